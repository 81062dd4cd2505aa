#!/bin/bash
# call L: where does the training forward's extra time go?  (stores compiled out / copy compiled out / masks off / both)
cd $GRAFT_REPO_ROOT
for v in "" _xnostore _xnocopy _xnomask _xnone; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep -E "static=|Error|error" 
done
