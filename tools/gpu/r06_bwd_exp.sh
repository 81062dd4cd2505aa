#!/bin/bash
# round 6: what bounds nsff_field_bwd_kernel_h3b -- timing of body variants that do less (results are garbage; only the time is
# read): built with  H3B_EXP=<exp> python tools/h3asm/gen_bwd.py  +  make variant NAME=b<exp> DEFS=-DH3B_BODY_FILE=...
#   usage: bash tools/gpu/r06_bwd_exp.sh <tag> [exp...]
TAG=${1:-a}; shift || true
EXPS=${*:-nostore nocopy noepi norefill nomfma}
O=gpurun_out/r06_$TAG; mkdir -p $O
for rnd in 1 2; do
  echo "== round $rnd: the product's body" >> $O/bwd_exp.txt
  python tools/debug/bwd_bench.py 196608 20 2>&1 | grep "field_backward" | sed 's/.*| field_backward/  field_backward/' >> $O/bwd_exp.txt
  for e in $EXPS; do
    echo "== round $rnd: $e" >> $O/bwd_exp.txt
    NSFF_LIB=nsff_pl_amd/libnsff_hip_b$e.so python tools/debug/bwd_bench.py 196608 20 2>&1 | grep "field_backward" | sed 's/.*| field_backward/  field_backward/' >> $O/bwd_exp.txt
  done
done
cat $O/bwd_exp.txt
