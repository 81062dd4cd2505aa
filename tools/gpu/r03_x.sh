#!/bin/bash
# call X: split-K factor of the weight-gradient GEMMs (fewer partials to reduce vs balance)
cd $GRAFT_REPO_ROOT
for sp in 32 16 24 48; do
  echo "== NSFF_WGRAD_SPLITS=$sp"
  NSFF_WGRAD_SPLITS=$sp timeout 600 python bench.py --workload train --graph --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c150-260
done
