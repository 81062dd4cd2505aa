#!/bin/bash
# call T: weight-gradient GEMMs forked onto the side stream inside the hipGraph capture as well?
cd $GRAFT_REPO_ROOT
for f in 0 1; do
  echo "== NSFF_WGRAD_FORK_IN_GRAPH=$f"
  NSFF_WGRAD_FORK_IN_GRAPH=$f timeout 600 python bench.py --workload train --graph --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
timeout 600 python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
