#!/bin/bash
# call P: flow glue + time-code gathers as native nodes: gradient suites, then the training step
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_field_grad.py tests/test_gradients.py tests/test_losses.py tests/test_optim.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -12
timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 600 python bench.py --workload train --graph --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
