#!/bin/bash
# round 5: GPU suite + smoke + the default bench line (one gpurun call).  usage: bash tools/gpu/r05_check.sh <tag> [pytest args]
TAG=${1:-a}; shift || true
mkdir -p gpurun_out/r05_$TAG
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/r05_$TAG/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_$TAG/pytest.log
tail -5 gpurun_out/r05_$TAG/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_$TAG/smoke.log 2>&1; echo "smoke rc $?"
python bench.py > gpurun_out/r05_$TAG/bench.json 2> gpurun_out/r05_$TAG/bench.err; echo "bench rc $?"
tail -c 3000 gpurun_out/r05_$TAG/bench.json
