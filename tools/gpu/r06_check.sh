#!/bin/bash
# round 6: GPU suite + smoke + the default bench line (one gpurun call).  usage: bash tools/gpu/r06_check.sh <tag> [pytest args]
TAG=${1:-a}; shift || true
O=gpurun_out/r06_$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 3000 $O/bench.json
