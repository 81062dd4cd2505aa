#!/bin/bash
# round 3, call I: whole GPU suite (training-limit lift, fused loss options)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest -m gpu rc=$?" >> $O/summary.txt
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -30 >> $O/summary.txt
grep -E "^E  " $O/pytest.log | head -40 >> $O/summary.txt
cat $O/summary.txt
