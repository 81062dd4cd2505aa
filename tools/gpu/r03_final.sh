#!/bin/bash
# round 3 evidence pass: whole GPU suite, smoke, the driver's bench line, rocprofv3 stats + counters (profiles/collect_r03.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-final}
O=gpurun_out/r03_$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest -m gpu rc=$?" > $O/summary.txt
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -30 >> $O/summary.txt
grep -E "^E  " $O/pytest.log | head -40 >> $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log >> $O/summary.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
bash profiles/collect_r03.sh $TAG stats pmc interp train > $O/collect.log 2>&1
python profiles/summarize_r03.py $O > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +6M -delete
du -sh $O >> $O/summary.txt
cat $O/summary.txt; head -c 1500 $O/bench_line.json
