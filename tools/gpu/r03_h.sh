#!/bin/bash
# round 3, call H: whole GPU suite after the training-limit lift + head prefetch; A/B vs the base library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest -m gpu rc=$?" >> $O/summary.txt
tail -12 $O/pytest.log >> $O/summary.txt
for rep in 1 2; do
  for tile in 0; do
    echo "== bench_field main tile=$tile" >> $O/summary.txt
    timeout 300 python tools/bench_field.py --tile-points $tile --iters 20 2>&1 | grep "C2 mix" >> $O/summary.txt
  done
done
echo "== bench.py (full line)" >> $O/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> $O/summary.txt
import json
try:
    d = json.loads([l for l in open("gpurun_out/r03_h/bench_line.json") if l.startswith("{")][0])
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "launch ms", d["roofline"]["avg_launch_ms"])
    print(json.dumps(d.get("aux"), indent=0)[:3000])
except Exception as e:
    print("bench parse failed", e)
PY
cat $O/summary.txt
