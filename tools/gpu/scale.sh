#!/bin/bash
# The scaling table of one node in ONE command (for whoever gets an 8-GPU MI355X node: the build pool has single-GPU boxes only).
#   bash tools/gpu/scale.sh [out_dir]        (run from the repository root; ~6 minutes)
# For N in 1 2 4 8 and workload in render (C2, weak: 1024 rays per rank) / eval (C3, strong: one 512x288 frame sharded) / train (C4,
# weak: data-parallel step, one flat gradient all-reduce): bench.py's JSON line -> value, ms per step, per-rank min / max step time,
# gather (all-reduce) ms; then the A/B legs of the two multi-GPU choices that were reasoned on one GPU and never measured on eight:
#   NSFF_GATHER_ASYNC=1 / NSFF_GATHER_SYNC=1   pixel all-gather on a side stream (overlapped) / on the render stream
#   NSFF_PERSIST_MULTI=1                       keep persistent field launches beside a side-stream collective
# Expected: render weak >= 7.8x at N = 8 (per step +36 us of gather + wire on 1.95 ms); eval strong ~7.3-7.6x
# (18 432 rays per rank: tile quantisation of the shards' launches, non-persistent form -1.5 %); train weak ~7.7x (one 9.2 MB all-reduce).
OUT=${1:-gpurun_out/scale}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
run() { # tag, n, extra env..., -- bench args
  local tag=$1 n=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  if [ "$n" -gt "$NG" ]; then echo "$tag n=$n: skipped ($NG GPUs visible)"; return; fi
  if [ "$n" -eq 1 ]; then env "${envs[@]}" python bench.py --gpus 1 --no-cpu-baseline --no-aux "$@" > $OUT/$tag.n$n.json 2> $OUT/$tag.n$n.err
  else env "${envs[@]}" python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
         bench.py --gpus $n --no-cpu-baseline --no-aux "$@" > $OUT/$tag.n$n.json 2> $OUT/$tag.n$n.err; fi
  python - "$OUT/$tag.n$n.json" "$tag" "$n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pr = d.get("per_rank", {})
    print(f"{sys.argv[2]:28s} N={sys.argv[3]}  value {d['value']:.4g} {d['unit']:18s} ms/step {d['ms_per_step']:.3f}  "
          f"per-rank ms min {pr.get('ms_per_step_min', float('nan')):.3f} max {pr.get('ms_per_step_max', float('nan')):.3f}  "
          f"gather ms/step min {pr.get('gather_ms_per_step_min', float('nan')):.3f} max {pr.get('gather_ms_per_step_max', float('nan')):.3f}  launch form: {d['config'].get('field_launch', '-')}")
except Exception as e:
    print(f"{sys.argv[2]:28s} N={sys.argv[3]}  FAILED ({e}): see {sys.argv[1].replace('.json', '.err')}")
PY
}
for n in 1 2 4 8; do
  run render $n -- --steps 100 --warmup 10
  run eval $n -- --workload eval --steps 10 --warmup 2
  run train $n -- --workload train --steps 30 --warmup 5
done
for n in 2 8; do
  run render_gather_async $n NSFF_GATHER_ASYNC=1 -- --steps 100 --warmup 10
  run eval_gather_sync $n NSFF_GATHER_SYNC=1 -- --workload eval --steps 10 --warmup 2
  run eval_persist_multi $n NSFF_PERSIST_MULTI=1 -- --workload eval --steps 10 --warmup 2
done
echo "scaling efficiency = value(N) / (N x value(1)) for render / train (weak), value(N) / value(1) / N for eval (strong); lines above are in $OUT/"
