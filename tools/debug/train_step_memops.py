"""Which torch ops of one eager C2 training step launch memcpy / memset / small elementwise kernels (debugging aid).
    python tools/debug/train_step_memops.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B
from torch.profiler import profile, ProfilerActivity

b = B.Bench(0, 1, torch.device("cuda:0"))
b.to_device()
step = b.train_step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type.name == "CUDA" or not e.kernels:
        continue
    for k in e.kernels:
        n = k.name
        if "Memcpy" in n or "Memset" in n or "copyBuffer" in n or "fillBuffer" in n or "elementwise" in n or "at::native" in n:
            rows.append((e.name, n[:60], k.duration, tuple(map(tuple, e.input_shapes)) if e.input_shapes else (), [s for s in (e.stack or []) if "nsff_pl_amd" in s or "bench.py" in s][:2]))
from collections import Counter
c = Counter((r[0], r[1], str(r[3])[:60], str(r[4])[:160]) for r in rows)
for (op, kern, shp, st), n in sorted(c.items(), key=lambda kv: -kv[1]):
    print(n, op, "|", kern, "|", shp, "|", st)
