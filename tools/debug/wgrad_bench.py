"""Time the weight-gradient launches of ONE fine field node in isolation (HIP events; random operands): the job list of a
both-trunk C2 fine node -- 14 hidden-layer jobs (256 x 256), 2 input jobs (256 x 128), 2 head jobs (32 x 256).
    python tools/debug/wgrad_bench.py [n_points] [reps] [n_splits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from nsff_pl_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
tiles = P // 64
g = torch.Generator(device="cpu").manual_seed(1)
dpre = (torch.randn(18, tiles, 64 * 256, generator=g) * 30).to(torch.float16).to(dev)
acts = torch.rand(18, tiles, 64 * 256, generator=g).to(torch.float16).to(dev)
xin = torch.rand(tiles, 64 * 128, generator=g).to(torch.float16).to(dev)
dhead = torch.randn(2, tiles, 64 * 32, generator=g).to(torch.float16).to(dev)
gmax = torch.ones(16, device=dev)
jobs, off = [], 0
for t in range(2):
    base = 9 * t
    jobs.append([dpre[base].data_ptr(), xin.data_ptr(), 256, 128, 0, t])
    for l in range(1, 8):
        jobs.append([dpre[base + l].data_ptr(), acts[base + l - 1].data_ptr(), 256, 256, 0, t])
    jobs.append([dhead[t].data_ptr(), acts[base + 7].data_ptr(), 32, 256, 0, t])
for j in jobs:
    j[4] = off
    off += j[2] * j[3]
out = torch.empty(off, device=dev)
bias = torch.empty(len(jobs), 256, device=dev)
jt = [tuple(j) for j in jobs]


def run():
    _lib.weight_grad(jt, tiles, NS, out, bias, gmax)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / REPS * 1e3
byt = (14 * 2 * 100.7e6 + 2 * (100.7e6 + 50.3e6) + 2 * (100.7e6 + 12.6e6)) * P / 196608
print(f"weight_grad of a fine node ({len(jobs)} jobs, {P} points, n_splits {NS}): {us:7.1f} us   operands {byt / 1e9:.2f} GB -> {byt / us / 1e6:.2f} TB/s")
