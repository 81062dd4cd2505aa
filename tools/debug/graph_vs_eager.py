"""Where the hipGraph replay of a C2 render_rays call spends its time against the eager call:
eager and replayed calls back to back, and the replay floor of a one-node graph.
    python tools/debug/graph_vs_eager.py [steps]
Round-5 measurement (one box): eager 1.932 ms, replay 1.946 ms per call; a one-node graph replays in 9.8 us where an eager launch of
the same kernel costs 4.4 us -- the replay's fixed cost is what the eager path, whose host runs 1.3 ms ahead of the GPU, never
pays.  (Tried with it: the nine generator kernels on a side stream, joined before their consumers -- eager 2.069 ms: every
cross-stream join costs ~35 us on this stack; a replay is indifferent to it, 1.966 ms.  Not kept.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes                                   # noqa: E402
import nsff_pl_amd as A                         # noqa: E402
from nsff_pl_amd.graphs import GraphedRender    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=1024)
rays, ts = scenes.synthetic_rays(1024, 42)
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
for m in list(models.values()) + [emb["t"]]:
    m.to(dev)
rays, ts, kw = rays.to(dev), ts.to(dev), scenes.render_kwargs(cfg)


def eager():
    with torch.no_grad():
        return A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, 64, 1.0, 1.0, 64, test_time=False, **kw)


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    time.sleep(0.3)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g = GraphedRender(models, emb, scenes.N_FRAMES - 1, 64, 1.0, 1.0, 64, test_time=False, **kw)
g(rays, ts)
for rnd in range(2):
    print(f"eager {timed(eager, steps):.4f} ms / call   graph replay {timed(lambda: g(rays, ts), steps):.4f} ms / call")
# the floor of a replay: a graph of one trivial kernel
x = torch.zeros(64, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    x.add_(1)
torch.cuda.current_stream().wait_stream(s)
one = torch.cuda.CUDAGraph()
with torch.cuda.graph(one):
    x.add_(1)
print(f"one-node graph: {timed(one.replay, 2000) * 1e3:.2f} us / replay;  eager one kernel: {timed(lambda: x.add_(1), 2000) * 1e3:.2f} us")
