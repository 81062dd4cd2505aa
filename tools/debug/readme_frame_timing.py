"""Time the reference's README configuration (512x288, 128 samples, view directions, flows, chunk 16384) resident on the GPU,
with the view-direction static trunk on the hand-scheduled kernel (default) and in the earlier two-launch form (NSFF_NO_SIDE_BIAS=1).
    python tools/debug/readme_frame_timing.py [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes                                   # noqa: E402
import nsff_pl_amd as A                         # noqa: E402
from nsff_pl_amd import _lib, evaluate         # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
cfg = dict(scenes.CASES["g6_readme_viewdir"], appearance=False)
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
for m in list(models.values()) + [emb["t"]]:
    m.to(dev)
H, W = 288, 512
K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]])
c2w = torch.tensor([[1.0, 0, 0, 0.02], [0, 1.0, 0, -0.01], [0, 0, 1.0, 0.0]])
rays = evaluate.frame_rays(K, c2w, H, W, device=dev)
ts = torch.full((H * W,), 8, device=dev, dtype=torch.long)
kw = dict(output_transient=True, output_transient_flow=['fw', 'bw'])


def frame():
    return evaluate.render_frame(models, emb, rays, ts, scenes.N_FRAMES - 1, 128, 0, 1024 * 16, keys=("rgb_fine",), **kw)


for rnd in range(2):
    for label, env in (("side rows (h3a for both trunks)", None), ("two launches (static on the eight-wave kernel)", "1")):
        if env:
            os.environ["NSFF_NO_SIDE_BIAS"] = env
        else:
            os.environ.pop("NSFF_NO_SIDE_BIAS", None)
        frame(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{label:50s} {dt * 1e3:8.2f} ms / frame  {H * W * 128 / dt / 1e6:7.1f} M ray-samples/s  last kernel {_lib.last_field_kernel()}")
