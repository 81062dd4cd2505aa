"""Debug: HIP interpolate vs the oracle on crafted multi-tile frames of growing size; where do they differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_interpolate as T
import nsff_pl_amd as A
from oracle import nsff_oracle as orc
dev = torch.device("cuda:0")
for shape in [(256, 72, 24), (512, 72, 24), (96, 40, 64), (512, 288, 24), (256, 144, 64)]:
    res_t, res_tp1, dt, K, c2w, wh, (sx, sy) = T.multi_tile_case(*shape)
    with np.errstate(all="ignore"):
        o_rgb, o_depth = orc.interpolate(res_t, res_tp1, dt, K, c2w, wh)
    rgb, depth = A.interpolate({k: torch.from_numpy(v).to(dev) for k, v in res_t.items()},
                               {k: torch.from_numpy(v).to(dev) for k, v in res_tp1.items()}, dt, K, c2w, wh)
    g = rgb.cpu().numpy()
    d = np.abs(g - o_rgb).max(-1)
    bad = np.argwhere(d > 1e-4 * np.abs(o_rgb).max())
    print(shape, "max diff", float(d.max()), "bad pixels", len(bad), "of", d.size, flush=True)
    for (y, x) in bad[:8]:
        print("   pixel y,x", int(y), int(x), "got", g[y, x], "want", o_rgb[y, x], flush=True)
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("   y range", int(ys.min()), int(ys.max()), "x range", int(xs.min()), int(xs.max()),
              "x mod 32 hist", np.bincount(xs % 32, minlength=32).tolist(), "y mod 8 hist", np.bincount(ys % 8, minlength=8).tolist(), flush=True)
