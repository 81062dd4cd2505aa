"""Debug: binned far path vs device-scope atomics on crafted multi-tile frames of growing size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_interpolate as T
import nsff_pl_amd as A
import nsff_pl_amd.interpolation as I
dev = torch.device("cuda:0")
for shape in [(96, 40, 24), (256, 72, 24), (512, 72, 24), (512, 288, 24), (96, 40, 64), (512, 288, 64)]:
    res_t, res_tp1, dt, K, c2w, wh, _ = T.multi_tile_case(*shape)
    a = {k: torch.from_numpy(v).to(dev) for k, v in res_t.items()}
    b = {k: torch.from_numpy(v).to(dev) for k, v in res_tp1.items()}
    out = {}
    for binning in (True, False):
        I._FAR_BINNING = binning
        rgb, depth = A.interpolate(a, b, dt, K, c2w, wh)
        torch.cuda.synchronize()
        out[binning] = rgb.cpu().numpy()
    d = np.abs(out[True] - out[False]).max(-1)
    print(shape, "max diff", float(d.max()), "pixels > 1e-4:", int((d > 1e-4).sum()), "of", d.size, flush=True)
