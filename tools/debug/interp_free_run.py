import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import scenes, common
import nsff_pl_amd as A
from test_interpolate import _golden
from test_gpu_parity import _to_dev, DEV
res_t, res_tp1, K, c2w, wh, outs, rays, checksum = _golden()
cfg = dict(scenes.INTERP_CFG, n_rays=wh[0] * wh[1])
for prec in ("f32", "f16x3"):
    A.set_precision(prec)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    r = torch.from_numpy(rays).to(DEV)
    both = []
    for t in (scenes.INTERP_T, scenes.INTERP_T + 1):
        with torch.no_grad():
            both.append(A.render_rays(models, emb, r, torch.full((r.shape[0],), t, device=DEV), scenes.N_FRAMES - 1,
                                      cfg["N_samples"], 0, 0, cfg["N_importance"], 1024 * 32, test_time=True, **scenes.render_kwargs(cfg)))
    zs_err = np.abs(both[0]["zs_fine"].cpu().numpy() - res_t["zs_fine"])
    print(prec, "zs_fine max abs diff vs golden", zs_err.max(), "n>1e-4:", (zs_err > 1e-4).sum(), "of", zs_err.size)
    for dt, (rgb, depth) in outs.items():
        g_rgb, g_depth = A.interpolate(both[0], both[1], dt, K, c2w, wh)
        e = np.abs(g_rgb.cpu().numpy() - rgb).max(-1).ravel()
        print(prec, dt, "rgb err max %.2e p99 %.2e p90 %.2e median %.2e; n>1e-3: %d of %d" % (e.max(), np.percentile(e, 99), np.percentile(e, 90), np.median(e), (e > 1e-3).sum(), e.size))
