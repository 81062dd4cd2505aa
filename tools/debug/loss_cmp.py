import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes, parity
import nsff_pl_amd as A
from nsff_pl_amd import fused_loss
from nsff_pl_amd.losses import NeRFWLoss
DEV = "cuda:0"
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(scenes.CASES["g7_nsff_train_noise"], n_rays=n_rays)
rays, ts = scenes.synthetic_rays(n_rays, 21)
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
for m in list(models.values()) + [emb["t"]]:
    m.to(DEV)
kw = scenes.render_kwargs(cfg)
torch.manual_seed(3)
with torch.no_grad():
    res = A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), scenes.N_FRAMES - 1, 64, 1.0, 1.0, 64, 32768, test_time=False, **kw)
loss_fn = NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0, static_shapes=True)
Ks, Ps, max_t = scenes.camera_buffers()
loss_fn.register_buffer("Ks", Ks); loss_fn.register_buffer("Ps", Ps); loss_fn.max_t = max_t
loss_fn.to(DEV)
targets = {k: v.to(DEV) for k, v in scenes.synthetic_targets(n_rays, ts, 5).items()}
out = {}
only = sys.argv[2] if len(sys.argv) > 2 else None
for fused in ("1", "0"):
    os.environ["NSFF_FUSED_LOSS"] = fused
    leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.startswith(("zs_", "disocc")) and k != "xyzs_fine") for k, v in res.items()}
    terms = loss_fn(leaves, targets, epoch=3, **kw)
    (terms[only] if only else sum(terms.values())).backward()
    out[fused] = {k: v.grad.detach().cpu().numpy() for k, v in leaves.items() if v.grad is not None}
for k in sorted(out["0"]):
    a, b = out["1"][k], out["0"][k]
    err = np.abs(a - b)
    i = np.unravel_index(err.argmax(), err.shape)
    print(f"{k:26s} rel {err.max() / max(np.abs(b).max(), 1e-30):.3e}  at {i}: fused {a[i]:.6e} torch {b[i]:.6e}  max|torch| {np.abs(b).max():.3e}")
    if err.max() / max(np.abs(b).max(), 1e-30) > 1e-3 and a.ndim == 3:
        n, s, c = i
        z = res["zs_fine"][n].cpu().numpy()
        print("     zs around:", z[max(s - 2, 0):s + 3], " n_keep", int(192 * 0.95), " fused row", a[n, s], " torch row", b[n, s])
        print("     xyzs_fine", res["xyzs_fine"][n, s].cpu().numpy(), "xyzs_bw", res["xyzs_bw"][n, s].cpu().numpy(), "xyzs_fw", res["xyzs_fw"][n, s].cpu().numpy())
