"""Debug: gradient of <output key, random cotangent> w.r.t. a static-trunk weight -- the build's backward graph against autograd of the
torch restatement (tests/torch_path.py) on the same device, for the README training configuration (N_importance = 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes, torch_path
import nsff_pl_amd as A
DEV = torch.device("cuda:0")
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(scenes.README_TRAIN_CASE, n_rays=n_rays)
rays, ts = scenes.synthetic_rays(cfg["n_rays"], cfg["seed"])
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
models = {"fine": models["fine"].to(DEV)}
for e in emb.values():
    e.to(DEV)
kw = scenes.render_kwargs(cfg)
rays, ts = rays.to(DEV), ts.to(DEV)
probe = [("static W1", models["fine"].static_xyz_encoding_1[0].weight), ("static W8", models["fine"].static_xyz_encoding_8[0].weight),
         ("dir W", models["fine"].static_dir_encoding[0].weight), ("dyn W1", models["fine"].transient_xyz_encoding_1[0].weight)]
res = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, 0, 1024 * 32, test_time=False, **kw)
zs = res["zs_fine"].detach()
rec = dict(N_importance=0, noise_std=0.0, output_transient=True, flows=list(cfg["flow"]), zs_coarse=zs, zs_fine=zs, view_dir=rays[:, 3:6],
           t_embedded_override=None, a_embedded_override=None)
ref = torch_path.recompute(models, emb, rays, ts, scenes.N_FRAMES - 1, rec)
g = torch.Generator().manual_seed(1)
print(f"{'key':28s} " + "  ".join(f"{n:>22s}" for n, _ in probe) + "      (|g|_1 build / torch restatement)")
for k in sorted(res):
    if not res[k].requires_grad:
        continue
    cot = torch.randn(res[k].shape, generator=g).to(DEV)
    row = []
    for which, r in (("build", res), ("torch", ref)):
        for _, p in probe:
            p.grad = None
        if k not in r or not r[k].requires_grad:
            row.append([float("nan")] * len(probe)); continue
        (r[k] * cot).sum().backward(retain_graph=True)
        torch.cuda.synchronize()
        row.append([0.0 if p.grad is None else float(p.grad.abs().sum()) for _, p in probe])
    bad = any(abs(a - b) > 0.02 * max(abs(b), 1e-9) for a, b in zip(*row))
    print(f"{k:28s} " + "  ".join(f"{a:10.4g}/{b:10.4g}" for a, b in zip(*row)) + ("   <-- differs" if bad else ""))

# ---- the whole objective: NeRFWLoss on the build's dict against NeRFWLoss on the torch restatement's dict
from nsff_pl_amd.losses import NeRFWLoss
for fused in ("1", "0"):
    os.environ["NSFF_FUSED_LOSS"] = fused
    row = []
    for which, r in (("build", res), ("torch", ref)):
        loss_fn = NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0)
        Ks, Ps, max_t = scenes.camera_buffers()
        loss_fn.register_buffer("Ks", Ks); loss_fn.register_buffer("Ps", Ps); loss_fn.max_t = max_t
        loss_fn.to(DEV)
        targets = {k: v.to(DEV) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts.cpu(), cfg["seed"]).items()}
        for _, p in probe:
            p.grad = None
        terms = loss_fn(r, targets, epoch=scenes.LOSS_EPOCH, **kw)
        sum(terms.values()).backward(retain_graph=True)
        torch.cuda.synchronize()
        row.append([float(p.grad.abs().sum()) for _, p in probe])
    print(f"NeRFWLoss total (fused={fused})    " + "  ".join(f"{a:10.4g}/{b:10.4g}" for a, b in zip(*row)))
