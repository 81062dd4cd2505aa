"""Debug: per-parameter difference between the native field backward and the torch backward on a golden case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import common, scenes
import nsff_pl_amd as A
from test_gpu_parity import _Replay, _to_dev, DEV
import nsff_pl_amd.rendering as R

name = sys.argv[1] if len(sys.argv) > 1 else "g7_nsff_train_noise"
A.set_precision("f16x3")
out = {}
for native in ("0", "1"):
    os.environ["NSFF_NATIVE_BACKWARD"] = native
    cfg, meta, rays, ts, models, emb, _, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    _to_dev(models, emb)
    draws = scenes.replay_draws(cfg, meta["draw_seed"])
    kw = scenes.render_kwargs(cfg)
    orig = (R.torch.rand, R.torch.randn)
    if cfg.get("perturb", 0) or cfg.get("noise_std", 0):
        rp = _Replay(cfg, draws)
        R.torch.rand, R.torch.randn = rp.rand, rp.randn
    res = common.render_rays_at(want["zs_fine"])(models, emb, rays.to(DEV), ts.to(DEV), scenes.N_FRAMES - 1, cfg["N_samples"],
                        cfg.get("perturb", 0), cfg.get("noise_std", 0), cfg["N_importance"], 1024 * 32,
                        test_time=False, **kw)
    R.torch.rand, R.torch.randn = orig
    scenes.cotangent_loss(res).backward()
    out[native] = {n: p.grad.detach().double().cpu() for n, p in scenes.named_grad_params(models, emb) if p.grad is not None}
for n in out["0"]:
    a, b = out["0"][n], out["1"][n]
    print(f"{n:45s} |g|1 {float(a.abs().sum()):10.3e}  sum torch {float(a.sum()):+.5e} native {float(b.sum()):+.5e}  "
          f"maxdiff/max {float((a - b).abs().max() / a.abs().max().clamp_min(1e-300)):.2e}")
