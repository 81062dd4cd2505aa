"""Measured max-norm relative error of every result key of the scenes whose chained re-query keys are held to more than 1e-4
(tests/common.py::key_rtol), at the reference's fine depths, per arithmetic / kernel.  Prints a table to paste into tests/common.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import common, parity, scenes
import nsff_pl_amd as A
from nsff_pl_amd import config

DEV = "cuda:0"
for name in ("g3b_nsff_train_gain3", "g18_wide_inputs_train", "g3_nsff_train", "g19_c2_subset"):
    cfg, meta, rays, ts, models, emb, dataset, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    for m in list(models.values()) + [emb[k] for k in ("t", "a") if k in emb]:
        m.to(DEV)
    kw = scenes.render_kwargs(cfg, dataset)
    worst = {}
    for prec, tile in (("f32", 0), ("f16x3", 0), ("f16x3", 130), ("f16x3", 131)):
        for grad in (True, False):
            config.set_precision(prec); config.set_tile_points(tile)
            with torch.set_grad_enabled(grad):
                out = common.render_rays_at(want["zs_fine"])(models, emb, rays.to(DEV), ts.to(DEV), scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"],
                                    1024 * 32, test_time=cfg["test_time"], **kw)
            for k in want:
                if k in ("static_zs_fine", "transient_zs_fine"):
                    continue
                e = parity.max_rel_err(out[k].detach().cpu().numpy(), want[k])
                worst[k] = max(worst.get(k, 0.0), e)
    config.set_precision("f16x3"); config.set_tile_points(0)
    big = {k: v for k, v in worst.items() if v > 5e-5}
    print(name, "keys above 5e-5 (worst over f32 / f16x3 kernels, saving and inference launches):")
    for k, v in sorted(big.items(), key=lambda kv: -kv[1]):
        print(f"    {k:22s} {v:.2e}")
