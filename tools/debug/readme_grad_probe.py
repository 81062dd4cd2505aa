"""Debug: |g|_1 of a few static-trunk gradients of the README training configuration under the kernel switches."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes, common
import nsff_pl_amd as A
from nsff_pl_amd import _lib
from nsff_pl_amd.losses import NeRFWLoss
DEV = torch.device("cuda:0")
z = np.load(common.GOLDEN_DIR + "/g20_loss_readme_train_512.npz")
s64 = json.loads(bytes(z["stats64"]).decode())
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = dict(scenes.README_TRAIN_CASE, n_rays=n_rays)
names = ["fine.static_xyz_encoding_1.0.weight", "fine.static_xyz_encoding_8.0.weight", "fine.static_dir_encoding.0.weight", "fine.static_sigma.weight",
         "fine.transient_xyz_encoding_1.0.weight", "fine.static_rgb.0.weight"]
for env in ({}, {"NSFF_NO_SIDE_BIAS": "1"}, {"NSFF_BWD_KERNEL": "c"}, {"NSFF_NO_SIDE_BIAS": "1", "NSFF_BWD_KERNEL": "c"}):
    for k in ("NSFF_NO_SIDE_BIAS", "NSFF_BWD_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rays, ts = scenes.synthetic_rays(cfg["n_rays"], cfg["seed"])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    models = {"fine": models["fine"].to(DEV)}
    for e in emb.values():
        e.to(DEV)
    kw = scenes.render_kwargs(cfg)
    res = A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, 0, 1024 * 32, test_time=False, **kw)
    loss_fn = NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0)
    Ks, Ps, max_t = scenes.camera_buffers()
    loss_fn.register_buffer("Ks", Ks); loss_fn.register_buffer("Ps", Ps); loss_fn.max_t = max_t
    loss_fn.to(DEV)
    targets = {k: v.to(DEV) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
    terms = loss_fn(res, targets, epoch=scenes.LOSS_EPOCH, **kw)
    sum(terms.values()).backward()
    torch.cuda.synchronize()
    stats, _ = scenes.grad_stats(models, emb)
    print(env, _lib.last_field_kernel(), _lib.last_bwd_kernel())
    for n in names:
        print(f"   {n:42s} |g|_1 {stats[n][1]:10.4f}   reference fp64 {s64[n][1]:10.4f}" if n_rays == 512 else f"   {n:42s} |g|_1 {stats[n][1]:10.4f}")
