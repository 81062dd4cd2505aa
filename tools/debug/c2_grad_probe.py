"""Debug / evidence: gradient statistics of the C2 training configuration at 512 rays (golden g21) with the package of the tree given
as argv[1] (default: this tree; e.g. base_r05 = a `git archive` of the round-5 commit, built) -- |g|_1 of a few tensors against the
reference's fp64 value."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TREE = os.path.join(ROOT, sys.argv[1]) if len(sys.argv) > 1 else ROOT
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, TREE)
import numpy as np, torch
import scenes, common
import nsff_pl_amd as A
from nsff_pl_amd.losses import NeRFWLoss
print("package:", os.path.dirname(A.__file__))
DEV = torch.device("cuda:0")
z = np.load(common.GOLDEN_DIR + "/g21_loss_c2_train_512.npz")
s64 = json.loads(bytes(z["stats64"]).decode())
cfg = scenes.C2_TRAIN_CASE
rays, ts = scenes.synthetic_rays(cfg["n_rays"], cfg["seed"])
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
for m in list(models.values()) + list(emb.values()):
    m.to(DEV)
kw = scenes.render_kwargs(cfg)
res = common.render_rays_at(z["zs_fine"])(models, emb, rays.to(DEV), ts.to(DEV), scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"],
                                          1024 * 32, test_time=False, **kw)
loss_fn = NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0)
Ks, Ps, max_t = scenes.camera_buffers()
loss_fn.register_buffer("Ks", Ks); loss_fn.register_buffer("Ps", Ps); loss_fn.max_t = max_t
loss_fn.to(DEV)
targets = {k: v.to(DEV) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
sum(loss_fn(res, targets, epoch=scenes.LOSS_EPOCH, **kw).values()).backward()
torch.cuda.synchronize()
stats, _ = scenes.grad_stats(models, emb)
worst = max(((abs(stats[n][1] - s64[n][1]) / max(s64[n][1], 1e-30), n) for n in s64))
for n in ["fine.static_xyz_encoding_1.0.weight", "fine.static_rgb.0.weight", "fine.static_sigma.weight", "coarse.static_xyz_encoding_4.0.weight",
          "fine.transient_sigma.weight", "fine.transient_rgb.0.weight", "fine.transient_xyz_encoding_1.0.weight", "fine.transient_flow_fw.0.weight"]:
    print(f"   {n:42s} |g|_1 {stats[n][1]:14.5f}   reference fp64 {s64[n][1]:14.5f}   ratio {stats[n][1] / s64[n][1]:.4f}")
print(f"   worst |g|_1 ratio deviation over all {len(s64)} tensors: {worst[0]:.3f} ({worst[1]})")
