import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import scenes, common
import nsff_pl_amd as A
from nsff_pl_amd.training import NSFFTrainer
from nsff_pl_amd import _lib
DEV = torch.device("cuda:0")
A.config.set_grad_precision("f16x3")
name = "g3_nsff_train"
out = {}
for graph in (False, True):
    cfg, meta, rays, ts, models, emb, _, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    Ks, Ps, _ = scenes.camera_buffers()
    hp = dict(N_samples=cfg["N_samples"], N_importance=cfg["N_importance"], perturb=0, noise_std=0)
    tr = NSFFTrainer(models, emb, scenes.N_FRAMES, hp, Ks, Ps, output_transient_flow=cfg["flow"], graph=graph).to(DEV)
    tr.on_train_epoch_start(scenes.LOSS_EPOCH)
    batch = {k: v.to(DEV) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
    batch["rays"] = rays.to(DEV)
    logs = [float(tr.step(batch)["train/loss"]) for _ in range(6)]
    torch.cuda.synchronize()
    out[graph] = logs
    print("graph" if graph else "eager", _lib.last_bwd_kernel(), ["%.6f" % l for l in logs])
assert abs(out[True][0] - out[False][0]) < 1e-5 * abs(out[False][0]) and abs(out[True][-1] - out[False][-1]) < 0.05 * abs(out[False][-1])
print("ok")
