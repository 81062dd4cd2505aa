"""Debug / evidence: a TRAINED synthetic scene.  A teacher field (seeded "sharp" init) renders 1024 rays; a student with another
seed and plain torch init is trained on the teacher's colours with NSFFTrainer (native forward / loss / backward / Adam).
Prints the PSNR trajectory and, on the trained weights, parity-grade f16x3 vs exact f32.
    python tools/debug/teacher_student.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd.training import NSFFTrainer

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
DEV = torch.device("cuda:0")
N = 1024
base = dict(scenes.CASES["g3_nsff_train"], n_rays=N)


def to_dev(models, emb):
    for m in models.values():
        m.to(DEV)
    for k in emb:
        if isinstance(emb[k], torch.nn.Module):
            emb[k].to(DEV)


def render(models, emb, rays, ts, prec):
    A.set_precision(prec)
    try:
        with torch.no_grad():
            return A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, base["N_samples"], 0, 0, base["N_importance"],
                                 1024 * 32, test_time=True, output_transient=True)
    finally:
        A.set_precision(A.config.DEFAULT_PRECISION)


def psnr(a, b):
    return float(-10 * torch.log10(((a - b) ** 2).mean()))


def main():
    rays, ts = scenes.synthetic_rays(N, 77)
    rays, ts = rays.to(DEV), ts.to(DEV)
    teacher, emb_t = scenes.build_scene(A.NeRF, A.PosEmbedding, dict(base, seed=101))
    to_dev(teacher, emb_t)
    target = render(teacher, emb_t, rays, ts, "f32")["rgb_fine"]
    student, emb_s = scenes.build_scene(A.NeRF, A.PosEmbedding, dict(base, seed=202, gain=1.0))
    Ks, Ps, _ = scenes.camera_buffers()
    hp = dict(N_samples=base["N_samples"], N_importance=base["N_importance"], perturb=1.0, noise_std=0.0, lambda_geo_init=0.0)
    tr = NSFFTrainer(student, emb_s, scenes.N_FRAMES, hp, Ks, Ps, output_transient_flow=base["flow"]).to(DEV)
    tr.on_train_epoch_start(0)
    batch = {k: v.to(DEV) for k, v in scenes.synthetic_targets(N, ts.cpu(), 9).items()}
    batch["rgbs"], batch["rays"], batch["ts"] = target.clone(), rays, ts
    print(f"target colour std {float(target.std()):.3f}; PSNR of the untrained student {psnr(render(student, emb_s, rays, ts, 'f16x3')['rgb_fine'], target):.2f} dB", flush=True)
    for i in range(STEPS):
        log = tr.step(batch)
        if i % 50 == 0 or i == STEPS - 1:
            print(f"step {i:4d}  loss {float(log['train/loss']):.5f}  train/psnr {float(log['train/psnr']):.2f} dB", flush=True)
    out = {p: render(student, emb_s, rays, ts, p) for p in ("f32", "f16x3")}
    for p, o in out.items():
        print(f"trained student, {p:6s}: PSNR vs teacher {psnr(o['rgb_fine'], target):.3f} dB", flush=True)
    for p in ("f16x3",):
        d = (out[p]["rgb_fine"] - out["f32"]["rgb_fine"]).abs().max() / out["f32"]["rgb_fine"].abs().max()
        dd = (out[p]["depth_fine"] - out["f32"]["depth_fine"]).abs().max() / out["f32"]["depth_fine"].abs().max()
        print(f"trained student, {p:6s} vs f32: rgb_fine max-norm rel {float(d):.2e}, depth_fine {float(dd):.2e}", flush=True)


if __name__ == "__main__":
    main()
