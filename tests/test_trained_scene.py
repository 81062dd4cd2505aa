"""A TRAINED synthetic scene (SURVEY.md 8f row N1 end to end): the native training step really fits a scene, and the parity
figures hold on trained weights, not only on the seeded "sharp" initialisations of the golden scenes.

A teacher field renders 1024 rays (exact fp32); a student with another seed and plain torch initialisation is trained on the
teacher's colours by NSFFTrainer -- HIP training forward, fused loss, native backward, native Adam (reference train.py:174-198).
The reference's real-data PSNR (kid-running, 35.02 dB) stays unpinned: dataset and checkpoint are not on the mount."""
import pytest
import torch

import scenes
import nsff_pl_amd as A

N_RAYS, STEPS = 1024, 300


def _render(models, emb, rays, ts, cfg, precision):
    A.set_precision(precision)
    try:
        with torch.no_grad():
            return A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"],
                                 1024 * 32, test_time=True, output_transient=True)
    finally:
        A.set_precision(A.config.DEFAULT_PRECISION)


def _psnr(a, b):
    return float(-10 * torch.log10(((a - b) ** 2).mean()))


@pytest.mark.gpu
def test_student_fits_the_teacher_and_precisions_agree_on_trained_weights(hip_lib):
    """Both backward arithmetics side by side: the default one-product backward and the three-product one
    (config.set_grad_precision) train the same student from the same seed on the same draws; final loss and PSNR are printed
    for both and must agree -- the fp16 backward's 1e-3 gradient noise does not show in what the scene learns."""
    from test_gpu_parity import _to_dev, DEV
    from nsff_pl_amd.training import NSFFTrainer
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=N_RAYS)
    rays, ts = scenes.synthetic_rays(N_RAYS, 77)
    rays, ts = rays.to(DEV), ts.to(DEV)
    teacher, emb_t = scenes.build_scene(A.NeRF, A.PosEmbedding, dict(cfg, seed=101))
    _to_dev(teacher, emb_t)
    target = _render(teacher, emb_t, rays, ts, cfg, "f32")["rgb_fine"]
    assert float(target.std()) > 0.05                                   # a scene, not a constant
    Ks, Ps, _ = scenes.camera_buffers()
    hp = dict(N_samples=cfg["N_samples"], N_importance=cfg["N_importance"], perturb=1.0, noise_std=0.0, lambda_geo_init=0.0)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    result = {}
    for mode in A.config.GRAD_PRECISIONS:
        A.config.set_grad_precision(mode)
        try:
            student, emb_s = scenes.build_scene(A.NeRF, A.PosEmbedding, dict(cfg, seed=202, gain=1.0))
            tr = NSFFTrainer(student, emb_s, scenes.N_FRAMES, hp, Ks, Ps, output_transient_flow=cfg["flow"]).to(DEV)
            tr.on_train_epoch_start(0)
            batch = {k: v.to(DEV) for k, v in scenes.synthetic_targets(N_RAYS, ts.cpu(), 9).items()}
            batch["rgbs"], batch["rays"], batch["ts"] = target.clone(), rays, ts
            before = _psnr(_render(student, emb_s, rays, ts, cfg, "f16x3")["rgb_fine"], target)
            torch.manual_seed(1234)                                      # (the stratified-sampling draws of the 300 steps)
            for _ in range(STEPS):
                log = tr.step(batch)
            loss = float(log["train/loss"])
            assert loss == loss and abs(loss) < float("inf")
        finally:
            A.config.set_grad_precision("f16")
        out = {p: _render(student, emb_s, rays, ts, cfg, p) for p in ("f32", "f16x3")}
        after = _psnr(out["f16x3"]["rgb_fine"], target)
        # measured: 18.3 dB untrained -> 33.9 dB after 300 steps (37-40 dB after 600)
        assert after > before + 10.0 and after > 28.0, (mode, before, after)
        for key in ("rgb_fine", "depth_fine"):
            assert rel(out["f16x3"][key], out["f32"][key]) <= 1e-4, key          # measured 7e-7 / 1e-6: the parity bar on trained weights
        assert abs(_psnr(out["f16x3"]["rgb_fine"], target) - _psnr(out["f32"]["rgb_fine"], target)) < 0.01
        result[mode] = (before, after, loss)
    print("\nbackward arithmetic   PSNR before -> after %d steps   final loss" % STEPS)
    for mode, (b, a_, l) in result.items():
        print(f"  {mode:6s}              {b:6.2f} -> {a_:6.2f} dB                {l:.6f}")
    # (two training runs diverge in their digits after a few steps whatever the arithmetic -- and a run does not repeat itself to
    #  the digit from process to process either; what they reach must not differ.  Measured, two processes: one product 18.29 ->
    #  33.10 / 33.25 dB, last step's loss 0.002716 / 0.003208; three products 18.29 -> 34.47 / 33.67 dB, 0.002709 / 0.002566 -- the
    #  PSNR and the loss of a single step wander by a dB / 20 % that late in such a run)
    assert abs(result["f16"][1] - result["f16x3"][1]) < 2.5, result
    assert abs(result["f16"][2] - result["f16x3"][2]) < 0.5 * result["f16x3"][2], result
