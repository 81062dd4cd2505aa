"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE.

Build-container only: imports kwea123/nsff_pl from /root/reference (read-only mount,
absent on the GPU box) on CPU torch and records inputs + every output key of
``render_rays`` / ``NeRF.forward`` / ``PosEmbedding`` / ``sample_pdf`` for the seeded
scenes of ``tests/scenes.py``.  Only data is written (inputs, expected outputs, a weight
checksum); no reference source travels.

    python tests/golden/make_golden.py                      # rewrites tests/golden/*.npz
    python tests/golden/make_golden.py --only g19_c2_subset # (re)writes the named render cases only

Import recipe: models/rendering.py pulls kornia, cupy (via
models/softsplat) and the datasets package (cv2, torchvision) at import time although the
render path needs none of them; three stub modules are inserted before the import.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def import_reference():
    sys.path.insert(0, REF)
    kornia = types.ModuleType("kornia")

    def create_meshgrid(H, W, normalized_coordinates=False, device=None):   # functional stand-in (ray-gen / interpolate goldens)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        return torch.stack([xs, ys], -1)[None]
    kornia.create_meshgrid = create_meshgrid
    sys.modules["kornia"] = kornia
    kfilters = types.ModuleType("kornia.filters")

    def filter2d(x, kernel, border_type="constant"):             # only the default thickness=1 (1x1x1 box) is pinned
        assert kernel.numel() == 1 and float(kernel) == 1.0
        return x
    kfilters.filter2d = filter2d
    kornia.filters = kfilters
    sys.modules["kornia.filters"] = kfilters
    datasets = types.ModuleType("datasets")
    datasets.__path__ = [REF + "/datasets"]
    sys.modules["datasets"] = datasets
    import models  # noqa: F401  (reference package)
    softsplat = types.ModuleType("models.softsplat")
    softsplat.FunctionSoftsplat = None
    sys.modules["models.softsplat"] = softsplat
    from models.nerf import NeRF, PosEmbedding
    from models.rendering import render_rays, sample_pdf
    return NeRF, PosEmbedding, render_rays, sample_pdf


def to_np(d):
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in d.items()}


def make_g20(scenes, NeRF, PosEmbedding, render_rays, ref_losses):
    # ---- the README training configuration at batch size 512 (g20): statistics only, fp32 and fp64 ----
    torch.set_grad_enabled(True)
    cfg = scenes.README_TRAIN_CASE
    rays, ts = scenes.synthetic_rays(cfg["n_rays"], cfg["seed"])
    save = {}
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        models, embeddings = scenes.build_scene(NeRF, PosEmbedding, cfg)
        for m in list(models.values()) + [embeddings["t"]]:
            m.to(dt)
        kw = scenes.render_kwargs(cfg)
        loss_fn = ref_losses.NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0)
        Ks, Ps, max_t = scenes.camera_buffers()
        loss_fn.register_buffer("Ks", Ks.to(dt)); loss_fn.register_buffer("Ps", Ps.to(dt)); loss_fn.max_t = max_t
        targets = {k: (v.to(dt) if v.is_floating_point() else v)
                   for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
        res = render_rays({"fine": models["fine"]}, embeddings, rays.to(dt), ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, 0,
                          1024 * 32, test_time=False, **kw)
        ld = loss_fn(res, targets, epoch=scenes.LOSS_EPOCH, **kw)
        total = sum(ld.values())
        total.backward()
        stats, _ = scenes.grad_stats({"fine": models["fine"]}, embeddings)
        save["terms" + tag] = np.frombuffer(json.dumps({k: float(v) for k, v in ld.items()}).encode(), dtype=np.uint8)
        save["stats" + tag] = np.frombuffer(json.dumps(stats).encode(), dtype=np.uint8)
        print(f"g20 README training configuration fp{tag}: total {float(total):.6f}  terms {len(ld)}  {len(stats)} parameter tensors")
    np.savez_compressed(os.path.join(HERE, "g20_loss_readme_train_512.npz"), **save)
    torch.set_grad_enabled(False)



def make_g21(scenes, NeRF, PosEmbedding, render_rays, ref_losses):
    # ---- the C2 training configuration at 512 rays (g21): statistics + the fine depths, fp32 and fp64 ----
    torch.set_grad_enabled(True)
    import models.rendering as R
    cfg = scenes.C2_TRAIN_CASE
    rays, ts = scenes.synthetic_rays(cfg["n_rays"], cfg["seed"])
    save = {}
    zs32 = None
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        models, embeddings = scenes.build_scene(NeRF, PosEmbedding, cfg)
        for m in list(models.values()) + [embeddings["t"]]:
            m.to(dt)
        kw = scenes.render_kwargs(cfg)
        loss_fn = ref_losses.NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0)
        Ks, Ps, max_t = scenes.camera_buffers()
        loss_fn.register_buffer("Ks", Ks.to(dt)); loss_fn.register_buffer("Ps", Ps.to(dt)); loss_fn.max_t = max_t
        targets = {k: (v.to(dt) if v.is_floating_point() else v)
                   for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
        orig_sort = R.torch.sort
        if dt == torch.float64:             # the fp64 run at the fp32 run's depths (sampling is not differentiated)
            R.torch.sort = lambda *a, **k_: (zs32.double(), None)
        try:
            res = render_rays(models, embeddings, rays.to(dt), ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"],
                              1024 * 32, test_time=False, **kw)
        finally:
            R.torch.sort = orig_sort
        if dt == torch.float32:
            zs32 = res["zs_fine"].detach().clone()
            save["zs_fine"] = zs32.numpy()
        ld = loss_fn(res, targets, epoch=scenes.LOSS_EPOCH, **kw)
        total = sum(ld.values())
        total.backward()
        stats, _ = scenes.grad_stats(models, embeddings)
        save["terms" + tag] = np.frombuffer(json.dumps({k: float(v) for k, v in ld.items()}).encode(), dtype=np.uint8)
        save["stats" + tag] = np.frombuffer(json.dumps(stats).encode(), dtype=np.uint8)
        print(f"g21 C2 training configuration fp{tag}: total {float(total):.6f}  terms {len(ld)}  {len(stats)} parameter tensors")
    np.savez_compressed(os.path.join(HERE, "g21_loss_c2_train_512.npz"), **save)
    torch.set_grad_enabled(False)


def main():
    import scenes
    from oracle import nsff_oracle as orc
    NeRF, PosEmbedding, render_rays, sample_pdf = import_reference()
    torch.set_grad_enabled(False)
    DRAW_SEED = 4242

    if "--g20" in sys.argv or "--g21" in sys.argv:      # only the training-configuration statistics (python tests/golden/make_golden.py --g20 | --g21)
        sys.path.insert(0, REF)
        import losses as ref_losses
        if "--g20" in sys.argv:
            make_g20(scenes, NeRF, PosEmbedding, render_rays, ref_losses)
        if "--g21" in sys.argv:
            make_g21(scenes, NeRF, PosEmbedding, render_rays, ref_losses)
        return
    only = None
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1].split(",")
    worst = 0.0
    for name in scenes.CASES:
        if only is not None and name not in only:
            continue
        cfg, rays, ts = scenes.case_inputs(name)
        models, embeddings = scenes.build_scene(NeRF, PosEmbedding, cfg)
        dataset = scenes.DatasetStub(cfg["seed"]) if cfg.get("dataset") else None
        kw = scenes.render_kwargs(cfg, dataset)
        torch.manual_seed(DRAW_SEED)
        res = render_rays(models, embeddings, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"],
                          cfg.get("perturb", 0), cfg.get("noise_std", 0), cfg["N_importance"],
                          1024 * 32, test_time=cfg["test_time"], **kw)
        res = to_np(res)
        meta = dict(case=name, cfg={k: v for k, v in cfg.items()}, draw_seed=DRAW_SEED,
                    weight_checksum=scenes.weight_checksum(models, embeddings),
                    torch=torch.__version__, keys=sorted(res))
        save = {"out/" + k: v for k, v in res.items()}
        save["in/rays"] = rays.numpy()
        if ts is not None:
            save["in/ts"] = ts.numpy()
        save["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **save)

        # cross-check the oracle against the reference right here
        fields = {k: orc.field_from_module(m) for k, m in models.items()}
        draws = scenes.replay_draws(cfg, DRAW_SEED)
        o = orc.render_rays(
            fields, embeddings["xyz"].freqs.numpy(), embeddings["dir"].freqs.numpy(), rays.numpy(),
            None if ts is None else ts.numpy(), scenes.N_FRAMES - 1,
            emb_t=embeddings["t"].weight.numpy() if "t" in embeddings else None,
            emb_a=embeddings["a"].weight.numpy() if "a" in embeddings else None,
            N_samples=cfg["N_samples"], perturb=cfg.get("perturb", 0), noise_std=cfg.get("noise_std", 0),
            N_importance=cfg["N_importance"], test_time=cfg["test_time"],
            z_lin=torch.linspace(0, 1, cfg["N_samples"]).numpy(),
            u_lin=torch.linspace(0, 1, max(cfg["N_importance"], 1)).numpy(), draws=draws,
            output_transient_flow=cfg["flow"], dataset=dataset.as_oracle_dict() if dataset else None)
        assert sorted(o) == sorted(res), (sorted(set(o) ^ set(res)))
        case_worst = 0.0
        for k in res:
            assert o[k].shape == res[k].shape, (k, o[k].shape, res[k].shape)
            err = float(np.abs(o[k] - res[k]).max() / (np.abs(res[k]).max() + 1e-12))
            case_worst = max(case_worst, err)
        worst = max(worst, case_worst)
        size = os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024
        print(f"{name:28s} keys={len(res):2d} oracle-vs-reference max-norm rel err {case_worst:.2e}  ({size:.0f} KiB)")

    if only is not None:
        return
    # ---- gradient goldens: d<outputs, fixed cotangents>/d(parameters) from the reference ----
    torch.set_grad_enabled(True)
    for name in scenes.GRAD_CASES:
        cfg, rays, ts = scenes.case_inputs(name)
        models, embeddings = scenes.build_scene(NeRF, PosEmbedding, cfg)
        kw = scenes.render_kwargs(cfg)
        torch.manual_seed(DRAW_SEED)
        res = render_rays(models, embeddings, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"],
                          cfg.get("perturb", 0), cfg.get("noise_std", 0), cfg["N_importance"],
                          1024 * 32, test_time=False, **kw)
        loss = scenes.cotangent_loss(res)
        loss.backward()
        stats, full = scenes.grad_stats(models, embeddings)
        save = {"full/" + k: v for k, v in full.items()}
        save["stats"] = np.frombuffer(json.dumps(stats).encode(), dtype=np.uint8)
        save["loss"] = np.float64(float(loss))
        if not (cfg.get("perturb", 0) or cfg.get("noise_std", 0)):
            # the same gradient from the reference evaluated in float64 (the fp32 one scatters by up to a few
            # 1e-3 of |g|_1 around it, machine to machine: sin(512 x) under the warped re-queries)
            models64, embeddings64 = scenes.build_scene(NeRF, PosEmbedding, cfg)
            for m in list(models64.values()) + [e for k_, e in embeddings64.items() if k_ in ("t", "a")]:
                m.double()
            # zs_fine must be the fp32 run's (sampling is not differentiated): feed it through a patched sort
            import models.rendering as R
            zs32 = res["zs_fine"].detach().double()
            orig_sort = R.torch.sort
            R.torch.sort = lambda *a, **k_: (zs32, None)
            try:
                res64 = render_rays(models64, embeddings64, rays.double(), ts, scenes.N_FRAMES - 1,
                                    cfg["N_samples"], 0, 0, cfg["N_importance"], 1024 * 32, test_time=False, **kw)
            finally:
                R.torch.sort = orig_sort
            scenes.cotangent_loss(res64).backward()
            stats64, _ = scenes.grad_stats(models64, embeddings64)
            save["stats64"] = np.frombuffer(json.dumps(stats64).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, "g9_grads_" + name + ".npz"), **save)
        print(f"g9 {name}: loss {float(loss):.6f}, {len(stats)} parameter tensors")
    torch.set_grad_enabled(False)

    # ---- loss goldens (row N1): reference NeRFWLoss on the reference render, values + parameter gradients ----
    torch.set_grad_enabled(True)
    sys.path.insert(0, REF)
    import losses as ref_losses
    for name in scenes.LOSS_CASES:
        cfg, rays, ts = scenes.case_inputs(name)
        save = {}
        for tag, dt in (("32", torch.float32), ("64", torch.float64)):
            if dt == torch.float64 and (cfg.get("perturb", 0) or cfg.get("noise_std", 0)):
                continue
            models, embeddings = scenes.build_scene(NeRF, PosEmbedding, cfg)
            for m in list(models.values()) + [embeddings["t"]]:
                m.to(dt)
            kw = scenes.render_kwargs(cfg)
            loss_fn = ref_losses.NeRFWLoss(lambda_geo=0.04, thickness=1, topk=1.0)
            Ks, Ps, max_t = scenes.camera_buffers()
            loss_fn.register_buffer("Ks", Ks.to(dt)); loss_fn.register_buffer("Ps", Ps.to(dt)); loss_fn.max_t = max_t
            targets = {k: (v.to(dt) if v.is_floating_point() else v)
                       for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
            torch.manual_seed(DRAW_SEED)
            if dt == torch.float64:
                import models.rendering as R
                zs32 = torch.from_numpy(np.load(os.path.join(HERE, name + ".npz"))["out/zs_fine"]).double()
                orig_sort = R.torch.sort
                R.torch.sort = lambda *a, **k_: (zs32, None)
            try:
                res = render_rays(models, embeddings, rays.to(dt), ts, scenes.N_FRAMES - 1, cfg["N_samples"],
                                  cfg.get("perturb", 0), cfg.get("noise_std", 0), cfg["N_importance"],
                                  1024 * 32, test_time=False, **kw)
            finally:
                if dt == torch.float64:
                    R.torch.sort = orig_sort
            ld = loss_fn(res, targets, epoch=scenes.LOSS_EPOCH, **kw)
            total = sum(ld.values())
            total.backward()
            stats, _ = scenes.grad_stats(models, embeddings)
            save["terms" + tag] = np.frombuffer(json.dumps({k: float(v) for k, v in ld.items()}).encode(), dtype=np.uint8)
            save["stats" + tag] = np.frombuffer(json.dumps(stats).encode(), dtype=np.uint8)
            print(f"loss golden {name} fp{tag}: total {float(total):.6f}  terms {len(ld)}")
        np.savez_compressed(os.path.join(HERE, "g10_loss_" + name + ".npz"), **save)
    torch.set_grad_enabled(False)

    make_g20(scenes, NeRF, PosEmbedding, render_rays, ref_losses)
    make_g21(scenes, NeRF, PosEmbedding, render_rays, ref_losses)

    # ---- stage goldens ----
    g = torch.Generator().manual_seed(77)
    stage = {}
    x = torch.cat([torch.rand(96, 3, generator=g) * 2.6 - 1.3, torch.tensor([[0., 0., 0.], [1.2, -1.2, 1.0]])], 0)
    stage["posenc/x"] = x.numpy()
    stage["posenc/xyz_9_10"] = PosEmbedding(9, 10)(x).numpy()
    stage["posenc/dir_3_4"] = PosEmbedding(3, 4)(x).numpy()

    cfg = dict(seed=11, transient=True, appearance=True, viewdir=True, flow=['fw', 'bw'], N_importance=64, gain=2.5)
    models, _ = scenes.build_scene(NeRF, PosEmbedding, cfg)
    fine, coarse = models["fine"], models["coarse"]
    B = 70   # not a multiple of the 64-point tile
    xin = torch.randn(B, 63 + 27 + scenes.N_A + scenes.N_TAU, generator=g) * 0.7
    stage["nerf/x_full"] = xin.numpy()
    stage["nerf/weight_checksum"] = np.float64(scenes.weight_checksum(models, {}))
    x_sig_t = torch.cat([xin[:, :63], xin[:, -scenes.N_TAU:]], 1)
    x_static = xin[:, :63 + 27 + scenes.N_A]
    x_coarse = torch.cat([xin[:, :63 + 27], xin[:, -scenes.N_TAU:]], 1)   # coarse: no appearance code
    stage["nerf/fine_sigma_only_static"] = fine(xin[:, :63], sigma_only=True, output_transient=False).numpy()
    stage["nerf/fine_sigma_only_both"] = fine(x_sig_t, sigma_only=True, output_transient=True).numpy()
    stage["nerf/fine_static"] = fine(x_static, output_transient=False).numpy()
    stage["nerf/fine_both_noflow"] = fine(xin, output_transient=True, output_transient_flow=[]).numpy()
    stage["nerf/fine_both_flow"] = fine(xin, output_transient=True, output_transient_flow=['fw', 'bw', 'disocc']).numpy()
    stage["nerf/fine_transient_bw"] = fine(xin, output_static=False, output_transient=True, output_transient_flow=['bw']).numpy()
    stage["nerf/fine_transient_fw"] = fine(xin, output_static=False, output_transient=True, output_transient_flow=['fw']).numpy()
    stage["nerf/coarse_both_noflow"] = coarse(x_coarse, output_transient=True).numpy()
    stage["nerf/coarse_sigma_only_both"] = coarse(x_sig_t, sigma_only=True, output_transient=True).numpy()

    n, m = 37, 62
    bins = torch.sort(torch.rand(n, m + 1, generator=g), 1)[0]
    w = torch.rand(n, m, generator=g) ** 4
    w[3] = 0                       # all-zero weights
    w[4, :] = 0; w[4, 17] = 1.0    # a single spike: every other bin has pdf < eps
    stage["pdf/bins"], stage["pdf/weights"] = bins.numpy(), w.numpy()
    stage["pdf/det_64"] = sample_pdf(bins, w, 64, det=True).numpy()
    torch.manual_seed(99)
    stage["pdf/rand_40"] = sample_pdf(bins, w, 40, det=False).numpy()
    torch.manual_seed(99)
    stage["pdf/u_40"] = torch.rand(n, 40).numpy()
    # ---- N3: frame ray generation (datasets/ray_utils.py as called by monocular.py:268-276) ----
    from datasets import ray_utils
    H, W = 18, 32
    K = torch.tensor([[25.0, 0, W / 2], [0, 24.0, H / 2 + 0.5], [0, 0, 1]])
    ang = 0.3
    c2w = torch.tensor([[np.cos(ang), 0.05, np.sin(ang), 0.12], [0.02, 0.999, -0.03, -0.07],
                        [-np.sin(ang), 0.03, np.cos(ang), -1.3]], dtype=torch.float32)
    dirs = ray_utils.get_ray_directions(H, W, K)
    ro, rd = ray_utils.get_rays(dirs, c2w)
    shift_near = -min(-1.0, c2w[2, 3])
    ro, rd = ray_utils.get_ndc_rays(K, 1.0, shift_near, ro, rd)
    stage["rays/K"], stage["rays/c2w"], stage["rays/HW"] = K.numpy(), c2w.numpy(), np.array([H, W])
    stage["rays/ndc"] = torch.cat([ro, rd], 1).numpy()
    assert np.abs(orc.frame_rays(K.numpy(), c2w.numpy(), H, W) - stage["rays/ndc"]).max() < 1e-5
    np.savez_compressed(os.path.join(HERE, "g8_stages.npz"), **stage)

    # ---- N2: interpolate (rendering.py:365-460) on that frame.  The reference's splat is a cupy/CUDA kernel that
    # cannot run here, so FunctionSoftsplat is stood in by the oracle's restatement of it: this golden pins the
    # projection / optical-flow / MPI-compositing code of the reference, NOT the splat (known-answer tests do). ----
    import models.rendering as R

    def splat_stub(tenInput, tenFlow, tenMetric, strType):
        assert tenMetric is None and strType == "average" and tenInput.shape[0] == 1
        return torch.from_numpy(orc.softsplat_average(tenInput[0].numpy(), tenFlow[0].numpy()))[None]
    R.FunctionSoftsplat = splat_stub
    cfg = dict(scenes.INTERP_CFG, n_rays=H * W)
    models, embeddings = scenes.build_scene(NeRF, PosEmbedding, cfg)
    frame_rays_ = torch.from_numpy(stage["rays/ndc"])
    both = []
    for t in (scenes.INTERP_T, scenes.INTERP_T + 1):
        both.append(render_rays(models, embeddings, frame_rays_, torch.full((H * W,), t), scenes.N_FRAMES - 1,
                                cfg["N_samples"], 0, 0, cfg["N_importance"], 1024 * 32, test_time=True,
                                **scenes.render_kwargs(cfg)))
    g11 = {"weight_checksum": np.array(scenes.weight_checksum(models, embeddings))}
    for k in scenes.INTERP_KEYS_T:
        g11["t/" + k] = both[0][k].numpy()
    for k in scenes.INTERP_KEYS_TP1:
        g11["tp1/" + k] = both[1][k].numpy()
    cuda_orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for dt in scenes.INTERP_DTS:
            img, dep = R.interpolate(both[0], both[1], dt, K, c2w, (W, H))
            g11[f"out/rgb_{dt}"], g11[f"out/depth_{dt}"] = img.numpy(), dep.numpy()
            o_img, o_dep = orc.interpolate(to_np(both[0]), to_np(both[1]), dt, K.numpy(), c2w.numpy(), (W, H))
            print(f"interpolate dt={dt}: oracle vs reference(+splat stub) rgb {np.abs(o_img - img.numpy()).max():.2e} "
                  f"depth {np.abs(o_dep - dep.numpy()).max():.2e}; rgb range {img.min():.3f}..{img.max():.3f}")
    finally:
        torch.Tensor.cuda = cuda_orig
    np.savez_compressed(os.path.join(HERE, "g11_interpolate.npz"), **g11)
    print("g8_stages written;", "oracle worst", f"{worst:.2e}")


if __name__ == "__main__":
    main()
