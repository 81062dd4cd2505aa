"""Parity of the HIP path (through the C-ABI) against the reference goldens and the oracle.

All tests here need the MI355X (`-m gpu`).  They never read /root/reference: expected
values come from tests/golden/*.npz (generated from the reference in the build container)
and from oracle/nsff_oracle.py run on the same seeded inputs.
"""
import os

import numpy as np
import pytest
import torch

import common
import parity
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib
from oracle import nsff_oracle as orc
from test_oracle_golden import nerf_mode_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["f32", "f16x3", "f16x3-130", "f16x3-131"])
def precision(request):
    """Every test runs on the two parity-grade arithmetics: exact fp32 MFMA and the fp16-split kernels -- the latter in
    every form the library has: what it picks by launch size ("f16x3": 64-point tiles for the small golden scenes, 128-point
    tiles for the full-size tests), 128-point tiles forced ("f16x3-130": the HAND-SCHEDULED body nsff_field_kernel_h3a
    wherever a launch's trunks qualify -- every scene without a view-direction branch -- else the compiler-scheduled
    eight-wave kernel) and the compiler-scheduled eight-wave kernel forced ("f16x3-131")."""
    from nsff_pl_amd import config
    name, _, tile = request.param.partition("-")
    config.set_precision(name)
    config.set_tile_points(int(tile) if tile else 0)
    yield name if not tile else request.param
    config.set_precision(config.DEFAULT_PRECISION)
    config.set_tile_points(0)


def _to_dev(models, emb):
    for m in models.values():
        m.to(DEV)
    for k in ("t", "a"):
        if k in emb:
            emb[k].to(DEV)


def _np(d):
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


class _Replay:
    """Feeds render_rays the recorded torch draws (moved to the GPU) in order."""

    def __init__(self, cfg, draws):
        self.queue = [(key, kind, torch.from_numpy(draws[key])) for key, _, kind in scenes.draw_plan(cfg)]

    def _pop(self, kind, shape):
        key, k, t = self.queue.pop(0)
        assert k == kind and tuple(t.shape) == tuple(shape), (key, k, kind, t.shape, shape)
        return t.to(DEV)

    def rand(self, *shape, **kw):
        return self._pop("rand", shape)

    def randn(self, *shape, **kw):
        return self._pop("randn", shape)


def _render(cfg, models, emb, rays, ts, dataset, monkeypatch, draws=None, zs_fine=None):
    kw = scenes.render_kwargs(cfg, dataset)
    replay = None
    if draws is not None:
        replay = _Replay(cfg, draws)
        import nsff_pl_amd.rendering as R
        monkeypatch.setattr(R.torch, "rand", replay.rand)
        monkeypatch.setattr(R.torch, "randn", replay.randn)
    try:
        out = common.render_rays_at(zs_fine)(models, emb, rays.to(DEV), None if ts is None else ts.to(DEV),
                            scenes.N_FRAMES - 1, cfg["N_samples"], cfg.get("perturb", 0),
                            cfg.get("noise_std", 0), cfg["N_importance"], 1024 * 32,
                            test_time=cfg["test_time"], **kw)
    finally:
        monkeypatch.undo()
    if replay is not None:
        assert not replay.queue, "render_rays drew fewer random tensors than the reference"
    return _np(out)


@pytest.mark.parametrize("name", list(scenes.CASES))
def test_render_rays_matches_reference_goldens(name, hip_lib, monkeypatch):
    cfg, meta, rays, ts, models, emb, dataset, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    _to_dev(models, emb)
    draws = scenes.replay_draws(cfg, meta["draw_seed"])
    use_draws = draws if (cfg.get("perturb", 0) or cfg.get("noise_std", 0)) else None

    got = _render(cfg, models, emb, rays, ts, dataset, monkeypatch, use_draws)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape and got[k].dtype == np.float32, k
    first_pass = [k for k in want if k.endswith("_coarse")] if cfg["N_importance"] > 0 else list(want)
    for k in first_pass:
        parity.assert_close(k, got[k], want[k], common.key_rtol(k, cfg))

    if cfg["N_importance"] > 0:
        u_lin = torch.linspace(0, 1, cfg["N_importance"]).numpy()
        u_s, u_t = draws.get("u_static", u_lin), draws.get("u_transient", u_lin)
        tol_s, tol_t = common.fine_sample_tolerances(cfg, want, u_s, u_t)
        if "static_zs_fine" in want:
            parity.assert_samples_close("static_zs_fine", got["static_zs_fine"], want["static_zs_fine"], tol_s)
        if "transient_zs_fine" in want:
            parity.assert_samples_close("transient_zs_fine", got["transient_zs_fine"],
                                        want["transient_zs_fine"], tol_t)
        tol = tol_s.max(1, keepdims=True)
        if tol_t is not None:
            tol = np.maximum(tol, tol_t.max(1, keepdims=True))
        parity.assert_samples_close("zs_fine", got["zs_fine"], want["zs_fine"], tol)
        assert (np.diff(got["zs_fine"], axis=1) >= 0).all(), "zs_fine not sorted"
        # merge + sort (rendering.py:346-348), bit for bit and independent of the sampling's conditioning: the merged depths
        # are exactly the kernel's own coarse depths and inverse-CDF samples, sorted -- on every ray, near-empty ones included
        parts = [got["zs_coarse"]] + [got[k] for k in ("static_zs_fine", "transient_zs_fine") if k in got and k in want]
        if len(parts) > 1 and sum(p.shape[1] for p in parts) == got["zs_fine"].shape[1]:
            assert np.array_equal(got["zs_fine"], np.sort(np.concatenate(parts, 1), 1)), "zs_fine is not sort(cat(coarse, samples))"

        got = _render(cfg, models, emb, rays, ts, dataset, monkeypatch, use_draws, zs_fine=want["zs_fine"])
        for k in want:
            if k in ("static_zs_fine", "transient_zs_fine"):
                continue
            parity.assert_close(k, got[k], want[k], common.key_rtol(k, cfg))


# Per-ray keys north_star names, FREE-RUNNING (no fine-depth override): the whole chain coarse field ->
# compositing -> inverse-CDF sampling -> merge/sort -> fine field -> compositing against the reference's own
# outputs (models/rendering.py:335-362).  On the well-conditioned scenes a 1e-6 depth shift stays small in the
# per-ray expectations; the gain-3 stress scene is asserted at 3e-3 (twice the measured worst), not at 1e-4.
FREE_RUN_KEYS = ("rgb_fine", "depth_fine", "transient_flow_fw", "transient_flow_bw", "xyz_fw", "xyz_bw",
                 "_static_rgb_fine", "rgb_coarse", "depth_coarse")
FREE_RUN_STRICT = ("g2_static_c2f", "g3_nsff_train", "g4_nsff_test", "g5_nsff_test_vis", "g7_nsff_train_noise", "g13_viewdir_train",
                   "g7b_static_noise_odd", "g12_other_arch")
# gain 3: sigma up to 33, weights near 0/1 -- a 1e-6 depth shift moves per-ray values by 1e-3 (the numpy oracle itself
# sits 3e-4..1.5e-3 from the reference there); measured 3e-4 .. 1.6e-3 on the MI355X, asserted at twice the worst: 3e-3
FREE_RUN_REPORTED = {"g3b_nsff_train_gain3": 3e-3}


@pytest.mark.parametrize("name", FREE_RUN_STRICT + tuple(FREE_RUN_REPORTED))
def test_free_running_per_ray_keys_match_reference(name, hip_lib, monkeypatch, precision, record_property):
    cfg, meta, rays, ts, models, emb, dataset, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    _to_dev(models, emb)
    draws = scenes.replay_draws(cfg, meta["draw_seed"])
    use_draws = draws if (cfg.get("perturb", 0) or cfg.get("noise_std", 0)) else None
    got = _render(cfg, models, emb, rays, ts, dataset, monkeypatch, use_draws)      # zs_fine=None: nothing overridden
    import nsff_pl_amd.rendering as R
    assert not hasattr(R, "_FINE_DEPTHS_OVERRIDE")        # the seam is a per-call keyword now: no module state
    rtol = parity.RTOL if name in FREE_RUN_STRICT else FREE_RUN_REPORTED[name]
    errs = {}
    for k in FREE_RUN_KEYS:
        if k in want:
            errs[k] = parity.max_rel_err(got[k], want[k])
    worst = max(errs, key=errs.get)
    print(f"free-run {name} [{precision}]: worst {worst} {errs[worst]:.2e}  " +
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    record_property("free_run_worst", f"{worst}={errs[worst]:.3e}")
    record_property("free_run_errors", " ".join(f"{k}={v:.3e}" for k, v in errs.items()))
    for k, e in errs.items():
        assert np.isfinite(got[k]).all() and e <= rtol, f"{name} {k}: free-running max-norm rel err {e:.3e} > {rtol:g}"


@pytest.fixture(scope="module")
def stages():
    return np.load(common.GOLDEN_DIR + "/g8_stages.npz")


def test_pos_embedding(stages, hip_lib):
    x = torch.from_numpy(stages["posenc/x"]).to(DEV)
    for key, (ms, nf) in {"xyz_9_10": (9, 10), "dir_3_4": (3, 4)}.items():
        got = A.PosEmbedding(ms, nf)(x).cpu().numpy()
        parity.assert_close(key, got, stages["posenc/" + key], 2e-6)
    # edge cases: empty batch, large magnitudes (args up to ~1e4 rad)
    assert A.PosEmbedding(9, 10)(torch.empty(0, 3, device=DEV)).shape == (0, 63)
    big = (torch.rand(257, 3, device=DEV) - 0.5) * 40
    want = orc.pos_embedding(big.cpu().numpy(), A.PosEmbedding(9, 10).freqs.numpy())
    parity.assert_close("big", A.PosEmbedding(9, 10)(big).cpu().numpy(), want, 2e-6)


def test_nerf_forward_modes(stages, hip_lib):
    cfg = dict(seed=11, transient=True, appearance=True, viewdir=True, flow=['fw', 'bw'],
               N_importance=64, gain=2.5)
    models, _ = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    assert abs(scenes.weight_checksum(models, {}) - float(stages["nerf/weight_checksum"])) < 1e-6
    _to_dev(models, {})
    for key, typ, x, kw in nerf_mode_inputs(stages):
        got = models[typ](torch.from_numpy(np.ascontiguousarray(x)).to(DEV), **kw).cpu().numpy()
        parity.assert_close(key, got, stages["nerf/" + key], parity.RTOL)


def test_nerf_forward_rejects_cpu_and_bad_shapes(hip_lib):
    m = A.NeRF("fine", use_viewdir=False)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 63 + 27))                       # CPU tensor: no fallback
    m.to(DEV)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 10, device=DEV), output_transient=False)
    assert m(torch.zeros(0, 63 + 27, device=DEV), output_transient=False).shape == (0, 4)


def test_skip_lists_render_and_train(hip_lib):
    """Several skip layers / none (nerf.py:34-40,163-167) run through the inference kernels (goldens g14 / g15) and through
    the training kernels (gradient goldens g16 / g17, tests/test_gradients.py): a differentiated call carries a graph."""
    for name in ("g14_two_skips", "g15_no_skip"):
        cfg = dict(scenes.CASES[name], test_time=False)
        rays, ts = scenes.synthetic_rays(cfg["n_rays"], cfg["seed"])
        models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
        _to_dev(models, emb)
        res = A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0,
                            cfg["N_importance"], 32768, test_time=False, **scenes.render_kwargs(cfg))
        assert res["rgb_fine"].requires_grad
        res["rgb_fine"].sum().backward()
        torch.cuda.synchronize()
        g = models["fine"].transient_xyz_encoding_1[0].weight.grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0


def test_weight_repack_follows_parameter_updates(hip_lib):
    torch.manual_seed(3)
    m = A.NeRF("fine", use_viewdir=False).to(DEV)
    x = torch.randn(130, 63 + 27, device=DEV)
    y0 = m(x, output_transient=False).clone()
    with torch.no_grad():
        m.static_rgb[0].bias.add_(0.5)                   # bumps the version counter
    y1 = m(x, output_transient=False)
    assert (y1[:, :3] - y0[:, :3]).abs().max() > 1e-3 and torch.equal(y1[:, 3], y0[:, 3])
    f = orc.field_from_module(m)
    parity.assert_close("after update", y1.cpu().numpy(),
                        orc.nerf_forward(f["params"], f["cfg"], x.cpu().numpy(), output_transient=False))


@pytest.mark.parametrize("logscale,n_freqs", [(True, 10), (False, 10), (True, 4)])
def test_field_query_encodes_raw_points_for_any_frequency_list(logscale, n_freqs, hip_lib):
    """Raw-position input of nsff_field_query (rendering.py:153-175 = PosEmbedding + cat + NeRF.forward): octave
    frequencies take the angle-doubling encoder of the f16x3 kernel, any other list (PosEmbedding(logscale=False),
    nerf.py:9-12) the generic one; both against the oracle, on a ragged point count."""
    from nsff_pl_amd import _lib
    emb = A.PosEmbedding(n_freqs - 1, n_freqs, logscale=logscale)
    torch.manual_seed(5)
    m = A.NeRF("fine", in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True, output_flow=True).to(DEV)
    S, n_rays = 50, 7                                       # 350 points: five full 64-point tiles + 30
    P = S * n_rays
    g = torch.Generator().manual_seed(2)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
    t_rows = torch.randn(n_rays, m.in_channels_t, generator=g).to(DEV)
    raw = torch.empty(P, _lib.RAW_STRIDE, device=DEV)
    freqs = [float(f) for f in emb.freqs]
    _lib.field_query(m, raw, P, S, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows)
    f = orc.field_from_module(m)
    x_emb = np.concatenate([orc.pos_embedding(xyz.cpu().numpy(), np.asarray(freqs, np.float32)),
                            np.repeat(t_rows.cpu().numpy(), S, 0)], 1)
    cfg = dict(f["cfg"], in_dir=0, in_a=0)
    want = orc.nerf_forward(f["params"], cfg, x_emb, output_transient=True, output_transient_flow=("fw", "bw"))
    got = raw.cpu().numpy()
    got = np.concatenate([got[:, 0:4], got[:, 4:8], got[:, 8:14]], 1)
    parity.assert_close(f"raw records logscale={logscale} n_freqs={n_freqs}", got, want, parity.RTOL)


def test_neighbour_time_rows_equal_the_embedding_gathers(hip_lib):
    """nsff_time_rows vs embedding_t(clamp(ts +- 1)) (rendering.py:218,224): bit-equal, both ends of the clamp, both widths
    of store (16-byte and 4-byte), an empty batch."""
    from nsff_pl_amd import _lib
    g = torch.Generator().manual_seed(4)
    for width in (48, 18):
        table = torch.randn(30, width, generator=g).to(DEV)
        ts = torch.cat([torch.tensor([0, 0, 29, 29, 28, 1]), torch.randint(0, 30, (500,), generator=g)]).to(DEV)
        nxt, prv = _lib.time_rows(table, ts, 29)
        assert torch.equal(nxt, table[torch.clamp(ts + 1, max=29)]) and torch.equal(prv, table[torch.clamp(ts - 1, min=0)])
        nxt, prv = _lib.time_rows(table, ts, 12)                    # max_t below the table size, as in a cropped sequence
        assert torch.equal(nxt, table[torch.clamp(ts + 1, max=12)])
    e, _ = _lib.time_rows(table, ts[:0], 29, want_prev=False)
    assert e.shape == (0, 18) and _ is None


def test_sample_pdf(stages, hip_lib, monkeypatch):
    bins, w = stages["pdf/bins"], stages["pdf/weights"]
    tb, tw = torch.from_numpy(bins).to(DEV), torch.from_numpy(w).to(DEV)
    u_det = torch.linspace(0, 1, 64).numpy()
    got = A.sample_pdf(tb, tw, 64, det=True).cpu().numpy()
    parity.assert_samples_close("det", got, stages["pdf/det_64"], parity.sample_tolerance(bins, w, u_det))
    u = stages["pdf/u_40"]
    import nsff_pl_amd.rendering as R
    monkeypatch.setattr(R.torch, "rand", lambda *s, **k: torch.from_numpy(u).to(DEV))
    got = A.sample_pdf(tb, tw, 40, det=False).cpu().numpy()
    monkeypatch.undo()
    parity.assert_samples_close("rand", got, stages["pdf/rand_40"], parity.sample_tolerance(bins, w, u))
    parity.assert_samples_close("rand-vs-oracle", got, orc.sample_pdf(bins, w, u),
                                parity.sample_tolerance(bins, w, u))


def test_rng_draw_order_matches_reference_on_device(hip_lib):
    """Same seed -> render_rays consumes the device generator exactly like the reference would."""
    cfg = dict(scenes.CASES["g7_nsff_train_noise"], n_rays=8)
    rays, ts = scenes.synthetic_rays(8, 7)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    torch.manual_seed(123)
    A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), 29, 64, 1.0, 1.0, 64, 32768, test_time=False,
                  **scenes.render_kwargs(cfg))
    after = torch.rand(4, device=DEV)
    torch.manual_seed(123)
    for _, shape, kind in scenes.draw_plan(cfg):
        (torch.rand if kind == "rand" else torch.randn)(*shape, device=DEV)
    assert torch.equal(after, torch.rand(4, device=DEV))


def test_fused_draws_are_torchs(hip_lib):
    """nsff_rng_draws (one launch for all draws of a call) against torch.rand / torch.randn themselves: the same bits from the same
    generator state, the same state afterwards -- for sizes below and above one grid-stride pass of torch's launch geometry, odd
    sizes, empty draws, and draws whose values are skipped (they only advance the generator)."""
    plan = [("rand", (1024, 64)), ("randn", (1024, 64)), ("randn", (1024, 64)), ("rand", (1024, 64)), ("randn", (1024, 192)),
            ("randn", (7, 33)), ("rand", (1, 1)), ("randn", (0, 64)), ("rand", (3001, 997)), ("randn", (2100, 1111)), ("rand", (5,)),
            ("randn", (1024, 192)), ("randn", (1024, 192)), ("rand", (255,)), ("randn", (257,))]
    for seed in (0, 123456789):
        torch.manual_seed(seed)
        torch.rand(3, device=DEV)                                   # (an offset that is not zero)
        want = [(torch.rand if k == "rand" else torch.randn)(*shape, device=DEV) for k, shape in plan]
        after = torch.rand(4, device=DEV)
        torch.manual_seed(seed)
        torch.rand(3, device=DEV)
        got = _lib.fused_draws(plan, torch.device(DEV))
        assert torch.equal(after, torch.rand(4, device=DEV)), "the generator is not where the separate calls leave it"
        for (k, shape), w, g in zip(plan, want, got):
            assert g.shape == w.shape and g.dtype == torch.float32
            assert torch.equal(g, w), (seed, k, shape, (g != w).float().mean().item())
        # skipped values: None in their place, the others unchanged, the same state afterwards
        torch.manual_seed(seed)
        torch.rand(3, device=DEV)
        need = [i % 3 != 1 for i in range(len(plan))]
        part = _lib.fused_draws(plan, torch.device(DEV), need)
        assert torch.equal(after, torch.rand(4, device=DEV))
        for i, (w, g) in enumerate(zip(want, part)):
            assert (g is None) == (not need[i] and w.numel() > 0)
            if g is not None:
                assert torch.equal(g, w)


def test_render_with_fused_draws_equals_render_with_torch_draws(hip_lib, monkeypatch):
    """the C2 training call (perturb = noise_std = 1, fw / bw warps: nine draws) with its draws made by one launch and by the nine
    torch calls (NSFF_TORCH_RNG=1): every output bit-identical, the generator in the same state; noise_std = 0: the skipped
    draws leave the same state too"""
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=256)
    rays, ts = scenes.synthetic_rays(256, 5)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    for noise_std in (1.0, 0.0):
        got = {}
        for form in ("fused", "torch"):
            if form == "torch":
                monkeypatch.setenv("NSFF_TORCH_RNG", "1")
            else:
                monkeypatch.delenv("NSFF_TORCH_RNG", raising=False)
            torch.manual_seed(77)
            with torch.no_grad():
                out = A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), 29, 64, 1.0, noise_std, 64, 32768, test_time=False,
                                    **scenes.render_kwargs(cfg))
            got[form] = ({k: v.clone() for k, v in out.items()}, torch.rand(4, device=DEV))
        monkeypatch.delenv("NSFF_TORCH_RNG", raising=False)
        (a, sa), (b, sb) = got["fused"], got["torch"]
        assert torch.equal(sa, sb)
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (noise_std, k)


def test_chunk_parallel_composite_equals_the_serial_form(hip_lib, monkeypatch):
    """65..256 samples per ray: one ray per workgroup, one wave per 64-sample chunk (composite_kernel_chunks) against the one-wave-
    per-ray form (NSFF_SERIAL_COMPOSITE=1): every per-sample output bit-identical (a chunk's transmittance starts at its
    predecessors' products, multiplied in the serial form's order), the per-ray sums equal up to their summation order"""
    for name, n_samples, n_imp in (("g3_nsff_train", 64, 64), ("g3_nsff_train", 80, 17), ("g5_nsff_test_vis", 64, 32), ("g1_static_c1", 128, 0),
                                   ("g1_static_c1", 65, 0), ("g1_static_c1", 255, 0), ("g1_static_c1", 256, 0)):
        cfg = dict(scenes.CASES[name], n_rays=96)
        rays, ts = scenes.synthetic_rays(96, 11)
        models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
        _to_dev(models, emb)
        got = {}
        for form in ("chunks", "serial"):
            if form == "serial":
                monkeypatch.setenv("NSFF_SERIAL_COMPOSITE", "1")
            else:
                monkeypatch.delenv("NSFF_SERIAL_COMPOSITE", raising=False)
            torch.manual_seed(3)
            with torch.no_grad():
                out = A.render_rays(models, emb, rays.to(DEV), None if ts is None else ts.to(DEV), 29, n_samples, cfg.get("perturb", 0),
                                    cfg.get("noise_std", 0), n_imp, 32768, test_time=cfg["test_time"], **scenes.render_kwargs(cfg))
            got[form] = {k: v.clone() for k, v in out.items()}
        monkeypatch.delenv("NSFF_SERIAL_COMPOSITE", raising=False)
        a, b = got["chunks"], got["serial"]
        assert a.keys() == b.keys()
        S_f = n_samples + (2 if cfg["transient"] else 1) * n_imp if n_imp else n_samples
        assert 64 < S_f <= 256
        for k in a:
            per_sample = a[k].dim() >= 2 and a[k].shape[1] in (n_samples, S_f)
            if per_sample:
                assert torch.equal(a[k], b[k]), (name, k)
            else:
                scale = max(b[k].abs().max().item(), 1e-30)
                assert (a[k] - b[k]).abs().max().item() <= 3e-6 * scale, (name, k)


# ---- full-size configuration (BASELINE.json configs[1]): size-independent properties ----
def test_c2_full_size_properties(hip_lib, precision):
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=1024)
    rays, ts = scenes.synthetic_rays(1024, 42)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    kw = scenes.render_kwargs(cfg)
    rd, td = rays.to(DEV), ts.to(DEV)
    full = A.render_rays(models, emb, rd, td, 29, 64, 0, 0, 64, 32768, test_time=False, **kw)
    # (1) rays are independent: a render of two half batches concatenates to the full render, bit for bit
    halves = [A.render_rays(models, emb, rd[i:i + 512], td[i:i + 512], 29, 64, 0, 0, 64, 32768,
                            test_time=False, **kw) for i in (0, 512)]
    for k, v in full.items():
        assert torch.equal(v, torch.cat([h[k] for h in halves], 0)), k
    out = _np(full)
    assert out["zs_fine"].shape == (1024, 192) and out["disoccs_fw"].shape == (1024, 192, 1)
    for k, v in out.items():
        assert np.isfinite(v).all(), k
    # (2) compositing invariants
    assert (np.diff(out["zs_fine"], axis=1) >= 0).all()
    w = out["weights_fine"]
    assert (w >= 0).all() and (w.sum(1) <= 1 + 1e-5).all()
    np.testing.assert_allclose(out["depth_fine"], (w * out["zs_fine"]).sum(1), rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["transient_alpha_fine"], out["transient_weights_fine"].sum(1), atol=2e-6)
    np.testing.assert_allclose(out["xyz_fw"], out["xyz_fine"] + out["transient_flow_fw"], atol=1e-6)
    np.testing.assert_allclose(out["xyzs_fw"], out["xyzs_fine"] + out["transient_flows_fw"], atol=1e-6)
    assert (np.abs(out["transient_flows_fw"]) <= 0.2 + 1e-6).all()
    assert (out["transient_flows_fw"][out["zs_fine"] > 0.95] == 0).all()
    # (3) 24 rows of THIS batch against the reference itself (golden g19_c2_subset = the reference's outputs for these rays,
    # tests/golden/make_golden.py), every key at 1e-4 -- the chained re-query keys included.  Per-sample fine keys are
    # only comparable at identical depths (tests/parity.py), so the batch is rendered once more with those 24 rows
    # evaluated at the reference's zs_fine (the other 1000 rows keep the depths the first render sampled).
    sub = scenes.CASES["g19_c2_subset"]
    assert sub["batch"] == (1024, 42, 0) and {k: v for k, v in sub.items() if k not in ("n_rays", "batch")} == \
        {k: v for k, v in cfg.items() if k != "n_rays"}
    idx = scenes.subset_rows(sub)
    _, gold = common.load_golden("g19_c2_subset")
    zs = full["zs_fine"].clone()
    zs[torch.from_numpy(idx).to(DEV)] = torch.from_numpy(gold["zs_fine"]).to(DEV)
    at = _np(common.render_rays_at(zs)(models, emb, rd, td, 29, 64, 0, 0, 64, 32768, test_time=False, **kw))
    rest = np.setdiff1d(np.arange(1024), idx)
    assert np.array_equal(at["rgb_fine"][rest], out["rgb_fine"][rest])        # untouched rows: bit-identical
    worst = {}
    for k in gold:
        worst[k] = parity.assert_close(k, at[k][idx], gold[k], common.key_rtol(k, cfg))     # 1e-4, every key
    print("C2 subset vs reference, worst keys:", sorted(worst.items(), key=lambda kv: -kv[1])[:6])
    # ... and once more with autograd off: the calls above ran the activation-saving kernels (gradients were possible); this is
    # what bench.py times -- the inference launches, i.e. the hand-scheduled kernel at this size -- against the same reference rows
    with torch.no_grad():
        inf = _np(common.render_rays_at(zs)(models, emb, rd, td, 29, 64, 0, 0, 64, 32768, test_time=False, **kw))
    if precision.startswith("f16x3"):
        assert _lib.last_field_kernel() == ("h3_8wave" if precision.endswith("131") else "h3a_tb")
    for k in gold:
        parity.assert_close(k + " (inference launches)", inf[k][idx], gold[k], common.key_rtol(k, cfg))


def test_large_inference_launches_run_the_hand_scheduled_kernel(hip_lib, precision):
    """No silent fallback: a C2-sized f16x3 call (every launch >= 32768 points, no view-direction branch) must take
    nsff_field_kernel_h3a by default and with tile_points = 130, the eight-wave kernel with 131, the 64-point tiling below the
    size threshold, and the activation-saving kernels when gradients are wanted (the hand-scheduled body in its SAVE build)."""
    if not precision.startswith("f16x3"):
        pytest.skip("f16x3 kernels only")
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=1024)
    rays, ts = scenes.synthetic_rays(1024, 42)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    kw = scenes.render_kwargs(cfg)
    # (h3a_tb: the hand-scheduled kernel with the time code folded into per-ray bias rows -- 64 / 192 samples per ray)
    want = {"f16x3": "h3a_tb", "f16x3-130": "h3a_tb", "f16x3-131": "h3_8wave"}[precision]
    with torch.no_grad():
        A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), 29, 64, 0, 0, 64, 32768, test_time=False, **kw)
    assert _lib.last_field_kernel() == want
    with torch.no_grad():
        A.render_rays(models, emb, rays[:16].to(DEV), ts[:16].to(DEV), 29, 64, 0, 0, 64, 32768, test_time=False, **kw)
    assert _lib.last_field_kernel() == ("h3_64" if precision == "f16x3" else want)
    A.render_rays(models, emb, rays[:256].to(DEV), ts[:256].to(DEV), 29, 64, 0, 0, 64, 32768, test_time=False, **kw)   # autograd on
    assert _lib.last_field_kernel() == ("h3_save" if precision.endswith("131") else "h3a_save")     # (the training forward's body)


def test_ragged_and_empty_batches(hip_lib):
    cfg = dict(scenes.CASES["g4_nsff_test"])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    kw = scenes.render_kwargs(cfg)
    rays, ts = scenes.synthetic_rays(7, 5)            # 7*64 and 7*192 points: partial 64-point tiles
    out = _np(A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), 29, 64, 0, 0, 64, 32768,
                            test_time=True, **kw))
    want = common.oracle_render(cfg, models, emb, rays.numpy(), ts.numpy(), zs_fine_override=out["zs_fine"])
    for k in want:
        if k not in ("static_zs_fine", "transient_zs_fine"):
            parity.assert_close(k, out[k], want[k])
    empty = A.render_rays(models, emb, rays[:0].to(DEV), ts[:0].to(DEV), 29, 64, 0, 0, 64, 32768,
                          test_time=True, **kw)
    assert empty["rgb_fine"].shape == (0, 3) and empty["zs_fine"].shape == (0, 192)
    # odd sample counts (not multiples of the 64-lane wave)
    out = _np(A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), 29, 37, 0, 0, 21, 32768, test_time=True, **kw))
    cfg2 = dict(cfg, N_samples=37, N_importance=21)
    want = common.oracle_render(cfg2, models, emb, rays.numpy(), ts.numpy(), zs_fine_override=out["zs_fine"])
    for k in want:
        if k not in ("static_zs_fine", "transient_zs_fine"):
            parity.assert_close(k, out[k], want[k])


def test_merged_requery_launch_equals_two_launches(hip_lib, precision, monkeypatch):
    """Inference calls issue the two scene-flow re-queries (x + fw at t + 1, x + bw at t - 1) as ONE field launch over 2 N rays;
    NSFF_NO_MERGED_REQUERY=1 issues them separately.  Per-point arithmetic does not depend on which tile a point sits in: every
    key is bit-identical -- for a ray count whose point count is not a multiple of the 128-point tile, and at the C2 size."""
    cfg = dict(scenes.CASES["g3_nsff_train"])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    kw = scenes.render_kwargs(cfg)
    for n in (37, 1024):
        rays, ts = scenes.synthetic_rays(n, 7)
        outs = []
        for off in ("", "1"):
            if off:
                monkeypatch.setenv("NSFF_NO_MERGED_REQUERY", off)
            else:
                monkeypatch.delenv("NSFF_NO_MERGED_REQUERY", raising=False)
            torch.manual_seed(3)
            with torch.no_grad():
                outs.append(_np(A.render_rays(models, emb, rays.to(DEV), ts.to(DEV), 29, 64, 1, 1, 64, 32768, test_time=False, **kw)))
        assert set(outs[0]) == set(outs[1]) and "rgb_fw" in outs[0]
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), (n, k)
    monkeypatch.delenv("NSFF_NO_MERGED_REQUERY", raising=False)


def test_native_library_is_the_one_loaded(hip_lib):
    with open("/proc/self/maps") as f:
        assert any("libnsff_hip.so" in line for line in f), "HIP extension not loaded in this process"
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(os.path.dirname(A.__file__), "libnsff_hip.so"))


# ---- N3 / eval loop: frame ray generation and the chunked frame renderer (BASELINE configs[2]) ----
def test_frame_rays_match_reference(stages, hip_lib):
    from nsff_pl_amd import evaluate
    H, W = [int(v) for v in stages["rays/HW"]]
    got = evaluate.frame_rays(stages["rays/K"], stages["rays/c2w"], H, W, device=DEV).cpu().numpy()
    parity.assert_close("ndc rays", got, stages["rays/ndc"], 1e-5)
    part = evaluate.frame_rays(stages["rays/K"], stages["rays/c2w"], H, W, device=DEV, first_pixel=100, n_pixels=77)
    assert np.array_equal(part.cpu().numpy(), got[100:177])


def test_full_frame_eval_512x288(hip_lib, precision):
    """One 512x288 frame, test-time flags, 32768-ray chunks like eval.f; PSNR against the oracle on a subset."""
    if precision not in ("f32", "f16x3"):
        pytest.skip("full-frame run only on the two shipped modes (f16x3 = the hand-scheduled 128-point kernel at this size)")
    from nsff_pl_amd import evaluate
    cfg = dict(scenes.CASES["g4_nsff_test"])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    H, W = 288, 512
    K = np.array([[400., 0, W / 2], [0, 400., H / 2], [0, 0, 1]], np.float32)
    c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    rays = evaluate.frame_rays(K, c2w, H, W, device=DEV)
    ts = torch.full((H * W,), 7, dtype=torch.long, device=DEV)
    kw = scenes.render_kwargs(cfg)
    keys = ("rgb_fine", "depth_fine", "transient_alpha_fine", "zs_fine")
    out = evaluate.render_frame(models, emb, rays, ts, 29, 64, 64, chunk=32768, keys=keys, **kw)
    assert set(out) == set(keys) and out["rgb_fine"].shape == (H * W, 3)
    img = out["rgb_fine"].view(H, W, 3)
    assert torch.isfinite(img).all()
    # a frame rendered in one call equals the chunked one (rays are independent)
    one = evaluate.render_frame(models, emb, rays[:40000], ts[:40000], 29, 64, 64, chunk=40000, keys=("rgb_fine",), **kw)
    assert torch.equal(one["rgb_fine"], out["rgb_fine"][:40000])
    # oracle on 96 pixels spread over the frame, at the same fine depths
    idx = np.linspace(0, H * W - 1, 96).astype(np.int64)
    want = common.oracle_render(cfg, models, emb, rays.cpu().numpy()[idx], ts.cpu().numpy()[idx],
                                zs_fine_override=out["zs_fine"].cpu().numpy()[idx])
    got = out["rgb_fine"].cpu().numpy()[idx]
    parity.assert_close("rgb_fine", got, want["rgb_fine"])
    parity.assert_close("depth_fine", out["depth_fine"].cpu().numpy()[idx], want["depth_fine"])
    p = float(evaluate.psnr(torch.from_numpy(got), torch.from_numpy(want["rgb_fine"])))
    assert p > 80.0, f"PSNR(build, oracle) = {p:.1f} dB"
    # the same frame WITH the a6 branch (rendering.py:190-200: dataset= passed at test time -> the dynamic density of every
    # sample no training camera of frame ts[0] sees becomes -10): same 96 pixels against the oracle with visibility
    ds = scenes.DatasetStub(5)
    vkeys = keys + ("transient_weights_fine",)
    vis = evaluate.render_frame(models, emb, rays, ts, 29, 64, 64, chunk=32768, keys=vkeys, dataset=ds, **kw)
    assert not torch.equal(vis["rgb_fine"], out["rgb_fine"]), "the visibility mask changed nothing: the test is vacuous"
    want = common.oracle_render(cfg, models, emb, rays.cpu().numpy()[idx], ts.cpu().numpy()[idx], dataset=ds,
                                zs_fine_override=vis["zs_fine"].cpu().numpy()[idx])
    from oracle import nsff_oracle as orc
    dsd, r, z = ds.as_oracle_dict(), rays.cpu().numpy()[idx], vis["zs_fine"].cpu().numpy()[idx]
    world = orc.ndc_to_world((r[:, None, :3] + r[:, None, 3:] * z[..., None]).reshape(-1, 3), dsd["K"])
    masked = 1.0 - float(np.mean(orc.world_visibility(world, dsd["K"], dsd["H"], dsd["W"], dsd["poses"][7]) > 0))
    assert 0.05 < masked < 0.95, f"{masked:.2f} of the checked samples masked: not a test of both sides of the frustum"
    for k in ("rgb_fine", "depth_fine", "transient_alpha_fine", "transient_weights_fine"):
        parity.assert_close(k + " (visibility)", vis[k].cpu().numpy()[idx], want[k])


def test_readme_configuration_frame_512x288(hip_lib, precision):
    """The reference's own documented configuration (README.md:226-233, test.ipynb:78-85): use_viewdir, N_samples = 128,
    N_importance = 0, flows fw + bw, chunk 16384, one 512x288 frame -- rendered by the DEFAULT kernel selection (every launch is
    2.1 M points: the hand-scheduled kernel for both trunks, the view-direction static trunk through its per-ray rows) and checked
    on 96 pixels spread over the frame against the oracle, per-ray and per-sample keys."""
    if precision not in ("f32", "f16x3"):
        pytest.skip("full-frame run only on the two shipped modes")
    from nsff_pl_amd import evaluate
    cfg = dict(scenes.CASES["g6_readme_viewdir"])
    assert cfg["viewdir"] and cfg["N_samples"] == 128 and cfg["N_importance"] == 0 and cfg["flow"] == ["fw", "bw"]
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    H, W = 288, 512
    K = np.array([[400., 0, W / 2], [0, 400., H / 2], [0, 0, 1]], np.float32)
    c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    rays = evaluate.frame_rays(K, c2w, H, W, device=DEV)
    ts = torch.full((H * W,), 7, dtype=torch.long, device=DEV)
    kw = scenes.render_kwargs(cfg)
    keys = ("rgb_fine", "depth_fine", "transient_alpha_fine", "transient_flow_fw", "transient_flow_bw", "static_sigmas_fine",
            "static_rgbs_fine", "transient_sigmas_fine", "weights_fine", "_static_rgb_fine")
    out = evaluate.render_frame(models, emb, rays, ts, scenes.N_FRAMES - 1, 128, 0, chunk=16384, keys=keys, **kw)
    if precision == "f16x3":
        assert _lib.last_field_kernel() == "h3a_side", _lib.last_field_kernel()
    assert out["rgb_fine"].shape == (H * W, 3) and torch.isfinite(out["rgb_fine"]).all()
    idx = np.linspace(0, H * W - 1, 96).astype(np.int64)
    want = common.oracle_render(cfg, models, emb, rays.cpu().numpy()[idx], ts.cpu().numpy()[idx])
    worst = {}
    for k in keys:
        worst[k] = parity.assert_close(k, out[k].cpu().numpy()[idx], want[k])
    print(f"README-configuration frame [{precision}]: worst keys", sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    p = float(evaluate.psnr(out["rgb_fine"].cpu()[idx], torch.from_numpy(want["rgb_fine"])))
    assert p > 80.0, f"PSNR(build, oracle) = {p:.1f} dB"


def test_frame_egress_to_pinned_host_buffers(hip_lib, precision):
    """Row N4 (reference eval.py:87-110,222: per-chunk .cpu() of every key): the asynchronous egress returns exactly
    the GPU-resident values, in pinned memory, and does not slow the frame loop down."""
    if precision != "f16x3":
        pytest.skip("one arithmetic is enough for the copy path")
    import time
    from nsff_pl_amd import evaluate
    cfg = dict(scenes.CASES["g4_nsff_test"])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    H, W = 144, 256
    K = np.array([[200., 0, W / 2], [0, 200., H / 2], [0, 0, 1]], np.float32)
    c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    rays = evaluate.frame_rays(K, c2w, H, W, device=DEV)
    ts = torch.full((H * W,), 7, dtype=torch.long, device=DEV)
    kw = scenes.render_kwargs(cfg)
    keys = ("rgb_fine", "depth_fine", "static_rgbs_fine", "zs_fine")           # pixels and two per-sample tensors
    args = (models, emb, rays, ts, 29, 64, 64)
    gpu = evaluate.render_frame(*args, chunk=8192, keys=keys, **kw)
    host = evaluate.render_frame(*args, chunk=8192, keys=keys, to_host=True, **kw)
    assert isinstance(host, evaluate.HostFrame) and set(host) == set(keys)
    for k in keys:
        assert not host[k].is_cuda and host[k].is_pinned() and torch.equal(host[k], gpu[k].cpu()), k
    # caller-provided buffers, copies left in flight
    mine = {"rgb_fine": torch.empty(H * W, 3, pin_memory=True)}
    out = evaluate.render_frame(*args, chunk=8192, keys=("rgb_fine",), to_host=mine, sync=False, **kw)
    assert out["rgb_fine"] is mine["rgb_fine"]
    assert torch.equal(out.wait()["rgb_fine"], gpu["rgb_fine"].cpu())
    with pytest.raises(ValueError):
        evaluate.render_frame(*args, chunk=8192, keys=("rgb_fine",), to_host={"rgb_fine": torch.empty(5, 3, pin_memory=True)}, **kw)
    # two frames in flight at once (the pattern of the time interpolation: frame t is held while t + 1 renders): fresh
    # buffers per call by default, and a PinnedPool(depth=2) rotates its sets -- neither may let frame t + 1 land in frame t
    ts2 = ts + 1
    gpu2 = evaluate.render_frame(models, emb, rays, ts2, 29, 64, 64, chunk=8192, keys=keys, **kw)
    assert not torch.equal(gpu2["rgb_fine"], gpu["rgb_fine"])
    for to_host in (True, evaluate.PinnedPool(depth=2)):
        fa = evaluate.render_frame(*args, chunk=8192, keys=keys, to_host=to_host, sync=False, **kw)
        fb = evaluate.render_frame(models, emb, rays, ts2, 29, 64, 64, chunk=8192, keys=keys, to_host=to_host, sync=False, **kw)
        fa.wait(); fb.wait()
        for k in keys:
            assert fa[k].data_ptr() != fb[k].data_ptr(), k
            assert torch.equal(fa[k], gpu[k].cpu()) and torch.equal(fb[k], gpu2[k].cpu()), k
        assert not fa.stale and not fb.stale
    pool = evaluate.PinnedPool(depth=2)
    frames = [evaluate.render_frame(*args, chunk=8192, keys=("rgb_fine",), to_host=pool, sync=False, **kw) for _ in range(3)]
    assert frames[0].stale and not frames[1].stale and not frames[2].stale          # the third frame took the first one's set
    assert frames[2]["rgb_fine"].data_ptr() == frames[0]["rgb_fine"].data_ptr()
    assert torch.equal(frames[2].wait()["rgb_fine"], gpu["rgb_fine"].cpu())

    def timed(**extra):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            evaluate.render_frame(*args, chunk=8192, keys=keys, **extra, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3
    pool = evaluate.PinnedPool(depth=2)
    timed(); timed(to_host=pool)                              # warm both paths
    # best of three alternating measurements (a wall-clock comparison on a shared host: one slow repetition must not fail it)
    t_gpu, t_host = min(timed() for _ in range(3)), min(timed(to_host=pool) for _ in range(3))
    t_gpu = min(t_gpu, timed())
    t_block = timed(to_cpu=True)
    print(f"frame {H}x{W}: resident {t_gpu * 1e3:.1f} ms, async pinned egress {t_host * 1e3:.1f} ms, blocking .cpu() {t_block * 1e3:.1f} ms")
    assert t_host <= 1.05 * t_gpu + 2e-3, (t_host, t_gpu)


def test_f16x3_value_domain(hip_lib):
    """f16x3 carries every operand as hi + lo halfs (hi = rtz_f16(x), csrc/field_h3.hip: h3a_split2 / the epilogues): inside the fp16
    range the result is fp32-grade, beyond ~1.3e5 a pre-activation saturates silently.  INTEGRATION.md documents the limit; this
    test pins both sides of it against the exact-fp32 kernel: hidden activations of ~3e4 still agree at 1e-4, activations of
    ~1e6 do not (finite, wrong) -- whoever lifts the limit updates the document."""
    torch.manual_seed(5)
    m = A.NeRF("coarse", D=3, skips=[], use_viewdir=False).to(DEV)
    emb = A.PosEmbedding(9, 10)
    x = emb((torch.rand(4096, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV))
    x_in = torch.cat([x, torch.zeros(x.shape[0], m.in_channels_dir, device=DEV)], 1)     # [xyz | dir] rows of NeRF.forward
    w0 = m.static_xyz_encoding_1[0].weight.detach().clone()
    b0 = m.static_xyz_encoding_1[0].bias.detach()

    def run(gain, precision):
        with torch.no_grad():
            m.static_xyz_encoding_1[0].weight.copy_(w0 * gain)
        A.set_precision(precision)
        try:
            out = m(x_in, sigma_only=False, output_transient=False)
            torch.cuda.synchronize()
            return out.cpu().numpy()
        finally:
            A.set_precision(A.config.DEFAULT_PRECISION)
    act = lambda gain: float(torch.relu(x @ (w0 * gain).T + b0).abs().max())
    g_in = 3.0e4 / act(1.0)
    g_out = 1.0e6 / act(1.0)
    assert 2.0e4 < act(g_in) < 6.5e4 and act(g_out) > 5e5
    inside = parity.max_rel_err(run(g_in, "f16x3")[:, 3], run(g_in, "f32")[:, 3])
    outside_a, outside_b = run(g_out, "f16x3"), run(g_out, "f32")
    assert inside <= parity.RTOL, f"inside the fp16 range (|act| ~ 3e4): {inside:.2e}"
    assert np.isfinite(outside_a).all()
    outside = parity.max_rel_err(outside_a[:, 3], outside_b[:, 3])
    assert outside > 1e-2, f"|act| ~ 1e6 was expected to saturate the hi halfs (documented limit), got {outside:.2e}"
