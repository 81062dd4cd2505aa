"""Time interpolation: oracle and HIP `interpolate` against the reference's
rendering.py:365-460 (golden G11) and hand-derived splat cases.

The reference's splat kernel (models/softsplat.py, cupy/CUDA) cannot run in the build container; the golden was
produced by the reference's `interpolate` with the oracle's splat standing in for it, so G11 pins the projection,
plane optical flow and MPI compositing; the splat itself is pinned by the known-answer cases below.
"""
import numpy as np
import pytest
import torch

import common
import parity
import scenes
import nsff_pl_amd as A
from oracle import nsff_oracle as orc


def _golden():
    z = np.load(common.GOLDEN_DIR + "/g11_interpolate.npz")
    st = np.load(common.GOLDEN_DIR + "/g8_stages.npz")
    H, W = [int(v) for v in st["rays/HW"]]
    res_t = {k[2:]: z[k] for k in z.files if k.startswith("t/")}
    res_tp1 = {k[4:]: z[k] for k in z.files if k.startswith("tp1/")}
    outs = {dt: (z[f"out/rgb_{dt}"], z[f"out/depth_{dt}"]) for dt in scenes.INTERP_DTS}
    return res_t, res_tp1, st["rays/K"], st["rays/c2w"], (W, H), outs, st["rays/ndc"], float(z["weight_checksum"])


# ---- known-answer cases of the 'average' forward splat (softsplat.py:12-43, 307-326) ----
def test_splat_integer_shift_moves_pixels():
    inp = np.arange(24, dtype=np.float32).reshape(2, 3, 4) + 1
    flow = np.zeros((2, 3, 4), np.float32)
    flow[0], flow[1] = 1, -1                                       # one pixel right, one up
    out = orc.softsplat_average(inp, flow)
    want = np.zeros_like(inp)
    want[:, :2, 1:] = inp[:, 1:, :3]
    assert np.array_equal(out, want)                                # vacated / never-hit pixels stay 0 (norm 0 -> 1)


def test_splat_half_pixel_averages_neighbours():
    inp = np.array([[[2.0, 4.0, 8.0, 16.0]]], np.float32)
    flow = np.zeros((2, 1, 4), np.float32)
    flow[0] = 0.5
    out = orc.softsplat_average(inp, flow)
    # target x: 0.5*src[x-1] + 0.5*src[x] over weight 1 (interior), 0.5*src[0]/0.5 at x=0; the last source's
    # east half falls outside and is dropped
    assert np.allclose(out, [[[2.0, 3.0, 6.0, 12.0]]], rtol=1e-6)


def test_splat_collision_is_weighted_average_and_bounds_are_dropped():
    inp = np.array([[[1.0, 5.0, 9.0]]], np.float32)
    flow = np.zeros((2, 1, 3), np.float32)
    flow[0] = [1.0, 0.0, -7.0]                                      # 0 -> 1 (collides with 1), 2 -> outside
    out = orc.softsplat_average(inp, flow)
    assert np.allclose(out, [[[0.0, 3.0, 0.0]]])
    flow[0] = [np.nan, 0.0, 1e30]
    with np.errstate(all="ignore"):
        out = orc.softsplat_average(inp, flow)
    assert np.allclose(out, [[[0.0, 5.0, 0.0]]])


def test_splat_bilinear_weights_sum_to_one():
    rng = np.random.RandomState(0)
    inp = np.ones((1, 9, 11), np.float32)
    flow = rng.uniform(-0.9, 0.9, (2, 9, 11)).astype(np.float32)
    out = orc.softsplat_average(inp, flow)
    assert np.all((np.abs(out - 1) < 1e-5) | (out == 0))            # average of ones is one wherever anything landed


def test_oracle_interpolate_matches_reference_golden():
    res_t, res_tp1, K, c2w, wh, outs, _, _ = _golden()
    for dt, (rgb, depth) in outs.items():
        o_rgb, o_depth = orc.interpolate(res_t, res_tp1, dt, K, c2w, wh)
        parity.assert_close(f"rgb dt={dt}", o_rgb, rgb, parity.RTOL)
        parity.assert_close(f"depth dt={dt}", o_depth, depth, parity.RTOL)
    flow_px = np.abs(res_t["transient_flows_fw"]).max() * wh[0] / 2
    assert flow_px > 1.0                                            # the case really moves samples across pixels


def test_interpolate_needs_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    res_t, res_tp1, K, c2w, wh, *_ = _golden()
    with pytest.raises(RuntimeError):
        A.interpolate({k: torch.from_numpy(v) for k, v in res_t.items()},
                      {k: torch.from_numpy(v) for k, v in res_tp1.items()}, 0.3, K, c2w, wh)


# ---- HIP path ----
@pytest.mark.gpu
def test_hip_interpolate_matches_reference_golden(hip_lib):
    res_t, res_tp1, K, c2w, wh, outs, _, _ = _golden()
    dev = torch.device("cuda:0")
    rt = {k: torch.from_numpy(v).to(dev) for k, v in res_t.items()}
    rt1 = {k: torch.from_numpy(v).to(dev) for k, v in res_tp1.items()}
    for dt, (rgb, depth) in outs.items():
        g_rgb, g_depth = A.interpolate(rt, rt1, dt, torch.from_numpy(K), torch.from_numpy(c2w), wh)
        assert g_rgb.shape == (wh[1], wh[0], 3) and g_depth.shape == (wh[1], wh[0])
        parity.assert_close(f"rgb dt={dt}", g_rgb.cpu().numpy(), rgb, parity.RTOL)
        parity.assert_close(f"depth dt={dt}", g_depth.cpu().numpy(), depth, parity.RTOL)
    # CPU tensors (what eval.f of the reference hands over, eval.py:101-104) are accepted and moved
    c_rgb, _ = A.interpolate({k: torch.from_numpy(v) for k, v in res_t.items()},
                             {k: torch.from_numpy(v) for k, v in res_tp1.items()}, 0.3, K, c2w, wh)
    parity.assert_close("rgb from cpu dicts", c_rgb.cpu().numpy(), outs[0.3][0], parity.RTOL)


@pytest.mark.gpu
def test_hip_splat_against_oracle_random_and_degenerate_flows(hip_lib):
    """nsff_splat_planes + nsff_mpi_composite on crafted inputs: huge / NaN / out-of-frame flows, all-transparent
    planes, dt at the ends of (0,1)."""
    res_t, res_tp1, K, c2w, wh, *_ = _golden()
    rng = np.random.RandomState(5)
    res_t, res_tp1 = dict(res_t), dict(res_tp1)
    f = res_t["transient_flows_fw"].copy()
    # in-plane flows only: scaling the z flow too pushes points through the NDC far plane (z -> 1), where the
    # projection w = 2/(z-1-eps) is ill-conditioned and 1-ulp differences move samples by pixels
    f[..., :2] *= rng.choice([0.0, 1.0, 8.0, -30.0], size=f.shape[:2] + (1,)).astype(np.float32)
    f[3, 5] = np.nan
    f[10, 2] = 1e30
    res_t["transient_flows_fw"] = f
    res_tp1["transient_alphas_fine"] = res_tp1["transient_alphas_fine"] * (rng.rand(*res_tp1["transient_alphas_fine"].shape) > 0.5)
    res_t["static_alphas_fine"] = np.zeros_like(res_t["static_alphas_fine"])
    dev = torch.device("cuda:0")
    for dt in (1e-3, 0.5, 0.999):
        with np.errstate(all="ignore"):
            o_rgb, o_depth = orc.interpolate(res_t, res_tp1, dt, K, c2w, wh)
        g_rgb, g_depth = A.interpolate({k: torch.from_numpy(v).to(dev) for k, v in res_t.items()},
                                       {k: torch.from_numpy(v).to(dev) for k, v in res_tp1.items()}, dt, K, c2w, wh)
        parity.assert_close(f"rgb dt={dt}", g_rgb.cpu().numpy(), o_rgb, parity.RTOL)
        parity.assert_close(f"depth dt={dt}", g_depth.cpu().numpy(), o_depth, parity.RTOL)


@pytest.mark.gpu
def test_hip_render_then_interpolate_end_to_end(hip_lib):
    """eval.py:199-213 on the device: render t and t+1 with the HIP path, interpolate, compare with the golden
    (which went reference render -> reference interpolate)."""
    from test_gpu_parity import _to_dev, DEV
    res_t, res_tp1, K, c2w, wh, outs, rays, checksum = _golden()
    cfg = dict(scenes.INTERP_CFG, n_rays=wh[0] * wh[1])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    assert abs(scenes.weight_checksum(models, emb) - checksum) <= 1e-9 * abs(checksum)
    _to_dev(models, emb)
    rays = torch.from_numpy(rays).to(DEV)
    both = []
    for t in (scenes.INTERP_T, scenes.INTERP_T + 1):
        with torch.no_grad():
            both.append(A.render_rays(models, emb, rays, torch.full((rays.shape[0],), t, device=DEV), scenes.N_FRAMES - 1,
                                      cfg["N_samples"], 0, 0, cfg["N_importance"], 1024 * 32, test_time=True,
                                      **scenes.render_kwargs(cfg)))
    for dt, (rgb, depth) in outs.items():
        g_rgb, g_depth = A.interpolate(both[0], both[1], dt, K, c2w, wh)
        # Free-running (own fine depths, own flows).  The reference's 'average' splat is DISCONTINUOUS where an output
        # cell receives only an epsilon of bilinear weight: its value is (src * eps) / eps = src, and 0 when the cell
        # is not touched at all -- a 1e-6 difference in a flow flips it.  Such cells are isolated pixels: 99 % of the
        # pixels must agree to 1e-4 of the range, the few outliers (3 of 576 at the time of writing) to 5e-2.
        for name, got, want in (("rgb", g_rgb.cpu().numpy(), rgb), ("depth", g_depth.cpu().numpy()[..., None], depth[..., None])):
            err = np.abs(got - want).max(-1).ravel() / np.abs(want).max()
            assert np.isfinite(got).all()
            assert np.percentile(err, 99) <= parity.RTOL, (name, dt, float(np.percentile(err, 99)))
            assert (err > 1e-3).mean() <= 0.01 and err.max() <= 5e-2, (name, dt, float(err.max()), int((err > 1e-3).sum()))


@pytest.mark.gpu
def test_render_sequence_mirrors_the_eval_loop(hip_lib):
    """eval.py:171-222 with split test_fixview*_interp3 on three samples: names, frame reuse and in-between frames."""
    from test_gpu_parity import _to_dev, DEV
    from nsff_pl_amd import evaluate
    res_t, res_tp1, K, c2w, wh, outs, rays, checksum = _golden()
    cfg = dict(scenes.INTERP_CFG, n_rays=wh[0] * wh[1])
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    _to_dev(models, emb)
    rays = torch.from_numpy(rays).to(DEV)
    ts = torch.full((rays.shape[0],), scenes.INTERP_T, device=DEV)
    samples = [dict(rays=rays, ts=ts + k, c2w=torch.from_numpy(c2w)) for k in range(3)]
    kw = scenes.render_kwargs(cfg)
    frames = list(evaluate.render_sequence(models, emb, samples, scenes.N_FRAMES - 1, cfg["N_samples"], cfg["N_importance"],
                                           wh, interp=3, K=torch.from_numpy(K), **kw))
    assert [f[0] for f in frames] == ["000_000", "000_033", "000_066", "001_000", "001_033", "001_066", "002_000"]
    a = evaluate.render_frame(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], cfg["N_importance"], **kw)
    b = evaluate.render_frame(models, emb, rays, ts + 1, scenes.N_FRAMES - 1, cfg["N_samples"], cfg["N_importance"], **kw)
    assert torch.equal(frames[0][1], torch.clip(a["rgb_fine"].view(wh[1], wh[0], 3), 0, 1))
    assert torch.equal(frames[3][1], torch.clip(b["rgb_fine"].view(wh[1], wh[0], 3), 0, 1))     # reused as left end
    img, dep = A.interpolate(a, b, 1 / 3, K, c2w, wh)
    parity.assert_close("in-between", frames[1][1].cpu().numpy(), torch.clip(img, 0, 1).cpu().numpy(), 1e-5)
    # plain split: one image per sample
    plain = list(evaluate.render_sequence(models, emb, samples[:2], scenes.N_FRAMES - 1, cfg["N_samples"],
                                          cfg["N_importance"], wh, **dict(kw, output_transient_flow=[])))
    assert [f[0] for f in plain] == ["000", "001"] and plain[0][1].shape == (wh[1], wh[0], 3)


# ---- multi-tile frame: the splat's tile ownership, halo and far path (csrc/interp.hip: TILE 32x8, HALO 4, 8 planes
# per workgroup) against the oracle's scatter-add restatement of softsplat.py:6-44 / :303-326 ----
SPLAT_TILE_X, SPLAT_TILE_Y, SPLAT_HALO, SPLAT_PLANES = 32, 8, 4, 4
# landing-cell offsets (pixels, after the dt scaling): inside the halo, on its last cell (3.75 -> +3, -3.75 -> -4),
# first cell beyond it (4.25 -> +4, -4.25 -> -5), far, several tiles away, out of the frame.  No value is near an
# integer: a cell that only receives an epsilon weight is normalised to a full value ('average'), i.e. the floor of a
# near-integer landing would be a legitimate one-ulp discontinuity, not what this test is about.
# every shift has a fractional part: the reference's 'average' splat divides by the splatted ones with exact zeros replaced by
# one (softsplat.py:303-326), so a sample landing ON a pixel centre gives its neighbour either no weight or an epsilon weight
# that the normalisation turns into the full value -- a discontinuity of the reference itself, decided by the last bit of the
# projection (seen at 512 x 288 with a shift of exactly 150: 3 % of the lower half's pixels differed by up to 0.07)
SPLAT_SHIFTS = (0.25, -0.5, 2.3, -2.3, 3.75, -3.75, 4.25, -4.25, 7.6, -7.6, 33.5, -33.5, 150.4)


def multi_tile_case(W=96, H=40, S=24, seed=3):
    """Crafted test-time dicts on a (W,H) frame with identity pose: sample (px,py,s) projects onto its own pixel, and
    its scene flow moves it by a chosen number of pixels at dt=0.4 (forward) / 0.6 (backward)."""
    rng = np.random.RandomState(seed)
    f = 80.0
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32)
    c2w = np.eye(4, dtype=np.float32)[:3]
    dt = 0.4
    n = H * W
    px, py = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x_ndc = (px / (W / 2) - 1).reshape(n, 1)
    y_ndc = (1 - py / (H / 2)).reshape(n, 1)
    zs = np.sort(rng.uniform(-0.9, 0.8, (n, S)), 1)
    xyz = np.stack([np.broadcast_to(x_ndc, (n, S)), np.broadcast_to(y_ndc, (n, S)), zs], -1).astype(np.float32)

    def flows(scale):
        sx = rng.choice(SPLAT_SHIFTS, size=(n, S))
        sy = rng.choice(SPLAT_SHIFTS, size=(n, S))
        fl = np.zeros((n, S, 3), np.float32)
        fl[..., 0] = sx / scale / (W / 2)                      # pixel shift -> NDC, undone by the dt scaling
        fl[..., 1] = -sy / scale / (H / 2)
        return fl, sx, sy
    f_fw, sx, sy = flows(dt)
    f_bw, _, _ = flows(1 - dt)
    u = lambda *s: rng.uniform(0.05, 0.95, s).astype(np.float32)
    res_t = dict(xyzs_fine=xyz, zs_fine=zs.astype(np.float32), static_rgbs_fine=u(n, S, 3),
                 static_alphas_fine=u(n, S) * 0.2, transient_flows_fw=f_fw, transient_rgbs_fine=u(n, S, 3),
                 transient_alphas_fine=u(n, S) * 0.3)
    res_tp1 = dict(transient_flows_bw=f_bw, transient_rgbs_fine=u(n, S, 3), transient_alphas_fine=u(n, S) * 0.3)
    return res_t, res_tp1, dt, K, c2w, (W, H), (sx, sy)


def test_multi_tile_case_exercises_every_splat_path():
    """The crafted flows really land where intended (checked with the oracle's projection) and cover: near samples
    that cross tile boundaries in x and y, the last halo cell, the first far cell, far moves and out-of-frame."""
    res_t, res_tp1, dt, K, c2w, (W, H), (sx, sy) = multi_tile_case()
    n, S = sx.shape
    xyz = res_t["xyzs_fine"].reshape(-1, 3)
    pw = orc.ndc_to_world(xyz, K)
    qw = orc.ndc_to_world(xyz + res_t["transient_flows_fw"].reshape(-1, 3), K)
    qw = pw + np.float32(dt) * (qw - pw)
    w2c = np.eye(4, dtype=np.float32)[:3].copy()
    w2c[1:] *= -1
    uvd = (K @ w2c)[:, :3] @ qw.T
    u, v = (uvd[0] / uvd[2]).reshape(n, S), (uvd[1] / uvd[2]).reshape(n, S)
    px, py = np.meshgrid(np.arange(W), np.arange(H))
    px, py = px.reshape(n, 1), py.reshape(n, 1)
    assert np.abs(u - (px + sx)).max() < 1e-3 and np.abs(v - (py + sy)).max() < 1e-3
    dx, dy = np.floor(u).astype(int) - px, np.floor(v).astype(int) - py
    near = (dx >= -SPLAT_HALO) & (dx < SPLAT_HALO) & (dy >= -SPLAT_HALO) & (dy < SPLAT_HALO)
    inside = (px + dx >= 0) & (px + dx + 1 < W) & (py + dy >= 0) & (py + dy + 1 < H)
    cross_x = (px // SPLAT_TILE_X) != ((px + dx + 1) // SPLAT_TILE_X)
    cross_y = (py // SPLAT_TILE_Y) != ((py + dy + 1) // SPLAT_TILE_Y)
    assert W // SPLAT_TILE_X >= 3 and H // SPLAT_TILE_Y >= 5 and S // SPLAT_PLANES >= 3
    for what, m in {"near, crosses a tile boundary in x": near & cross_x & inside,
                    "near, crosses a tile boundary in y": near & cross_y & inside,
                    "near, crosses both": near & cross_x & cross_y & inside,
                    "last halo cell (+3 / -4) across a boundary": near & ((dx == 3) | (dx == -4)) & cross_x & inside,
                    "last halo row (+3 / -4) across a boundary": near & ((dy == 3) | (dy == -4)) & cross_y & inside,
                    "first cell beyond the halo (+4 / -5)": ((dx == 4) | (dx == -5)) & inside,
                    "first row beyond the halo (+4 / -5)": ((dy == 4) | (dy == -5)) & inside,
                    "far in x only": (np.abs(dx) > 5) & (np.abs(dy) < 4) & inside,
                    "several tiles away": (np.abs(dx) > 32) & inside,
                    "out of the frame": ~inside}.items():
        assert m.sum() > 50, f"crafted case has only {m.sum()} samples for: {what}"


@pytest.mark.gpu
@pytest.mark.parametrize("shape,binning", [((96, 40, 24), True), ((70, 19, 11), True), ((96, 40, 24), False),
                                           ((512, 288, 64), True)])
def test_hip_interpolate_multi_tile_matches_oracle(shape, binning, hip_lib, monkeypatch):
    """whole tiles / ragged tiles and plane groups / the atomic far route / the C5 frame size (BASELINE.json configs[4]:
    512 x 288, 64 planes here -- the oracle's numpy scatter-add needs ~20 s for it)."""
    import nsff_pl_amd.interpolation as I
    monkeypatch.setattr(I, "_FAR_BINNING", binning)
    W, H, S = shape
    res_t, res_tp1, dt, K, c2w, wh, _ = multi_tile_case(W, H, S)
    dev = torch.device("cuda:0")
    with np.errstate(all="ignore"):
        o_rgb, o_depth = orc.interpolate(res_t, res_tp1, dt, K, c2w, wh)
    g_rgb, g_depth = A.interpolate({k: torch.from_numpy(v).to(dev) for k, v in res_t.items()},
                                   {k: torch.from_numpy(v).to(dev) for k, v in res_tp1.items()}, dt, K, c2w, wh)
    assert np.abs(o_rgb).max() > 0.3
    parity.assert_close(f"rgb {shape}", g_rgb.cpu().numpy(), o_rgb, parity.RTOL)
    parity.assert_close(f"depth {shape}", g_depth.cpu().numpy(), o_depth, parity.RTOL)
    # per-pixel, not only max-norm: count pixels off by more than 1e-4 of the range
    bad = (np.abs(g_rgb.cpu().numpy() - o_rgb).max(-1) > 1e-4 * np.abs(o_rgb).max()).sum()
    assert bad == 0, f"{bad} pixels differ"
