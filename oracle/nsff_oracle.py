"""CPU oracle for the NSFF render hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

A plain numpy (fp32) restatement of the reference algorithm of kwea123/nsff_pl:

    pos_embedding      <- models/nerf.py:4-30        (PosEmbedding)
    nerf_forward       <- models/nerf.py:118-213     (NeRF.forward, all call modes)
    sample_pdf         <- models/rendering.py:10-49
    ndc_to_world       <- datasets/ray_utils.py:127-151
    world_visibility   <- datasets/ray_utils.py:154-181
    frame_rays         <- datasets/ray_utils.py:7-106 (get_ray_directions, get_rays, get_ndc_rays)
    render_rays        <- models/rendering.py:52-362 (inference :83-300,
                          render_transient_warping :98-140)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  The product path
(``nsff_pl_amd``) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
importing ``/root/reference`` (``tests/golden/make_golden.py``) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every key.

The reference draws torch RNG inside the path; here every random tensor is an explicit
input (``draws``), in the reference's draw order:
    perturb (N,S) | coarse_static, coarse_transient (N,S) | u_static, u_transient (N,Ni)
    | fine_static, fine_transient (N,Sf) | warp_fw, warp_bw (N,Sf)
"""
import numpy as np

F = np.float32
Z_FAR = F(0.95)


# ----------------------------------------------------------------------------- nerf.py
def pos_embedding(x, freqs):
    """[x, sin(f0 x), cos(f0 x), sin(f1 x), ...]  (nerf.py:25-30)."""
    x = np.asarray(x, F)
    out = [x]
    for f in np.asarray(freqs, F):
        fx = f * x
        out += [np.sin(fx), np.cos(fx)]
    return np.concatenate(out, -1).astype(F)


_TORCH_DENSE = False


def use_torch_dense(on):
    """Timing aid for bench.py's cpu_baseline: evaluate the dense layers with torch's CPU BLAS / threaded elementwise
    ops (what the reference's own ``nn.Linear`` + ReLU run on) instead of numpy's.  Same fp32 arithmetic up to
    summation order; the parity tests always run the numpy form."""
    global _TORCH_DENSE
    _TORCH_DENSE = bool(on)


def _lin(p, name, x):
    if _TORCH_DENSE:
        import torch
        return torch.addmm(torch.from_numpy(p[name + ".bias"]), torch.from_numpy(np.ascontiguousarray(x, F)),
                           torch.from_numpy(p[name + ".weight"]).t()).numpy()
    return x @ p[name + ".weight"].T + p[name + ".bias"]


def _sigmoid(x):
    return (F(1) / (F(1) + np.exp(-x))).astype(F)


def _trunk(p, prefix, x_in, D, skips):
    if _TORCH_DENSE:
        import torch
        x_t = torch.from_numpy(np.ascontiguousarray(x_in, F))
        h = x_t
        for i in range(D):
            if i in skips:
                h = torch.cat([x_t, h], 1)
            name = f"{prefix}_xyz_encoding_{i + 1}.0"
            h = torch.relu_(torch.addmm(torch.from_numpy(p[name + ".bias"]), h, torch.from_numpy(p[name + ".weight"]).t()))
        return h.numpy()
    h = x_in
    for i in range(D):
        if i in skips:
            h = np.concatenate([x_in, h], 1)
        h = np.maximum(_lin(p, f"{prefix}_xyz_encoding_{i + 1}.0", h), F(0))
    return h


def nerf_forward(p, cfg, x, sigma_only=False, output_static=True, output_transient=True,
                 output_transient_flow=()):
    """NeRF.forward (nerf.py:118-213).  p: state_dict as numpy, cfg: constructor facts.

    cfg keys: D, skips, in_xyz, in_dir, in_a, in_t, use_viewdir, flow_scale.
    """
    x = np.asarray(x, F)
    cx, cd, ca, ct = cfg["in_xyz"], cfg["in_dir"], cfg["in_a"], cfg["in_t"]
    D, skips = cfg["D"], cfg["skips"]
    if sigma_only:
        xyz = x[:, :cx]
        t = x[:, cx:cx + ct] if output_transient else None
        d = a = None
    else:
        xyz, d, a = x[:, :cx], x[:, cx:cx + cd], x[:, cx + cd:cx + cd + ca]
        t = x[:, cx + cd + ca:cx + cd + ca + ct] if output_transient else None

    static = None
    if output_static:
        h = _trunk(p, "static", xyz, D, skips)
        s_sigma = _lin(p, "static_sigma", h)                       # before *_final (:169)
        if sigma_only:
            if not output_transient:
                return s_sigma
            ht = _trunk(p, "transient", np.concatenate([xyz, t], 1), D, skips)
            t_sigma = _lin(p, "transient_sigma", _lin(p, "transient_xyz_encoding_final", ht))
            return np.concatenate([s_sigma, t_sigma], 1)
        feat = _lin(p, "static_xyz_encoding_final", h)
        if cfg["use_viewdir"]:
            feat = np.maximum(_lin(p, "static_dir_encoding.0", np.concatenate([feat, d, a], 1)), F(0))
        static = np.concatenate([_sigmoid(_lin(p, "static_rgb.0", feat)), s_sigma], 1)
        if not output_transient:
            return static

    ht = _trunk(p, "transient", np.concatenate([xyz, t], 1), D, skips)
    feat = _lin(p, "transient_xyz_encoding_final", ht)
    parts = [_sigmoid(_lin(p, "transient_rgb.0", feat)), _lin(p, "transient_sigma", feat)]
    for name in ("fw", "bw"):
        if name in output_transient_flow:
            parts.append(F(cfg["flow_scale"]) * np.tanh(_lin(p, f"transient_flow_{name}.0", feat)))
    transient = np.concatenate(parts, 1).astype(F)
    return np.concatenate([static, transient], 1) if output_static else transient


# ------------------------------------------------------------------------ rendering.py
def sample_pdf(bins, weights, u, eps=1e-5):
    """Inverse-CDF sampling with explicit u (rendering.py:10-49).  u: (Ni,) or (N,Ni)."""
    bins, weights = np.asarray(bins, F), np.asarray(weights, F)
    n, m = weights.shape
    w = weights + F(eps)
    pdf = w / w.sum(1, keepdims=True, dtype=F)
    # torch's CPU cumsum accumulates in double and rounds every output to fp32
    cdf = np.concatenate([np.zeros((n, 1), F), np.cumsum(pdf.astype(np.float64), 1).astype(F)], 1)
    u = np.broadcast_to(np.asarray(u, F), (n, u.shape[-1]))
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1)          # searchsorted(right=True)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, m)
    cdf_b, cdf_a = np.take_along_axis(cdf, below, 1), np.take_along_axis(cdf, above, 1)
    bin_b, bin_a = np.take_along_axis(bins, below, 1), np.take_along_axis(bins, above, 1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < F(eps), F(1), denom)
    return (bin_b + (u - cdf_b) / denom * (bin_a - bin_b)).astype(F)


def softplus(x):
    """torch.nn.Softplus(beta=1, threshold=20)."""
    x = np.asarray(x, F)
    return np.where(x > F(20), x, np.log1p(np.exp(np.minimum(x, F(20))))).astype(F)


def _excl_cumprod(one_minus_alpha):
    sh = np.concatenate([np.ones_like(one_minus_alpha[:, :1]), one_minus_alpha], 1)
    return np.cumprod(sh[:, :-1], 1, dtype=F)


def ndc_to_world(xyz, K, eps=1e-6):
    fx, fy, cx, cy = F(K[0, 0]), F(K[1, 1]), F(K[0, 2]), F(K[1, 2])
    rz = F(2) / (xyz[:, 2] - F(1) - F(eps))
    rx = -rz * xyz[:, 0] * cx / fx
    ry = -rz * xyz[:, 1] * cy / fy
    return np.stack([rx, ry, rz], 1).astype(F)


def world_visibility(xyz_w, K, H, W, c2w):
    pose = np.eye(4, dtype=F)
    pose[:3] = c2w
    w2c = np.linalg.inv(pose).astype(F)
    cam = w2c[:3, :3] @ xyz_w.T + w2c[:3, 3:]
    front = cam[2] < 0
    cam = np.stack([cam[0], -cam[1], -cam[2]], 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        img = np.asarray(K, F) @ cam
        u, v = img[0] / img[2], img[1] / img[2]
        inside = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    return (front & inside).astype(F)


def _zeros_like_draw(draws, key, shape):
    d = None if draws is None else draws.get(key)
    return np.zeros(shape, F) if d is None else np.asarray(d, F).reshape(shape)


def _inference(results, field, xyz, zs, *, test_time, output_transient, flows, noise_std,
               emb_xyz_freqs, dir_embedded, a_embedded, t_embedded, t_next, t_prev,
               draws, draw_prefix, dataset, ts):
    """rendering.py:83-300 for one model."""
    p, cfg, typ = field["params"], field["cfg"], field["typ"]
    n, s = zs.shape
    results[f"zs_{typ}"], results[f"xyzs_{typ}"] = zs, xyz
    pts = xyz.reshape(-1, 3)
    rep = lambda e: np.repeat(e, s, 0)                              # 'n1 c -> (n1 n2) c'
    emb = pos_embedding(pts, emb_xyz_freqs)
    noise_std = F(noise_std)

    if typ == "coarse" and test_time:
        cols = [emb] + ([rep(t_embedded)] if output_transient else [])
        out = nerf_forward(p, cfg, np.concatenate(cols, 1), sigma_only=True,
                           output_transient=output_transient).reshape(n, s, -1)
        s_sig_raw = out[..., 0]
        t_sig_raw = out[..., 1] if output_transient else None
        s_rgb = t_rgb = None
    else:
        cols = [emb, rep(dir_embedded)]
        if field["encode_appearance"]:
            cols.append(rep(a_embedded))
        if output_transient:
            cols.append(rep(t_embedded))
        out = nerf_forward(p, cfg, np.concatenate(cols, 1), output_transient=output_transient,
                           output_transient_flow=flows).reshape(n, s, -1)
        results[f"static_rgbs_{typ}"] = s_rgb = out[..., :3]
        s_sig_raw = out[..., 3]
        if output_transient:
            results[f"transient_rgbs_{typ}"] = t_rgb = out[..., 4:7]
            t_sig_raw = out[..., 7]
            if flows:
                far = (zs > Z_FAR)[..., None]
                results["transient_flows_fw"] = f_fw = np.where(far, F(0), out[..., 8:11])
                results["transient_flows_bw"] = f_bw = np.where(far, F(0), out[..., 11:14])

    if test_time and output_transient and dataset is not None:       # :190-200
        K = np.asarray(dataset["K"], F)
        vis = np.zeros(len(pts), F)
        world = ndc_to_world(pts, K)
        for i in range(dataset["n_cam_train"]):
            vis += world_visibility(world, K, dataset["H"], dataset["W"],
                                    dataset["poses"][i * dataset["N_frames"] + int(ts[0])])
        t_sig_raw = np.where(vis.reshape(n, s) == 0, F(-10), t_sig_raw)

    deltas = zs[:, 1:] - zs[:, :-1]
    d_static = np.concatenate([deltas, np.full((n, 1), 100, F)], 1)
    d_trans = np.concatenate([deltas, np.full((n, 1), 1e-3, F)], 1)

    results[f"static_sigmas_{typ}"] = s_sig = softplus(
        s_sig_raw + _zeros_like_draw(draws, draw_prefix + "_static", (n, s)) * noise_std)
    alphas = F(1) - np.exp(-d_static * s_sig)

    if output_transient:
        s_alpha = alphas
        results[f"transient_sigmas_{typ}"] = t_sig = softplus(
            t_sig_raw + _zeros_like_draw(draws, draw_prefix + "_transient", (n, s)) * noise_std)
        t_alpha = F(1) - np.exp(-d_trans * t_sig)
        alphas = F(1) - (F(1) - s_alpha) * (F(1) - t_alpha)

        if (not test_time) and flows:                                # :217-232
            def warp(xyz_w, t_emb_w, head, key):
                cols = [pos_embedding(xyz_w.reshape(-1, 3), emb_xyz_freqs), rep(dir_embedded)]
                if field["encode_appearance"]:
                    cols.append(rep(a_embedded))
                cols.append(rep(t_emb_w))
                o = nerf_forward(p, cfg, np.concatenate(cols, 1), output_static=False,
                                 output_transient=True, output_transient_flow=[head]).reshape(n, s, -1)
                rgb_w, sig_w = o[..., :3], o[..., 3]
                flow_w = np.where((zs > Z_FAR)[..., None], F(0), o[..., 4:7])
                al_w = F(1) - np.exp(-d_trans * softplus(
                    sig_w + _zeros_like_draw(draws, key, (n, s)) * noise_std))
                al = F(1) - (F(1) - s_alpha) * (F(1) - al_w)
                T = _excl_cumprod(F(1) - al)
                rgb = ((s_alpha * T)[..., None] * s_rgb).sum(1, dtype=F) + \
                      ((al_w * T)[..., None] * rgb_w).sum(1, dtype=F)
                return rgb.astype(F), flow_w, (al_w * T).astype(F)

            results["xyzs_fw"] = xyz_fw = xyz + f_fw
            results["rgb_fw"], flow_fw_bw, tw_fw = warp(xyz_fw, t_next, "bw", "warp_fw")
            results["xyzs_bw"] = xyz_bw = xyz + f_bw
            results["rgb_bw"], flow_bw_fw, tw_bw = warp(xyz_bw, t_prev, "fw", "warp_bw")
            results["xyzs_fw_bw"] = xyz_fw + flow_fw_bw
            results["xyzs_bw_fw"] = xyz_bw + flow_bw_fw

    T = _excl_cumprod(F(1) - alphas)
    weights = alphas * T
    if output_transient:
        results[f"static_weights_{typ}"] = s_w = s_alpha * T
        results[f"transient_weights_{typ}"] = t_w = t_alpha * T
        results[f"weights_{typ}"] = weights
    else:
        results[f"static_weights_{typ}"] = weights
    if test_time:
        if output_transient:
            results[f"static_alphas_{typ}"] = s_alpha
            results[f"transient_alphas_{typ}"] = t_alpha
        if typ == "coarse":
            return

    results[f"depth_{typ}"] = (weights * zs).sum(1, dtype=F)
    if not output_transient:
        results[f"rgb_{typ}"] = (weights[..., None] * s_rgb).sum(1, dtype=F)
        return
    t_map = (t_w[..., None] * t_rgb).sum(1, dtype=F)
    results[f"rgb_{typ}"] = (s_w[..., None] * s_rgb).sum(1, dtype=F) + t_map
    results[f"transient_alpha_{typ}"] = ta = t_w.sum(1, dtype=F)
    results[f"transient_rgb_{typ}"] = t_map + F(0.8) * (F(1) - ta[:, None])
    so_w = s_alpha * _excl_cumprod(F(1) - s_alpha)                   # :270-278
    results[f"_static_rgb_{typ}"] = (so_w[..., None] * s_rgb).sum(1, dtype=F)
    results[f"_static_depth_{typ}"] = (so_w * zs).sum(1, dtype=F)
    if flows:
        w3 = weights[..., None]
        results["xyz_fine"] = (w3 * xyz).sum(1, dtype=F)
        results["transient_flow_fw"] = (w3 * f_fw).sum(1, dtype=F)
        results["xyz_fw"] = results["xyz_fine"] + results["transient_flow_fw"]
        results["transient_flow_bw"] = (w3 * f_bw).sum(1, dtype=F)
        results["xyz_bw"] = results["xyz_fine"] + results["transient_flow_bw"]
        if (not test_time) and "disocc" in flows:
            occ_fw, occ_bw = tw_fw - t_w, tw_bw - t_w
            results["disocc_fw"] = F(1) - np.abs(occ_fw.sum(1, keepdims=True, dtype=F))
            results["disoccs_fw"] = (F(1) - np.abs(occ_fw))[..., None]
            results["disocc_bw"] = F(1) - np.abs(occ_bw.sum(1, keepdims=True, dtype=F))
            results["disoccs_bw"] = (F(1) - np.abs(occ_bw))[..., None]


def render_rays(fields, freqs_xyz, freqs_dir, rays, ts, max_t, emb_t=None, emb_a=None,
                N_samples=64, perturb=0, noise_std=0, N_importance=0, test_time=False,
                z_lin=None, u_lin=None, draws=None, output_transient=True,
                output_transient_flow=(), view_dir=None, dataset=None, zs_fine_override=None):
    """render_rays (rendering.py:52-362) with explicit random draws.

    fields: {'fine': field[, 'coarse': field]}, field = dict(typ, params, cfg,
    encode_appearance, encode_transient).  emb_t / emb_a: embedding tables (rows indexed
    by ts).  z_lin / u_lin: torch.linspace(0,1,N_samples / N_importance) if the caller
    wants bit-identical sample positions (defaults to np.linspace in fp32).
    zs_fine_override: (N, S_fine) depths that replace the merged fine samples -- used by
    the tests to compare the fine pass at identical positions, because the inverse-CDF
    draw is ill-conditioned in near-empty bins (see tests/parity.py).
    """
    rays = np.asarray(rays, F)
    n = rays.shape[0]
    o, d = rays[:, None, 0:3], rays[:, None, 3:6]
    dir_embedded = pos_embedding(rays[:, 3:6] if view_dir is None else view_dir, freqs_dir)
    z_lin = np.linspace(0, 1, N_samples, dtype=F) if z_lin is None else np.asarray(z_lin, F)
    zs = np.broadcast_to(z_lin, (n, N_samples)).astype(F)
    zs_mid = F(0.5) * (zs[:, :-1] + zs[:, 1:])
    if perturb > 0:
        upper = np.concatenate([zs_mid, zs[:, -1:]], 1)
        lower = np.concatenate([zs[:, :1], zs_mid], 1)
        zs = lower + (upper - lower) * (F(perturb) * np.asarray(draws["perturb"], F))
    results = {}
    common = dict(test_time=test_time, noise_std=noise_std, emb_xyz_freqs=freqs_xyz,
                  dir_embedded=dir_embedded, draws=draws, dataset=dataset, ts=ts)
    t_embedded = None
    if N_importance > 0:
        field = fields["coarse"]
        out_t = bool(output_transient and field["encode_transient"])
        if out_t:
            t_embedded = emb_t[ts]
        _inference(results, field, o + d * zs[..., None], zs, output_transient=out_t, flows=[],
                   a_embedded=None, t_embedded=t_embedded, t_next=None, t_prev=None,
                   draw_prefix="coarse", **common)
        if perturb == 0:
            u = np.linspace(0, 1, N_importance, dtype=F) if u_lin is None else np.asarray(u_lin, F)
            u_s = u_t = u
        else:
            u_s, u_t = draws["u_static"], draws.get("u_transient")
        zs_static = sample_pdf(zs_mid, results["static_weights_coarse"][:, 1:-1], u_s)
        zs_list = [zs, zs_static]
        if test_time:
            results["static_zs_fine"] = zs_static
        if out_t:
            zs_transient = sample_pdf(zs_mid, results["transient_weights_coarse"][:, 1:-1], u_t)
            zs_list.append(zs_transient)
            if test_time:
                results["transient_zs_fine"] = zs_transient
        zs = np.sort(np.concatenate(zs_list, 1), 1)
        if zs_fine_override is not None:
            zs = np.asarray(zs_fine_override, F)
    field = fields["fine"]
    a_embedded = emb_a[ts] if field["encode_appearance"] else None
    if N_importance == 0:
        out_t = bool(output_transient and field["encode_transient"])
        if out_t:
            t_embedded = emb_t[ts]
    flows = list(output_transient_flow) if out_t else []
    t_next = t_prev = None
    if out_t and flows and not test_time:
        t_next = emb_t[np.minimum(ts + 1, max_t)]
        t_prev = emb_t[np.maximum(ts - 1, 0)]
    _inference(results, field, o + d * zs[..., None], zs, output_transient=out_t, flows=flows,
               a_embedded=a_embedded, t_embedded=t_embedded, t_next=t_next, t_prev=t_prev,
               draw_prefix="fine", **common)
    return results


# ------------------------------------------------------------------------- ray_utils.py
def frame_rays(K, c2w, H, W, near=1.0):
    """NDC rays of a full frame: get_ray_directions -> get_rays -> get_ndc_rays
    (datasets/ray_utils.py:7-106 as called from datasets/monocular.py:268-276)."""
    K, c2w = np.asarray(K, F), np.asarray(c2w, F)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    j, i = np.meshgrid(np.arange(H, dtype=F), np.arange(W, dtype=F), indexing="ij")
    dirs = np.stack([(i - cx) / fx, -(j - cy) / fy, -np.ones_like(i)], -1).reshape(-1, 3)
    d = (dirs @ c2w[:, :3].T).astype(F)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(c2w[:, 3], d.shape)
    shift_near = F(-min(-1.0, float(c2w[2, 3])))
    t = -(shift_near + o[:, 2]) / d[:, 2]
    o = o + t[:, None] * d
    ox_oz, oy_oz = o[:, 0] / o[:, 2], o[:, 1] / o[:, 2]
    sx, sy = F(-1.0) / (cx / fx), F(-1.0) / (cy / fy)
    o2 = F(1.0) + F(2.0) * F(near) / o[:, 2]
    return np.stack([sx * ox_oz, sy * oy_oz, o2, sx * (d[:, 0] / d[:, 2] - ox_oz),
                     sy * (d[:, 1] / d[:, 2] - oy_oz), F(1.0) - o2], -1).astype(F)


# ------------------------------------------------------------------ helpers for callers
def softsplat_average(inp, flow):
    """FunctionSoftsplat(..., tenMetric=None, strType='average') for one image (models/softsplat.py:6-44,
    303-326): forward bilinear splat of `inp` (C,h,w) plus a ones channel along `flow` (2,h,w) [x then y],
    accumulated with += (atomicAdd there), then divided by the splatted ones (zeros -> 1).

    The reference kernel is CUDA-only (cupy; the CPU branch raises NotImplementedError, softsplat.py:238-239),
    so this function cannot be checked against a reference run: PARITY UNPINNED for the splat itself
    (hand-derived known-answer cases in tests/test_interpolate.py)."""
    C, h, w = inp.shape
    src = np.concatenate([inp.astype(F), np.ones((1, h, w), F)], 0)
    out = np.zeros((C + 1, h, w), F)
    xs, ys = np.meshgrid(np.arange(w, dtype=F), np.arange(h, dtype=F))
    ox, oy = (xs + flow[0].astype(F)).astype(F), (ys + flow[1].astype(F)).astype(F)
    with np.errstate(invalid="ignore"):
        fx, fy = np.floor(ox), np.floor(oy)
    ok = np.isfinite(fx) & np.isfinite(fy) & (np.abs(fx) < 2 ** 30) & (np.abs(fy) < 2 ** 30)
    nwx, nwy = np.where(ok, fx, -5).astype(np.int64), np.where(ok, fy, -5).astype(np.int64)
    sex, sey = (nwx + 1).astype(F), (nwy + 1).astype(F)
    corners = [(nwx, nwy, (sex - ox) * (sey - oy)), (nwx + 1, nwy, (ox - nwx.astype(F)) * (sey - oy)),
               (nwx, nwy + 1, (sex - ox) * (oy - nwy.astype(F))), (nwx + 1, nwy + 1, (ox - nwx.astype(F)) * (oy - nwy.astype(F)))]
    for cx, cy, wgt in corners:
        m = (cx >= 0) & (cx < w) & (cy >= 0) & (cy < h)
        for c in range(C + 1):
            np.add.at(out[c], (cy[m], cx[m]), (src[c][m] * wgt[m].astype(F)).astype(F))
    norm = out[-1:].copy()
    norm[norm == 0] = 1
    return (out[:-1] / norm).astype(F)


def interpolate(results_t, results_tp1, dt, K, c2w, img_wh):
    """models/rendering.py:365-460: time interpolation t -> t+dt by per-plane flow splatting + front-to-back
    MPI compositing.  results_*: dicts of numpy arrays (test-time render_rays outputs with flows).
    Returns (h,w,3) rgb and (h,w) NDC depth."""
    w, h = img_wh
    dt = F(dt)
    xyzs = results_t["xyzs_fine"].astype(F)
    n, S = xyzs.shape[:2]
    pose = np.eye(4, dtype=F)
    pose[:3] = c2w
    w2c = np.linalg.inv(pose)[:3].astype(F)
    w2c[1:] *= -1
    P = (np.asarray(K, F) @ w2c).astype(F)
    gx, gy = np.meshgrid(np.arange(w, dtype=F), np.arange(h, dtype=F))

    def plane_flows(flow, scale):
        pw = ndc_to_world(xyzs.reshape(-1, 3), K)
        qw = ndc_to_world((xyzs + flow.astype(F)).reshape(-1, 3), K)
        qw = (pw + scale * (qw - pw)).astype(F)
        uvd = (P[:, :3] @ qw.T + P[:, 3:]).astype(F)
        with np.errstate(divide="ignore", invalid="ignore"):
            uv = (uvd[:2] / uvd[2]).astype(F).reshape(2, h, w, S)
        return np.stack([uv[0] - gx[..., None], uv[1] - gy[..., None]], 0)          # (2,h,w,S)

    of_fw = plane_flows(results_t["transient_flows_fw"], dt)
    of_bw = plane_flows(results_tp1["transient_flows_bw"], F(1) - dt)
    zs = results_t["zs_fine"].reshape(h, w, S).astype(F)
    s_rgb = results_t["static_rgbs_fine"].reshape(h, w, S, 3).astype(F)
    s_a = results_t["static_alphas_fine"].reshape(h, w, S, 1).astype(F)

    def rgba_planes(res):
        rgb = res["transient_rgbs_fine"].reshape(h, w, S, 3)
        a = res["transient_alphas_fine"].reshape(h, w, S, 1)
        return np.concatenate([rgb, a], -1).astype(F).transpose(2, 3, 0, 1)           # (S,4,h,w)
    src_t, src_tp1 = rgba_planes(results_t), rgba_planes(results_tp1)

    rgba = np.zeros((h, w, 4), F)
    depth = np.zeros((h, w), F)
    for s in range(S):
        fw = softsplat_average(src_t[s], of_fw[..., s]).transpose(1, 2, 0)
        bw = softsplat_average(src_tp1[s], of_bw[..., s]).transpose(1, 2, 0)
        c_rgb = fw[..., :3] * fw[..., 3:] * (F(1) - dt) + bw[..., :3] * bw[..., 3:] * dt + s_rgb[:, :, s] * s_a[:, :, s]
        c_a = F(1) - (F(1) - (fw[..., 3:] * (F(1) - dt) + bw[..., 3:] * dt)) * (F(1) - s_a[:, :, s])
        rgba[..., :3] += (F(1) - rgba[..., 3:]) * c_rgb
        depth += (F(1) - rgba[..., 3]) * c_a[..., 0] * zs[..., s]
        rgba[..., 3:] += (F(1) - rgba[..., 3:]) * c_a
    return rgba[..., :3].astype(F), depth.astype(F)


def field_from_module(module):
    """Describe a NeRF nn.Module (reference's or the build's: same attribute names)."""
    params = {k: v.detach().cpu().numpy().astype(F) for k, v in module.state_dict().items()}
    cfg = dict(D=module.D, skips=list(module.skips), in_xyz=module.in_channels_xyz,
               in_dir=module.in_channels_dir, in_a=module.in_channels_a, in_t=module.in_channels_t,
               use_viewdir=bool(module.use_viewdir), flow_scale=float(getattr(module, "flow_scale", 0.0)))
    return dict(typ=module.typ, params=params, cfg=cfg,
                encode_appearance=bool(module.encode_appearance),
                encode_transient=bool(module.encode_transient))
