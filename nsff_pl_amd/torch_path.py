"""Differentiable torch expression of the render path, used ONLY to obtain gradients.

``render_rays`` computes its outputs with the gfx950 kernels (no autograd graph).  When the
caller needs gradients (training, reference ``train.py:178-198``), :mod:`nsff_pl_amd.autograd`
re-evaluates the same mathematics here with ordinary torch ops on the GPU -- same sample depths,
same random draws -- and lets autograd differentiate it (activation checkpointing at the granularity
of one ``render_rays`` call).  Nothing in this module is on the forward / inference path.

The dense layers of this backward path run on rocBLAS through ``torch.nn.functional.linear``
(plain library GEMMs); replacing them by native MFMA backward kernels is the planned next step
 and only needs :func:`field` to become an ``autograd.Function``.

Algorithm references: models/nerf.py:118-213 (field), models/rendering.py:98-140, 187-188,
202-298 (compositing / warping / disocclusion).
"""
import torch
import torch.nn.functional as F

Z_FAR = 0.95


def pos_embed(x, freqs):
    out = [x]
    for f in freqs:
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def _lin(mod, x):
    layer = mod[0] if isinstance(mod, torch.nn.Sequential) else mod
    return F.linear(x, layer.weight, layer.bias)


def _trunk(model, prefix, x_in):
    h = x_in
    for i in range(model.D):
        if i in model.skips:
            h = torch.cat([x_in, h], 1)
        h = torch.relu(_lin(getattr(model, f"{prefix}_xyz_encoding_{i + 1}"), h))
    return h


def field(model, emb_xyz, dir_rows, a_rows, t_rows, static=True, transient=True, flows=()):
    """NeRF.forward on embedded point rows; returns the slot-ordered columns that exist:
    dict(rgb_s, sigma_s, rgb_t, sigma_t, fw, bw) with (P,3)/(P,) tensors."""
    out = {}
    if static:
        h = _trunk(model, "static", emb_xyz)
        out["sigma_s"] = _lin(model.static_sigma, h)[:, 0]
        feat = _lin(model.static_xyz_encoding_final, h)
        if model.use_viewdir:
            cols = [feat, dir_rows] + ([a_rows] if model.in_channels_a > 0 else [])
            feat = torch.relu(_lin(model.static_dir_encoding, torch.cat(cols, 1)))
        out["rgb_s"] = torch.sigmoid(_lin(model.static_rgb, feat))
    if transient:
        h = _trunk(model, "transient", torch.cat([emb_xyz, t_rows], 1))
        feat = _lin(model.transient_xyz_encoding_final, h)
        out["sigma_t"] = _lin(model.transient_sigma, feat)[:, 0]
        out["rgb_t"] = torch.sigmoid(_lin(model.transient_rgb, feat))
        if "fw" in flows:
            out["fw"] = model.flow_scale * torch.tanh(_lin(model.transient_flow_fw, feat))
        if "bw" in flows:
            out["bw"] = model.flow_scale * torch.tanh(_lin(model.transient_flow_bw, feat))
    return out


def query(model, xyz, freqs_xyz, dir_embedded, a_embedded, t_embedded, s, static, transient, flows, saved=None):
    """Field outputs for (P,3) points, `s` consecutive points per ray.  On the GPU (models without view
    directions) this is the native node of :mod:`nsff_pl_amd.field_grad`; otherwise the torch expression."""
    from . import field_grad
    if field_grad.supported(model, xyz):
        raw = field_grad.field(model, xyz, freqs_xyz, t_embedded if transient else None, s, static, transient, saved)
        out = {}
        if static:
            out["rgb_s"], out["sigma_s"] = raw[:, 0:3], raw[:, 3]
        if transient:
            out["rgb_t"], out["sigma_t"] = raw[:, 4:7], raw[:, 7]
            if "fw" in flows:
                out["fw"] = raw[:, 8:11]
            if "bw" in flows:
                out["bw"] = raw[:, 11:14]
        return out
    rep = lambda e: None if e is None else e.repeat_interleave(s, 0)
    return field(model, pos_embed(xyz, freqs_xyz), rep(dir_embedded), rep(a_embedded), rep(t_embedded),
                 static, transient, flows)


def _excl_cumprod_raw(x):
    return torch.cumprod(torch.cat([torch.ones_like(x[:, :1]), x], 1)[:, :-1], 1)


def _rev_excl_cumsum(v):
    return torch.flip(torch.cumsum(torch.flip(v, [1]), 1), [1]) - v


class _ExclCumprod(torch.autograd.Function):
    """T_i = prod_{j<i} x_j along dim 1 (rendering.py:226-229).  torch.cumprod's own backward asks the host whether
    the input holds zeros (a device sync, which also forbids hipGraph capture); this backward is the same
    mathematics without the question: sum_{i>j} g_i T_i / x_j where x_j != 0, and for the first zero of a row the
    products are re-formed with that factor left out (entries behind a zero get exactly 0, as they must)."""

    @staticmethod
    def forward(ctx, x):
        T = _excl_cumprod_raw(x)
        ctx.save_for_backward(x, T)
        return T

    @staticmethod
    def backward(ctx, g):
        x, T = ctx.saved_tensors
        zero = x == 0
        first = zero & (torch.cumsum(zero.to(torch.int32), 1) == 1)
        T1 = _excl_cumprod_raw(torch.where(first, torch.ones_like(x), x))
        plain = _rev_excl_cumsum(g * T) / torch.where(zero, torch.ones_like(x), x)
        return torch.where(first, _rev_excl_cumsum(g * T1), torch.where(zero, torch.zeros_like(x), plain))


def _excl_cumprod(x):
    return _ExclCumprod.apply(x)


def _softplus(x):
    return F.softplus(x)          # beta = 1, threshold = 20 like torch.nn.Softplus()


def render_pass_native(results, model, typ, freqs_xyz, rays, zs, t_embedded, t_next, t_prev, output_transient, flows,
                       noise_std, noise, saved, values):
    """render_pass with the field AND the compositing as native nodes (GPU, models without view directions, train
    mode).  `values`: the result dict the HIP forward of render_rays produced (the compositing node hands those
    numbers out instead of recomputing them).  What stays in torch is the glue between the nodes: far-masking of the
    flows, the warped query points, and sums of node outputs."""
    from . import composite_grad, field_grad
    n, s = zs.shape
    saved = saved or {}
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * zs[..., None]
    results[f"zs_{typ}"], results[f"xyzs_{typ}"] = zs, xyz
    raw = field_grad.field(model, xyz.reshape(-1, 3), freqs_xyz, t_embedded if output_transient else None, s,
                           True, output_transient, saved.get(typ))
    results[f"static_rgbs_{typ}"] = raw[:, 0:3].view(n, s, 3)
    raw_fw = raw_bw = f_fw = f_bw = None
    if output_transient:
        results[f"transient_rgbs_{typ}"] = raw[:, 4:7].view(n, s, 3)
        if flows:
            far = (zs > Z_FAR)[..., None]
            zero = torch.zeros((), device=zs.device)
            f_fw = results["transient_flows_fw"] = torch.where(far, zero, raw[:, 8:11].view(n, s, 3))
            f_bw = results["transient_flows_bw"] = torch.where(far, zero, raw[:, 11:14].view(n, s, 3))
            xyz_fw = results["xyzs_fw"] = xyz + f_fw
            xyz_bw = results["xyzs_bw"] = xyz + f_bw
            raw_fw = field_grad.field(model, xyz_fw.reshape(-1, 3), freqs_xyz, t_next, s, False, True,
                                      saved.get(f"{typ}_warp_fw"))
            raw_bw = field_grad.field(model, xyz_bw.reshape(-1, 3), freqs_xyz, t_prev, s, False, True,
                                      saved.get(f"{typ}_warp_bw"))
            results["xyzs_fw_bw"] = xyz_fw + torch.where(far, zero, raw_fw[:, 11:14].view(n, s, 3))
            results["xyzs_bw_fw"] = xyz_bw + torch.where(far, zero, raw_bw[:, 8:11].view(n, s, 3))
    results.update(composite_grad.composite(values, typ, raw, raw_fw, raw_bw, f_fw, f_bw, zs, xyz if flows else None,
                                            output_transient, noise_std, noise))
    if output_transient and flows:
        results["xyz_fw"] = results["xyz_fine"] + results["transient_flow_fw"]
        results["xyz_bw"] = results["xyz_fine"] + results["transient_flow_bw"]


def render_pass(results, model, typ, freqs_xyz, rays, zs, dir_embedded, a_embedded, t_embedded, t_next, t_prev,
                output_transient, flows, noise_std, noise, test_time, saved=None, values=None):
    from . import composite_grad, field_grad
    if (values is not None and not test_time and composite_grad.enabled() and field_grad.supported(model, zs)
            and (not flows or "fw" in flows and "bw" in flows)):
        return render_pass_native(results, model, typ, freqs_xyz, rays, zs, t_embedded, t_next, t_prev,
                                  output_transient, flows, noise_std, noise, saved, values)
    """One model pass (reference ``inference``): fills `results` with differentiable tensors.

    noise: dict with keys static / transient / warp_fw / warp_bw -> (N,S) standard normal draws (or None).
    Only the train-time branches are needed (gradients are never taken at test time).
    """
    n, s = zs.shape
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * zs[..., None]
    results[f"zs_{typ}"], results[f"xyzs_{typ}"] = zs, xyz
    saved = saved or {}
    f = query(model, xyz.reshape(-1, 3), freqs_xyz, dir_embedded, a_embedded, t_embedded, s,
              True, output_transient, flows, saved.get(typ))
    g = lambda k, c=None: f[k].view(n, s) if c is None else f[k].view(n, s, c)
    s_rgb = results[f"static_rgbs_{typ}"] = g("rgb_s", 3)
    far = (zs > Z_FAR)[..., None]
    if output_transient:
        t_rgb = results[f"transient_rgbs_{typ}"] = g("rgb_t", 3)
        if flows:
            f_fw = results["transient_flows_fw"] = torch.where(far, torch.zeros_like(g("fw", 3)), g("fw", 3))
            f_bw = results["transient_flows_bw"] = torch.where(far, torch.zeros_like(g("bw", 3)), g("bw", 3))

    deltas = zs[:, 1:] - zs[:, :-1]
    d_static = torch.cat([deltas, torch.full_like(deltas[:, :1], 100.0)], 1)
    d_trans = torch.cat([deltas, torch.full_like(deltas[:, :1], 1e-3)], 1)
    nz = lambda k: 0.0 if noise.get(k) is None else noise[k] * noise_std

    s_sig = results[f"static_sigmas_{typ}"] = _softplus(g("sigma_s") + nz("static"))
    alphas = 1 - torch.exp(-d_static * s_sig)
    if output_transient:
        s_alpha = alphas
        t_sig = results[f"transient_sigmas_{typ}"] = _softplus(g("sigma_t") + nz("transient"))
        t_alpha = 1 - torch.exp(-d_trans * t_sig)
        alphas = 1 - (1 - s_alpha) * (1 - t_alpha)

        if flows and not test_time:
            def warp(xyz_w, t_rows, head, key):
                fw_ = query(model, xyz_w.reshape(-1, 3), freqs_xyz, dir_embedded, a_embedded, t_rows, s,
                            False, True, [head], saved.get(f"{typ}_{key}"))
                rgb_w, sig_w = fw_["rgb_t"].view(n, s, 3), fw_["sigma_t"].view(n, s)
                flow_w = torch.where(far, torch.zeros_like(rgb_w), fw_[head].view(n, s, 3))
                al_w = 1 - torch.exp(-d_trans * _softplus(sig_w + nz(key)))
                al = 1 - (1 - s_alpha) * (1 - al_w)
                T = _excl_cumprod(1 - al)
                rgb = ((s_alpha * T)[..., None] * s_rgb).sum(1) + ((al_w * T)[..., None] * rgb_w).sum(1)
                return rgb, flow_w, al_w * T
            xyz_fw = results["xyzs_fw"] = xyz + f_fw
            results["rgb_fw"], flow_fw_bw, tw_fw = warp(xyz_fw, t_next, "bw", "warp_fw")
            xyz_bw = results["xyzs_bw"] = xyz + f_bw
            results["rgb_bw"], flow_bw_fw, tw_bw = warp(xyz_bw, t_prev, "fw", "warp_bw")
            results["xyzs_fw_bw"] = xyz_fw + flow_fw_bw
            results["xyzs_bw_fw"] = xyz_bw + flow_bw_fw

    T = _excl_cumprod(1 - alphas)
    weights = alphas * T
    if output_transient:
        s_w = results[f"static_weights_{typ}"] = s_alpha * T
        t_w = results[f"transient_weights_{typ}"] = t_alpha * T
        results[f"weights_{typ}"] = weights
    else:
        results[f"static_weights_{typ}"] = weights
    results[f"depth_{typ}"] = (weights * zs).sum(1)
    if not output_transient:
        results[f"rgb_{typ}"] = (weights[..., None] * s_rgb).sum(1)
        return
    t_map = (t_w[..., None] * t_rgb).sum(1)
    results[f"rgb_{typ}"] = (s_w[..., None] * s_rgb).sum(1) + t_map
    ta = results[f"transient_alpha_{typ}"] = t_w.sum(1)
    results[f"transient_rgb_{typ}"] = t_map + 0.8 * (1 - ta[:, None])
    so_w = s_alpha * _excl_cumprod(1 - s_alpha)
    results[f"_static_rgb_{typ}"] = (so_w[..., None] * s_rgb).sum(1)
    results[f"_static_depth_{typ}"] = (so_w * zs).sum(1)
    if flows:
        w3 = weights[..., None]
        results["xyz_fine"] = (w3 * xyz).sum(1)
        results["transient_flow_fw"] = (w3 * f_fw).sum(1)
        results["xyz_fw"] = results["xyz_fine"] + results["transient_flow_fw"]
        results["transient_flow_bw"] = (w3 * f_bw).sum(1)
        results["xyz_bw"] = results["xyz_fine"] + results["transient_flow_bw"]
        if not test_time and "disocc" in flows:
            occ_fw, occ_bw = (tw_fw - t_w).detach(), (tw_bw - t_w).detach()     # rendering.py:290-291
            results["disocc_fw"] = 1 - torch.abs(occ_fw.sum(1, keepdim=True))
            results["disoccs_fw"] = (1 - torch.abs(occ_fw))[..., None]
            results["disocc_bw"] = 1 - torch.abs(occ_bw.sum(1, keepdim=True))
            results["disoccs_bw"] = (1 - torch.abs(occ_bw))[..., None]
