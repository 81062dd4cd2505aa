"""TEST INFRASTRUCTURE, NOT THE PRODUCT: a differentiable torch expression of the render path.

The product (``nsff_pl_amd``) takes gradients only through its HIP nodes (``field_grad``, ``composite_grad``) and
refuses what they do not cover.  This module re-expresses the same mathematics with ordinary torch ops so that
tests can (a) check the autograd graph of the product against float64 autograd on the CPU (``recompute``, used by
tests/test_gradients.py with the reference's gradient goldens) and (b) A/B the native compositing backward against
autograd of the elementwise expression on the GPU (``field_fn`` = the native field node).

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


def query(model, xyz, freqs_xyz, dir_embedded, a_embedded, t_embedded, s, static, transient, flows, field_fn=None):
    """Field outputs for (P,3) points, `s` consecutive points per ray.  field_fn (optional): a replacement returning
    the (P,16) raw record -- tests pass the product's native node to isolate the compositing."""
    if field_fn is not None:
        raw = field_fn(model, xyz, freqs_xyz, t_embedded if transient else None, s, static, transient,
                       dir_embedded, a_embedded)
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


def render_pass(results, model, typ, freqs_xyz, rays, zs, dir_embedded, a_embedded, t_embedded, t_next, t_prev,
                output_transient, flows, noise_std, noise, test_time, field_fn=None):
    """One model pass (reference ``inference``): fills `results` with differentiable tensors.

    noise: dict with keys static / transient / warp_fw / warp_bw -> (N,S) standard normal draws (or None).
    Only the train-time branches are needed (gradients are never taken at test time).
    """
    n, s = zs.shape
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * zs[..., None]
    results[f"zs_{typ}"], results[f"xyzs_{typ}"] = zs, xyz
    f = query(model, xyz.reshape(-1, 3), freqs_xyz, dir_embedded, a_embedded, t_embedded, s,
              True, output_transient, flows, field_fn)
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
                            False, True, [head], field_fn)
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


def recompute(models, embeddings, rays, ts, max_t, rec, field_fn=None):
    """Differentiable torch evaluation of a recorded train-time call (same record as nsff_pl_amd.autograd.recompute:
    depths, draws, flags); returns the result dict."""
    results = {}
    freqs_xyz = [float(f) for f in embeddings["xyz"].freqs]
    dir_embedded = None
    if any(m.use_viewdir for m in models.values()):
        dir_embedded = pos_embed(rec["view_dir"], [float(f) for f in embeddings["dir"].freqs])
    t_embedded = None
    out_t = rec["output_transient"]
    if out_t:
        t_embedded = rec["t_embedded_override"] if rec["t_embedded_override"] is not None else embeddings["t"](ts)
    if rec["N_importance"] > 0:
        render_pass(results, models["coarse"], "coarse", freqs_xyz, rays, rec["zs_coarse"], dir_embedded,
                    None, t_embedded, None, None, out_t, [], rec["noise_std"],
                    dict(static=rec.get("coarse_static"), transient=rec.get("coarse_transient")), False, field_fn)
    fine = models["fine"]
    a_embedded = None
    if fine.encode_appearance:
        a_embedded = rec["a_embedded_override"] if rec["a_embedded_override"] is not None else embeddings["a"](ts)
    flows = rec["flows"]
    t_next = t_prev = None
    if out_t and flows:
        t_next = embeddings["t"](torch.clamp(ts + 1, max=max_t))
        t_prev = embeddings["t"](torch.clamp(ts - 1, min=0))
    zs = rec["zs_fine"] if rec["N_importance"] > 0 else rec["zs_coarse"]
    render_pass(results, fine, "fine", freqs_xyz, rays, zs, dir_embedded, a_embedded, t_embedded,
                t_next, t_prev, out_t, flows, rec["noise_std"],
                dict(static=rec.get("fine_static"), transient=rec.get("fine_transient"),
                     warp_fw=rec.get("fine_warp_fw"), warp_bw=rec.get("fine_warp_bw")), False, field_fn)
    return results


def folded_grads_reference(model, static, transient, meta, plist, result, bias_of):
    """The algebra of nsff_pl_amd.field_grad._folded_grads in torch (any device / dtype): *_xyz_encoding_final is a Linear without
    activation (reference nerf.py:170,195), so with G = sum_p dpre_p (x) h_p and gb = sum_p dpre_p of the FOLDED heads
        dW_head = G W_final^T + gb (x) b_final,  db_head = gb,  dW_final = W_head^T G,  db_final = W_head^T gb.
    Test infrastructure: the CPU suite holds it to float64 autograd of the two layers, the GPU suite holds the product's HIP
    kernels (nsff_fold_grads, nsff_fold_grads_dense) to it."""
    from nsff_pl_amd import field_grad as fg
    viewdir = bool(model.use_viewdir and static)
    index = {id(q): i for i, q in enumerate(plist)}
    res = {tag: i for i, tag in enumerate(meta)}
    out = {}
    for t in ([0] if static else []) + ([1] if transient else []):
        prefix = "static" if t == 0 else "transient"
        fin = fg._lin(getattr(model, f"{prefix}_xyz_encoding_final"))
        w_f, b_f = fin.weight.detach(), fin.bias.detach()
        if t == 0 and viewdir:
            ih, ix = res[("dir_h", 0, 0)], res[("dir_x", 0, 0)]
            g, gb = result(ih), bias_of(ih)
            layer = fg._lin(model.static_dir_encoding)
            w_dh = layer.weight.detach()[:, :256]
            n_side = model.in_channels_dir + model.in_channels_a
            out[index[id(layer.weight)]] = torch.cat([torch.addmm(torch.outer(gb, b_f), g, w_f.t()), result(ix)[:, :n_side]], 1)
            out[index[id(layer.bias)]] = gb.clone()
            out[index[id(fin.weight)]] = w_dh.t() @ g
            out[index[id(fin.bias)]] = w_dh.t() @ gb
            continue
        i = res[("head", t, 0)]
        hw, hb = result(i), bias_of(i)
        g, gb = hw[0:16] + hw[16:32], hb[0:16] + hb[16:32]           # fp16 value + rounding remainder rows
        heads = fg._fold_heads(model, t)
        n_rows = heads[-1][2]
        w_h = torch.cat([fg._lin(m).weight.detach() for m, _, _ in heads], 0)
        g_r, gb_r = g[:n_rows], gb[:n_rows]
        d_heads = torch.addmm(torch.outer(gb_r, b_f), g_r, w_f.t())
        k = 0
        for m, a, b in heads:
            lin = fg._lin(m)
            out[index[id(lin.weight)]] = d_heads[k:k + (b - a)]
            out[index[id(lin.bias)]] = gb[a:b].clone()
            k += b - a
        out[index[id(fin.weight)]] = w_h.t() @ g_r
        out[index[id(fin.bias)]] = w_h.t() @ gb_r
    return out
