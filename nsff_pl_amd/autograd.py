"""Gradients for ``render_rays``.

Forward values always come from the gfx950 kernels.  When gradients are needed, the same quantities are also
expressed as a differentiable graph at the SAME sample depths and random draws (recorded during the HIP forward),
built from two kinds of native nodes:

* the field network (:mod:`nsff_pl_amd.field_grad`: training forward = ``render_rays``' own launches, backward =
  ``nsff_field_backward`` + ``nsff_weight_grad``), one node per field launch;
* the compositing of a pass (:mod:`nsff_pl_amd.composite_grad`: backward = ``nsff_composite_backward``).

The glue between them -- far-masking of the flows, the warped query points, the cycle points -- is a third node
(:class:`_FlowFn`, backward = ``nsff_flow_grad``) and a tail of the compositing node; what stays in torch is the per-ray
``xyz + expected flow`` sums and the embedding gathers.  Every returned tensor keeps the kernel's value and takes its gradient route from that graph
(:class:`_Graft`).  ``sample_pdf`` and the disocclusion weights carry no gradient in the reference either
(``.detach()`` at rendering.py:336,343,290-291).

There is no torch fallback: a model or call the native nodes do not cover is refused by name
(:func:`why_not_differentiable`).  The all-torch expression of the same mathematics lives in ``tests/torch_path.py``
(test infrastructure).
"""
import torch

from . import _lib, composite_grad, field_grad

Z_FAR = 0.95
# outputs that do not depend on any parameter
_NON_DIFF = ("zs_coarse", "xyzs_coarse", "zs_fine", "xyzs_fine")


def grad_parameters(models, embeddings):
    """Parameters a render_rays call can send gradients to (order is part of the Function signature)."""
    params = []
    for key in sorted(models):
        params += [p for p in models[key].parameters() if p.requires_grad]
    for key in ("t", "a"):
        if key in embeddings and isinstance(embeddings[key], torch.nn.Module):
            params += [p for p in embeddings[key].parameters() if p.requires_grad]
    return params


class _EmbedRows(torch.autograd.Function):
    """weight[idx] with a dense backward: one_hot(idx)^T . grad as a small GEMM.  torch's embedding backward takes
    ~200 us for 1024 rows that hit only 30 distinct codes (atomics on a handful of rows); the render graph needs it
    three times per step (t, t+1, t-1)."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.n = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        onehot = (idx[None, :] == torch.arange(ctx.n, device=idx.device)[:, None]).to(grad.dtype)      # (rows, N)
        return onehot @ grad, None


class _TimeRows(torch.autograd.Function):
    """(E[ts], E[clamp(ts + 1, max=max_t)], E[clamp(ts - 1, min=0)]) of the time-code table -- the three gathers of a training
    step (rendering.py:162,218,224) -- as one node: forward = one index_select + ``nsff_time_rows``, backward = one
    ``nsff_time_rows_backward`` launch (deterministic) instead of three one-hot GEMMs and two adds."""

    @staticmethod
    def forward(ctx, weight, ts, max_t):
        ts = ts.contiguous()
        ctx.save_for_backward(ts)
        ctx.max_t, ctx.n_table = int(max_t), weight.shape[0]
        ctx.set_materialize_grads(False)
        w = weight.detach()
        nxt, prv = _lib.time_rows(w, ts, max_t)
        return w.index_select(0, ts), nxt, prv

    @staticmethod
    def backward(ctx, g_cur, g_next, g_prev):
        if g_cur is None and g_next is None and g_prev is None:
            return None, None, None
        (ts,) = ctx.saved_tensors
        return _lib.time_rows_backward(g_cur, g_next, g_prev, ts, ctx.max_t, ctx.n_table), None, None


def _plain_table(module, idx):
    return (isinstance(module, torch.nn.Embedding) and module.padding_idx is None and module.max_norm is None
            and not module.sparse and module.weight.is_cuda and module.weight.dtype == torch.float32 and idx.dim() == 1)


def time_rows(module, ts, max_t, neighbours):
    """(module(ts), module(clamp(ts + 1, max=max_t)), module(clamp(ts - 1, min=0))) -- the last two None unless `neighbours`."""
    if _plain_table(module, ts) and neighbours and ts.dtype == torch.int64:
        return _TimeRows.apply(module.weight, ts, max_t)
    cur = embed_rows(module, ts)
    if not neighbours:
        return cur, None, None
    return cur, embed_rows(module, torch.clamp(ts + 1, max=max_t)), embed_rows(module, torch.clamp(ts - 1, min=0))


def embed_rows(module, idx):
    """module(idx) -- through the dense-backward gather when `module` is a plain nn.Embedding on the GPU."""
    if (isinstance(module, torch.nn.Embedding) and module.padding_idx is None and module.max_norm is None
            and not module.sparse and module.weight.is_cuda and idx.dim() == 1 and module.weight.shape[0] <= 4096):
        return _EmbedRows.apply(module.weight, idx)
    return module(idx)


def why_not_differentiable(models, rays, flows):
    """None, or the reason the native backward cannot serve this call."""
    if not rays.is_cuda:
        return "rays are not on the GPU (the backward, like the forward, runs only on the HIP kernels)"
    if not field_grad.enabled() or not composite_grad.enabled():
        return "NSFF_NATIVE_BACKWARD=0 / NSFF_NATIVE_COMPOSITE_BWD=0 disable the native nodes (debug switches)"
    for key, m in models.items():
        why = field_grad.why_unsupported(m)
        if why is not None:
            return f"models['{key}']: {why}"
    if flows and not ("fw" in flows and "bw" in flows):
        return f"output_transient_flow={list(flows)}: the training path needs both 'fw' and 'bw'"
    return None


class _FlowFn(torch.autograd.Function):
    """The scene-flow glue between the fine field query and its warped re-queries (reference rendering.py:187-188, 218, 224) as
    ONE node: forward hands out the flows (zeroed beyond z = 0.95) and the warped points the HIP forward already wrote
    (nothing is launched) -- the warped points three times, once per consumer (result key, re-query, cycle point), so that
    the engine never has to add their cotangents; backward is one ``nsff_flow_grad`` launch that sums whatever arrived,
    masks it and writes the (P,16) record gradient.  As torch ops (where / slice / add and their backward: zeros, copy,
    where, add) the same glue was ~35 small kernels per step."""

    @staticmethod
    def forward(ctx, cfg, raw):
        v = cfg["values"]
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        again = lambda t: t.detach().view_as(t)
        return ((again(v["transient_flows_fw"]), again(v["transient_flows_bw"]))
                + tuple(again(v["xyzs_fw"]) for _ in range(3)) + tuple(again(v["xyzs_bw"]) for _ in range(3)))

    @staticmethod
    def backward(ctx, g_f_fw, g_f_bw, *gx):
        g_fw = [g for g in (g_f_fw,) + tuple(gx[:3]) if g is not None]
        g_bw = [g for g in (g_f_bw,) + tuple(gx[3:]) if g is not None]
        if not g_fw and not g_bw:
            return None, None
        zs = ctx.cfg["zs"]
        d_raw = torch.empty(zs.numel(), 16, device=zs.device)
        _lib.flow_grad(zs, Z_FAR, d_raw, False, 8 if g_fw else -1, g_fw, 11 if g_bw else -1, g_bw)
        return None, d_raw


def _render_pass(results, model, typ, freqs_xyz, rays, zs, dir_embedded, a_embedded, t_embedded, t_next, t_prev,
                 output_transient, flows, noise_std, noise, saved, values):
    """One model pass (reference ``inference``, rendering.py:83-300) as native nodes + glue.  `values`: the result
    dict the HIP forward of render_rays produced (the compositing node hands those numbers out again)."""
    n, s = zs.shape
    saved = saved or {}
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * zs[..., None]
    results[f"zs_{typ}"], results[f"xyzs_{typ}"] = zs, xyz
    side = dict(dir_rows=dir_embedded if model.use_viewdir else None,
                a_rows=a_embedded if (model.use_viewdir and model.in_channels_a > 0) else None)
    raw = field_grad.field(model, xyz.reshape(-1, 3), freqs_xyz, t_embedded if output_transient else None, s,
                           True, output_transient, saved.get(typ), **side)
    results[f"static_rgbs_{typ}"] = raw[:, 0:3].view(n, s, 3)
    raw_fw = raw_bw = f_fw = f_bw = cyc_fw = cyc_bw = None
    if output_transient:
        results[f"transient_rgbs_{typ}"] = raw[:, 4:7].view(n, s, 3)
        if flows:
            (f_fw, f_bw, results["xyzs_fw"], xyz_fw, cyc_fw, results["xyzs_bw"], xyz_bw, cyc_bw) = \
                _FlowFn.apply(dict(values=values, zs=zs.contiguous()), raw)
            results["transient_flows_fw"], results["transient_flows_bw"] = f_fw, f_bw
            raw_fw = field_grad.field(model, xyz_fw.reshape(-1, 3), freqs_xyz, t_next, s, False, True,
                                      saved.get(f"{typ}_warp_fw"))
            raw_bw = field_grad.field(model, xyz_bw.reshape(-1, 3), freqs_xyz, t_prev, s, False, True,
                                      saved.get(f"{typ}_warp_bw"))
    # (the cycle points xyzs_fw_bw / xyzs_bw_fw = warped point + masked flow of the re-query come out of the compositing node)
    results.update(composite_grad.composite(values, typ, raw, raw_fw, raw_bw, f_fw, f_bw, zs, xyz if flows else None,
                                            output_transient, noise_std, noise, cyc_fw, cyc_bw))
    if output_transient and flows:
        results["xyz_fw"] = results["xyz_fine"] + results["transient_flow_fw"]
        results["xyz_bw"] = results["xyz_fine"] + results["transient_flow_bw"]


def recompute(models, embeddings, rays, ts, max_t, rec):
    """Differentiable graph of a recorded train-time call; returns the result dict (values handed out by the nodes)."""
    why = why_not_differentiable(models, rays, rec["flows"] if rec["output_transient"] else [])
    if why is not None:
        raise RuntimeError("render_rays cannot be differentiated for this call: " + why +
                           ".  Run it under torch.no_grad() / test_time=True, or freeze the parameters.")
    results = {}
    freqs_xyz = [float(f) for f in embeddings["xyz"].freqs]
    dir_embedded = rec.get("dir_embedded")        # the rows render_rays' forward used (view directions carry no gradient)
    t_embedded = t_next = t_prev = None
    out_t = rec["output_transient"]
    if out_t:
        t_embedded, t_next, t_prev = time_rows(embeddings["t"], ts, max_t, bool(rec["flows"]))
        if rec["t_embedded_override"] is not None:
            t_embedded = rec["t_embedded_override"]
    if rec["N_importance"] > 0:
        _render_pass(results, models["coarse"], "coarse", freqs_xyz, rays, rec["zs_coarse"], dir_embedded,
                     None, t_embedded, None, None, out_t, [], rec["noise_std"],
                     dict(static=rec.get("coarse_static"), transient=rec.get("coarse_transient")),
                     rec.get("saved"), rec["values"])
    fine = models["fine"]
    a_embedded = None
    if fine.encode_appearance:
        a_embedded = rec["a_embedded_override"] if rec["a_embedded_override"] is not None else embed_rows(embeddings["a"], ts)
    flows = rec["flows"]
    zs = rec["zs_fine"] if rec["N_importance"] > 0 else rec["zs_coarse"]
    _render_pass(results, fine, "fine", freqs_xyz, rays, zs, dir_embedded, a_embedded, t_embedded,
                 t_next, t_prev, out_t, flows, rec["noise_std"],
                 dict(static=rec.get("fine_static"), transient=rec.get("fine_transient"),
                      warp_fw=rec.get("fine_warp_fw"), warp_bw=rec.get("fine_warp_bw")),
                 rec.get("saved"), rec["values"])
    return results


class _Graft(torch.autograd.Function):
    """value (computed by the HIP render kernels) with the gradient route of `carrier` (the same quantity expressed
    by the differentiable graph): forward returns the kernel's number untouched, backward hands the incoming gradient
    to the carrier."""

    @staticmethod
    def forward(ctx, value, carrier):
        return value.detach().view_as(value)

    @staticmethod
    def backward(ctx, grad):
        return None, grad


def attach(results, models, embeddings, rays, ts, max_t, rec):
    """Return `results` with an autograd graph to the parameters (values unchanged).

    The graph is built right away from the native nodes, which -- when render_rays' own launches were training
    forwards (rec['saved']) -- launch nothing here.  No re-entrant backward, no host syncs: the whole step can be
    captured in a hipGraph."""
    if not grad_parameters(models, embeddings):
        return results
    rec = dict(rec, values=results)          # the compositing node hands these numbers out again
    with torch.enable_grad():
        res = recompute(models, embeddings, rays, ts, max_t, rec)
    out = {}
    for k, v in results.items():
        if k in _NON_DIFF or k not in res or not res[k].requires_grad:
            out[k] = v
        else:
            out[k] = _Graft.apply(v, res[k].view_as(v))
    return out
