"""Gradients for ``render_rays``.

Forward values always come from the gfx950 kernels.  When gradients are needed, the same quantities are also
expressed as a differentiable graph at the SAME sample depths and random draws (recorded during the HIP forward),
built from two kinds of native nodes:

* the field network (:mod:`nsff_pl_amd.field_grad`: training forward = ``render_rays``' own launches, backward =
  ``nsff_field_backward`` + ``nsff_weight_grad``), one node per field launch;
* the compositing of a pass (:mod:`nsff_pl_amd.composite_grad`: backward = ``nsff_composite_backward``).

What stays in torch between the nodes is glue -- far-masking of the flows, the warped query points, sums of node
outputs.  Every returned tensor keeps the kernel's value and takes its gradient route from that graph
(:class:`_Graft`).  ``sample_pdf`` and the disocclusion weights carry no gradient in the reference either
(``.detach()`` at rendering.py:336,343,290-291).

There is no torch fallback: a model or call the native nodes do not cover is refused by name
(:func:`why_not_differentiable`).  The all-torch expression of the same mathematics lives in ``tests/torch_path.py``
(test infrastructure).
"""
import torch

from . import composite_grad, field_grad

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


def recompute(models, embeddings, rays, ts, max_t, rec):
    """Differentiable graph of a recorded train-time call; returns the result dict (values handed out by the nodes)."""
    why = why_not_differentiable(models, rays, rec["flows"] if rec["output_transient"] else [])
    if why is not None:
        raise RuntimeError("render_rays cannot be differentiated for this call: " + why +
                           ".  Run it under torch.no_grad() / test_time=True, or freeze the parameters.")
    results = {}
    freqs_xyz = [float(f) for f in embeddings["xyz"].freqs]
    dir_embedded = rec.get("dir_embedded")        # the rows render_rays' forward used (view directions carry no gradient)
    t_embedded = None
    out_t = rec["output_transient"]
    if out_t:
        t_embedded = rec["t_embedded_override"] if rec["t_embedded_override"] is not None else embed_rows(embeddings["t"], ts)
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
    t_next = t_prev = None
    if out_t and flows:
        t_next = embed_rows(embeddings["t"], torch.clamp(ts + 1, max=max_t))
        t_prev = embed_rows(embeddings["t"], torch.clamp(ts - 1, min=0))
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
