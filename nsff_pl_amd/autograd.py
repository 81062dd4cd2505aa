"""Gradients for ``render_rays``.

Forward values always come from the gfx950 kernels.  When gradients are needed, the same quantities are also
expressed as a differentiable graph at the SAME sample depths and random draws (recorded during the HIP forward):
compositing / warping with the torch ops of :mod:`nsff_pl_amd.torch_path`, the field network as the native
node of :mod:`nsff_pl_amd.field_grad`; every returned tensor keeps the kernel's value and takes its gradient
route from that graph (:class:`_Graft`).  ``sample_pdf`` and the disocclusion weights carry no gradient in the
reference either (``.detach()`` at rendering.py:336,343,290-291).
"""
import torch

from . import torch_path

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


def recompute(models, embeddings, rays, ts, max_t, rec):
    """Differentiable re-evaluation of a recorded train-time call; returns the result dict."""
    results = {}
    freqs_xyz = [float(f) for f in embeddings["xyz"].freqs]
    dir_embedded = None
    if any(m.use_viewdir for m in models.values()):
        dir_embedded = torch_path.pos_embed(rec["view_dir"], [float(f) for f in embeddings["dir"].freqs])
    t_embedded = None
    out_t = rec["output_transient"]
    if out_t:
        t_embedded = rec["t_embedded_override"] if rec["t_embedded_override"] is not None else embeddings["t"](ts)
    if rec["N_importance"] > 0:
        torch_path.render_pass(results, models["coarse"], "coarse", freqs_xyz, rays, rec["zs_coarse"], dir_embedded,
                               None, t_embedded, None, None, out_t, [], rec["noise_std"],
                               dict(static=rec.get("coarse_static"), transient=rec.get("coarse_transient")), False,
                               rec.get("saved"), rec.get("values"))
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
    torch_path.render_pass(results, fine, "fine", freqs_xyz, rays, zs, dir_embedded, a_embedded, t_embedded,
                           t_next, t_prev, out_t, flows, rec["noise_std"],
                           dict(static=rec.get("fine_static"), transient=rec.get("fine_transient"),
                                warp_fw=rec.get("fine_warp_fw"), warp_bw=rec.get("fine_warp_bw")), False,
                           rec.get("saved"), rec.get("values"))
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

    The graph is built right away: compositing / warping as torch ops on the recorded depths and draws, the
    field network as the native node of :mod:`nsff_pl_amd.field_grad`, which -- when render_rays' own launches
    were training forwards (rec['saved']) -- launches nothing here.  No re-entrant backward, no host syncs: the
    whole step can be captured in a hipGraph."""
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
