"""Gradients for ``render_rays`` (SURVEY.md section 8f, row N1 -- first stage).

Forward values always come from the gfx950 kernels.  When gradients are needed the call is wrapped
in :class:`_RenderGrad`: ``backward`` re-evaluates the path with the differentiable torch expression of
:mod:`nsff_pl_amd.torch_path` at the SAME sample depths and random draws (recorded during the HIP
forward) and back-propagates the incoming gradients to the module parameters -- activation
checkpointing over one ``render_rays`` call.  ``sample_pdf`` and the disocclusion weights carry no
gradient in the reference either (``.detach()`` at rendering.py:336,343,290-291).
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
                               rec.get("saved"))
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
                           rec.get("saved"))
    return results


class _RenderGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *params):
        ctx.state = state
        ctx.n_params = len(params)
        keys, values = state["keys"], state["values"]
        outs = tuple(values[k] for k in keys)
        ctx.mark_non_differentiable(*[values[k] for k in keys if k in _NON_DIFF])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.state
        params = st["params"]
        with torch.enable_grad():
            res = recompute(st["models"], st["embeddings"], st["rays"], st["ts"], st["max_t"], st["rec"])
            outs, gouts = [], []
            for k, g in zip(st["keys"], grads):
                if g is None or k in _NON_DIFF or not res[k].requires_grad:
                    continue
                outs.append(res[k])
                gouts.append(g)
            pg = torch.autograd.grad(outs, params, gouts, allow_unused=True) if outs else [None] * len(params)
        return (None,) + tuple(pg)


def attach(results, models, embeddings, rays, ts, max_t, rec):
    """Return `results` with an autograd graph to the parameters (values unchanged)."""
    params = grad_parameters(models, embeddings)
    if not params:
        return results
    keys = sorted(results)
    state = dict(keys=keys, values=results, params=params, models=models, embeddings=embeddings,
                 rays=rays, ts=ts, max_t=max_t, rec=rec)
    outs = _RenderGrad.apply(state, *params)
    return dict(zip(keys, outs))
