"""Native compositing node of the backward graph.

One ``autograd.Function`` per render pass: forward hands out the values the compositing kernel of ``render_rays``
already produced (nothing is launched), backward is ``nsff_composite_backward`` (csrc/rays_bwd.hip) -- one launch
instead of the ~250 small torch kernels the elementwise expression of the same mathematics needs
(reference models/rendering.py:122-140,200-298).
"""
import os

import torch

from . import _lib

# (result-key template, per-sample?, gradient argument of nsff_composite_backward)
_OUTS_STATIC = [("static_sigmas_{t}", "g_static_sigmas"), ("static_weights_{t}", "g_static_weights"),
                ("depth_{t}", "g_depth"), ("rgb_{t}", "g_rgb")]
_OUTS_TRANSIENT = [("static_sigmas_{t}", "g_static_sigmas"), ("transient_sigmas_{t}", "g_transient_sigmas"),
                   ("static_weights_{t}", "g_static_weights"), ("transient_weights_{t}", "g_transient_weights"),
                   ("weights_{t}", "g_weights"), ("depth_{t}", "g_depth"), ("rgb_{t}", "g_rgb"),
                   ("transient_alpha_{t}", "g_transient_alpha"), ("transient_rgb_{t}", "g_transient_rgb"),
                   ("_static_rgb_{t}", "g_so_rgb"), ("_static_depth_{t}", "g_so_depth")]
_OUTS_FLOW = [("xyz_fine", "g_xyz_exp"), ("transient_flow_fw", "g_flow_fw_exp"), ("transient_flow_bw", "g_flow_bw_exp")]
_OUTS_WARP = [("rgb_fw", "g_rgb_fw"), ("rgb_bw", "g_rgb_bw")]


CYCLE_KEYS = ("xyzs_fw_bw", "xyzs_bw_fw")
Z_FAR = 0.95


def enabled():
    return os.environ.get("NSFF_NATIVE_COMPOSITE_BWD", "1") != "0"


def output_spec(typ, transient, flows, warps):
    spec = list(_OUTS_TRANSIENT if transient else _OUTS_STATIC)
    if transient and flows:
        spec += _OUTS_FLOW
        if warps:
            spec += _OUTS_WARP
    return [(k.format(t=typ), g) for k, g in spec]


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, raw, raw_fw, raw_bw, f_fw, f_bw, cyc_fw, cyc_bw):
        values = cfg["values"]
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)         # outputs nobody differentiates arrive as None -> NULL, not as zero tensors
        ctx.save_for_backward(*[t for t in (raw, raw_fw, raw_bw, f_fw, f_bw) if t is not None])
        ctx.present = [t is not None for t in (raw, raw_fw, raw_bw, f_fw, f_bw)]
        ctx.cycle = cyc_fw is not None
        keys = [k for k, _ in cfg["spec"]] + (list(CYCLE_KEYS) if ctx.cycle else [])
        return tuple(values[k].detach().view_as(values[k]) for k in keys)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        if all(g is None for g in grads):
            return (None,) * 8
        g_cyc_fw = g_cyc_bw = None
        if ctx.cycle:
            grads, (g_cyc_fw, g_cyc_bw) = grads[:-2], grads[-2:]
        it = iter(ctx.saved_tensors)
        raw, raw_fw, raw_bw, f_fw, f_bw = [next(it) if p else None for p in ctx.present]
        n, s = cfg["zs"].shape
        dev = raw.device
        tens = dict(raw=raw, raw_fw=raw_fw, raw_bw=raw_bw, zs=cfg["zs"], noise_static=cfg["noise"].get("static"),
                    noise_transient=cfg["noise"].get("transient"), noise_fw=cfg["noise"].get("warp_fw"),
                    noise_bw=cfg["noise"].get("warp_bw"))
        if cfg["noise_std"] == 0:
            for k in ("noise_static", "noise_transient", "noise_fw", "noise_bw"):
                tens[k] = None
        if f_fw is not None:
            tens.update(xyz=cfg["xyz"], f_fw=f_fw.contiguous(), f_bw=f_bw.contiguous())
        for (_, gname), g in zip(cfg["spec"], grads):
            tens[gname] = None if g is None else g.contiguous()
        tens["scratch"] = torch.empty(n, s, 4, device=dev)
        d_raw = torch.empty_like(raw)
        tens["d_raw"] = d_raw
        d_raw_fw = d_raw_bw = d_f_fw = d_f_bw = None
        if raw_fw is not None:
            d_raw_fw, d_raw_bw = torch.empty_like(raw_fw), torch.empty_like(raw_bw)
            tens.update(d_raw_fw=d_raw_fw, d_raw_bw=d_raw_bw)
        if f_fw is not None:
            d_f_fw, d_f_bw = torch.empty_like(f_fw), torch.empty_like(f_bw)
            tens.update(d_f_fw=d_f_fw, d_f_bw=d_f_bw)
        flow_mode = 0 if f_fw is None else (2 if raw_fw is not None else 1)
        _lib.composite_backward(n, s, cfg["transient"], flow_mode, cfg["noise_std"], **tens)
        # cycle points (rendering.py:226-232): xyzs_fw_bw = x_fw + [z <= 0.95] bw(x_fw), xyzs_bw_fw = x_bw + [z <= 0.95] fw(x_bw):
        # the cotangent goes to the warped point as it is and, masked, into the flow columns of the re-query's record
        if g_cyc_fw is not None:
            _lib.flow_grad(cfg["zs"], Z_FAR, d_raw_fw, True, 11, [g_cyc_fw.reshape(-1, 3)])
        if g_cyc_bw is not None:
            _lib.flow_grad(cfg["zs"], Z_FAR, d_raw_bw, True, 8, [g_cyc_bw.reshape(-1, 3)])
        return None, d_raw, d_raw_fw, d_raw_bw, d_f_fw, d_f_bw, g_cyc_fw, g_cyc_bw


def composite(values, typ, raw, raw_fw, raw_bw, f_fw, f_bw, zs, xyz, transient, noise_std, noise, cyc_fw=None, cyc_bw=None):
    """Differentiable per-ray / per-sample compositing outputs of one pass, as a dict keyed like render_rays.
    cyc_fw / cyc_bw: the warped points (their own copies out of autograd._FlowFn); with them the node also hands out the
    cycle points ``xyzs_fw_bw`` / ``xyzs_bw_fw``."""
    spec = output_spec(typ, transient, f_fw is not None, raw_fw is not None)
    if cyc_fw is not None and raw_fw is None:
        raise ValueError("cycle points need the warped re-queries")
    cfg = dict(values=values, spec=spec, zs=zs.contiguous(), xyz=None if xyz is None else xyz.contiguous(),
               transient=bool(transient), noise_std=float(noise_std), noise=noise)
    outs = _CompositeFn.apply(cfg, raw, raw_fw, raw_bw, f_fw, f_bw, cyc_fw, cyc_bw)
    keys = [k for k, _ in spec] + (list(CYCLE_KEYS) if cyc_fw is not None else [])
    return dict(zip(keys, outs))
