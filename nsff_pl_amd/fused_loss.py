"""``NeRFWLoss`` as six small HIP launches forward + two backward (csrc/loss.hip, ``nsff_nerfw_loss``).

The torch expression in :mod:`nsff_pl_amd.losses` is ~280 small kernels forward and ~300 backward -- a quarter of the
C4 training step's device time.  For the NSFF train-mode configuration (flows + disocclusion in the dict, <= 4096 rays on
the GPU) the same eleven scalars and their gradients w.r.t. the render dict come from the kernels instead, for every
reduction the reference's constructor can ask for: plain means, per-ray ``weights``, ``topk < 1`` (mean of the
int(topk * M) largest per-ray values, losses.py:162-169) and ``thickness > 1`` (dilated cross entropy, losses.py:91-95).
The torch expression stays what static-only models and CPU tensors use, and what the kernels are tested against
(tests/test_losses.py).  ``NSFF_FUSED_LOSS=0`` switches the kernels off.
"""
import os

import torch

from . import _lib

TERMS = ("col_l", "disp_l", "entropy_l", "cross_entropy_l", "flow_fw_l", "flow_bw_l", "pho_l", "cyc_l",
         "reg_temp_sm_l", "reg_min_l", "reg_sp_sm_l")
# (render-dict key, kernel argument, gradient argument or None)
_INPUTS = (("rgb_fine", "rgb_fine", "g_rgb_fine"), ("rgb_coarse", "rgb_coarse", "g_rgb_coarse"),
           ("depth_fine", "depth_fine", "g_depth_fine"), ("depth_coarse", "depth_coarse", "g_depth_coarse"),
           ("transient_weights_fine", "t_weights", "g_t_weights"), ("static_weights_fine", "s_weights", "g_s_weights"),
           ("xyz_fw", "xyz_fw", "g_xyz_fw"), ("xyz_bw", "xyz_bw", "g_xyz_bw"), ("rgb_fw", "rgb_fw", "g_rgb_fw"),
           ("rgb_bw", "rgb_bw", "g_rgb_bw"), ("xyzs_fw_bw", "xyzs_fw_bw", "g_xyzs_fw_bw"),
           ("xyzs_bw_fw", "xyzs_bw_fw", "g_xyzs_bw_fw"), ("xyzs_fw", "xyzs_fw", "g_xyzs_fw"), ("xyzs_bw", "xyzs_bw", "g_xyzs_bw"),
           ("disocc_fw", "disocc_fw", None), ("disocc_bw", "disocc_bw", None), ("disoccs_fw", "disoccs_fw", None),
           ("disoccs_bw", "disoccs_bw", None), ("xyzs_fine", "xyzs_fine", None))
_OPTIONAL = ("rgb_coarse", "depth_coarse")
MAX_RAYS = 4096
_CONST = {}


def enabled():
    return os.environ.get("NSFF_FUSED_LOSS", "1") != "0"


def applicable(loss, inputs, targets, kwargs):
    if not enabled() or not kwargs.get("output_transient_flow"):
        return False
    x = inputs.get("rgb_fine")
    w = kwargs.get("weights")
    if w is not None and not (torch.is_tensor(w) and x is not None and w.numel() == x.shape[0]):
        return False
    if x is None or not x.is_cuda or x.dtype != torch.float32 or not 1 <= x.shape[0] <= MAX_RAYS:
        return False
    need = [k for k, _, _ in _INPUTS if k not in _OPTIONAL]
    if any(k not in inputs for k in need) or ("rgb_coarse" in inputs) != ("depth_coarse" in inputs):
        return False
    return all(k in targets for k in ("rgbs", "disps", "ts", "cam_ids", "uv_fw", "uv_bw"))


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, hyper, *tens):
        names = cfg["names"]
        args = {arg: t.detach().contiguous() for (key, arg, _), t in zip(names, tens)}
        dev = tens[0].device
        stats = torch.empty(24, device=dev)
        terms = torch.empty(len(TERMS), device=dev)
        common = dict(cfg["targets"], hyper=hyper, stats=stats, weights=cfg["weights"],
                      per_ray=torch.empty(len(TERMS), cfg["n"], device=dev), coef=torch.empty(len(TERMS), cfg["n"], device=dev),
                      topk=cfg["topk"], thickness=cfg["thickness"])
        _lib.nerfw_loss(1, cfg["n"], cfg["s"], cfg["n_keep"], cfg["n_frames"], cfg["max_t"], terms=terms, **args, **common)
        ctx.cfg, ctx.args, ctx.common = cfg, args, common
        return terms

    @staticmethod
    def backward(ctx, g_terms):
        cfg, args = ctx.cfg, ctx.args
        grads = {}
        for (key, arg, garg), need in zip(cfg["names"], ctx.needs_input_grad[2:]):
            if garg is not None:
                grads[garg] = torch.empty_like(args[arg])        # the kernel writes every element
        _lib.nerfw_loss(2, cfg["n"], cfg["s"], cfg["n_keep"], cfg["n_frames"], cfg["max_t"], term_w=g_terms.contiguous(),
                        **args, **ctx.common, **grads)
        out = []
        for (key, arg, garg), need in zip(cfg["names"], ctx.needs_input_grad[2:]):
            out.append(grads[garg] if (garg is not None and need) else None)
        return (None, None) + tuple(out)


def nerfw_loss(loss, inputs, targets, kwargs):
    """The eleven reduced terms of ``NeRFWLoss.forward`` (reference losses.py:72-171) as a dict of 0-d tensors."""
    x = inputs["rgb_fine"]
    dev = x.device
    n, s = inputs["xyzs_fine"].shape[:2]
    names = [(k, a, g) for k, a, g in _INPUTS if k in inputs]
    tens = [inputs[k].reshape(inputs[k].shape[0], -1) if inputs[k].dim() == 3 and inputs[k].shape[-1] == 1 else inputs[k]
            for k, _, _ in names]

    def scalar(v):
        if torch.is_tensor(v):
            return v.to(device=dev, dtype=torch.float32).reshape(())
        key = (dev, float(v))
        if key not in _CONST:                 # uploaded once per value (the eager warm-up precedes any graph capture)
            _CONST[key] = torch.tensor(float(v), device=dev)
        return _CONST[key]
    ramp = kwargs.get("epoch_ramp", min(kwargs.get("epoch", 0) / 10, 1.0))
    hyper = torch.stack([scalar(loss.lambda_geo_d), scalar(loss.lambda_geo_f), scalar(ramp) * (loss.lambda_ent / 5),
                         scalar(loss.lambda_reg), scalar(loss.lambda_ent)])
    tg = dict(rgbs=targets["rgbs"].contiguous().float(), disps=targets["disps"].contiguous().float(),
              ts=targets["ts"].contiguous().long(), cam_ids=targets["cam_ids"].contiguous().long(),
              uv_fw=targets["uv_fw"].contiguous().float(), uv_bw=targets["uv_bw"].contiguous().float(),
              Ks=loss.Ks.contiguous().float().reshape(-1, 3, 3), Ps=loss.Ps.contiguous().float())
    w = kwargs.get("weights")
    cfg = dict(names=names, targets=tg, n=n, s=s, n_keep=int(s * loss.z_far), n_frames=int(loss.Ps.shape[1]),
               max_t=int(loss.max_t), topk=float(loss.topk), thickness=int(loss.thickness),
               weights=None if w is None else w.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous())
    terms = _LossFn.apply(cfg, hyper, *tens)
    return LossTerms(terms)


class LossTerms(dict):
    """The loss dict of ``NeRFWLoss.forward`` ({term: 0-d tensor}) backed by ONE vector: ``sum(d.values())`` -- what the
    reference's training_step does (train.py:184) -- works as on any dict, but costs eleven select / add nodes forward and
    thirty-odd tiny kernels backward; :meth:`total` is the same number through a single sum node."""

    def __init__(self, terms):
        super().__init__({k: terms[i] for i, k in enumerate(TERMS)})
        self.vector = terms

    def total(self):
        return self.vector.sum()
