#!/usr/bin/env python
"""CPU emulation of candidate dense-layer arithmetics through the WHOLE render path (test infrastructure: it drives
the numpy oracle with a replaced Linear), to decide what a faster field kernel may compute without leaving the
1e-4 parity bar.  Prints the worst max-norm relative error over all result keys against the reference golden.

  f16x3        W.x ~ Wh.xh + Wh.xl + Wl.xh, fp16 operands, fp32 accumulate            (what field_h3.hip does)
  f16x2w/x     drop Wl.xh / drop Wh.xl
  f16+fp8x2    main product in fp16, both 2^-11 corrections with fp8 (e4m3) operands, power-of-two scaled
               (an fp8 MFMA runs at twice the f16 rate: 2 instead of 3 MFMA units per product)
  f16+bf8x2    same with e5m2
  f16+mxfp8    the corrections on MX operands (OCP microscaling: blocks of 32 along K share a power-of-two scale, e4m3 elements --
               v_mfma_scale_f32_32x32x64_f8f6f4 at twice the f16 rate): 2 pipe-slots per product
  f16+mxfp6    the same with e2m3 elements (FP6 runs at the FP4 rate, four times f16): 1.5 pipe-slots per product
  f16+i8row    the corrections in 8-bit FIXED POINT (v_mfma_i32_*_i8, twice the f16 rate, exact int32 accumulation): the top 7 bits
               of Wh / xh relative to their row maximum, the top 7 bits of Wl / xl relative to 2^-11 of the same maximum -- so both
               corrections share one scale s_w[n] s_x[p] 2^-11 and one K-concatenated accumulator: 2 pipe-slots per product
  f16+i8tensor the same with ONE scale per tensor (no per-row maximum to find in the epilogue)
  f16x2w+i8 / f16x2x+i8   one correction in f16 (Wh.xl / Wl.xh), the other in 8-bit fixed point: 2.5 pipe-slots per product
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import parity  # noqa: E402
import scenes  # noqa: E402
import nsff_pl_amd as A  # noqa: E402
from oracle import nsff_oracle as orc  # noqa: E402


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def f8(x, dtype):
    """Round to an 8-bit float after scaling the tensor's max to ~2^7 (power of two), undo the scale."""
    amax = float(np.abs(x).max())
    if amax == 0:
        return x
    s = 2.0 ** np.floor(np.log2(128.0 / amax))
    t = torch.from_numpy((x * s).astype(np.float32)).to(dtype).float().numpy()
    return (t / s).astype(np.float32)


def mx(x, kind):
    """OCP MX emulation along the last axis (K): blocks of 32 share a power-of-two scale; elements e4m3 ('fp8') or e2m3 ('fp6')."""
    k = x.shape[-1]
    pad = (-k) % 32
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, pad)]).reshape(x.shape[:-1] + (-1, 32)).astype(np.float32)
    amax = np.abs(xp).max(-1, keepdims=True)
    top = 448.0 if kind == "fp8" else 7.5                 # largest element value of the format
    sc = 2.0 ** np.ceil(np.log2(np.where(amax > 0, amax, 1.0) / top))      # the block's maximum lands in (top / 2, top]
    v = xp / sc
    if kind == "fp8":
        q = torch.from_numpy(v.astype(np.float32)).to(torch.float8_e4m3fn).float().numpy()
    else:                                                 # e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa bits: normals 1.0 .. 7.5, subnormal step 0.125
        a = np.abs(v)
        ex = np.clip(np.floor(np.log2(np.maximum(a, 1e-30))), 0, 2)
        step = 2.0 ** (ex - 3)
        q = np.sign(v) * np.minimum(np.round(a / step) * step, 7.5)
    out = (q * sc).reshape(x.shape[:-1] + (-1,))
    return out[..., :k].astype(np.float32)


def i8_pair(hi, lo, axis_scale):
    """(hi, lo) fp32 arrays (value = hi + lo, |lo| <= 2^-11 |hi| elementwise) -> their 8-bit fixed-point tops on the scales s and
    s 2^-11, s = a power of two with max |hi| / s in [64, 128): per row (axis_scale='row') or per tensor."""
    amax = np.abs(hi).max(-1, keepdims=True) if axis_scale == "row" else np.full(hi.shape[:-1] + (1,), np.abs(hi).max(), np.float32)
    s = 2.0 ** (np.floor(np.log2(np.where(amax > 0, amax, 1.0))) - 6)          # |hi| / s < 128
    qh = np.clip(np.round(hi / s), -127, 127)
    ql = np.clip(np.round(lo / (s * 2.0 ** -11)), -127, 127)
    return qh.astype(np.float64), ql.astype(np.float64), s.astype(np.float64)


def make_lin(mode):
    def lin(p, name, x):
        W, b = p[name + ".weight"], p[name + ".bias"]
        if mode == "f32":
            return x @ W.T + b
        xh, Wh = f16(x), f16(W)
        xl, Wl = f16(x - xh), f16(W - Wh)
        mm = lambda a, w: (a.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)
        y = mm(xh, Wh)
        if mode == "f16x3":
            y = y + mm(xl, Wh) + mm(xh, Wl)
        elif mode == "f16x2w":
            y = y + mm(xl, Wh)
        elif mode == "f16x2x":
            y = y + mm(xh, Wl)
        elif mode in ("f16+fp8x2", "f16+bf8x2"):
            dt = torch.float8_e4m3fn if mode == "f16+fp8x2" else torch.float8_e5m2
            y = y + mm(f8(xl, dt), f8(Wh, dt)) + mm(f8(xh, dt), f8(Wl, dt))
        elif mode in ("f16+mxfp8", "f16+mxfp6"):
            kind = "fp8" if mode == "f16+mxfp8" else "fp6"
            y = y + mm(mx(xl, kind), mx(Wh, kind)) + mm(mx(xh, kind), mx(Wl, kind))
        elif mode in ("f16+i8row", "f16+i8tensor"):
            ax = "row" if mode == "f16+i8row" else "tensor"
            xq, xlq, sx = i8_pair(xh, xl, ax)
            wq, wlq, sw = i8_pair(Wh, Wl, ax)
            acc = xlq @ wq.T + xq @ wlq.T                                     # exact integer sums (one K-concatenated chain)
            y = y + (acc * (sx * sw.T) * 2.0 ** -11).astype(np.float32)
        elif mode in ("f16x2w+i8", "f16x2x+i8"):
            # one correction in f16, the other in 8-bit fixed point (row scales): 2.5 pipe-slots per product
            xq, xlq, sx = i8_pair(xh, xl, "row")
            wq, wlq, sw = i8_pair(Wh, Wl, "row")
            if mode == "f16x2w+i8":
                y = y + mm(xl, Wh) + ((xq @ wlq.T) * (sx * sw.T) * 2.0 ** -11).astype(np.float32)
            else:
                y = y + mm(xh, Wl) + ((xlq @ wq.T) * (sx * sw.T) * 2.0 ** -11).astype(np.float32)
        elif mode == "f16":
            pass
        else:
            raise ValueError(mode)
        return (y + b).astype(np.float32)
    return lin


def main():
    modes = sys.argv[1:] or ["f16x3", "f16x2w", "f16x2x", "f16+fp8x2", "f16+bf8x2", "f16+mxfp8", "f16+mxfp6", "f16+i8row", "f16+i8tensor", "f16"]
    names = ["g3_nsff_train", "g3b_nsff_train_gain3", "g7_nsff_train_noise", "g19_c2_subset", "g4_nsff_test"]
    orig = orc._lin
    for mode in modes:
        worst = {}
        for name in names:
            cfg, meta, rays, ts, models, emb, _, want = common.build_case(name, A.NeRF, A.PosEmbedding)
            orc._lin = make_lin(mode)
            try:
                draws = scenes.replay_draws(cfg, meta["draw_seed"]) if (cfg.get("perturb", 0) or cfg.get("noise_std", 0)) else None
                got = common.oracle_render(cfg, models, emb, rays, ts, draws=draws, zs_fine_override=want["zs_fine"])
            finally:
                orc._lin = orig
            errs = {k: parity.max_rel_err(got[k], want[k]) for k in want if k in got}
            k = max(errs, key=errs.get)
            plain = {q: v for q, v in errs.items() if q not in common.CHAINED_KEYS}
            worst[name] = (errs[k], k, max(plain.values()))
        print(f"{mode:11s} " + "  ".join(f"{n.split('_', 1)[0]}: all {w[0]:.1e} ({w[1]}) non-chained {w[2]:.1e}" for n, w in worst.items()))


if __name__ == "__main__":
    main()
