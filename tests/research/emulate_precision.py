#!/usr/bin/env python
"""CPU emulation of candidate dense-layer arithmetics through the WHOLE render path (test infrastructure: it drives
the numpy oracle with a replaced Linear), to decide what a faster field kernel may compute without leaving the
1e-4 parity bar.  Prints the worst max-norm relative error over all result keys against the reference golden.

  f16x3        W.x ~ Wh.xh + Wh.xl + Wl.xh, fp16 operands, fp32 accumulate            (what field_h3.hip does)
  f16x2w/x     drop Wl.xh / drop Wh.xl
  f16+fp8x2    main product in fp16, both 2^-11 corrections with fp8 (e4m3) operands, power-of-two scaled
               (an fp8 MFMA runs at twice the f16 rate: 2 instead of 3 MFMA units per product)
  f16+bf8x2    same with e5m2
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
        elif mode == "f16":
            pass
        else:
            raise ValueError(mode)
        return (y + b).astype(np.float32)
    return lin


def main():
    modes = sys.argv[1:] or ["f16x3", "f16x2w", "f16x2x", "f16+fp8x2", "f16+bf8x2", "f16"]
    names = ["g3_nsff_train", "g3b_nsff_train_gain3", "g4_nsff_test"]
    orig = orc._lin
    for mode in modes:
        worst = {}
        for name in names:
            cfg, meta, rays, ts, models, emb, _, want = common.build_case(name, A.NeRF, A.PosEmbedding)
            orc._lin = make_lin(mode)
            try:
                got = common.oracle_render(cfg, models, emb, rays, ts, zs_fine_override=want["zs_fine"])
            finally:
                orc._lin = orig
            errs = {k: parity.max_rel_err(got[k], want[k]) for k in want if k in got}
            k = max(errs, key=errs.get)
            plain = {q: v for q, v in errs.items() if q not in common.CHAINED_KEYS}
            worst[name] = (errs[k], k, max(plain.values()))
        print(f"{mode:11s} " + "  ".join(f"{n.split('_', 1)[0]}: all {w[0]:.1e} ({w[1]}) non-chained {w[2]:.1e}" for n, w in worst.items()))


if __name__ == "__main__":
    main()
