"""Which training-forward configurations write the value planes differently when the remainder planes are requested (debugging aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nsff_pl_amd as A                         # noqa: E402
from nsff_pl_amd import _lib, config, field_grad         # noqa: E402

dev = torch.device("cuda:0")
freqs = [2.0 ** k for k in range(10)]
for D, skips, n_rays, s, tp in ((8, [4], 21, 64, 131), (8, [4], 30, 64, 131), (8, [4], 48, 40, 131), (6, [2], 30, 64, 131), (6, [2], 48, 40, 131),
                                (6, [2], 48, 40, 64), (8, [4], 48, 40, 64)):
    torch.manual_seed(3)
    m = A.NeRF("fine", D=D, skips=skips, use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True).to(dev)
    g = torch.Generator().manual_seed(4)
    P = n_rays * s
    tiles = (P + 63) // 64
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    t_rows = torch.randn(n_rays, 48, generator=g).to(dev)
    out = {}
    for lo in (False, True):
        raw = torch.zeros(P, _lib.RAW_STRIDE, device=dev)
        acts, xin, masks, side = field_grad.alloc_saves(m, P, dev, True, True, lo)
        acts.zero_(); masks.zero_()
        config.set_tile_points(tp)
        _lib.field_query(m, raw, P, s, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, save_acts=acts, save_xin=xin, save_masks=masks,
                         save_side=side, precision=config.PRECISIONS["f16x3"], save_lo=lo)
        config.set_tile_points(0)
        torch.cuda.synchronize()
        out[lo] = (raw, acts, xin, masks)
    diff = [sl for sl in range(acts.shape[0]) if not torch.equal(out[False][1][sl].view(torch.int16), out[True][1][sl].view(torch.int16))]
    frac = [float((out[False][1][sl] != out[True][1][sl]).float().mean()) for sl in diff]
    print(f"D {D} skips {skips} rays {n_rays} x {s} = {tiles} tiles, tile_points {tp}: kernel {_lib.last_field_kernel()}  slots that differ {diff} {['%.3f' % f for f in frac]}"
          f"  xin equal {torch.equal(out[False][2].view(torch.int16), out[True][2].view(torch.int16))}  masks equal {torch.equal(out[False][3], out[True][3])}"
          f"  raw equal {torch.equal(out[False][0], out[True][0])}")

a0 = out[False][1][0].cpu().numpy().reshape(tiles, 4, 8, 64, 8)
a1 = out[True][1][0].cpu().numpy().reshape(tiles, 4, 8, 64, 8)
d = (a0.view(np.uint16) != a1.view(np.uint16))
print("differing fraction by tile", d.mean((1, 2, 3, 4))[:8], "\n by ks", d.mean((0, 2, 3, 4)), "\n by row block", d.mean((0, 1, 3, 4)), "\n by lane/16", d.reshape(tiles, 4, 8, 4, 16, 8).mean((0, 1, 2, 4, 5)),
      "\n by t", d.mean((0, 1, 2, 3)))
idx = np.argwhere(d)[:6]
for i in idx:
    print(tuple(i), a0[tuple(i)], a1[tuple(i)])
lo = torch.as_strided(out[True][1], out[True][1].shape, out[True][1].stride(), out[True][1].numel())[0].cpu().numpy().reshape(tiles, 4, 8, 64, 8)
print("lo plane sample", lo[0, 0, 0, :4], "\nhi f16", a0[0, 0, 0, :4], "\nhi x3", a1[0, 0, 0, :4])
