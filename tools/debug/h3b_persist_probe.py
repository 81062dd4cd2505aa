"""Persistent data-gradient launch, step by step (debugging aid)."""
import os, sys, faulthandler
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nsff_pl_amd as A
from nsff_pl_amd import _lib, field_grad
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = A.NeRF("fine", use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True).to(dev)
g = torch.Generator().manual_seed(2)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
static = (sys.argv[2] == "1") if len(sys.argv) > 2 else True
tiles = (P + 63) // 64
d_raw = torch.randn(P, _lib.RAW_STRIDE, generator=g).to(dev)
raw = (torch.rand(P, _lib.RAW_STRIDE, generator=g) * 0.2).to(dev)
masks = torch.randint(-2 ** 62, 2 ** 62, (field_grad.n_slots(m), tiles, 256), generator=g, dtype=torch.int64).to(dev)
xin_rows, _, _ = _lib.train_dims(m)
res = {}
for mode in ("0", None, None):
    if mode is None:
        os.environ.pop("NSFF_BWD_PERSIST", None)
    else:
        os.environ["NSFF_BWD_PERSIST"] = mode
    dpre = torch.full((field_grad.n_slots(m), tiles, 64 * 256), 7.0, device=dev, dtype=torch.float16)
    dhead = torch.full((2, tiles, 64 * 32), 7.0, device=dev, dtype=torch.float16)
    d_xin = torch.full((P, xin_rows), 7.0, device=dev)
    gmax = _lib.absmax(d_raw)
    torch.cuda.synchronize()
    print("launch", mode, flush=True)
    _lib.field_backward(m, P, static, True, d_raw, raw, gmax, masks, dpre, dhead, d_xin)
    print("launched grid", _lib.last_bwd_grid(), _lib.last_bwd_kernel(), flush=True)
    torch.cuda.synchronize()
    print("done", flush=True)
    cur = (dpre.cpu().numpy().view(np.uint16), dhead.cpu().numpy().view(np.uint16), d_xin.cpu().numpy().view(np.uint32))
    if res:
        print("equal to one-workgroup-per-item:", [bool(np.array_equal(a, b)) for a, b in zip(cur, res["0"])], flush=True)
    else:
        res["0"] = cur
    if res and not all(np.array_equal(a, b) for a, b in zip(cur, res["0"])):
        a, b = cur, res["0"]
        d0 = (a[0] != b[0]).reshape(a[0].shape[0], tiles, -1).mean(2)
        for sl in range(d0.shape[0]):
            nz = np.nonzero(d0[sl])[0]
            if len(nz):
                print(f"  dpre slot {sl}: tiles that differ {nz[:8].tolist()} ... {len(nz)} tiles, fraction in the first {d0[sl][nz[0]]:.3f}")
        d1 = (a[1] != b[1]).reshape(2, tiles, -1).mean(2)
        for t in range(2):
            nz = np.nonzero(d1[t])[0]
            if len(nz):
                print(f"  dhead trunk {t}: tiles {nz[:8].tolist()} ... {len(nz)}, fraction {d1[t][nz[0]]:.3f}")
                x = a[1].reshape(2, tiles, -1)[t, nz[0]].view(np.float16)[:16]; y = b[1].reshape(2, tiles, -1)[t, nz[0]].view(np.float16)[:16]
                print("    got ", x, "\n    want", y)
        d2 = (a[2] != b[2]).reshape(-1, 64, a[2].shape[1]).mean((1, 2))
        nz = np.nonzero(d2)[0]
        print(f"  d_xin: 64-point tiles that differ {nz[:8].tolist()} ... {len(nz)}")
        break
