"""Debug: where do nsff_field_bwd_kernel_h3b and nsff_field_bwd_kernel differ when gradients overflow fp16?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nsff_pl_amd as A
from nsff_pl_amd import _lib, field_grad
DEV = torch.device("cuda:0")
torch.manual_seed(9)
m = A.NeRF("fine", D=4, skips=[2], use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True)
with torch.no_grad():
    for name, p in m.named_parameters():
        if name.endswith(".weight") and "encoding" in name:
            p.mul_(40.0)
m.to(DEV)
g = torch.Generator().manual_seed(10)
P = 128 * 11
tiles = P // 64
d_raw = torch.randn(P, _lib.RAW_STRIDE, generator=g).to(DEV)
raw = (torch.rand(P, _lib.RAW_STRIDE, generator=g) * 0.2).to(DEV)
masks = torch.full((field_grad.n_slots(m), tiles, 256), -1, dtype=torch.int64).to(DEV)
out = {}
for k in ("h", "c"):
    os.environ["NSFF_BWD_KERNEL"] = k
    dpre = torch.full((field_grad.n_slots(m), tiles, 64 * 256), 7.0, device=DEV, dtype=torch.float16)
    dhead = torch.full((2, tiles, 64 * 32), 7.0, device=DEV, dtype=torch.float16)
    d_xin = torch.full((P, 128), 7.0, device=DEV)
    _lib.field_backward(m, P, True, True, d_raw, raw, _lib.absmax(d_raw), masks, dpre, dhead, d_xin)
    torch.cuda.synchronize()
    out[k] = (dpre.cpu().numpy(), dhead.cpu().numpy(), d_xin.cpu().numpy(), _lib.last_bwd_kernel())
a, b = out["h"], out["c"]
print(a[3], b[3])
for s_ in range(a[0].shape[0]):
    da, db = a[0][s_].astype(np.float32), b[0][s_].astype(np.float32)
    ne = (a[0][s_].view(np.uint16) != b[0][s_].view(np.uint16))
    print(f"slot {s_}: mismatches {int(ne.sum())} / {ne.size}   nonfinite h {int((~np.isfinite(da)).sum())} c {int((~np.isfinite(db)).sum())}   max |h| {np.nanmax(np.abs(da)):.1f} max |c| {np.nanmax(np.abs(db)):.1f}")
    if ne.any():
        idx = np.argwhere(ne)[:6]
        for i in idx:
            print("    ", tuple(i), da[tuple(i)], db[tuple(i)])
print("dhead mismatches", int((a[1].view(np.uint16) != b[1].view(np.uint16)).sum()), " d_xin mismatches", int((a[2].view(np.uint32) != b[2].view(np.uint32)).sum()),
      "nonfinite d_xin h", int((~np.isfinite(a[2])).sum()), "c", int((~np.isfinite(b[2])).sum()))
