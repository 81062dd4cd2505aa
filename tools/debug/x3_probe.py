"""Three-product backward, piece by piece, against float64 restatements of the same sums (debugging aid).
    python tools/debug/x3_probe.py [spread]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes                                   # noqa: E402
import nsff_pl_amd as A                         # noqa: E402
from nsff_pl_amd import _lib, config, field_grad         # noqa: E402

spread = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = torch.device("cuda:0")
cfg = scenes.CASES["g12_other_arch"]
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
m = models["fine"].to(dev)
D = m.D
g = torch.Generator().manual_seed(5)
n_rays, s = 48, 40
P = n_rays * s
tiles = (P + 63) // 64
xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
t_rows = torch.randn(n_rays, m.in_channels_t, generator=g).to(dev)
cot = torch.randn(P, 16, generator=g)
cot *= 10.0 ** (-spread * torch.rand(P, 1, generator=g))
cot = cot.to(dev)
freqs = [float(f) for f in emb["xyz"].freqs]


def unfrag(buf, rows):
    a = buf.float().cpu().numpy().reshape(-1, 4, rows // 32, 2, 32, 8)
    return a.transpose(0, 1, 3, 5, 2, 4).reshape(-1, rows).astype(np.float64)


lo_of = lambda t: torch.as_strided(t, t.shape, t.stride(), t.storage_offset() + t.numel())
out = {}
for x3 in (False, True):
    raw = torch.empty(P, _lib.RAW_STRIDE, device=dev)
    acts, xin, masks, side = field_grad.alloc_saves(m, P, dev, True, True, x3)
    config.set_tile_points(131)
    _lib.field_query(m, raw, P, s, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, save_acts=acts, save_xin=xin, save_masks=masks,
                     save_side=side, precision=config.PRECISIONS["f16x3"], save_lo=x3)
    config.set_tile_points(0)
    torch.cuda.synchronize()
    snap = acts.clone()
    if x3:
        print("after the x3 forward: hi plane equals the f16 run's:", torch.equal(acts.view(torch.int16), out[False]["acts"].view(torch.int16)),
              " last forward kernel", _lib.last_field_kernel())
    gmax = _lib.absmax(cot)
    dpre, _ = field_grad._planes(x3, field_grad.n_slots(m), tiles, 64 * 256, device=dev)
    dpre.zero_()
    if x3:
        lo_of(dpre).zero_()
    dhead = torch.zeros(2, tiles, 64 * 32, device=dev, dtype=torch.float16)
    xin_rows, t_row0, side_rows = _lib.train_dims(m)
    d_xin = torch.zeros(P, xin_rows, device=dev)
    _lib.field_backward(m, P, True, True, cot, raw, gmax, masks, dpre, dhead, d_xin, None, x3=x3)
    torch.cuda.synchronize()
    print("x3" if x3 else "f16", "kernel", _lib.last_bwd_kernel(), " acts unchanged by the backward:", torch.equal(snap.view(torch.int16), acts.view(torch.int16)))
    out[x3] = dict(acts=acts, xin=xin, dpre=dpre, dhead=dhead, d_xin=d_xin, gmax=gmax, masks=masks)

a, b = out[False], out[True]
print("d_xin: x3 vs f16 rel", float((a["d_xin"] - b["d_xin"]).abs().max() / a["d_xin"].abs().max()))
for slot in range(field_grad.n_slots(m)):
    h0 = unfrag(a["dpre"][slot], 256)
    h1 = unfrag(b["dpre"][slot], 256)
    l1 = unfrag(lo_of(b["dpre"])[slot], 256)
    sc = np.abs(h0).max()
    if sc == 0:
        continue
    print(f"slot {slot:2d}  max |dpre| {sc:9.2f}   x3.hi vs f16 {np.abs(h1 - h0).max() / sc:.2e}   x3.(hi+lo) vs f16 {np.abs(h1 + l1 - h0).max() / sc:.2e}"
          f"   max|lo|/max|hi| {np.abs(l1).max() / sc:.2e}")

# weight-gradient GEMM of one hidden layer against float64 of the decoded operands
t, l = 0, 3
base = t * (D + 1)
A1 = unfrag(b["dpre"][base + l], 256) + unfrag(lo_of(b["dpre"])[base + l], 256)
B1 = unfrag(b["acts"][base + l - 1], 256) + unfrag(lo_of(b["acts"])[base + l - 1], 256)
want = A1.T @ B1
jobs = [(b["dpre"][base + l].data_ptr(), b["acts"][base + l - 1].data_ptr(), 256, 256, 0, t, _lib.lo_delta(b["dpre"]), _lib.lo_delta(b["acts"]))]
outw = torch.zeros(256 * 256, device=dev)
bias = torch.zeros(1, 256, device=dev)
_lib.weight_grad(jobs, tiles, 4, outw, bias, b["gmax"])
torch.cuda.synchronize()
G = 2.0 ** np.floor(np.log2(2048.0 / float(b["gmax"][:4].max()))) if False else None
got = outw.view(256, 256).double().cpu().numpy()
ratio = (got * want).sum() / (want * want).sum()
print("wgrad x3: scale ratio got/want", ratio, " rel err after scale", np.abs(got / ratio - want).max() / np.abs(want).max())
A0 = unfrag(a["dpre"][base + l], 256)
B0 = unfrag(a["acts"][base + l - 1], 256)
jobs0 = [(a["dpre"][base + l].data_ptr(), a["acts"][base + l - 1].data_ptr(), 256, 256, 0, t)]
_lib.weight_grad(jobs0, tiles, 4, outw, bias, a["gmax"])
torch.cuda.synchronize()
got0 = outw.view(256, 256).double().cpu().numpy()
want0 = A0.T @ B0
r0 = (got0 * want0).sum() / (want0 * want0).sum()
print("wgrad f16: scale ratio", r0, " rel err vs its own operands", np.abs(got0 / r0 - want0).max() / np.abs(want0).max(),
      " vs the x3 operands", np.abs(got0 / r0 - want).max() / np.abs(want).max())

# ---- float64 truth of the pre-activation gradients (the torch expression's F.linear outputs, recorded in call order)
import copy
import torch_path
m64 = copy.deepcopy(m).double()
store = []
orig_lin = torch_path._lin
def rec_lin(mod, x):
    y = orig_lin(mod, x)
    y.retain_grad()
    store.append(y)
    return y
torch_path._lin = rec_lin
x64 = xyz.double()
outs = torch_path.field(m64, torch_path.pos_embed(x64, freqs), None, None, t_rows.double().repeat_interleave(s, 0), True, True, ("fw", "bw"))
torch_path._lin = orig_lin
cols = {"rgb_s": slice(0, 3), "sigma_s": 3, "rgb_t": slice(4, 7), "sigma_t": 7, "fw": slice(8, 11), "bw": slice(11, 14)}
sum((outs[k] * cot[:, c].double()).sum() for k, c in cols.items() if k in outs).backward()
gm = b["gmax"].double().cpu().numpy()
def pow2(amax):
    return 2.0 ** (11 - np.frexp(amax)[1])
Gt = [pow2(gm[0:4].max()), pow2(gm[4:14].max())]
print("G", Gt)
# call order: static trunk D layers, sigma, final, rgb; transient trunk D layers, final, ...
pre = {(0, i): store[i] for i in range(D)}
pre.update({(1, i): store[D + 3 + i] for i in range(D)})
for (t_, i), v in sorted(pre.items()):
    truth = v.grad.cpu().numpy()[:P] * Gt[t_]
    slot = t_ * (D + 1) + i
    h0 = unfrag(a["dpre"][slot], 256)[:P]
    h1 = unfrag(b["dpre"][slot], 256)[:P]
    l1 = unfrag(lo_of(b["dpre"])[slot], 256)[:P]
    rowmax = np.abs(truth).max(1, keepdims=True) + 1e-300
    big = (rowmax[:, 0] > 1e-3 * rowmax.max())
    e0 = (np.abs(h0 - truth) / rowmax)[big].max()
    e1 = (np.abs(h1 - truth) / rowmax)[big].max()
    e3 = (np.abs(h1 + l1 - truth) / rowmax)[big].max()
    print(f"trunk {t_} layer {i}: per-point relative error  f16 {e0:.2e}   x3.hi {e1:.2e}   x3.hi+lo {e3:.2e}   points judged {int(big.sum())}")
    if i >= 1:
        act = torch.relu(pre[(t_, i - 1)].detach()).cpu().numpy()[:P]
        s_ = t_ * (D + 1) + i - 1
        b0 = unfrag(a["acts"][s_], 256)[:P]; b1 = unfrag(b["acts"][s_], 256)[:P]; bl = unfrag(lo_of(b["acts"])[s_], 256)[:P]
        sc = np.abs(act).max()
        print(f"      activation slot {s_}: f16 {np.abs(b0 - act).max() / sc:.2e}  x3.hi {np.abs(b1 - act).max() / sc:.2e}  x3.hi+lo {np.abs(b1 + bl - act).max() / sc:.2e}")
wtruth = getattr(m64, "static_xyz_encoding_4")[0].weight.grad.cpu().numpy()
print("dW static layer 3: x3 GEMM vs float64 autograd", np.abs(got / ratio / Gt[0] - wtruth[:, :256]).max() / np.abs(wtruth).max(),
      "  f16 GEMM", np.abs(got0 / r0 / Gt[0] - wtruth[:, :256]).max() / np.abs(wtruth).max())
