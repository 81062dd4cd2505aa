"""Debug: per-phase cycle breakdown of nsff_field_bwd_kernel (needs `make -C nsff_pl_amd/csrc timing`).
Run as  NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so python tools/debug/bwd_timing.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config, field_grad

config.set_precision("f16x3")
dev = torch.device("cuda:0")
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, scenes.CASES["g3_nsff_train"])
model = models["fine"].to(dev)
freqs = [float(f) for f in emb["xyz"].freqs]
P, S = 131072, 128
g = torch.Generator().manual_seed(3)
xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
t_rows = torch.randn(P // S, scenes.N_TAU, generator=g).to(dev)
d_raw = (torch.randn(P, 16, generator=g) * torch.exp(torch.randn(P, 1, generator=g) * 3)).to(dev)
raw = torch.empty(P, 16, device=dev)
acts, xin, masks, side = field_grad.alloc_saves(model, P, dev, True, True)
_lib.field_query(model, raw, P, S, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, precision=config.PRECISIONS["f16x3"],
                 save_acts=acts, save_xin=xin, save_masks=masks)
tiles = (P + 63) // 64
gmax = _lib.absmax(d_raw)
dpre = torch.empty(field_grad.n_slots(model), tiles, 64 * 256, device=dev, dtype=torch.float16)
dhead = torch.empty(2, tiles, 64 * 32, device=dev, dtype=torch.float16)
d_xin = torch.empty(P, 128, device=dev)
for _ in range(3):
    _lib.field_backward(model, P, True, True, d_raw, raw, gmax, masks, dpre, dhead, d_xin, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.field_backward(model, P, True, True, d_raw, raw, gmax, masks, dpre, dhead, d_xin, None)
e1.record()
torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, {tiles} tiles")
lib = _lib.load()
n = 256 * 4 * 32 * 6
buf = (C.c_uint * n)()
assert lib.nsff_debug_read_bwd_timing(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint32).reshape(256, 4, 32, 6).astype(np.int64)
nsteps = 22                               # static 10 + dynamic 12 steps (both trunks, d_xin)
d = lambda x, y: ((y - x) & 0xffffffff).astype(np.float64)
tot = d(t[:, :, 0, 0], t[:, :, 31, 0])
print(f"whole workgroup (first step stamp -> end) mean cycles {tot.mean():.0f}; per step {tot.mean() / nsteps:.0f}")
names = ["gemm issue (0->1)", "prefetch+flush (1->2)", "barrier 1 (2->3)", "epilogue (3->4)", "barrier 2 (4->5)"]
epi = t[0, 0, :nsteps, 3] != 0
acc = 0.0
for k, nm in enumerate(names):
    x = d(t[:, :, :nsteps, k], t[:, :, :nsteps, k + 1])
    if k >= 2:
        x[:, :, ~epi] = 0
    acc += x.sum()
    print(f"{nm:26s} share {x.sum() / tot.sum() * 100:5.1f}%   per-step means: " + " ".join(f"{v:6.0f}" for v in x.mean((0, 1))))
last = np.where(epi, 5, 2)
gaps = []
for s_ in range(nsteps):
    nxt = t[:, :, s_ + 1, 0] if s_ + 1 < nsteps else t[:, :, 31, 0]
    gaps.append(d(t[:, :, s_, last[s_]], nxt).mean())
print("gap to the next step (head stage / mask load / end flush): " + " ".join(f"{v:6.0f}" for v in gaps))
print(f"{'rest (gaps)':26s} share {(tot.sum() - acc) / tot.sum() * 100:5.1f}%")
