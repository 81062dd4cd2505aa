"""Debug: per-phase cycle breakdown of the f16x3 field kernel (needs `make -C nsff_pl_amd/csrc timing`).
Run as  NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so python tools/debug/h3_timing.py [tile_points]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
save = len(sys.argv) > 3 and sys.argv[3] == "save"          # the training forward (keeps activations) instead of inference
config.set_precision(prec); config.set_tile_points(tile)
dev = torch.device("cuda:0")
cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=1024, seed=0)
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
for m in list(models.values()) + [emb["t"]]:
    m.to(dev)
model = models["fine"]
P, S = 1024 * 192, 192
xyz = (torch.rand(P, 3, device=dev) * 2 - 1)
t_rows = torch.randn(1024, scenes.N_TAU, device=dev)
raw = torch.zeros(P, 16, device=dev)
freqs = [float(f) for f in emb["xyz"].freqs]
kw = {}
if save:
    from nsff_pl_amd import field_grad
    acts, xin, masks, _ = field_grad.alloc_saves(model, P, dev, True, True)
    kw = dict(save_acts=acts, save_xin=xin, save_masks=masks)
for _ in range(3):
    _lib.field_query(model, raw, P, S, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, **kw)
torch.cuda.synchronize()
lib = _lib.load()
# tick calibration: one launch bracketed by events vs first/last s_memtime stamp inside it
span = (C.c_ulonglong * 2)()
lib.nsff_debug_span.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.nsff_debug_span(span, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.field_query(model, raw, P, S, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, **kw)
e1.record()
torch.cuda.synchronize()
lib.nsff_debug_span(span, 0)
ms = e0.elapsed_time(e1)
print(f"launch {ms:.4f} ms by events, {span[1] - span[0]} ticks first->last stamp  =>  {(span[1] - span[0]) / ms / 1e6:.4f} GHz tick rate (lower bound)")
n = 256 * 8 * 32 * 6
buf = (C.c_uint * n)()
assert lib.nsff_debug_read_timing(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint32).reshape(256, 8, 32, 6).astype(np.int64)
waves = 4 if tile == 64 else 8
t = t[:, :waves]
# the first 256 workgroups of a two-trunk launch each run the STATIC trunk of their tile (grid = 2 x tiles, static first)
nsteps = int((t[0, 0, :31, 1] != 0).sum())
post = [bool(t[0, 0, s_, 3] != 0) for s_ in range(nsteps)]       # steps with an epilogue (the skip layer's first half has none)
print("steps per workgroup:", nsteps, " with epilogue:", post)
d = lambda a, b: ((t[:, :, :nsteps, b] - t[:, :, :nsteps, a]) & 0xffffffff).astype(np.float64)
tot = ((t[:, :, 31, 0] - t[:, :, 0, 0]) & 0xffffffff).astype(np.float64)
print(f"whole workgroup (first stamp -> end of kernel) mean cycles {tot.mean():.0f}")
rows = [("pre (barrier+build)", d(0, 1)), ("gemm issue", d(1, 2))]
mask = np.array(post)
dr, st_, b2 = d(2, 3), d(3, 4), d(4, 5)
for x in (dr, st_, b2):
    x[:, :, ~mask] = 0
rows += [("barrier 1 (gemm drain)", dr), ("acc_store", st_), ("barrier 2", b2)]
acc = 0.0
for nm, x in rows:
    acc += x.sum()
    print(f"{nm:26s} share {x.sum() / tot.sum() * 100:5.1f}%   per-step means: " + " ".join(f"{v:6.0f}" for v in x.mean((0, 1))))
last = np.where(mask, 5, 2)                                  # last stamp a step leaves
gaps = []
for s_ in range(nsteps):
    nxt_t = t[:, :, s_ + 1, 0] if s_ + 1 < nsteps else t[:, :, 31, 0]
    gaps.append((((nxt_t - t[:, :, s_, last[s_]]) & 0xffffffff).astype(np.float64)).mean())
print("after each step (heads / next pre-barrier / end of kernel): " + " ".join(f"{v:6.0f}" for v in gaps))
print(f"{'heads, raw records, rest':26s} share {(tot.sum() - acc) / tot.sum() * 100:5.1f}%")

# ---- inside / behind the last head call of each workgroup (slot 30: 0 entry, 1 MFMAs done, 2 k-split exchange done,
# 3 activations written to the record image, 4 before the final barrier, 5 after it; slot 31 stamp 0 = records stored) ----
h = t[:, :, 30, :]
last5 = ((t[:, :, nsteps - 1, 5] if post[nsteps - 1] else t[:, :, nsteps - 1, 2]))
seg = lambda a_, b_: ((b_ - a_) & 0xffffffff).astype(np.float64)
kh0 = slice(0, waves // 2)                                   # waves of the lower k half run the whole head
print("last head call, waves of the lower k half (mean cycles):",
      f"barrier-2 of the last layer -> head entry {seg(last5, h[..., 0])[:, kh0].mean():.0f} |",
      f"weights + MFMAs {seg(h[..., 0], h[..., 1])[:, kh0].mean():.0f} |",
      f"k-split exchange (barrier) {seg(h[..., 1], h[..., 2])[:, kh0].mean():.0f} |",
      f"bias + activations + record image {seg(h[..., 2], h[..., 3])[:, kh0].mean():.0f} |",
      f"-> final barrier entered {seg(h[..., 3], h[..., 4])[:, kh0].mean():.0f} | barrier {seg(h[..., 4], h[..., 5]).mean():.0f} |",
      f"records to HBM {seg(h[..., 5], t[:, :, 31, 0]).mean():.0f}")

# ---- per-wave view of one mid-trunk layer (step 2) of a few workgroups: when does each wave start / finish issuing its GEMM,
# how long does it wait at the barrier behind it? ----
for wg in (0, 97, 200):
    base = t[wg, :, 2, 1].min()
    rel = lambda k: ((t[wg, :, 2, k] - base) & 0xffffffff)
    print(f"workgroup {wg:3d} layer step 2: GEMM start {rel(1).tolist()}  GEMM issued {rel(2).tolist()}  past barrier 1 {rel(3).tolist()}")
