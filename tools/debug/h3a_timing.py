"""Debug: per-phase cycles of the hand-scheduled field kernel (needs `make -C nsff_pl_amd/csrc timing`).
    NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so python tools/debug/h3a_timing.py
Stamps (s_memtime, shader cycles) of the first 256 workgroups of the C2 fine launch = static-trunk workgroups, per wave:
[0] kernel entry, [1] input tile built, [2] body entered, [3 + i] phase i entered, [62] body left, [63] records stored."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config

which = sys.argv[1] if len(sys.argv) > 1 else "static"       # static | dynamic | dynamic_tb: a launch of that trunk alone (131 072 x 1.5 points)
config.set_precision("f16x3"); config.set_tile_points(130)
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
sm = 2 if which == "static" else 0                            # dynamic: the warp launch shape (dynamic trunk only)
tb = _lib.time_bias([(model, t_rows)])[0] if which == "dynamic_tb" else None     # dynamic_tb: time code folded into bias rows
for _ in range(3):
    _lib.field_query(model, raw, P, S, sm, 0 if which == "static" else 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, t_bias=tb)
torch.cuda.synchronize()
lib = _lib.load()
TBASE = 256 * 8 * 32 * 6
n = TBASE + 256 * 4 * 256
buf = (C.c_uint * n)()
assert lib.nsff_debug_read_timing(buf, n) == 0
allt = np.frombuffer(buf, dtype=np.uint32)
t = allt[TBASE:].reshape(256, 4, 256).astype(np.int64)
rec = t[:, :, 64:64 + 6 * 30].reshape(256, 4, 30, 6)           # record r >= 1: [0] start of phase r - 1, [1..5] its five stamps
dd = lambda a, b: ((b - a) & 0xffffffff).astype(np.float64)
d = lambda a, b: dd(t[:, :, a], t[:, :, b])
names = {1: "A16R", 2: "B16R", 3: "B16X", 4: "A4", 5: "A8", 6: "B4", 7: "B8", 9: "EPI_B", 10: "A4F", 11: "A8F", 12: "B16L", 13: "HEAD"}
if which in ("static", "dynamic_tb"):
    seq = [10, 6] + [1, 2] * 3 + [1, 3, 4, 6] + [1, 2] * 2 + [1, 12, 9, 13]
else:
    seq = [11, 7] + [1, 2] * 3 + [1, 3, 5, 7] + [1, 2] * 2 + [1, 12, 9, 13]
print(f"{which} trunk, mean cycles over 256 workgroups x 4 waves")
print(f"  kernel entry -> input built   {d(0, 1).mean():8.0f}   = bias-row + head requests {d(0, 52).mean():.0f}, point pinned {d(52, 53).mean():.0f}, "
      f"encoder (weight slots 3..7 riding) + bias rows to LDS {d(53, 57).mean():.0f}, time-code pointers + barrier {d(57, 1).mean():.0f}")
print(f"  body prologue                 {dd(rec[:, :, 0, 0], rec[:, :, 1, 0]).mean():8.0f}   (asm start -> first dispatch)")
tot = 0.0
for i, b in enumerate(seq):
    r = rec[:, :, i + 1]
    whole = dd(r[..., 0], rec[:, :, i + 2, 0])
    tot += whole.mean()
    line = f"  phase {i:2d} {names[b]:6s} {whole.mean():7.0f}"
    if b not in (9, 13):
        line += (f" | dispatch {dd(r[..., 0], r[..., 1]).mean():5.0f} | entry waits {dd(r[..., 1], r[..., 2]).mean():5.0f} | MFMAs to the barrier "
                 f"{dd(r[..., 2], r[..., 3]).mean():6.0f} | lgkmcnt(0) + barrier {dd(r[..., 3], r[..., 4]).mean():5.0f} | rest of the body "
                 f"{dd(r[..., 4], r[..., 5]).mean():5.0f} | end -> next dispatch {dd(r[..., 5], rec[:, :, i + 2, 0]).mean():5.0f}")
    print(line)
print(f"  phases total                  {tot:8.0f}")
print(f"  body left    -> records stored {d(62, 63).mean():7.0f}   = barrier {d(62, 54).mean():.0f}, heads {d(54, 55).mean():.0f}, barrier {d(55, 56).mean():.0f}, "
      f"records {d(56, 63).mean():.0f}")
print(f"  whole workgroup               {d(0, 63).mean():8.0f}   ({_lib.last_field_grid()} workgroups for {P // 128} tiles)")
# a persistent workgroup's LAST tile (its stamps overwrite the earlier tiles'): the steady state of the tile loop
print(f"  last tile of the workgroup    {d(58, 59).mean():8.0f}   = requests {d(58, 52).mean():.0f}, point pinned {d(52, 53).mean():.0f}, encoder + bias rows to LDS {d(53, 57).mean():.0f}, "
      f"barrier {d(57, 1).mean():.0f}, body {d(1, 62).mean():.0f}, barrier + next tile's slots 0..2 {d(62, 54).mean():.0f}, records {d(56, 59).mean():.0f}")
