"""Debug: dispatcher-visit stamps of nsff_field_bwd_kernel_h3b (needs `make -C nsff_pl_amd/csrc timing`; regenerate the bodies and
`make all` afterwards).  Run as  NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so python tools/debug/h3b_timing.py [static|dynamic|both]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "h3asm"))
import numpy as np, torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config, field_grad
import gen_bwd

which = sys.argv[1] if len(sys.argv) > 1 else "both"
config.set_precision("f16x3")
dev = torch.device("cuda:0")
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, scenes.CASES["g3_nsff_train"])
model = models["fine"].to(dev)
P = 196608
g = torch.Generator().manual_seed(3)
d_raw = (torch.randn(P, 16, generator=g) * torch.exp(torch.randn(P, 1, generator=g) * 3)).to(dev)
raw = (torch.rand(P, 16, generator=g) * 0.2).to(dev)
tiles = P // 64
masks = torch.randint(-2 ** 62, 2 ** 62, (field_grad.n_slots(model), tiles, 256), generator=g, dtype=torch.int64).to(dev)
gmax = _lib.absmax(d_raw)
dpre = torch.empty(field_grad.n_slots(model), tiles, 64 * 256, device=dev, dtype=torch.float16)
dhead = torch.empty(2, tiles, 64 * 32, device=dev, dtype=torch.float16)
d_xin = torch.empty(P, 128, device=dev)
static, transient = which in ("static", "both"), which in ("dynamic", "both")
for _ in range(3):
    _lib.field_backward(model, P, static, transient, d_raw, raw, gmax, masks, dpre, dhead, d_xin, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.field_backward(model, P, static, transient, d_raw, raw, gmax, masks, dpre, dhead, d_xin, None)
e1.record()
torch.cuda.synchronize()
assert _lib.last_bwd_kernel() == "h3b"
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us ({which}; stamps of the LAST workgroup that used each of the 256 x 4 records)")
lib = _lib.load()
n = 256 * 4 * 32 * 6
buf = (C.c_uint * n)()
assert lib.nsff_debug_read_bwd_timing(buf, n) == 0
later = len(sys.argv) > 2 and sys.argv[2] == "later"      # the workgroups' LAST later item instead of their first one (persistent launches)
t = np.frombuffer(buf, dtype=np.uint32)[65536 if later else 0:][:256 * 4 * 64].reshape(256, 4, 64).astype(np.int64)
print("items:", "a workgroup's last later item (entry = the top of its loop iteration)" if later else "a workgroup's first item", " grid", _lib.last_bwd_grid())
names = {v: k for k, v in gen_bwd.BODY.items()}
ph = {}
for dyn in (False, True):
    prog, _ = _lib.field_bwd_phase_program(model, dyn, dyn, tiles)
    ph[dyn] = [names[int(b)] for b in prog[1:, 0]]
d = lambda x, y: ((y - x) & 0xffffffff).astype(np.float64)
# workgroup b: trunk (b & 7) >> 2 when both trunks run
for dyn in ([False, True] if which == "both" else [which == "dynamic"]):
    sel = [b for b in range(256) if which != "both" or (b & 1) == int(dyn)]        # (items alternate between the trunks; 256 workgroups: a workgroup keeps its parity)
    tt = t[sel]
    seq = ph[dyn]
    n_ph = seq.index("END")
    print(f"-- {'dynamic' if dyn else 'static'} trunk: {n_ph} phases; mean cycles (s_memtime ticks at 100 MHz x clock ratio are NOT cycles: see below)")
    print(f"   entry -> head stage done {d(tt[:, :, 0], tt[:, :, 1]).mean():8.0f}")
    hs = ["entry -> top of the item's loop iteration (first item: the record DMA and its wait)", "barrier (every wave done with the previous item)",
          "counter fetch issued, records read from LDS, the pre-issue statement (8 head-segment loads)",
          "head arithmetic, half A", "head arithmetic, half B", "barrier", "head-gradient stores issued"]
    prev = tt[:, :, 0]
    for k, nm in enumerate(hs):
        print(f"      {nm:90s} {d(prev, tt[:, :, 56 + k]).mean():8.0f}")
        prev = tt[:, :, 56 + k]
    print(f"   head stage done -> first dispatch {d(tt[:, :, 1], tt[:, :, 2]).mean():8.0f}   (prologue: addresses, 32 weight-slot loads)")
    for i in range(n_ph):
        print(f"   {seq[i]:10s} {d(tt[:, :, 2 + i], tt[:, :, 3 + i]).mean():8.0f}")
    print(f"   whole workgroup {d(tt[:, :, 0], tt[:, :, 2 + n_ph]).mean():8.0f}")
