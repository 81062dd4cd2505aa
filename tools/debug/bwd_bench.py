"""Debug: time the training kernels of one field node in isolation (HIP events, nothing else on the GPU):
training forward, nsff_field_backward, nsff_weight_grad, for the fine-pass shapes of the C2 batch.
    python tools/debug/bwd_bench.py [n_points] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config, field_grad

P = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S = 128
config.set_precision("f16x3")
dev = torch.device("cuda:0")
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, scenes.CASES["g3_nsff_train"])
model = models["fine"].to(dev)
freqs = [float(f) for f in emb["xyz"].freqs]
g = torch.Generator().manual_seed(3)
xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
t_rows = torch.randn(P // S, scenes.N_TAU, generator=g).to(dev)
d_raw = (torch.randn(P, 16, generator=g) * torch.exp(torch.randn(P, 1, generator=g) * 3)).to(dev)


def timed(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for static in (True, False):
    raw = torch.empty(P, 16, device=dev)
    acts, xin, masks, side = field_grad.alloc_saves(model, P, dev, True, static)

    def fwd(save=True):
        kw = dict(save_acts=acts, save_xin=xin, save_masks=masks) if save else {}
        _lib.field_query(model, raw, P, S, 2 if static else 0, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows,
                         precision=config.PRECISIONS["f16x3"], **kw)
    tiles = (P + 63) // 64
    gmax = _lib.absmax(d_raw)
    dpre = torch.empty(field_grad.n_slots(model), tiles, 64 * 256, device=dev, dtype=torch.float16)
    dhead = torch.empty(2, tiles, 64 * 32, device=dev, dtype=torch.float16)
    d_xin = torch.empty(P, 128, device=dev)

    def bwd():
        _lib.field_backward(model, P, static, True, d_raw, raw, gmax, masks, dpre, dhead, d_xin, None)
    t_inf, t_fwd = timed(lambda: fwd(False)), timed(fwd)
    t_bwd = timed(bwd)
    steps = (10 if not static else 20)
    flop = P * steps * 2 * 256 * 256
    wbytes = dpre[:(2 if static else 1) * (model.D + 1)].numel() * 2 if static else dpre[model.D + 1:].numel() * 2
    print(f"static={static}: inference fwd {t_inf:7.1f} us | training fwd {t_fwd:7.1f} us | field_backward {t_bwd:7.1f} us "
          f"= {flop / t_bwd / 1e6:6.1f} TFLOP/s, dpre {wbytes / 1e6:.0f} MB -> {wbytes / t_bwd / 1e6:.2f} TB/s written", flush=True)
print("absmax", f"{timed(lambda: _lib.absmax(d_raw)):.1f} us   torch abs().max() {timed(lambda: d_raw.abs().max()):.1f} us")
