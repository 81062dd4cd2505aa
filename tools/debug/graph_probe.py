"""Debug: which part of the training step breaks hipGraph capture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, traceback
import common, scenes
import nsff_pl_amd as A
from nsff_pl_amd.losses import NeRFWLoss
DEV = torch.device("cuda:0")
A.set_precision("f16x3")
name = "g3_nsff_train"
cfg, meta, rays, ts, models, emb, _, want = common.build_case(name, A.NeRF, A.PosEmbedding)
for m in list(models.values()) + [emb["t"]]:
    m.to(DEV)
rays, ts = rays.to(DEV), ts.to(DEV)
kw = scenes.render_kwargs(cfg)
Ks, Ps, max_t = scenes.camera_buffers()
loss_fn = NeRFWLoss(static_shapes=True); loss_fn.register_buffer("Ks", Ks); loss_fn.register_buffer("Ps", Ps); loss_fn.max_t = max_t
loss_fn.to(DEV)
targets = {k: v.to(DEV) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts.cpu(), cfg["seed"]).items()}
params = [p for m in models.values() for p in m.parameters()] + list(emb["t"].parameters())
for p in params: p.grad = torch.zeros_like(p)

def fwd():
    return A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 1.0, 1.0, cfg["N_importance"], 1024 * 32, test_time=False, **kw)
def v_fwd():
    with torch.no_grad(): fwd()
def v_cot():
    scenes.cotangent_loss(fwd()).backward()
def v_static_only():
    res = fwd(); (res["rgb_coarse"].sum()).backward()
def v_loss():
    res = fwd(); sum(loss_fn(res, targets, epoch=5, epoch_ramp=torch.tensor(0.5, device=DEV), **kw).values()).backward()
def v_emb():
    e = emb["t"](ts); (e * e).sum().backward()
def v_median():
    x = torch.rand(64, device=DEV, requires_grad=True); torch.median(x).backward()
def v_cumprod():
    x = torch.rand(16, 64, device=DEV, requires_grad=True); torch.cumprod(x, 1).sum().backward()
def v_field():
    from nsff_pl_amd import field_grad
    m = models["fine"]
    xyz = torch.rand(16 * 64, 3, device=DEV, requires_grad=True); t = torch.randn(16, scenes.N_TAU, device=DEV, requires_grad=True)
    raw = field_grad.field(m, xyz, [float(f) for f in emb["xyz"].freqs], t, 64, False, True)
    raw.sum().backward()
def v_softplus():
    x = torch.randn(16, 64, device=DEV, requires_grad=True); torch.nn.functional.softplus(x).sum().backward()
variants = dict(cumprod=v_cumprod, softplus=v_softplus, field=v_field, static_only=v_static_only, cot=v_cot, loss=v_loss)
mode = sys.argv[1] if len(sys.argv) > 1 else "global"
for nm, fn in variants.items():
    try:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=mode):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(nm, "OK")
    except Exception as e:
        print(nm, "FAILED:", str(e).splitlines()[0][:150])
        torch.cuda.synchronize()
