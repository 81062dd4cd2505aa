"""Debug: training forward (save_*) vs inference forward of the same points; saved trunk input vs the embedding."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config, field_grad

config.set_precision("f16x3")
dev = torch.device("cuda:0")
models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, scenes.CASES["g3_nsff_train"])
model = models["fine"].to(dev)
freqs = [float(f) for f in emb["xyz"].freqs]
g = torch.Generator().manual_seed(5)
n_rays, S = 48, 40
P = n_rays * S
xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
t_rows = torch.randn(n_rays, scenes.N_TAU, generator=g).to(dev)
for static in (True, False):
    raw0 = torch.empty(P, 16, device=dev); raw1 = torch.empty(P, 16, device=dev)
    acts, xin, masks, side = field_grad.alloc_saves(model, P, dev, True, static)
    kw = dict(xyz=xyz, freqs=freqs, t_emb=t_rows, precision=config.PRECISIONS["f16x3"])
    _lib.field_query(model, raw0, P, S, 2 if static else 0, 2, 2, **kw)
    _lib.field_query(model, raw1, P, S, 2 if static else 0, 2, 2, save_acts=acts, save_xin=xin, save_masks=masks, **kw)
    torch.cuda.synchronize()
    print("static", static, "max |raw_train - raw_inference|", float((raw0 - raw1).abs().max()))
    x = field_grad._unfragment(xin[None], 128)[0][:P].float()
    want = torch.cat([emb["xyz"].to(dev)(xyz) if hasattr(emb["xyz"], "to") else emb["xyz"](xyz), torch.zeros(P, 1, device=dev),
                      t_rows.repeat_interleave(S, 0)], 1)
    print("   saved xin vs embedding: xyz part", float((x[:, :64] - want[:, :64]).abs().max()), " t part",
          float((x[:, 64:64 + scenes.N_TAU] - want[:, 64:]).abs().max()))
    bad = (x[:, :64] - want[:, :64]).abs().amax(0)
    print("   worst columns:", [(int(i), float(bad[i])) for i in torch.argsort(bad, descending=True)[:6]])
