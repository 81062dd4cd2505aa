"""GraphedRender (nsff_pl_amd/graphs.py): render_rays replayed as one hipGraph gives the eager call's values, follows weight
updates, and serves the chunk loop of render_frame."""
import numpy as np
import pytest
import torch

import common
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import config, evaluate
from nsff_pl_amd.graphs import GraphedRender

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(name):
    cfg, meta, rays, ts, models, emb, dataset, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    for m in list(models.values()) + [emb[k] for k in ("t", "a") if k in emb]:
        m.to(DEV)
    return cfg, rays.to(DEV), ts.to(DEV), models, emb


def test_graphed_render_equals_the_eager_call(hip_lib):
    cfg, rays, ts, models, emb = _scene("g4_nsff_test")
    kw = scenes.render_kwargs(cfg)
    eager = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"], test_time=True, **kw)
    g = GraphedRender(models, emb, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"], test_time=True, **kw)
    for rep in range(3):                                    # capture, then two replays -- the last one on other rays
        r = rays if rep < 2 else rays.flip(0).contiguous()
        t = ts if rep < 2 else ts.flip(0).contiguous()
        out = g(r, t)
        want = eager if rep < 2 else {k: v.flip(0) for k, v in eager.items()}
        assert sorted(out) == sorted(want)
        for k in want:
            assert torch.equal(out[k], want[k]), (rep, k)
    # a weight update between two calls is seen by the next replay (the pack buffers keep their addresses)
    with torch.no_grad():
        models["fine"].static_rgb[0].bias.add_(0.25)
    after = g(rays, ts)
    ref = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"], test_time=True, **kw)
    assert torch.equal(after["rgb_fine"], ref["rgb_fine"]) and not torch.equal(ref["rgb_fine"], eager["rgb_fine"])


def test_graph_is_recaptured_when_parameters_move(hip_lib):
    """The captured launches hold raw parameter addresses (nsff_time_bias reads the input-layer weights, the embedding gathers
    their tables): FlatAdam.adopt() re-points every parameter at a slice of its flat buffer -- the next call must re-capture, not
    replay against freed memory, and must follow the optimizer step."""
    from nsff_pl_amd.optim import FlatAdam
    cfg, rays, ts, models, emb = _scene("g3_nsff_train")
    # the C2-like launch mix: force the hand-scheduled body with per-ray time-bias rows also on this small batch
    config.set_tile_points(130)
    try:
        kw = scenes.render_kwargs(cfg)
        g = GraphedRender(models, emb, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"], test_time=False, **kw)
        before = {k: v.clone() for k, v in g(rays, ts).items()}
        old_ptr = models["fine"].transient_xyz_encoding_1[0].weight.data_ptr()
        params = [p for m in list(models.values()) + [emb["t"]] for p in m.parameters()]
        opt = FlatAdam(params, lr=1e-2)                       # adopt(): every parameter now lives in the flat buffer
        assert models["fine"].transient_xyz_encoding_1[0].weight.data_ptr() != old_ptr
        again = g(rays, ts)                                   # same values, new addresses: a fresh capture
        assert len(g._graphs) == 1
        for k in before:
            assert torch.equal(again[k], before[k]), k
        opt.flat_grad.normal_(0, 1e-2)
        opt.step()
        got = {k: v.clone() for k, v in g(rays, ts).items()}
        with torch.no_grad():
            want = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"],
                                 test_time=False, **kw)
        assert not torch.equal(got["rgb_fine"], before["rgb_fine"])
        for k in want:
            assert torch.equal(got[k], want[k]), k
    finally:
        config.set_tile_points(0)


def test_graphed_train_mode_call_with_draws(hip_lib):
    """train-mode flags with perturb / noise: the generator kernels are part of the graph; same seed -> the eager call's values"""
    cfg = dict(scenes.CASES["g7_nsff_train_noise"], n_rays=32)
    rays, ts = scenes.synthetic_rays(32, 3)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    for m in list(models.values()) + [emb["t"]]:
        m.to(DEV)
    rays, ts, kw = rays.to(DEV), ts.to(DEV), scenes.render_kwargs(cfg)
    g = GraphedRender(models, emb, scenes.N_FRAMES - 1, 64, 1.0, 1.0, 64, test_time=False, **kw)
    g(rays, ts)                                             # capture (consumes generator state during warm-up)
    torch.manual_seed(77)
    got = {k: v.clone() for k, v in g(rays, ts).items()}
    torch.manual_seed(77)
    with torch.no_grad():
        want = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, 64, 1.0, 1.0, 64, test_time=False, **kw)
    for k in ("rgb_fine", "depth_fine", "zs_fine", "transient_flow_fw", "rgb_fw"):
        assert torch.equal(got[k], want[k]), k


def test_render_frame_through_graphs(hip_lib):
    cfg, _, _, models, emb = _scene("g4_nsff_test")
    H, W = 40, 64
    K = np.array([[50., 0, W / 2], [0, 50., H / 2], [0, 0, 1]], np.float32)
    c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    rays = evaluate.frame_rays(K, c2w, H, W, device=DEV)
    ts = torch.full((H * W,), 7, dtype=torch.long, device=DEV)
    kw = scenes.render_kwargs(cfg)
    keys = ("rgb_fine", "depth_fine")
    plain = evaluate.render_frame(models, emb, rays, ts, 29, 64, 64, chunk=1024, keys=keys, **kw)
    g = GraphedRender(models, emb, 29, 64, 0, 0, 64, test_time=True, **kw)
    viag = evaluate.render_frame(models, emb, rays, ts, 29, 64, 64, chunk=1024, keys=keys, graph=g, **kw)   # 2 full chunks + 512
    assert len(g._graphs) == 2
    for k in keys:
        assert torch.equal(plain[k], viag[k]), k
