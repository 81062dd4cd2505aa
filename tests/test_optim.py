"""The native Adam step (nsff_pl_amd/optim.py, csrc/optim.hip) against torch.optim.Adam as the reference configures it
(utils/__init__.py:45-47: lr, eps=1e-8, weight_decay; train.py:143-146: MultiStepLR)."""
import numpy as np
import pytest
import torch

import common
from nsff_pl_amd.optim import FlatAdam

SHAPES = [(256, 63), (256,), (3, 256), (1,), (5, 7, 3), (48, 30)]          # total not a multiple of 4


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(dev)) for s in SHAPES]


def _run(opt_factory, dev, wd, steps=6, lr_drop_at=3, unused=()):
    """`unused`: indices of parameters that never receive a gradient (torch: .grad stays None; flat: the slice stays zero)."""
    params = _params(dev)
    opt = opt_factory(params, wd)
    g = torch.Generator().manual_seed(7)
    for i in range(steps):
        opt.zero_grad()
        for k, p in enumerate(params):
            grad = (torch.randn(*p.shape, generator=g) * 10.0 ** float(torch.randint(-4, 2, (1,), generator=g))).to(dev)
            if k in unused:
                continue
            if p.grad is None:
                p.grad = grad
            else:
                p.grad.copy_(grad)
        opt.step()
        if i + 1 == lr_drop_at:
            if isinstance(opt, FlatAdam):
                opt.lr.mul_(0.1)
            else:
                opt.param_groups[0]["lr"] *= 0.1
    return [p.detach().cpu().numpy().copy() for p in params]


def _torch_adam(params, wd):
    return torch.optim.Adam(params, lr=5e-4, eps=1e-8, weight_decay=wd)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_torch_op_twin_equals_torch_adam_on_cpu(wd):
    Twin = common.cpu_flat_adam()
    got = _run(lambda ps, w: Twin(ps, lr=5e-4, eps=1e-8, weight_decay=w, decay_unused=True), torch.device("cpu"), wd)
    want = _run(_torch_adam, torch.device("cpu"), wd)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)


def test_twin_skips_parameters_without_gradient_like_torch_adam():
    """weight_decay > 0 (the reference's --weight_decay, opt.py:84): torch.optim.Adam leaves a parameter whose .grad is None
    alone; the flat step does the same for a tensor whose gradient slice is zero, unless decay_unused=True asks otherwise."""
    Twin = common.cpu_flat_adam()
    cpu, unused = torch.device("cpu"), (2, 3)
    want = _run(_torch_adam, cpu, 0.01, unused=unused)
    got = _run(lambda ps, w: Twin(ps, lr=5e-4, eps=1e-8, weight_decay=w), cpu, 0.01, unused=unused)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)
    start = [p.detach().numpy() for p in _params(cpu)]
    for k in unused:
        np.testing.assert_array_equal(got[k], start[k])
    decayed = _run(lambda ps, w: Twin(ps, lr=5e-4, eps=1e-8, weight_decay=w, decay_unused=True), cpu, 0.01, unused=unused)
    assert not np.array_equal(decayed[2], start[2])        # the every-element step does decay them


def test_trainer_takes_the_reference_weight_decay_option():
    """NSFFTrainer(weight_decay > 0) (opt.py:84 --weight_decay) builds its optimizer and steps; parameters the step never
    uses stay exactly where they were (round-3 advisor finding: the constructor raised).  Checkpoints: before the first
    step, partial, and in Lightning's ``optimizer_states`` layout."""
    import nsff_pl_amd as A
    from nsff_pl_amd import training
    torch.manual_seed(0)
    models = {"fine": A.NeRF("fine", use_viewdir=False)}
    emb = {"xyz": A.PosEmbedding(9, 10), "dir": A.PosEmbedding(3, 4)}

    def cpu_render(models_, embeddings_, rays, ts, max_t, N_samples, *a, **kw):
        m = models_["fine"]
        h = torch.sigmoid(rays[:, :3] @ m.static_xyz_encoding_1[0].weight[:3, :3] + m.static_rgb[0].bias)
        return {"rgb_fine": h, "depth_fine": (rays[:, 3:] ** 2).sum(1) * m.static_sigma.bias.abs().sum()}
    old, training.render_rays = training.render_rays, cpu_render
    try:
        tr = training.NSFFTrainer(models, emb, 30, dict(N_samples=8, perturb=0, noise_std=0, weight_decay=0.01),
                                  output_transient=False, optimizer_cls=common.cpu_flat_adam())
        ck0 = tr.checkpoint()                                 # no optimizer yet: must not raise
        assert ck0["optimizer"] is None
        tr.on_train_epoch_start(0)
        before = {n: p.detach().clone() for n, p in models["fine"].named_parameters()}
        g = torch.Generator().manual_seed(3)
        for _ in range(2):
            tr.step(dict(rays=torch.randn(16, 6, generator=g), rgbs=torch.rand(16, 3, generator=g),
                         disps=torch.rand(16, generator=g) + 0.1))
        after = dict(models["fine"].named_parameters())
        assert not torch.equal(after["static_rgb.0.bias"].detach(), before["static_rgb.0.bias"])
        assert torch.equal(after["static_xyz_encoding_5.0.weight"].detach(), before["static_xyz_encoding_5.0.weight"])  # unused: not decayed
        # a partial checkpoint (one tensor missing) loads non-strictly like the reference's load_ckpt, strictly raises
        ck = tr.checkpoint()
        ck["state_dict"].pop("nerf_fine.static_sigma.bias")
        assert tr.load_checkpoint(ck) == ["nerf_fine.static_sigma.bias"]
        with pytest.raises(KeyError):
            tr.load_checkpoint(ck, strict=True)
        # Lightning layout: optimizer state under optimizer_states[0]
        full = tr.checkpoint()
        steps = float(tr.optimizer.state[0])
        pl = {"state_dict": full["state_dict"], "optimizer_states": [full["optimizer"]], "epoch": 3}
        tr.optimizer.reset_state()
        tr.load_checkpoint(pl)
        assert float(tr.optimizer.state[0]) == steps and tr.current_epoch == 3
        assert float(tr.optimizer.exp_avg.abs().sum()) > 0
    finally:
        training.render_rays = old


def test_trainer_graph_auto_is_decided_from_the_batch():
    """NSFFTrainer(graph="auto"): replayed hipGraphs for steps the host's launch rate bounds (few field-point evaluations), the
    eager step for large ones, for CPU tensors and for topk < 1; nothing else is accepted as a string"""
    import nsff_pl_amd as A
    from nsff_pl_amd.training import NSFFTrainer
    models = {"fine": A.NeRF("fine", use_viewdir=False, encode_transient=True, output_flow=True)}
    emb = {"xyz": A.PosEmbedding(9, 10), "dir": A.PosEmbedding(3, 4), "t": torch.nn.Embedding(30, 48)}

    class Rays:                                  # (what _resolve_graph reads of batch["rays"]; no GPU in this test)
        def __init__(self, n, cuda):
            self.shape, self.is_cuda = (n, 6), cuda
    cases = [(dict(N_samples=128, N_importance=0), 512, True, True),        # README.md:226-233: 512 x 128 x 3 = 197 k evaluations
             (dict(N_samples=64, N_importance=64), 1024, True, False),      # C2: 1024 x (64 + 192 x 3) = 655 k
             (dict(N_samples=128, N_importance=0), 512, False, False),      # CPU tensors
             (dict(N_samples=128, N_importance=0, topk=0.5), 512, True, False)]
    for hp, n, cuda, want in cases:
        tr = NSFFTrainer(models, emb, 30, hp, torch.eye(3), torch.zeros(1, 30, 3, 4), graph="auto", optimizer_cls=common.cpu_flat_adam())
        assert tr.graph is False
        tr.on_train_epoch_start = lambda epoch: None            # (the captured loss's device scalars: not on a CPU run)
        tr._resolve_graph({"rays": Rays(n, cuda)})
        assert tr.graph is want and tr._graph_auto is False and tr.loss.static_shapes is want, (hp, n, cuda)
    with pytest.raises(ValueError):
        NSFFTrainer(models, emb, 30, None, torch.eye(3), torch.zeros(1, 30, 3, 4), graph="sometimes")


def test_flat_adam_refuses_cpu_parameters():
    with pytest.raises(RuntimeError, match="HIP device only"):
        FlatAdam(_params(torch.device("cpu")))


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_native_adam_equals_torch_adam(wd, hip_lib):
    dev = torch.device("cuda:0")
    got = _run(lambda ps, w: FlatAdam(ps, lr=5e-4, eps=1e-8, weight_decay=w, decay_unused=True), dev, wd)
    want = _run(_torch_adam, dev, wd)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9)


@pytest.mark.gpu
def test_native_adam_skips_parameters_without_gradient_like_torch_adam(hip_lib):
    """nsff_adam_step_segments: with weight_decay > 0 a tensor whose gradient slice is zero keeps value and moments (torch:
    grad is None -> skipped); tensor boundaries are not float4-aligned in SHAPES, so straddling float4s are exercised."""
    dev, unused = torch.device("cuda:0"), (1, 3, 4)
    want = _run(_torch_adam, dev, 0.01, unused=unused)
    got = _run(lambda ps, w: FlatAdam(ps, lr=5e-4, eps=1e-8, weight_decay=w), dev, 0.01, unused=unused)
    start = [p.detach().cpu().numpy() for p in _params(dev)]
    for k, (a, b) in enumerate(zip(got, want)):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9)
        if k in unused:
            np.testing.assert_array_equal(a, start[k])


@pytest.mark.gpu
def test_native_adam_views_state_and_graph_capture(hip_lib):
    dev = torch.device("cuda:0")
    params = _params(dev)
    before = [p.detach().clone() for p in params]
    opt = FlatAdam(params, lr=1e-3)
    assert opt.in_place() and all(torch.equal(p.detach(), b) for p, b in zip(params, before))   # adoption keeps the values
    assert opt.flat_param.numel() % 4 == 0 and opt.numel == sum(int(np.prod(s)) for s in SHAPES)
    for p in params:
        p.grad.fill_(0.5)                                   # .grad is a view of the flat buffer
    assert float(opt.flat_grad[:opt.numel].min()) == 0.5
    # eager step, then the same step replayed from a hipGraph: identical trajectories
    ref = FlatAdam(_params(dev), lr=1e-3)
    ref.flat_grad.copy_(opt.flat_grad)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step()                                          # warm-up
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()
    graph.replay()
    graph.replay()                                          # warm-up + two replays = three steps
    for _ in range(3):
        ref.step()
    torch.cuda.synchronize()
    assert float(opt.state[0]) == float(ref.state[0]) == 3.0
    assert torch.equal(opt.flat_param, ref.flat_param)
    # a caller that swaps a tensor is noticed and re-adopted without losing its values
    params[2].data = torch.full_like(params[2], 2.0)
    assert not opt.in_place()
    opt.gather()
    assert float(params[2].min()) == 2.0 and opt.in_place()
    sd = opt.state_dict()
    other = FlatAdam(_params(dev), lr=5e-4)
    other.load_state_dict(sd)
    assert float(other.state[0]) == 3.0 and torch.equal(other.exp_avg, opt.exp_avg) and float(other.lr) == float(opt.lr)


@pytest.mark.gpu
def test_optimizer_state_round_trips_through_torch_adam(hip_lib):
    """Resume a torch / reference run here and hand a run back: moments, step count and hyper-parameters convert both ways,
    and the continued trajectories agree."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    grads = [[(torch.randn(*s, generator=g) * 0.1).to(dev) for s in SHAPES] for _ in range(5)]

    def run(opt, params, which):
        for i in which:
            for p, gr in zip(params, grads[i]):
                if p.grad is None:
                    p.grad = gr.clone()
                else:
                    p.grad.copy_(gr)
            opt.step()
    # three steps in torch, then two here
    pt = _params(dev)
    ot = _torch_adam(pt, 0.0)
    run(ot, pt, range(3))
    pf = [torch.nn.Parameter(p.detach().clone()) for p in pt]
    of = FlatAdam(pf, lr=1.0)                               # (hyper-parameters come from the state)
    of.load_torch_state_dict(ot.state_dict())
    assert float(of.state[0]) == 3.0 and abs(float(of.lr) - 5e-4) < 1e-9
    run(of, pf, range(3, 5))
    run(ot, pt, range(3, 5))
    for a, b in zip(pf, pt):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-9)
    # ... and back: a fresh torch Adam continues from this state
    back = of.torch_state_dict()
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in pf]
    o2 = _torch_adam(p2, 0.0)
    o2.load_state_dict(back)
    assert int(o2.state[p2[0]]["step"]) == 5
    assert torch.equal(o2.state[p2[1]]["exp_avg"], back["state"][1]["exp_avg"])
    # clones, not views of the flat buffers
    keep = back["state"][0]["exp_avg"].clone()
    of.exp_avg.add_(1.0)
    assert torch.equal(back["state"][0]["exp_avg"], keep)


@pytest.mark.gpu
def test_trainer_checkpoint_is_detached_and_restores(hip_lib):
    import scenes
    import nsff_pl_amd as A
    from nsff_pl_amd.training import NSFFTrainer
    dev = "cuda:0"
    cfg, meta, rays, ts, models, emb, _, _ = common.build_case("g3_nsff_train", A.NeRF, A.PosEmbedding)
    Ks, Ps, _ = scenes.camera_buffers()
    hp = dict(N_samples=cfg["N_samples"], N_importance=cfg["N_importance"], perturb=0, noise_std=0)
    tr = NSFFTrainer(models, emb, scenes.N_FRAMES, hp, Ks, Ps, output_transient_flow=cfg["flow"]).to(dev)
    tr.on_train_epoch_start(0)
    batch = {k: v.to(dev) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
    batch["rays"] = rays.to(dev)
    tr.step(batch)
    ck = tr.checkpoint()
    assert {k.split(".")[0] for k in ck["state_dict"]} == {"nerf_fine", "nerf_coarse", "embedding_t"}
    flat_lo, flat_hi = tr.optimizer.flat_param.data_ptr(), tr.optimizer.flat_param.data_ptr() + 4 * tr.optimizer.flat_param.numel()
    assert all(not (flat_lo <= v.data_ptr() < flat_hi) for v in ck["state_dict"].values())     # clones, not views
    at_ckpt = {k: v.clone() for k, v in ck["state_dict"].items()}
    tr.step(batch); tr.step(batch)
    assert all(torch.equal(ck["state_dict"][k], at_ckpt[k]) for k in at_ckpt)               # training did not touch it
    after = tr.checkpoint()
    assert any(not torch.equal(after["state_dict"][k], at_ckpt[k]) for k in at_ckpt)
    tr.load_checkpoint(ck)
    back = tr.checkpoint()
    assert all(torch.equal(back["state_dict"][k], at_ckpt[k]) for k in at_ckpt)
    assert int(back["optimizer"]["state"][0]["step"]) == 1 and tr.optimizer.in_place()
