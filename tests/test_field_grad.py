"""The native field node of the backward graph (nsff_pl_amd/field_grad.py) against float64 autograd of the
torch expression of the same network: gradients w.r.t. every parameter, the points and the time codes."""
import copy

import numpy as np
import pytest
import torch

import scenes
import nsff_pl_amd as A
import torch_path
from nsff_pl_amd import field_grad

pytestmark = pytest.mark.gpu


def _reference_grads(model, xyz, t_rows, s, freqs, cot, static, transient, dt=torch.float64):
    m64 = copy.deepcopy(model).to(dt)
    x = xyz.detach().clone().to(dt).requires_grad_(True)
    t = t_rows.detach().clone().to(dt).requires_grad_(True)
    out = torch_path.field(m64, torch_path.pos_embed(x, freqs), None, None, t.repeat_interleave(s, 0),
                           static, transient, ("fw", "bw") if (transient and m64.output_flow) else ())
    cols = {"rgb_s": slice(0, 3), "sigma_s": 3, "rgb_t": slice(4, 7), "sigma_t": 7, "fw": slice(8, 11), "bw": slice(11, 14)}
    loss = sum((out[k] * cot[:, c].to(dt)).sum() for k, c in cols.items() if k in out)
    loss.backward()
    grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m64.named_parameters()}
    return grads, x.grad, t.grad


@pytest.mark.parametrize("static,transient", [(True, True), (False, True)])
@pytest.mark.parametrize("spread", [0.0, 6.0])
def test_field_node_gradients(static, transient, spread, hip_lib):
    """spread > 0: per-point cotangent magnitudes spanning 10^spread (the dynamic range real losses produce)."""
    dev = torch.device("cuda:0")
    cfg = scenes.CASES["g3_nsff_train"]
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    model = models["fine"].to(dev)
    g = torch.Generator().manual_seed(5)
    n_rays, s = 48, 40
    P = n_rays * s
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    t_rows = torch.randn(n_rays, scenes.N_TAU, generator=g).to(dev)
    cot = torch.randn(P, 16, generator=g)
    cot *= 10.0 ** (-spread * torch.rand(P, 1, generator=g))
    cot = cot.to(dev)
    freqs = [float(f) for f in emb["xyz"].freqs]

    ref_p, ref_x, ref_t = _reference_grads(model, xyz, t_rows, s, freqs, cot, static, transient)
    # the same network differentiated by torch in fp32: how far fp32 itself sits from the fp64 gradient
    f32_p, f32_x, f32_t = _reference_grads(model, xyz, t_rows, s, freqs, cot, static, transient, torch.float32)

    for p in model.parameters():
        p.grad = None
    # inside render_pass the static trunk only ever sees points without gradient: static + d(points) is not built
    x = xyz.detach().clone().requires_grad_(not static)
    t = t_rows.detach().clone().requires_grad_(True)
    raw = field_grad.field(model, x, freqs, t, s, static, transient)
    (raw * cot).sum().backward()
    torch.cuda.synchronize()

    def rel(a, b, scale=None):
        return float((a.double() - b).abs().max() / (scale if scale is not None else b.abs().max()).clamp_min(1e-300))
    worst, base = {}, {}
    for n, p in model.named_parameters():
        if ref_p[n].abs().max() == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n
            continue
        # a bias gradient is a plain sum over points that may cancel: measure it on the scale of its layer
        layer = n.rsplit(".", 1)[0]
        scale = torch.stack([ref_p[layer + ".weight"].abs().max(), ref_p[layer + ".bias"].abs().max()]).max()
        worst[n], base[n] = rel(p.grad, ref_p[n], scale), rel(f32_p[n], ref_p[n], scale)
    worst["t"], base["t"] = rel(t.grad, ref_t), rel(f32_t, ref_t)
    if x.requires_grad:
        worst["xyz"], base["xyz"] = rel(x.grad, ref_x), rel(f32_x, ref_x)
    # fp16 operands: 5e-4 per rounding, a handful of roundings along the chain; with `spread` a few points
    # dominate every sum, so the rounding noise of single terms is not averaged away
    tol = 2e-3 if spread == 0 else 6e-3
    bad = {k: (v, base[k]) for k, v in worst.items() if v > tol + 3 * base[k]}
    k = max(worst, key=worst.get)
    print("\nworst native", static, spread, worst[k], k, "fp32 torch there", base[k], "| fp32 torch worst", max(base.values()))
    assert not bad, bad
    if spread == 0:
        # self-test of this tolerance: a 1 % systematic error in ANY weight gradient (or in the input gradients) must be
        # flagged -- the end-to-end gradient tests cannot promise that for the ill-conditioned flow paths, this one does
        blind = []
        for n, p in list(model.named_parameters()) + [("t", t)]:
            if n.endswith(".bias") or n not in worst:
                continue                       # (a bias gradient is judged on the scale of its layer: its own 1 % can be below it;
                                               #  d/d(xyz) runs through sin(512 x): torch's own fp32 result is 3e-3 off there)
            if n == "t":
                err = rel(1.01 * p.grad, ref_t)
            else:
                layer = n.rsplit(".", 1)[0]
                scale = torch.stack([ref_p[layer + ".weight"].abs().max(), ref_p[layer + ".bias"].abs().max()]).max()
                err = rel(1.01 * p.grad, ref_p[n], scale)
            if not err > tol + 3 * base[n]:
                blind.append((n, err, tol + 3 * base[n]))
        assert not blind, blind


def test_failed_backward_does_not_lose_later_weight_gradients(hip_lib):
    """A backward pass that raises after a field node ran leaves deferred weight gradients behind (the engine drops
    the queued end-of-pass callbacks); the next pass must still deliver every trunk gradient, and the default
    (non-deferred) node must hand its gradients to torch.autograd.grad."""
    dev = torch.device("cuda:0")
    cfg = scenes.CASES["g3_nsff_train"]
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    model = models["fine"].to(dev)
    freqs = [float(f) for f in emb["xyz"].freqs]
    g = torch.Generator().manual_seed(9)
    xyz = (torch.rand(256, 3, generator=g) * 2 - 1).to(dev)
    t_rows = torch.randn(4, scenes.N_TAU, generator=g).to(dev)
    trunk = model.transient_xyz_encoding_3[0].weight

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, gr):
            raise RuntimeError("boom")

    t_leaf = t_rows.clone().requires_grad_(True)

    def loss(with_failure):
        # the time codes pass through a node whose backward raises: it runs AFTER the field node (it is upstream of it)
        t_in = Boom.apply(t_leaf) if with_failure else t_rows
        return field_grad.field(model, xyz, freqs, t_in, 64, True, True).sum()

    # default: gradients come back through autograd itself
    got = torch.autograd.grad(loss(False), [trunk])[0]
    assert got is not None and float(got.abs().sum()) > 0
    with field_grad.deferred_weight_grads():
        for p in model.parameters():
            p.grad = None
        try:
            loss(True).backward()
        except RuntimeError as e:
            assert "boom" in str(e)
        for p in model.parameters():
            p.grad = None
        loss(False).backward()
        torch.cuda.synchronize()
        assert trunk.grad is not None
        assert torch.allclose(trunk.grad, got, rtol=1e-5, atol=1e-6 * float(got.abs().max()))
        # what the failed pass queued is never delivered (the assert above) and stays in the list only until the next
        # drop_stale_pending() (NSFFTrainer.step calls it): the end-of-pass flush removes ITS pass's entries and nothing
        # else, so that a re-entrant backward cannot throw away what the enclosing pass has queued
        field_grad.drop_stale_pending()
        assert not field_grad._PENDING


def test_a_forward_inside_a_backward_pass_keeps_the_queued_weight_gradients(hip_lib):
    """Deferred mode: a field forward that runs WHILE a backward pass is in flight (activation recomputation, a hook that
    renders) must not discard the weight gradients the pass has already queued."""
    dev = torch.device("cuda:0")
    cfg = scenes.CASES["g3_nsff_train"]
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    model = models["fine"].to(dev)
    freqs = [float(f) for f in emb["xyz"].freqs]
    g = torch.Generator().manual_seed(13)
    xyz = (torch.rand(256, 3, generator=g) * 2 - 1).to(dev)
    t_rows = torch.randn(4, scenes.N_TAU, generator=g).to(dev)
    trunk = model.transient_xyz_encoding_3[0].weight
    want = torch.autograd.grad(field_grad.field(model, xyz, freqs, t_rows, 64, True, True).sum(), [trunk])[0]
    t_leaf = t_rows.clone().requires_grad_(True)
    ran = []

    def hook(grad):                                # runs after the field node's backward (the time codes are upstream of it)
        with torch.no_grad():
            field_grad.field(model, xyz, freqs, t_rows, 64, True, True)
        ran.append(len(field_grad._PENDING))
        return grad
    t_leaf.register_hook(hook)
    for p in model.parameters():
        p.grad = None
    with field_grad.deferred_weight_grads():
        field_grad.field(model, xyz, freqs, t_leaf, 64, True, True).sum().backward()
    torch.cuda.synchronize()
    assert ran == [1] and not field_grad._PENDING
    assert trunk.grad is not None and torch.allclose(trunk.grad, want, rtol=1e-5, atol=1e-6 * float(want.abs().max()))


def test_a_backward_inside_a_backward_pass_keeps_the_enclosing_pass_entries(hip_lib):
    """Deferred mode, re-entrant: a hook of the outer pass renders AND differentiates (its own backward pass, its own
    end-of-pass flush).  That inner flush must deliver the inner gradients and leave what the outer pass has queued alone:
    afterwards .grad holds outer + inner (round-3 advisor finding: the inner flush emptied the whole list)."""
    dev = torch.device("cuda:0")
    cfg = scenes.CASES["g3_nsff_train"]
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    model = models["fine"].to(dev)
    freqs = [float(f) for f in emb["xyz"].freqs]
    g = torch.Generator().manual_seed(17)
    xyz = (torch.rand(256, 3, generator=g) * 2 - 1).to(dev)
    xyz2 = (torch.rand(128, 3, generator=g) * 2 - 1).to(dev)
    t_rows = torch.randn(4, scenes.N_TAU, generator=g).to(dev)
    trunk = model.transient_xyz_encoding_3[0].weight
    outer = torch.autograd.grad(field_grad.field(model, xyz, freqs, t_rows, 64, True, True).sum(), [trunk])[0]
    inner = torch.autograd.grad(field_grad.field(model, xyz2, freqs, t_rows[:2], 64, True, True).square().sum(), [trunk])[0]
    t_leaf = t_rows.clone().requires_grad_(True)
    seen = []

    def hook(grad):                                # runs after the outer field node's backward queued its entry
        before = len(field_grad._PENDING)
        with torch.enable_grad():
            field_grad.field(model, xyz2, freqs, t_rows[:2], 64, True, True).square().sum().backward()
        seen.append((before, len(field_grad._PENDING)))
        return grad
    t_leaf.register_hook(hook)
    for p in model.parameters():
        p.grad = None
    with field_grad.deferred_weight_grads():
        field_grad.field(model, xyz, freqs, t_leaf, 64, True, True).sum().backward()
    torch.cuda.synchronize()
    assert seen == [(1, 1)] and not field_grad._PENDING          # the inner pass took its own entry, the outer one survived it
    want = outer + inner
    assert torch.allclose(trunk.grad, want, rtol=1e-5, atol=2e-6 * float(want.abs().max()))


@pytest.fixture(params=["f16", "f16x3"])
def grad_precision(request):
    """both backward arithmetics (config.set_grad_precision): one f16 product per multiply-accumulate, and three"""
    A.config.set_grad_precision(request.param)
    try:
        yield request.param
    finally:
        A.config.set_grad_precision("f16")


@pytest.mark.gpu
@pytest.mark.parametrize("scene,static", [("g3_nsff_train", True), ("g3_nsff_train", False), ("g13_viewdir_train", True)])
def test_in_place_accumulation_equals_the_returned_gradients(scene, static, hip_lib, grad_precision):
    """Deferred mode with .grad tensors in place: nsff_weight_grad_accumulate adds every gradient element straight into
    the parameters' own memory.  Result must be bit-identical to (existing .grad) + (what the node returns through
    autograd), for scattered .grad tensors and for views of one flat buffer, twice in a row (accumulation).  The FOLDED
    parameters (*_final and the heads / the view-direction layer that read it) are products of the folded heads' gradient --
    nsff_fold_grads in place, torch GEMMs on the returned path: the same sums in another order, equal to fp32 rounding."""
    dev = torch.device("cuda:0")
    cfg = scenes.CASES[scene]
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    model = models["fine"].to(dev)
    freqs = [float(f) for f in emb["xyz"].freqs]
    g = torch.Generator().manual_seed(11)
    n_rays, s = 8, 64
    xyz = (torch.rand(n_rays * s, 3, generator=g) * 2 - 1).to(dev)
    t_rows = torch.randn(n_rays, scenes.N_TAU, generator=g).to(dev)
    side = {}
    if model.use_viewdir and static:
        side["dir_rows"] = torch.randn(n_rays, model.in_channels_dir, generator=g).to(dev)
        if model.in_channels_a > 0:
            side["a_rows"] = torch.randn(n_rays, model.in_channels_a, generator=g).to(dev)
    cot = torch.randn(n_rays * s, 16, generator=g).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    names = {id(p): n for n, p in model.named_parameters()}
    viewdir = model.use_viewdir and static
    folded = lambda n: ("xyz_encoding_final" in n or n.startswith(("transient_rgb", "transient_sigma", "transient_flow"))
                        or (n.startswith("static_dir_encoding") if viewdir else n.startswith("static_rgb")))

    def loss():
        return (field_grad.field(model, xyz, freqs, t_rows, s, static, True, **side) * cot).sum()

    want = torch.autograd.grad(loss(), params, allow_unused=True)
    start = [torch.randn(p.shape, generator=g).to(dev) * 1e-3 for p in params]
    for layout in ("scattered", "flat"):
        if layout == "flat":
            flat = torch.cat([t.reshape(-1) for t in start])
            off = 0
            for p, t in zip(params, start):
                p.grad = flat[off:off + t.numel()].view_as(p)
                off += t.numel()
        else:
            for p, t in zip(params, start):
                p.grad = t.clone()
        with field_grad.deferred_weight_grads():
            loss().backward()
            loss().backward()
        torch.cuda.synchronize()
        assert not field_grad._PENDING
        n_folded = 0
        for p, t, w in zip(params, start, want):
            expect = t if w is None else (t + w) + w
            if w is not None and folded(names[id(p)]):
                n_folded += 1
                assert float((p.grad - expect).abs().max()) <= 4e-6 * float(w.abs().max()), (layout, names[id(p)])
            else:
                assert torch.equal(p.grad, expect), (layout, names[id(p)])
        assert n_folded >= (10 if static or not viewdir else 0)
    assert field_grad._GRAD_MAPS, "the in-place path was not taken"
    from nsff_pl_amd import _lib
    assert (_lib.last_bwd_kernel() == "x3") == (grad_precision == "f16x3")


@pytest.mark.gpu
def test_flow_grad_matches_the_torch_expression(hip_lib):
    """nsff_flow_grad (the backward of the flow glue, rendering.py:187-188,218,224,226-232) against autograd of the torch ops
    it replaces: where(z > 0.95, 0, raw[:, c:c+3]) consumed by several users, overwrite and accumulate form."""
    from nsff_pl_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    P = 1000                                               # not a multiple of the block size
    zs = torch.rand(P, generator=g).to(dev)
    raw = torch.randn(P, 16, generator=g).to(dev).requires_grad_(True)
    cot = [torch.randn(P, 3, generator=g).to(dev) for _ in range(5)]
    far = (zs > 0.95)[:, None]
    assert 0 < int(far.sum()) < P
    f_fw = torch.where(far, torch.zeros((), device=dev), raw[:, 8:11])
    f_bw = torch.where(far, torch.zeros((), device=dev), raw[:, 11:14])
    want, = torch.autograd.grad((f_fw * (cot[0] + cot[1] + cot[2])).sum() + (f_bw * (cot[3] + cot[4])).sum(), raw)
    got = torch.full((P, 16), float("nan"), device=dev)
    _lib.flow_grad(zs, 0.95, got, False, 8, cot[:3], 11, cot[3:])
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    # accumulate: only the named columns change
    base = torch.randn(P, 16, generator=g).to(dev)
    acc = base.clone()
    _lib.flow_grad(zs, 0.95, acc, True, 11, [cot[0]])
    want2 = base.clone()
    want2[:, 11:14] += torch.where(far, torch.zeros((), device=dev), cot[0])
    assert torch.allclose(acc, want2, rtol=1e-6, atol=1e-6)
    # one group only, overwrite: the other group's columns are zero
    one = torch.full((P, 16), float("nan"), device=dev)
    _lib.flow_grad(zs, 0.95, one, False, -1, [], 11, [cot[3]])
    assert torch.equal(one[:, :11], torch.zeros(P, 11, device=dev)) and torch.equal(one[:, 14:], torch.zeros(P, 2, device=dev))


@pytest.mark.gpu
def test_time_rows_node_matches_embedding_autograd(hip_lib):
    """autograd._TimeRows (E[ts], E[clamp(ts+1)], E[clamp(ts-1)] with one native backward) against nn.Embedding autograd, frames
    at both ends of the clamp included, one cotangent absent."""
    from nsff_pl_amd import autograd as ag
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n_frames, width, n = 12, 48, 700
    emb = torch.nn.Embedding(n_frames, width).to(dev)
    ts = torch.randint(0, n_frames, (n,), generator=g).to(dev)
    ts[:3] = torch.tensor([0, n_frames - 1, n_frames - 2], device=dev)
    max_t = n_frames - 1
    cots = [torch.randn(n, width, generator=g).to(dev) for _ in range(3)]
    cur, nxt, prv = ag.time_rows(emb, ts, max_t, True)
    ref = [emb(ts), emb(torch.clamp(ts + 1, max=max_t)), emb(torch.clamp(ts - 1, min=0))]
    for a, b in zip((cur, nxt, prv), ref):
        assert torch.equal(a, b)
    want, = torch.autograd.grad(sum((r * c).sum() for r, c in zip(ref, cots)), emb.weight)
    got, = torch.autograd.grad(sum((r * c).sum() for r, c in zip((cur, nxt, prv), cots)), emb.weight)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    cur, nxt, prv = ag.time_rows(emb, ts, max_t, True)
    ref = [emb(ts), None, emb(torch.clamp(ts - 1, min=0))]
    want, = torch.autograd.grad((ref[0] * cots[0]).sum() + (ref[2] * cots[2]).sum(), emb.weight)
    got, = torch.autograd.grad((cur * cots[0]).sum() + (prv * cots[2]).sum(), emb.weight)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    only, _, _ = ag.time_rows(emb, ts, max_t, False)                     # no neighbours: the plain gather
    assert torch.equal(only, emb(ts))


@pytest.mark.gpu
@pytest.mark.parametrize("viewdir", [False, True])
def test_fold_gradient_kernels_match_the_torch_algebra(viewdir, hip_lib):
    """The gradients of *_xyz_encoding_final and of the layers that read it are small products of the folded layer's weight-gradient
    sums.  Until round 5 two of the three code paths computed them with torch.addmm / `@` (rocBLAS); now every path is a HIP
    kernel: nsff_fold_grads (heads, accumulating), nsff_fold_grads_dense (the 256-row view-direction layer, and every caller that
    wants tensors).  Both forms -- tensors returned (field_grad._folded_grads) and accumulated in place (_fold_in_place, onto
    non-zero gradients) -- against tests/torch_path.py::folded_grads_reference in float64; fp32 products of 256 terms: 2e-6."""
    from nsff_pl_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = A.NeRF('fine', use_viewdir=viewdir, encode_appearance=viewdir, in_channels_a=48 if viewdir else 0, encode_transient=True,
                   output_flow=True).to(dev)
    plist = _lib.param_list(model)
    meta = field_grad._wgrad_jobs(model, True, True)
    mats = {i: torch.randn(field_grad.job_shape(model, k), device=dev) for i, (k, _, _) in enumerate(meta)}
    rows = {i: torch.randn(256, device=dev) for i in mats}
    m64 = copy.deepcopy(model).double()
    ref = torch_path.folded_grads_reference(m64, True, True, meta, _lib.param_list(m64), lambda i: mats[i].double(), lambda i: rows[i].double())
    got = field_grad._folded_grads(model, True, True, meta, plist, lambda i: mats[i], lambda i: rows[i])
    torch.cuda.synchronize()
    assert set(got) == set(ref) and len(got) == 14
    names = {id(p): n for n, p in model.named_parameters()}
    for i, g in got.items():
        r = ref[i]
        assert g.shape == r.shape == plist[i].shape, names[id(plist[i])]
        err = (g.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        assert err < 2e-6, (names[id(plist[i])], err)
    # in place, onto gradients that are already there
    for p in plist:
        p.grad = torch.randn_like(p)
    before = {i: plist[i].grad.clone() for i in ref}
    field_grad._fold_in_place(model, True, True, meta, plist, lambda i: mats[i], lambda i: rows[i])
    torch.cuda.synchronize()
    for i, r in ref.items():
        want = before[i].double() + r
        err = (plist[i].grad.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
        assert err < 2e-6, (names[id(plist[i])], err)
    # a frozen member of the fold: the trunk takes the tensor route, frozen parameters receive nothing
    fin = model.transient_xyz_encoding_final
    fin.weight.requires_grad_(False)
    for p in plist:
        p.grad = torch.zeros_like(p) if p.requires_grad else None
    field_grad._fold_in_place(model, True, True, meta, plist, lambda i: mats[i], lambda i: rows[i])
    torch.cuda.synchronize()
    assert fin.weight.grad is None
    for i, r in ref.items():
        if plist[i] is fin.weight:
            continue
        err = (plist[i].grad.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        assert err < 2e-6, (names[id(plist[i])], err)
