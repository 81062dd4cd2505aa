"""CPU-only checks of the host side: module/state_dict contract, flag plumbing, the C-ABI
library's exported symbols and its argument validation (no GPU compute is launched)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import scenes
import nsff_pl_amd as A
from nsff_pl_amd import _lib, dist as ndist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_contract_matches_reference_probe():
    """Key names, shapes and parameter counts recorded from the reference."""
    fine = A.NeRF('fine', use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True)
    coarse = A.NeRF('coarse', use_viewdir=False, encode_transient=True, in_channels_t=48)
    assert sum(p.numel() for p in fine.parameters()) == 1145870
    assert sum(p.numel() for p in coarse.parameters()) == 1144328
    sd = fine.state_dict()
    assert sd['static_xyz_encoding_1.0.weight'].shape == (256, 63)
    assert sd['static_xyz_encoding_5.0.weight'].shape == (256, 319)
    assert sd['transient_xyz_encoding_1.0.weight'].shape == (256, 111)
    assert sd['transient_xyz_encoding_5.0.weight'].shape == (256, 367)
    assert sd['static_sigma.weight'].shape == (1, 256) and sd['static_rgb.0.weight'].shape == (3, 256)
    assert sd['transient_flow_fw.0.weight'].shape == (3, 256) and 'transient_flow_bw.0.bias' in sd
    assert 'transient_flow_fw.0.weight' not in coarse.state_dict()
    assert 'static_dir_encoding.0.weight' not in sd
    vd = A.NeRF('fine', use_viewdir=True, encode_appearance=True, encode_transient=True, in_channels_t=48,
                output_flow=True)
    assert vd.state_dict()['static_dir_encoding.0.weight'].shape == (256, 256 + 27 + 48)
    assert sum(p.numel() for p in A.NeRF('fine', use_viewdir=True, encode_transient=True, in_channels_t=48,
                                         output_flow=True).parameters()) == 1218574
    # coarse models never take the appearance code (reference nerf.py:67)
    assert A.NeRF('coarse', encode_appearance=True).encode_appearance is False
    assert A.PosEmbedding(9, 10).state_dict() == {} and len(A.PosEmbedding(9, 10).freqs) == 10
    assert torch.equal(A.PosEmbedding(3, 4).freqs, torch.tensor([1., 2., 4., 8.]))


def test_seeded_weights_reproduce_golden_checksums():
    import common
    for name in scenes.CASES:
        common.build_case(name, A.NeRF, A.PosEmbedding)      # asserts the checksum


def test_header_symbols_are_exported_and_bound():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "nsff_render.h")).read()
    declared = set(re.findall(r"\b(nsff_[a-z_]+)\s*\(", header))
    declared -= set(re.findall(r"static inline [a-z0-9_ ]*?\b(nsff_[a-z_]+)\s*\(", header))     # header-only helpers
    assert declared, "no declarations found in the header"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), f"libnsff_hip.so does not export {sym}"
    assert lib.nsff_abi_version() == _lib.ABI_VERSION == 32


def test_struct_layouts_match_the_header_sizes():
    # natural-alignment layout of the C structs (pointer = 8 bytes)
    assert C.sizeof(_lib.ModelDesc) == 48
    assert C.sizeof(_lib.FieldArgs) == 8 + 24 + 8 + 4 + 4 * _lib.MAX_FREQS + 4 + 3 * 8 + 8 + 4 * 5 + 4 + 8 + 4 * 8 + 8 + 8 + 16 + 24 and _lib.MAX_FREQS == 24
    assert C.sizeof(_lib.TimeBiasJob) == 8 + 64 + 64 + 16 + 16 + 16 + 8 + 8
    assert C.sizeof(_lib.RngJob) == 32 and _lib.MAX_RNG_JOBS == 12 and C.sizeof(_lib.RngCoarse) == 48
    assert C.sizeof(_lib.FieldBwdArgs) == 16 + 8 * 9 and C.sizeof(_lib.WgradJob) == 56
    assert C.sizeof(_lib.SplatArgs) == 12 + 16 + 48 + 4 + 5 * 8 + 16 and C.sizeof(_lib.MpiArgs) == 16 + 7 * 8
    assert C.sizeof(_lib.LossArgs) == 24 + 8 + 8 + 8 * (len(_lib._LOSS_IN) + 3 + len(_lib.LOSS_GRADS))
    n_ptr = len(_lib._COMPOSITE_PTRS_IN) + len(_lib._COMPOSITE_PTRS_OUT)
    assert C.sizeof(_lib.FrustumArgs) == 16 + 16 + 16
    assert C.sizeof(_lib.CompositeArgs) == 40 + 8 * n_ptr + C.sizeof(_lib.FrustumArgs)


def test_layout_and_argument_validation_without_gpu():
    lib = _lib.load()
    m = A.NeRF('fine', use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True)
    d = _lib.model_desc(m)
    n = C.c_size_t()
    assert lib.nsff_packed_bytes(C.byref(d), 1, C.byref(n)) == 0       # f16x3: hi+lo halfs = same bytes/weight
    assert 1.0 <= n.value / 4 / sum(p.numel() for p in m.parameters()) < 1.08
    assert lib.nsff_packed_bytes(C.byref(d), 7, C.byref(n)) == -1
    assert lib.nsff_packed_bytes(C.byref(d), 0, C.byref(n)) == 0
    # every Linear element appears once (+ zero padding of K to multiples of 8, vectors to 4)
    assert n.value // 4 >= sum(p.numel() for p in m.parameters())
    assert n.value // 4 < 1.03 * sum(p.numel() for p in m.parameters())
    assert lib.nsff_param_count(C.byref(d)) == len(_lib.param_list(m)) == 48
    bad = _lib.model_desc(m); bad.W = 128
    assert lib.nsff_packed_bytes(C.byref(bad), 0, C.byref(n)) == -1         # NSFF_ERR_INVALID
    bad = _lib.model_desc(m); bad.skip = 8                                     # a skip layer must be one of 1..D-1
    assert lib.nsff_packed_bytes(C.byref(bad), 0, C.byref(n)) == -1
    bad = _lib.model_desc(m); bad.skip = 0; bad.skip_mask = 0b101              # (layer 0 cannot be a skip layer)
    assert lib.nsff_packed_bytes(C.byref(bad), 1, C.byref(n)) == -1
    assert lib.nsff_packed_bytes(None, 0, C.byref(n)) == -2                    # NSFF_ERR_NULL
    a = _lib.FieldArgs()
    a.n_points, a.pts_per_ray, a.static_mode = 64, 1, 2
    assert lib.nsff_field_query(C.byref(d), None, C.byref(a), None) == -2
    bad = _lib.FieldArgs(); bad.launch_form, bad.n_points = 2, 4
    assert lib.nsff_field_query(C.byref(d), (C.c_char * 64)(), C.byref(bad), None) == -1   # launch_form: 0 (library's choice) or 1 (one workgroup per tile)
    from nsff_pl_amd import config
    assert config.get_persistent() is True
    assert lib.nsff_time_bias_rows(C.byref(d)) == 2                            # layer 0 + the skip layer
    two = _lib.model_desc(m); two.skip = 0; two.skip_mask = 0b100100
    assert lib.nsff_time_bias_rows(C.byref(two)) == 3
    st = _lib.model_desc(A.NeRF('coarse', use_viewdir=False, encode_transient=False))
    assert lib.nsff_time_bias_rows(C.byref(st)) == 0
    assert lib.nsff_time_bias(None, 1, 4, None) == -2
    jobs = (_lib.TimeBiasJob * 1)()
    assert lib.nsff_time_bias(jobs, 0, 4, None) == -1 and lib.nsff_time_bias(jobs, 5, 4, None) == -1
    assert lib.nsff_time_bias(jobs, 1, 4, None) == -2                          # (null members)
    rj = (_lib.RngJob * 2)()
    assert lib.nsff_rng_draws(None, 0, 1, None) == 0 and lib.nsff_rng_draws(None, 1, 1, None) == -2
    assert lib.nsff_rng_draws(rj, _lib.MAX_RNG_JOBS + 1, 1, None) == -1 and lib.nsff_rng_draws(rj, 1, 1, None) == -2    # (null out)
    rj[0].out, rj[0].numel, rj[0].grid, rj[0].offset = 64, 1000, 4, 2        # an offset that is not a multiple of 4
    assert lib.nsff_rng_draws(rj, 1, 1, None) == -1
    rj[0].offset, rj[0].grid = 4, 5                                          # more blocks than torch launches for 1000 elements
    assert lib.nsff_rng_draws(rj, 1, 1, None) == -1
    c = _lib.CompositeArgs()
    c.n_rays, c.n_samples = 4, 0
    assert lib.nsff_composite(C.byref(c), None) == -1
    assert lib.nsff_coarse_samples(None, 4, None, 0, 0.0, None, None, None, None) == -1
    fg = _lib.FlowGradArgs(n_points=8, col_a=8, col_b=11)
    assert C.sizeof(_lib.FlowGradArgs) == 8 + 8 + 4 * 5 + 4 + 8 * 9
    assert lib.nsff_flow_grad(None, None) == -2 and lib.nsff_flow_grad(C.byref(fg), None) == -2      # zs / out missing
    fg.col_b = 9                                                                                     # overlapping groups
    assert lib.nsff_flow_grad(C.byref(fg), None) == -1
    fg.col_a = fg.col_b = -1
    assert lib.nsff_flow_grad(C.byref(fg), None) == -1
    assert lib.nsff_coarse_samples(None, 0, None, 8, 0.0, None, None, None, None) == 0   # empty batch is ok


def test_product_path_refuses_cpu_tensors():
    models = {'fine': A.NeRF('fine', use_viewdir=False)}
    emb = {'xyz': A.PosEmbedding(9, 10), 'dir': A.PosEmbedding(3, 4)}
    with pytest.raises(RuntimeError, match="GPU"):
        A.render_rays(models, emb, torch.zeros(4, 6), None, 0, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        A.PosEmbedding(9, 10)(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        A.sample_pdf(torch.zeros(2, 5), torch.zeros(2, 4), 8, det=True)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nsff_pl_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU oracle", ""), fn


def test_draw_plan_matches_reference_order():
    cfg = scenes.CASES["g7_nsff_train_noise"]
    keys = [k for k, _, _ in scenes.draw_plan(cfg)]
    assert keys == ["perturb", "coarse_static", "coarse_transient", "u_static", "u_transient",
                    "fine_static", "fine_transient", "warp_fw", "warp_bw"]
    keys = [k for k, _, _ in scenes.draw_plan(scenes.CASES["g1_static_c1"])]
    assert keys == ["fine_static"]


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 1024, 147456):
        for world in (1, 2, 3, 8):
            b = [ndist.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_compositing_node_outputs_are_result_keys_of_the_reference():
    """Every tensor the native compositing node hands out is a key the reference's render_rays returns in that mode
    (golden key sets), and the node covers every differentiable per-ray key."""
    import common
    from nsff_pl_amd import composite_grad
    for name, typ_flags in (("g3_nsff_train", dict(coarse=(True, False, False), fine=(True, True, True))),
                            ("g2_static_c2f", dict(coarse=(False, False, False), fine=(False, False, False)))):
        want = common.load_golden(name)[-1] if hasattr(common, "load_golden") else common.build_case(name, A.NeRF, A.PosEmbedding)[-1]
        for typ, (tr, fl, wp) in typ_flags.items():
            keys = [k for k, _ in composite_grad.output_spec(typ, tr, fl, wp)]
            assert len(keys) == len(set(keys))
            assert all(k in want for k in keys), [k for k in keys if k not in want]
    per_ray = {"rgb_fine", "depth_fine", "transient_alpha_fine", "transient_rgb_fine", "_static_rgb_fine",
               "_static_depth_fine", "xyz_fine", "transient_flow_fw", "transient_flow_bw", "rgb_fw", "rgb_bw"}
    assert per_ray <= {k for k, _ in composite_grad.output_spec("fine", True, True, True)}


def test_unsupported_architectures_are_refused_by_name():
    """models/nerf.py:34-40 accepts any W and a list of skips; the kernels take W = 256 and any skip layers among
    1..D-1, for inference and training alike -- the error must say which field."""
    from nsff_pl_amd import field_grad
    for kw, needle in ((dict(W=128), "W=128"), (dict(D=8, skips=[8]), "skips="), (dict(D=8, skips=[0, 4]), "skips="),
                       (dict(D=1, skips=[]), "D=1")):
        m = A.NeRF('fine', use_viewdir=False, **kw)
        with pytest.raises(RuntimeError, match="unsupported NeRF architecture.*" + needle):
            _lib.model_desc(m)
        assert needle.rstrip("=") in field_grad.why_unsupported(m)
    for skips, mask in (([2, 5], 0b100100), ([], 0), ([1, 2, 3], 0b1110)):
        m = A.NeRF('fine', D=6, skips=skips, use_viewdir=False, encode_transient=True)
        d = _lib.model_desc(m)                                                 # several / no skip layers
        assert (d.skip, d.skip_mask) == (0, mask)
        assert field_grad.why_unsupported(m) is None
        n = C.c_size_t(0)
        assert _lib.load().nsff_packed_bytes(C.byref(d), 1, C.byref(n)) == 0 and n.value > 0
        assert _lib.load().nsff_bwd_packed_bytes(C.byref(d), C.byref(n)) == 0 and n.value > 0      # ... train as well
    d = _lib.model_desc(A.NeRF('fine', D=6, skips=[2], use_viewdir=False))
    assert (d.skip, d.skip_mask) == (2, 0)
    # the reference's CLI can widen the embeddings (opt.py:25,41,45): the saved input tiles then have 256 rows
    assert _lib.train_dims(A.NeRF('fine', use_viewdir=False, encode_transient=True)) == (128, 64, 128)
    assert _lib.train_dims(A.NeRF('fine', use_viewdir=False, in_channels_xyz=3 + 6 * 12, encode_transient=True, in_channels_t=96)) == (256, 128, 128)
    assert _lib.train_dims(A.NeRF('fine', use_viewdir=True, in_channels_dir=99, encode_appearance=True, in_channels_a=48)) == (128, 64, 256)
    assert _lib.train_dims(A.NeRF("fine", use_viewdir=False, in_channels_xyz=3 + 6 * 20, encode_transient=True, in_channels_t=128)) == (256, 128, 128)
    too_wide = A.NeRF("fine", use_viewdir=False, in_channels_xyz=3 + 6 * 20, encode_transient=True, in_channels_t=192)   # 128 + 192 > 256
    assert "must fit 256 columns" in field_grad.why_unsupported(too_wide)


@pytest.mark.parametrize("kw,static,transient", [
    (dict(use_viewdir=False, encode_transient=True, output_flow=True), True, True),
    (dict(use_viewdir=False, encode_transient=True, output_flow=True), False, True),
    (dict(use_viewdir=True, encode_appearance=True, in_channels_a=48, encode_transient=True, output_flow=True), True, True),
    (dict(use_viewdir=True, encode_transient=False, D=5, skips=[2]), True, False),
    (dict(use_viewdir=False, encode_transient=True, output_flow=True, skips=[2, 5]), True, True),
    (dict(use_viewdir=False, encode_transient=True, output_flow=True, D=4, skips=[]), True, True),
    (dict(use_viewdir=True, encode_appearance=True, in_channels_a=48, in_channels_dir=99, in_channels_xyz=75,
          encode_transient=True, in_channels_t=96, output_flow=True), True, True)])
def test_gradient_map_names_the_same_elements_as_the_tensor_assembly(kw, static, transient):
    """nsff_weight_grad_accumulate's map (field_grad._grad_map) is built by running the tensor assembly on index
    objects: gathering through those indices must reproduce the tensors, element for element, with one owner each."""
    from nsff_pl_amd import field_grad as fg
    model = A.NeRF('fine', **kw)
    plist = _lib.param_list(model)
    meta = fg._wgrad_jobs(model, static, transient)
    rng = np.random.default_rng(5)
    mats = [rng.standard_normal(fg.job_shape(model, k)).astype(np.float32) for k, _, _ in meta]
    rows = [rng.standard_normal(256).astype(np.float32) for _ in meta]
    sizes = [m.size for m in mats]
    tens = fg._assemble(model, static, transient, meta, plist, lambda i: torch.from_numpy(mats[i]),
                        lambda i: torch.from_numpy(rows[i]), torch.cat)
    idx = fg._assemble(model, static, transient, meta, plist,
                       lambda i: fg._Idx(np.full(mats[i].shape, i, np.int16),
                                         np.arange(sizes[i], dtype=np.int32).reshape(mats[i].shape)),
                       lambda i: fg._Idx(np.full(256, i, np.int16), sizes[i] + np.arange(256, dtype=np.int32)), fg._Idx.cat)

    def fetch(job, e):                      # what the kernel reads: e < size -> matrix element, else row sum e - size
        out = np.zeros(job.shape, np.float32)
        for j in np.unique(job[job >= 0]):
            sel = job == j
            flat = np.concatenate([mats[j].reshape(-1), rows[j]])
            out[sel] = flat[e[sel]]
        return out
    n_grads = 0
    for p, g, ix in zip(plist, tens, idx):
        assert (g is None) == (ix is None)
        if g is None:
            continue
        n_grads += 1
        assert tuple(g.shape) == tuple(p.shape) == ix.ja.shape, (g.shape, p.shape)
        assert (ix.ja >= 0).all()
        got = fetch(ix.ja, ix.ea) + fetch(ix.jb, ix.eb)
        np.testing.assert_array_equal(got, g.numpy())
    # every parameter of the evaluated trunks has exactly one source: a slice of the jobs (above) or the fold's products
    # (which parameters those are: the torch restatement of the fold; the GPU suite holds the product's kernels to the same key set)
    import torch_path
    folded = torch_path.folded_grads_reference(model, static, transient, meta, plist, lambda i: torch.from_numpy(mats[i]),
                                               lambda i: torch.from_numpy(rows[i]))
    direct = {i for i, g in enumerate(tens) if g is not None}
    assert not (direct & set(folded)) and all(tuple(g.shape) == tuple(plist[i].shape) for i, g in folded.items())
    expect = sum(1 for n, _ in model.named_parameters()
                 if (static and n.startswith("static")) or (transient and n.startswith("transient")))
    assert n_grads + len(folded) == expect
    assert sorted(fg._fold_job_indices(model, static, transient, meta)) == sorted(
        i for i, (k, t, l) in enumerate(meta) if k in ("dir_h", "dir_x") or (k == "head" and l == 0 and not (t == 0 and model.use_viewdir and static)))


@pytest.mark.parametrize("viewdir", [False, True])
def test_folded_gradients_are_autograds(viewdir):
    """*_xyz_encoding_final is never executed as a layer: the folded heads' weight gradient G = sum_p dpre_p (x) h_p becomes the
    gradients of *_final and of the heads (of static_dir_encoding with view directions) by four small products -- the algebra
    (tests/torch_path.py::folded_grads_reference) against float64 autograd of the unfolded layers on random activations and
    cotangents; the product's HIP kernels are held to the same algebra by tests/test_field_grad.py (GPU)."""
    from nsff_pl_amd import field_grad as fg
    torch.manual_seed(3)
    model = A.NeRF('fine', use_viewdir=viewdir, encode_appearance=viewdir, in_channels_a=48 if viewdir else 0, encode_transient=True,
                   output_flow=True).double()
    plist = _lib.param_list(model)
    meta = fg._wgrad_jobs(model, True, True)
    res = {tag: i for i, tag in enumerate(meta)}
    P, n_side = 37, model.in_channels_dir + model.in_channels_a
    mats = {i: torch.zeros(fg.job_shape(model, k), dtype=torch.float64) for i, (k, _, _) in enumerate(meta)}
    rows = {i: torch.zeros(256, dtype=torch.float64) for i in mats}
    for p in model.parameters():
        p.grad = None
    loss = 0
    for t, prefix in ((0, "static"), (1, "transient")):
        h = torch.randn(P, 256, dtype=torch.float64)
        final = getattr(model, f"{prefix}_xyz_encoding_final")(h)
        if t == 0 and viewdir:
            side = torch.randn(P, n_side, dtype=torch.float64)
            pre = model.static_dir_encoding[0](torch.cat([final, side], 1))
            cot = torch.randn_like(pre)                           # d loss / d pre-activation of static_dir_encoding
            loss = loss + (pre * cot).sum()
            mats[res[("dir_h", 0, 0)]] = cot.t() @ h
            rows[res[("dir_h", 0, 0)]] = cot.sum(0)
            mats[res[("dir_x", 0, 0)]][:, :n_side] = cot.t() @ side
            continue
        heads = [model.static_rgb] if t == 0 else [model.transient_rgb, model.transient_sigma, model.transient_flow_fw, model.transient_flow_bw]
        pre = torch.cat([m[0](final) if isinstance(m, torch.nn.Sequential) else m(final) for m in heads], 1)
        cot = torch.randn_like(pre)
        loss = loss + (pre * cot).sum()
        i = res[("head", t, 0)]
        mats[i][:pre.shape[1]] = cot.t() @ h                      # rows 0..R-1 (the remainder rows 16.. stay zero)
        rows[i][:pre.shape[1]] = cot.sum(0)
    loss.backward()
    import torch_path
    got = torch_path.folded_grads_reference(model, True, True, meta, plist, lambda i: mats[i], lambda i: rows[i])
    names = {id(p): n for n, p in model.named_parameters()}
    assert len(got) == (4 + 8 + 2 if not viewdir else 4 + 8 + 2)          # static: final + (rgb | dir layer); dynamic: final + four heads
    for i, g in got.items():
        ref = plist[i].grad
        assert ref is not None, names[id(plist[i])]
        np.testing.assert_allclose(g.numpy(), ref.numpy(), rtol=1e-10, atol=1e-10, err_msg=names[id(plist[i])])
