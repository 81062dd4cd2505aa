"""nsff_field_bwd_kernel_h3b -- the hand-scheduled data-gradient kernel (tools/h3asm/gen_bwd.py; 128-point workgroups, one wave per
SIMD, resident transposed weights, epilogues / fragment copies / refills riding in the other half's MFMA gaps) -- against
nsff_field_bwd_kernel, the compiler-scheduled kernel it replaces for the launches it covers (NSFF_BWD_KERNEL=c): the same step
programs, the same arithmetic in the same order, so everything the launch leaves in HBM -- the pre-activation-gradient fragments of
every layer, the head gradients, d(trunk input) -- must be BIT-IDENTICAL.  Random architectures (depth 2..8, any skip set), both
trunks / one trunk, with and without the trunk-input gradient, ragged point counts; what the body does not cover must fall back
(and say so through nsff_last_bwd_kernel).  The gradients' parity with the reference is tests/test_gradients.py /
test_field_grad.py, which run this kernel by default."""
import numpy as np
import pytest
import torch

import nsff_pl_amd as A
from nsff_pl_amd import _lib, field_grad

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

ARCHS = [  # D, skips, n_freqs, n_tau
    (8, [4], 10, 48), (8, [2, 5], 10, 48), (4, [], 10, 48), (2, [], 6, 16), (2, [1], 10, 64), (6, [1, 2, 3, 4, 5], 4, 32),
    (8, [7], 10, 48), (5, [3], 8, 20), (8, [1, 3, 5, 7], 10, 36), (3, [2], 10, 4),
]


def _run(m, P, static, transient, want_xin, d_raw, raw, masks, force_c, monkeypatch):
    if force_c:
        monkeypatch.setenv("NSFF_BWD_KERNEL", "c")
    else:
        monkeypatch.delenv("NSFF_BWD_KERNEL", raising=False)
    tiles = (P + 63) // 64
    xin_rows, _, _ = _lib.train_dims(m)
    dpre = torch.full((field_grad.n_slots(m), tiles, 64 * 256), 7.0, device=DEV, dtype=torch.float16)
    dhead = torch.full((2, tiles, 64 * 32), 7.0, device=DEV, dtype=torch.float16)
    d_xin = torch.full((P, xin_rows), 7.0, device=DEV) if want_xin else None
    gmax = _lib.absmax(d_raw)
    _lib.field_backward(m, P, static, transient, d_raw, raw, gmax, masks, dpre, dhead, d_xin)
    torch.cuda.synchronize()
    return dpre.cpu().numpy().view(np.uint16), dhead.cpu().numpy().view(np.uint16), None if d_xin is None else d_xin.cpu().numpy().view(np.uint32), _lib.last_bwd_kernel()


@pytest.mark.parametrize("arch", range(len(ARCHS)))
def test_hand_scheduled_backward_is_bit_identical_to_the_compiler_scheduled_kernel(arch, hip_lib, monkeypatch):
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(400 + arch)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight"):
                p.mul_(2.0)
    m.to(DEV)
    g = torch.Generator().manual_seed(70 + arch)
    sk = sorted(set(skips))
    for k, (static, transient, want_xin, P) in enumerate([(True, True, True, 128 * 37), (False, True, True, 128 * 21 - 40), (True, False, False, 128 * 9),
                                                           (True, True, False, 128 * 5 - 1), (False, True, True, 64 * 7)]):
        tiles = (P + 63) // 64
        d_raw = (torch.randn(P, _lib.RAW_STRIDE, generator=g) * 10.0 ** (-4 * torch.rand(P, 1, generator=g))).to(DEV)
        raw = torch.rand(P, _lib.RAW_STRIDE, generator=g).to(DEV) * 0.2
        masks = torch.randint(-2 ** 62, 2 ** 62, (field_grad.n_slots(m), tiles, 256), generator=g, dtype=torch.int64).to(DEV)
        a = _run(m, P, static, transient, want_xin, d_raw, raw, masks, False, monkeypatch)
        b = _run(m, P, static, transient, want_xin, d_raw, raw, masks, True, monkeypatch)
        assert b[3] == "c"
        # covered: an even number of 64-point tiles, at most one skip layer when the trunk-input gradient is wanted, none at D - 1
        tail = transient and want_xin
        covered = tiles % 2 == 0 and not (tail and (len(sk) > 1 or (D - 1) in sk)) and not (D - 1 in sk and False)
        if tail and sk and sk[0] == D - 1:
            covered = False
        assert a[3] == ("h3b" if covered else "c"), (ARCHS[arch], k, a[3], covered)
        assert np.array_equal(a[1], b[1]), f"arch {ARCHS[arch]} case {k}: head gradients differ"
        assert np.array_equal(a[0], b[0]), (f"arch {ARCHS[arch]} case {k}: fragments differ in slots "
                                             f"{sorted(set(np.nonzero((a[0] != b[0]).reshape(a[0].shape[0], -1).any(1))[0].tolist()))}")
        if want_xin:
            assert np.array_equal(a[2], b[2]), f"arch {ARCHS[arch]} case {k}: d_xin differs"


def test_c2_sized_backward_launches_take_the_hand_scheduled_kernel(hip_lib, monkeypatch):
    """the launches of a C2 training step (65 536 / 196 608 points; both trunks, and the dynamic trunk alone with its trunk-input
    gradient): no silent fallback, and the PERSISTENT form (one workgroup per compute unit, items from a device counter, the next
    item's records brought into LDS behind the current body) -- same bits as one workgroup per item (NSFF_BWD_PERSIST=0), launch
    after launch (the counter is left at zero by the launch's last fetch); a ragged point count and a launch too small to persist
    beside them"""
    monkeypatch.delenv("NSFF_BWD_KERNEL", raising=False)
    torch.manual_seed(1)
    m = A.NeRF("fine", use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True).to(DEV)
    g = torch.Generator().manual_seed(2)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    for P, static in ((65536, True), (196608, True), (196608, False), (128 * 1531 - 13, True), (128 * (2 * n_cus - 1), False)):
        tiles = (P + 63) // 64
        items = tiles // 2 * (2 if static else 1)
        d_raw = torch.randn(P, _lib.RAW_STRIDE, generator=g).to(DEV)
        raw = (torch.rand(P, _lib.RAW_STRIDE, generator=g) * 0.2).to(DEV)
        masks = torch.randint(-2 ** 62, 2 ** 62, (field_grad.n_slots(m), tiles, 256), generator=g, dtype=torch.int64).to(DEV)
        monkeypatch.delenv("NSFF_BWD_PERSIST", raising=False)
        out = _run(m, P, static, True, True, d_raw, raw, masks, False, monkeypatch)
        assert out[3] == "h3b"
        assert _lib.last_bwd_grid() == (n_cus if items >= 2 * n_cus else items), (P, static, _lib.last_bwd_grid())
        assert np.isfinite(out[2].view(np.float32)).all()
        again = _run(m, P, static, True, True, d_raw, raw, masks, False, monkeypatch)
        monkeypatch.setenv("NSFF_BWD_PERSIST", "0")
        one = _run(m, P, static, True, True, d_raw, raw, masks, False, monkeypatch)
        assert one[3] == "h3b" and _lib.last_bwd_grid() == items
        for k in range(3):
            assert np.array_equal(out[k], one[k]) and np.array_equal(out[k], again[k]), (P, static, k)
    monkeypatch.delenv("NSFF_BWD_PERSIST", raising=False)


def test_overflowing_gradients_clamp_like_the_compiler_scheduled_kernel(hip_lib, monkeypatch):
    """Pre-activation gradients that leave the fp16 range: nsff_field_bwd_kernel clamps with v_med3_f32 in front of the conversion, the
    hand-scheduled body runs with MODE.FP16_OVFL set (the conversion itself saturates at +-65504, no instruction).  Weights x 40 drive
    thousands of values per tile out of range; both kernels must leave the same bits -- no infinity, no NaN."""
    torch.manual_seed(9)
    m = A.NeRF("fine", D=4, skips=[2], use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight") and "encoding" in name:
                p.mul_(40.0)
    m.to(DEV)
    g = torch.Generator().manual_seed(10)
    P = 128 * 11
    tiles = P // 64
    d_raw = torch.randn(P, _lib.RAW_STRIDE, generator=g).to(DEV)
    raw = (torch.rand(P, _lib.RAW_STRIDE, generator=g) * 0.2).to(DEV)
    masks = torch.full((field_grad.n_slots(m), tiles, 256), -1, dtype=torch.int64).to(DEV)       # (every ReLU open: nothing masks the growth)
    a = _run(m, P, True, True, True, d_raw, raw, masks, False, monkeypatch)
    b = _run(m, P, True, True, True, d_raw, raw, masks, True, monkeypatch)
    assert a[3] == "h3b" and b[3] == "c"
    written = [s_ for s_ in range(field_grad.n_slots(m)) if not (b[0][s_] == np.float16(7.0).view(np.uint16)).all()]
    bits = b[0][written]
    vals = bits.view(np.float16).astype(np.float32)
    # (what reaches HBM is the tile times the point's factor G / s_p = 2^-k: a saturated value shows as 65504 * 2^-k -- an fp16 whose
    #  ten mantissa bits are all ones; among random values one in 1024 looks like that)
    assert np.isfinite(vals).all() and np.isfinite(b[2].view(np.float32)).all()
    big = np.abs(vals) >= 1024
    assert ((bits[big] & 0x3FF) == 0x3FF).mean() > 0.02, ((bits[big] & 0x3FF) == 0x3FF).mean()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_view_direction_model_runs_its_dynamic_trunk_on_the_hand_scheduled_kernel(hip_lib, monkeypatch):
    """The reference's README configuration (use_viewdir: static_dir_encoding between the static trunk and static_rgb, nerf.py:183-186):
    the body does not execute that static trunk, so a launch of both trunks runs as two -- the static trunk on nsff_field_bwd_kernel,
    the dynamic one on the hand-scheduled body -- and the re-query launches (dynamic trunk alone) on the body; same bits as the
    compiler-scheduled kernel for everything."""
    torch.manual_seed(21)
    m = A.NeRF("fine", use_viewdir=True, encode_appearance=False, encode_transient=True, in_channels_t=48, output_flow=True).to(DEV)
    g = torch.Generator().manual_seed(22)
    for static, want in ((True, "c+h3b"), (False, "h3b")):
        P = 128 * 19 - 7
        tiles = (P + 63) // 64
        d_raw = torch.randn(P, _lib.RAW_STRIDE, generator=g).to(DEV)
        raw = (torch.rand(P, _lib.RAW_STRIDE, generator=g) * 0.2).to(DEV)
        masks = torch.randint(-2 ** 62, 2 ** 62, (field_grad.n_slots(m), tiles, 256), generator=g, dtype=torch.int64).to(DEV)
        a = _run(m, P, static, True, True, d_raw, raw, masks, False, monkeypatch)
        b = _run(m, P, static, True, True, d_raw, raw, masks, True, monkeypatch)
        assert (a[3], b[3]) == (want, "c")
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
