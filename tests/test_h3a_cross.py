"""The hand-scheduled field kernel (nsff_field_kernel_h3a, tile_points 130) against the compiler-scheduled eight-wave kernel (131)
and the oracle on RANDOM architectures and launch shapes: depth 2..8, any skip set, 4..10 embedding frequencies, time codes of
16..64 columns, every static / dynamic mode combination the render path uses, point counts that are not multiples of the
128-point tile.  The two kernels evaluate the same f16x3 layers in different orders, with different sin / cos (Cody-Waite vs the
library) and exp (v_exp_f32 vs expf) implementations: records must agree to 2e-5 of the record's largest value (measured worst
2.2e-6 on these cases), five times below the 1e-4 parity bar; one configuration per architecture is also held to the oracle at 1e-4.  Each launch is
checked to have taken the kernel it was meant to take."""
import numpy as np
import pytest
import torch

import parity
import nsff_pl_amd as A
from nsff_pl_amd import _lib, config
from oracle import nsff_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

ARCHS = [  # D, skips, n_freqs, n_tau
    (8, [4], 10, 48), (8, [2, 5], 10, 48), (4, [], 10, 48), (2, [], 6, 16), (2, [1], 10, 64), (6, [1, 2, 3, 4, 5], 4, 32),
    (8, [7], 10, 48), (5, [3], 8, 20), (8, [1, 3, 5, 7], 10, 36), (3, [2], 10, 4),
]
MODES = [(2, 2, 2), (2, 2, 0), (0, 2, 1), (1, 1, 0), (2, 0, 0), (1, 0, 0)]      # (static_mode, transient_mode, flow heads)


def _query(m, P, S, sm, tm, fh, xyz, freqs, t_rows, tile, t_bias=None):
    config.set_precision("f16x3")
    config.set_tile_points(tile)
    raw = torch.empty(P, _lib.RAW_STRIDE, device=DEV)
    try:
        _lib.field_query(m, raw, P, S, sm, tm, fh, xyz=xyz, freqs=freqs, t_emb=t_rows if tm else None, t_bias=t_bias)
        torch.cuda.synchronize()
        return raw, _lib.last_field_kernel()
    finally:
        config.set_tile_points(0)


@pytest.mark.parametrize("arch", range(len(ARCHS)))
def test_hand_scheduled_kernel_equals_eight_wave_kernel(arch, hip_lib):
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(100 + arch)
    emb = A.PosEmbedding(n_freqs - 1, n_freqs)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight"):
                p.mul_(2.5)
    m.to(DEV)
    freqs = [float(f) for f in emb.freqs]
    g = torch.Generator().manual_seed(arch)
    rng = np.random.RandomState(arch)
    checked_oracle = False
    for k, (sm, tm, fh) in enumerate(MODES):
        S = int(rng.choice([37, 64, 192]))
        n_rays = int(rng.randint(3, 40))
        P = S * n_rays                                      # rarely a multiple of 128: the last tile is partial
        xyz = (torch.rand(P, 3, generator=g) * 2.4 - 1.2).to(DEV)
        t_rows = torch.randn(n_rays, n_tau, generator=g).to(DEV)
        got, kern = _query(m, P, S, sm, tm, fh, xyz, freqs, t_rows, 130)
        ref, kern_ref = _query(m, P, S, sm, tm, fh, xyz, freqs, t_rows, 131)
        assert kern == "h3a" and kern_ref == "h3_8wave", (kern, kern_ref)
        a, b = got.cpu().numpy(), ref.cpu().numpy()
        assert np.isfinite(a).all()
        for lo, hi, what in ((0, 4, "static"), (4, 8, "dynamic"), (8, 14, "flows")):
            scale = max(np.abs(b[:, lo:hi]).max(), 1e-30)
            err = np.abs(a[:, lo:hi] - b[:, lo:hi]).max() / scale
            assert err <= 2e-5, f"arch {ARCHS[arch]} mode {(sm, tm, fh)} P={P}: {what} columns differ by {err:.2e}"
        if tm:
            # the same launch with the time code folded into per-ray bias rows: taken when no 64-point half straddles two rays,
            # silently the plain launch otherwise; the folded product is exact fp32 instead of f16x3 -- same 2e-5 contract
            (tb,) = _lib.time_bias([(m, t_rows)])
            w0 = getattr(m, "transient_xyz_encoding_1")[0]
            want_tb0 = (t_rows.double() @ w0.weight.double()[:, 3 + 6 * n_freqs:].T + w0.bias.double()).float()
            assert tb.shape == (n_rays, 1 + len(skips), 256)
            assert (tb[:, 0] - want_tb0).abs().max().item() <= 2e-6 * want_tb0.abs().max().item()
            got_tb, kern_tb = _query(m, P, S, sm, tm, fh, xyz, freqs, t_rows, 130, t_bias=tb)
            assert kern_tb == ("h3a_tb" if S % 64 == 0 else "h3a"), (kern_tb, S)
            c = got_tb.cpu().numpy()
            for lo, hi, what in ((0, 4, "static"), (4, 8, "dynamic"), (8, 14, "flows")):
                scale = max(np.abs(b[:, lo:hi]).max(), 1e-30)
                err = np.abs(c[:, lo:hi] - b[:, lo:hi]).max() / scale
                assert err <= 2e-5, f"arch {ARCHS[arch]} mode {(sm, tm, fh)} P={P} S={S}: {what} columns with t_bias differ by {err:.2e}"
            if S % 64:
                assert np.array_equal(c, a)
        if sm == 2 and tm == 2 and fh == 2 and not checked_oracle:
            f = orc.field_from_module(m)
            x_emb = np.concatenate([orc.pos_embedding(xyz.cpu().numpy(), np.asarray(freqs, np.float32)),
                                    np.repeat(t_rows.cpu().numpy(), S, 0)], 1)
            want = orc.nerf_forward(f["params"], dict(f["cfg"], in_dir=0, in_a=0), x_emb, output_transient=True,
                                    output_transient_flow=("fw", "bw"))
            parity.assert_close(f"arch {ARCHS[arch]} vs oracle", np.concatenate([a[:, 0:4], a[:, 4:8], a[:, 8:14]], 1), want, parity.RTOL)
            checked_oracle = True
    assert checked_oracle


def test_trunks_the_body_does_not_cover_take_the_eight_wave_kernel(hip_lib):
    """view-direction branch in the launch, a 12-frequency embedding (128 padded columns): fallback, by name, same results contract"""
    torch.manual_seed(3)
    emb = A.PosEmbedding(11, 12)
    m = A.NeRF("fine", in_channels_xyz=3 + 6 * 12, use_viewdir=False, encode_transient=True, output_flow=True).to(DEV)
    g = torch.Generator().manual_seed(1)
    P, S = 64 * 9, 64
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
    t_rows = torch.randn(9, 48, generator=g).to(DEV)
    _, kern = _query(m, P, S, 2, 2, 2, xyz, [float(f) for f in emb.freqs], t_rows, 130)
    assert kern == "h3_8wave"


def test_view_direction_model_runs_as_two_launches(hip_lib):
    """A launch with a view-direction static trunk (not covered by the hand-scheduled body) next to a dynamic trunk: the static
    workgroups take the eight-wave kernel, the dynamic ones the hand-scheduled kernel, each writing its part of the records --
    the records equal those of the all-eight-wave launch (static part bit for bit, dynamic part to 2e-5)."""
    torch.manual_seed(11)
    emb, emb_d = A.PosEmbedding(9, 10), A.PosEmbedding(3, 4)
    m = A.NeRF("fine", use_viewdir=True, encode_appearance=True, in_channels_a=48, encode_transient=True, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight"):
                p.mul_(2.5)
    m.to(DEV)
    g = torch.Generator().manual_seed(2)
    S, n_rays = 128, 11
    P = S * n_rays
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
    t_rows = torch.randn(n_rays, 48, generator=g).to(DEV)
    a_rows = torch.randn(n_rays, 48, generator=g).to(DEV)
    dirs = emb_d(torch.randn(n_rays, 3, generator=g).to(DEV)).contiguous()
    freqs = [float(f) for f in emb.freqs]
    out = {}
    for tile in (130, 131):
        config.set_precision("f16x3"); config.set_tile_points(tile)
        raw = torch.empty(P, _lib.RAW_STRIDE, device=DEV)
        try:
            _lib.field_query(m, raw, P, S, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, dir_emb=dirs, a_emb=a_rows)
            torch.cuda.synchronize()
            out[tile] = (raw.cpu().numpy(), _lib.last_field_kernel())
        finally:
            config.set_tile_points(0)
    (a, ka), (b, kb) = out[130], out[131]
    assert ka == "h3a" and kb == "h3_8wave"
    assert np.array_equal(a[:, 0:4], b[:, 0:4])                                   # static part: the same kernel in both
    for lo, hi in ((4, 8), (8, 14)):
        assert np.abs(a[:, lo:hi] - b[:, lo:hi]).max() <= 2e-5 * np.abs(b[:, lo:hi]).max()
    assert np.abs(b[:, 4:14]).max() > 0


@pytest.mark.parametrize("in_a", [0, 48])
def test_view_direction_static_trunk_on_the_hand_scheduled_kernel(in_a, hip_lib):
    """Given the per-ray [dir | a] rows (nsff_side_bias), the static trunk of a view-direction model runs on the hand-scheduled
    kernel: static_dir_encoding as one more 256-wide layer with a bias row per ray, sigma accumulated by the epilogues of the last
    trunk layer (fp32 FMA on the ReLU outputs).  Records equal the eight-wave launch's (which multiplies the [dir | a] columns
    and evaluates sigma as an f16x3 head) to 2e-5 of the record's largest value; split launches (static + dynamic trunk), static-only
    launches (whole records, zeros where the ride parked its partial sums) and ragged point counts."""
    torch.manual_seed(21 + in_a)
    emb, emb_d = A.PosEmbedding(9, 10), A.PosEmbedding(3, 4)
    m = A.NeRF("fine", use_viewdir=True, encode_appearance=in_a > 0, in_channels_a=max(in_a, 1), encode_transient=True, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight"):
                p.mul_(2.5)
    m.to(DEV)
    g = torch.Generator().manual_seed(2)
    freqs = [float(f) for f in emb.freqs]
    config.set_precision("f16x3")
    for S, n_rays, sm, tm, fh in ((128, 11, 2, 2, 2), (64, 9, 2, 0, 0), (192, 3, 2, 2, 0), (64, 7, 2, 2, 2)):
        P = S * n_rays                                          # (11 x 128, 9 x 64, ...: the last 128-point tile is partial)
        xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
        t_rows = torch.randn(n_rays, 48, generator=g).to(DEV)
        a_rows = torch.randn(n_rays, in_a, generator=g).to(DEV) if in_a else None
        dirs = emb_d(torch.randn(n_rays, 3, generator=g).to(DEV)).contiguous()
        sb = _lib.side_bias(m, dirs, a_rows)
        w = m.static_dir_encoding[0].weight.double()
        fold_b = w[:, :256] @ m.static_xyz_encoding_final.bias.double() + m.static_dir_encoding[0].bias.double()
        side = dirs.double() if a_rows is None else torch.cat([dirs.double(), a_rows.double()], 1)
        want_sb = (side @ w[:, 256:].T + fold_b).float()
        assert sb.shape == (n_rays, 1, 256)
        assert (sb[:, 0] - want_sb).abs().max().item() <= 3e-6 * want_sb.abs().max().item()
        out = {}
        for tile, s_bias in ((130, sb), (131, None)):
            config.set_tile_points(tile)
            raw = torch.full((P, _lib.RAW_STRIDE), float("nan"), device=DEV)
            try:
                _lib.field_query(m, raw, P, S, sm, tm, fh, xyz=xyz, freqs=freqs, t_emb=t_rows if tm else None, dir_emb=dirs,
                                 a_emb=a_rows, s_bias=s_bias)
                torch.cuda.synchronize()
                out[tile] = (raw.cpu().numpy(), _lib.last_field_kernel())
            finally:
                config.set_tile_points(0)
        (a, ka), (b, kb) = out[130], out[131]
        assert ka == "h3a_side" and kb == "h3_8wave", (ka, kb)
        if tm == 0:
            assert np.isfinite(a).all() and not a[:, 4:].any(), "a static-only launch stores whole records: zeros behind the static slots"
        for lo, hi, what in ((0, 3, "static rgb"), (3, 4, "static sigma"), (4, 8, "dynamic"), (8, 14, "flows")):
            if hi > 4 and tm == 0:
                continue
            scale = max(np.abs(b[:, lo:hi]).max(), 1e-30)
            err = np.abs(a[:, lo:hi] - b[:, lo:hi]).max() / scale
            assert err <= 2e-5, f"S={S} rays={n_rays} modes {(sm, tm, fh)} in_a={in_a}: {what} differ by {err:.2e}"
    # samples per ray not a multiple of 64: a 64-point half may straddle two rays -- the rows are ignored, the old two-launch form runs
    S, n_rays = 37, 20
    P = S * n_rays
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
    t_rows = torch.randn(n_rays, 48, generator=g).to(DEV)
    a_rows = torch.randn(n_rays, in_a, generator=g).to(DEV) if in_a else None
    dirs = emb_d(torch.randn(n_rays, 3, generator=g).to(DEV)).contiguous()
    config.set_tile_points(130)
    try:
        raw = torch.empty(P, _lib.RAW_STRIDE, device=DEV)
        _lib.field_query(m, raw, P, S, 2, 2, 2, xyz=xyz, freqs=freqs, t_emb=t_rows, dir_emb=dirs, a_emb=a_rows,
                         s_bias=_lib.side_bias(m, dirs, a_rows))
        torch.cuda.synchronize()
        assert _lib.last_field_kernel() == "h3a"
    finally:
        config.set_tile_points(0)


@pytest.mark.parametrize("arch", [0, 1, 2, 4])
def test_training_forward_on_the_hand_scheduled_kernel(arch, hip_lib):
    """nsff_field_kernel_h3a_save (the SAVE build of the body: activation copies and ReLU sign words riding in the phases) against
    the eight-wave training forward: the same records (2e-5), the same encoded input tile (one range-reduced sin / cos per column
    in both: equal up to fp16 rounding flips), every layer's saved activation equal up to one fp16 rounding on a fraction of a per cent of the values (the
    two kernels add the same products in different orders), sign words that differ only where the activation is a rounding
    error away from zero -- and nothing written outside the evaluated trunks' slots."""
    from nsff_pl_amd import field_grad
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(300 + arch)
    emb = A.PosEmbedding(n_freqs - 1, n_freqs)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight"):
                p.mul_(2.5)
    m.to(DEV)
    freqs = [float(f) for f in emb.freqs]
    g = torch.Generator().manual_seed(arch)
    config.set_precision("f16x3")
    for S, n_rays, sm, tm in ((64, 10, 2, 2), (192, 4, 0, 2), (128, 3, 2, 0)):
        P = S * n_rays                                      # an even number of 64-point tiles, not always a multiple of 128 points...
        xyz = (torch.rand(P, 3, generator=g) * 2.4 - 1.2).to(DEV)
        t_rows = torch.randn(n_rays, n_tau, generator=g).to(DEV)
        out = {}
        for tile in (130, 131):
            config.set_tile_points(tile)
            raw = torch.full((P, _lib.RAW_STRIDE), float("nan"), device=DEV)
            acts, xin, masks, _ = field_grad.alloc_saves(m, P, DEV, bool(tm), bool(sm))
            acts.view(torch.int16).fill_(0x7e01); masks.fill_(-2)             # (fill patterns: what the launch does not write keeps them)
            try:
                _lib.field_query(m, raw, P, S, sm, tm, 2 if tm else 0, xyz=xyz, freqs=freqs, t_emb=t_rows if tm else None,
                                 save_acts=acts, save_xin=xin, save_masks=masks)
                torch.cuda.synchronize()
                out[tile] = (raw.cpu().numpy(), acts.cpu(), xin.cpu(), masks.cpu(), _lib.last_field_kernel())
            finally:
                config.set_tile_points(0)
        (a, acts_a, xin_a, masks_a, ka), (b, acts_b, xin_b, masks_b, kb) = out[130], out[131]
        assert ka == "h3a_save" and kb == "h3_save", (ka, kb)
        for lo, hi in ((0, 4), (4, 8), (8, 14)):
            scale = max(np.abs(b[:, lo:hi]).max(), 1e-30)
            assert np.abs(a[:, lo:hi] - b[:, lo:hi]).max() <= 2e-5 * scale, (ARCHS[arch], S, sm, tm, lo)
        # the encoded input tile: one range-reduced sin / cos per column in both kernels (Cody-Waite here, the library call there):
        # equal after the rounding to fp16 except where an fp32 last-place difference straddles a rounding boundary
        dx = (xin_a.float() - xin_b.float()).abs()
        assert float(dx.max()) <= 2.0 ** -10 and float((dx > 0).float().mean()) < 0.01, (float(dx.max()), float((dx > 0).float().mean()))
        written = [t * (D + 1) + l for t, on in ((0, sm), (1, tm)) if on for l in range(D)]
        for slot in range(acts_a.shape[0]):
            xa, xb = acts_a[slot].float(), acts_b[slot].float()
            if slot not in written:
                assert torch.equal(acts_a[slot].view(torch.int16), acts_b[slot].view(torch.int16)) and (masks_a[slot] == -2).all()
                assert (acts_a[slot].view(torch.int16) == 0x7e01).all()
                continue
            dif = (xa - xb).abs()
            assert bool((dif <= xb.abs() * 2.0 ** -9 + 4e-6 * float(xb.abs().max())).all()), (ARCHS[arch], S, slot, float(dif.max()))
            assert float((dif > 0).float().mean()) < 0.01
            # sign words: [tile][thread] x 64 bits; a differing bit belongs to an activation that is ~0 in one of the kernels
            diff_bits = (masks_a[slot] ^ masks_b[slot])
            n_diff = sum(bin(int(v) & 0xFFFFFFFFFFFFFFFF).count("1") for v in diff_bits.reshape(-1)[diff_bits.reshape(-1) != 0].tolist())
            assert n_diff <= 2e-4 * masks_a[slot].numel() * 64, (ARCHS[arch], S, slot, n_diff)


@pytest.mark.parametrize("in_a", [0, 48])
def test_view_direction_training_forward_on_the_hand_scheduled_kernel(in_a, hip_lib):
    """Round 6: the TRAINING forward of a view-direction model -- the reference's documented training configuration (README.md:226-233:
    --use_viewdir --N_samples 128 --N_importance 0) -- on the SAVE build of the hand-scheduled body: given the per-ray [dir | a] rows
    (nsff_side_bias), nsff_field_kernel_h3a_save runs the static trunk with static_dir_encoding as a folded segment and sigma as a ride
    of the last trunk layer's epilogues, and saves what the backward kernels read: every trunk layer's activation and sign words,
    static_dir_encoding's in the static trunk's slot D, the [dir | a] input tile (a small launch of its own).  Against the eight-wave
    training forward, which multiplies the [dir | a] columns: records 2e-5, saves equal up to fp16 rounding on < 1 % of the values."""
    from nsff_pl_amd import field_grad
    torch.manual_seed(77 + in_a)
    emb, emb_d = A.PosEmbedding(9, 10), A.PosEmbedding(3, 4)
    m = A.NeRF("fine", use_viewdir=True, encode_appearance=in_a > 0, in_channels_a=max(in_a, 1), encode_transient=True, output_flow=True)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".weight"):
                p.mul_(2.5)
    m.to(DEV)
    D = m.D
    freqs = [float(f) for f in emb.freqs]
    g = torch.Generator().manual_seed(5 + in_a)
    config.set_precision("f16x3")
    for S, n_rays, sm, tm in ((128, 12, 2, 2), (64, 6, 2, 0)):
        P = S * n_rays
        xyz = (torch.rand(P, 3, generator=g) * 2.4 - 1.2).to(DEV)
        t_rows = torch.randn(n_rays, 48, generator=g).to(DEV)
        dirs = emb_d(torch.randn(n_rays, 3, generator=g).to(DEV)).contiguous()
        a_rows = torch.randn(n_rays, in_a, generator=g).to(DEV) if in_a else None
        out = {}
        for tile, rows in ((130, True), (131, False)):
            config.set_tile_points(tile)
            raw = torch.full((P, _lib.RAW_STRIDE), float("nan"), device=DEV)
            acts, xin, masks, side = field_grad.alloc_saves(m, P, DEV, bool(tm), True)
            acts.view(torch.int16).fill_(0x7e01); masks.fill_(-2)             # (fill patterns: what the launch does not write keeps them)
            try:
                _lib.field_query(m, raw, P, S, sm, tm, 2 if tm else 0, xyz=xyz, freqs=freqs, t_emb=t_rows if tm else None, dir_emb=dirs, a_emb=a_rows,
                                 save_acts=acts, save_xin=xin, save_masks=masks, save_side=side,
                                 s_bias=_lib.side_bias(m, dirs, a_rows) if rows else None)
                torch.cuda.synchronize()
                out[tile] = (raw.cpu().numpy(), acts.cpu(), masks.cpu(), side.cpu(), _lib.last_field_kernel())
            finally:
                config.set_tile_points(0)
        (a, acts_a, masks_a, side_a, ka), (b, acts_b, masks_b, side_b, kb) = out[130], out[131]
        assert ka == "h3a_save" and kb == "h3_save", (ka, kb)
        for lo, hi in ((0, 4),) + (((4, 8), (8, 14)) if tm else ()):
            scale = max(np.abs(b[:, lo:hi]).max(), 1e-30)
            assert np.abs(a[:, lo:hi] - b[:, lo:hi]).max() <= 2e-5 * scale, (in_a, S, sm, tm, lo, np.abs(a[:, lo:hi] - b[:, lo:hi]).max() / scale)
        # the [dir | a] tile: the eight-wave kernel rounds hi + lo halfs, the side-tile launch the fp32 row -- one fp16 rounding apart at most
        n_side = m.in_channels_dir + (in_a if in_a else 0)
        ds = (side_a.float() - side_b.float()).abs()
        assert float(ds.max()) <= 2.0 ** -9 * float(side_b.float().abs().max()) and float((ds > 0).float().mean()) < 0.01
        assert float(side_a.float().abs().max()) > 0.5
        written = [l for l in range(D + 1)] + ([D + 1 + l for l in range(D)] if tm else [])       # (static: the trunk's slots AND slot D, static_dir_encoding)
        for slot in range(acts_a.shape[0]):
            xa, xb = acts_a[slot].float(), acts_b[slot].float()
            if slot not in written:
                assert (acts_a[slot].view(torch.int16) == 0x7e01).all() and (masks_a[slot] == -2).all(), slot
                assert (acts_b[slot].view(torch.int16) == 0x7e01).all(), slot
                continue
            dif = (xa - xb).abs()
            assert bool((dif <= xb.abs() * 2.0 ** -9 + 4e-6 * float(xb.abs().max())).all()), (in_a, S, slot, float(dif.max()))
            assert float((dif > 0).float().mean()) < 0.01
            diff_bits = (masks_a[slot] ^ masks_b[slot])
            n_diff = sum(bin(int(v) & 0xFFFFFFFFFFFFFFFF).count("1") for v in diff_bits.reshape(-1)[diff_bits.reshape(-1) != 0].tolist())
            assert n_diff <= 2e-4 * masks_a[slot].numel() * 64, (in_a, S, slot, n_diff)


def test_persistent_launch_is_bit_identical_to_one_workgroup_per_tile(hip_lib, monkeypatch):
    """Large launches of the hand-scheduled kernel are persistent (include/nsff_render.h::nsff_last_field_grid): one workgroup per
    compute unit walks tiles of ONE trunk -- both trunks split by XCD; when they do not cost the same (time code through the matrix
    pipe; a view-direction static trunk with its per-ray rows) the longer trunk's last tiles are a SECOND ROUND of workgroups
    (grid = 2 x compute units) that the shorter trunk's XCDs run behind their own tiles -- one trunk in static-only /
    dynamic-only launches and for the dynamic half of a view-direction model without rows.
    Every tile is computed by the same instruction stream from the same inputs whichever workgroup runs it: the records are
    bit-identical to the one-workgroup-per-tile form (NSFF_NO_PERSIST=1), including a ragged last tile; the grid tells which form ran."""
    torch.manual_seed(5)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    emb, emb_d = A.PosEmbedding(9, 10), A.PosEmbedding(3, 4)
    freqs = [float(f) for f in emb.freqs]
    g = torch.Generator().manual_seed(9)
    plain = A.NeRF("fine", use_viewdir=False, encode_transient=True, in_channels_t=48, output_flow=True).to(DEV)
    viewdir = A.NeRF("fine", use_viewdir=True, encode_transient=True, in_channels_t=48, output_flow=True).to(DEV)
    config.set_precision("f16x3")
    S = 64
    cases = [  # model, rays, (static, transient, flow heads), per-ray time rows?, per-ray [dir | a] rows?, grid of the persistent form (None: never persistent)
        (plain, 2 * n_cus + 3, (2, 2, 2), True, False, n_cus),     # both trunks, equal cost: trunk by XCD
        (plain, 2 * n_cus + 3, (2, 2, 2), False, False, 2 * n_cus),  # time code through the matrix pipe: the dynamic trunk is 7 % longer
                                                                     # -> trunk by XCD + a second round of workgroups for its tail
        (plain, 2 * n_cus + 5, (2, 0, 0), False, False, n_cus),    # static only
        (plain, 2 * n_cus + 5, (0, 2, 1), True, False, n_cus),     # dynamic only
        (viewdir, 2 * n_cus + 1, (2, 2, 2), True, False, n_cus),   # static trunk on the eight-wave kernel, dynamic one persistent
        (viewdir, 2 * n_cus + 7, (2, 2, 2), True, True, 2 * n_cus),  # view-direction static trunk (23 % longer) beside the dynamic one:
                                                                     # its tail tiles are run by the dynamic trunk's XCDs (second round)
        (viewdir, 9 * n_cus + 5, (2, 2, 2), True, True, 2 * n_cus),  # ... several tiles per second-round workgroup, ragged
        (viewdir, 2 * n_cus + 7, (2, 0, 0), False, True, n_cus),   # ... alone
        (plain, n_cus // 8, (2, 2, 2), True, False, None),         # fewer tiles than workgroups
    ]
    for m, n_rays, (sm, tm, fh), rows, side, grid_p in cases:
        P = S * n_rays
        tiles = (P + 127) // 128
        xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
        t_rows = torch.randn(n_rays, 48, generator=g).to(DEV)
        dirs = emb_d(torch.randn(n_rays, 3, generator=g).to(DEV)).contiguous() if m is viewdir else None
        tb = _lib.time_bias([(m, t_rows)])[0] if (rows and tm) else None
        sb = _lib.side_bias(m, dirs, None) if side else None
        got = {}
        config.set_tile_points(130)
        try:
            for form in ("persistent", "tile"):
                if form == "tile":
                    monkeypatch.setenv("NSFF_NO_PERSIST", "1")
                else:
                    monkeypatch.delenv("NSFF_NO_PERSIST", raising=False)
                raw = torch.full((P, _lib.RAW_STRIDE), float("nan"), device=DEV)
                _lib.field_query(m, raw, P, S, sm, tm, fh, xyz=xyz, freqs=freqs, t_emb=t_rows if tm else None, dir_emb=dirs, t_bias=tb,
                                 s_bias=sb)
                torch.cuda.synchronize()
                got[form] = (raw.cpu().numpy(), _lib.last_field_grid(), _lib.last_field_kernel())
        finally:
            config.set_tile_points(0)
            monkeypatch.delenv("NSFF_NO_PERSIST", raising=False)
        # the per-call switch (NsffFieldArgs::launch_form through the scoped config.launch_form: what a sharded frame loop selects)
        config.set_tile_points(130)
        try:
            with config.launch_form(persistent=False):
                raw = torch.full((P, _lib.RAW_STRIDE), float("nan"), device=DEV)
                _lib.field_query(m, raw, P, S, sm, tm, fh, xyz=xyz, freqs=freqs, t_emb=t_rows if tm else None, dir_emb=dirs, t_bias=tb,
                                 s_bias=sb)
                torch.cuda.synchronize()
                assert _lib.last_field_grid() == got["tile"][1] and np.array_equal(raw.cpu().numpy().view(np.uint32), got["tile"][0].view(np.uint32))
            assert config.get_persistent() is True
        finally:
            config.set_tile_points(0)
        (a, ga, ka), (b, gb, kb) = got["persistent"], got["tile"]
        both = sm and tm and (m is not viewdir or side)
        assert ka == kb and ka.startswith("h3a"), (ka, kb)
        assert gb == (2 * tiles if both else tiles), (gb, tiles)
        assert ga == (grid_p if grid_p is not None else gb), (ga, grid_p, gb)
        used = slice(0, 4) if tm == 0 else (slice(4, 4 + 4 + 3 * fh) if sm == 0 else slice(0, 4 + 4 + 3 * fh))
        assert np.isfinite(a[:, used]).all()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"rays={n_rays} modes {(sm, tm, fh)} rows={rows}"


@pytest.mark.parametrize("arch", range(len(ARCHS)))
def test_persistent_launch_on_random_architectures(arch, hip_lib, monkeypatch):
    """the persistent form on every architecture of this file (depth 2..8, any skip set, 4..10 frequencies, 16..64 time-code
    columns): records bit-identical to one workgroup per tile for a both-trunk launch (time code through the matrix pipe and as
    per-ray rows) and a dynamic-only one; a trunk that ends with a skip layer's input part has no B16L phase to turn into B16LP
    and keeps one workgroup per tile (the grid says which form ran)"""
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(300 + arch)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    emb = A.PosEmbedding(n_freqs - 1, n_freqs)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True).to(DEV)
    freqs = [float(f) for f in emb.freqs]
    g = torch.Generator().manual_seed(50 + arch)
    S, n_rays = 64, 2 * n_cus + 3
    P = S * n_rays
    tiles = (P + 127) // 128
    xyz = (torch.rand(P, 3, generator=g) * 2.4 - 1.2).to(DEV)
    t_rows = torch.randn(n_rays, n_tau, generator=g).to(DEV)
    rows_ok = n_tau % 4 == 0 and n_tau <= 64
    can = (D - 1) not in skips                     # (the trunk ends with a 256-wide segment)
    config.set_precision("f16x3")
    for (sm, tm, fh), rows in (((2, 2, 2), False), ((2, 2, 2), True), ((0, 2, 1), True)):
        if rows and not rows_ok:
            continue
        tb = _lib.time_bias([(m, t_rows)])[0] if rows else None
        got = {}
        config.set_tile_points(130)
        try:
            for form in ("persistent", "tile"):
                if form == "tile":
                    monkeypatch.setenv("NSFF_NO_PERSIST", "1")
                else:
                    monkeypatch.delenv("NSFF_NO_PERSIST", raising=False)
                raw = torch.full((P, _lib.RAW_STRIDE), float("nan"), device=DEV)
                _lib.field_query(m, raw, P, S, sm, tm, fh, xyz=xyz, freqs=freqs, t_emb=t_rows, t_bias=tb)
                torch.cuda.synchronize()
                got[form] = (raw.cpu().numpy(), _lib.last_field_grid(), _lib.last_field_kernel())
        finally:
            config.set_tile_points(0)
            monkeypatch.delenv("NSFF_NO_PERSIST", raising=False)
        (a, ga, ka), (b, gb, kb) = got["persistent"], got["tile"]
        assert ka == kb and ka.startswith("h3a"), (ka, kb)
        assert gb == (2 * tiles if sm else tiles)
        # (both trunks of unequal cost -- the time code through the matrix pipe -- run a second round of workgroups: 2 x compute units)
        assert (ga in ((n_cus, 2 * n_cus) if sm else (n_cus,))) if can else ga == gb, (ARCHS[arch], (sm, tm, fh), rows, ga, gb)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (ARCHS[arch], (sm, tm, fh), rows)
