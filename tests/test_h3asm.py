"""The hand-scheduled trunk body of nsff_field_kernel_h3a (tools/h3asm): the committed csrc/field_h3a_body.inc is what the generator
produces, the stream passes the wait-state lint, and the functional four-wave simulator (pending-load registers, barrier
rendezvous, cross-wave LDS race detection) reproduces a numpy evaluation of the same layers -- static trunk with a skip layer,
dynamic trunk with the time-code rebuild.  CPU only; the GPU parity suite runs the assembled kernel (fixture "f16x3-130")."""
import io
import os
import sys
import contextlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "h3asm"))
import gen      # noqa: E402
import check    # noqa: E402
import isa      # noqa: E402


def test_committed_body_is_the_generators_output(tmp_path, monkeypatch):
    committed = open(os.path.join(ROOT, "nsff_pl_amd", "csrc", "field_h3a_body.inc")).read()
    monkeypatch.setattr(sys, "argv", ["gen.py"])
    real_open = open
    captured = {}

    class Sink(io.StringIO):
        def close(self):
            captured["text"] = self.getvalue()
            super().close()

    def fake_open(path, mode="r", *a, **k):
        if "w" in mode and str(path).endswith(".inc"):
            return Sink()
        return real_open(path, mode, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    with contextlib.redirect_stdout(io.StringIO()):
        gen.main()
    assert captured["text"] == committed, "run `python tools/h3asm/gen.py` and commit nsff_pl_amd/csrc/field_h3a_body.inc"


def test_stream_passes_the_hazard_lint():
    pre, prog, bodies = gen.build()
    assert gen.lint(bodies, prog) == []
    assert sum(i.kind == "mfma" for i in bodies["A16R"]) == 192 and sum(i.kind == "mfma" for i in bodies["B8"]) == 96
    assert all(i.kind != "mfma" for i in pre)


@pytest.mark.parametrize("kind", ["static", "dynamic", "dynamic_tb", "twoskips_tb", "viewdir", "static_save", "dynamic_save", "twoskips_save", "static_persist",
                                  "dynamic_tb_persist", "viewdir_persist"])
def test_simulated_trunk_matches_numpy(kind):
    assert check.run_case(kind, verbose=False) < 2e-6


def test_simulator_notices_a_missing_wait(monkeypatch):
    """the test of the test: drop the waits for the lo fragments -> an MFMA reads a register with a load still in flight"""
    orig = gen.Stream.need_lds
    monkeypatch.setattr(gen.Stream, "need_lds", lambda self, tag: None if tag[0] == "xl" else orig(self, tag))
    with pytest.raises(isa.SimError, match="outstanding"):
        check.run_case("static", verbose=False)


# ---- the C++ host-side program builder (csrc/field_h3.hip: h3_step_program + h3a_build_program) against the Python builder the
# simulator runs (check.build_program), through the host-only C-ABI nsff_field_phase_program: no GPU involved
ARCHS = [(8, [4], 10, 48), (8, [2, 5], 10, 48), (4, [], 10, 48), (2, [], 6, 16), (2, [1], 10, 64), (6, [1, 2, 3, 4, 5], 4, 32),
         (8, [7], 10, 48), (8, [1, 3, 5, 7], 10, 36)]


@pytest.mark.parametrize("arch", range(len(ARCHS)))
def test_cxx_phase_program_equals_the_simulated_builder(arch):
    import torch
    import nsff_pl_amd as A
    from nsff_pl_amd import _lib
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(0)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True)
    for sm, tm in ((2, 2), (1, 1), (2, 0), (0, 2)):
        steps, n_static, ph_s, ph_d = _lib.h3a_program(m, sm, tm)
        assert (len(ph_s) > 0) == (sm > 0) and (len(ph_d) > 0) == (tm > 0)
        for lo, hi, got, in_t in ((0, n_static, ph_s, 0), (n_static, len(steps), ph_d, n_tau)):
            if hi == lo:
                continue
            segs, nb = [], 0
            for i, (w, b, nks, pre, post, head) in enumerate(steps[lo:hi]):
                segs.append(dict(nks=nks, off=4 * w, bias=None if b is None else nb, post="relu" if post == 1 else "none",
                                 rebuild=i > 0 and pre != 0))
                nb += b is not None
            # the heads of this trunk: rows and first record float (field_h3.hip::h3a_head_sel); the tile's offset is not compared
            head = dict(off=0, n_rows=(4 if sm == 2 else 1) if in_t == 0 else 10, slot0=(0 if sm == 2 else 3) if in_t == 0 else 4)   # (10: a model with flow heads)
            want = [list(map(int, r)) for r in check.build_program(segs, in_t, head)]
            assert len(want) == len(got)
            B = gen.BODY
            uses_streams = {B["B16R"], B["B16X"], B["B4"], B["B8"], B["A4F"], B["A8F"]}        # (+ descriptor 0: the pre-issue)
            for i, (w_, g_) in enumerate(zip(want, got)):
                # body, flags, bias row (HEAD: 4 x first record float)[, rows of the heads]; stream fields where they are read
                n_cmp = 8 if (i == 0 or w_[0] in uses_streams) else (4 if w_[0] in (B["EPI_B"], B["HEAD"]) else 3)
                assert w_[:n_cmp] == g_[:n_cmp], (ARCHS[arch], sm, tm, i, w_, g_)
        if tm == 0:
            continue
        # the same launch with the time code folded into per-ray bias rows (NsffFieldArgs::t_bias): input segments run their
        # position part only, the layer's first segment gets a second table row for half B
        steps, n_static, _, ph_f = _lib.h3a_program(m, sm, tm, fold_t=True)
        segs, nb = [], 0
        for i, (w, b, nks, pre, post, head) in enumerate(steps[n_static:]):
            segs.append(dict(nks=nks, off=4 * w, bias=None if b is None else nb, post="relu" if post == 1 else "none",
                             rebuild=i > 0 and pre != 0))
            nb += b is not None
        for t, sg in enumerate(segs):
            if sg["nks"] == 16:
                continue
            sg.update(wstride=sg["nks"] * 4096, nks=4)
            first = segs[0 if t == 0 else t - 1]
            first["bias_b"] = nb
            nb += 1
        want = [list(map(int, r)) for r in check.build_program(segs, 0, dict(off=0, n_rows=10, slot0=4))]
        assert len(want) == len(ph_f) and nb <= 16
        for i, (w_, g_) in enumerate(zip(want, ph_f)):
            n_cmp = 8 if (i == 0 or w_[0] in uses_streams) else (4 if w_[0] in (B["EPI_B"], B["HEAD"]) else 3)
            assert w_[:n_cmp] == g_[:n_cmp], (ARCHS[arch], sm, tm, "fold_t", i, w_, g_)


@pytest.mark.parametrize("arch", range(len(ARCHS)))
def test_cxx_persistent_program_equals_the_simulated_patch(arch):
    """the programs of a persistent launch (fold_t bit 2): the C++ patch (h3a_make_persistent) against check.make_persistent on
    the C++ builder's own plain program -- B16L -> B16LP with descriptor 0's stream fields, nothing else; a trunk that ends with a
    skip layer's input part has no such phase and is refused (one workgroup per tile)"""
    import numpy as np
    import torch
    import nsff_pl_amd as A
    from nsff_pl_amd import _lib
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(0)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True)
    for fold_t in (False, True):
        _, _, ph_s, ph_d = _lib.h3a_program(m, 2, 2, fold_t=fold_t)
        _, _, pp_s, pp_d = _lib.h3a_program(m, 2, 2, fold_t=fold_t, persist=True)
        for plain, pers in ((ph_s, pp_s), (ph_d, pp_d)):
            assert plain
            want = np.array(plain, np.uint32)
            if check.make_persistent(want):
                assert [list(map(int, r)) for r in want] == pers
                at = [i for i, r in enumerate(pers) if r[0] == gen.BODY["B16LP"]]
                assert len(at) == 1 and pers[at[0]][3:8] == pers[0][3:8] and not any(r[0] == gen.BODY["B16L"] for r in pers)
            else:
                assert pers == [] and (D - 1) in skips


@pytest.mark.parametrize("arch", [(8, [4], 48), (4, [2], 0), (3, [], 12)])
def test_cxx_side_fold_program_equals_the_simulated_builder(arch):
    """a view-direction static trunk given per-ray [dir | a] rows (NsffFieldArgs::s_bias): static_dir_encoding is one more
    256-wide segment with a row per half, the sigma head is the ride of the last trunk layer's epilogues (B16RS / A16RS)"""
    import torch
    import nsff_pl_amd as A
    from nsff_pl_amd import _lib
    D, skips, in_a = arch
    torch.manual_seed(0)
    m = A.NeRF("fine", D=D, skips=skips, use_viewdir=True, encode_appearance=in_a > 0, in_channels_a=max(in_a, 1), encode_transient=True,
               in_channels_t=48, output_flow=True)
    steps, n_static, ph_s, ph_d = _lib.h3a_program(m, 2, 2, fold_t=True, side_fold=True)
    assert len(ph_s) > 0 and len(ph_d) > 0
    assert [st[5] for st in steps[:n_static]].count(1) == 1 and steps[n_static - 1][5] == 2       # HEAD_S_SIGMA mid-trunk, HEAD_S_RGB last
    segs, nb = [], 0
    for i, (w, b, nks, pre, post, head) in enumerate(steps[:n_static]):
        segs.append(dict(nks=nks, off=4 * w, bias=None if b is None else nb, post="relu" if post == 1 else "none",
                         rebuild=i > 0 and pre != 0))
        nb += b is not None
    segs[-1]["bias_b"] = nb                       # half B's row of static_dir_encoding, then the sigma weights' row
    want = [list(map(int, r)) for r in check.build_program(segs, 0, dict(off=0, n_rows=3, slot0=0), sig_row=nb + 1)]
    B = gen.BODY
    assert [w_[0] for w_ in want].count(B["B16RS"]) == 1 and [w_[0] for w_ in want].count(B["A16RS"]) == 1
    assert len(want) == len(ph_s)
    uses_streams = {B["B16R"], B["B16RS"], B["B16X"], B["B4"], B["B8"], B["A4F"], B["A8F"]}
    for i, (w_, g_) in enumerate(zip(want, ph_s)):
        n_cmp = 8 if (i == 0 or w_[0] in uses_streams) else (4 if w_[0] in (B["EPI_B"], B["HEAD"]) else 3)
        assert w_[:n_cmp] == g_[:n_cmp], (arch, i, w_, g_)
    # without the rows the static trunk of a view-direction launch is not one the body executes
    assert _lib.h3a_program(m, 2, 2)[2] == []


def test_compiled_kernel_audit():
    """512 registers, no scratch, no spilled VGPRs, one body statement, and no compiler use of accumulation registers between the
    pre-issue statement and the body (the weight loads are in flight there)"""
    import shutil
    import subprocess
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "h3asm", "audit.py"), os.path.join(ROOT, "nsff_pl_amd", "csrc", "field_h3.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


# ---- the backward body (round 6): tools/h3asm/gen_bwd.py -> csrc/field_bwd_h3b_body.inc, nsff_field_bwd_kernel_h3b
import gen_bwd      # noqa: E402
import check_bwd    # noqa: E402


def test_committed_backward_body_is_the_generators_output(monkeypatch):
    committed = open(os.path.join(ROOT, "nsff_pl_amd", "csrc", "field_bwd_h3b_body.inc")).read()
    monkeypatch.setattr(sys, "argv", ["gen_bwd.py"])
    real_open = open
    captured = {}

    class Sink(io.StringIO):
        def close(self):
            captured["text"] = self.getvalue()
            super().close()

    def fake_open(path, mode="r", *a, **k):
        if "w" in mode and str(path).endswith(".inc"):
            return Sink()
        return real_open(path, mode, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    with contextlib.redirect_stdout(io.StringIO()):
        gen_bwd.main()
    assert captured["text"] == committed, "run `python tools/h3asm/gen_bwd.py` and commit nsff_pl_amd/csrc/field_bwd_h3b_body.inc"


def test_backward_stream_passes_the_hazard_lint():
    pre, prog, bodies = gen_bwd.build()
    assert gen_bwd.lint(bodies) == [] and all(i.kind != "mfma" for i in pre)
    for name in ("A16", "B16", "A16S", "B16S", "A16F", "B16L", "AX", "BX", "AXS", "BXS", "BXD"):
        assert sum(i.kind == "mfma" for i in bodies[name]) == 64, name
    assert sum(i.kind == "mfma" for i in bodies["AH"]) == 16 and sum(i.kind == "mfma" for i in bodies["BH"]) == 16


@pytest.mark.parametrize("kind", ["static", "dynamic", "noskip", "short", "ragged", "overflow"])
def test_simulated_backward_chain_is_bit_identical_to_numpy(kind):
    """every fragment slot of the chain, d(trunk input) under its exec masks and the last tile in LDS, bit for bit; nothing else
    written (the simulator refuses a global dword stored twice or outside the mapped buffers)"""
    assert check_bwd.run_case(kind, verbose=False)


def test_backward_simulator_notices_a_missing_wait(monkeypatch):
    """the test of the test: drop the waits for the sign words -> the epilogue reads a register with a load still in flight"""
    orig = gen_bwd.LogStream.need_vm
    monkeypatch.setattr(gen_bwd.LogStream, "need_vm", lambda self, tag: None if isinstance(tag, str) and tag.startswith("msk") else orig(self, tag))
    with pytest.raises(isa.SimError, match="outstanding"):
        check_bwd.run_case("static", verbose=False)


@pytest.mark.parametrize("arch", range(len(ARCHS)))
def test_cxx_backward_phase_program_equals_the_simulated_builder(arch):
    """csrc/field_bwd.hip::h3b_build_program (through the host-only C-ABI nsff_field_bwd_phase_program) against
    check_bwd.build_program, the builder the simulator runs; trunks the body does not cover are refused by both"""
    import torch
    import nsff_pl_amd as A
    from nsff_pl_amd import _lib
    D, skips, n_freqs, n_tau = ARCHS[arch]
    torch.manual_seed(0)
    m = A.NeRF("fine", D=D, skips=skips, in_channels_xyz=3 + 6 * n_freqs, use_viewdir=False, encode_transient=True,
               in_channels_t=n_tau, output_flow=True)
    n_tiles = 48
    sk = sorted(set(skips))
    for dynamic, want_xin in ((False, False), (True, True), (True, False)):
        got, offs = _lib.field_bwd_phase_program(m, dynamic, want_xin, n_tiles)
        tail = dynamic and want_xin
        stash_l = sk[0] if (tail and sk) else None
        # the step list as csrc/field_bwd.hip::bwd_step_program builds it: head, layers D-1 .. 1, (x0, xskip)
        steps = [dict(off=int(offs[0]), nks=4, stash=(stash_l == D - 1))]
        covered = True
        j = 1
        for l in range(D - 1, 0, -1):
            if tail and l in sk and l != stash_l:
                covered = False             # (a further skip layer: its trunk-input step sits inside the chain)
                j += 1
            steps.append(dict(off=int(offs[j]), nks=16, stash=(stash_l == l - 1)))
            j += 1
        if tail and (D - 1) in sk and (D - 1) != stash_l:
            covered = False
        if tail:
            steps.append(dict(off=int(offs[j]), nks=16, stash=False)); j += 1
            if stash_l is not None:
                steps.append(dict(off=int(offs[j]), nks=16, stash=False)); j += 1
        want = check_bwd.build_program(steps, tail, stash_l is not None, slot_bytes=(n_tiles * 64 * 256 * 2, n_tiles * 256 * 8)) if covered else None
        if D < 3 and want is not None and len(steps) - (2 if stash_l is not None else (1 if tail else 0)) < 2:
            want = None
        if want is None:
            assert got is None, (ARCHS[arch], dynamic, want_xin)
        else:
            assert got is not None, (ARCHS[arch], dynamic, want_xin)
            assert np.array_equal(got, want[:len(got)]) and len(got) == len(want), (ARCHS[arch], dynamic, want_xin, got, want)
