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


@pytest.mark.parametrize("kind", ["static", "dynamic"])
def test_simulated_trunk_matches_numpy(kind):
    assert check.run_case(kind, verbose=False) < 2e-6


def test_simulator_notices_a_missing_wait(monkeypatch):
    """the test of the test: drop the waits for the lo fragments -> an MFMA reads a register with a load still in flight"""
    orig = gen.Stream.need_lds
    monkeypatch.setattr(gen.Stream, "need_lds", lambda self, tag: None if tag[0] == "xl" else orig(self, tag))
    with pytest.raises(isa.SimError, match="outstanding"):
        check.run_case("static", verbose=False)
