"""Comparison helpers shared by the parity tests.

Tolerance: BASELINE.json's north_star asks for 1e-4 relative fp32.  "Relative" is taken in
the max-norm, per output key:  max|got-want| <= rtol * max|want|  (the survey's measure,
SURVEY.md 8c) -- per-element relative error is meaningless for weights that are ~0.

One stage of the reference is ill-conditioned by construction and needs its own bound:
``sample_pdf`` (rendering.py:10-49) places a sample inside bin j at
``bins[j] + (u - cdf[j]) / (cdf[j+1]-cdf[j]) * width`` where cdf is an fp32 running sum.
For a near-empty bin the denominator is ~1e-5 while cdf carries ~1e-7 of rounding that
depends on the summation order (torch's CPU cumsum accumulates in double, its CUDA scan
and our wavefront scan in fp32 trees), so two correct fp32 implementations differ by up
to ``width * 1e-6/denom`` there.  The weights themselves are ``(1-exp(-delta*sigma))*T``,
i.e. they carry ~1e-7 of ABSOLUTE rounding (1-exp(-x) is quantised to 6e-8 for tiny x);
when a whole ray is near-empty (sum(w+eps) ~ 6e-4, e.g. behind the eval visibility mask)
that noise is a percent-level perturbation of the pdf.  ``sample_tolerance`` returns the
first-order bound of both effects per sample; everywhere else the plain 1e-4 applies.  For the same reason the
end-to-end fine-pass keys are compared at identical depths (``tests/common.py::render_rays_at``).
"""
import numpy as np

RTOL = 1e-4


def max_rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def assert_close(name, got, want, rtol=RTOL):
    assert np.isfinite(np.asarray(got)).all(), f"{name}: non-finite values"
    err = max_rel_err(got, want)
    assert err <= rtol, f"{name}: max-norm relative error {err:.3e} > {rtol:g}"
    return err


def sample_tolerance(bins, weights, u, eps=1e-5, cdf_noise=1e-6, weight_noise=1.2e-7):
    """Per-sample absolute tolerance for sample_pdf outputs (see module docstring)."""
    bins, weights = np.asarray(bins, np.float64), np.asarray(weights, np.float64)
    n, m = weights.shape
    w = weights + eps
    pdf = w / w.sum(1, keepdims=True)
    cdf = np.concatenate([np.zeros((n, 1)), np.cumsum(pdf, 1)], 1)
    u = np.broadcast_to(np.asarray(u, np.float64), (n, np.asarray(u).shape[-1]))
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1)
    below, above = np.maximum(inds - 1, 0), np.minimum(inds, m)
    denom = np.take_along_axis(cdf, above, 1) - np.take_along_axis(cdf, below, 1)
    width = np.take_along_axis(bins, above, 1) - np.take_along_axis(bins, below, 1)
    # neighbours: a sample within cdf_noise of a bin edge may fall in either bin
    width_n = np.maximum(width, np.abs(np.take_along_axis(bins, np.minimum(above + 1, m), 1) -
                                       np.take_along_axis(bins, np.maximum(below - 1, 0), 1)))
    noise = cdf_noise + weight_noise * m / w.sum(1, keepdims=True)
    # bins with pdf below ~eps sit on the reference's `denom < eps -> 1` switch
    # (rendering.py:45): the sample may land anywhere in the bin or its neighbours
    cond = np.where(denom < 2 * eps, 1.0, noise / np.maximum(denom, eps))
    return RTOL * np.abs(bins).max() + np.minimum(width_n * cond, width_n)


def assert_samples_close(name, got, want, tol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.abs(got - want) > tol
    assert not bad.any(), (f"{name}: {bad.sum()} of {bad.size} samples outside the conditioning bound; "
                           f"worst excess {np.max(np.abs(got - want) - tol):.3e}")
