"""Execution options of the field kernel (not part of the reference's interface).

precision
    "f16x3" -- (default) every fp32 operand split into two halfs, three f16 MFMAs per product, fp32
               accumulation; agrees with "f32" to fp32-rounding level and passes the same 1e-4 parity tests,
               ~2.7x faster.
    "f32"   -- exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
Select with ``set_precision`` or the environment variable ``NSFF_PRECISION``.
"""
import os

PRECISIONS = {"f32": 0, "f16x3": 1}
DEFAULT_PRECISION = "f16x3"


def precision_code(model):
    return PRECISIONS[_precision]


_precision = os.environ.get("NSFF_PRECISION", DEFAULT_PRECISION)
_tile_points = int(os.environ.get("NSFF_TILE_POINTS", "0"))
if _precision not in PRECISIONS:
    raise RuntimeError(f"NSFF_PRECISION must be one of {sorted(PRECISIONS)}")


def set_precision(name):
    global _precision
    if name not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    _precision = name


def get_precision():
    return _precision


GRAD_PRECISIONS = ("f16", "f16x3")
_grad_precision = os.environ.get("NSFF_GRAD_PRECISION", "f16")
if _grad_precision not in GRAD_PRECISIONS:
    raise RuntimeError(f"NSFF_GRAD_PRECISION must be one of {list(GRAD_PRECISIONS)}")


def set_grad_precision(name):
    """Arithmetic of the BACKWARD field kernels (csrc/field_bwd.hip).
    "f16"   -- (default, the measured path) one fp16 product per multiply-accumulate, block floating point, fp32 accumulation:
               gradients within ~1e-3 of the reference's fp32 autograd (tests/test_gradients.py).
    "f16x3" -- what the forward does: gradient tiles, transposed weights and saved activations as fp16 value + fp16 remainder,
               three products per multiply-accumulate in the data-gradient chain and in the weight-gradient GEMMs.  Gradients within
               ~1e-5 of the reference's; twice the saved-activation memory, the backward kernels at about a third of the speed.
               The training forward runs on the compiler-scheduled eight-wave kernel (it writes the remainder planes).
    Read when the forward of a field node runs (the node's backward follows its forward)."""
    global _grad_precision
    if name not in GRAD_PRECISIONS:
        raise ValueError(f"grad precision must be one of {list(GRAD_PRECISIONS)}")
    _grad_precision = name


def get_grad_precision():
    return _grad_precision


def grad_x3():
    return _grad_precision == "f16x3"


def set_tile_points(n):
    """f16x3 only: tiling of the field kernel.  0 = library default (130; inference launches below 32768 points: 64);
    64 = 64 points, four waves of 64 neurons, two workgroups per CU; 130 = 128 points per workgroup: the hand-scheduled
    body (four waves, one per SIMD, resident weights, two 64-point halves half a layer apart) for inference launches whose
    trunks it executes, else eight waves of 32 neurons (also the training forward's default); 131 = 128 points, always the
    compiler-scheduled eight-wave form (A/B comparisons, and a second implementation for the parity suite)."""
    global _tile_points
    if n not in (0, 64, 130, 131):
        raise ValueError("tile_points must be 0 (library default), 64, 130 (128 points: hand-scheduled body where it applies) "
                         "or 131 (128 points, 8 waves x 32 neurons, compiler-scheduled)")
    _tile_points = n


def get_tile_points():
    return _tile_points


_persistent = True


def set_persistent(on):
    """Large launches of the hand-scheduled field kernel are persistent by default (one workgroup per compute unit walking its
    tiles; same records bit for bit, ~1.4 % faster).  False: one workgroup per 128-point tile.  This is the process default;
    code that needs the other form for a stretch of calls uses :func:`launch_form` (scoped, restored on exit) -- nothing in the
    package changes this default behind the caller's back."""
    global _persistent
    _persistent = bool(on)


def get_persistent():
    return _persistent


class launch_form:
    """``with config.launch_form(persistent=False): ...`` -- field launches issued inside the block take the given form
    (``NsffFieldArgs::launch_form``, a per-call field of the C-ABI), the previous setting is restored on exit, also on an
    exception.  ``persistent=None`` leaves the setting alone (callers that decide at run time).

    Who uses it: the sharded frame loops (``evaluate.render_sequence_sharded``, ``bench.py --workload eval``) at world sizes
    above one -- there a frame's pixel all-gather (a RCCL kernel that waits for its peers) runs on a side stream BESIDE the next
    frame's render, and a persistent launch holds every compute unit until it ends; with one workgroup per tile the collective
    gets a compute unit at the next tile boundary (:func:`nsff_pl_amd.dist.beside_a_collective` makes the choice)."""

    def __init__(self, persistent=None):
        self.want = persistent

    def __enter__(self):
        global _persistent
        self.old = _persistent
        if self.want is not None:
            _persistent = bool(self.want)
        return self

    def __exit__(self, *exc):
        global _persistent
        _persistent = self.old
        return False
