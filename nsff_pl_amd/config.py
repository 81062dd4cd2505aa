"""Execution options of the field kernel (not part of the reference's interface).

precision
    "f32"   -- exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
    "f16x3" -- every fp32 operand split into two halfs, three f16 MFMAs per product, fp32
               accumulation; agrees with "f32" to fp32-rounding level and passes the same
               1e-4 parity tests, ~2.7x faster.
    "f16x3_ra" -- same arithmetic, register-resident activations + LDS weight ring (experimental).
Select with ``set_precision`` or the environment variable ``NSFF_PRECISION``.
"""
import os

# "f16x3" = LDS-activation kernel (fastest so far); "f16x3_ra" = register-resident-activation
# kernel (experimental, reference depth D = 8 only; falls back to "f16x3" for other models).
PRECISIONS = {"f32": 0, "f16x3": 1, "f16x3_ra": 2}


def precision_code(model):
    code = PRECISIONS[_precision]
    if code == 2 and (model.D != 8 or model.in_channels_xyz > 63 or model.in_channels_t > 64):
        code = 1
    return code
_precision = os.environ.get("NSFF_PRECISION", "f32")
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


def set_tile_points(n):
    """f16x3 only: points per workgroup (0 = library default, 64 or 128)."""
    global _tile_points
    if n not in (0, 64, 128, 129, 130):
        raise ValueError("tile_points must be 0, 64 or 128")
    _tile_points = n


def get_tile_points():
    return _tile_points
