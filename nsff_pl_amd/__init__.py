"""MI355X-native Neural Scene Flow Fields renderer (one hot path of kwea123/nsff_pl).

Public surface = the reference's own interface for that path:

    from nsff_pl_amd import PosEmbedding, NeRF, render_rays, sample_pdf, interpolate

All arithmetic runs in the gfx950 kernels of ``csrc/`` behind the C-ABI declared in
``include/nsff_render.h``; see DESIGN.md / INTEGRATION.md.
"""
from .config import get_precision, set_precision
from .nerf import NeRF, PosEmbedding
from .rendering import render_rays, sample_pdf
from .interpolation import interpolate

__all__ = ["NeRF", "PosEmbedding", "render_rays", "sample_pdf", "interpolate", "set_precision", "get_precision"]
