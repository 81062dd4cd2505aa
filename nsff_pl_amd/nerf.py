"""Host-side mirror of the reference field interface (models/nerf.py).

Two classes with the reference's constructor arguments, attribute names and
``state_dict`` keys, so checkpoints and callers are interchangeable:

* :class:`PosEmbedding`  -- reference ``models/nerf.py:4-30``
* :class:`NeRF`          -- reference ``models/nerf.py:33-213``

Neither class computes anything with torch: ``forward`` hands device pointers
to the gfx950 kernels in ``csrc/`` through the C-ABI of ``include/nsff_render.h``
and fails loudly when the HIP library or a GPU tensor is missing.  The modules
still own ordinary ``nn.Parameter`` tensors (PyTorch ``Linear`` layout
``(out, in)``), which the kernels read after a one-off repack into MFMA tiles
(:mod:`nsff_pl_amd.packing`).
"""
import math

import torch
from torch import nn

from . import _lib
from . import packing

# Raw output slot order of the field kernel (one 16-float record per point).
SLOT_RGB_S, SLOT_SIGMA_S, SLOT_RGB_T, SLOT_SIGMA_T, SLOT_FW, SLOT_BW = 0, 3, 4, 7, 8, 11
RAW_STRIDE = 16


class PosEmbedding(nn.Module):
    """sin/cos frequency encoding, layout ``[x, sin(f0 x), cos(f0 x), sin(f1 x), ...]``.

    Same constructor as the reference (``models/nerf.py:5``): ``max_logscale``,
    ``N_freqs``, ``logscale``.  ``freqs`` is kept as a plain tensor attribute
    (not a buffer) exactly like the reference, so ``state_dict`` stays empty.
    """

    def __init__(self, max_logscale, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs = int(N_freqs)
        if logscale:
            self.freqs = 2 ** torch.linspace(0, max_logscale, N_freqs)
        else:
            self.freqs = torch.linspace(1, 2 ** max_logscale, N_freqs)

    @property
    def out_channels(self):
        return 3 + 6 * self.N_freqs

    def forward(self, x):
        """x: (B, 3) fp32 on the GPU -> (B, 6*N_freqs+3)  (reference nerf.py:17-30)."""
        _lib.require_gpu_tensor(x, "PosEmbedding input")
        x = x.contiguous().float()
        out = torch.empty(x.shape[0], self.out_channels, device=x.device, dtype=torch.float32)
        _lib.posenc(x, self.freqs, out)
        return out


def _relu_linear(n_in, n_out):
    return nn.Sequential(nn.Linear(n_in, n_out), nn.ReLU(True))


class NeRF(nn.Module):
    """Static (+ optional dynamic) field with the reference's parameter names.

    Constructor mirrors ``models/nerf.py:34-40``.  Parameters are created in the
    reference's order so that ``torch.manual_seed(s); NeRF(...)`` yields the same
    initial weights as the reference constructor under the same seed.
    """

    def __init__(self, typ,
                 D=8, W=256, skips=[4],
                 in_channels_xyz=63,
                 use_viewdir=True, in_channels_dir=27,
                 encode_appearance=False, in_channels_a=48,
                 encode_transient=False, in_channels_t=16,
                 output_flow=False, flow_scale=0.2):
        super().__init__()
        self.typ = typ
        self.D, self.W, self.skips = D, W, list(skips)
        self.in_channels_xyz = in_channels_xyz
        self.use_viewdir = use_viewdir
        self.in_channels_dir = in_channels_dir
        # appearance code is a fine-model-only input (reference nerf.py:67-68)
        self.encode_appearance = encode_appearance and typ != 'coarse'
        self.in_channels_a = in_channels_a if encode_appearance else 0
        self.encode_transient = encode_transient
        self.in_channels_t = in_channels_t if encode_transient else 0
        self.output_flow = bool(encode_transient and output_flow)

        self._make_trunk("static", in_channels_xyz)
        if use_viewdir:
            self.static_dir_encoding = _relu_linear(W + in_channels_dir + self.in_channels_a, W)
        self.static_sigma = nn.Linear(W, 1)
        self.static_rgb = nn.Sequential(nn.Linear(W, 3), nn.Sigmoid())

        if encode_transient:
            self._make_trunk("transient", in_channels_xyz + in_channels_t)
            self.transient_sigma = nn.Linear(W, 1)
            self.transient_rgb = nn.Sequential(nn.Linear(W, 3), nn.Sigmoid())
            if typ == 'fine' and self.output_flow:
                self.flow_scale = flow_scale
                self.transient_flow_fw = nn.Sequential(nn.Linear(W, 3), nn.Tanh())
                self.transient_flow_bw = nn.Sequential(nn.Linear(W, 3), nn.Tanh())

        self._pack_cache = packing.PackCache()

    def _make_trunk(self, prefix, n_in):
        for i in range(self.D):
            fan_in = n_in if i == 0 else (self.W + n_in if i in self.skips else self.W)
            setattr(self, f"{prefix}_xyz_encoding_{i + 1}", _relu_linear(fan_in, self.W))
        setattr(self, f"{prefix}_xyz_encoding_final", nn.Linear(self.W, self.W))

    # ------------------------------------------------------------------
    @property
    def has_flow_heads(self):
        return hasattr(self, "transient_flow_fw")

    def packed(self, precision=0):
        """Device buffer with this model's weights in the kernel's MFMA tile order."""
        return self._pack_cache.get(self, precision)

    def forward(self, x, sigma_only=False, output_static=True, output_transient=True,
                output_transient_flow=[]):
        """Same call modes and output column order as reference ``nerf.py:118-213``.

        ``x`` holds already-embedded rows ``[xyz | dir | a | t]`` (or ``[xyz | t]`` /
        ``[xyz]`` when ``sigma_only``); the rows go to the field kernel in its
        "pre-embedded input" mode.
        """
        _lib.require_gpu_tensor(x, "NeRF input")
        x = x.contiguous().float()
        B = x.shape[0]
        cx, cd, ca, ct = self.in_channels_xyz, self.in_channels_dir, self.in_channels_a, self.in_channels_t
        if sigma_only:
            off_dir, off_a, off_t = -1, -1, (cx if output_transient else -1)
            need = cx + (ct if output_transient else 0)
        else:
            off_dir, off_a = cx, cx + cd
            off_t = cx + cd + ca if output_transient else -1
            need = cx + cd + ca + (ct if output_transient else 0)
        if x.shape[1] != need:
            raise RuntimeError(f"NeRF.forward: expected {need} input channels, got {x.shape[1]}")
        if output_transient and not self.encode_transient:
            raise AttributeError("this NeRF has no transient branch")
        flows = [f for f in output_transient_flow if f in ('fw', 'bw')]
        if flows and not self.has_flow_heads:
            raise AttributeError("this NeRF has no flow heads")

        static_mode = 0 if not output_static else (1 if sigma_only else 2)
        transient_mode = 0 if not output_transient else (1 if sigma_only else 2)
        raw = torch.empty(B, RAW_STRIDE, device=x.device, dtype=torch.float32)
        _lib.field_query(self, raw, n_points=B, pts_per_ray=1,
                         static_mode=static_mode, transient_mode=transient_mode,
                         x_emb=x, emb_offsets=(0, off_dir, off_a, off_t))
        if sigma_only:
            cols = [raw[:, SLOT_SIGMA_S:SLOT_SIGMA_S + 1]]
            if output_transient:
                cols.append(raw[:, SLOT_SIGMA_T:SLOT_SIGMA_T + 1])
            return torch.cat(cols, 1)
        cols = []
        if output_static:
            cols.append(raw[:, 0:4])
        if output_transient:
            cols.append(raw[:, 4:8])
            if 'fw' in flows:
                cols.append(raw[:, SLOT_FW:SLOT_FW + 3])
            if 'bw' in flows:
                cols.append(raw[:, SLOT_BW:SLOT_BW + 3])
        return torch.cat(cols, 1)
