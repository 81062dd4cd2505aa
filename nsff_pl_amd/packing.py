"""Per-model cache of the MFMA-tiled weight buffer.

The field kernel reads weights as pre-packed B-operand tiles (layout in
``csrc/nsff_layout.h`` / DESIGN.md).  Packing costs one small kernel batch, so it is
redone only when a parameter changed: the cache key is every parameter's
``(data_ptr, _version)`` pair, which an optimizer step or ``load_state_dict`` bumps.
"""
import torch

from . import _lib


class PackCache:
    def __init__(self):
        self._key = None
        self._buf = None

    def invalidate(self):
        self._key = None

    def get(self, model):
        params = _lib.param_list(model)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key:
            dev = params[0].device
            _lib.require_gpu_tensor(params[0], "model parameter")
            desc = _lib.model_desc(model)
            nbytes = _lib.packed_bytes(desc)
            if self._buf is None or self._buf.numel() * 4 != nbytes or self._buf.device != dev:
                self._buf = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.pack_weights(desc, params, self._buf)
            self._key = key
        return self._buf
