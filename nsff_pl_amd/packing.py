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
        self._key = {}
        self._buf = {}

    def invalidate(self):
        self._key = {}

    def get(self, model, precision=0):
        """The packed buffer of `model` for `precision` (a forward arithmetic of config.PRECISIONS, or _lib.BWD_PACK: the transposed
        tiles of the backward kernel).  Forward packs always carry the folded head rows (nsff_fold_heads): inference AND training
        launches evaluate the heads on the last trunk activation."""
        params = _lib.param_list(model)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key.get(precision):
            dev = params[0].device
            _lib.require_gpu_tensor(params[0], "model parameter")
            desc = _lib.model_desc(model)
            nbytes = _lib.bwd_packed_bytes(desc) if precision == _lib.BWD_PACK else _lib.packed_bytes(desc, precision)
            buf = self._buf.get(precision)
            if buf is None or buf.numel() * 4 != nbytes or buf.device != dev:
                buf = self._buf[precision] = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                if precision == _lib.BWD_PACK:
                    # (the folded products W_head W_final come from the forward pack of the same weights)
                    _lib.pack_weights_bwd(desc, params, buf, self.get(model, 1))
                else:
                    _lib.pack_weights(desc, params, buf, precision)
            self._key[precision] = key
        return self._buf[precision]
