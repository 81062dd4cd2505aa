"""ctypes binding of ``libnsff_hip.so`` (C-ABI declared in ``include/nsff_render.h``).

PyTorch is used here only as the owner of device memory and of the HIP stream; every
call below passes raw device pointers and the current stream handle to the library.
There is no CPU fallback: a missing library, a CPU tensor or a non-zero return code
raises ``RuntimeError``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NSFF_LIB overrides the library path (kernel A/B experiments only)
LIB_PATH = os.environ.get("NSFF_LIB") or os.path.join(_HERE, "libnsff_hip.so")

RAW_STRIDE = 16
ABI_VERSION = 32
MAX_FREQS = 24

_ERR = {-1: "NSFF_ERR_INVALID (bad shape/flag/unsupported architecture)",
        -2: "NSFF_ERR_NULL (required pointer missing)",
        -3: "NSFF_ERR_ALIGN (pointer not 16-byte aligned)",
        -4: "NSFF_ERR_HIP"}

_fp = C.c_void_p  # device pointers travel as integers


class ModelDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("skip", C.c_int32),
                ("in_xyz", C.c_int32), ("in_dir", C.c_int32), ("in_a", C.c_int32),
                ("in_t", C.c_int32), ("use_viewdir", C.c_int32),
                ("has_transient", C.c_int32), ("has_flow", C.c_int32),
                ("flow_scale", C.c_float), ("skip_mask", C.c_int32)]


class FieldArgs(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("precision", C.c_int32), ("tile_points", C.c_int32),
                ("pts_per_ray", C.c_int32),
                ("static_mode", C.c_int32), ("transient_mode", C.c_int32),
                ("flow_heads", C.c_int32),
                ("xyz", _fp), ("n_freqs", C.c_int32), ("freqs", C.c_float * MAX_FREQS),
                ("dir_emb", _fp), ("a_emb", _fp), ("t_emb", _fp),
                ("x_emb", _fp), ("ld_emb", C.c_int32),
                ("off_xyz", C.c_int32), ("off_dir", C.c_int32), ("off_a", C.c_int32),
                ("off_t", C.c_int32), ("raw", _fp), ("save_acts", _fp), ("save_xin", _fp), ("save_masks", _fp),
                ("save_side", _fp), ("t_bias", _fp), ("t_bias_rows", C.c_int32), ("reserved0", C.c_int32),
                ("s_bias", _fp), ("s_bias_rows", C.c_int32), ("launch_form", C.c_int32), ("save_lo_delta", C.c_int64 * 3)]


class TimeBiasJob(C.Structure):
    _fields_ = [("desc", C.POINTER(ModelDesc)), ("w", _fp * 8), ("b", _fp * 8), ("t_rows", _fp), ("out", _fp),
                ("table", _fp), ("ts", _fp), ("n_table", C.c_int64), ("max_t", C.c_int64), ("delta", C.c_int32), ("pad_", C.c_int32),
                ("rows_out", _fp)]


MAX_TIME_BIAS_JOBS = 4


class RngJob(C.Structure):
    _fields_ = [("out", _fp), ("numel", C.c_int64), ("offset", C.c_uint64), ("kind", C.c_int32), ("grid", C.c_uint32)]


MAX_RNG_JOBS = 12


class RngCoarse(C.Structure):
    _fields_ = [("rays", _fp), ("z_lin", _fp), ("zs", _fp), ("xyz", _fp), ("n_samples", C.c_int32), ("perturb", C.c_float),
                ("job", C.c_int32), ("pad_", C.c_int32)]


_COMPOSITE_PTRS_IN = ["raw", "raw_fw", "raw_bw", "zs", "xyz", "xyz_fw", "xyz_bw",
                      "noise_static", "noise_transient", "noise_fw", "noise_bw", "visibility"]
_COMPOSITE_PTRS_OUT = ["static_rgbs", "transient_rgbs", "flows_fw", "flows_bw",
                       "static_sigmas", "transient_sigmas", "static_alphas", "transient_alphas",
                       "static_weights", "transient_weights", "weights",
                       "xyzs_fw_bw", "xyzs_bw_fw", "disoccs_fw", "disoccs_bw",
                       "depth", "rgb", "transient_alpha", "transient_rgb",
                       "static_only_rgb", "static_only_depth",
                       "xyz_exp", "flow_fw_exp", "flow_bw_exp", "xyz_fw_exp", "xyz_bw_exp",
                       "rgb_fw", "rgb_bw", "disocc_fw", "disocc_bw"]


class FrustumArgs(C.Structure):
    _fields_ = [("w2c", _fp), ("ts", _fp), ("K4", C.c_float * 4), ("n_cams", C.c_int32), ("n_frames", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32)]


class CompositeArgs(C.Structure):
    _fields_ = ([("n_rays", C.c_int64), ("n_samples", C.c_int32), ("has_transient", C.c_int32),
                 ("has_rgb", C.c_int32), ("flow_mode", C.c_int32), ("want_disocc", C.c_int32),
                 ("noise_std", C.c_float), ("z_far", C.c_float)]
                + [(n, _fp) for n in _COMPOSITE_PTRS_IN]
                + [(n, _fp) for n in _COMPOSITE_PTRS_OUT]
                + [("vis", FrustumArgs)])


_CBWD_IN = ["raw", "raw_fw", "raw_bw", "zs", "xyz", "f_fw", "f_bw", "noise_static", "noise_transient", "noise_fw", "noise_bw"]
_CBWD_G = ["g_static_sigmas", "g_transient_sigmas", "g_static_weights", "g_transient_weights", "g_weights", "g_depth",
           "g_rgb", "g_transient_alpha", "g_transient_rgb", "g_so_rgb", "g_so_depth", "g_xyz_exp", "g_flow_fw_exp",
           "g_flow_bw_exp", "g_rgb_fw", "g_rgb_bw"]
_CBWD_OUT = ["scratch", "d_raw", "d_raw_fw", "d_raw_bw", "d_f_fw", "d_f_bw"]


class CompositeBwdArgs(C.Structure):
    _fields_ = ([("n_rays", C.c_int64), ("n_samples", C.c_int32), ("has_transient", C.c_int32), ("flow_mode", C.c_int32),
                 ("noise_std", C.c_float)] + [(n, _fp) for n in _CBWD_IN + _CBWD_G + _CBWD_OUT])


class FieldBwdArgs(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("static_mode", C.c_int32), ("transient_mode", C.c_int32),
                ("d_raw", _fp), ("raw", _fp), ("gmax", _fp), ("masks", _fp), ("dpre", _fp), ("dhead", _fp),
                ("d_xin", _fp), ("d_side", _fp), ("dpre_lo_delta", C.c_int64)]


class WgradJob(C.Structure):
    _fields_ = [("a", _fp), ("b", _fp), ("a_rows", C.c_int32), ("b_rows", C.c_int32), ("out_off", C.c_int64), ("trunk", C.c_int32),
                ("pad_", C.c_int32), ("a_lo_delta", C.c_int64), ("b_lo_delta", C.c_int64)]


class FoldGradArgs(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("pad_", C.c_int32), ("g", _fp), ("gb", _fp), ("w_final", _fp), ("b_final", _fp),
                ("w_head", _fp * 16), ("d_w_head", _fp * 16), ("d_b_head", _fp * 16), ("d_w_final", _fp), ("d_b_final", _fp)]


class FoldDenseArgs(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("accumulate", C.c_int32), ("ld_head", C.c_int32), ("ld_dhead", C.c_int32),
                ("g", _fp), ("g2", _fp), ("gb", _fp), ("gb2", _fp), ("w_head", _fp), ("w_final", _fp), ("b_final", _fp),
                ("d_w_head", _fp), ("d_b_head", _fp), ("d_w_final", _fp), ("d_b_final", _fp)]


class SplatArgs(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("n_planes", C.c_int32), ("K4", C.c_float * 4),
                ("P", C.c_float * 12), ("scale", C.c_float),
                ("xyz", _fp), ("flow", _fp), ("rgb", _fp), ("alpha", _fp), ("accum", _fp), ("work", _fp),
                ("work_bytes", C.c_int64)]


class MpiArgs(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("n_planes", C.c_int32), ("dt", C.c_float),
                ("accum_fw", _fp), ("accum_bw", _fp), ("static_rgb", _fp), ("static_alpha", _fp), ("zs", _fp),
                ("rgb", _fp), ("depth", _fp)]


_LOSS_IN = ["rgb_fine", "rgb_coarse", "depth_fine", "depth_coarse", "t_weights", "s_weights", "xyz_fw", "xyz_bw", "rgb_fw",
            "rgb_bw", "disocc_fw", "disocc_bw", "disoccs_fw", "disoccs_bw", "xyzs_fw_bw", "xyzs_bw_fw", "xyzs_fine",
            "xyzs_fw", "xyzs_bw", "rgbs", "disps", "ts", "cam_ids", "uv_fw", "uv_bw", "Ks", "Ps", "hyper", "stats", "terms",
            "term_w"]
LOSS_GRADS = ["g_rgb_fine", "g_rgb_coarse", "g_depth_fine", "g_depth_coarse", "g_t_weights", "g_s_weights", "g_xyz_fw",
              "g_xyz_bw", "g_rgb_fw", "g_rgb_bw", "g_xyzs_fw_bw", "g_xyzs_bw_fw", "g_xyzs_fw", "g_xyzs_bw"]


class LossArgs(C.Structure):
    _fields_ = ([("n_rays", C.c_int64), ("n_samples", C.c_int32), ("n_keep", C.c_int32), ("n_frames", C.c_int32),
                 ("max_t", C.c_int32), ("topk", C.c_double), ("thickness", C.c_int32), ("pad_", C.c_int32)]
                + [(n, _fp) for n in _LOSS_IN] + [("weights", _fp), ("per_ray", _fp), ("coef", _fp)]
                + [(n, _fp) for n in LOSS_GRADS])


class FlowGradArgs(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("zs", _fp), ("z_far", C.c_float), ("accumulate", C.c_int32), ("col_a", C.c_int32),
                ("col_b", C.c_int32), ("pad_", C.c_int32), ("g_a", _fp * 4), ("g_b", _fp * 4), ("out", _fp)]


# name -> (restype, argtypes); also the list of symbols the header declares
_SIGNATURES = {
    "nsff_abi_version": (C.c_int, []),
    "nsff_last_field_kernel": (C.c_int, []),
    "nsff_last_field_grid": (C.c_int, []),
    "nsff_field_phase_program": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "nsff_time_bias_rows": (C.c_int, [C.POINTER(ModelDesc)]),
    "nsff_time_bias": (C.c_int, [C.POINTER(TimeBiasJob), C.c_int32, C.c_int64, C.c_void_p]),
    "nsff_rng_draws": (C.c_int, [C.POINTER(RngJob), C.c_int32, C.c_uint64, C.c_void_p]),
    "nsff_rng_draws_coarse": (C.c_int, [C.POINTER(RngJob), C.c_int32, C.c_uint64, C.POINTER(RngCoarse), C.c_void_p]),
    "nsff_side_bias": (C.c_int, [C.POINTER(ModelDesc), _fp, _fp, _fp, _fp, C.c_int64, _fp, C.c_void_p]),
    "nsff_last_hip_error": (C.c_char_p, []),
    "nsff_packed_bytes": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(C.c_size_t)]),
    "nsff_param_count": (C.c_int, [C.POINTER(ModelDesc)]),
    "nsff_pack_weights": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_fp), _fp, _fp]),
    "nsff_pack_weights_ex": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_fp), _fp, C.c_int32, _fp]),
    "nsff_fold_heads": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_fp), _fp, _fp]),
    "nsff_posenc": (C.c_int, [_fp, C.c_int64, C.POINTER(C.c_float), C.c_int, _fp, _fp]),
    "nsff_time_rows": (C.c_int, [_fp, C.c_int64, C.c_int32, _fp, C.c_int64, C.c_int64, _fp, _fp, _fp]),
    "nsff_time_rows_backward": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _fp, _fp]),
    "nsff_field_query": (C.c_int, [C.POINTER(ModelDesc), _fp, C.POINTER(FieldArgs), _fp]),
    "nsff_coarse_samples": (C.c_int, [_fp, C.c_int64, _fp, C.c_int32, C.c_float, _fp, _fp, _fp, _fp]),
    "nsff_fine_samples": (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp,
                                    C.c_int32, _fp, _fp, _fp, _fp, _fp]),
    "nsff_sample_pdf": (C.c_int, [_fp, _fp, C.c_int64, C.c_int32, _fp, C.c_int32, C.c_int32,
                                  C.c_float, _fp, _fp]),
    "nsff_warp_points": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_float, _fp, _fp, _fp]),
    "nsff_composite": (C.c_int, [C.POINTER(CompositeArgs), _fp]),
    "nsff_frustum_visibility": (C.c_int, [C.POINTER(FrustumArgs), _fp, C.c_int64, _fp, _fp]),
    "nsff_frame_rays": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_float,
                                  C.c_float, C.c_int64, C.c_int64, _fp, _fp]),
    "nsff_bwd_packed_bytes": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_size_t)]),
    "nsff_pack_weights_bwd": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_fp), _fp, _fp]),
    "nsff_field_backward": (C.c_int, [C.POINTER(ModelDesc), _fp, C.POINTER(FieldBwdArgs), _fp]),
    "nsff_field_input_backward": (C.c_int, [_fp, C.c_int32, C.c_int32, _fp, C.c_int64, C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                            C.c_int32, _fp, _fp, _fp]),
    "nsff_train_dims": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nsff_weight_grad_scratch": (C.c_int64, [C.POINTER(WgradJob), C.c_int32, C.c_int64, C.c_int32]),
    "nsff_weight_grad": (C.c_int, [C.POINTER(WgradJob), C.c_int32, C.c_int64, C.c_int32, _fp, _fp, _fp, _fp, _fp]),
    "nsff_weight_grad_accumulate": (C.c_int, [C.POINTER(WgradJob), C.c_int32, C.c_int64, C.c_int32, _fp, _fp, C.c_int64,
                                              _fp, _fp, _fp]),
    "nsff_weight_grad_accumulate_aux": (C.c_int, [C.POINTER(WgradJob), C.c_int32, C.c_int64, C.c_int32, _fp, _fp, C.c_int64,
                                                  _fp, _fp, _fp, _fp]),
    "nsff_absmax_raw": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "nsff_last_bwd_kernel": (C.c_int, []),
    "nsff_last_bwd_grid": (C.c_int, []),
    "nsff_field_bwd_phase_program": (C.c_int, [C.POINTER(ModelDesc), C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_uint32), C.c_int32,
                                                C.POINTER(C.c_uint32), C.c_int32]),
    "nsff_fold_grads": (C.c_int, [C.POINTER(FoldGradArgs), _fp]),
    "nsff_fold_grads_dense": (C.c_int, [C.POINTER(FoldDenseArgs), _fp]),
    "nsff_pack_weights_bwd_ex": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_fp), _fp, _fp, _fp]),
    "nsff_absmax": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "nsff_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, _fp, _fp, C.c_double, C.c_double, C.c_double, C.c_double, _fp]),
    "nsff_adam_step_segments": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, _fp, _fp, C.c_double, C.c_double, C.c_double, C.c_double,
                                          _fp, C.c_int, _fp, _fp]),
    "nsff_composite_backward": (C.c_int, [C.POINTER(CompositeBwdArgs), _fp]),
    "nsff_flow_grad": (C.c_int, [C.POINTER(FlowGradArgs), _fp]),
    "nsff_nerfw_loss": (C.c_int, [C.POINTER(LossArgs), C.c_int, _fp]),
    "nsff_splat_planes": (C.c_int, [C.POINTER(SplatArgs), _fp]),
    "nsff_splat_work_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "nsff_mpi_composite": (C.c_int, [C.POINTER(MpiArgs), _fp]),
    "nsff_prof_enable": (C.c_int, [C.c_int]),
    "nsff_prof_collect": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nsff_prof_collect_clock": (C.c_int, [C.POINTER(C.c_int64)] + [C.POINTER(C.c_double)] * 5),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C nsff_pl_amd/csrc`).  There is no CPU fallback for the render path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.nsff_abi_version() != ABI_VERSION:
            raise RuntimeError("libnsff_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f"error {rc}")
        if rc == -4:
            msg += ": " + load().nsff_last_hip_error().decode()
        raise RuntimeError(f"{what} failed: {msg}")


def require_gpu_tensor(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{what} must be a GPU tensor: the NSFF render path runs only on the "
                           "HIP kernels of libnsff_hip.so (no CPU fallback)")


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "need contiguous fp32 GPU tensor"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------------
def model_desc(model):
    """NsffModelDesc of a NeRF module.  The reference constructor (models/nerf.py:34-40) takes any width W and a
    list of skip layers; the gfx950 kernels are built for W = 256 trunks, 2..8 layers deep, with any set of skip layers
    among 1..D-1 (inference and training alike) -- anything else is
    refused here, by name."""
    if model.W != 256:
        raise RuntimeError(f"unsupported NeRF architecture: W={model.W} (the gfx950 field kernels tile W=256 trunks "
                           "into 4 x 64-neuron wave blocks; other widths are not built)")
    skips = sorted(set(int(s) for s in model.skips))
    if not 2 <= model.D <= 8 or any(not 1 <= s < model.D for s in skips):
        raise RuntimeError(f"unsupported NeRF architecture: D={model.D}, skips={list(model.skips)} (need 2 <= D <= 8 "
                           "and every skip layer in 1..D-1; a skip at layer 0 would concatenate the input with itself)")
    mask = sum(1 << s for s in skips)
    return ModelDesc(D=model.D, W=model.W, skip=skips[0] if len(skips) == 1 else 0, skip_mask=mask if len(skips) != 1 else 0,
                     in_xyz=model.in_channels_xyz,
                     in_dir=model.in_channels_dir,
                     in_a=model.in_channels_a if model.use_viewdir else 0,
                     in_t=model.in_channels_t, use_viewdir=int(model.use_viewdir),
                     has_transient=int(model.encode_transient),
                     has_flow=int(getattr(model, "has_flow_heads", hasattr(model, "transient_flow_fw"))),
                     flow_scale=float(getattr(model, "flow_scale", 0.0)))


def param_list(model):
    """Parameter tensors in the order nsff_pack_weights documents."""
    def lin(m):
        layer = m[0] if isinstance(m, torch.nn.Sequential) else m
        return [layer.weight, layer.bias]
    out = []
    for i in range(model.D):
        out += lin(getattr(model, f"static_xyz_encoding_{i + 1}"))
    out += lin(model.static_xyz_encoding_final)
    if model.use_viewdir:
        out += lin(model.static_dir_encoding)
    out += lin(model.static_sigma) + lin(model.static_rgb)
    if model.encode_transient:
        for i in range(model.D):
            out += lin(getattr(model, f"transient_xyz_encoding_{i + 1}"))
        out += lin(model.transient_xyz_encoding_final)
        out += lin(model.transient_sigma) + lin(model.transient_rgb)
        if hasattr(model, "transient_flow_fw"):
            out += lin(model.transient_flow_fw) + lin(model.transient_flow_bw)
    return out


def packed_bytes(desc, precision=0):
    n = C.c_size_t(0)
    _check(load().nsff_packed_bytes(C.byref(desc), int(precision), C.byref(n)), "nsff_packed_bytes")
    return n.value


PACK_SKIP_FOLD = 1


def pack_weights(desc, params, packed, precision=0, fold=True):
    """fold=False: without the folded head rows inference launches read (include/nsff_render.h: NSFF_PACK_SKIP_FOLD)."""
    lib = load()
    if lib.nsff_param_count(C.byref(desc)) != len(params):
        raise RuntimeError("parameter list does not match the model description")
    keep = [p.detach().contiguous().float() for p in params]
    for p in keep:
        require_gpu_tensor(p, "model parameter")
    arr = (_fp * len(keep))(*[p.data_ptr() for p in keep])
    _check(lib.nsff_pack_weights_ex(C.byref(desc), int(precision), arr, _ptr(packed), 0 if fold else PACK_SKIP_FOLD,
                                    _stream()), "nsff_pack_weights_ex")
    return keep  # caller keeps temporaries alive until the stream has consumed them


def fold_heads(desc, params, packed, precision):
    keep = [p.detach().contiguous().float() for p in params]
    arr = (_fp * len(keep))(*[p.data_ptr() for p in keep])
    _check(load().nsff_fold_heads(C.byref(desc), int(precision), arr, _ptr(packed), _stream()), "nsff_fold_heads")
    return keep


def time_rows_backward(g_cur, g_next, g_prev, ts, max_t, n_table):
    """d_table (n_table, width) of E[ts], E[clamp(ts + 1)], E[clamp(ts - 1)] for the (n, width) cotangents given (None = absent)."""
    gs = [None if g is None else g.contiguous() for g in (g_cur, g_next, g_prev)]
    width = next(g for g in gs if g is not None).shape[1]
    assert ts.dtype == torch.int64 and ts.is_cuda and ts.is_contiguous()
    d = torch.empty(int(n_table), width, device=ts.device)
    _check(load().nsff_time_rows_backward(_ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), C.c_void_p(ts.data_ptr()), ts.shape[0],
                                          int(max_t), int(n_table), width, _ptr(d), _stream()), "nsff_time_rows_backward")
    return d


def time_rows(table, ts, max_t, want_next=True, want_prev=True):
    """(E[clamp(ts + 1, max=max_t)], E[clamp(ts - 1, min=0)]) of an embedding table in one launch (rendering.py:218,224)."""
    n, width = ts.shape[0], table.shape[1]
    if want_next and want_prev:     # (one buffer: the two re-queries of a call can then run as ONE field launch over 2 n rays)
        both = torch.empty(2, n, width, device=table.device)
        nxt, prv = both[0], both[1]
    else:
        nxt = torch.empty(n, width, device=table.device) if want_next else None
        prv = torch.empty(n, width, device=table.device) if want_prev else None
    assert ts.dtype == torch.int64 and ts.is_cuda and ts.is_contiguous()
    _check(load().nsff_time_rows(_ptr(table), table.shape[0], width, C.c_void_p(ts.data_ptr()), n, int(max_t),
                                 _ptr(nxt), _ptr(prv), _stream()), "nsff_time_rows")
    return nxt, prv


def posenc(x, freqs, out):
    f = [float(v) for v in freqs]
    arr = (C.c_float * len(f))(*f)
    _check(load().nsff_posenc(_ptr(x), x.shape[0], arr, len(f), _ptr(out), _stream()), "nsff_posenc")


def field_query(model, raw, n_points, pts_per_ray, static_mode, transient_mode, flow_heads=0,
                xyz=None, freqs=None, dir_emb=None, a_emb=None, t_emb=None,
                x_emb=None, emb_offsets=(0, -1, -1, -1), save_acts=None, save_xin=None, save_masks=None, save_side=None,
                precision=None, t_bias=None, s_bias=None, save_lo=False):
    from . import config
    desc = model_desc(model)
    prec = config.precision_code(model) if precision is None else precision
    saves = not (save_acts is None and save_xin is None and save_masks is None and save_side is None)
    # (training forwards run the folded step program too: one pack form)
    packed = model.packed(prec)
    a = FieldArgs()
    a.precision, a.tile_points = prec, config.get_tile_points()
    a.launch_form = 0 if config.get_persistent() else 1
    a.n_points, a.pts_per_ray = int(n_points), int(pts_per_ray)
    a.static_mode, a.transient_mode, a.flow_heads = int(static_mode), int(transient_mode), int(flow_heads)
    a.xyz = _ptr(xyz)
    if freqs is not None:
        f = [float(v) for v in freqs]
        if len(f) > MAX_FREQS:
            raise RuntimeError("too many embedding frequencies")
        a.n_freqs = len(f)
        for i, v in enumerate(f):
            a.freqs[i] = v
    a.dir_emb, a.a_emb, a.t_emb = _ptr(dir_emb), _ptr(a_emb), _ptr(t_emb)
    a.x_emb = _ptr(x_emb)
    a.ld_emb = int(x_emb.shape[1]) if x_emb is not None else 0
    a.off_xyz, a.off_dir, a.off_a, a.off_t = [int(v) for v in emb_offsets]
    a.raw = _ptr(raw)
    a.save_acts = None if save_acts is None else save_acts.data_ptr()
    a.save_xin = None if save_xin is None else save_xin.data_ptr()
    a.save_masks = None if save_masks is None else save_masks.data_ptr()
    a.save_side = None if save_side is None else save_side.data_ptr()
    if save_lo:                                 # (field_grad.alloc_saves: the remainder plane of a saved tensor directly behind it)
        for i, t in enumerate((save_acts, save_xin, save_side)):
            a.save_lo_delta[i] = 0 if t is None else lo_delta(t)
    if t_bias is not None:                      # (n_rays, rows, 256) from time_bias(): the time code's part of the input layers
        a.t_bias, a.t_bias_rows = _ptr(t_bias), int(t_bias.shape[1])
    if s_bias is not None:                      # (n_rays, 1, 256) from side_bias(): [dir | a]'s part of static_dir_encoding
        a.s_bias, a.s_bias_rows = _ptr(s_bias), int(s_bias.shape[1])
    _check(load().nsff_field_query(C.byref(desc), _ptr(packed), C.byref(a), _stream()), "nsff_field_query")


def lo_delta(t):
    """Elements from a tensor to its remainder twin: the three-product buffers are allocated as (2, ...) -- plane 0 the fp16 values
    (the tensor every caller holds), plane 1 value - fp16(value) -- so the twin of any slice of plane 0 sits numel(plane) further on."""
    n = t.numel()
    if t.untyped_storage().nbytes() < (t.storage_offset() + 2 * n) * t.element_size():
        raise RuntimeError("three-product backward: this buffer has no remainder plane behind it (field_grad.alloc_saves(..., x3=True))")
    return n


def side_bias(model, dir_rows, a_rows=None):
    """(n_rays, 1, 256) fp32: per ray, the folded bias + the [dir | a] columns' product of static_dir_encoding (nsff_side_bias);
    field_query(..., s_bias=) then runs the view-direction static trunk on the hand-scheduled kernel."""
    from . import config
    desc = model_desc(model)
    n_rays = int(dir_rows.shape[0])
    assert dir_rows.shape == (n_rays, desc.in_dir) and (desc.in_a == 0 or (a_rows is not None and a_rows.shape == (n_rays, desc.in_a)))
    packed = model.packed(config.PRECISIONS["f16x3"])      # (the folded bias row lives in the inference pack)
    w = model.static_dir_encoding[0].weight.detach()
    out = torch.empty(n_rays, 1, 256, device=dir_rows.device, dtype=torch.float32)
    _check(load().nsff_side_bias(C.byref(desc), _ptr(packed), _ptr(w), _ptr(dir_rows), _ptr(a_rows) if desc.in_a > 0 else None,
                                 n_rays, _ptr(out), _stream()), "nsff_side_bias")
    return out


_DEVICE_RNG_GEOMETRY = {}


def fused_draws(plan, device, values=True, coarse=None):
    """The draws ``[(kind, shape)]`` (kind "rand" | "randn") of torch's default generator of `device`, in order, by ONE launch
    (nsff_rng_draws): bit-identical to calling torch.rand / torch.randn in that order, and the generator is left where those
    calls would leave it.  values=False for entries whose numbers nobody reads: give a list of booleans -- such a draw only
    advances the generator (its tensor is None).  -> list of float32 tensors.
    coarse = (rays, z_lin, perturb, zs, xyz): plan[0] is the stratified-sampling draw of the call (perturb > 0); the launch
    computes the coarse depths from it where it is drawn (nsff_rng_draws_coarse) -- `coarse_samples` is not needed then."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _DEVICE_RNG_GEOMETRY:
        p = torch.cuda.get_device_properties(idx)
        _DEVICE_RNG_GEOMETRY[idx] = p.multi_processor_count * (p.max_threads_per_multi_processor // 256)
    max_grid = _DEVICE_RNG_GEOMETRY[idx]
    gen = torch.cuda.default_generators[idx]
    seed, offset = gen.initial_seed(), gen.get_offset()
    wanted = values if isinstance(values, (list, tuple)) else [bool(values)] * len(plan)
    sizes = []
    for kind, shape in plan:
        n = 1
        for s in shape:
            n *= int(s)
        sizes.append(n)
    total = sum(n for n, w in zip(sizes, wanted) if w)
    buf = torch.empty(max(total, 1), device=dev, dtype=torch.float32)        # ONE allocation; the draws are views of it
    outs, jobs, at = [], [], 0
    for (kind, shape), numel, want in zip(plan, sizes, wanted):
        if numel == 0:                              # (torch returns before it touches the generator)
            outs.append(torch.empty(*shape, device=dev, dtype=torch.float32))
            continue
        grid = min(max_grid, (numel + 255) // 256)
        if coarse is not None and not outs and not want:     # the perturbation draw: consumed by the launch itself
            jobs.append((None, numel, offset, 0, grid))
            outs.append(None)
        elif want:
            t = buf[at:at + numel].view(*shape)
            at += numel
            jobs.append((t, numel, offset, 1 if kind == "randn" else 0, grid))
            outs.append(t)
        else:
            outs.append(None)
        offset += ((numel - 1) // (1024 * grid) + 1) * 4
    hook = None
    if coarse is not None:
        rays, z_lin, perturb, zs, xyz = coarse
        assert plan and plan[0][0] == "rand" and tuple(plan[0][1]) == (rays.shape[0], z_lin.shape[0]) and perturb > 0
        if sizes[0]:
            hook = RngCoarse(_ptr(rays), _ptr(z_lin), _ptr(zs), _ptr(xyz), int(z_lin.shape[0]), float(perturb), 0, 0)
    for k in range(0, len(jobs), MAX_RNG_JOBS):
        part = jobs[k:k + MAX_RNG_JOBS]
        arr = (RngJob * len(part))()
        for j, (t, numel, off, kind, grid) in enumerate(part):
            arr[j].out, arr[j].numel, arr[j].offset, arr[j].kind, arr[j].grid = (None if t is None else t.data_ptr()), numel, off, kind, grid
        with torch.cuda.device(idx):
            _check(load().nsff_rng_draws_coarse(arr, len(part), seed, C.byref(hook) if (hook is not None and k == 0) else None, _stream()),
                   "nsff_rng_draws")
    gen.set_offset(offset)
    return outs


_FUSED_DRAWS_OK = {}


def fused_draws_match_torch(device):
    """True when :func:`fused_draws` reproduces torch's generator on `device` -- checked ONCE per device, on first use: fused_draws
    restates three ATen internals (the launch geometry of calc_execution_policy, the Philox offset accounting, the uniform /
    normal transforms), and a torch or hiprand upgrade could change any of them without an error.  A few thousand rand / randn
    numbers (several grid sizes, a size that is not a multiple of four) are drawn both ways from the same saved generator state
    and compared bit for bit, the generator's final offset included; the state is restored afterwards, so the check draws
    nothing from the caller's stream of numbers.  On a mismatch the render path uses the torch calls (as NSFF_TORCH_RNG=1 does)
    and says so once."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _FUSED_DRAWS_OK:
        return _FUSED_DRAWS_OK[idx]
    if torch.cuda.is_current_stream_capturing():          # (never decided inside a capture: the caller draws with torch there anyway)
        return False
    gen = torch.cuda.default_generators[idx]
    state = gen.get_state()
    ok = True
    try:
        plan = [("rand", (7, 64)), ("randn", (1031,)), ("rand", (3, 5, 2)), ("randn", (70000,)), ("rand", (300001,))]
        gen.manual_seed(0x5EED1234)
        gen.set_offset(8)
        mine = fused_draws(plan, torch.device("cuda", idx))
        end_mine = gen.get_offset()
        gen.manual_seed(0x5EED1234)
        gen.set_offset(8)
        with torch.cuda.device(idx):
            ref = [(torch.rand if k == "rand" else torch.randn)(*shape, device=torch.device("cuda", idx)) for k, shape in plan]
        end_ref = gen.get_offset()
        ok = end_mine == end_ref and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(mine, ref))
    except Exception:                                      # an API that moved is a mismatch as well
        ok = False
    finally:
        gen.set_state(state)
    if not ok:
        import warnings
        warnings.warn("nsff_pl_amd: the fused generator kernel (nsff_rng_draws) no longer reproduces torch.rand / torch.randn on "
                      f"cuda:{idx} (torch {torch.__version__}); render_rays draws with the torch calls instead (slower by ~1 %, same "
                      "numbers as the reference)")
    _FUSED_DRAWS_OK[idx] = ok
    return ok


def time_bias_rows(model):
    """rows per ray of time_bias() for this model (0: no dynamic trunk)"""
    return load().nsff_time_bias_rows(C.byref(model_desc(model)))


def time_bias(jobs, index=None):
    """jobs: [(model, t_rows (n_rays, in_t))] (at most MAX_TIME_BIAS_JOBS, one n_rays) -> [(n_rays, rows, 256) fp32]: per ray,
    bias + time-code columns' product of the dynamic trunk's layer 0 and skip layers (nsff_time_bias: ONE launch for all
    jobs); field_query(..., t_bias=) then multiplies no time-code column.
    index = (table (n_table, in_t), ts (n_rays,) int64, max_t): jobs are [(model, delta)], delta in (0, +1, -1) -- the launch
    gathers table[clamp(ts + delta)] itself (no separate gather / neighbour-row launches) and ALSO returns the gathered rows:
    -> (outs, {delta: (n_rays, in_t)}); the rows of +1 and -1 are the two halves of one buffer."""
    assert 1 <= len(jobs) <= MAX_TIME_BIAS_JOBS
    arr = (TimeBiasJob * len(jobs))()
    keep, outs, per_model, pair = [], [], {}, None
    rows_of = {}
    if index is not None:
        table, ts, max_t = index
        assert ts.dtype == torch.int64 and ts.is_cuda and ts.is_contiguous() and ts.dim() == 1, "ts: one int64 frame index per ray"
        n_rays, width = int(ts.shape[0]), int(table.shape[1])
        deltas = sorted({int(d) for _, d in jobs})
        if 1 in deltas and -1 in deltas:
            both = torch.empty(2, n_rays, width, device=table.device, dtype=torch.float32)
            rows_of[1], rows_of[-1] = both[0], both[1]
        for d in deltas:
            if d not in rows_of:
                rows_of[d] = torch.empty(n_rays, width, device=table.device, dtype=torch.float32)
        written = set()
    else:
        n_rays = int(jobs[0][1].shape[0])
    for j, (model, t_rows) in enumerate(jobs):
        if id(model) not in per_model:
            desc = model_desc(model)
            lins = [getattr(model, f"transient_xyz_encoding_{l + 1}")[0] for l in [0] + sorted(set(int(v) for v in model.skips))]
            wb = [(m.weight.detach(), m.bias.detach()) for m in lins]
            for w, b in wb:
                _ptr(w), _ptr(b)                # (contiguous fp32 GPU tensors, or an assertion)
            per_model[id(model)] = (desc, C.pointer(desc), wb)
        desc, pdesc, wb = per_model[id(model)]
        if index is not None:
            delta, t_rows = int(t_rows), None
            assert width == desc.in_t
        else:
            assert t_rows.shape == (n_rays, desc.in_t)
        # (consecutive jobs of one model get adjacent halves of one buffer: see rendering._inference's merged re-query)
        dev = wb[0][0].device
        if j + 1 < len(jobs) and jobs[j + 1][0] is model and pair is None:
            pair = torch.empty(2, n_rays, len(wb), 256, device=dev, dtype=torch.float32)
            out = pair[0]
        elif pair is not None:
            out, pair = pair[1], None
        else:
            out = torch.empty(n_rays, len(wb), 256, device=dev, dtype=torch.float32)
        arr[j].desc = pdesc
        for i, (w, b) in enumerate(wb):
            arr[j].w[i], arr[j].b[i] = w.data_ptr(), b.data_ptr()
        arr[j].t_rows, arr[j].out = _ptr(t_rows), _ptr(out)
        if index is not None:
            arr[j].table, arr[j].ts = _ptr(table), ts.data_ptr()
            arr[j].n_table, arr[j].max_t, arr[j].delta = int(table.shape[0]), int(max_t), delta
            if delta not in written:                # (the first job of a delta leaves the gathered rows behind)
                arr[j].rows_out = _ptr(rows_of[delta])
                written.add(delta)
        outs.append(out)
    keep.append(per_model)
    _check(load().nsff_time_bias(arr, len(jobs), n_rays, _stream()), "nsff_time_bias")
    return outs if index is None else (outs, rows_of)


def coarse_samples(rays, z_lin, perturb, perturb_rand, zs, xyz):
    _check(load().nsff_coarse_samples(_ptr(rays), rays.shape[0], _ptr(z_lin), z_lin.shape[0],
                                      float(perturb), _ptr(perturb_rand), _ptr(zs), _ptr(xyz), _stream()),
           "nsff_coarse_samples")


def fine_samples(rays, z_lin, zs_coarse, n_importance, w_static, w_transient, u_static, u_transient,
                 u_per_ray, samples_static, samples_transient, zs_fine, xyz_fine):
    _check(load().nsff_fine_samples(_ptr(rays), rays.shape[0], _ptr(z_lin), _ptr(zs_coarse),
                                    z_lin.shape[0], int(n_importance), _ptr(w_static), _ptr(w_transient),
                                    _ptr(u_static), _ptr(u_transient), int(u_per_ray),
                                    _ptr(samples_static), _ptr(samples_transient),
                                    _ptr(zs_fine), _ptr(xyz_fine), _stream()), "nsff_fine_samples")


def sample_pdf(bins, weights, u, u_per_ray, eps, samples):
    _check(load().nsff_sample_pdf(_ptr(bins), _ptr(weights), weights.shape[0], weights.shape[1],
                                  _ptr(u), samples.shape[1], int(u_per_ray), float(eps),
                                  _ptr(samples), _stream()), "nsff_sample_pdf")


def warp_points(raw, xyz, zs, z_far, xyz_fw, xyz_bw):
    _check(load().nsff_warp_points(_ptr(raw), _ptr(xyz), _ptr(zs), zs.numel(), float(z_far),
                                   _ptr(xyz_fw), _ptr(xyz_bw), _stream()), "nsff_warp_points")


def frustum_args(w2c, ts, K4, n_cams, n_frames, H, W):
    """NsffFrustumArgs: w2c (n_cams * n_frames, 12) fp32 and ts (>= 1,) int64 on the device (the frame is ts[0])."""
    assert w2c.is_cuda and w2c.dtype == torch.float32 and w2c.is_contiguous() and w2c.shape == (n_cams * n_frames, 12)
    assert ts.is_cuda and ts.dtype == torch.int64 and ts.is_contiguous() and ts.numel() >= 1
    a = FrustumArgs(w2c=w2c.data_ptr(), ts=ts.data_ptr(), n_cams=int(n_cams), n_frames=int(n_frames), H=int(H), W=int(W))
    a.K4[:] = [float(v) for v in K4]
    a._keep = (w2c, ts)                      # the structure only holds addresses
    return a


def frustum_visibility(vis, xyz, out):
    _check(load().nsff_frustum_visibility(C.byref(vis), _ptr(xyz), xyz.shape[0], _ptr(out), _stream()),
           "nsff_frustum_visibility")


def composite(vis=None, **kw):
    a = CompositeArgs()
    for k, v in kw.items():
        if k in _COMPOSITE_PTRS_IN or k in _COMPOSITE_PTRS_OUT:
            setattr(a, k, _ptr(v))
        else:
            setattr(a, k, v)
    if vis is not None:
        a.vis = vis
    _check(load().nsff_composite(C.byref(a), _stream()), "nsff_composite")


def frame_rays(K4, c2w12, H, W, near, shift_near, first, count, rays):
    k = (C.c_float * 4)(*[float(v) for v in K4])
    m = (C.c_float * 12)(*[float(v) for v in c2w12])
    _check(load().nsff_frame_rays(k, m, int(H), int(W), float(near), float(shift_near), int(first), int(count),
                                  _ptr(rays), _stream()), "nsff_frame_rays")


BWD_PACK = 100          # PackCache key of the transposed fp16 weight pack (nsff_pack_weights_bwd)


def bwd_packed_bytes(desc):
    n = C.c_size_t(0)
    _check(load().nsff_bwd_packed_bytes(C.byref(desc), C.byref(n)), "nsff_bwd_packed_bytes")
    return n.value


def pack_weights_bwd(desc, params, packed, fwd_packed=None):
    """fwd_packed: the f16x3 forward pack of the same weights (folded heads): its fp32 scratch supplies the folded products."""
    arr = (_fp * len(params))(*[p.data_ptr() for p in params])
    _check(load().nsff_pack_weights_bwd_ex(C.byref(desc), arr, _ptr(packed), _ptr(fwd_packed), _stream()), "nsff_pack_weights_bwd_ex")


def last_bwd_kernel():
    """'h3b' (the hand-scheduled body), 'c' (compiler-scheduled) or 'c+h3b' (a view-direction model's both-trunk launch: static trunk
    on the compiler-scheduled kernel, dynamic trunk on the hand-scheduled one): which kernel(s) the last field_backward launch took"""
    return {0: "c", 1: "h3b", 2: "c+h3b", 3: "x3"}[load().nsff_last_bwd_kernel()]      # x3: the three-product kernel (config.set_grad_precision)


def last_bwd_grid():
    """workgroups of the last hand-scheduled data-gradient launch (= compute units for a persistent one; 0: none)"""
    return load().nsff_last_bwd_grid()


def field_bwd_phase_program(model, dynamic, want_xin, n_tiles, max_phases=32):
    """(descriptors (n, 8) uint32 numpy, segment byte offsets) of the hand-scheduled backward body for one trunk, or (None, offsets)
    when it does not cover the trunk (host-only: no GPU needed)."""
    import numpy as np
    desc = model_desc(model)
    out = (C.c_uint32 * (8 * max_phases))()
    segs = (C.c_uint32 * 48)()
    n = load().nsff_field_bwd_phase_program(C.byref(desc), 1 if dynamic else 0, 1 if want_xin else 0, int(n_tiles), out, max_phases, segs, 48)
    if n < 0:
        raise RuntimeError(f"nsff_field_bwd_phase_program failed ({n})")
    offs = np.array(list(segs), np.uint32)
    return (np.array(list(out), np.uint32).reshape(max_phases, 8)[:n] if n > 0 else None), offs


def field_backward(model, n_points, static, transient, d_raw, raw, gmax, masks, dpre, dhead, d_xin, d_side=None, x3=False):
    desc = model_desc(model)
    a = FieldBwdArgs(n_points=int(n_points), static_mode=2 if static else 0, transient_mode=2 if transient else 0,
                     d_raw=_ptr(d_raw), raw=_ptr(raw), gmax=_ptr(gmax), masks=masks.data_ptr(), dpre=dpre.data_ptr(),
                     dhead=dhead.data_ptr(), d_xin=_ptr(d_xin), d_side=_ptr(d_side), dpre_lo_delta=lo_delta(dpre) if x3 else 0)
    _check(load().nsff_field_backward(C.byref(desc), _ptr(model.packed(BWD_PACK)), C.byref(a), _stream()),
           "nsff_field_backward")


def train_dims(model):
    """(xin_rows, t_row0, side_rows) of the training buffers of `model` (include/nsff_render.h: nsff_train_dims)."""
    desc = model_desc(model)
    xr, t0, sr = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _check(load().nsff_train_dims(C.byref(desc), C.byref(xr), C.byref(t0), C.byref(sr)), "nsff_train_dims")
    return xr.value, t0.value, sr.value


def field_input_backward(d_xin, t_row0, xyz, pts_per_ray, freqs, in_t, want_xyz, want_t):
    """(d_xyz (P,3) or None, d_t (n_rays, in_t) or None) from the (P, xin_rows) trunk-input gradient."""
    P, xin_rows = d_xin.shape
    n_rays = P // pts_per_ray
    d_xyz = torch.empty(P, 3, device=d_xin.device) if want_xyz else None
    d_t = torch.empty(n_rays, in_t, device=d_xin.device) if want_t else None
    f = [float(v) for v in freqs]
    arr = (C.c_float * max(len(f), 1))(*f)
    _check(load().nsff_field_input_backward(_ptr(d_xin), int(xin_rows), int(t_row0), _ptr(xyz), n_rays, int(pts_per_ray), arr,
                                            len(f), int(in_t), _ptr(d_xyz), _ptr(d_t), _stream()), "nsff_field_input_backward")
    return d_xyz, d_t


def _wgrad_jobs(jobs, with_off):
    return (WgradJob * len(jobs))(*[WgradJob(a=j[0], b=j[1], a_rows=j[2], b_rows=j[3], out_off=j[4] if with_off else 0, trunk=j[5],
                                             a_lo_delta=j[6] if len(j) > 6 else 0, b_lo_delta=j[7] if len(j) > 7 else 0) for j in jobs])


def weight_grad(jobs, n_tiles, n_splits, out, bias, gmax):
    """jobs: list of (a_ptr, b_ptr, a_rows, b_rows, out_off, trunk[, a_lo_delta, b_lo_delta]); out / bias receive the final (summed,
    unscaled) gradients; gmax: the 16 column maxima of the raw-record gradient (absmax)."""
    arr = _wgrad_jobs(jobs, True)
    n = load().nsff_weight_grad_scratch(arr, len(jobs), int(n_tiles), int(n_splits))
    if n < 0:
        raise RuntimeError("nsff_weight_grad_scratch failed")
    scratch = torch.empty(n, device=out.device)
    _check(load().nsff_weight_grad(arr, len(jobs), int(n_tiles), int(n_splits), _ptr(scratch), _ptr(out), _ptr(bias),
                                   _ptr(gmax), _stream()), "nsff_weight_grad")


def weight_grad_accumulate(jobs, n_tiles, n_splits, grad_map, grad_base_ptr, gmax, aux=None):
    """The same GEMMs, accumulated straight into the parameters' gradient memory.  grad_map: (n,4) int32 device tensor of
    NsffGradMapEntry rows; grad_base_ptr: device address the map's `dst` offsets count from; aux: fp32 tensor that receives
    the entries with dst < 0 (dense sums for the folded parameters)."""
    arr = _wgrad_jobs(jobs, False)
    n = load().nsff_weight_grad_scratch(arr, len(jobs), int(n_tiles), int(n_splits))
    if n < 0:
        raise RuntimeError("nsff_weight_grad_scratch failed")
    assert grad_map.dtype == torch.int32 and grad_map.is_contiguous() and grad_map.shape[1] == 4
    scratch = torch.empty(n, device=gmax.device)
    _check(load().nsff_weight_grad_accumulate_aux(arr, len(jobs), int(n_tiles), int(n_splits), _ptr(scratch),
                                                  C.c_void_p(grad_map.data_ptr()), grad_map.shape[0], C.c_void_p(grad_base_ptr),
                                                  _ptr(aux), _ptr(gmax), _stream()), "nsff_weight_grad_accumulate_aux")
    return scratch


def fold_grads(g, gb, w_final, b_final, head_rows, d_w_final, d_b_final):
    """nsff_fold_grads: head_rows = [(w_row (256,) tensor, d_w_row (256,) view of .grad, d_b (1,) view of .grad)] per folded row;
    every tensor fp32 contiguous on the GPU; the d_* are accumulated into."""
    a = FoldGradArgs(n_rows=len(head_rows), g=_ptr(g), gb=_ptr(gb), w_final=_ptr(w_final), b_final=_ptr(b_final),
                     d_w_final=_ptr(d_w_final), d_b_final=_ptr(d_b_final))
    for r, (w, dw, db) in enumerate(head_rows):
        a.w_head[r], a.d_w_head[r], a.d_b_head[r] = w.data_ptr(), dw.data_ptr(), db.data_ptr()
    _check(load().nsff_fold_grads(C.byref(a), _stream()), "nsff_fold_grads")


def fold_grads_dense(g, gb, w_head, w_final, b_final, d_w_head, d_b_head, d_w_final, d_b_final, accumulate, g2=None, gb2=None):
    """nsff_fold_grads_dense: g (R, 256) [+ g2], gb (R,) [+ gb2]: dense sum / row sums of the folded layer's job; w_head (R, >= 256)
    a (possibly column-sliced) view of the folded layer's weight -- its row stride is passed on --, d_w_head likewise; w_final
    (256, 256), b_final (256,).  accumulate: add to / store into the d_* tensors (fp32 GPU tensors, rows contiguous)."""
    R = int(g.shape[0])
    for t in (g, g2, gb, gb2, w_final, b_final, d_w_final, d_b_final, d_b_head):
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
    for t in (w_head, d_w_head):
        assert t.is_cuda and t.dtype == torch.float32 and t.shape[0] == R and t.shape[1] == 256 and t.stride(1) == 1
    a = FoldDenseArgs(n_rows=R, accumulate=1 if accumulate else 0, ld_head=int(w_head.stride(0)), ld_dhead=int(d_w_head.stride(0)),
                      g=g.data_ptr(), g2=None if g2 is None else g2.data_ptr(), gb=gb.data_ptr(), gb2=None if gb2 is None else gb2.data_ptr(),
                      w_head=w_head.data_ptr(), w_final=w_final.data_ptr(), b_final=b_final.data_ptr(), d_w_head=d_w_head.data_ptr(),
                      d_b_head=d_b_head.data_ptr(), d_w_final=d_w_final.data_ptr(), d_b_final=d_b_final.data_ptr())
    _check(load().nsff_fold_grads_dense(C.byref(a), _stream()), "nsff_fold_grads_dense")


def absmax(x):
    """max |x| as a device scalar (one launch, no host round trip).  For a raw-record gradient (P, RAW_STRIDE): the 16 COLUMN
    maxima -- what field_backward / weight_grad* derive their scales from (one per trunk for the fragments, one per head row)."""
    x = x.contiguous()
    if x.dim() == 2 and x.shape[1] == RAW_STRIDE:
        out = torch.empty(RAW_STRIDE, device=x.device, dtype=torch.float32)
        _check(load().nsff_absmax_raw(_ptr(x), x.shape[0], _ptr(out), _stream()), "nsff_absmax_raw")
        return out
    out = torch.empty((), device=x.device, dtype=torch.float32)
    _check(load().nsff_absmax(_ptr(x), x.numel(), _ptr(out), _stream()), "nsff_absmax")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, state, lr, beta1, beta2, eps, weight_decay, seg_start=None, seg_used=None):
    """One Adam step on flat buffers (include/nsff_render.h: nsff_adam_step); state / lr are device tensors.  With
    ``seg_start`` (int64, n_seg + 1 offsets) and ``seg_used`` (int32, n_seg): parameter tensors whose gradient slice is
    identically zero this step are skipped as torch.optim.Adam skips ``grad is None`` (nsff_adam_step_segments)."""
    if seg_start is None:
        _check(load().nsff_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), _ptr(state), _ptr(lr),
                                     float(beta1), float(beta2), float(eps), float(weight_decay), _stream()), "nsff_adam_step")
        return
    _check(load().nsff_adam_step_segments(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), _ptr(state),
                                          _ptr(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                          C.c_void_p(seg_start.data_ptr()), int(seg_used.numel()),
                                          C.c_void_p(seg_used.data_ptr()), _stream()),
           "nsff_adam_step_segments")


def composite_backward(n_rays, n_samples, has_transient, flow_mode, noise_std, **tensors):
    a = CompositeBwdArgs(n_rays=int(n_rays), n_samples=int(n_samples), has_transient=int(has_transient),
                         flow_mode=int(flow_mode), noise_std=float(noise_std))
    for k, v in tensors.items():
        setattr(a, k, _ptr(v))
    _check(load().nsff_composite_backward(C.byref(a), _stream()), "nsff_composite_backward")


def flow_grad(zs, z_far, out, accumulate, col_a=-1, g_a=(), col_b=-1, g_b=()):
    """out (P,16) (+)= masked sums of the (P,3) cotangents g_a into columns col_a.., g_b into col_b.. (nsff_flow_grad)."""
    a = FlowGradArgs(n_points=int(out.shape[0]), zs=_ptr(zs), z_far=float(z_far), accumulate=int(bool(accumulate)),
                     col_a=int(col_a), col_b=int(col_b), out=_ptr(out))
    if len(g_a) > 4 or len(g_b) > 4:
        raise ValueError("flow_grad: at most four cotangents per group")
    keep = [g.contiguous() for g in list(g_a) + list(g_b)]          # (alive until the launch is enqueued)
    for i, g in enumerate(keep[:len(g_a)]):
        a.g_a[i] = _ptr(g)
    for i, g in enumerate(keep[len(g_a):]):
        a.g_b[i] = _ptr(g)
    _check(load().nsff_flow_grad(C.byref(a), _stream()), "nsff_flow_grad")


def nerfw_loss(mode, n_rays, n_samples, n_keep, n_frames, max_t, topk=1.0, thickness=1, **tensors):
    """mode 1: the terms into tensors['terms']; mode 2: gradients into tensors['g_*'] (include/nsff_render.h)."""
    a = LossArgs(n_rays=int(n_rays), n_samples=int(n_samples), n_keep=int(n_keep), n_frames=int(n_frames), max_t=int(max_t),
                 topk=float(topk), thickness=int(thickness))
    for k, v in tensors.items():
        if v is None:
            continue
        if not torch.is_tensor(v):
            raise TypeError(f"nerfw_loss: {k} must be a tensor")
        if v.dtype == torch.int64:
            assert v.is_cuda and v.is_contiguous()
            setattr(a, k, v.data_ptr())
        else:
            setattr(a, k, _ptr(v))
    _check(load().nsff_nerfw_loss(C.byref(a), int(mode), _stream()), "nsff_nerfw_loss")


def splat_work_bytes(H, W, S):
    return int(load().nsff_splat_work_bytes(int(H), int(W), int(S)))


def splat_planes(H, W, S, K4, P12, scale, xyz, flow, rgb, alpha, accum, work=None):
    """work: uint8 device tensor of splat_work_bytes(H, W, S) bytes (binned far path) or None (device-scope atomics)."""
    a = SplatArgs(H=int(H), W=int(W), n_planes=int(S), scale=float(scale), xyz=_ptr(xyz), flow=_ptr(flow),
                  rgb=_ptr(rgb), alpha=_ptr(alpha), accum=_ptr(accum))
    if work is not None:
        assert work.is_cuda and work.is_contiguous() and work.dtype == torch.uint8
        a.work, a.work_bytes = work.data_ptr(), work.numel()
    a.K4[:] = [float(v) for v in K4]
    a.P[:] = [float(v) for v in P12]
    _check(load().nsff_splat_planes(C.byref(a), _stream()), "nsff_splat_planes")


def mpi_composite(H, W, S, dt, accum_fw, accum_bw, static_rgb, static_alpha, zs, rgb, depth):
    a = MpiArgs(H=int(H), W=int(W), n_planes=int(S), dt=float(dt), accum_fw=_ptr(accum_fw), accum_bw=_ptr(accum_bw),
                static_rgb=_ptr(static_rgb), static_alpha=_ptr(static_alpha), zs=_ptr(zs), rgb=_ptr(rgb),
                depth=_ptr(depth))
    _check(load().nsff_mpi_composite(C.byref(a), _stream()), "nsff_mpi_composite")


KERNEL_NAMES = {0: None, 1: "f32", 2: "h3_64", 3: "h3_8wave", 4: "h3a", 5: "h3_save", 7: "h3a_tb", 8: "h3a_side", 9: "h3a_save"}


def last_field_kernel():
    """name of the kernel the last field_query of this process launched (include/nsff_render.h: NSFF_KERNEL_*)"""
    return KERNEL_NAMES[load().nsff_last_field_kernel()]


def last_field_grid():
    """Workgroups of the last field launch when it ran the hand-scheduled inference kernel (0 otherwise): the CU count for a
    persistent launch, one (two: both trunks) per 128-point tile otherwise (include/nsff_render.h::nsff_last_field_grid)."""
    return int(load().nsff_last_field_grid())


def h3a_program(model, static_mode, transient_mode, fold_t=False, side_fold=False, persist=False):
    """(steps, n_static_steps, phases_static, phases_dynamic) of an f16x3 inference launch, from the host-side builders alone
    (no GPU): steps = [(w_off_words, bias_off_words or None, nks, pre, post, head)], phases_* = [[8 dwords]] or [].
    fold_t: the dynamic trunk's program of a launch that is given t_bias (time code folded into per-ray bias rows);
    side_fold: the static trunk's program of a view-direction launch that is given s_bias ([dir | a] folded into per-ray rows);
    persist: the programs of a persistent launch (the last segment's B phase requests the next tile's first weight slots; [] for a
    trunk that does not end with a 256-wide segment)."""
    desc = model_desc(model)
    steps = (C.c_uint32 * (28 * 4))()
    ps, pd = (C.c_uint32 * (36 * 8))(), (C.c_uint32 * (36 * 8))()
    n, ns, nph = C.c_int(0), C.c_int(0), (C.c_int * 2)()
    _check(load().nsff_field_phase_program(C.byref(desc), int(static_mode), int(transient_mode), int(bool(fold_t)) | (2 if side_fold else 0) | (4 if persist else 0), steps, C.byref(n), C.byref(ns), ps, pd, nph),
           "nsff_field_phase_program")
    out = []
    for i in range(n.value):
        w, b, packed = steps[4 * i], steps[4 * i + 1], steps[4 * i + 2]
        out.append((w, None if b == 0xFFFFFFFF else b, packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF, packed >> 24))
    rows = lambda arr, k: [[int(arr[8 * i + j]) for j in range(8)] for i in range(k)]
    return out, ns.value, rows(ps, nph[0]), rows(pd, nph[1])


def prof_enable(on):
    _check(load().nsff_prof_enable(int(bool(on))), "nsff_prof_enable")


def prof_collect():
    """(launches, ms, algorithmic flops, executed flops, shader clock in GHz or None) of the field launches since prof_enable."""
    n, ms, fl, ex, tk, tms = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    _check(load().nsff_prof_collect_clock(C.byref(n), C.byref(ms), C.byref(fl), C.byref(ex), C.byref(tk), C.byref(tms)),
           "nsff_prof_collect_clock")
    ghz = tk.value / tms.value * 1e-6 if tms.value > 0 else None
    if ghz is not None and not (0.5 < ghz < 3.0):       # counters of different XCDs disagreeing, a wrapped stamp: no claim
        ghz = None
    return n.value, ms.value, fl.value, ex.value, ghz
