"""Time interpolation between two test-time renders.

``interpolate`` has the signature of the reference's ``models/rendering.py:365-460``.  The reference splats
each of the S sample planes separately (2*S cupy launches, each with a ``.cuda()``/``.cpu()`` round trip,
``rendering.py:439-449``); here the whole frame is two ``nsff_splat_planes`` launches (t forward by dt, t+1
backward by 1-dt) into (pixel, plane, 8) fp32 accumulators -- output blocks owned by workgroups, LDS adds, far samples
binned per destination block (csrc/interp.hip) -- and one ``nsff_mpi_composite`` launch (wavefront per pixel, product scan
over the planes).
"""
import os

import torch

from . import _lib


_FAR_BINNING = os.environ.get("NSFF_SPLAT_BINNING", "1") != "0"      # 0: far samples through device-scope atomics (A/B)


def _dev(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def interpolate(results_t, results_tp1, dt, K, c2w, img_wh):
    """results_t / results_tp1: test-time ``render_rays`` dicts of frame t and t+1 for the SAME rays (pose c2w,
    intrinsics K), rendered with ``output_transient_flow=['fw','bw']``.  dt in (0,1).
    Returns ((h,w,3) rgb, (h,w) NDC depth) on the GPU."""
    xyzs = results_t['xyzs_fine']
    device = xyzs.device if xyzs.is_cuda else (torch.device('cuda') if torch.cuda.is_available() else None)
    if device is None:
        raise RuntimeError("interpolate runs only on the HIP kernels of libnsff_hip.so (no CPU fallback)")
    w, h = img_wh
    n_rays, S = xyzs.shape[:2]
    if n_rays != h * w:
        raise ValueError(f"results hold {n_rays} rays, img_wh={img_wh} needs {h * w}")
    K = torch.as_tensor(K, dtype=torch.float32).cpu().reshape(3, 3)
    pose = torch.eye(4)
    pose[:3] = torch.as_tensor(c2w, dtype=torch.float32).cpu()
    w2c = torch.inverse(pose)[:3]
    w2c[1:] *= -1                                   # "right up back" -> "right down forward" (rendering.py:393)
    P = (K @ w2c).reshape(-1).tolist()
    K4 = [float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])]

    xyz = _dev(xyzs, device)
    accum = torch.empty(2, n_rays, S, 8, device=device)
    # workspace of the binned far path (samples that move more than 4 pixels): one allocation serves both splats, and the
    # caching allocator hands the same block back frame after frame
    work = torch.empty(_lib.splat_work_bytes(h, w, S), device=device, dtype=torch.uint8) if _FAR_BINNING else None
    _lib.splat_planes(h, w, S, K4, P, dt, xyz, _dev(results_t['transient_flows_fw'], device),
                      _dev(results_t['transient_rgbs_fine'], device), _dev(results_t['transient_alphas_fine'], device),
                      accum[0], work)
    _lib.splat_planes(h, w, S, K4, P, 1 - dt, xyz, _dev(results_tp1['transient_flows_bw'], device),
                      _dev(results_tp1['transient_rgbs_fine'], device),
                      _dev(results_tp1['transient_alphas_fine'], device), accum[1], work)
    rgb = torch.empty(h, w, 3, device=device)
    depth = torch.empty(h, w, device=device)
    _lib.mpi_composite(h, w, S, dt, accum[0], accum[1], _dev(results_t['static_rgbs_fine'], device),
                       _dev(results_t['static_alphas_fine'], device), _dev(results_t['zs_fine'], device), rgb, depth)
    return rgb, depth
