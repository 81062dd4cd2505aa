"""Eval-only frustum-visibility mask of render_rays (reference rendering.py:190-200).

At test time the reference suppresses the dynamic field at sample points that no training camera of the current
frame can see: NDC points go back to world space (``datasets/ray_utils.py:127-151``), are projected into each
training camera of frame ``ts[0]`` (``ray_utils.py:154-181``) and points seen by none get raw transient sigma -10.

Here that is arithmetic INSIDE the compositing kernel (``NsffCompositeArgs.vis``, csrc/rays.hip ``frustum_count``): the
sample point is already in registers there, so the mask costs no memory traffic and no launch.  What this module does
is host-side preparation, once per dataset: the world-to-camera matrices of ALL training poses (a batched 4x4
inverse, what ``compute_world_visiblility`` recomputes per call) are uploaded as one (n_cams * n_frames, 12) table; the
kernel picks row ``i * n_frames + ts[0]`` with ``ts[0]`` read on the device -- the call never synchronises with the host.
"""
import numpy as np
import torch

from . import _lib

_TABLES = {}          # (device, pose bytes) -> (n_poses, 12) fp32 device tensor


def world_to_camera_table(poses, device):
    """(n_poses, 12) rows of inverse([c2w; 0 0 0 1])[:3] for every (3,4) pose, on `device`; cached by content."""
    p = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 3, 4))
    key = (str(device), p.tobytes())
    hit = _TABLES.get(key)
    if hit is None:
        pose4 = torch.eye(4).repeat(p.shape[0], 1, 1)
        pose4[:, :3] = torch.from_numpy(p)
        hit = torch.linalg.inv(pose4)[:, :3].reshape(-1, 12).contiguous().to(device)
        if len(_TABLES) > 16:
            _TABLES.clear()
        _TABLES[key] = hit
    return hit


def frustum_args(dataset, ts, device):
    """The ``NsffFrustumArgs`` of one render_rays call: attributes read from ``kwargs['dataset']`` exactly as the reference
    does (``Ks[0]``, ``img_wh``, ``cam_train``, ``N_frames``, ``poses``; rendering.py:192-199)."""
    K = torch.as_tensor(dataset.Ks[0], dtype=torch.float32).cpu()
    n_cams, n_frames = len(dataset.cam_train), int(dataset.N_frames)
    table = world_to_camera_table(np.asarray(dataset.poses)[:n_cams * n_frames], device)
    ts = ts.to(device=device, dtype=torch.int64).contiguous()
    return _lib.frustum_args(table, ts, [K[0, 0], K[1, 1], K[0, 2], K[1, 2]], n_cams, n_frames,
                             dataset.img_wh[1], dataset.img_wh[0])


def training_view_visibility(xyz_ndc, dataset, ts):
    """(P,) number of training cameras of frame ts[0] that see each NDC sample point -- the reference's
    ``visibilities`` as a tensor (the stage on its own: ``nsff_frustum_visibility``; render_rays itself never
    materialises it)."""
    xyz = xyz_ndc.contiguous().float()
    out = torch.empty(xyz.shape[0], device=xyz.device)
    if xyz.shape[0]:
        with torch.cuda.device(xyz.device):
            _lib.frustum_visibility(frustum_args(dataset, ts, xyz.device), xyz, out)
    return out
