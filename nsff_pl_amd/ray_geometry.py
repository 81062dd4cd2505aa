"""Eval-only frustum-visibility mask of render_rays (reference rendering.py:190-200).

At test time the reference suppresses the dynamic field at sample points that no
training camera of the current frame can see: NDC points go back to world space
(``datasets/ray_utils.py:127-151``), are projected into each training camera
(``ray_utils.py:154-181``) and points seen by none get raw transient sigma -10.
This is one camera, a 4x4 inverse and a few elementwise ops per call, so it stays in
torch ops on the device; the mask is handed to the
compositing kernel, which applies the -10 override.
"""
import torch


def ndc_to_world(xyz, K, eps=1e-6):
    """(P,3) NDC -> (P,3) world for pinhole intrinsics K (ray_utils.py:127-151)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    wz = 2 / (xyz[:, 2] - 1 - eps)
    wx = -wz * xyz[:, 0] * cx / fx
    wy = -wz * xyz[:, 1] * cy / fy
    return torch.stack([wx, wy, wz], 1)


def count_in_frustum(xyz_world, K, H, W, c2w):
    """1.0 where a world point lies in front of and inside the image of camera c2w (3,4)."""
    pose = torch.eye(4, device=xyz_world.device)
    pose[:3] = c2w
    w2c = torch.inverse(pose)
    cam = w2c[:3, :3] @ xyz_world.T + w2c[:3, 3:]          # (3,P), camera looks along -z
    in_front = cam[2] < 0
    cam = torch.stack([cam[0], -cam[1], -cam[2]], 0)       # right-down-front
    img = K @ cam
    u, v = img[0] / img[2], img[1] / img[2]
    inside = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    return (in_front & inside).float()


def training_view_visibility(xyz_ndc, dataset, ts):
    """(P,) number of training cameras of frame ts[0] that see each NDC sample point."""
    K = dataset.Ks[0].to(xyz_ndc.device)
    world = ndc_to_world(xyz_ndc, K)
    vis = torch.zeros(xyz_ndc.shape[0], device=xyz_ndc.device)
    frame = int(ts[0])
    for i in range(len(dataset.cam_train)):
        c2w = torch.as_tensor(dataset.poses[i * dataset.N_frames + frame], dtype=torch.float32,
                              device=xyz_ndc.device)
        vis += count_in_frustum(world, K, dataset.img_wh[1], dataset.img_wh[0], c2w)
    return vis.contiguous()
