"""Objective of the NSFF trainer, consuming the ``render_rays`` result dict.

A restatement of the reference's ``losses.py`` (``shiftscale_invariant_depthloss`` :8-28, ``NeRFWLoss``
:31-171) with the same constructor, term names, weights and reductions.  On the GPU the NSFF train configuration
is evaluated -- forward and backward -- by the fused kernels of ``csrc/loss.hip`` (:mod:`nsff_pl_amd.fused_loss`);
the torch expression below is the general form (top-k mining, per-ray weights, dilated cross entropy, static-only
models, CPU tensors) and what the kernels are tested against.  ``kornia.filter2d`` (1 x thickness box filter, zero padded) is replaced by ``conv1d``.
``Ks`` (n_cam,3,3), ``Ps`` (n_cam,N_frames,3,4) and ``max_t`` are attached by the trainer exactly like
``train.py:136-138`` does.
"""
import torch
import torch.nn.functional as F
from torch import nn


def shiftscale_invariant_depthloss(depth, disp):
    """(N,) NDC depth vs (N,) image-based disparity, both median/MAD normalised (losses.py:8-28)."""
    def normalise(x):
        shift = torch.median(x)
        return (x - shift) / torch.mean(torch.abs(x - shift))
    return (normalise(depth) - normalise(-disp)) ** 2


def ndc2world(xyz, K, eps=1e-6):
    """NDC -> world for (N,3) points with one K (3,3), or (N,M,3) points with per-ray K (N,3,3)
    (datasets/ray_utils.py:127-151)."""
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    wz = 2 / (xyz[..., 2] - 1 - eps)
    if xyz.dim() == 3:
        wx = -wz * xyz[..., 0] * (cx / fx)[:, None]
        wy = -wz * xyz[..., 1] * (cy / fy)[:, None]
    else:
        wx = -wz * xyz[..., 0] * cx / fx
        wy = -wz * xyz[..., 1] * cy / fy
    return torch.stack([wx, wy, wz], -1)


class NeRFWLoss(nn.Module):
    """col_l, disp_l, entropy_l, cross_entropy_l, flow_fw_l / flow_bw_l, pho_l, cyc_l, reg_temp_sm_l,
    reg_min_l, reg_sp_sm_l -- each reduced to a scalar mean (after optional per-ray weights / top-k)."""

    def __init__(self, lambda_geo=0.04, lambda_reg=0.1, thickness=1, topk=1.0, static_shapes=False):
        """static_shapes: evaluate the two masked flow terms as masked means instead of boolean indexing (same value;
        no data-dependent shapes or host syncs, so the step can be captured in a hipGraph).  Needs topk == 1 and no
        per-ray `weights`."""
        super().__init__()
        self.static_shapes = static_shapes
        self.lambda_geo_d = self.lambda_geo_f = lambda_geo
        self.lambda_reg = lambda_reg
        self.lambda_ent = 1e-3
        self.z_far = 0.95
        self.thickness = max(thickness, 1)
        self.topk = topk

    def _project(self, xyz_ndc, Ks, cam_ids, ts_neighbour):
        world = ndc2world(xyz_ndc, Ks)
        P = self.Ps[cam_ids, ts_neighbour]                                  # (N,3,4)
        uvd = (P[:, :3, :3] @ world[..., None])[..., 0] + P[:, :3, 3]
        return uvd[:, :2] / (torch.abs(uvd[:, 2:]) + 1e-8), uvd[:, 2]

    def forward(self, inputs, targets, **kwargs):
        from . import fused_loss
        if fused_loss.applicable(self, inputs, targets, kwargs):      # NSFF train configuration on the GPU: csrc/loss.hip
            return fused_loss.nerfw_loss(self, inputs, targets, kwargs)
        ret, population = {}, {}            # population: the rays a masked (flow) term's values belong to
        rgbs = targets['rgbs']
        ret['col_l'] = ((inputs['rgb_fine'] - rgbs) ** 2).mean(1)
        if 'rgb_coarse' in inputs:
            ret['col_l'] = ret['col_l'] + 0.1 * ((inputs['rgb_coarse'] - rgbs) ** 2).mean(1)
        ret['disp_l'] = self.lambda_geo_d * shiftscale_invariant_depthloss(inputs['depth_fine'], targets['disps'])
        if 'depth_coarse' in inputs:
            ret['disp_l'] = ret['disp_l'] + self.lambda_geo_d * \
                shiftscale_invariant_depthloss(inputs['depth_coarse'], targets['disps'])

        if kwargs['output_transient_flow']:
            t_w, s_w = inputs['transient_weights_fine'], inputs['static_weights_fine']
            ret['entropy_l'] = self.lambda_ent * (-t_w * torch.log(t_w + 1e-8)).sum(1)
            # static weights are pushed away from where the (dilated) dynamic weights peak; ramps up over 10 epochs
            cross_w = self.lambda_ent / 5 * kwargs.get('epoch_ramp', min(kwargs['epoch'] / 10, 1.0))
            dil = t_w.detach()
            if self.thickness > 1:
                box = torch.ones(1, 1, self.thickness, device=dil.device, dtype=dil.dtype)
                pad = (self.thickness - 1) // 2
                dil = F.conv1d(F.pad(dil[:, None], (pad, self.thickness - 1 - pad)), box)[:, 0]
            ret['cross_entropy_l'] = cross_w * (dil * torch.log(s_w + 1e-8)).sum(1)

            cam_ids, ts = targets['cam_ids'], targets['ts']
            Ks = self.Ks[cam_ids]
            Ks = Ks if Ks.dim() == 3 else Ks[None].expand(len(ts), 3, 3)
            # 2D-3D flow consistency: expected scene-flow end points projected into the neighbouring frames
            uv_fw, d_fw = self._project(inputs['xyz_fw'], Ks, cam_ids, torch.clamp(ts + 1, max=self.max_t))
            uv_bw, d_bw = self._project(inputs['xyz_bw'], Ks, cam_ids, torch.clamp(ts - 1, min=0))
            ok_fw = (d_fw > 0) & (ts < self.max_t)
            ok_bw = (d_bw > 0) & (ts > 0)
            scalars = {}
            if self.static_shapes:
                assert self.topk >= 1 and 'weights' not in kwargs, "static_shapes needs topk == 1 and no ray weights"
                for key, ok, uv, tgt in (('flow_fw_l', ok_fw, uv_fw, targets['uv_fw']), ('flow_bw_l', ok_bw, uv_bw, targets['uv_bw'])):
                    err = torch.where(ok[:, None], torch.abs(uv - tgt), torch.zeros_like(uv))
                    scalars[key] = self.lambda_geo_f / 2 * err.sum() / (2 * ok.sum().clamp_min(1))
            elif ok_fw.any():
                ret['flow_fw_l'] = (self.lambda_geo_f / 2 * torch.abs(uv_fw[ok_fw] - targets['uv_fw'][ok_fw])).mean(1)
                population['flow_fw_l'] = ok_fw
            if not self.static_shapes and ok_bw.any():
                ret['flow_bw_l'] = (self.lambda_geo_f / 2 * torch.abs(uv_bw[ok_bw] - targets['uv_bw'][ok_bw])).mean(1)
                population['flow_bw_l'] = ok_bw

            # photometric + cycle consistency of the warped renders, weighted by disocclusion
            pho = inputs['disocc_fw'] * (inputs['rgb_fw'] - rgbs) ** 2 / inputs['disocc_fw'].mean() + \
                inputs['disocc_bw'] * (inputs['rgb_bw'] - rgbs) ** 2 / inputs['disocc_bw'].mean()
            ret['pho_l'] = pho.mean(1)
            cyc = inputs['disoccs_fw'] * torch.abs(inputs['xyzs_fw_bw'] - inputs['xyzs_fine']) / inputs['disoccs_fw'].mean() + \
                inputs['disoccs_bw'] * torch.abs(inputs['xyzs_bw_fw'] - inputs['xyzs_fine']) / inputs['disoccs_bw'].mean()
            ret['cyc_l'] = cyc.mean((1, 2))

            # scene-flow regularisers in world space, near samples only
            n_keep = int(inputs['xyzs_fine'].shape[1] * self.z_far)
            p0 = ndc2world(inputs['xyzs_fine'][:, :n_keep], Ks)
            pf = ndc2world(inputs['xyzs_fw'][:, :n_keep], Ks)
            pb = ndc2world(inputs['xyzs_bw'][:, :n_keep], Ks)
            ret['reg_temp_sm_l'] = (self.lambda_reg * torch.abs(pf + pb - 2 * p0)).mean((1, 2))
            ret['reg_min_l'] = (self.lambda_reg * (torch.abs(pf - p0) + torch.abs(pb - p0))).mean((1, 2))
            near = torch.exp(-2 * torch.norm(p0[:, 1:] - p0[:, :-1], dim=-1, keepdim=True))
            sf_f, sf_b = pf - p0, pb - p0
            ret['reg_sp_sm_l'] = (self.lambda_reg * (torch.abs(sf_f[:, 1:] - sf_f[:, :-1]) * near +
                                                     torch.abs(sf_b[:, 1:] - sf_b[:, :-1]) * near)).mean((1, 2))

        if kwargs['output_transient_flow'] and self.static_shapes:
            ret.update(scalars)             # already reduced (mean of a 0-d tensor is itself)
        for k, loss in ret.items():
            if 'weights' in kwargs:
                # (the reference multiplies every term by the (N,) weights, losses.py:163-164 -- which only broadcasts for the
                #  masked flow terms when every ray is valid; here a masked term takes the weights of its own rays)
                w = kwargs['weights']
                loss = loss * (w[population[k]] if (k in population and torch.is_tensor(w) and w.dim() > 0) else w)
            if self.topk < 1:
                loss, _ = torch.topk(loss.flatten(), int(self.topk * loss.numel()))
            ret[k] = loss.mean()
        return ret
