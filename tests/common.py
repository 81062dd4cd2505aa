"""Fixture loading and oracle invocation shared by CPU and GPU tests."""
import contextlib
import json
import os

import numpy as np
import torch

import scenes
from oracle import nsff_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# keys produced by a second, chained field query at positions that already carry fp32
# rounding (x + flow, then sin(512 x)): two correct fp32 paths differ more there.
CHAINED_KEYS = ("xyzs_fw_bw", "xyzs_bw_fw", "rgb_fw", "rgb_bw", "disocc_fw", "disocc_bw",
                "disoccs_fw", "disoccs_bw")
SAMPLE_KEYS = ("static_zs_fine", "transient_zs_fine", "zs_fine", "xyzs_fine")


def render_rays_at(zs_fine):
    """``nsff_pl_amd.render_rays`` with the fine pass of ONE call evaluated at the given (N_rays, S_fine) depths (numpy / tensor;
    None = the plain function, free-running).  sample_pdf is ill-conditioned in near-empty bins (tests/parity.py), so per-sample
    fine keys of two correct fp32 implementations are only comparable at identical depths.  The substitution (and the point
    arithmetic o + d z that goes with it) lives here: the product's body only offers the injection point between its fine
    sampling stage and its fine field pass (rendering._render_rays(fine_points=)); nothing is patched, no state outlives the call."""
    import nsff_pl_amd.rendering as R
    if zs_fine is None:
        return R.render_rays

    def at_depths(rays, zs_sampled, xyz_sampled):
        zs = torch.as_tensor(zs_fine).to(rays.device).contiguous().float()
        return zs, (rays[:, None, 0:3] + rays[:, None, 3:6] * zs[..., None]).contiguous()

    def call(models, embeddings, rays, ts, max_t, N_samples=64, perturb=0, noise_std=0, N_importance=0, chunk=1024 * 32,
             test_time=False, **kwargs):
        return R._render_rays(models, embeddings, rays, ts, max_t, N_samples, perturb, noise_std, N_importance, chunk, test_time,
                              kwargs, fine_points=at_depths)
    return call


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    out = {k[4:]: z[k] for k in z.files if k.startswith("out/")}
    return meta, out


def build_case(name, nerf_cls, posemb_cls):
    """Scene of a golden case built with the given classes, weights verified by checksum."""
    meta, want = load_golden(name)
    cfg, rays, ts = scenes.case_inputs(name)
    models, emb = scenes.build_scene(nerf_cls, posemb_cls, cfg)
    chk = scenes.weight_checksum(models, emb)
    assert abs(chk - meta["weight_checksum"]) <= 1e-9 * abs(chk), \
        "seeded weights differ from the ones the golden was generated with (torch init/RNG drift)"
    dataset = scenes.DatasetStub(cfg["seed"]) if cfg.get("dataset") else None
    return cfg, meta, rays, ts, models, emb, dataset, want


def oracle_render(cfg, models, emb, rays, ts, draws=None, dataset=None, zs_fine_override=None):
    fields = {k: orc.field_from_module(m) for k, m in models.items()}
    return orc.render_rays(
        fields, emb["xyz"].freqs.numpy(), emb["dir"].freqs.numpy(), np.asarray(rays),
        None if ts is None else np.asarray(ts), scenes.N_FRAMES - 1,
        emb_t=emb["t"].weight.detach().cpu().numpy() if "t" in emb else None,
        emb_a=emb["a"].weight.detach().cpu().numpy() if "a" in emb else None,
        N_samples=cfg["N_samples"], perturb=cfg.get("perturb", 0), noise_std=cfg.get("noise_std", 0),
        N_importance=cfg["N_importance"], test_time=cfg["test_time"],
        z_lin=torch.linspace(0, 1, cfg["N_samples"]).numpy(),
        u_lin=torch.linspace(0, 1, max(cfg["N_importance"], 1)).numpy(), draws=draws,
        output_transient_flow=cfg["flow"], dataset=dataset.as_oracle_dict() if dataset else None,
        zs_fine_override=zs_fine_override)


# The two keys that cannot hold 1e-4 on two scenes, with the error MEASURED on the MI355X (tools/debug/chained_key_errors.py:
# worst over fp32 / f16x3, every kernel, saving and inference launches, at the reference's depths; gpurun_out/r04_chained.txt) and
# the bound asserted = 2 x that.  They are the cycle points x + fw(x) + bw(x + fw(x)): a second MLP pass at a position that already
# carries fp32 rounding, through sin(2^9 x) (2^11 on g18).  Every other chained key (rgb_fw / rgb_bw, disocc*, ...) holds 1e-4 on
# every scene, and these two hold it on every other scene (6.3e-5 on g3, < 5e-5 on the C2 subset g19).
MEASURED_ABOVE_1E4 = {
    ("gain3", "xyzs_fw_bw"): 6.71e-4, ("gain3", "xyzs_bw_fw"): 6.66e-4,          # g3b: weights x 3, sigma up to 33
    ("emb11", "xyzs_bw_fw"): 1.45e-4, ("emb11", "xyzs_fw_bw"): 1.05e-4,          # g18: 12-frequency embedding (2^11 x)
}


def key_rtol(key, cfg):
    """1e-4 for every key of every scene, except the per-key measured bounds above (2 x the measurement)."""
    import parity
    scene = "gain3" if cfg["gain"] > 2.5 else ("emb11" if cfg.get("xyz_emb", (9, 10))[0] > 9 else None)
    m = MEASURED_ABOVE_1E4.get((scene, key))
    return parity.RTOL if m is None else 2 * m


def fine_sample_tolerances(cfg, coarse, u_s, u_t):
    """Conditioning-aware tolerance of the fine samples given the coarse weights (parity.py)."""
    import parity
    z = torch.linspace(0, 1, cfg["N_samples"]).numpy()
    n = coarse["static_weights_coarse"].shape[0]
    mids = np.broadcast_to(0.5 * (z[:-1] + z[1:]), (n, cfg["N_samples"] - 1))
    tol_s = parity.sample_tolerance(mids, coarse["static_weights_coarse"][:, 1:-1], u_s)
    tol_t = None
    if "transient_weights_coarse" in coarse:
        tol_t = parity.sample_tolerance(mids, coarse["transient_weights_coarse"][:, 1:-1], u_t)
    return tol_s, tol_t


def cpu_flat_adam():
    """A twin of nsff_pl_amd.optim.FlatAdam whose step is written with torch ops (torch.optim.Adam's single-tensor
    formulas, amsgrad off) so that NSFFTrainer's data-parallel step can be driven on CPU by the gloo tests, and the HIP
    step has something to be compared with."""
    import math
    import torch
    from nsff_pl_amd.optim import FlatAdam

    class TorchFlatAdam(FlatAdam):
        @staticmethod
        def _check_device(dev):
            pass

        @torch.no_grad()
        def step(self):
            b1, b2 = self.betas
            p, g, m, v = self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq
            self.state[0] += 1
            t = float(self.state[0])
            live = None
            if self.weight_decay and not self.decay_unused:     # the segment form: tensors with an all-zero gradient are skipped
                live = torch.zeros_like(p, dtype=torch.bool)
                offs = self.seg_start.tolist()
                for a, b in zip(offs[:-1], offs[1:]):
                    live[a:b] = bool((g[a:b] != 0).any())
                keep = (p.clone(), m.clone(), v.clone())
            if self.weight_decay:
                g = g + self.weight_decay * p
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            step_size = float(self.lr) / (1 - b1 ** t)
            denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)
            if live is not None:
                for buf, old in zip((p, m, v), keep):
                    buf.copy_(torch.where(live, buf, old))
    return TorchFlatAdam
