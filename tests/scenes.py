"""Seeded synthetic scenes shared by the golden generator and the tests.

The SAME construction code is run with the reference's classes (in
``tests/golden/make_golden.py``, build container only) and with the build's classes (in
the tests), so ``torch.manual_seed`` yields bit-identical weights on both sides and the
fixtures only need to store inputs, expected outputs and a weight checksum.

Synthetic inputs follow SURVEY.md section 8(d): NDC-like rays o=(U(-1,1),U(-1,1),-1),
d=(N(0,.1),N(0,.1),2); ts ~ randint(0,30); max_t=29; N_tau=48; default init with every
``*.weight`` of the NeRFs scaled by ``gain`` ("sharp" init) so outputs are not
near-constant.
"""
import numpy as np
import torch

N_FRAMES = 30
N_TAU = 48
N_A = 48

# name -> config.  'flow' is output_transient_flow; 'draws' marks cases with perturb/noise.
CASES = {
    "g1_static_c1": dict(n_rays=32, N_samples=64, N_importance=0, transient=False, viewdir=False,
                         appearance=False, test_time=False, flow=[], gain=2.5, seed=0),
    "g2_static_c2f": dict(n_rays=16, N_samples=64, N_importance=64, transient=False, viewdir=False,
                          appearance=False, test_time=False, flow=[], gain=2.5, seed=1),
    "g3_nsff_train": dict(n_rays=16, N_samples=64, N_importance=64, transient=True, viewdir=False,
                          appearance=False, test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=2),
    "g3b_nsff_train_gain3": dict(n_rays=8, N_samples=64, N_importance=64, transient=True, viewdir=False,
                                 appearance=False, test_time=False, flow=['fw', 'bw', 'disocc'], gain=3.0, seed=3),
    "g4_nsff_test": dict(n_rays=16, N_samples=64, N_importance=64, transient=True, viewdir=False,
                         appearance=False, test_time=True, flow=[], gain=2.5, seed=4),
    "g5_nsff_test_vis": dict(n_rays=24, N_samples=64, N_importance=64, transient=True, viewdir=False,
                             appearance=False, test_time=True, flow=['fw', 'bw'], gain=2.5, seed=5,
                             dataset=True),
    "g6_readme_viewdir": dict(n_rays=8, N_samples=128, N_importance=0, transient=True, viewdir=True,
                              appearance=True, test_time=True, flow=['fw', 'bw'], gain=2.5, seed=6),
    "g7_nsff_train_noise": dict(n_rays=16, N_samples=64, N_importance=64, transient=True, viewdir=False,
                                appearance=False, test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5,
                                seed=7, perturb=1.0, noise_std=1.0),
    # a different architecture: 6 layers, skip at 2, 16-wide time code, 8 / 3 embedding frequencies (in_xyz = 51)
    "g12_other_arch": dict(n_rays=12, N_samples=32, N_importance=24, transient=True, viewdir=False,
                           appearance=False, test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=12,
                           D=6, skips=[2], n_tau=16, xyz_emb=(7, 8), dir_emb=(2, 3)),
    # view directions + appearance code in TRAIN mode (static_dir_encoding on the gradient path; ctor default use_viewdir=True)
    "g13_viewdir_train": dict(n_rays=16, N_samples=64, N_importance=64, transient=True, viewdir=True, appearance=True,
                              test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=13),
    # the reference's `skips` is a list (nerf.py:34-40,163-167): two skip layers / none at all, at test time ...
    "g14_two_skips": dict(n_rays=10, N_samples=32, N_importance=24, transient=True, viewdir=False, appearance=False,
                          test_time=True, flow=['fw', 'bw'], gain=2.5, seed=14, D=8, skips=[2, 5]),
    "g15_no_skip": dict(n_rays=10, N_samples=48, N_importance=0, transient=True, viewdir=True, appearance=False,
                        test_time=True, flow=[], gain=2.5, seed=15, D=4, skips=[]),
    # ... and in TRAIN mode (gradient goldens g9_*): two skip layers, no skip layer, and the widths the reference's CLI can
    # reach beyond the defaults (opt.py:25 --N_emb_xyz, :45 --N_tau; a 16-frequency direction embedding + appearance code):
    # in_xyz = 75 > 64, in_t = 96 > 64, in_dir + in_a = 147 > 128 -- the saved input tiles of the backward pass get 256 rows
    "g16_two_skips_train": dict(n_rays=12, N_samples=32, N_importance=24, transient=True, viewdir=False, appearance=False,
                                test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=16, D=8, skips=[2, 5]),
    "g17_no_skip_train": dict(n_rays=12, N_samples=32, N_importance=24, transient=True, viewdir=False, appearance=False,
                              test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=17, D=4, skips=[]),
    "g18_wide_inputs_train": dict(n_rays=12, N_samples=32, N_importance=24, transient=True, viewdir=True, appearance=True,
                                  test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=18,
                                  n_tau=96, xyz_emb=(11, 12), dir_emb=(15, 16)),
    # 24 rays OF THE C2 BENCH / FULL-SIZE TEST BATCH (synthetic_rays(1024, 42), the g3 scene's weights): the full-size GPU test
    # compares exactly these rows of its 1024-ray render with the reference's outputs, chained re-query keys included
    "g19_c2_subset": dict(n_rays=24, N_samples=64, N_importance=64, transient=True, viewdir=False,
                          appearance=False, test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=2,
                          batch=(1024, 42, 0)),
    "g7b_static_noise_odd": dict(n_rays=9, N_samples=48, N_importance=40, transient=False, viewdir=True,
                                 appearance=False, test_time=False, flow=[], gain=2.5, seed=8,
                                 perturb=0.5, noise_std=0.7),
}


def synthetic_rays(n_rays, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    o = torch.cat([torch.rand(n_rays, 2, generator=g) * 2 - 1, -torch.ones(n_rays, 1)], 1)
    d = torch.cat([torch.randn(n_rays, 2, generator=g) * 0.1, 2 * torch.ones(n_rays, 1)], 1)
    ts = torch.randint(0, N_FRAMES, (n_rays,), generator=g)
    return torch.cat([o, d], 1).float(), ts


def build_scene(nerf_cls, posemb_cls, cfg):
    """Construct embeddings + models in a fixed order under cfg['seed'] and apply the gain."""
    torch.manual_seed(cfg["seed"])
    xyz_emb, dir_emb = cfg.get("xyz_emb", (9, 10)), cfg.get("dir_emb", (3, 4))
    n_tau = cfg.get("n_tau", N_TAU)
    embeddings = {"xyz": posemb_cls(*xyz_emb), "dir": posemb_cls(*dir_emb)}
    if cfg["transient"]:
        embeddings["t"] = torch.nn.Embedding(N_FRAMES, n_tau)
    if cfg["appearance"]:
        embeddings["a"] = torch.nn.Embedding(N_FRAMES, N_A)
    flow = bool(cfg["flow"])
    arch = dict(D=cfg.get("D", 8), skips=cfg.get("skips", [4]), in_channels_xyz=3 + 6 * xyz_emb[1],
                in_channels_dir=3 + 6 * dir_emb[1])
    models = {"fine": nerf_cls("fine", use_viewdir=cfg["viewdir"],
                               encode_appearance=cfg["appearance"], in_channels_a=N_A,
                               encode_transient=cfg["transient"], in_channels_t=n_tau,
                               output_flow=flow, **arch)}
    if cfg["N_importance"] > 0:
        models["coarse"] = nerf_cls("coarse", use_viewdir=cfg["viewdir"],
                                    encode_transient=cfg["transient"], in_channels_t=n_tau, **arch)
    with torch.no_grad():
        for m in models.values():
            for name, p in m.named_parameters():
                if name.endswith(".weight"):
                    p.mul_(cfg["gain"])
    return models, embeddings


def weight_checksum(models, embeddings):
    """Order-sensitive float64 checksum of every parameter (detects init / RNG drift)."""
    tot, k = 0.0, 1
    mods = [models[k_] for k_ in sorted(models)] + [embeddings[k_] for k_ in sorted(embeddings)
                                                     if isinstance(embeddings[k_], torch.nn.Embedding)]
    for m in mods:
        for _, p in sorted(m.state_dict().items()):
            tot += float((p.double().abs().sum() + p.double().sum() * 0.5)) * (1 + 0.001 * (k % 97))
            k += 1
    return tot


class DatasetStub:
    """The attributes render_rays reads from kwargs['dataset'] (rendering.py:192-199)."""

    def __init__(self, seed):
        rng = np.random.RandomState(seed)
        W, H, f = 512, 288, 400.0
        K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32)
        self.Ks = torch.from_numpy(K)[None]
        self.img_wh = (W, H)
        self.cam_train = [0]
        self.N_frames = N_FRAMES
        poses = np.tile(np.eye(4, dtype=np.float32)[None, :3], (N_FRAMES, 1, 1))
        poses[:, :, 3] = rng.uniform(-0.15, 0.15, (N_FRAMES, 3)).astype(np.float32)
        ang = rng.uniform(-0.2, 0.2, N_FRAMES).astype(np.float32)
        poses[:, 0, 0], poses[:, 0, 2] = np.cos(ang), np.sin(ang)
        poses[:, 2, 0], poses[:, 2, 2] = -np.sin(ang), np.cos(ang)
        self.poses = poses

    def as_oracle_dict(self):
        return dict(K=self.Ks[0].numpy(), H=self.img_wh[1], W=self.img_wh[0], N_frames=self.N_frames,
                    n_cam_train=len(self.cam_train), poses=self.poses)


def wide_rays(n_rays, seed):
    """Rays covering a wider NDC range so that the frustum test of case g5 is not trivial."""
    g = torch.Generator().manual_seed(2000 + seed)
    o = torch.cat([torch.rand(n_rays, 2, generator=g) * 3 - 1.5, -torch.ones(n_rays, 1)], 1)
    d = torch.cat([torch.randn(n_rays, 2, generator=g) * 0.4, 2 * torch.ones(n_rays, 1)], 1)
    ts = torch.full((n_rays,), int(torch.randint(0, N_FRAMES, (1,), generator=g)))
    return torch.cat([o, d], 1).float(), ts


def subset_rows(cfg):
    """Row indices (into the larger synthetic batch) of a case with cfg['batch'] = (batch rays, ray seed, subset seed)."""
    n_batch, _, sub_seed = cfg["batch"]
    return np.random.RandomState(sub_seed).choice(n_batch, cfg["n_rays"], replace=False).astype(np.int64)


def case_inputs(name):
    cfg = CASES[name]
    if cfg.get("batch"):
        n_batch, ray_seed, sub_seed = cfg["batch"]
        rays, ts = synthetic_rays(n_batch, ray_seed)
        idx = torch.from_numpy(subset_rows(cfg))
        rays, ts = rays[idx], ts[idx]
    elif cfg.get("dataset"):
        rays, ts = wide_rays(cfg["n_rays"], cfg["seed"])
    else:
        rays, ts = synthetic_rays(cfg["n_rays"], cfg["seed"])
    return cfg, rays, (ts if cfg["transient"] else None)


def render_kwargs(cfg, dataset=None):
    kw = {}
    if cfg["transient"]:
        kw["output_transient"] = True
        kw["output_transient_flow"] = list(cfg["flow"])
    if dataset is not None:
        kw["dataset"] = dataset
    return kw


def draw_plan(cfg):
    """(key, shape, kind) of every torch RNG draw render_rays makes, in order."""
    n, S, Ni = cfg["n_rays"], cfg["N_samples"], cfg["N_importance"]
    tr = cfg["transient"]
    plan = []
    if cfg.get("perturb", 0) > 0:
        plan.append(("perturb", (n, S), "rand"))
    if Ni > 0:
        plan.append(("coarse_static", (n, S), "randn"))
        if tr:
            plan.append(("coarse_transient", (n, S), "randn"))
        if cfg.get("perturb", 0) != 0:
            plan.append(("u_static", (n, Ni), "rand"))
            if tr:
                plan.append(("u_transient", (n, Ni), "rand"))
    Sf = S + (2 if tr else 1) * Ni if Ni > 0 else S
    plan.append(("fine_static", (n, Sf), "randn"))
    if tr:
        plan.append(("fine_transient", (n, Sf), "randn"))
        if cfg["flow"] and not cfg["test_time"]:
            plan += [("warp_fw", (n, Sf), "randn"), ("warp_bw", (n, Sf), "randn")]
    return plan


def replay_draws(cfg, seed):
    torch.manual_seed(seed)
    out = {}
    for key, shape, kind in draw_plan(cfg):
        out[key] = (torch.rand(*shape) if kind == "rand" else torch.randn(*shape)).numpy()
    return out


# ---- gradient goldens (G9): a fixed random cotangent per differentiable output key ----
GRAD_CASES = ("g3_nsff_train", "g7_nsff_train_noise", "g2_static_c2f", "g13_viewdir_train",
              "g16_two_skips_train", "g17_no_skip_train", "g18_wide_inputs_train")
NON_DIFF_KEYS = ("zs_coarse", "xyzs_coarse", "zs_fine", "xyzs_fine")
FULL_GRAD_PARAMS = ("t.weight", "fine.transient_flow_fw.0.weight", "fine.static_sigma.weight",
                    "fine.static_xyz_encoding_5.0.bias", "coarse.transient_rgb.0.bias", "coarse.static_rgb.0.weight",
                    "a.weight", "fine.static_dir_encoding.0.weight", "coarse.static_dir_encoding.0.bias")


def cotangent_loss(results):
    """sum_k <results[k], C_k> with C_k ~ N(0,1) drawn in sorted-key order from a fixed generator."""
    gen = torch.Generator().manual_seed(555)
    loss = 0.0
    for k in sorted(results):
        c = torch.randn(results[k].shape, generator=gen)
        if k in NON_DIFF_KEYS or not results[k].requires_grad:
            continue
        loss = loss + (results[k] * c.to(device=results[k].device, dtype=results[k].dtype)).sum()
    return loss


def named_grad_params(models, embeddings):
    out = []
    for key in sorted(models):
        out += [(f"{key}.{n}", p) for n, p in models[key].named_parameters()]
    for key in ("t", "a"):
        if key in embeddings:
            out += [(f"{key}.{n}", p) for n, p in embeddings[key].named_parameters()]
    return out


def grad_stats(models, embeddings):
    """{name: [sum g, sum |g|, <g, r>]} with r ~ N(0,1) from a fixed generator, plus a few full gradients."""
    gen = torch.Generator().manual_seed(777)
    stats, full = {}, {}
    for name, p in named_grad_params(models, embeddings):
        r = torch.randn(p.shape, generator=gen)
        g = torch.zeros_like(p).cpu() if p.grad is None else p.grad.detach().cpu()
        stats[name] = [float(g.double().sum()), float(g.double().abs().sum()), float((g.double() * r.double()).sum())]
        if name in FULL_GRAD_PARAMS:
            full[name] = g.numpy().copy()
    return stats, full


# ---- trainer-side synthetic data for the loss goldens (reference losses.py / train.py:136-138,178-198) ----
LOSS_CASES = ("g3_nsff_train", "g7_nsff_train_noise")
LOSS_EPOCH = 5
# The reference's documented TRAINING configuration at its real batch size (README.md:226-233: --use_viewdir --N_samples 128
# --N_importance 0 --batch_size 512, encode_t): the golden g20 holds STATISTICS only -- loss terms and per-parameter gradient
# statistics of the reference in fp32 and fp64 (the per-sample outputs of 512 x 128 points would be 15 MB)
# ... and the bench's own training configuration (C2 / C4: 64 coarse + 64 importance samples -> 192 fine points, coarse and fine model,
# flows + disocclusion) at 512 rays: golden g21 -- statistics, plus the reference's fine depths (sample_pdf is ill-conditioned:
# gradients are compared at identical depths, tests/parity.py)
C2_TRAIN_CASE = dict(n_rays=512, N_samples=64, N_importance=64, transient=True, viewdir=False, appearance=False,
                     test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=21)
README_TRAIN_CASE = dict(n_rays=512, N_samples=128, N_importance=0, transient=True, viewdir=True, appearance=False,
                         test_time=False, flow=['fw', 'bw', 'disocc'], gain=2.5, seed=20)


def camera_buffers():
    """Ks (1,3,3), Ps (1,N_frames,3,4) world->image, max_t -- what train.py registers on the loss module."""
    g = torch.Generator().manual_seed(31337)
    W, H, f = 512.0, 288.0, 400.0
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]])
    Ps = []
    for _ in range(N_FRAMES):
        ang = float(torch.rand(1, generator=g)) * 0.2 - 0.1
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        t = (torch.rand(3, 1, generator=g) - 0.5) * 0.2
        flip = torch.diag(torch.tensor([1.0, -1.0, -1.0]))          # world (right-up-back) -> camera (right-down-front)
        Ps.append(K @ torch.cat([flip @ R, flip @ t], 1))
    return K[None], torch.stack(Ps)[None], N_FRAMES - 1


def synthetic_targets(n_rays, ts, seed):
    g = torch.Generator().manual_seed(4000 + seed)
    return dict(rgbs=torch.rand(n_rays, 3, generator=g), disps=torch.rand(n_rays, generator=g) * 2 + 0.1,
                ts=ts, cam_ids=torch.zeros(n_rays, dtype=torch.long),
                uv_fw=torch.rand(n_rays, 2, generator=g) * torch.tensor([512.0, 288.0]),
                uv_bw=torch.rand(n_rays, 2, generator=g) * torch.tensor([512.0, 288.0]))


# ---- N2: time interpolation (reference rendering.py:365-460) on the small N3 frame ----
INTERP_CFG = dict(N_samples=16, N_importance=16, transient=True, viewdir=False, appearance=False,
                  test_time=True, flow=['fw', 'bw'], gain=2.5, seed=11)
INTERP_T, INTERP_DTS = 7, (0.3, 0.75)
INTERP_KEYS_T = ("xyzs_fine", "zs_fine", "static_rgbs_fine", "static_alphas_fine", "transient_flows_fw",
                 "transient_rgbs_fine", "transient_alphas_fine")
INTERP_KEYS_TP1 = ("transient_flows_bw", "transient_rgbs_fine", "transient_alphas_fine")
