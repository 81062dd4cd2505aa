#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of render_rays on BASELINE.json configs[1] (C2).

One "step" = one ``render_rays`` call on a synthetic 1024-ray batch per GPU: static+dynamic
NSFF, 64 coarse + 64 importance samples (-> 192 fine points per ray), train-mode flags
(fw/bw flow warp into t+-1 with re-query, disocclusion), all 47 result tensors left on
the device.  value = n_gpus * 1024 * (64+64) * steps / time  ("nominal" ray-samples, the
accounting of SURVEY.md 8d).

    python bench.py [--gpus N --steps K --warmup W]

``--gpus N`` with N > 1 starts N ranks itself (one process per GPU, RCCL) when it was not already started by
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`` (both work).
Multi-GPU is weak scaling: every rank renders its own 1024 rays (no data-path collective)
and the step ends with ONE RCCL all-gather of the rendered pixels (rgb_fine, depth_fine).

Extra objects on the JSON line:
  roofline     -- the field (MLP) kernel: algorithmic FLOPs (2*MACs, unpadded K)
                  per launch / average launch duration from HIP events recorded on the
                  launch stream inside the timed region, against the dense MFMA peak of the dtype.
  cpu_baseline -- the CPU oracle (a port of the reference algorithm, dense layers on torch's CPU BLAS like
                  the reference's nn.Linear) timed on this host's cores on a bounded sample of the same
                  workload (rank 0, N=1 only).
  aux          -- (N=1, default workload) the other BASELINE.json configurations, measured in the same run
                  AFTER the headline timed region, a few steps each: training step (C4 per GPU), full-frame
                  evaluation (C3), the reference's README configuration, time interpolation (C5 inner loop).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_RAYS, N_SAMPLES, N_IMPORTANCE = 1024, 64, 64
# MI355X_MICROARCH.md dense peaks: fp32 MFMA (v_mfma_f32_32x32x2_f32) and f16 MFMA (32x32x16)
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0}
MFMA_PER_PRODUCT = {"f32": 1, "f16x3": 3}
KERNEL_NAME = {"f32": "nsff_field_kernel", "f16x3": "nsff_field_kernel_h3a"}
TRAIN_KERNEL_NAME = {"f16x3": "nsff_field_kernel_h3a_save"}
DTYPE_TEXT = {"f32": "f32",
              "f16x3": "f16x3 (fp32 operands split into 2 halfs, 3 f16 MFMAs per product, fp32 accumulate; same 1e-4 parity as f32)"}
FLOP_PER_RAY_C2_TRAIN = 1031.80e6


def build_scene():
    import scenes
    import nsff_pl_amd as A
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=N_RAYS, seed=0)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    return cfg, models, emb


def cpu_baseline(cfg, models_cpu, emb_cpu, n_rays=1024, reps=3):
    """Time the oracle on a bounded sample (same per-ray workload, fewer rays).  The dense layers go through
    torch's CPU BLAS on all host threads -- what the reference's own nn.Linear runs on."""
    import scenes
    from oracle import nsff_oracle as orc
    rays, ts = scenes.synthetic_rays(n_rays, 0)
    fields = {k: orc.field_from_module(m) for k, m in models_cpu.items()}
    kw = dict(emb_t=emb_cpu["t"].weight.detach().numpy(), N_samples=N_SAMPLES, perturb=1.0, noise_std=1.0,
              N_importance=N_IMPORTANCE, test_time=False, output_transient_flow=cfg["flow"])
    def run(n):
        k = dict(kw, draws=scenes.replay_draws(dict(cfg, n_rays=n, perturb=1.0, noise_std=1.0), 1))
        t0 = time.perf_counter()
        orc.render_rays(fields, emb_cpu["xyz"].freqs.numpy(), emb_cpu["dir"].freqs.numpy(), rays.numpy()[:n],
                        ts.numpy()[:n], 29, **k)
        return time.perf_counter() - t0
    default_threads = torch.get_num_threads()
    best, cores = float("inf"), default_threads
    orc.use_torch_dense(True)
    try:
        # skinny GEMMs (N = 256) stop scaling long before 128 threads: probe a few thread counts on a 128-ray sample,
        # then time the full sample with the best one
        probe = {}
        for th in sorted({t for t in (8, 16, 32, 64, default_threads) if t <= default_threads}):
            torch.set_num_threads(th)
            run(128)
            probe[th] = run(128)
        cores = min(probe, key=probe.get)
        torch.set_num_threads(cores)
        for _ in range(reps):
            best = min(best, run(n_rays))
    finally:
        orc.use_torch_dense(False)
        torch.set_num_threads(default_threads)
    return dict(value=n_rays * (N_SAMPLES + N_IMPORTANCE) / best, unit="ray-samples/s", cores=int(cores),
                kind="port", sample=f"{n_rays} rays of the same C2 train-mode workload (fwd), CPU oracle with the dense "
                                    f"layers on torch's CPU BLAS; threads chosen among {sorted(probe)} by a 128-ray probe "
                                    f"(host has {os.cpu_count()} CPUs), best of {reps}")


class Bench:
    """Builds the synthetic scene once and hands out the step functions of the workloads."""

    def __init__(self, rank, world, device, graph=False, standin=False):
        import scenes
        self.scenes, self.rank, self.world, self.device, self.graph, self.standin = scenes, rank, world, device, graph, standin
        self.cfg, self.models, self.emb = build_scene()
        self.cpu = None
        import torch.distributed as dist
        self.live = dist.is_initialized()       # a process group exists (torchrun / self-launch; world size 1 included)
        self.gather_spans = []                  # (start, end) of every pixel all-gather: HIP events on the GPU, seconds on the CPU

    def gather(self, out, keys, counts=None, overlap=True):
        """The step's one collective.  overlap=True (frames: SURVEY 8e): issued on the gather side stream behind this step's
        render and joined one step later, so it runs beside the next step's kernels; bracketed by events on that stream so
        that rank 0 can report the collective's own time.  `finish()` joins what is still in flight (timed() calls it inside
        the timed region, before the closing synchronize).
        overlap=False (the 2 ms render step): the collective on the render stream itself.  Measured at world size 1 under
        torchrun, same box, 100 steps: no gather 1.959 ms, on the render stream 1.995 ms, overlapped 2.030 ms -- the side
        stream's pack / collective kernels can only run in the gaps of a render stream whose field launches hold every CU, and
        the two cross-stream joins cost more than the 20 us collective they hide; a 10-30 ms frame shard is a different
        trade.  NSFF_GATHER_ASYNC=1 / NSFF_GATHER_SYNC=1 force either form (A/B on a multi-GPU node)."""
        from nsff_pl_amd import dist as ndist
        if os.environ.get("NSFF_GATHER_ASYNC"):
            overlap = True
        if os.environ.get("NSFF_GATHER_SYNC"):
            overlap = False
        if self.device.type == "cuda" and not overlap:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            merged = ndist.all_gather_pixels(out, keys, counts=counts)
            ev[1].record()
            self.gather_spans.append(ev)
            return merged
        if self.device.type == "cuda":
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            handle = ndist.all_gather_pixels_async(out, keys, counts=counts, events=ev)
            self.gather_spans.append(ev)
        else:
            t0 = time.perf_counter()
            handle = ndist.all_gather_pixels_async(out, keys, counts=counts)
            handle.wait()                                   # (CPU stand-in: nothing to overlap with; time the collective)
            self.gather_spans.append((t0, time.perf_counter()))
        prev, self.pending = getattr(self, "pending", None), handle
        return prev.wait() if prev is not None else None

    def render_scope(self, overlap=True):
        """Scope of the renders of a step whose gather is issued with gather(..., overlap=overlap): when that gather will run
        BESIDE the next step's render (and the world is larger than one rank) the field launches inside take the
        one-workgroup-per-tile form -- chosen here, per step, restored on exit (nsff_pl_amd.dist.beside_a_collective)."""
        import contextlib
        from nsff_pl_amd import dist as ndist
        if os.environ.get("NSFF_GATHER_ASYNC"):
            overlap = True
        if os.environ.get("NSFF_GATHER_SYNC"):
            overlap = False
        return ndist.beside_a_collective() if (overlap and self.live) else contextlib.nullcontext()

    def field_launch_form(self, workload):
        """What the timed steps' field launches ran as (for the JSON line): the form render_scope selects for this workload."""
        from nsff_pl_amd import config
        overlap = workload in ("eval", "eval_interp")
        with self.render_scope(overlap=overlap):
            persistent = config.get_persistent()
        return ("persistent (one workgroup per CU)" if persistent else
                "one workgroup per tile (a collective kernel runs beside the render stream)")

    def finish(self):
        prev, self.pending = getattr(self, "pending", None), None
        return prev.wait() if prev is not None else None

    def gather_ms(self):
        """Total milliseconds of the recorded gathers (call after a synchronize); clears the record."""
        spans, self.gather_spans = self.gather_spans, []
        if self.device.type == "cuda":
            return sum(a.elapsed_time(b) for a, b in spans)
        return sum((b - a) * 1e3 for a, b in spans)

    def to_device(self):
        for m in list(self.models.values()) + [self.emb["t"]]:
            m.to(self.device)
        rays, ts = self.scenes.synthetic_rays(N_RAYS, 100 + self.rank)
        self.rays, self.ts = rays.to(self.device), ts.to(self.device)
        self.kw = self.scenes.render_kwargs(self.cfg)

    # -- C2: the headline render step
    def render_step(self, train_forward=False):
        import nsff_pl_amd as A
        from nsff_pl_amd import dist as ndist
        scenes, live = self.scenes, self.live
        if self.standin:                                   # CPU plumbing check only (tests): no kernels, constant pixels
            px = {"rgb_fine": torch.full((N_RAYS, 3), float(self.rank)), "depth_fine": torch.zeros(N_RAYS)}

            def step():
                return self.gather(px, ("rgb_fine", "depth_fine")) if live else px
            return step

        def step():
            # the render workload measures the forward path (inference launches); train_forward=True keeps autograd on, so
            # the same call runs the TRAINING forward kernels (every layer executed, activations kept for the backward pass)
            with torch.set_grad_enabled(train_forward), self.render_scope(overlap=False):
                out = A.render_rays(self.models, self.emb, self.rays, self.ts, scenes.N_FRAMES - 1, N_SAMPLES, 1.0, 1.0,
                                    N_IMPORTANCE, 1024 * 32, test_time=False, **self.kw)
            if live:
                self.gather(out, ("rgb_fine", "depth_fine"), overlap=False)
            return out
        return step

    # -- C4 per GPU: the same batch through one training step
    def train_step(self):
        from nsff_pl_amd.training import NSFFTrainer
        scenes, device = self.scenes, self.device
        Ks, Ps, _ = scenes.camera_buffers()
        trainer = NSFFTrainer(self.models, self.emb, scenes.N_FRAMES, dict(N_samples=N_SAMPLES, N_importance=N_IMPORTANCE),
                              Ks, Ps, output_transient_flow=self.cfg["flow"], graph=self.graph).to(device)
        trainer.on_train_epoch_start(0)
        batch = {k: v.to(device) for k, v in scenes.synthetic_targets(N_RAYS, self.ts.cpu(), 100 + self.rank).items()}
        batch["rays"] = self.rays
        self.trainer = trainer
        return lambda: trainer.step(batch)

    # -- C3 / C5: full 512x288 frames
    def frame_steps(self, interp, to_host=False, visibility=True, flow_scale=None):
        """visibility: pass `dataset` like eval.py:134 always does, i.e. run the a6 frustum test of every sample against the
        frame's training camera (C3 is defined with it).  flow_scale: override of the models' scene-flow scale
        for the interpolation workload (random-init flow heads saturate at +-flow_scale NDC = +-50 px at the default 0.2)."""
        from nsff_pl_amd import dist as ndist, evaluate, interpolate
        scenes, device, world, rank = self.scenes, self.device, self.world, self.rank
        H, W = 288, 512
        K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]])
        c2w = torch.tensor([[1.0, 0, 0, 0.02], [0, 1.0, 0, -0.01], [0, 0, 1.0, 0.0]])
        lo, hi = (0, H * W) if interp else ndist.shard_bounds(H * W, world, rank)
        ekw = dict(output_transient=True, output_transient_flow=['fw', 'bw'] if interp else [])
        if visibility:
            ekw["dataset"] = scenes.DatasetStub(5)
        if flow_scale is not None:
            for m in self.models.values():
                if hasattr(m, "flow_scale"):
                    m.flow_scale = float(flow_scale)
                m._pack_cache.invalidate()
        self.frame = dict(H=H, W=W, lo=lo, hi=hi)

        pool = evaluate.PinnedPool(depth=2) if to_host else None     # steady state of a frame loop: no allocation per frame

        def render_t(t, keys):
            rays_f = evaluate.frame_rays(K, c2w, H, W, device=device, first_pixel=lo, n_pixels=hi - lo)
            ts_f = torch.full((hi - lo,), t, device=device, dtype=torch.long)
            return evaluate.render_frame(self.models, self.emb, rays_f, ts_f, scenes.N_FRAMES - 1, 128, 64, 1024 * 32,
                                         keys=keys, to_host=pool, **ekw)

        def eval_step():
            if self.standin:                               # CPU plumbing check (tests): constant pixels of this rank's block
                out = {"rgb_fine": torch.full((hi - lo, 3), float(rank)), "depth_fine": torch.zeros(hi - lo)}
            else:
                with self.render_scope(overlap=not to_host):
                    out = render_t(7, ("rgb_fine", "depth_fine"))
            if self.live and not to_host:
                counts = [b - a_ for a_, b in (ndist.shard_bounds(H * W, world, r) for r in range(world))]
                self.gather(out, ("rgb_fine", "depth_fine"), counts=counts)
            return out

        def interp_step():
            keys_t = ("xyzs_fine", "zs_fine", "static_rgbs_fine", "static_alphas_fine", "transient_flows_fw",
                      "transient_flows_bw", "transient_rgbs_fine", "transient_alphas_fine", "rgb_fine", "depth_fine")
            a_, b_ = render_t(7, keys_t), render_t(8, keys_t)
            return [interpolate(a_, b_, dt / 10, K, c2w, (W, H)) for dt in range(1, 10)]
        return interp_step if interp else eval_step


LAST_STEP_EVENTS = []        # sorted per-step HIP-event milliseconds of the most recent timed() on this rank
LAST_HOST_ISSUE_MS = [None]  # host milliseconds per step spent ISSUING the most recent timed() loop (before any synchronize)


def median_step_ms():
    ev = LAST_STEP_EVENTS
    if not ev:
        return None
    n = len(ev)
    return ev[n // 2] if n % 2 else 0.5 * (ev[n // 2 - 1] + ev[n // 2])


def timed(step, steps, warmup, world, device, prof=False, bench=None):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; returns (seconds [max over ranks],
    field-kernel (launches, ms, flops) from HIP events on the launch stream when prof, per-rank report).  The report
    (`bench` given and a process group alive) holds every rank's own seconds for the K steps and the milliseconds it
    spent in the pixel all-gather: rank 0 prints their min / max, so a scaling line explains itself."""
    import torch.distributed as dist
    from nsff_pl_amd import _lib
    gpu = device.type == "cuda"
    live = dist.is_initialized()

    def fence():
        if gpu:
            torch.cuda.synchronize()
        if live:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    if bench is not None:
        bench.finish()
    fence()
    if bench is not None:
        bench.gather_ms()                       # (drop the warm-up's record)
    if prof:
        _lib.prof_enable(True)
    # hipEvent pairs around every step on torch's current stream, median -- reported beside the
    # wall-clock figure the driver's contract asks for (K steps between two barrier + synchronize brackets)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if gpu else None
    t0 = time.perf_counter()
    for i in range(steps):
        if gpu:
            marks[i].record()
        step()
    LAST_HOST_ISSUE_MS[0] = (time.perf_counter() - t0) / max(steps, 1) * 1e3
    if bench is not None:
        bench.finish()                          # the last step's overlapped pixel gather joins inside the timed region
    if gpu:
        marks[steps].record()
        torch.cuda.synchronize()
    own = time.perf_counter() - t0              # this rank's own K steps, before waiting for the others
    fence()
    elapsed = time.perf_counter() - t0
    LAST_STEP_EVENTS.clear()
    if gpu and steps > 0:
        LAST_STEP_EVENTS.extend(sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps)))
    kern = (0, 0.0, 0.0, 0.0, None)
    if prof:
        kern = _lib.prof_collect()
        _lib.prof_enable(False)
    per_rank = None
    if live:
        mine = torch.tensor([elapsed, own, bench.gather_ms() if bench is not None else 0.0], device=device, dtype=torch.float64)
        allv = torch.empty(world * 3, device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(allv, mine)
        allv = allv.view(world, 3).cpu()
        elapsed = float(allv[:, 0].max())
        per_rank = {"ms_per_step_min": float(allv[:, 1].min()) / steps * 1e3, "ms_per_step_max": float(allv[:, 1].max()) / steps * 1e3,
                    "gather_ms_per_step_min": float(allv[:, 2].min()) / steps, "gather_ms_per_step_max": float(allv[:, 2].max()) / steps,
                    "note": "each rank's own time for the K steps (host clock around its loop + device synchronize, before the "
                            "closing barrier) and the part of it inside the pixel all-gather (HIP events around the collective)"}
    return elapsed, kern, per_rank


NOMINAL_GHZ = 2.4            # the clock the dense peaks of MI355X_MICROARCH.md are quoted at


def roofline_block(precision, kern):
    launches, kernel_ms, kernel_flops, executed_flops, clock_ghz = kern
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    executed = executed_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    try:    # measured separately with rocprofv3 --pmc (profiles/collect_pmc.sh): bytes cannot be counted from here
        traffic = json.load(open(os.path.join(ROOT, "profiles", "field_traffic.json"))).get(precision)
    except Exception:
        traffic = None
    return {"bound": "mfma", "kernel": KERNEL_NAME[precision], "achieved": achieved, "peak": PEAK_TFLOPS[precision],
            "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[precision],
            "clock_ghz": clock_ghz,
            "frac_at_clock": None if not clock_ghz else achieved / (PEAK_TFLOPS[precision] * clock_ghz / NOMINAL_GHZ),
            "clock_note": "shader clock measured inside the timed field launches (s_memtime ticks per XCD / HIP-event time); "
                          f"`peak` and `frac` use the nominal {NOMINAL_GHZ} GHz peak, frac_at_clock scales the peak to the clock the part held",
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/field_traffic.json)",
            "executed": executed,
            "mfma_issue_frac": executed * MFMA_PER_PRODUCT[precision] / PEAK_TFLOPS[precision],
            "note": "achieved = algorithmic FLOPs of the reference network (2*MACs of its fp32 Linear layers) / kernel time; "
                    "executed = what the kernel actually multiplies: the f16 kernels (inference AND training forward) fold the two "
                    "activation-free *_xyz_encoding_final layers (nerf.py:170,195) into pre-multiplied head rows, i.e. skip "
                    "2 of the 18 256x256 layers; the f16x3 mode issues 3 f16 MFMAs per executed product (mfma_issue_frac)",
            "launches": launches, "avg_launch_ms": kernel_ms / max(launches, 1),
            "flop_per_launch": kernel_flops / max(launches, 1)}


def readme_eval(bench, steps=2):
    import scenes
    import nsff_pl_amd as A
    from nsff_pl_amd import evaluate
    dev = bench.device
    cfg = dict(scenes.CASES["g6_readme_viewdir"], appearance=False)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    for m in list(models.values()) + [emb["t"]]:
        m.to(dev)
    H, W = 288, 512
    K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]])
    c2w = torch.tensor([[1.0, 0, 0, 0.02], [0, 1.0, 0, -0.01], [0, 0, 1.0, 0.0]])
    rays = evaluate.frame_rays(K, c2w, H, W, device=dev)
    ts = torch.full((H * W,), 8, device=dev, dtype=torch.long)
    kw = dict(output_transient=True, output_transient_flow=['fw', 'bw'])
    out = {"reference_published": {"seconds_per_frame": 8.04, "ray_samples_per_s": 2.35e6, "hardware": "RTX 2080 Ti (implied)",
                                   "source": "test.ipynb:122"},
           "config": "512x288, N_samples=128, N_importance=0, use_viewdir=True, flows fw+bw, chunk=16384, test_time, every key "
                     "(24 tensors, 1.75 GB per frame) delivered to the host; random-init weights, synthetic pose"}
    pool = evaluate.PinnedPool(depth=2)
    forms = (("resident", dict()), ("blocking_cpu_every_chunk", dict(to_cpu=True)), ("pinned_async", dict(to_host=pool)))
    for name, egress in forms:
        def step():
            return evaluate.render_frame(models, emb, rays, ts, scenes.N_FRAMES - 1, 128, 0, 1024 * 16, **egress, **kw)
        t, _, _ = timed(step, steps, 1, 1, dev)
        out[name] = {"seconds_per_frame": t / steps, "ray_samples_per_s": H * W * 128 * steps / t,
                     "vs_published": (H * W * 128 * steps / t) / 2.35e6}
    out["blocking_cpu_every_chunk"]["note"] = "the reference's own egress (eval.py:106-107 / test.ipynb cell 1): .cpu() of every key per chunk"
    return out


def readme_train(bench, steps=10):
    """The reference's documented TRAINING configuration (README.md:226-233: --use_viewdir --N_samples 128 --N_importance 0
    --batch_size 512, encode_t, the NSFF flow outputs) as one full step of NSFFTrainer -- HIP forward, fused NeRFWLoss, native
    backward, Adam -- on 512 synthetic rays with random-init weights.  Which kernels the step's field launches took is reported
    beside the time (a view-direction static trunk still trains on the eight-wave SAVE kernel and the compiler-scheduled
    data-gradient kernel; the dynamic trunk and the re-queries on the hand-scheduled ones)."""
    import scenes
    import nsff_pl_amd as A
    from nsff_pl_amd import _lib
    from nsff_pl_amd.training import NSFFTrainer
    dev = bench.device
    cfg = dict(scenes.CASES["g13_viewdir_train"], appearance=False, n_rays=512, N_samples=128, N_importance=0)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    models = {"fine": models["fine"]}                      # (N_importance = 0: the reference builds no coarse model, train.py:75-86)
    Ks, Ps, _ = scenes.camera_buffers()
    trainer = NSFFTrainer(models, emb, scenes.N_FRAMES, dict(N_samples=128, N_importance=0), Ks, Ps,
                          output_transient_flow=cfg["flow"], graph=False).to(dev)
    trainer.on_train_epoch_start(0)
    rays, ts = scenes.synthetic_rays(512, 321)
    batch = {k: v.to(dev) for k, v in scenes.synthetic_targets(512, ts, 321).items()}
    batch["rays"] = rays.to(dev)
    kernels = {"forward": set(), "backward": set()}
    orig_q, orig_b = _lib.field_query, _lib.field_backward

    def q(*a, **k):
        r = orig_q(*a, **k)
        kernels["forward"].add(_lib.last_field_kernel())
        return r

    def b(*a, **k):
        r = orig_b(*a, **k)
        kernels["backward"].add(_lib.last_bwd_kernel())
        return r
    _lib.field_query, _lib.field_backward = q, b
    try:
        trainer.step(batch)
    finally:
        _lib.field_query, _lib.field_backward = orig_q, orig_b
    t, _, _ = timed(lambda: trainer.step(batch), steps, 2, 1, dev)
    out = {"config": "README.md:226-233: use_viewdir, N_samples=128, N_importance=0, batch_size=512, encode_t, flows fw/bw/disocc; one "
                     "NSFFTrainer.step (forward, NeRFWLoss, backward, Adam), eager; random-init weights, synthetic rays and targets",
           "ms_per_step": t / steps * 1e3, "ray_samples_per_s": 512 * 128 * steps / t,
           "forward_kernels": sorted(kernels["forward"]), "data_gradient_kernels": sorted(kernels["backward"])}
    # the same step as the trainer's two replayed hipGraphs (NSFFTrainer(graph=True), what graph="auto" picks at this size): a step is
    # ~100 launches of 2.5 ms of kernels here -- the eager step is bound by the host's launch rate, the replay is not
    try:
        models_g, emb_g = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
        tg = NSFFTrainer({"fine": models_g["fine"]}, emb_g, scenes.N_FRAMES, dict(N_samples=128, N_importance=0), Ks, Ps,
                         output_transient_flow=cfg["flow"], graph=True).to(dev)
        tg.on_train_epoch_start(0)
        for _ in range(3):
            tg.step(batch)
        t, _, _ = timed(lambda: tg.step(batch), steps, 1, 1, dev)
        out["ms_per_step_graph"] = t / steps * 1e3
    except Exception as e:                                   # (reported, not fatal: the eager figure above is the entry's subject)
        out["ms_per_step_graph"] = None
        out["graph_error"] = repr(e)[:200]
    return out


def backward_rooflines(bench, args, steps=4):
    """The two other MFMA kernels of a training step against the f16 peak, per field node.  Each is timed with HIP events around its
    C-ABI call during eager training steps, TWICE: `serialised` -- the weight-gradient launches on the backward pass's own stream
    (NSFF_WGRAD_OVERLAP=0), so an event pair brackets that kernel's own execution -- and `overlapped`, the trainer's default (the
    weight-gradient launches of node k on a side stream beside the data-gradient kernel of node k + 1): there both kernels take a
    whole compute unit per workgroup and share the HBM, so an event pair on either stream also contains the time its kernel waited
    for the other one -- the sum of the two is what counts, neither span alone is a kernel time (round 5 reported only those).
    `frac` is the serialised figure."""
    from nsff_pl_amd import _lib
    orig = {n: getattr(_lib, n) for n in ("field_backward", "weight_grad", "weight_grad_accumulate", "field_query")}

    def measure(overlap):
        spans = {"nsff_field_backward": [], "nsff_weight_grad": []}

        def timed_call(name, key):
            def f(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig[name](*a, **k)
                e1.record()
                spans[key].append((e0, e1))
                return r
            return f
        keep_models = bench.models
        bench.graph = False
        old_env = os.environ.get("NSFF_WGRAD_OVERLAP")
        os.environ["NSFF_WGRAD_OVERLAP"] = "1" if overlap else "0"
        try:
            step = bench.train_step()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            _lib.field_backward = timed_call("field_backward", "nsff_field_backward")
            _lib.weight_grad = timed_call("weight_grad", "nsff_weight_grad")
            _lib.weight_grad_accumulate = timed_call("weight_grad_accumulate", "nsff_weight_grad")
            _lib.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            step_ms = (time.perf_counter() - t0) / steps * 1e3
            launches, ms, flops, _ex, ghz = _lib.prof_collect()
            _lib.prof_enable(False)
        finally:
            for n, f in orig.items():
                setattr(_lib, n, f)
            if old_env is None:
                os.environ.pop("NSFF_WGRAD_OVERLAP", None)
            else:
                os.environ["NSFF_WGRAD_OVERLAP"] = old_env
        bench.models = keep_models
        per_node = flops / max(launches, 1)
        res = {}
        for key, sp in spans.items():
            t = sum(a.elapsed_time(b) for a, b in sp)
            n = len(sp)
            tf = per_node * n / (t * 1e-3) / 1e12 if t > 0 else 0.0
            res[key] = (n, t / max(n, 1), tf)
        return res, launches, ms / max(launches, 1), step_ms
    ser, launches, fwd_ms, step_ser = measure(False)
    ovl, _, _, step_ovl = measure(True)
    out = {"note": "per field node of a training step (four per step); FLOPs = 2 x MACs of the differentiated layers = those of the "
                   "node's training forward; nsff_weight_grad = the GEMM launches + the split-K reduction that accumulates into .grad. "
                   "avg_ms / achieved / frac: SERIALISED (each kernel alone on the GPU: its own execution time); avg_ms_overlapped: the "
                   "event span in the trainer's default form, where it also contains the time spent waiting for the other stream's kernel",
           "forward_launches": launches, "forward_ms_per_launch": fwd_ms,
           "step_ms_serialised": step_ser, "step_ms_overlapped": step_ovl,
           "data_gradient_kernel": "nsff_field_bwd_kernel_h3b (hand-scheduled body)" if _lib.last_bwd_kernel() == "h3b" else "nsff_field_bwd_kernel"}
    for key in ser:
        n, avg, tf = ser[key]
        out[key] = {"calls": n, "avg_ms": avg, "avg_ms_overlapped": ovl[key][1], "achieved": tf, "peak": PEAK_TFLOPS["f16x3"], "unit": "TFLOP/s",
                    "frac": tf / PEAK_TFLOPS["f16x3"], "bound": "mfma (f16, one product per MAC)" if key == "nsff_field_backward"
                    else "hbm / mfma balanced (128 FLOP per byte of saved activations)"}
    return out


def aux_block(bench, args):
    """The other configurations, a few steps each, after the headline (single GPU only)."""
    from nsff_pl_amd import config
    aux = {"note": "measured in this run after the headline timed region; steps/warmup per entry"}
    dev = bench.device
    # (1a) the headline call as ONE replayed hipGraph (nsff_pl_amd.graphs.GraphedRender): same kernels, no per-launch host work
    from nsff_pl_amd.graphs import GraphedRender
    gr = GraphedRender(bench.models, bench.emb, bench.scenes.N_FRAMES - 1, N_SAMPLES, 1.0, 1.0, N_IMPORTANCE, test_time=False, **bench.kw)
    gr(bench.rays, bench.ts)
    t, _, _ = timed(lambda: gr(bench.rays, bench.ts), 20, 3, 1, dev)
    aux["render_as_hip_graph"] = {"ms_per_step": t / 20 * 1e3, "ray_samples_per_s": N_RAYS * (N_SAMPLES + N_IMPORTANCE) * 20 / t,
                                  "note": "the C2 call captured once and replayed (17 kernel nodes incl. the generator kernels); a replay has a fixed "
                                          "cost of ~10 us (a one-node graph: 9.8 us, an eager launch 4.4 us -- tools/debug/graph_vs_eager.py) that "
                                          "the eager path, whose host runs ahead of the GPU, never pays"}
    # (1b) the TRAINING forward of the same C2 call (what a training step launches: every layer executed, activations and
    # ReLU sign bits kept for the backward pass) next to the headline's inference launches
    t, kern, _ = timed(bench.render_step(train_forward=True), 10, 2, 1, dev, prof=True)
    rf = roofline_block(args.precision, kern)
    aux["train_forward"] = {"label": "C2 forward as a training step runs it (autograd on): the SAVE build of the hand-scheduled body -- every trunk "
                                     "layer's activation and ReLU sign words go to HBM while the phases multiply; *_final folded into the heads as in inference",
                            "ms_per_step": t / 10 * 1e3, "ray_samples_per_s": N_RAYS * (N_SAMPLES + N_IMPORTANCE) * 10 / t,
                            "roofline": dict({k: rf[k] for k in ("achieved", "executed", "peak", "unit", "frac", "clock_ghz", "frac_at_clock",
                                                                 "avg_launch_ms", "launches")},
                                             kernel=TRAIN_KERNEL_NAME.get(args.precision, KERNEL_NAME[args.precision]))}
    # (1c) the two other MFMA kernels of a training step, each against the same f16 peak: nsff_field_backward (data gradients,
    # one f16 product per MAC) and the weight-gradient GEMMs (nsff_weight_grad*: dW = dY^T X over the saved activations).
    # Algorithmic FLOPs of either = 2 x MACs of the layers it differentiates = the FLOPs of the training forward launch of the
    # same field node; time = HIP events around the C-ABI calls (on the stream they are issued on) during eager training steps.
    aux["train_backward_kernels"] = backward_rooflines(bench, args)
    # (2) C3 as SURVEY 8d defines it: one 512x288 test-time frame WITH the frustum-visibility test of every sample point
    # (eval.py:134 always passes `dataset`), and without it for comparison
    t, _, _ = timed(bench.frame_steps(False), 3, 1, 1, dev)
    aux["eval_ms_per_frame"] = t / 3 * 1e3
    aux["eval_ray_samples_per_s"] = 288 * 512 * (128 + 64) * 3 / t
    aux["eval_note"] = "visibility on (dataset passed, a6 inside the compositing kernel)"
    t, _, _ = timed(bench.frame_steps(False, visibility=False), 3, 1, 1, dev)
    aux["eval_ms_per_frame_no_visibility"] = t / 3 * 1e3
    # ... and with the pixels (rgb_fine, depth_fine) delivered to pinned host memory chunk by chunk on a copy stream (row N4)
    t, _, _ = timed(bench.frame_steps(False, to_host=True), 3, 1, 1, dev)
    aux["eval_ms_per_frame_pixels_to_pinned_host"] = t / 3 * 1e3
    # (2b) the reference's ONE published timing, like for like (test.ipynb:122, config :34,78-85 -- SURVEY section 6): one
    # 512x288 frame, N_samples=128, N_importance=0, NeRF('fine', encode_transient, output_flow) with the constructor's
    # default use_viewdir=True, output_transient_flow=['fw','bw'], chunk=16384, EVERY result key brought to the host after
    # every chunk: 8.04 s on an RTX 2080 Ti = 2.35 M ray-samples/s
    aux["readme_eval"] = readme_eval(bench)
    # (2c) ... and the reference's documented TRAINING configuration, one full step (README.md:226-233)
    aux["readme_train"] = readme_train(bench)
    # (3) C5 inner loop: 2 rendered + 9 interpolated frames.  Random-init flow heads saturate at +-flow_scale: at the default
    # 0.2 NDC every sample moves ~+-50 px and the splat runs on its FAR path; a trained field moves a few pixels -- the NEAR
    # figure uses flow_scale 0.02 (+-5 px) on the same weights.  Both are reported, each labelled.
    for label, fs in (("far_50px", 0.2), ("near_5px", 0.02)):
        t, _, _ = timed(bench.frame_steps(True, flow_scale=fs), 2, 1, 1, dev)
        aux[f"interp_ms_per_11_frames_{label}"] = t / 2 * 1e3
        aux[f"interp_frames_per_s_{label}"] = 10 * 2 / t
    bench.frame_steps(True, flow_scale=0.2)                      # (restore the models' flow scale)
    # (4) C4 per GPU: training step (changes the weights, so it goes last)
    for graph in (False, True):
        bench.graph = graph
        cfg_models = bench.models
        # (ten timed steps behind five warm-up ones: the first steps of a fresh trainer still grow the caching allocator's pools --
        #  five timed steps behind three read 5.96 ms where `--workload train` reads 5.65)
        t, _, _ = timed(bench.train_step(), 10, 5 if not graph else 2, 1, dev)
        aux["train_ms_per_step" + ("_graph" if graph else "_eager")] = t / 10 * 1e3
        bench.models = cfg_models
    # (4b) the same eager step with the parity-grade backward (config.set_grad_precision("f16x3"): three products per multiply-
    # accumulate in the data-gradient chain and the weight-gradient GEMMs, the forward on the eight-wave SAVE kernel that writes
    # the remainder planes) -- what the option costs
    bench.graph = False
    config.set_grad_precision("f16x3")
    try:
        cfg_models = bench.models
        t, _, _ = timed(bench.train_step(), 5, 3, 1, dev)
        aux["train_ms_per_step_eager_grad_f16x3"] = t / 5 * 1e3
        bench.models = cfg_models
    finally:
        config.set_grad_precision("f16")
    # (the trainer's default form is the eager step -- NSFFTrainer(graph=False); the two-hipGraph replay is 2-3 % slower on this
    #  stack: a replayed node costs what an eager launch costs, the host already runs ahead of the GPU, and the replay adds a fixed
    #  cost per graph launch: DESIGN.md section 7)
    aux["train_ray_samples_per_s"] = N_RAYS * (N_SAMPLES + N_IMPORTANCE) / (aux["train_ms_per_step_eager"] * 1e-3)
    return aux


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=600.0,
                    help="untimed steps run for this long BEFORE the W warm-up steps: the part comes out of its idle state (95 MHz "
                         "while the CPU baseline runs) and its power management settles; 0 = none.  Reported in config.settle_ms")
    ap.add_argument("--no-aux", action="store_true", help="skip the aux block (other configurations after the headline)")
    ap.add_argument("--precision", default=os.environ.get("NSFF_PRECISION", "f16x3"), choices=["f32", "f16x3"],
                    help="arithmetic of the dense layers; f32 (exact fp32 MFMA) and f16x3 (three f16 products per MAC, the "
                         "default) pass the same 1e-4 parity tests")
    ap.add_argument("--workload", default="render", choices=["render", "train", "eval", "eval_interp"],
                    help="render = C2 (headline, default); train = C4: the same batch through NSFFTrainer.step "
                         "(HIP forward, NeRFWLoss, native HIP backward, flat RCCL gradient all-reduce, Adam); eval = C3: one "
                         "512x288 test-time frame per step (on-device ray generation, 32768-ray chunks, 128+64 "
                         "samples), rays sharded over the ranks; eval_interp = C5's inner loop: two frames (t, t+1) "
                         "+ 9 interpolated frames per step")
    ap.add_argument("--graph", action="store_true", help="train workload: replay the step as captured hipGraphs")
    ap.add_argument("--tile-points", type=int, default=int(os.environ.get("NSFF_TILE_POINTS", "0")))
    ap.add_argument("--standin", action="store_true",
                    help="CPU plumbing check used by tests/test_bench_launch.py: spawn / rendezvous / barrier / max-over-ranks "
                         "/ rank-0 JSON line with a constant stand-in for the render step (gloo); never a measurement")
    args = ap.parse_args()

    from nsff_pl_amd import config, dist as ndist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by torchrun: be the launcher (one rank per GPU), rank 0 prints the JSON line
        sys.exit(ndist.launch_local(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    import torch.distributed as dist
    config.set_precision(args.precision)
    config.set_tile_points(args.tile_points if args.precision == "f16x3" else 0)

    rank, world, device = ndist.init_from_env("gloo" if args.standin else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.standin:
        device = torch.device("cpu")
    else:
        assert device.type == "cuda", "bench.py needs the MI355X"

    bench = Bench(rank, world, device, graph=args.graph and args.workload == "train", standin=args.standin)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.standin:
        cpu = cpu_baseline(bench.cfg, bench.models, bench.emb)
    if not args.standin:
        bench.to_device()
    step = {"render": bench.render_step, "train": bench.train_step,
            "eval": lambda: bench.frame_steps(False), "eval_interp": lambda: bench.frame_steps(True)}[args.workload]()

    if args.settle_ms > 0 and not args.standin:
        # the same number of settle steps on every rank (a step may end in a collective): sized from ten steps on this rank
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        n_settle = int(args.settle_ms * 1e-3 / max((time.perf_counter() - t0) / 10, 1e-6)) + 1
        if dist.is_initialized():
            nt = torch.tensor([n_settle], device=device)
            dist.all_reduce(nt, op=dist.ReduceOp.MAX)
            n_settle = int(nt.item())
        for _ in range(n_settle):
            step()
        bench.finish()
        torch.cuda.synchronize()
    elapsed, kern, per_rank = timed(step, args.steps, args.warmup, world, device, prof=not args.standin, bench=bench)
    median_ms = median_step_ms()               # (of the headline's steps: the aux block below times other things)
    host_issue_ms = LAST_HOST_ISSUE_MS[0]

    aux = None
    if rank == 0 and world == 1 and args.workload == "render" and not args.no_aux and not args.standin:
        aux = aux_block(bench, args)

    if rank == 0:
        value = world * N_RAYS * (N_SAMPLES + N_IMPORTANCE) * args.steps / elapsed
        line = {
            "metric": "ray-samples/sec (coarse+fine, static+dynamic)" + (" -- TRAINING step" if args.workload == "train" else ""),
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median_events": median_ms,
            # host time per step to ISSUE the launches (no synchronize inside): close to ms_per_step = the host is the limit
            "host_issue_ms_per_step": host_issue_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
            "config": {"workload": "C2 (BASELINE.json configs[1]): static+dynamic NSFF, 1024 rays/GPU x (64 coarse + "
                                   "64 importance -> 192 fine pts), train-mode fwd, fw/bw flow warp t+-1, "
                                   "perturb=1 noise_std=1, 8x256 MLPs, N_tau=48, all 47 outputs on device.  `value` is "
                                   "render_rays' throughput with autograd DISABLED (the train-mode FLAGS, inference launches: "
                                   "nothing kept for a backward pass); what a trainer's forward costs is aux.train_forward, "
                                   "the whole training step aux.train_ms_per_step_*",
                       "settle_ms": args.settle_ms,
                       "rays_per_gpu": N_RAYS, "N_samples": N_SAMPLES, "N_importance": N_IMPORTANCE,
                       "parallelism": f"ray-shard x{world}, pixel all-gather" if bench.live else "single GPU",
                       "field_launch": bench.field_launch_form(args.workload),
                       "rays_per_s": world * N_RAYS * args.steps / elapsed,
                       "mlp_tflops_whole_step": world * N_RAYS * args.steps * FLOP_PER_RAY_C2_TRAIN / elapsed / 1e12},
        }
        if args.standin:
            line["data"] = "STAND-IN (CPU plumbing check, not a measurement)"
            line["config"] = {"workload": "stand-in"}
        else:
            line["roofline"] = roofline_block(args.precision, kern)
        if args.workload in ("eval", "eval_interp"):
            n_px = bench.frame["H"] * bench.frame["W"]
            if args.workload == "eval":
                line["metric"] = "ray-samples/sec (coarse+fine, static+dynamic) -- full-frame EVALUATION"
                line["value"] = n_px * (128 + 64) * args.steps / elapsed
                line["config"] = {"workload": "C3 (BASELINE.json configs[2]): 512x288 test-time frame per step, rays generated "
                                              "on the device, 32768-ray chunks, 128 coarse + 64 importance samples -> 256 fine "
                                              "points/ray, static+dynamic, rays sharded over the ranks + one pixel all-gather",
                                  "frames_per_s": args.steps / elapsed, "rays_per_frame": n_px,
                                  "parallelism": f"ray-shard x{world}" if world > 1 else "single GPU"}
                line["scaling"] = "strong"
            else:
                line["metric"] = "frames/sec -- fixed-view time interpolation x10 (2 rendered + 9 interpolated frames per step)"
                line["unit"] = "frames/s"
                line["value"] = world * 10 * args.steps / elapsed
                line["config"] = {"workload": "C5 inner loop (BASELINE.json configs[4]): render t and t+1 (512x288, 128+64 samples, "
                                              "flows), 9 x interpolate (plane splat + MPI composite, 256 planes); every rank does "
                                              "its own frame pair", "parallelism": f"frame-pair per rank x{world}"}
        if args.workload == "train":
            line["config"]["workload"] = ("C4 (BASELINE.json configs[3]) per GPU: the C2 batch through one training step = "
                                          "HIP forward (training variant: keeps activations) + NeRFWLoss (11 terms) + native HIP "
                                          "backward (nsff_composite_backward, nsff_field_backward, nsff_weight_grad) + "
                                          "flat RCCL gradient all-reduce + Adam")
            line["config"]["parallelism"] = f"data-parallel x{world}, one flat gradient all-reduce per step"
            line["config"]["hip_graph"] = bool(bench.trainer.graph)
            line["config"].pop("mlp_tflops_whole_step")
            if "roofline" in line:      # (the field launches of a training step are the training forward's)
                line["roofline"]["kernel"] = TRAIN_KERNEL_NAME.get(args.precision, line["roofline"]["kernel"])
        if per_rank is not None:
            line["per_rank"] = per_rank
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if aux is not None:
            line["aux"] = aux
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
