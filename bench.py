#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of render_rays on BASELINE.json configs[1] (C2).

One "step" = one ``render_rays`` call on a synthetic 1024-ray batch per GPU: static+dynamic
NSFF, 64 coarse + 64 importance samples (-> 192 fine points per ray), train-mode flags
(fw/bw flow warp into t+-1 with re-query, disocclusion), all 47 result tensors left on
the device.  value = n_gpus * 1024 * (64+64) * steps / time  ("nominal" ray-samples, the
accounting of SURVEY.md 8d).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU is weak scaling: every rank renders its own 1024 rays (no data-path collective)
and the step ends with ONE RCCL all-gather of the rendered pixels (rgb_fine, depth_fine).

Extra objects on the JSON line:
  roofline     -- the field (MLP) kernel: algorithmic FLOPs (2*MACs, unpadded K)
                  per launch / average launch duration from HIP events recorded on the
                  launch stream inside the timed region, against the fp32 MFMA peak.
  cpu_baseline -- the numpy oracle (a port of the reference algorithm) timed on this host's
                  cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_RAYS, N_SAMPLES, N_IMPORTANCE = 1024, 64, 64
# MI355X_MICROARCH.md dense peaks: fp32 MFMA (v_mfma_f32_32x32x2_f32) and f16 MFMA (32x32x16)
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16x3_ra": 2500.0}
FLOP_PER_RAY_C2_TRAIN = 1031.80e6


def build(device):
    import scenes
    import nsff_pl_amd as A
    cfg = dict(scenes.CASES["g3_nsff_train"], n_rays=N_RAYS, seed=0)
    models, emb = scenes.build_scene(A.NeRF, A.PosEmbedding, cfg)
    host = dict(models={k: m for k, m in models.items()}, emb=emb, cfg=cfg)
    return cfg, models, emb, host


def cpu_baseline(cfg, models_cpu, emb_cpu, n_rays=24, reps=2):
    """Time the oracle on a bounded sample (same per-ray workload, fewer rays)."""
    import scenes
    from oracle import nsff_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    rays, ts = scenes.synthetic_rays(n_rays, 0)
    fields = {k: orc.field_from_module(m) for k, m in models_cpu.items()}
    draws = scenes.replay_draws(dict(cfg, n_rays=n_rays, perturb=1.0, noise_std=1.0), 1)
    kw = dict(emb_t=emb_cpu["t"].weight.detach().numpy(), N_samples=N_SAMPLES, perturb=1.0, noise_std=1.0,
              N_importance=N_IMPORTANCE, test_time=False, draws=draws, output_transient_flow=cfg["flow"])
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.render_rays(fields, emb_cpu["xyz"].freqs.numpy(), emb_cpu["dir"].freqs.numpy(), rays.numpy(),
                        ts.numpy(), 29, **kw)
        best = min(best, time.perf_counter() - t0)
    return dict(value=n_rays * (N_SAMPLES + N_IMPORTANCE) / best, unit="ray-samples/s", cores=int(cores),
                kind="port", sample=f"{n_rays} rays of the same C2 train-mode workload, numpy oracle, best of {reps}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("NSFF_PRECISION", "f16x3"), choices=["f32", "f16x3", "f16x3_ra"],
                    help="arithmetic of the dense layers; both modes pass the same 1e-4 parity tests")
    ap.add_argument("--workload", default="render", choices=["render", "train", "eval", "eval_interp"],
                    help="render = C2 (headline, default); train = C4: the same batch through NSFFTrainer.step "
                         "(HIP forward, NeRFWLoss, backward, flat RCCL gradient all-reduce, Adam); eval = C3: one "
                         "512x288 test-time frame per step (on-device ray generation, 32768-ray chunks, 128+64 "
                         "samples), rays sharded over the ranks; eval_interp = C5's inner loop: two frames (t, t+1) "
                         "+ 9 interpolated frames per step")
    ap.add_argument("--graph", action="store_true", help="train workload: replay the step as one captured hipGraph")
    ap.add_argument("--tile-points", type=int, default=int(os.environ.get("NSFF_TILE_POINTS", "0")))
    args = ap.parse_args()

    import scenes
    import nsff_pl_amd as A
    from nsff_pl_amd import _lib, config, dist as ndist
    import torch.distributed as dist
    config.set_precision(args.precision)
    config.set_tile_points(args.tile_points if args.precision == "f16x3" else 0)

    rank, world, device = ndist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert device.type == "cuda", "bench.py needs the MI355X"

    cfg, models, emb, _ = build(device)
    cpu_models = {k: m for k, m in models.items()}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, cpu_models, emb)
    for m in list(models.values()) + [emb["t"]]:
        m.to(device)
    rays, ts = scenes.synthetic_rays(N_RAYS, 100 + rank)
    rays, ts = rays.to(device), ts.to(device)
    kw = scenes.render_kwargs(cfg)

    trainer = None
    if args.workload == "train":
        from nsff_pl_amd.training import NSFFTrainer
        Ks, Ps, _ = scenes.camera_buffers()
        trainer = NSFFTrainer(models, emb, scenes.N_FRAMES, dict(N_samples=N_SAMPLES, N_importance=N_IMPORTANCE),
                              Ks, Ps, output_transient_flow=cfg["flow"], graph=args.graph and world == 1).to(device)
        trainer.on_train_epoch_start(0)
        batch = {k: v.to(device) for k, v in scenes.synthetic_targets(N_RAYS, ts.cpu(), 100 + rank).items()}
        batch["rays"] = rays

    frame = None
    if args.workload in ("eval", "eval_interp"):
        from nsff_pl_amd import evaluate, interpolate
        H, W = 288, 512
        K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]])
        c2w = torch.tensor([[1.0, 0, 0, 0.02], [0, 1.0, 0, -0.01], [0, 0, 1.0, 0.0]])
        lo, hi = ndist.shard_bounds(H * W, world, rank) if args.workload == "eval" else (0, H * W)
        ekw = dict(output_transient=True, output_transient_flow=['fw', 'bw'] if args.workload == "eval_interp" else [])
        frame = dict(H=H, W=W, lo=lo, hi=hi)

        def render_t(t, keys):
            rays_f = evaluate.frame_rays(K, c2w, H, W, device=device, first_pixel=lo, n_pixels=hi - lo)
            ts_f = torch.full((hi - lo,), t, device=device, dtype=torch.long)
            return evaluate.render_frame(models, emb, rays_f, ts_f, scenes.N_FRAMES - 1, 128, 64, 1024 * 32,
                                         keys=keys, **ekw)

    def step():
        if frame is not None:
            if args.workload == "eval":
                out = render_t(7, ("rgb_fine", "depth_fine"))
                if world > 1:
                    counts = [b - a_ for a_, b in (ndist.shard_bounds(frame["H"] * frame["W"], world, r) for r in range(world))]
                    ndist.all_gather_pixels(out, ("rgb_fine", "depth_fine"), counts=counts)
                return out
            keys_t = ("xyzs_fine", "zs_fine", "static_rgbs_fine", "static_alphas_fine", "transient_flows_fw",
                      "transient_flows_bw", "transient_rgbs_fine", "transient_alphas_fine", "rgb_fine", "depth_fine")
            a_, b_ = render_t(7, keys_t), render_t(8, keys_t)
            return [interpolate(a_, b_, dt / 10, K, c2w, (frame["W"], frame["H"])) for dt in range(1, 10)]
        if trainer is not None:
            return trainer.step(batch)
        with torch.no_grad():      # the render workload measures the forward path; --workload train the full step
            out = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, N_SAMPLES, 1.0, 1.0,
                                N_IMPORTANCE, 1024 * 32, test_time=False, **kw)
        if world > 1:
            ndist.all_gather_pixels(out, ("rgb_fine", "depth_fine"))
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = _lib.prof_collect()
    _lib.prof_enable(False)

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = world * N_RAYS * (N_SAMPLES + N_IMPORTANCE) * args.steps / elapsed
        achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        try:    # measured separately with rocprofv3 --pmc (profiles/collect_pmc.sh); not measurable from here
            traffic = json.load(open(os.path.join(ROOT, "profiles", "field_traffic.json"))).get(args.precision)
        except Exception:
            traffic = None
        line = {
            "metric": "ray-samples/sec (coarse+fine, static+dynamic)" + (" -- TRAINING step" if trainer else ""),
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f16x3 (fp32 operands split into 2 halfs, 3 f16 MFMAs per product, fp32 accumulate; same 1e-4 parity as f32)",
            "data": "synthetic",
            "config": {"workload": "C2 (BASELINE.json configs[1]): static+dynamic NSFF, 1024 rays/GPU x (64 coarse + "
                                   "64 importance -> 192 fine pts), train-mode fwd, fw/bw flow warp t+-1, "
                                   "perturb=1 noise_std=1, 8x256 MLPs, N_tau=48, all 47 outputs on device",
                       "rays_per_gpu": N_RAYS, "N_samples": N_SAMPLES, "N_importance": N_IMPORTANCE,
                       "parallelism": f"ray-shard x{world}, pixel all-gather" if world > 1 else "single GPU",
                       "rays_per_s": world * N_RAYS * args.steps / elapsed,
                       "mlp_tflops_whole_step": world * N_RAYS * args.steps * FLOP_PER_RAY_C2_TRAIN / elapsed / 1e12},
            "roofline": {"bound": "mfma", "kernel": {"f32": "nsff_field_kernel", "f16x3": "nsff_field_kernel_h3<2,1>", "f16x3_ra": "nsff_field_kernel_ra"}[args.precision],
                         "achieved": achieved, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC, profiles/field_traffic.json)",
                         "mfma_issue_frac": achieved * (1 if args.precision == "f32" else 3) / PEAK_TFLOPS[args.precision],
                         "note": "achieved = algorithmic FLOPs (2*MACs of the fp32 Linear layers); the f16x3 mode "
                                 "issues 3 f16 MFMAs per algorithmic product, so frac <= 1/3 there",
                         "launches": launches, "avg_launch_ms": kernel_ms / max(launches, 1),
                         "flop_per_launch": kernel_flops / max(launches, 1)},
        }
        if frame is not None:
            n_px = frame["H"] * frame["W"]
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
        if trainer is not None:
            line["config"]["workload"] = ("C4 (BASELINE.json configs[3]) per GPU: the C2 batch through one training step = "
                                          "HIP forward + NeRFWLoss (11 terms) + backward (torch/rocBLAS re-evaluation) + "
                                          "flat RCCL gradient all-reduce + Adam")
            line["config"]["parallelism"] = f"data-parallel x{world}, one flat gradient all-reduce per step"
            line["config"]["hip_graph"] = bool(trainer.graph)
            line["config"].pop("mlp_tflops_whole_step")
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
