"""The multi-GPU path driven through RCCL (torch.distributed backend "nccl") on the ONE GPU a test box has: world size 1.

What the eight-GPU job runs -- ``dist.init_from_env`` with the communicator bound to ``cuda:LOCAL_RANK``, the packed pixel
all-gather (even and ``counts=`` forms), ``render_frame_sharded``, ``NSFFTrainer(graph=True)`` with the flat gradient
all-reduce issued between its two hipGraphs, and ``bench.py --gpus 1`` in the form the driver starts it for N > 1
(``python -m torch.distributed.run``) -- executes here with a live process group; at world size 1 the collectives must
leave every bit of their buffers unchanged, so every result equals the group-less one.  The world-size-2 semantics (uneven shards, replicas kept in sync) are covered on
CPU with gloo (tests/test_dist_cpu.py).  Reference: DDP of train.py:294-301; the pixel gather has no counterpart there.
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist

import common
import scenes
import nsff_pl_amd as A
from nsff_pl_amd import dist as ndist, evaluate

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture()
def nccl_world1(monkeypatch):
    """A live one-rank RCCL group, created the way a launcher's environment asks for it."""
    assert not dist.is_initialized()
    tmp = tempfile.mkdtemp(prefix="nsff_rdzv_test_")
    for k, v in dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv(ndist.INIT_ENV, "file://" + os.path.join(tmp, "rdzv"))
    rank, world, device = ndist.init_from_env()
    assert (rank, world, str(device)) == (0, 1, DEV) and dist.is_initialized() and dist.get_backend() == "nccl"
    yield device
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _scene(name="g4_nsff_test"):
    cfg, meta, rays, ts, models, emb, dataset, want = common.build_case(name, A.NeRF, A.PosEmbedding)
    for m in list(models.values()) + [emb[k] for k in ("t", "a") if k in emb]:
        m.to(DEV)
    return cfg, rays.to(DEV), ts.to(DEV), models, emb


def test_pixel_all_gather_runs_through_rccl(hip_lib, nccl_world1, monkeypatch):
    calls = []
    real = dist.all_gather_into_tensor
    monkeypatch.setattr(dist, "all_gather_into_tensor", lambda out, inp, **kw: (calls.append(tuple(inp.shape)), real(out, inp, **kw))[1])
    cfg, rays, ts, models, emb = _scene()
    out = A.render_rays(models, emb, rays, ts, scenes.N_FRAMES - 1, cfg["N_samples"], 0, 0, cfg["N_importance"],
                        test_time=True, **scenes.render_kwargs(cfg))
    keys = ("rgb_fine", "depth_fine", "transient_alpha_fine")
    for counts in (None, [rays.shape[0]]):
        merged = ndist.all_gather_pixels(out, keys, counts=counts)
        torch.cuda.synchronize()
        for k in keys:
            assert merged[k].shape == out[k].shape and torch.equal(merged[k], out[k]), k
    assert calls == [(rays.shape[0], 5)] * 2            # ONE collective per gather: rgb(3) + depth(1) + alpha(1) packed


def test_overlapped_pixel_all_gather_on_a_side_stream(hip_lib, nccl_world1, monkeypatch):
    """all_gather_pixels_async: the RCCL collective is issued from a side stream (not the render stream), the merged pixels
    equal the synchronous gather bit for bit, and wait() orders the caller's stream behind it."""
    streams = []
    real = dist.all_gather_into_tensor
    monkeypatch.setattr(dist, "all_gather_into_tensor",
                        lambda out, inp, **kw: (streams.append(torch.cuda.current_stream().cuda_stream), real(out, inp, **kw))[1])
    cfg, rays, ts, models, emb = _scene()
    keys = ("rgb_fine", "depth_fine")
    render = torch.cuda.current_stream().cuda_stream
    outs, handles = [], []
    for shift in (0, 1):                                   # two frames: the second renders while the first one's gather runs
        out = A.render_rays(models, emb, rays, (ts + shift).clamp(max=scenes.N_FRAMES - 1), scenes.N_FRAMES - 1, cfg["N_samples"],
                            0, 0, cfg["N_importance"], test_time=True, **scenes.render_kwargs(cfg))
        outs.append(out)
        handles.append(ndist.all_gather_pixels_async(out, keys))
    for out, h in zip(outs, handles):
        merged = h.wait()
        assert all(torch.equal(merged[k], out[k]) for k in keys)
        (merged["rgb_fine"] * 2).sum().item()              # consumed on the render stream right after wait()
    assert len(streams) == 2 and all(st != render for st in streams)


def test_sharded_frame_equals_the_unsharded_one(hip_lib, nccl_world1):
    cfg, _, _, models, emb = _scene()
    H, W = 36, 64
    K = torch.tensor([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]])
    c2w = torch.tensor([[1.0, 0, 0, 0.02], [0, 1.0, 0, -0.01], [0, 0, 1.0, 0.0]])
    rays = evaluate.frame_rays(K, c2w, H, W, device=DEV)
    ts = torch.full((H * W,), 7, device=DEV, dtype=torch.long)
    kw = dict(output_transient=True, output_transient_flow=[])
    args = (scenes.N_FRAMES - 1, cfg["N_samples"], cfg["N_importance"])
    whole = evaluate.render_frame(models, emb, rays, ts, *args, chunk=1000, keys=ndist.DEFAULT_PIXEL_KEYS, **kw)
    shard = evaluate.render_frame_sharded(models, emb, rays, ts, *args, chunk=1000, **kw)
    torch.cuda.synchronize()
    for k in ndist.DEFAULT_PIXEL_KEYS:
        assert torch.equal(shard[k], whole[k]), k


def _batch(tr):
    return tr._test_batch


def _train(graph, steps=3):
    from nsff_pl_amd.training import NSFFTrainer
    name = "g3_nsff_train"
    cfg, meta, rays, ts, models, emb, _, _ = common.build_case(name, A.NeRF, A.PosEmbedding)
    Ks, Ps, _ = scenes.camera_buffers()
    hp = dict(N_samples=cfg["N_samples"], N_importance=cfg["N_importance"], perturb=0, noise_std=0)
    tr = NSFFTrainer(models, emb, scenes.N_FRAMES, hp, Ks, Ps, output_transient_flow=cfg["flow"], graph=graph).to(DEV)
    tr.on_train_epoch_start(scenes.LOSS_EPOCH)
    batch = {k: v.to(DEV) for k, v in scenes.synthetic_targets(cfg["n_rays"], ts, cfg["seed"]).items()}
    batch["rays"] = rays.to(DEV)
    tr._test_batch = batch
    losses = [float(tr.step(batch)["train/loss"]) for _ in range(steps)]
    torch.cuda.synchronize()
    return losses, tr._flat_grad.detach().clone(), tr


@pytest.mark.parametrize("graph", [False, True])
def test_trainer_step_with_a_live_group_equals_the_group_less_step(graph, hip_lib, monkeypatch, request):
    """The flat gradient all-reduce sits between graph A (zero_grad .. backward) and graph B (Adam): with a process group
    alive it must be issued once per step, on the gradient buffer itself, and at world size 1 leave every bit of it
    unchanged -- so the step is the group-less step.  Compared: the loss and the gradient buffer of the FIRST step (1e-6:
    the loss reductions use atomics).  Later steps are not comparable between ANY two runs, group or not: Adam's first
    update is lr * g / (|g| + 1e-8), so last-bit noise on near-zero gradient elements becomes +-lr parameter changes
    (tools/debug/graph_determinism.py: run-to-run differences of 1e-4 in the third loss, eager and graph alike)."""
    A.set_precision("f16x3")
    try:
        want_losses, want_grad, _ = _train(graph, steps=1)
        request.getfixturevalue("nccl_world1")
        calls, unchanged = [], []
        real = dist.all_reduce

        def spy(t, *a, **kw):
            before = t.clone()
            out = real(t, *a, **kw)
            torch.cuda.synchronize()
            calls.append((t.data_ptr(), t.numel()))
            unchanged.append(bool(torch.equal(before, t)) and bool(before.abs().sum() > 0))
            return out
        monkeypatch.setattr(dist, "all_reduce", spy)
        got_losses, got_grad, tr = _train(graph, steps=1)
        assert len(calls) == 1 and calls[0] == (tr._flat_grad.data_ptr(), tr._flat_grad.numel()) and all(unchanged)
        assert abs(got_losses[0] - want_losses[0]) <= 1e-6 * abs(want_losses[0]), (got_losses, want_losses)
        assert float((got_grad - want_grad).abs().max()) <= 1e-5 * float(want_grad.abs().max())
        more = [float(tr.step(_batch(tr))["train/loss"]) for _ in range(2)]
        assert len(calls) == 3 and len(set(calls)) == 1 and all(unchanged)      # once per step, always the same flat buffer
        assert all(l == l for l in more) and more[-1] < got_losses[0]
    finally:
        A.set_precision(A.config.DEFAULT_PRECISION)


def test_bench_in_its_torchrun_form_on_one_gpu(hip_lib):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            ndist.INIT_ENV)}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-aux", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 1e6 and line["roofline"]["frac"] > 0
    pr = line["per_rank"]                                  # the RCCL gather ran inside the timed steps and was timed by itself
    assert 0 < pr["gather_ms_per_step_max"] < pr["ms_per_step_max"] <= line["ms_per_step"] * 1.001
    assert "pixel all-gather" in line["config"]["parallelism"]
    # ... and started by itself with --gpus 1 it stays group-less (the driver's N = 1 form)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-aux",
                        "--no-cpu-baseline", "--workload", "train", "--graph"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert "per_rank" not in line and line["config"]["hip_graph"] is True


@pytest.mark.parametrize("extra", [["--workload", "train"], ["--workload", "train", "--graph"], ["--workload", "eval"]])
def test_bench_other_workloads_in_their_torchrun_form_on_one_gpu(extra, hip_lib):
    """The three forms the driver may start under the launcher besides the headline: the C4 training step (the flat-gradient
    RCCL all-reduce between the step's two hipGraphs / between backward and Adam) and the strong-scaling evaluation frame
    (ray shards + the overlapped pixel all-gather) -- world size 1 over RCCL on the one GPU there is; rank 0 prints ONE line
    that carries per_rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            ndist.INIT_ENV)}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29673", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-aux", "--no-cpu-baseline", "--settle-ms", "0"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 1e5
    assert line["per_rank"]["ms_per_step_max"] <= line["ms_per_step"] * 1.001
    if "train" in extra:
        assert "all-reduce" in line["config"]["parallelism"] and line["config"]["hip_graph"] is ("--graph" in extra)
        assert line["roofline"]["kernel"] == "nsff_field_kernel_h3a_save"
        if "--graph" not in extra:       # (a replayed graph does not pass through the C-ABI's launch profiler)
            assert line["roofline"]["frac"] > 0.05
    else:
        assert line["scaling"] == "strong" and line["per_rank"]["gather_ms_per_step_max"] > 0
