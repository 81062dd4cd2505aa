"""Ray sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

Rays are independent units of ``render_rays`` (no cross-ray state, read-only weights), so
the path shards with no data-path collective: every rank renders a contiguous block of
rays with its own replica of the (<= 9.2 MB) weights.  The only exchange is one
all-gather of the rendered PIXELS -- ``rgb_fine`` (3) + ``depth_fine`` (1) [+ per-ray flows]
= 16-40 B per ray, packed into a single buffer so it is ONE collective per frame; the
per-sample tensors never leave the GPU that produced them.  The reference has no explicit
collective (PL DDP only all-reduces gradients, train.py:294-301); this module is the
renderer-side counterpart for sharded eval / multi-GPU benchmarking.

``torch.distributed`` backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""
import os

import torch
import torch.distributed as dist

DEFAULT_PIXEL_KEYS = ("rgb_fine", "depth_fine")


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if use_gpu:
            kw["device_id"] = device
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


def launch_local(n_procs, argv, master_port=None, env=None, timeout=None):
    """Start `argv` (a command line, e.g. [sys.executable, "bench.py", ...]) as `n_procs` ranks on this node, the
    way ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment, one process per GPU (``init_from_env`` picks
    ``cuda:LOCAL_RANK``).  stdout / stderr are inherited, so rank 0's single JSON line reaches the caller's stdout.
    Returns the worst exit code; if one rank fails the others are terminated."""
    import socket
    import subprocess
    import time
    if master_port is None:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            master_port = sock.getsockname()[1]
    procs = []
    for rank in range(n_procs):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_procs), LOCAL_WORLD_SIZE=str(n_procs),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
        procs.append(subprocess.Popen(list(argv), env=e))
    deadline = None if timeout is None else time.monotonic() + timeout
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:                         # a dead rank would leave the others in a collective
                        q.terminate()
            if deadline is not None and time.monotonic() > deadline:
                worst = worst or 124
                for q in pending:
                    q.terminate()
                deadline = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def shard_bounds(n_items, world, rank):
    """Contiguous, balanced block [lo, hi) of rank `rank`; blocks differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _pack(results, keys, pad_to):
    cols = [results[k].reshape(results[k].shape[0], -1).float() for k in keys]
    buf = torch.cat(cols, 1).contiguous()
    if buf.shape[0] < pad_to:
        buf = torch.cat([buf, buf.new_zeros(pad_to - buf.shape[0], buf.shape[1])], 0)
    return buf, [c.shape[1] for c in cols]


def all_gather_pixels(results, keys=DEFAULT_PIXEL_KEYS, counts=None, group=None):
    """All-gather the per-ray tensors `keys` of every rank with ONE collective.

    results: this rank's render_rays dict.  counts: rays per rank (list, len world) when
    shards are uneven; None = every rank holds the same number.  Returns {key: (sum N, ...)}.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_local = results[keys[0]].shape[0]
    if counts is None:
        counts = [n_local] * world
    pad_to = max(counts)
    buf, widths = _pack(results, keys, pad_to)
    if world == 1:
        gathered = buf[None]
    else:
        out = buf.new_empty(world * pad_to, buf.shape[1])
        dist.all_gather_into_tensor(out, buf, group=group)
        gathered = out.view(world, pad_to, buf.shape[1])
    rows = torch.cat([gathered[r, :counts[r]] for r in range(world)], 0)
    merged, c0 = {}, 0
    for k, wdt in zip(keys, widths):
        merged[k] = rows[:, c0:c0 + wdt].reshape((rows.shape[0],) + tuple(results[k].shape[1:]))
        c0 += wdt
    return merged


def render_rays_sharded(render_fn, models, embeddings, rays, ts, *args,
                        gather_keys=DEFAULT_PIXEL_KEYS, group=None, **kwargs):
    """Strong-scaling render of one ray batch: rank r renders block r, pixels are all-gathered.

    Every rank passes the SAME full `rays` / `ts`; returns (merged_pixels, local_results)
    where merged_pixels[key] covers all rays (identical on every rank) and local_results is
    this rank's full render_rays dict for its block.  `render_fn` has the signature of
    ``render_rays`` (positional args after ts are forwarded unchanged).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays.shape[0]
    bounds = [shard_bounds(n, world, r) for r in range(world)]
    lo, hi = bounds[rank]
    kw = dict(kwargs)
    for per_ray in ("view_dir", "t_embedded", "a_embedded"):
        if per_ray in kw and kw[per_ray] is not None:
            kw[per_ray] = kw[per_ray][lo:hi]
    local = render_fn(models, embeddings, rays[lo:hi], None if ts is None else ts[lo:hi], *args, **kw)
    merged = all_gather_pixels(local, gather_keys, counts=[b - a for a, b in bounds], group=group)
    return merged, local
