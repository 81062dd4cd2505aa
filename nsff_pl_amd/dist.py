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


INIT_ENV = "NSFF_DIST_INIT"          # optional rendezvous URL for init_from_env (launch_local passes a file:// one)


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, device).

    The group is created whenever a launcher set WORLD_SIZE -- also for WORLD_SIZE=1, so that a one-GPU
    ``torchrun --nproc-per-node 1`` run drives the same RCCL calls (communicator bound to ``cuda:LOCAL_RANK`` through
    ``device_id``) as an eight-GPU one.  A plain ``python script.py`` (no WORLD_SIZE) stays group-less.
    ``NSFF_DIST_INIT`` (a ``file://`` / ``tcp://`` URL) replaces the MASTER_ADDR / MASTER_PORT rendezvous."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if (world > 1 or "WORLD_SIZE" in os.environ) and not dist.is_initialized():
        kw = {}
        if os.environ.get(INIT_ENV):
            kw["init_method"] = os.environ[INIT_ENV]
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if use_gpu and (backend in (None, "nccl")):
            kw["device_id"] = device
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


def launch_local(n_procs, argv, master_port=None, env=None, timeout=None, rocm_dmabuf_ipc=True):
    """Start `argv` (a command line, e.g. [sys.executable, "bench.py", ...]) as `n_procs` ranks on this node, the
    way ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would: RANK / LOCAL_RANK / WORLD_SIZE in the
    environment, one process per GPU (``init_from_env`` picks ``cuda:LOCAL_RANK``).  The ranks meet through a fresh
    FILE (``NSFF_DIST_INIT=file://...`` in a private temporary directory) -- no port to pick, so nothing another process
    or a second concurrent launch could grab between choosing and binding it; ``master_port`` selects the classic
    MASTER_ADDR=127.0.0.1 / MASTER_PORT rendezvous instead.  stdout / stderr are inherited, so rank 0's single JSON line
    reaches the caller's stdout.  Returns the worst exit code; if one rank fails the others are terminated.
    rocm_dmabuf_ipc: default ``HSA_ENABLE_IPC_MODE_LEGACY=0`` into the ranks' environment when the caller's does not set
    it (ROCm hosts whose driver only offers dmabuf IPC -- RCCL fails with hipIpcGetMemHandle errors otherwise)."""
    import shutil
    import subprocess
    import tempfile
    import time
    tmpdir = None
    if master_port is None:
        tmpdir = tempfile.mkdtemp(prefix="nsff_rdzv_")
    procs = []
    for rank in range(n_procs):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_procs), LOCAL_WORLD_SIZE=str(n_procs))
        if tmpdir is not None:
            e[INIT_ENV] = "file://" + os.path.join(tmpdir, "rendezvous")
        else:
            e.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port))
        if rocm_dmabuf_ipc and getattr(torch.version, "hip", None):
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
        if tmpdir is not None:
            shutil.rmtree(tmpdir, ignore_errors=True)
    return worst


def shard_bounds(n_items, world, rank):
    """Contiguous, balanced block [lo, hi) of rank `rank`; blocks differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _pack(results, keys, pad_to):
    """One (pad_to, sum of widths) buffer holding the per-ray tensors `keys` side by side (rows past this rank's count: 0)."""
    cols = [results[k].reshape(results[k].shape[0], -1) for k in keys]
    widths = [c.shape[1] for c in cols]
    n = cols[0].shape[0]
    if n == pad_to and all(c.dtype == torch.float32 for c in cols):
        return torch.cat(cols, 1), widths                     # even shards: ONE launch
    buf = (cols[0].new_empty if n == pad_to else cols[0].new_zeros)((pad_to, sum(widths)), dtype=torch.float32)
    c0 = 0
    for c, w in zip(cols, widths):
        buf[:n, c0:c0 + w] = c
        c0 += w
    return buf, widths


def _unpack(out, results, keys, widths, counts, pad_to, world):
    if all(c == pad_to for c in counts):
        rows = out                                            # even shards: the gathered buffer already is the row list
    else:
        gathered = out.view(world, pad_to, out.shape[1])
        rows = torch.cat([gathered[r, :counts[r]] for r in range(world)], 0)
    merged, c0 = {}, 0
    for k, wdt in zip(keys, widths):
        merged[k] = rows[:, c0:c0 + wdt].reshape((rows.shape[0],) + tuple(results[k].shape[1:]))
        c0 += wdt
    return merged


def all_gather_pixels(results, keys=DEFAULT_PIXEL_KEYS, counts=None, group=None):
    """All-gather the per-ray tensors `keys` of every rank with ONE collective.

    results: this rank's render_rays dict.  counts: rays per rank (list, len world) when
    shards are uneven; None = every rank holds the same number.  Returns {key: (sum N, ...)}.
    The collective runs whenever a process group exists (world size 1 included: same call path on one GPU as on eight).
    Synchronous on the caller's stream; :func:`all_gather_pixels_async` is the overlapped form.
    """
    live = dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    n_local = results[keys[0]].shape[0]
    if counts is None:
        counts = [n_local] * world
    pad_to = max(counts)
    buf, widths = _pack(results, keys, pad_to)
    if not live:
        out = buf
    else:
        out = buf.new_empty(world * pad_to, buf.shape[1])
        dist.all_gather_into_tensor(out, buf, group=group)
    return _unpack(out, results, keys, widths, counts, pad_to, world)


_GATHER_STREAM = {}


def beside_a_collective(group=None):
    """Scope for a frame loop whose pixel gathers overlap the NEXT frame's render (:func:`all_gather_pixels_async`):
    ``with dist.beside_a_collective(): <render frames, start gathers, join them>``.  At world sizes above one the field
    launches issued inside take the one-workgroup-per-tile form (``config.launch_form(persistent=False)`` -- the per-call
    ``NsffFieldArgs::launch_form``): the RCCL kernel waits for its peers BESIDE the render stream, and a persistent launch
    would hold every compute unit until it ends.  The previous setting is restored on exit; at world size one (or with
    ``NSFF_PERSIST_MULTI=1``, the A/B switch for a multi-GPU node) nothing changes.  The choice is made here, once, where the
    process group is known -- not inside the gather: the first frame's launches already have the form of the later ones, and
    a graph captured inside the scope is keyed on it (``graphs.GraphedRender``)."""
    from . import config
    multi = dist.is_initialized() and dist.get_world_size(group) > 1 and not os.environ.get("NSFF_PERSIST_MULTI")
    return config.launch_form(persistent=False if multi else None)


class PendingPixels:
    """Handle of an overlapped pixel all-gather (:func:`all_gather_pixels_async`); ``wait()`` returns the merged dict and
    makes the caller's current stream (GPU) / thread (CPU) wait for the collective -- not before."""

    def __init__(self, finish):
        self._finish, self._merged = finish, None

    def wait(self):
        if self._finish is not None:
            self._merged, self._finish = self._finish(), None
        return self._merged


def all_gather_pixels_async(results, keys=DEFAULT_PIXEL_KEYS, counts=None, group=None, events=None):
    """:func:`all_gather_pixels` off the render stream ("on a side stream overlapped with the next
    frame's render"): pack, collective and unpack are enqueued on a per-device side stream behind an event dependency on
    the caller's stream, so the caller can go on enqueueing the next frame's kernels at once; the returned
    :class:`PendingPixels` joins the side stream into the then-current stream when ``wait()`` is called.  Values are
    bit-identical to the synchronous form.  CPU tensors (gloo tests): the collective is issued with ``async_op=True`` and
    completed in ``wait()``.  events: optional (start, end) ``torch.cuda.Event`` pair recorded on the side stream around the
    gather (bench.py reports the collective's own time from them)."""
    ref = results[keys[0]]
    if not ref.is_cuda:
        live = dist.is_initialized()
        world = dist.get_world_size(group) if live else 1
        if counts is None:
            counts = [ref.shape[0]] * world
        pad_to = max(counts)
        buf, widths = _pack(results, keys, pad_to)
        out, work = buf, None
        if live:
            out = buf.new_empty(world * pad_to, buf.shape[1])
            work = dist.all_gather_into_tensor(out, buf, group=group, async_op=True)

        def finish_cpu():
            if work is not None:
                work.wait()
            return _unpack(out, results, keys, widths, counts, pad_to, world)
        return PendingPixels(finish_cpu)
    dev = ref.device
    if dev not in _GATHER_STREAM:
        _GATHER_STREAM[dev] = torch.cuda.Stream(device=dev)
    side, cur = _GATHER_STREAM[dev], torch.cuda.current_stream(dev)
    side.wait_stream(cur)                                   # the pixels are complete where the caller's stream stands now
    with torch.cuda.stream(side):
        if events is not None:
            events[0].record(side)
        merged = all_gather_pixels(results, keys, counts, group)      # RCCL orders itself behind the CURRENT (= side) stream
        done = events[1] if events is not None else torch.cuda.Event()
        done.record(side)
    for k in keys:
        results[k].record_stream(side)                      # (read by the side stream: not to be recycled under it)

    def finish_gpu():
        now = torch.cuda.current_stream(dev)
        now.wait_event(done)
        for v in merged.values():
            v.record_stream(now)
        return merged
    return PendingPixels(finish_gpu)


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
