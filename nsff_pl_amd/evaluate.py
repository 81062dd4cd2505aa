"""Frame-level callers of ``render_rays``: on-device ray generation, the chunked evaluation loop
and PSNR -- the renderer-side pieces of the reference's ``eval.py`` / ``datasets/monocular.py``.

* :func:`frame_rays`    -- NDC rays of a pinhole frame generated on the GPU
                           (reference datasets/ray_utils.py:7-106 as called at monocular.py:268-276);
                           the reference builds them on the CPU and uploads 3.5 MB per frame.
* :func:`render_frame`  -- the ray-chunk loop of ``eval.f`` (eval.py:81-110): ``test_time=True``,
                           ``perturb = noise_std = 0``, per-key concatenation.  Results stay on the GPU and
                           ``keys=`` selects what is kept (the reference copies EVERY key of every chunk
                           to the host with a blocking ``.cpu()``, eval.py:106-107; ``to_cpu=True`` does that).
                           ``to_host=`` is the asynchronous form of the same egress (row N4): the kept keys of
                           chunk i travel to PINNED host buffers on a copy stream while chunk i+1 renders.
* :func:`render_frame_sharded` -- the same frame split across the ranks of ``torch.distributed``
                           with one pixel all-gather (:mod:`nsff_pl_amd.dist`).
* :func:`psnr`          -- metrics.py:6-16.
"""
import torch

from . import _lib
from . import dist as ndist
from .rendering import render_rays


def frame_rays(K, c2w, H, W, near=1.0, device="cuda", first_pixel=0, n_pixels=None):
    """(n_pixels, 6) NDC rays of the frame with intrinsics K (3,3) and pose c2w (3,4).

    Same conventions as the reference dataset: pixel order row-major, no +0.5 centring,
    ``shift_near = -min(-1, c2w[2,3])`` (monocular.py:270).
    """
    K = torch.as_tensor(K, dtype=torch.float32).cpu()
    c2w = torch.as_tensor(c2w, dtype=torch.float32).cpu()
    n = H * W - first_pixel if n_pixels is None else n_pixels
    shift_near = -min(-1.0, float(c2w[2, 3]))
    rays = torch.empty(n, 6, device=device, dtype=torch.float32)
    with torch.cuda.device(rays.device):
        _lib.frame_rays([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], c2w.reshape(-1).tolist(), H, W, near, shift_near,
                        first_pixel, n, rays)
    return rays


_COPY_STREAM = {}


class PinnedPool:
    """Page-locked host buffers for ``render_frame(..., to_host=pool)``, reused frame after frame (hipHostMalloc is slow)
    WITHOUT handing the same memory to two live frames: the pool keeps ``depth`` buffer sets and rotates through them, so
    a returned :class:`HostFrame` stays valid until ``depth`` further frames have been rendered into the same pool
    (``depth=2``: frames t and t+1 of a time interpolation can be held together).  Before a set is written again the pool
    waits for the copies of the frame that last used it, so ``sync=False`` frames never race each other."""

    def __init__(self, depth=2):
        if depth < 1:
            raise ValueError("PinnedPool depth must be >= 1")
        self.depth, self._turn = int(depth), 0
        self._sets = [dict() for _ in range(self.depth)]     # slot -> {(key, shape): tensor}
        self._last = [None] * self.depth                     # slot -> the HostFrame that was handed this set

    def next_slot(self):
        slot = self._turn % self.depth
        self._turn += 1
        old = self._last[slot]
        if old is not None:
            old.wait()                                        # its copies must have landed before the set is rewritten
            old.stale = True
        return slot

    def buffer(self, slot, key, shape):
        ck = (key, tuple(shape))
        bufs = self._sets[slot]
        if ck not in bufs:
            bufs[ck] = torch.empty(*shape, dtype=torch.float32, pin_memory=True)
        return bufs[ck]


class HostFrame(dict):
    """Result of ``render_frame(..., to_host=...)``: {key: pinned host tensor}.  The device-to-host copies may still be
    in flight on the copy stream; ``wait()`` (or ``render_frame(..., sync=True)``, the default) blocks until they landed.
    ``stale`` turns True once a :class:`PinnedPool` has handed this frame's buffers to a later frame."""
    event = None
    stale = False

    def wait(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None
        return self


@torch.no_grad()
def render_frame(models, embeddings, rays, ts, max_t, N_samples, N_importance, chunk=1024 * 32,
                 keys=None, to_cpu=False, to_host=None, sync=True, graph=None, **kwargs):
    """Batched inference on the rays of one frame (reference eval.py:81-110).

    keys: iterable of result keys to keep (default: all, like the reference).
    to_cpu=True: the reference's egress -- a blocking ``.cpu()`` of every kept key after every chunk (eval.py:106-107).
    to_host=True | PinnedPool | {key: pinned tensor}: asynchronous egress -- the kept keys of each chunk are copied into
    frame-sized page-locked host buffers on a side stream while the next chunk renders; returns a :class:`HostFrame` of
    host tensors (``sync=False`` leaves the last copies in flight: call ``.wait()``).  ``True`` allocates FRESH buffers
    for this call (no two frames ever share memory); a :class:`PinnedPool` reuses ``depth`` rotating buffer sets (a frame
    is valid until ``depth`` later frames went through the pool); a dict supplies caller-owned buffers per key.
    graph: a :class:`nsff_pl_amd.graphs.GraphedRender` built with this call's flags (test_time=True, perturb = noise_std = 0):
    every chunk is one replayed hipGraph (one capture per distinct chunk size); kept tensors are copied out of the graph's
    buffers, so the returned frame is the caller's.
    """
    B = rays.shape[0]
    results = {}
    host, copy_stream = None, None
    if to_host:
        dev = rays.device
        if dev not in _COPY_STREAM:
            _COPY_STREAM[dev] = torch.cuda.Stream(device=dev)
        copy_stream = _COPY_STREAM[dev]
        host = HostFrame()
        given = to_host if isinstance(to_host, dict) else {}
        pool = to_host if isinstance(to_host, PinnedPool) else None
        slot = pool.next_slot() if pool is not None else None
    for i in range(0, B, chunk):
        kw = dict(kwargs)
        for per_ray in ("view_dir", "t_embedded", "a_embedded"):
            if per_ray in kw and kw[per_ray] is not None:
                kw[per_ray] = kw[per_ray][i:i + chunk]
        if graph is not None:
            out = graph(rays[i:i + chunk], None if ts is None else ts[i:i + chunk])
            out = {k: v.clone() for k, v in out.items() if keys is None or k in keys}    # (the graph owns its outputs)
        else:
            out = render_rays(models, embeddings, rays[i:i + chunk], None if ts is None else ts[i:i + chunk],
                              max_t, N_samples, 0, 0, N_importance, chunk, test_time=True, **kw)
        kept = {k: v for k, v in out.items() if keys is None or k in keys}
        if host is not None:
            done = torch.cuda.Event()
            done.record()                                     # this chunk's kernels, on the render stream
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                for k, v in kept.items():
                    if k not in host:
                        shape = (B,) + tuple(v.shape[1:])
                        if k in given:
                            host[k] = given[k]
                        elif pool is not None:
                            host[k] = pool.buffer(slot, k, shape)
                        else:
                            host[k] = torch.empty(*shape, dtype=torch.float32, pin_memory=True)
                        if tuple(host[k].shape) != shape or not host[k].is_pinned():
                            raise ValueError(f"to_host['{k}'] must be a pinned float32 tensor of shape {shape}")
                    host[k][i:i + v.shape[0]].copy_(v, non_blocking=True)
                    v.record_stream(copy_stream)              # the allocator must not recycle it before the copy ran
            continue
        for k, v in kept.items():
            results.setdefault(k, []).append(v.cpu() if to_cpu else v)
    if host is not None:
        host.event = torch.cuda.Event()
        host.event.record(copy_stream)
        if pool is not None:
            pool._last[slot] = host
        return host.wait() if sync else host
    return {k: v[0] if len(v) == 1 else torch.cat(v, 0) for k, v in results.items()}


@torch.no_grad()
def render_frame_sharded(models, embeddings, rays, ts, max_t, N_samples, N_importance, chunk=1024 * 32,
                         gather_keys=ndist.DEFAULT_PIXEL_KEYS, **kwargs):
    """Every rank passes the full frame; rank r renders block r; pixels are all-gathered (one collective)."""
    def fn(models_, embeddings_, rays_, ts_, *a, **kw):
        return render_frame(models_, embeddings_, rays_, ts_, max_t, N_samples, N_importance, chunk,
                            keys=gather_keys, **kw)
    merged, _ = ndist.render_rays_sharded(fn, models, embeddings, rays, ts, gather_keys=gather_keys, **kwargs)
    return merged


def render_sequence_sharded(models, embeddings, samples, max_t, N_samples, N_importance, img_wh, chunk=1024 * 32,
                            gather_keys=ndist.DEFAULT_PIXEL_KEYS, **kwargs):
    """The plain frame loop (``interp == 0``) of :func:`render_sequence` with the rays of every frame sharded over the ranks:
    rank r renders block r of each frame, and the ONE pixel all-gather of frame k runs on a side stream while frame k + 1
    renders (:func:`nsff_pl_amd.dist.all_gather_pixels_async`).  Yields ``(name, rgb (h,w,3), depth (h,w))``
    with the complete image on every rank, values clipped like eval.py:218-222 -- one frame behind the renders."""
    import torch.distributed as dist
    w, h = img_wh
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    pending = None

    def emit(item):
        name, handle = item
        px = handle.wait()
        return name, torch.clip(px['rgb_fine'].view(h, w, 3), 0, 1), px['depth_fine'].view(h, w)
    for i, sample in enumerate(samples):
        rays, ts = sample['rays'], sample.get('ts')
        bounds = [ndist.shard_bounds(rays.shape[0], world, r) for r in range(world)]
        lo, hi = bounds[rank]
        kw = dict(kwargs)
        for per_ray in ("view_dir", "t_embedded", "a_embedded"):
            if per_ray in kw and kw[per_ray] is not None:
                kw[per_ray] = kw[per_ray][lo:hi]
        # (the scope is entered per frame, not around the generator's yields: the caller's own launches between two frames keep
        # the process default, and an abandoned generator leaves nothing changed)
        with ndist.beside_a_collective():
            local = render_frame(models, embeddings, rays[lo:hi], None if ts is None else ts[lo:hi], max_t, N_samples,
                                 N_importance, chunk, keys=gather_keys, **kw)
        handle = ndist.all_gather_pixels_async(local, gather_keys, counts=[b - a for a, b in bounds])
        if pending is not None:
            yield emit(pending)                     # frame i - 1: its gather ran beside this frame's render
        pending = (f"{i:03d}", handle)
    if pending is not None:
        yield emit(pending)


def render_sequence(models, embeddings, samples, max_t, N_samples, N_importance, img_wh, chunk=1024 * 32, interp=0,
                    K=None, **kwargs):
    """The frame loop of the reference's ``eval.py`` (:171-222) as a generator of ``(name, rgb (h,w,3), depth (h,w))``
    GPU tensors, values clipped like there.

    samples: sequence of dicts ``{'rays': (h*w,6) [, 'ts': (h*w,) long] [, 'c2w': (3,4)]}`` (what the dataset yields).
    interp == 0: one image per sample (splits ``test`` / ``test_spiral``), named ``'{i:03d}'``.
    interp  > 0: the fixed-view time interpolation of split ``test_fixview*_interp{N}`` (:176-213): frame i is rendered
    once, re-used as the left end of the next pair (``last_results``), frame i+1 is rendered at ``ts + 1`` and
    ``interp - 1`` in-between images come from :func:`nsff_pl_amd.interpolate`; names ``'{i:03d}_{int(dt*100):03d}'``;
    the last sample only closes the sequence.  ``kwargs`` must then ask for ``output_transient_flow=['fw','bw']``."""
    from .interpolation import interpolate
    w, h = img_wh
    n = len(samples)
    last = None
    for i, sample in enumerate(samples):
        rays = sample['rays']
        ts = sample.get('ts')
        if interp > 0 and i == n - 1:                                   # eval.py:172-180: last frame closes the sequence
            yield f"{i:03d}_000", torch.clip(last['rgb_fine'].view(h, w, 3), 0, 1), last['depth_fine'].view(h, w)
            return
        res = last if last is not None else render_frame(models, embeddings, rays, ts, max_t, N_samples, N_importance,
                                                         chunk, **kwargs)
        if interp > 0:
            nxt = render_frame(models, embeddings, rays, ts + 1, max_t, N_samples, N_importance, chunk, **kwargs)
            for j in range(interp):
                dt = j / interp
                if j == 0:
                    img, depth = res['rgb_fine'].view(h, w, 3), res['depth_fine'].view(h, w)
                else:
                    img, depth = interpolate(res, nxt, dt, K, sample['c2w'], (w, h))
                yield f"{i:03d}_{int(dt * 100):03d}", torch.clip(img, 0, 1), depth
            last = nxt
        else:
            yield f"{i:03d}", torch.clip(res['rgb_fine'].view(h, w, 3), 0, 1), res['depth_fine'].view(h, w)


def psnr(image_pred, image_gt, valid_mask=None):
    """-10 log10(mean squared error) (reference metrics.py:6-16)."""
    err = (image_pred - image_gt) ** 2
    if valid_mask is not None:
        err = err[valid_mask]
    return -10 * torch.log10(err.mean())
