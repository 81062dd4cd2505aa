"""``render_rays`` as a replayed hipGraph (inference): one capture per batch size, then every call is one graph launch.

The render path has no host synchronisation and no data-dependent shape, so a call with fixed flags and a
fixed number of rays is a fixed sequence of ~23 launches -- the generator kernels of the reference's draw order included.
``GraphedRender`` captures that sequence once (``torch.cuda.CUDAGraph``: HIP stream capture) with static input buffers and replays
it; what a caller on a slow or busy host gains is the launch overhead of the whole sequence.  Values are those of the eager call
(the same kernels on the same inputs; ``tests/test_graphs.py``).

Contract: results are the graph's own output tensors -- valid until the next call with the same batch size (copy what must
outlive it; ``evaluate.render_frame(graph=...)`` does).  Weights may change between calls: the packed weight buffers keep their
addresses and are refreshed eagerly before the replay.  The graph also holds RAW PARAMETER ADDRESSES (``nsff_time_bias`` reads the
dynamic trunk's input-layer weights, the embedding gathers read their tables): the addresses of every model / embedding parameter
are part of the cache key, so a caller that re-points parameters (``FlatAdam.adopt``, ``model.to()``, ``load_state_dict(assign=True)``)
gets a fresh capture instead of a replay that reads freed memory.  Per-ray keyword tensors (``view_dir``, ``t_embedded``, ``a_embedded``) are
not supported -- they would have to be static buffers too; pass the plain arguments.
"""
import torch

from . import config
from .rendering import render_rays


class GraphedRender:
    def __init__(self, models, embeddings, max_t, N_samples=64, perturb=0, noise_std=0, N_importance=0, chunk=1024 * 32,
                 test_time=False, **kwargs):
        for k in ("view_dir", "t_embedded", "a_embedded"):
            if kwargs.get(k) is not None:
                raise ValueError(f"GraphedRender: per-ray keyword tensor '{k}' is not supported (static shapes only)")
        self.models, self.embeddings = models, embeddings
        self.args = (max_t, N_samples, perturb, noise_std, N_importance, chunk)
        self.test_time, self.kwargs = bool(test_time), dict(kwargs)
        self._graphs = {}                # (n_rays, has_ts, precision, tile, parameter addresses) -> (graph, rays, ts, results)

    def _param_addresses(self):
        """Addresses of every parameter a captured launch may read directly (not through the refreshed packs)."""
        mods = list(self.models.values()) + [m for m in self.embeddings.values() if isinstance(m, torch.nn.Module)]
        return tuple(p.data_ptr() for m in mods for p in m.parameters())

    def _refresh_packs(self):
        """Packed weights live in buffers whose addresses the graphs hold: re-pack (eagerly, in place) whatever changed."""
        prec = config.PRECISIONS[config.get_precision()]
        for m in self.models.values():
            m.packed(prec)

    def _capture(self, key, rays, ts):
        dev = rays.device
        s_rays = rays.detach().clone().contiguous().float()
        s_ts = None if ts is None else ts.detach().clone()

        def call():
            with torch.no_grad():
                return render_rays(self.models, self.embeddings, s_rays, s_ts, *self.args, test_time=self.test_time, **self.kwargs)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up off the capture: packs, caches, allocator pools
            call()
            call()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            results = call()
        self._graphs[key] = (graph, s_rays, s_ts, results)
        return self._graphs[key]

    def __call__(self, rays, ts=None):
        if not rays.is_cuda:
            raise RuntimeError("GraphedRender needs GPU tensors (there is no CPU path)")
        # (the launch form is part of the key: a graph captured with persistent launches keeps replaying them, so a caller inside
        # config.launch_form(persistent=False) -- a sharded frame loop beside a collective -- gets a capture of its own)
        shape_key = (int(rays.shape[0]), ts is not None, config.get_precision(), config.get_tile_points(), config.get_persistent())
        key = shape_key + (self._param_addresses(),)
        entry = self._graphs.get(key)
        if entry is None:
            for stale in [k for k in self._graphs if k[:5] == shape_key]:      # captured against parameters that moved since
                del self._graphs[stale]
            entry = self._capture(key, rays, ts)
        graph, s_rays, s_ts, results = entry
        s_rays.copy_(rays)
        if s_ts is not None:
            s_ts.copy_(ts)
        self._refresh_packs()
        graph.replay()
        return results
