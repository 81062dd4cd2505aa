"""Thin NSFF trainer step around the HIP renderer.

Mirrors what ``NSFFSystem`` of the reference's ``train.py`` does per batch -- ``forward`` (:99-123, the ray
chunk loop), ``on_train_epoch_start`` (:174-176), ``training_step`` (:178-198) and the Adam / MultiStepLR
defaults of ``utils/__init__.py:24-77`` + ``opt.py`` -- without pytorch-lightning: one process per GPU, the
renderer's forward on the gfx950 kernels, backward through :mod:`nsff_pl_amd.autograd`, and (world > 1)
ONE flat RCCL all-reduce of the gradients per step instead of DDP's per-bucket hooks (the models total
2.3 M parameters = 9.2 MB: a single bucket is already far below the xGMI latency/bandwidth knee, so
splitting it to overlap with backward would only add launches).

Logging, checkpoint callbacks, validation images and hard-sampling buffers of the reference are control
plane and out of scope.
"""
from collections import defaultdict

import torch
import torch.distributed as dist

from . import field_grad
from .autograd import grad_parameters
from .losses import NeRFWLoss
from .optim import FlatAdam
from .rendering import render_rays


def psnr(image_gt, image_pred):
    """metrics.py:6-16 (no mask)."""
    return -10 * torch.log10(torch.mean((image_gt - image_pred) ** 2))


def allreduce_gradients(params, group=None):
    """Average .grad over the ranks with one flat collective (missing grads count as zeros)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat /= world
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(g).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(g))
        off += n


class NSFFTrainer:
    """models = {'fine', 'coarse'?}, embeddings = {'xyz','dir','t'?,'a'?} exactly as train.py:40-84 builds them.

    hparams (attribute or dict access): N_samples, N_importance, perturb, noise_std, chunk, lambda_geo_init,
    thickness, topk, lr, weight_decay, decay_step, decay_gamma -- reference names and defaults (opt.py) -- and
    decay_unused (False: with weight_decay > 0 a parameter that receives no gradient is left alone, as torch.optim.Adam
    leaves ``grad is None`` parameters alone; True: the plain every-element step, see optim.FlatAdam).
    """

    DEFAULTS = dict(N_samples=128, N_importance=0, perturb=1.0, noise_std=1.0, chunk=32 * 1024,
                    lambda_geo_init=0.04, thickness=1, topk=1.0, lr=5e-4, weight_decay=0.0,
                    decay_step=[20], decay_gamma=0.1, decay_unused=False)

    def __init__(self, models, embeddings, n_frames, hparams=None, Ks=None, Ps=None,
                 output_transient=True, output_transient_flow=("fw", "bw", "disocc"), graph=False, optimizer_cls=FlatAdam):
        """graph=True: the step is captured once into two hipGraphs (``torch.cuda.CUDAGraph``) and replayed: graph A =
        zero_grad + forward kernels + loss + backward kernels, graph B = Adam; between them -- outside any capture --
        the flat RCCL gradient all-reduce when world > 1.  Needs fixed batch shapes and topk == 1.
        graph="auto": decided at the first step from the batch -- replayed graphs when the step is SMALL (a step is ~100 launches
        whatever its size: below ~450 k field-point evaluations the host's launch rate, not the GPU, bounds the eager step; the
        reference's README configuration -- 512 rays x 128 samples -- takes 2.4 ms replayed against 4.3 ms eager, bench.py
        aux.readme_train), the eager step otherwise (a replayed node costs what an eager launch costs and a replay adds a fixed
        cost: the C2 step is 3-4 % faster eager); eager on CPU tensors and with topk < 1."""
        hp = dict(self.DEFAULTS)
        if hparams is not None:
            given = hparams if isinstance(hparams, dict) else vars(hparams)
            hp.update({k: v for k, v in given.items() if k in hp})
        self.hp = hp
        self.models, self.embeddings, self.n_frames = models, embeddings, n_frames
        self.output_transient = output_transient
        self.output_transient_flow = list(output_transient_flow) if output_transient else []
        self._graph_auto = isinstance(graph, str) and graph == "auto"
        if isinstance(graph, str) and not self._graph_auto:
            raise ValueError("graph must be True, False or 'auto'")
        self.graph = False if self._graph_auto else bool(graph)
        self.optimizer_cls = optimizer_cls       # (tests drive the step on CPU with a torch-op twin of FlatAdam)
        self.loss = NeRFWLoss(lambda_geo=hp["lambda_geo_init"], thickness=hp["thickness"], topk=hp["topk"],
                              static_shapes=self.graph)
        if self.output_transient_flow:                                   # train.py:136-138
            self.loss.register_buffer("Ks", Ks)
            self.loss.register_buffer("Ps", Ps)
            self.loss.max_t = n_frames - 1
        self.params = grad_parameters(models, embeddings)
        self.optimizer = None
        self.current_epoch = 0
        self._graph = self._graph_opt = self._static_batch = self._static_log = None
        self._flat_grad = None
        self._geo = None                     # device scalars the captured loss reads (lambda_geo, epoch ramp)

    def _setup_flat_grads(self):
        """Every parameter and every .grad is a view of ONE flat buffer each (optim.FlatAdam): zero_grad is a single
        memset, the data-parallel all-reduce a single collective on the gradient buffer itself (no cat / copy-back of
        ~100 tensors), Adam one launch."""
        if self.optimizer is None:
            self._make_optimizer()
        elif not self.optimizer.in_place():      # a caller replaced parameter or gradient tensors
            self.optimizer.gather()
        self._flat_grad = self.optimizer.flat_grad

    def zero_grad(self):
        self._flat_grad.zero_()

    def allreduce(self):
        """One flat RCCL all-reduce (mean) of all gradients, in place on the gradient buffer.  Issued whenever a process
        group exists -- world size 1 included, so that a one-GPU run under torchrun exercises the same collective between
        the two hipGraphs as an eight-GPU one (the sum over one rank and the division by 1 leave every bit unchanged)."""
        if dist.is_initialized():
            dist.all_reduce(self._flat_grad)
            world = dist.get_world_size()
            if world > 1:
                self._flat_grad /= world

    def _make_optimizer(self):
        hp = self.hp
        self.optimizer = self.optimizer_cls(self.params, lr=hp["lr"], eps=1e-8, weight_decay=hp["weight_decay"],
                                            decay_unused=hp["decay_unused"])
        self._flat_grad = self.optimizer.flat_grad
        self._lr_epoch = -1

    def _invalidate_packs(self):
        for m in self.models.values():          # the native step changes the weights without bumping tensor versions
            m._pack_cache.invalidate()

    def to(self, device):
        for m in self.models.values():
            m.to(device)
        for k in ("t", "a"):
            if k in self.embeddings:
                self.embeddings[k].to(device)
        self.loss.to(device)
        self._make_optimizer()               # after the move: optimizer state must live where the parameters do
        return self

    # train.py:99-123
    def forward(self, rays, ts, test_time=False, **kwargs):
        hp, results = self.hp, defaultdict(list)
        for i in range(0, rays.shape[0], hp["chunk"]):
            chunk = render_rays(self.models, self.embeddings, rays[i:i + hp["chunk"]],
                                None if ts is None else ts[i:i + hp["chunk"]], self.n_frames - 1, hp["N_samples"],
                                0 if test_time else hp["perturb"], 0 if test_time else hp["noise_std"],
                                hp["N_importance"], hp["chunk"] // 4 if test_time else hp["chunk"],
                                test_time=test_time, **kwargs)
            for k, v in chunk.items():
                results[k].append(v)
        # (a single chunk -- every training batch -- is handed on as is: torch.cat of one tensor would copy all 47 keys)
        return {k: v[0] if len(v) == 1 else torch.cat(v, 0) for k, v in results.items()}

    # train.py:174-176
    def on_train_epoch_start(self, epoch):
        self.current_epoch = epoch
        geo = self.hp["lambda_geo_init"] * 0.1 ** (epoch // 10)
        if self.graph:                        # the captured graph reads these from device scalars
            dev = self.params[0].device
            if self._geo is None:
                self._geo = torch.zeros((), device=dev)
                self._ramp = torch.zeros((), device=dev)
            self._geo.fill_(geo)
            self._ramp.fill_(min(epoch / 10, 1.0))
            self.loss.lambda_geo_d = self.loss.lambda_geo_f = self._geo
        else:
            self.loss.lambda_geo_d = self.loss.lambda_geo_f = geo

    # train.py:178-198
    def training_step(self, batch):
        kwargs = dict(output_transient=self.output_transient, output_transient_flow=self.output_transient_flow)
        results = self.forward(batch["rays"], batch.get("ts"), **kwargs)
        if self.graph:
            kwargs["epoch_ramp"] = self._ramp
        loss_d = self.loss(results, batch, epoch=self.current_epoch, **kwargs)
        loss = loss_d.total() if hasattr(loss_d, "total") else sum(loss_d.values())
        with torch.no_grad():
            log = {f"train/{k}": v.detach() for k, v in loss_d.items()}
            log["train/loss"] = loss.detach()
            log["train/psnr"] = psnr(results["rgb_fine"].detach(), batch["rgbs"])
            log["lr"] = self.optimizer.param_groups[0]["lr"]
        return loss, log

    # train.py:200-214 (the metric part; image grids / SSIM maps of the reference are logging)
    @torch.no_grad()
    def validation_step(self, batch):
        """batch: {'rays': (H*W,6), 'rgbs': (H*W,3) [, 'ts']} of one full frame -> {'val_psnr'}."""
        kwargs = dict(output_transient=self.output_transient, output_transient_flow=[])
        results = self.forward(batch["rays"], batch.get("ts"), test_time=True, **kwargs)
        return {"val_psnr": psnr(results["rgb_fine"], batch["rgbs"])}

    def step(self, batch):
        """zero_grad -> training_step -> backward -> gradient all-reduce -> Adam; returns the log dict."""
        if self.optimizer is None:
            self._make_optimizer()
        if self._graph_auto:
            self._resolve_graph(batch)
        if self.graph:
            return self._graph_step(batch)
        self._setup_flat_grads()
        field_grad.drop_stale_pending()
        self.zero_grad()
        loss, log = self.training_step(batch)
        with field_grad.deferred_weight_grads():
            loss.backward()
        self.allreduce()
        self.optimizer.step()
        self._invalidate_packs()
        return log

    AUTO_GRAPH_POINT_EVALS = 450_000        # (see __init__: where the eager step stops being bound by the host)

    def _resolve_graph(self, batch):
        """graph='auto', first step: field-point evaluations of a step = rays x [coarse samples + (fine samples) x (1 + two
        scene-flow re-queries)]."""
        self._graph_auto = False
        hp, rays = self.hp, batch["rays"]
        fine = hp["N_samples"] + (2 if self.output_transient else 1) * hp["N_importance"]
        evals = rays.shape[0] * ((hp["N_samples"] if hp["N_importance"] > 0 else 0) + fine * (3 if self.output_transient_flow else 1))
        self.graph = bool(rays.is_cuda and hp["topk"] >= 1 and "weights" not in batch and evals <= self.AUTO_GRAPH_POINT_EVALS)
        if self.graph:
            self.loss.static_shapes = True
            self.on_train_epoch_start(self.current_epoch)        # (the captured loss reads lambda_geo / the ramp from device scalars)

    def _graph_body_backward(self):
        self.zero_grad()
        loss, log = self.training_step(self._static_batch)
        with field_grad.deferred_weight_grads():
            loss.backward()
        return log

    def _graph_step(self, batch):
        if self._graph is None:
            if self._geo is None:
                self.on_train_epoch_start(self.current_epoch)
            self._static_batch = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
            self._setup_flat_grads()
            keep = [p.detach().clone() for p in self.params]
            keep_opt = self.optimizer.state_dict()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):              # warm-up off the capture stream (allocator, pack caches, Adam state)
                for _ in range(3):
                    self._graph_body_backward()
                    self.optimizer.step()
                    self._invalidate_packs()
            torch.cuda.current_stream().wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_log = self._graph_body_backward()
            self._graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_opt):
                self.optimizer.step()
            # the warm-up steps were not part of the schedule: undo them (parameters and Adam moments / step count)
            with torch.no_grad():
                for p, k in zip(self.params, keep):
                    p.copy_(k)
            self.optimizer.load_state_dict(keep_opt)
        for k, v in self._static_batch.items():
            v.copy_(batch[k])
        self._graph.replay()
        self.allreduce()                        # outside the captures: RCCL is not captured
        self._graph_opt.replay()
        self._invalidate_packs()
        return self._static_log

    # utils/__init__.py:82-104 + PL checkpoint layout (train.py:55,59,76,87): nerf_fine. / nerf_coarse. / embedding_t. / embedding_a.
    def checkpoint(self):
        """{'state_dict': reference-style prefixed model / embedding tensors (cloned), 'optimizer': torch.optim.Adam format,
        'epoch'} -- independent of the flat training buffers, so it can be saved or kept while training continues."""
        from .optim import detached_state
        sd = {}
        for typ, m in self.models.items():
            sd.update({f"nerf_{typ}.{k}": v for k, v in detached_state(m).items()})
        for k in ("t", "a"):
            if k in self.embeddings:
                sd.update({f"embedding_{k}.{kk}": v for kk, v in detached_state(self.embeddings[k]).items()})
        opt = None                                  # (before the first step / to(): no optimizer state exists yet)
        if self.optimizer is not None:
            opt = self.optimizer.torch_state_dict() if hasattr(self.optimizer, "torch_state_dict") else self.optimizer.state_dict()
        return {"state_dict": sd, "optimizer": opt, "epoch": self.current_epoch}

    def load_checkpoint(self, ckpt, strict=False, prefixes_to_ignore=()):
        """Inverse of :meth:`checkpoint`; also takes a reference (Lightning) checkpoint: ``state_dict`` with the same
        prefixes and the ``torch.optim.Adam`` state under ``optimizer_states[0]`` (train.py:279-290).  Like the reference's
        ``load_ckpt`` (utils/__init__.py:82-104: ``load_state_dict(..., strict=False)``) tensors the checkpoint does not hold
        keep their current values and ``prefixes_to_ignore`` drops checkpoint entries by (un-prefixed) key prefix;
        ``strict=True`` raises KeyError on a missing tensor instead.  Returns the list of keys that were not loaded."""
        sd = ckpt["state_dict"]
        missing = []

        def load_into(module, prefix):
            sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
            sub = {k: v for k, v in sub.items() if not any(k.startswith(ig) for ig in prefixes_to_ignore)}
            for k, p in module.state_dict().items():
                if k in sub:
                    p.copy_(sub[k])                 # in place: parameters stay views of the flat buffer
                elif strict:
                    raise KeyError(f"checkpoint has no tensor {prefix}{k}")
                else:
                    missing.append(prefix + k)
        with torch.no_grad():
            for typ, m in self.models.items():
                load_into(m, f"nerf_{typ}.")
            for name in ("t", "a"):
                if name in self.embeddings:
                    load_into(self.embeddings[name], f"embedding_{name}.")
        if self.optimizer is None and self.params[0].is_cuda:
            self._make_optimizer()
        opt = ckpt.get("optimizer")
        if opt is None and ckpt.get("optimizer_states"):         # Lightning's layout: one entry per optimizer
            opt = ckpt["optimizer_states"][0]
        if opt is not None and self.optimizer is not None:
            if "param_groups" in opt:
                try:
                    self.optimizer.load_torch_state_dict(opt)
                except ValueError as e:                          # another parameter list (e.g. a partial warm start)
                    import warnings
                    warnings.warn(f"load_checkpoint: optimizer state not taken over ({e}); Adam moments start from zero")
            else:
                self.optimizer.load_state_dict(opt)
        elif opt is not None:
            import warnings
            warnings.warn("load_checkpoint: the checkpoint holds optimizer state but no optimizer exists yet (call .to(device) "
                          "first); Adam moments will start from zero")
        self.current_epoch = int(ckpt.get("epoch", self.current_epoch))
        self._invalidate_packs()
        return missing

    def on_train_epoch_end(self):
        """MultiStepLR(milestones=decay_step, gamma=decay_gamma) of train.py:143-146, stepped once per epoch."""
        if self.optimizer is not None and (self.current_epoch + 1) in list(self.hp["decay_step"]):
            self.optimizer.lr.mul_(self.hp["decay_gamma"])
