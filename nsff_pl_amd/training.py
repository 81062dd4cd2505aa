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

from .autograd import grad_parameters
from .losses import NeRFWLoss
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
    thickness, topk, lr, weight_decay, decay_step, decay_gamma -- reference names and defaults (opt.py).
    """

    DEFAULTS = dict(N_samples=128, N_importance=0, perturb=1.0, noise_std=1.0, chunk=32 * 1024,
                    lambda_geo_init=0.04, thickness=1, topk=1.0, lr=5e-4, weight_decay=0.0,
                    decay_step=[20], decay_gamma=0.1)

    def __init__(self, models, embeddings, n_frames, hparams=None, Ks=None, Ps=None,
                 output_transient=True, output_transient_flow=("fw", "bw", "disocc")):
        hp = dict(self.DEFAULTS)
        if hparams is not None:
            given = hparams if isinstance(hparams, dict) else vars(hparams)
            hp.update({k: v for k, v in given.items() if k in hp})
        self.hp = hp
        self.models, self.embeddings, self.n_frames = models, embeddings, n_frames
        self.output_transient = output_transient
        self.output_transient_flow = list(output_transient_flow) if output_transient else []
        self.loss = NeRFWLoss(lambda_geo=hp["lambda_geo_init"], thickness=hp["thickness"], topk=hp["topk"])
        if self.output_transient_flow:                                   # train.py:136-138
            self.loss.register_buffer("Ks", Ks)
            self.loss.register_buffer("Ps", Ps)
            self.loss.max_t = n_frames - 1
        self.params = grad_parameters(models, embeddings)
        self.optimizer = torch.optim.Adam(self.params, lr=hp["lr"], eps=1e-8, weight_decay=hp["weight_decay"])
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(hp["decay_step"]),
                                                              gamma=hp["decay_gamma"])
        self.current_epoch = 0

    def to(self, device):
        for m in self.models.values():
            m.to(device)
        for k in ("t", "a"):
            if k in self.embeddings:
                self.embeddings[k].to(device)
        self.loss.to(device)
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
        return {k: torch.cat(v, 0) for k, v in results.items()}

    # train.py:174-176
    def on_train_epoch_start(self, epoch):
        self.current_epoch = epoch
        self.loss.lambda_geo_d = self.loss.lambda_geo_f = self.hp["lambda_geo_init"] * 0.1 ** (epoch // 10)

    # train.py:178-198
    def training_step(self, batch):
        kwargs = dict(output_transient=self.output_transient, output_transient_flow=self.output_transient_flow)
        results = self.forward(batch["rays"], batch.get("ts"), **kwargs)
        loss_d = self.loss(results, batch, epoch=self.current_epoch, **kwargs)
        loss = sum(loss_d.values())
        with torch.no_grad():
            log = {f"train/{k}": v.detach() for k, v in loss_d.items()}
            log["train/loss"] = loss.detach()
            log["train/psnr"] = psnr(results["rgb_fine"].detach(), batch["rgbs"])
            log["lr"] = self.optimizer.param_groups[0]["lr"]
        return loss, log

    def step(self, batch):
        """zero_grad -> training_step -> backward -> gradient all-reduce -> Adam; returns the log dict."""
        self.optimizer.zero_grad(set_to_none=True)
        loss, log = self.training_step(batch)
        loss.backward()
        allreduce_gradients(self.params)
        self.optimizer.step()
        return log

    def on_train_epoch_end(self):
        self.scheduler.step()
