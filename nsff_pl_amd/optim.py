"""The optimizer of the training step: ``torch.optim.Adam(lr, eps=1e-8, weight_decay)`` as
the reference's ``get_optimizer`` builds it (utils/__init__.py:45-47, train.py:140-146), run by ONE native launch pair
(``nsff_adam_step``, csrc/optim.hip) on flat buffers.

Every parameter becomes a view of one flat fp32 buffer, every ``.grad`` a view of a second one (the buffer the
weight-gradient kernel accumulates into and RCCL all-reduces); the two moment buffers are flat as well.  Step count and
learning rate live on the device, so the step is the same two launches eagerly and inside a hipGraph.  Parameters keep
their shapes, ``state_dict`` keys and identity (``nn.Parameter`` objects are untouched, only their storage moves).

Where this differs from ``torch.optim.Adam`` -- read before swapping it in elsewhere:

* **Parameters without a gradient.**  torch skips parameters whose ``.grad`` is None; here ``.grad`` always exists (a
  slice of the flat buffer, zero after ``zero_grad``).  With ``weight_decay == 0`` (the reference's default, opt.py:84)
  every element is updated every step: a zero gradient only decays the moments, as torch would for a zero-valued gradient.
  With ``weight_decay > 0`` a parameter that never receives a gradient (an unused head) must NOT be decayed, so the step
  runs in its segment form (``nsff_adam_step_segments``): a parameter tensor whose gradient slice is identically zero this
  step keeps its value and its moments, which is what torch does for ``grad is None``.  ``decay_unused=True`` selects the
  plain every-element step instead (one launch fewer).  Known deviation of that test: "unused" is decided from gradient VALUES,
  so a parameter whose true gradient is exactly zero everywhere (a dead ReLU layer, a fully masked loss) is skipped too,
  where torch -- which sees a zero-valued ``.grad`` tensor, not ``None`` -- would still decay it and its moments.
* **One step counter.**  The bias corrections use one global step count (``state[0]``), not torch's per-parameter counts: a
  tensor that receives its first gradient at step N is corrected as at step N, not as at step 1.  The two agree whenever
  every parameter is used from the first step on (the reference's training loop) or ``weight_decay == 0`` with gradients
  that start at step 1; ``load_torch_state_dict`` refuses per-parameter counts that differ.
* **HIP device only.**  There is no CPU implementation (tests drive CPU runs with a torch-op twin, tests/common.py).
* **Shared storage.**  ``module.state_dict()`` tensors are views of the one flat buffer: ``torch.save`` of such a dict
  writes the whole buffer once per file.  Use :func:`detached_state` (or ``NSFFTrainer.checkpoint``) to get clones.
* **Optimizer state.**  ``state_dict`` / ``load_state_dict`` use this class's flat layout; ``torch_state_dict`` /
  ``load_torch_state_dict`` convert to and from ``torch.optim.Adam``'s per-parameter format (resuming a reference /
  torch checkpoint, or handing a run back to torch).
"""
import torch

from . import _lib


class FlatAdam:
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decay_unused=False):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
        self.decay_unused = bool(decay_unused)
        dev = self.params[0].device
        self._check_device(dev)
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise RuntimeError("FlatAdam: parameters must be fp32 tensors on one device")
        self.betas, self.eps, self.weight_decay = (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.numel = sum(p.numel() for p in self.params)
        padded = (self.numel + 3) // 4 * 4                     # the kernel works on float4
        self.flat_param = torch.zeros(padded, device=dev)
        self.flat_grad = torch.zeros(padded, device=dev)
        self.exp_avg = torch.zeros(padded, device=dev)
        self.exp_avg_sq = torch.zeros(padded, device=dev)
        self.state = torch.zeros(4, device=dev)                # [0] = steps taken
        self.lr = torch.tensor(float(lr), device=dev)
        # parameter tensor k = flat elements [seg_start[k], seg_start[k + 1]) -- the segment form of the step (see above)
        offs = [0]
        for p in self.params:
            offs.append(offs[-1] + p.numel())
        self.seg_start = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.seg_used = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        self.param_groups = [{"lr": self.lr, "params": self.params}]   # (what loggers / schedulers look at)
        with torch.no_grad():
            off = 0
            for p in self.params:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                off += n
        self.adopt()

    @staticmethod
    def _check_device(dev):
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the HIP device only (move the models first); there is no CPU path")

    # -- storage ----------------------------------------------------------------------------------------------
    def adopt(self):
        """(Re-)point every parameter and gradient at its slice of the flat buffers (values are taken from the flat
        buffers; call :meth:`gather` first if the parameters were replaced from outside, e.g. by load_state_dict on
        re-created tensors)."""
        off = 0
        for p in self.params:
            n = p.numel()
            p.data = self.flat_param[off:off + n].view(p.shape)
            p.grad = self.flat_grad[off:off + n].view(p.shape)
            off += n

    def in_place(self):
        """True while every parameter and gradient still aliases its slice of the flat buffers."""
        base_p, base_g, off = self.flat_param.data_ptr(), self.flat_grad.data_ptr(), 0
        for p in self.params:
            g = p.grad
            if p.data_ptr() != base_p + 4 * off or g is None or g.data_ptr() != base_g + 4 * off:
                return False
            off += p.numel()
        return True

    def gather(self):
        """Copy the current parameter values into the flat buffer (after something re-created the tensors), re-adopt."""
        with torch.no_grad():
            off = 0
            for p in self.params:
                n = p.numel()
                if p.data_ptr() != self.flat_param.data_ptr() + 4 * off:
                    self.flat_param[off:off + n].copy_(p.detach().reshape(-1).to(self.flat_param.device))
                off += n
        self.adopt()

    # -- torch.optim surface --------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    def step(self):
        skip_unused = self.weight_decay != 0 and not self.decay_unused
        _lib.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.state, self.lr,
                       self.betas[0], self.betas[1], self.eps, self.weight_decay,
                       self.seg_start if skip_unused else None, self.seg_used if skip_unused else None)

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def reset_state(self):
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.state.zero_()

    def state_dict(self):
        return dict(step=self.state[0:1].clone(), lr=self.lr.clone(), exp_avg=self.exp_avg[:self.numel].clone(),
                    exp_avg_sq=self.exp_avg_sq[:self.numel].clone(), betas=self.betas, eps=self.eps,
                    weight_decay=self.weight_decay)

    def torch_state_dict(self):
        """The same state in ``torch.optim.Adam.state_dict()`` format (cloned tensors, parameter order of ``params``)."""
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            state[i] = {"step": self.state[0].detach().clone().cpu(),
                        "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = {"lr": float(self.lr), "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_torch_state_dict(self, sd):
        """Take over moments, step count and hyper-parameters of a ``torch.optim.Adam.state_dict()`` over the same
        parameters in the same order (one param group, no amsgrad).  Parameters without state keep zero moments."""
        groups = sd["param_groups"]
        if len(groups) != 1 or groups[0].get("amsgrad", False):
            raise ValueError("FlatAdam takes one parameter group without amsgrad")
        g = groups[0]
        if len(g["params"]) != len(self.params):
            raise ValueError(f"optimizer state covers {len(g['params'])} parameters, this optimizer has {len(self.params)}")
        steps = set()
        with torch.no_grad():
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            off = 0
            for key, p in zip(g["params"], self.params):
                n = p.numel()
                st = sd["state"].get(key)
                if st is not None:
                    if tuple(st["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError(f"state of parameter {key} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
                    self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                off += n
            if len(steps) > 1:
                raise ValueError(f"parameters are at different step counts {sorted(steps)}: one shared counter here")
            self.state.zero_(); self.state[0] = float(steps.pop()) if steps else 0.0
            self.lr.fill_(float(g["lr"]))
        self.betas, self.eps = (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])
        self.weight_decay = float(g.get("weight_decay", 0.0))

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.state.zero_(); self.state[0:1].copy_(sd["step"])
            self.lr.copy_(sd["lr"])
            self.exp_avg[:self.numel].copy_(sd["exp_avg"]); self.exp_avg_sq[:self.numel].copy_(sd["exp_avg_sq"])
        self.betas, self.eps, self.weight_decay = tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])


def detached_state(module):
    """``module.state_dict()`` with every tensor cloned: safe to ``torch.save`` / keep while training goes on (the live
    tensors are views of FlatAdam's one flat buffer and change in place)."""
    return {k: v.detach().clone() for k, v in module.state_dict().items()}
