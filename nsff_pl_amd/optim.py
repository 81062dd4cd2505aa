"""The optimizer of the training step: ``torch.optim.Adam(lr, eps=1e-8, weight_decay)`` as
the reference's ``get_optimizer`` builds it (utils/__init__.py:45-47, train.py:140-146), run by ONE native launch pair
(``nsff_adam_step``, csrc/optim.hip) on flat buffers.

Every parameter becomes a view of one flat fp32 buffer, every ``.grad`` a view of a second one (the buffer the
weight-gradient kernel accumulates into and RCCL all-reduces); the two moment buffers are flat as well.  Step count and
learning rate live on the device, so the step is the same two launches eagerly and inside a hipGraph.  Parameters keep
their shapes, ``state_dict`` keys and identity (``nn.Parameter`` objects are untouched, only their storage moves).
"""
import torch

from . import _lib


class FlatAdam:
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
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
        _lib.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.state, self.lr,
                       self.betas[0], self.betas[1], self.eps, self.weight_decay)

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def reset_state(self):
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.state.zero_()

    def state_dict(self):
        return dict(step=self.state[0:1].clone(), lr=self.lr.clone(), exp_avg=self.exp_avg[:self.numel].clone(),
                    exp_avg_sq=self.exp_avg_sq[:self.numel].clone(), betas=self.betas, eps=self.eps,
                    weight_decay=self.weight_decay)

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.state.zero_(); self.state[0:1].copy_(sd["step"])
            self.lr.copy_(sd["lr"])
            self.exp_avg[:self.numel].copy_(sd["exp_avg"]); self.exp_avg_sq[:self.numel].copy_(sd["exp_avg_sq"])
        self.betas, self.eps, self.weight_decay = tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])
