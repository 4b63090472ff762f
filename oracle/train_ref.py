"""Oracle (test infrastructure): CPU restatement of one reference training step.

Follows ddpm_torch/utils/train.py:148-170 (Trainer.step), :300-305 (EMA.update) and
train.py:128-132 (Adam + linear warm-up).  Parameters live in a state dict of leaf tensors;
the denoiser is ``oracle.unet_ref.unet_forward``.  Not product code.
"""
import torch

from . import diffusion_ref as D
from . import unet_ref as U


class TrainState:
    def __init__(self, sd, cfg, lr, warmup, grad_norm=1.0, ema_decay=0.9999, betas=(0.9, 0.999), lr_lambda=None):
        self.cfg = cfg
        self.params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, betas=betas)       # train.py:128
        if lr_lambda is None and warmup > 0:
            lr_lambda = lambda s: min((s + 1) / warmup, 1.0)                               # train.py:130-132
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lr_lambda=lr_lambda) if lr_lambda is not None else None
        self.grad_norm = grad_norm
        self.ema_decay = ema_decay
        self.shadow = {k: v.detach().clone() for k, v in self.params.items()}              # utils/train.py:286-293
        self.num_updates = -1                                                               # utils/train.py:297

    def step(self, tables, x_0, t, noise, training=True, masks=None):
        """utils/train.py:148-170 with num_accum=1; returns the mean loss (python float)."""
        x_t = D.q_sample(tables, x_0, t, noise)
        eps_hat = U.unet_forward(self.params, self.cfg, x_t, t, training=training, masks=masks)
        loss = D.mse_eps_loss(eps_hat, noise).mean()
        loss.backward()
        self.last_grads = {k: p.grad.detach().clone() for k, p in self.params.items()}
        self.last_gnorm = float(torch.nn.utils.clip_grad_norm_(list(self.params.values()), max_norm=self.grad_norm))
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.sched is not None:
            self.sched.step()
        self.num_updates += 1                                                               # utils/train.py:300-305
        d = min(self.ema_decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for k, p in self.params.items():
                self.shadow[k] += (1 - d) * (p.data - self.shadow[k])
        return float(loss.detach())
