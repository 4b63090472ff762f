"""Training-step semantics of the reference's ``Trainer`` / ``EMA`` (``ddpm_torch/utils/train.py:64-346`` of
tqch/ddpm-torch) on top of the MI355X engine.

What one ``Trainer.step`` does is the contract (utils/train.py:148-170): per-rank (t, noise) from a seeded device
generator -> ``diffusion.train_losses`` -> mean loss / num_accum -> backward -> every ``num_accum`` steps: global-norm
clip, optimizer step, zero_grad(set_to_none), LR schedule step, EMA update -> loss reduce to rank 0.
The forward/backward under ``loss.backward()`` is the hand-written HIP engine (one autograd node for the UNet, one for
the eps-MSE); EMA runs as multi-tensor ops over all 304 parameters instead of a Python loop.
Epoch loop / checkpoint I/O are thin host plumbing kept compatible with the reference's checkpoint layout.
"""
import math
import os
import re
import weakref
from contextlib import nullcontext

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from .. import _hip

__all__ = ["Trainer", "EMA", "ModelWrapper", "DummyScheduler", "RunningStatistics"]


class _FusedUpdate:
    """clip_grad_norm_ -> Adam.step -> EMA.update as TWO launches over all parameters (utils/train.py:159-165,300-305).

    Transparent fast path for the reference's own objects: it keeps torch.optim.Adam's state layout (``state[p]`` =
    {step, exp_avg, exp_avg_sq}) and the EMA shadow dict, so ``state_dict()`` / checkpoints are unchanged.  Anything it
    does not recognise (other optimizer, weight decay, amsgrad, several param groups, CPU params) falls back to the
    generic torch calls.  The clip coefficient is computed on the device from the accumulated squared norm: there is no
    host synchronisation between backward and the update.
    """

    def __init__(self, optimizer, ema):
        self.opt, self.ema = optimizer, ema
        self.ok = (type(optimizer) is torch.optim.Adam and len(optimizer.param_groups) == 1)
        if self.ok:
            g = optimizer.param_groups[0]
            self.ok = (not g.get("amsgrad") and not g.get("maximize") and g.get("weight_decay", 0) == 0
                       and not g.get("capturable") and not g.get("differentiable") and not g.get("fused")
                       and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in g["params"]))
        self.table = None
        self.steps = 0

    def _build(self):
        g = self.opt.param_groups[0]
        self.params = [p for p in g["params"] if p.requires_grad]
        dev = self.params[0].device
        shadow = {}
        if isinstance(self.ema, EMA):
            by_param = {id(r()): k for k, r in self.ema._refs.items()}
            shadow = {id(p): self.ema.shadow[by_param[id(p)]] for p in self.params if id(p) in by_param}
        rows = []
        for p in self.params:
            st = self.opt.state[p]
            if "exp_avg" not in st:                   # torch's lazy state initialisation (adam.py _init_group)
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            sh = shadow.get(id(p))
            rows.append([p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), 0 if sh is None else sh.data_ptr(), p.numel()])
        self.host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        self.table = torch.empty_like(self.host, device=dev)
        self.total = torch.zeros(64, dtype=torch.float32, device=dev)      # striped ||g||^2 accumulators (ddpm_mt_grad_sumsq)
        self.key = [p.data_ptr() for p in self.params]
        self.steps = int(self.opt.state[self.params[0]]["step"].item())

    def __call__(self, max_norm, ema_w):
        """Returns False (nothing done) when the fast path does not apply."""
        if not self.ok:
            return False
        if self.table is None or any(p.data_ptr() != k for p, k in zip(self.params, self.key)):
            self._build()
        if any(p.grad is None or not p.grad.is_contiguous() or p.grad.dtype != torch.float32 for p in self.params):
            return False
        self.host[:, 1] = torch.tensor([p.grad.data_ptr() for p in self.params], dtype=torch.int64)
        self.table.copy_(self.host, non_blocking=True)
        g = self.opt.param_groups[0]
        self.steps += 1
        b1, b2 = g["betas"]
        n = len(self.params)
        s = _hip.stream()
        if max_norm and max_norm > 0:
            self.total.zero_()
            _hip.call("ddpm_mt_grad_sumsq", self.table.data_ptr(), n, self.total.data_ptr(), s)
        _hip.call("ddpm_mt_adam_ema", self.table.data_ptr(), n, self.total.data_ptr() if max_norm and max_norm > 0 else 0, float(max_norm or 0.0),
                  float(g["lr"]), b1, b2, g["eps"], 1 - b1 ** self.steps, 1 - b2 ** self.steps, float(ema_w), s)
        torch._foreach_add_([self.opt.state[p]["step"] for p in self.params], 1)
        self.opt._opt_called = True                    # what LR schedulers check before their own step()
        return True


class DummyScheduler:
    def step(self):
        pass

    def load_state_dict(self, state_dict):
        pass

    def state_dict(self):
        return None


class RunningStatistics:
    """Running sums -> per-sample averages (utils/train.py:29-58)."""

    def __init__(self, **kwargs):
        self.count = 0
        self.stats = {k: (v or 0) for k, v in kwargs.items()}

    def reset(self):
        self.count = 0
        self.stats = {k: 0 for k in self.stats}

    def update(self, n, **kwargs):
        self.count += n
        for k, v in kwargs.items():
            self.stats[k] = self.stats.get(k, 0) + v

    def extract(self):
        return {k: v / self.count for k, v in self.stats.items()}


class EMA:
    """Exponential moving average of the trainable parameters (utils/train.py:279-346).

    decay_t = min(decay, (1 + n) / (10 + n)), n = number of updates so far starting at 0; shadow keys carry no
    ``module.`` prefix.  ``with ema:`` swaps the shadow weights in (and back out) in place, which bumps the
    parameters' version counters so the engine refreshes its packed copies.
    """

    def __init__(self, model, decay=0.9999):
        self.shadow, self._refs = {}, {}
        for k, v in model.named_parameters():
            if v.requires_grad:
                self.shadow[k] = v.detach().clone()
                self._refs[k] = weakref.ref(v)
        self.decay = decay
        self.num_updates = -1
        self.backup = None

    def _live(self):
        ps = [r() for r in self._refs.values()]
        assert all(p is not None for p in ps), "referenced object no longer exists!"
        return ps

    def update(self):
        self.num_updates += 1
        decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():                       # shadow += (1 - decay) * (p - shadow), all tensors in one sweep
            torch._foreach_lerp_(list(self.shadow.values()), [p.data for p in self._live()], 1 - decay)

    def apply(self):
        ps = self._live()
        self.backup = {k: p.detach().clone() for k, p in zip(self._refs, ps)}
        with torch.no_grad():
            torch._foreach_copy_([p.data for p in ps], list(self.shadow.values()))

    def restore(self):
        with torch.no_grad():
            torch._foreach_copy_([p.data for p in self._live()], list(self.backup.values()))
        self.backup = None

    def __enter__(self):
        self.apply()

    def __exit__(self, *exc):
        self.restore()

    def state_dict(self):
        return {"decay": self.decay, "shadow": self.shadow, "num_updates": self.num_updates}

    @property
    def extra_states(self):
        return {"decay", "num_updates"}

    def load_state_dict(self, state_dict, strict=True):
        mine = set(self.shadow).union(self.extra_states)
        theirs = set(state_dict["shadow"]).union(self.extra_states)
        bad = (mine ^ theirs) if strict else (mine - theirs)
        if bad:
            raise RuntimeError(f"Key mismatch!\nMissing key(s): {', '.join(mine - theirs)}."
                               f"Unexpected key(s): {', '.join(theirs - mine)}")
        self.__dict__.update(state_dict)


class ModelWrapper(nn.Module):
    """Optional pre/post transforms around the denoiser (pixel-(un)shuffle; utils/train.py:349-367)."""

    def __init__(self, model, pre_transform=None, post_transform=None):
        super().__init__()
        self._model = model
        self.pre_transform = pre_transform
        self.post_transform = post_transform

    def forward(self, x, *args, **kwargs):
        if self.pre_transform is not None:
            x = self.pre_transform(x)
        out = self._model(x, *args, **kwargs)
        if self.post_transform is not None:
            out = self.post_transform(out)
        return out


class Trainer:
    def __init__(self, model, optimizer, diffusion, epochs, trainloader, sampler=None, scheduler=None, num_accum=1,
                 use_ema=False, grad_norm=1.0, shape=None, device=torch.device("cpu"), chkpt_intv=5, image_intv=1,
                 num_samples=64, ema_decay=0.9999, distributed=False, rank=0, dry_run=False):
        self.model, self.optimizer, self.diffusion = model, optimizer, diffusion
        self.epochs, self.start_epoch = epochs, 0
        self.trainloader, self.sampler = trainloader, sampler
        if shape is None:
            shape = next(iter(trainloader))[0].shape[1:]
        self.shape = tuple(shape)
        self.scheduler = DummyScheduler() if scheduler is None else scheduler
        self.num_accum, self.grad_norm = num_accum, grad_norm
        self.device = torch.device(device)
        self.chkpt_intv, self.image_intv, self.num_samples = chkpt_intv, image_intv, num_samples
        if distributed:
            assert sampler is not None
        self.distributed, self.rank, self.dry_run = distributed, rank, dry_run
        self.is_leader = rank == 0
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.generator = torch.Generator(self.device).manual_seed(8191 + self.rank)     # utils/train.py:115
        self.sample_seed = 131071 + self.rank                                            # utils/train.py:117
        self.use_ema = use_ema
        self.ema = EMA(model.module if isinstance(model, DDP) else model, decay=ema_decay) if use_ema else nullcontext()
        self.stats = RunningStatistics(loss=None)
        self._fused = _FusedUpdate(optimizer, self.ema)

    @property
    def timesteps(self):
        return self.diffusion.timesteps

    def get_input(self, x):
        """Draw order per step: t first, then noise (utils/train.py:134-141)."""
        x = x.to(self.device)
        t = torch.empty((x.shape[0],), dtype=torch.int64, device=self.device).random_(to=self.timesteps, generator=self.generator)
        noise = torch.empty_like(x).normal_(generator=self.generator)
        return {"x_0": x, "t": t, "noise": noise}

    def loss(self, x):
        loss = self.diffusion.train_losses(self.model, **self.get_input(x))
        assert loss.shape == (x.shape[0],)
        return loss

    def step(self, x, global_steps=1):
        loss = self.loss(x).mean()
        loss.div(self.num_accum).backward()
        if global_steps % self.num_accum == 0:
            ema_on = self.use_ema and hasattr(self.ema, "update")
            ema_w = 1 - min(self.ema.decay, (2 + self.ema.num_updates) / (11 + self.ema.num_updates)) if ema_on else 0.0
            if self._fused(self.grad_norm, ema_w):          # clip + Adam + EMA in two launches (after DDP averaging)
                if ema_on:
                    self.ema.num_updates += 1
            else:
                nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.grad_norm)
                self.optimizer.step()
                if ema_on:
                    self.ema.update()
            self.optimizer.zero_grad(set_to_none=True)
            self.scheduler.step()
        loss = loss.detach()
        if self.distributed:
            dist.reduce(loss, dst=0, op=dist.ReduceOp.SUM)
            loss.div_(self.world_size)
        self.stats.update(x.shape[0], loss=loss.item() * x.shape[0])

    def sample_fn(self, sample_size=None, noise=None, diffusion=None, sample_seed=None):
        shape = ((sample_size // self.world_size,) + self.shape) if noise is None else noise.shape
        diffusion = diffusion or self.diffusion
        with self.ema:
            sample = diffusion.p_sample(denoise_fn=self.model, shape=shape, device=self.device, noise=noise, seed=sample_seed)
        if self.distributed:
            gathered = [torch.zeros(shape, device=self.device) for _ in range(self.world_size)]
            dist.all_gather(gathered, sample)
            sample = torch.cat(gathered, dim=0)
        assert sample.grad is None
        return sample

    def train(self, evaluator=None, chkpt_path=None, image_dir=None):
        if self.num_samples:
            assert self.num_samples % self.world_size == 0, "Number of samples should be divisible by WORLD_SIZE!"
        if self.dry_run:
            self.start_epoch, self.epochs = 0, 1
        global_steps = 0
        for e in range(self.start_epoch, self.epochs):
            self.stats.reset()
            self.model.train()
            results = {}
            if hasattr(self.sampler, "set_epoch"):
                self.sampler.set_epoch(e)
            for x in self.trainloader:
                if isinstance(x, (list, tuple)):
                    x = x[0]
                global_steps += 1
                self.step(x.to(self.device), global_steps=global_steps)
                results.update(self.current_stats)
                if self.dry_run and not global_steps % self.num_accum:
                    break
            if not (e + 1) % self.image_intv and self.num_samples and image_dir:
                self.model.eval()
                x = self.sample_fn(sample_size=self.num_samples, sample_seed=self.sample_seed).cpu()
                if self.is_leader:
                    torch.save(x, os.path.join(image_dir, f"{e + 1}.pt"))     # image encoding is out of scope (no torchvision)
            if not (e + 1) % self.chkpt_intv and chkpt_path:
                self.model.eval()
                results.update(evaluator.eval(self.sample_fn, is_leader=self.is_leader) if evaluator is not None else {})
                if self.is_leader:
                    self.save_checkpoint(chkpt_path, epoch=e + 1, **results)
            if self.distributed:
                dist.barrier()

    @property
    def trainees(self):
        roster = ["model", "optimizer"]
        if self.use_ema:
            roster.append("ema")
        if self.scheduler is not None:
            roster.append("scheduler")
        return roster

    @property
    def current_stats(self):
        return self.stats.extract()

    def load_checkpoint(self, chkpt_path, map_location):
        """Same on-disk layout as the reference (utils/train.py:249-276): {model, optimizer, ema, scheduler, epoch, ...}."""
        chkpt = torch.load(chkpt_path, map_location=map_location)
        for name in self.trainees:
            try:
                getattr(self, name).load_state_dict(chkpt[name])
            except RuntimeError:
                sd = chkpt[name]["shadow"] if name == "ema" else chkpt[name]
                for k in list(sd.keys()):
                    if k.startswith("module."):
                        sd[k.split(".", maxsplit=1)[1]] = sd.pop(k)
                getattr(self, name).load_state_dict(chkpt[name])
            except AttributeError:
                continue
        self.start_epoch = chkpt["epoch"]

    def save_checkpoint(self, chkpt_path, **extra_info):
        chkpt = dict(self.named_state_dicts())
        chkpt.update(extra_info)
        if "epoch" in extra_info:
            chkpt_path = re.sub(r"(_\d+)?\.pt", f"_{extra_info['epoch']}.pt", chkpt_path)
        torch.save(chkpt, chkpt_path)

    def named_state_dicts(self):
        for k in self.trainees:
            yield k, getattr(self, k).state_dict()
