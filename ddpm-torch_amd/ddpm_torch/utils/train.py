"""``Trainer`` / ``EMA`` with the reference's constructor and call contracts (``ddpm_torch/utils/train.py:64-367`` of
tqch/ddpm-torch), rebuilt around the MI355X engine.

What one ``Trainer.step`` computes is the contract (utils/train.py:148-170): per-rank ``t`` then ``noise`` from a seeded
device generator -> ``diffusion.train_losses`` -> mean loss / num_accum -> backward -> every ``num_accum`` steps
global-norm clip, Adam step, ``zero_grad(set_to_none)``, LR-schedule step, EMA update -> loss reduce to rank 0 ->
``loss.item()`` into the running statistics.  How it is executed here:

* **direct step** (the normal case: the model is this package's ``UNet``, ``torch.optim.Adam`` with one plain parameter
  group, ``num_accum == 1``, eps/x_0/mean-MSE loss): no autograd graph at all — the engine's hand-written forward and
  backward are called back to back, the eps-MSE and its gradient are two kernels, clip + Adam + EMA are two multi-tensor
  launches over a pointer table, and the packed weight copies are re-derived at the end of the step.  All step-dependent
  scalars (learning rate, Adam bias corrections, EMA weight, the per-step part of the dropout seed) are read by the
  kernels from a small device buffer, so the whole step is shape-static.  It runs in one of three forms, chosen by measurement on the
  box (``DDPM_TORCH_AMD_TRAIN_GRAPH=auto``): eager launches from Python; a **launch plan** (``_plan.LaunchPlan`` / ``csrc/plan.hip``:
  the step's ~500 C-ABI calls recorded once and re-issued from C on both streams — the eager step without the interpreter; the
  default wherever it has been measured); or **hipGraphs** (``_graphs.SegmentedGraph``).  In data-parallel runs the RCCL calls stay
  outside the plan / the graphs, between segments.
* **autograd step** (anything else: DDP-wrapped or user-wrapped models, gradient accumulation, other optimisers):
  ``loss.backward()`` through the engine's single autograd node, then the fused multi-tensor update when the optimiser
  qualifies, else the plain torch calls.

Checkpoints keep the reference's on-disk layout ({model, optimizer, ema, scheduler, epoch, ...}, utils/train.py:249-276).
"""
import math
import os
import re
import time
import warnings
import weakref
from contextlib import nullcontext

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from .. import _hip
from .._graphs import SegmentedGraph
from .._plan import LaunchPlan

__all__ = ["Trainer", "EMA", "ModelWrapper", "DummyScheduler", "RunningStatistics"]

# "auto" (default): the captured step is used when it measures faster than the eager one on this workload; "1" / "0" force it
_MAIN_PRIORITY = os.environ.get("DDPM_MAIN_PRIORITY", "0") != "0"        # run the direct step's main chain on a high-priority stream
_ASYNC_LOSS = os.environ.get("DDPM_TORCH_AMD_ASYNC_LOSS", "1") != "0"     # 0: read the loss back synchronously in every step
# distributed runs: 1 = the per-step loss reduce is issued asynchronously and read back on a stream of its own (round 5).  Default 0 = the
# reference's synchronous reduce on the step's stream (utils/train.py:166-169): opt in after tests/test_multi_gpu.py passed on >= 2 GPUs.
_ASYNC_LOSS_REDUCE = os.environ.get("DDPM_TORCH_AMD_ASYNC_LOSS_REDUCE", "0") != "0"
_TRAIN_GRAPH = {"0": False, "1": True, "plan": "plan"}.get(os.environ.get("DDPM_TORCH_AMD_TRAIN_GRAPH", "auto"), "auto")


class DummyScheduler:
    """Stands in when no LR scheduler is given (utils/train.py:17-26)."""

    def step(self):
        return None

    def state_dict(self):
        return None

    def load_state_dict(self, state_dict):
        return None


class RunningStatistics:
    """Sample-weighted running means: ``update(n, loss=sum_of_n_losses)``; ``extract()`` divides by the sample count
    (utils/train.py:29-58)."""

    def __init__(self, **kwargs):
        self.count = 0
        self.stats = {name: (0 if value is None else value) for name, value in kwargs.items()}

    def reset(self):
        self.count = 0
        self.stats = dict.fromkeys(self.stats, 0)

    def update(self, n, **kwargs):
        self.count += n
        for name, value in kwargs.items():
            self.stats[name] = self.stats.get(name, 0) + value

    def extract(self):
        denom = self.count
        return {name: total / denom for name, total in self.stats.items()}

    def __repr__(self):
        return "RunningStatistics(" + ", ".join(f"{k}={v:.6g}" for k, v in self.extract().items()) + ")" if self.count else "RunningStatistics()"


class _StepStatistics(RunningStatistics):
    """The Trainer's statistics: the loss of the latest step may still be on its way from the device (see Trainer.step); every
    read or reset collects it first, so callers see exactly what the synchronous read-back would have produced."""

    def __init__(self, drain, **kwargs):
        super().__init__(**kwargs)
        self._drain = drain

    def reset(self):
        self._drain()
        super().reset()

    def extract(self):
        self._drain()
        return super().extract()


class EMA:
    """Exponential moving average of the trainable parameters (utils/train.py:279-346).

    ``decay_t = min(decay, (1 + n) / (10 + n))`` with ``n`` = updates so far (0 for the first); shadow keys are the
    model's ``named_parameters`` names (no ``module.`` prefix).  ``with ema:`` copies the shadow into the live
    parameters and back — in place and through the parameters themselves, which bumps their version counters, so the
    engine re-derives its packed copies (a write through ``p.data`` would go unnoticed).  Shadow tensors are never re-bound after construction (``load_state_dict`` copies INTO them, key by key):
    the fused update kernel keeps their addresses in its pointer table.
    """

    def __init__(self, model, decay=0.9999):
        self.shadow, self._refs = {}, {}
        for name, prm in model.named_parameters():
            if prm.requires_grad:
                self.shadow[name] = prm.detach().clone()
                self._refs[name] = weakref.ref(prm)
        self.decay = decay
        self.num_updates = -1
        self.backup = None

    def _pairs(self):
        """[(shadow tensor, live parameter)] matched by NAME."""
        out = []
        for name, ref in self._refs.items():
            prm = ref()
            assert prm is not None, "referenced object no longer exists!"
            out.append((self.shadow[name], prm))
        return out

    def weight_of_next_update(self):
        """1 - decay of the update that would run next (the fused kernel applies it; ``update`` uses the same value)."""
        n = self.num_updates + 1
        return 1.0 - min(self.decay, (1 + n) / (10 + n))

    def update(self):
        w = self.weight_of_next_update()
        self.num_updates += 1
        pairs = self._pairs()
        with torch.no_grad():                       # shadow += w * (p - shadow) over all tensors in one sweep
            torch._foreach_lerp_([s for s, _ in pairs], [p.data for _, p in pairs], w)

    def apply(self):
        pairs = self._pairs()
        self.backup = {name: ref().detach().clone() for name, ref in self._refs.items()}
        with torch.no_grad():
            # written through the parameters themselves (not ``.data``, whose writes the version counters do not see): the
            # engine's derived weight copies are keyed on those counters
            torch._foreach_copy_([p for _, p in pairs], [s for s, _ in pairs])

    def restore(self):
        with torch.no_grad():
            torch._foreach_copy_([ref() for ref in self._refs.values()], [self.backup[name] for name in self._refs])
        self.backup = None

    def __enter__(self):
        self.apply()

    def __exit__(self, *exc):
        self.restore()

    @property
    def extra_states(self):
        return {"decay", "num_updates"}

    def state_dict(self):
        return {"decay": self.decay, "shadow": self.shadow, "num_updates": self.num_updates}

    def load_state_dict(self, state_dict, strict=True):
        own = set(self.shadow) | self.extra_states
        given = set(state_dict["shadow"]) | self.extra_states
        missing, unexpected = own - given, given - own
        if missing or (strict and unexpected):
            raise RuntimeError(f"Key mismatch!\nMissing key(s): {', '.join(sorted(missing))}."
                               f"Unexpected key(s): {', '.join(sorted(unexpected))}")
        with torch.no_grad():
            for name, mine in self.shadow.items():
                mine.copy_(state_dict["shadow"][name])
        self.decay = state_dict.get("decay", self.decay)
        self.num_updates = state_dict.get("num_updates", self.num_updates)


class ModelWrapper(nn.Module):
    """Denoiser with an input transform before and an output transform after it (the reference wraps the UNet with
    pixel-unshuffle / pixel-shuffle when ``block_size > 1``, train.py:69-72; utils/train.py:349-367).  The wrapped
    module is registered as ``_model`` — checkpoints of wrapped models carry that prefix."""

    def __init__(self, model, pre_transform=None, post_transform=None):
        super().__init__()
        self._model = model
        self.pre_transform = pre_transform
        self.post_transform = post_transform

    def forward(self, x, *args, **kwargs):
        y = x if self.pre_transform is None else self.pre_transform(x)
        y = self._model(y, *args, **kwargs)
        return y if self.post_transform is None else self.post_transform(y)


# ========================================================================================== fused clip + Adam + EMA
class _FusedUpdate:
    """``clip_grad_norm_`` -> ``Adam.step`` -> ``EMA.update`` as TWO launches over all parameters
    (utils/train.py:159-165,300-305; train.py:128).

    Works on the reference's own objects: ``torch.optim.Adam``'s state layout (``state[p]`` = {step, exp_avg, exp_avg_sq})
    and the EMA shadow dict are updated in place, so ``state_dict()`` / checkpoints are unchanged.  Anything it does not
    recognise (another optimiser, weight decay, amsgrad, several parameter groups, CPU parameters) is declined and the
    caller uses the generic torch calls.  The clip coefficient is computed on the device from the accumulated squared
    norm: no host synchronisation between backward and the update.

    The kernels write parameters through raw pointers, which autograd's version counters cannot see; every launch is
    therefore followed by ``increment_version`` on the updated parameters so that version-keyed caches (the engine's
    packed weights, captured sampler graphs) notice.
    """

    def __init__(self, optimizer, ema):
        self.opt, self.ema = optimizer, ema
        self.table = self.key = self.sig = None
        self.step_block = None
        self.total = None
        self.generation = 0                  # bumped whenever the pointer table is rebuilt (captured steps compare it)

    def eligible(self):
        opt = self.opt
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("weight_decay", 0) != 0 or g.get("capturable") or g.get("differentiable") or g.get("fused"):
            return False
        return all(_hip.on_device(p) and p.dtype == torch.float32 and p.is_contiguous() for p in g["params"])

    def _state_tensors(self):
        """(params, exp_avg, exp_avg_sq, shadow-or-None) in optimiser order, creating torch's lazy Adam state if needed."""
        params = [p for p in self.opt.param_groups[0]["params"] if p.requires_grad]
        shadow_of = {}
        if isinstance(self.ema, EMA):
            shadow_of = {id(prm): sh for sh, prm in self.ema._pairs()}
        m, v, sh = [], [], []
        for p in params:
            st = self.opt.state[p]
            if "exp_avg" not in st:                       # adam.py _init_group
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m.append(st["exp_avg"]); v.append(st["exp_avg_sq"]); sh.append(shadow_of.get(id(p)))
        return params, m, v, sh

    def _signature(self, grad_ptrs):
        """Cheap per-step check that nothing the table points at has moved: first / last entries of every column.  (Whole
        columns move together: ``.to()`` re-creates all parameters, ``optimizer.load_state_dict`` all moments; the EMA never
        re-binds its shadow tensors.)"""
        group = self.opt.param_groups[0]["params"]
        sig = []
        for p in (group[0], group[-1]):
            st = self.opt.state.get(p, {})
            sig += [p.data_ptr(), st["exp_avg"].data_ptr() if "exp_avg" in st else 0, st["exp_avg_sq"].data_ptr() if "exp_avg_sq" in st else 0]
        if isinstance(self.ema, EMA) and self.ema.shadow:
            sig.append(next(iter(self.ema.shadow.values())).data_ptr())
        if grad_ptrs is not None:
            sig += [grad_ptrs[0], grad_ptrs[-1], len(grad_ptrs)]
        return tuple(sig)

    def prepare(self, grad_ptrs=None, stable_grads=False):
        """(Re)build the device pointer table when an address in it moved (``optimizer.load_state_dict`` and ``.to()``
        re-create state tensors).  ``grad_ptrs``: addresses of the gradients in optimiser order; None = ``p.grad``.
        ``stable_grads``: the gradients live in a persistent buffer — a cheap signature decides whether to look closer."""
        if stable_grads and self.table is not None and self._signature(grad_ptrs) == self.sig:
            return self.params
        params, m, v, sh = self._state_tensors()
        if grad_ptrs is None:
            grad_ptrs = [p.grad.data_ptr() for p in params]
        key = tuple(t.data_ptr() for t in params) + tuple(t.data_ptr() for t in m) + tuple(t.data_ptr() for t in v) + \
            tuple(0 if t is None else t.data_ptr() for t in sh) + tuple(grad_ptrs)
        if key != self.key:
            rows = [[p.data_ptr(), g, a.data_ptr(), b.data_ptr(), 0 if s is None else s.data_ptr(), p.numel()]
                    for p, g, a, b, s in zip(params, grad_ptrs, m, v, sh)]
            dev = params[0].device
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            if self.total is None or self.total.device != dev:
                # ||g||^2: bank of 64 (lane 0 = the norm, fixed-order sum) + the per-block partial slots behind it
                self.total = torch.zeros(_hip.SUMSQ_FLOATS, dtype=torch.float32, device=dev)
            self.key, self.params = key, params
            self.generation += 1
            # Adam keeps one 0-dim CPU "step" tensor per parameter; bumping ~300 of them costs the host 1.3 ms per step.  Re-bind them as
            # views of ONE block (same dtype, same values, still tensors in state_dict()): the bookkeeping of a step is one add_.
            steps = [self.opt.state[p]["step"] for p in params]
            if all(torch.is_tensor(t) and not t.is_cuda and t.dtype == steps[0].dtype for t in steps):
                block = torch.stack([t.detach().reshape(()) for t in steps])
                for i, p in enumerate(params):
                    self.opt.state[p]["step"] = block[i]
                self.step_block = block
            else:
                self.step_block = None
        self.sig = self._signature(grad_ptrs) if stable_grads else None
        return self.params

    def step_count(self):
        """Optimiser steps taken so far, read from the optimiser's own state (survives load_state_dict and fallbacks)."""
        first = next(p for p in self.opt.param_groups[0]["params"] if p.requires_grad)
        st = self.opt.state.get(first, {})
        return int(st["step"]) if "step" in st else 0

    def launch(self, max_norm, scalars=None, hyper_dev=0, have_sumsq=False):
        """The two launches.  ``scalars`` = (lr, bias_corr1, bias_corr2, ema_w) by value, or ``hyper_dev`` = address of the
        same four floats in device memory (captured steps).  ``have_sumsq``: ``self.total`` already holds the squared gradient
        norm (the engine's backward accumulated it while writing the flat gradient): only the update is launched."""
        g = self.opt.param_groups[0]
        b1, b2 = g["betas"]
        n, s = len(self.params), _hip.stream()
        clip = bool(max_norm) and max_norm > 0
        if clip and not have_sumsq:
            if n > _hip.SUMSQ_MAX_TENSORS:
                raise ValueError(f"{n} parameter tensors exceed the squared-norm buffer ({_hip.SUMSQ_MAX_TENSORS})")
            _hip.call("ddpm_mt_grad_sumsq", self.table.data_ptr(), n, self.total.data_ptr(), self.total.numel(), s)
        lr, bc1, bc2, ema_w = scalars if scalars is not None else (0.0, 1.0, 1.0, 0.0)
        _hip.call("ddpm_mt_adam_ema", self.table.data_ptr(), n, self.total.data_ptr() if clip else 0, float(max_norm or 0.0),
                  float(lr), b1, b2, g["eps"], float(bc1), float(bc2), float(ema_w), hyper_dev, s)

    def scalars(self, ema_w):
        g = self.opt.param_groups[0]
        b1, b2 = g["betas"]
        k = self.step_count() + 1
        return float(g["lr"]), 1 - b1 ** k, 1 - b2 ** k, float(ema_w)

    def committed(self):
        """Host-side bookkeeping after the update kernels were enqueued (eagerly or by a graph replay)."""
        block = getattr(self, "step_block", None)
        if block is not None and self.opt.state[self.params[0]]["step"].untyped_storage().data_ptr() == block.untyped_storage().data_ptr():
            block.add_(1)
        else:                                          # (state re-created behind our back and not yet re-bound by prepare())
            torch._foreach_add_([self.opt.state[p]["step"] for p in self.params], 1)
        torch.autograd.graph.increment_version(self.params)
        self.opt._opt_called = True                    # what LR schedulers check before their own step()

    def __call__(self, max_norm, ema_w):
        """Autograd path: update from ``p.grad``.  Returns False (nothing done) when the fast path does not apply."""
        if not self.eligible():
            return False
        params = [p for p in self.opt.param_groups[0]["params"] if p.requires_grad]
        if any(p.grad is None or not p.grad.is_contiguous() or p.grad.dtype != torch.float32 for p in params):
            return False
        self.prepare()
        self.launch(max_norm, scalars=self.scalars(ema_w))
        self.committed()
        return True


def _end_generator_capture(gen, device):
    """A capture that dies half-way never reaches the epilogue that switches a registered generator back from its in-graph offset to
    the ordinary one; the next eager draw from it then raises "Offset increment outside graph capture encountered unexpectedly" — the
    fallback to the eager step would crash exactly when it is needed (seen once in round 5).  A complete, EMPTY capture with the generator
    registered runs prologue and epilogue and leaves it usable; its seed and offset outside graphs were never touched."""
    if gen is None or gen.device.type != "cuda":
        return
    try:
        graph, stream = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=device)
        graph.register_generator_state(gen)
        with torch.cuda.stream(stream), warnings.catch_warnings():
            warnings.simplefilter("ignore")                 # ("The CUDA Graph is empty")
            graph.capture_begin(capture_error_mode="thread_local")
            graph.capture_end()
        torch.cuda.synchronize(device)
    except Exception as e:                                  # best effort: the eager step will report what is still wrong
        warnings.warn(f"could not return the training generator to eager use after the failed capture ({type(e).__name__}: {e})")


# ========================================================================================== the direct / captured step
class _DirectStep:
    """State of the autograd-free training step for ONE input shape: persistent buffers, the pointer table of the fused
    update over the engine's flat gradient buffer, and (on the GPU) the captured graphs."""

    def __init__(self, trainer, unet, shape):
        self.tr, self.unet = trainer, unet
        dev = trainer.device
        B = shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        self.x0 = torch.empty(shape, **f32)
        self.noise = torch.empty(shape, **f32)
        self.t = torch.empty((B,), dtype=torch.int64, device=dev)
        self.gloss = torch.full((B,), 1.0 / B, **f32)            # d(mean loss)/d(loss_b)
        self.loss = torch.zeros((), **f32)
        eng = unet.engine()
        self.gflat = torch.empty(eng.gtotal, **f32)
        # step-dependent scalars the kernels read from memory: floats {lr, bc1, bc2, ema_w} | uint64 dropout seed word | pad
        self.hyper_dev = torch.zeros(4, dtype=torch.int64, device=dev)
        self.hyper_host = [self._pinned(torch.zeros(4, dtype=torch.int64)) for _ in range(4)]
        self.grad_ptrs = None
        self.calls = 0
        self.graph = None
        self.graph_failed = False
        self.plan = None                     # _plan.LaunchPlan of this step (recorded once, replayed from C)
        self.plan_failed = False
        self.plan_stale = 0                  # how often a recorded plan had to be dropped because what it baked in moved
        self._plan_baked = None
        self.captures = 0                    # graph captures + plan recordings so far (bench.py asserts none happens inside a timed region)
        self._baked = None                   # identity of everything whose ADDRESS the captured graph holds (see _identity)
        self._warm = None                    # serial of the engine that has run an EAGER step here (lazy tables / slabs / workspaces exist)
        # auto mode: wall times (Trainer.step reports them) of a few eager and a few replayed steps decide which one stays
        self.times = {}                      # auto mode: seconds per step of the "eager" / "graph" probe phases
        self.last_kind, self.choice, self._phase = None, None, None

    @staticmethod
    def _pinned(t):
        return t.pin_memory() if torch.cuda.is_available() else t

    def _write_hyper(self):
        tr, eng = self.tr, self.unet.engine()
        host = self.hyper_host[self.calls % len(self.hyper_host)]
        ema_w = tr.ema.weight_of_next_update() if tr._ema_on else 0.0
        lr, bc1, bc2, w = tr._fused.scalars(ema_w)
        host.view(torch.float32)[:4] = torch.tensor([lr, bc1, bc2, w], dtype=torch.float32)
        host[2] = eng.next_dropout_seed() if self.unet.training and self.unet.drop_rate > 0 else 0
        self.hyper_dev.copy_(host, non_blocking=True)

    def draw(self):
        """This step's (t, noise) into the persistent buffers: t first, then noise (utils/train.py:138-140)."""
        tr = self.tr
        if tr.input_source is not None:
            tr.input_source(self.t, self.noise)                                # parity tests: an injected (t, noise) stream
        else:
            self.t.random_(to=tr.timesteps, generator=tr.generator)
            self.noise.normal_(generator=tr.generator)

    def body(self, cut=None, draw=True):
        """One training step on the persistent buffers (x0 and the hyper words are already in place).  Apart from the draw, every device
        operation is a C-ABI call (fills and stream edges included), which is what lets a launch plan record it."""
        tr, dif, eng = self.tr, self.tr.diffusion, self.unet.engine()
        B, n = self.x0.shape[0], self.x0[0].numel()
        s = _hip.stream
        if draw:
            self.draw()
        x_t = dif.q_sample(self.x0, self.t, noise=self.noise)
        target = dif.loss_target(self.x0, x_t, self.t, self.noise)
        tape = []
        out = eng.forward(x_t, self.t, self.unet.training, tape, seed_dev=self.hyper_dev.data_ptr() + 16)
        losses = _hip.retain(torch.empty(B, dtype=torch.float32, device=out.device))
        _hip.call("ddpm_mse_fwd", out.data_ptr(), target.data_ptr(), losses.data_ptr(), B, n, s())
        _hip.call("ddpm_weighted_sum_f32", losses.data_ptr(), self.gloss.data_ptr(), self.loss.data_ptr(), B, s())   # mean over the batch
        gout = _hip.retain(torch.empty_like(out))
        _hip.call("ddpm_mse_bwd", out.data_ptr(), target.data_ptr(), self.gloss.data_ptr(), gout.data_ptr(), B, n, s())
        clip = bool(tr.grad_norm) and tr.grad_norm > 0
        # (the backward's last launch, which rewrites the staging buffer into the flat gradient, also produces its squared norm: per-block
        # partials added in a fixed order, so two replicas with the same gradients clip by the same coefficient, bit for bit)
        eng.backward(tape, gout, gflat=self.gflat, cut=cut, want_views=False, sumsq=tr._fused.total.data_ptr() if clip else 0)
        tr._fused.launch(tr.grad_norm, hyper_dev=self.hyper_dev.data_ptr(), have_sumsq=clip)
        eng.refresh_unconditionally()                                          # the next forward reads the re-derived copies

    PROBE = 4
    FORMS = ("plan", "eager", "graph")          # order of preference among forms that measure within 2 % of the fastest

    # ---- auto mode: which form of the step is fastest HERE?  Each is timed the way it will run — launches pipelined, the loss read
    # back a step late — over PROBE consecutive steps bracketed by device synchronisations:
    #   eager  every launch issued from Python: ~7 ms of host time per step (on a slow or busy host that is the bound);
    #   plan   the same launches on the same two streams, walked by ddpm_plan_run in C (csrc/plan.hip): eager semantics, no interpreter;
    #   graph  hipGraph replay: no host cost either, but the executor overlaps the two stream branches less well.
    def _available(self):
        """Forms this step can take here, in probing order."""
        forms = ["eager"]
        if self._plan_ok() and (self.x0.is_cuda or _TRAIN_GRAPH == "plan"):     # (host-emulated runs: only when asked for)
            forms.append("plan")
        if self.x0.is_cuda and not self.graph_failed and self.tr.input_source is None:
            forms.append("graph")
        return forms

    def _plan_ok(self):
        # loss_target of model_mean_type "mean" gathers its coefficients with torch ops: invisible to a plan
        return not self.plan_failed and self.tr.diffusion.model_mean_type in ("eps", "x_0")

    def _candidates(self):
        """Forms the auto-probe measures.  With a process group the hipGraph form is not a candidate: between its segments sit the
        communicator calls, its replay serialises the two stream branches the exchange overlaps with (one-rank RCCL: 11.2 ms against 9.9
        for the plan), and a capture next to a live communicator is the one form whose failure modes cannot be exercised on one GPU.
        DDPM_TORCH_AMD_TRAIN_GRAPH=1 still forces it."""
        forms = self._available()
        if self.unet.engine().pg is not None:
            forms = [f for f in forms if f != "graph"]
        return forms

    def _form(self):
        """The form the NEXT step takes."""
        forms = self._available()
        if _TRAIN_GRAPH != "auto":
            want = {False: "eager", True: "graph", "plan": "plan"}[_TRAIN_GRAPH]
            return want if want in forms else "eager"
        forms = self._candidates()
        if self.choice is not None:
            return self.choice if self.choice in forms else "eager"
        for f in forms:
            if f not in self.times:
                return f
        self._decide()
        return self.choice

    def _decide(self):
        best = min(self.times.values())
        self.choice = next(f for f in self.FORMS if f in self.times and self.times[f] <= 1.02 * best)
        # the forms that lost own a private allocator pool with a full set of the step's activations each: give that memory back
        if self.choice != "graph":
            self.graph = None
        if self.choice != "plan":
            self.plan = None

    def _wants_graph(self):
        return self._form() == "graph"

    def _probe_begin(self, kind):
        if _TRAIN_GRAPH == "auto" and self.choice is None and not self.x0.is_cuda:
            self.choice = "eager"                           # (host-emulated runs: nothing to measure)
        if _TRAIN_GRAPH == "auto" and self.choice is None and kind not in self.times and self._phase is None:
            torch.cuda.synchronize()
            self._phase = [kind, 0, time.perf_counter()]

    def _probe_end(self, kind):
        ph = self._phase
        if ph is None or ph[0] != kind:
            return
        ph[1] += 1
        if ph[1] == self.PROBE:
            torch.cuda.synchronize()
            self.times[kind] = (time.perf_counter() - ph[2]) / self.PROBE
            self._phase = None
            if all(f in self.times for f in self._candidates()):
                self._decide()

    def observe(self, seconds):
        """Kept for callers of the earlier interface: the probe now times whole phases itself."""

    def settled(self):
        """True once the form of the step (eager launches, launch plan or graph replay) is final and recorded / captured."""
        if not self.x0.is_cuda:
            return True
        if _TRAIN_GRAPH == "auto" and self.choice is None:
            return False
        form = self._form()
        if form == "graph":
            return self.graph is not None
        if form == "plan":
            return self.plan is not None
        return True

    def _identity(self):
        """Everything a captured or recorded step addresses by a raw pointer baked into its kernel arguments and that can be RE-CREATED
        behind its back: the fused update's pointer table (``optimizer.load_state_dict`` / ``load_checkpoint`` re-create the Adam moments
        and with them the table), the engine (``model.to()`` / ``.float()`` drop it: packed weights, workspaces, staging buffers)
        and the engine's derived-copy tables.  A replay after any of these changed would write through dangling pointers."""
        eng, fused = self.unet.engine(), self.tr._fused
        return (eng.serial, fused.generation, fused.table.data_ptr(), eng.pack_table.data_ptr(), eng.fc_table.data_ptr(),
                self.gflat.data_ptr(), self.unet.training, eng.pg is not None)

    def run(self, x):
        tr, eng = self.tr, self.unet.engine()
        self.x0.copy_(x, non_blocking=True)
        eng.ensure_fresh(need_dgrad=True)                                      # external writes since the last step (load_state_dict, EMA swap)
        if self.grad_ptrs is None:
            self.grad_ptrs = [self.gflat.data_ptr() + 4 * eng.goff[id(p)] for p in tr._fused_param_order()]
        params = tr._fused.prepare(grad_ptrs=self.grad_ptrs, stable_grads=True)
        assert len(params) == len(eng.params)
        self._write_hyper()
        ident = None
        if self.graph is not None or self.plan is not None:
            ident = self._identity()
            if self.graph is not None and self._baked != ident:
                self.graph = None                               # stale addresses: capture again (or run eagerly) instead of replaying
            if self.plan is not None and self._plan_baked != (ident, _hip.stream()):
                self.plan = None                                # ... and a plan also holds the stream handle it was recorded on
                # Back-off: a plan that keeps going stale (stream handle or identity churn) would be re-recorded every step — an
                # eager run + ~500 ctypes appends + a new MemPool each time — and an auto-probe waiting for plan samples would never settle.
                self.plan_stale += 1
                self._phase = None                              # (a probe sample that spans a re-recording is not a sample)
                if self.plan_stale >= 4:
                    warnings.warn("the launch plan of the training step went stale 4 times (baked addresses or the stream keep changing); running it eagerly")
                    self.plan_failed = True                      # (no longer a candidate: the probe decides among the other forms)
                    self.times.pop("plan", None)
                    if self.choice == "plan":
                        self.choice = None
        # capture / record only what has run eagerly once with THIS engine: its first backward builds descriptor tables with host ->
        # device copies (illegal under stream capture, invisible to a plan) and allocates the persistent slabs / staging buffers
        warm = self._warm == eng.serial and not (self.x0.is_cuda and torch.cuda.is_current_stream_capturing())
        form = self._form() if warm else "eager"
        self.last_kind = None
        fresh = False                                           # this step paid for a capture / recording: not a probe sample
        if form == "graph" and self.graph is None:
            g = SegmentedGraph(self.x0.device)
            g.register_generator(tr.generator)
            try:
                self.graph = g.capture(self.body, on_abort=eng.join_forked_streams)
                self.captures += 1
                self._baked = self._identity()
            except Exception as e:                        # capture not possible here: keep training without it
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); running it eagerly")
                torch.cuda.synchronize()
                del g
                _end_generator_capture(tr.generator, self.x0.device)
                self.graph_failed, self.graph = True, None
                if self.choice == "graph":
                    self.choice = None
                    self.times.pop("graph", None)
                form = "eager"
            fresh = True
        if form == "plan" and self.plan is None:
            # recording IS a step: the body runs eagerly once more while every call is copied into the plan
            self.draw()
            plan = LaunchPlan(self.x0.device)
            plan.record(lambda cut: self.body(cut, draw=False))          # (an exception of the body itself propagates, as it would eagerly)
            self.last_kind = "eager"
            if plan.build_error is None:
                self.plan = plan
                self.captures += 1
                self._plan_baked = (self._identity(), _hip.stream())
            else:                                                        # the step is done; only its replays are not to be had
                warnings.warn(f"the training step could not be turned into a launch plan ({plan.build_error}); running it eagerly")
                self.plan_failed = True
                if self.choice == "plan":
                    self.choice = None
                    self.times.pop("plan", None)
        elif form == "plan":
            self._probe_begin("plan")
            self.draw()
            self.plan.replay()
            self.last_kind = "plan"
            self._probe_end("plan")
        elif form == "graph" and self.graph is not None:
            if not fresh:
                self._probe_begin("graph")
            self.graph.replay()
            self.last_kind = "graph"
            if not fresh:
                self._probe_end("graph")
        else:
            if self.calls >= 1 and not fresh:
                self._probe_begin("eager")                  # (nor is the very first step: lazy initialisation)
            self.body()
            self.last_kind = "eager"
            self._warm = eng.serial
            if self.calls >= 1 and not fresh:
                self._probe_end("eager")
        self.calls += 1
        tr._fused.committed()
        eng.mark_fresh(bumped=True)
        return self.loss


class Trainer:
    def __init__(self, model, optimizer, diffusion, epochs, trainloader, sampler=None, scheduler=None, num_accum=1,
                 use_ema=False, grad_norm=1.0, shape=None, device=torch.device("cpu"), chkpt_intv=5, image_intv=1,
                 num_samples=64, ema_decay=0.9999, distributed=False, rank=0, dry_run=False):
        self.model, self.optimizer, self.diffusion = model, optimizer, diffusion
        self.epochs, self.start_epoch = epochs, 0
        self.trainloader, self.sampler = trainloader, sampler
        if shape is None:
            first = next(iter(trainloader))
            first = first[0] if isinstance(first, (list, tuple)) else first      # (image, label) pairs or bare image batches
            shape = first.shape[1:]
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
        self.stats = _StepStatistics(self._collect_loss, loss=None)
        self._loss_pending, self._loss_host, self._loss_slot, self._stat_stream = None, None, 0, None
        self._fused = _FusedUpdate(optimizer, self.ema)
        self._direct = {}                               # (input shape, train/eval) -> _DirectStep
        self._hi_stream = None                          # high-priority stream of the direct step (DDPM_MAIN_PRIORITY)
        self.input_source = None                        # optional fn(t_buf, noise_buf) filling the step's (t, noise) in place (parity tests)

    # ------------------------------------------------------------------ the reference's small accessors
    @property
    def timesteps(self):
        return self.diffusion.timesteps

    @property
    def current_stats(self):
        return self.stats.extract()

    @property
    def trainees(self):
        names = ["model", "optimizer"]
        if self.use_ema:
            names.append("ema")
        if self.scheduler is not None:
            names.append("scheduler")
        return names

    @property
    def _ema_on(self):
        return self.use_ema and isinstance(self.ema, EMA)

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

    # ------------------------------------------------------------------ one optimisation step
    def _fused_param_order(self):
        return [p for p in self.optimizer.param_groups[0]["params"] if p.requires_grad]

    def _direct_unet(self):
        """The UNet behind ``self.model`` when the autograd-free step applies, else None."""
        from ..models.unet import UNet
        m = self.model
        group = self.optimizer.param_groups[0] if self.optimizer.param_groups else None
        # (the scans below walk ~300 parameters several times: 0.25 ms per step — done once per (model, engine, optimizer, parameter list,
        #  number of trainable parameters, hyper-flags))
        key = (id(m), id(getattr(m, "_eng", None)), id(self.optimizer), len(self.optimizer.param_groups), id(group["params"]) if group else 0,
               len(group["params"]) if group else 0, sum(1 for q in group["params"] if q.requires_grad) if group else 0, self.num_accum, type(self.optimizer),
               tuple(bool(group.get(k)) for k in ("amsgrad", "maximize", "weight_decay", "capturable", "differentiable", "fused")) if group else (),
               "get_input" in self.__dict__, "loss" in self.__dict__, id(self.diffusion), getattr(self.diffusion, "loss_type", None),
               getattr(self.diffusion, "model_mean_type", None))
        cached = getattr(self, "_direct_unet_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        result = self._direct_unet_scan(m, UNet)
        self._direct_unet_cache = (key, result)
        return result

    def _direct_unet_scan(self, m, UNet):
        if not isinstance(m, UNet) or self.num_accum != 1 or not self._fused.eligible():
            return None
        if "get_input" in self.__dict__ or "loss" in self.__dict__ or type(self).get_input is not Trainer.get_input or type(self).loss is not Trainer.loss:
            return None                                 # a customised input / loss: keep the generic autograd step
        if not getattr(self.diffusion, "supports_direct_step", False) or not self.diffusion.supports_direct_step():
            return None
        own = m.engine().params
        opt = self._fused_param_order()
        if len(own) != len(opt) or {id(p) for p in own} != {id(p) for p in opt}:
            return None
        return m

    def step(self, x, global_steps=1):
        t_begin = time.perf_counter()
        x = x.to(self.device)
        unet = self._direct_unet() if os.environ.get("DDPM_TORCH_AMD_DIRECT_STEP", "1") != "0" else None
        direct = None
        if unet is not None and x.dtype == torch.float32:
            key = (tuple(x.shape), bool(unet.training))
            if key not in self._direct:
                self._direct[key] = _DirectStep(self, unet, key[0])
            direct = self._direct[key]
            if _MAIN_PRIORITY and x.is_cuda:
                # The step's critical chain on a HIGH-priority HIP stream (the priority range of this part is 0 .. -1, so the weight-
                # gradient stream cannot be made lower than a default stream — the main chain has to be made higher): when a leaf
                # kernel of the side stream and the next kernel of the chain become ready together, the chain gets the compute units.
                if self._hi_stream is None:
                    self._hi_stream = torch.cuda.Stream(device=x.device, priority=-1)
                cur = torch.cuda.current_stream(x.device)
                self._hi_stream.wait_stream(cur)
                with torch.cuda.stream(self._hi_stream):
                    loss = direct.run(x).clone()
                cur.wait_stream(self._hi_stream)
            else:
                loss = direct.run(x).clone()
            if self._ema_on:
                self.ema.num_updates += 1
            self.scheduler.step()
        else:
            loss = self.loss(x).mean()
            loss.div(self.num_accum).backward()
            if global_steps % self.num_accum == 0:
                ema_w = self.ema.weight_of_next_update() if self._ema_on else 0.0
                if self._fused(self.grad_norm, ema_w):          # clip + Adam + EMA in two launches (after DDP averaging)
                    if self._ema_on:
                        self.ema.num_updates += 1
                else:
                    nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.grad_norm)
                    self.optimizer.step()
                    if self._ema_on:
                        self.ema.update()
                self.optimizer.zero_grad(set_to_none=True)
                self.scheduler.step()
            loss = loss.detach()
        off_chain = self.distributed and _ASYNC_LOSS and _ASYNC_LOSS_REDUCE and loss.is_cuda
        if self.distributed and not off_chain:
            dist.reduce(loss, dst=0, op=dist.ReduceOp.SUM)
            loss.div_(self.world_size)
        if _ASYNC_LOSS and loss.is_cuda:
            # The reference reads the loss back with .item() at this point (utils/train.py:170) and so parks the host until the GPU has
            # finished the step — after which the GPU idles while the host prepares the next one.  Same read-back, one step later: the
            # value goes to pinned memory asynchronously and is added to the statistics when the NEXT step gets here (or when the
            # statistics are read or reset), by which time it has long arrived.
            self._collect_loss()
            if self._loss_host is None:
                self._loss_host = [torch.zeros((), dtype=torch.float32).pin_memory() for _ in range(2)]
            buf = self._loss_host[self._loss_slot]
            self._loss_slot ^= 1
            if off_chain:
                # Data parallel: the loss reduce of the reference (utils/train.py:166-169) is a collective at the END of every step; waited
                # for on the step's own stream, its latency sits between this step's update and the next step's forward.  Nothing on the
                # chain needs the reduced value: the collective is issued asynchronously (the communicator orders it behind the step),
                # and the division + read-back follow it on a stream of their own.
                if self._stat_stream is None:
                    self._stat_stream = torch.cuda.Stream(device=loss.device)
                work = dist.reduce(loss, dst=0, op=dist.ReduceOp.SUM, async_op=True)
                with torch.cuda.stream(self._stat_stream):
                    work.wait()                              # (this stream waits for the communicator; the host and the step's stream do not)
                    loss.div_(self.world_size)
                    buf.copy_(loss, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                loss.record_stream(self._stat_stream)
            else:
                buf.copy_(loss, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._loss_pending = (x.shape[0], ev, buf)
        else:
            self._collect_loss()
            RunningStatistics.update(self.stats, x.shape[0], loss=loss.item() * x.shape[0])      # the host waits for the step here
        if direct is not None:
            direct.observe(time.perf_counter() - t_begin)

    def _collect_loss(self):
        """Add the loss of the step whose read-back is still pending (if any) to the statistics."""
        if self._loss_pending is not None:
            n, ev, buf = self._loss_pending
            self._loss_pending = None
            ev.synchronize()
            RunningStatistics.update(self.stats, n, loss=float(buf) * n)

    # ------------------------------------------------------------------ sampling with the (EMA) weights
    def sample_fn(self, sample_size=None, noise=None, diffusion=None, sample_seed=None):
        per_rank = ((sample_size // self.world_size,) + self.shape) if noise is None else tuple(noise.shape)
        process = self.diffusion if diffusion is None else diffusion
        with self.ema:
            sample = process.p_sample(denoise_fn=self.model, shape=per_rank, device=self.device, noise=noise, seed=sample_seed)
        if self.distributed:
            parts = [torch.zeros(per_rank, device=self.device) for _ in range(self.world_size)]
            dist.all_gather(parts, sample)
            sample = torch.cat(parts, dim=0)
        assert sample.grad is None
        return sample

    # ------------------------------------------------------------------ epoch loop
    def _epoch(self, e, first_global_step):
        """One pass over the loader; returns (#steps done, last running statistics)."""
        self.stats.reset()
        self.model.train()
        if hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(e)                   # DistributedSampler reshuffles per epoch
        steps, seen = first_global_step, {}
        for batch in self.trainloader:
            images = batch[0] if isinstance(batch, (list, tuple)) else batch       # unconditional: labels are dropped
            steps += 1
            self.step(images.to(self.device), global_steps=steps)
            if self.dry_run and steps % self.num_accum == 0:
                break
        # (the reference refreshes a progress-bar postfix from current_stats after every step, utils/train.py:209-212; reading the
        #  statistics drains the pending loss read-back and would park the host on the device in every step, so they are read once,
        #  when the epoch is over — callers that want a live figure use ``peek_stats``)
        return steps, dict(self.current_stats) if steps > first_global_step else {}

    def peek_stats(self):
        """Running statistics WITHOUT the step whose loss is still on its way from the device (no host wait)."""
        return RunningStatistics.extract(self.stats) if self.stats.count else {}

    def _write_samples(self, e, image_dir):
        self.model.eval()
        x = self.sample_fn(sample_size=self.num_samples, sample_seed=self.sample_seed).cpu()
        if self.is_leader:
            from . import save_image_grid
            save_image_grid(x, os.path.join(image_dir, f"{e + 1}.jpg"), nrow=max(1, math.floor(math.sqrt(self.num_samples))))

    def train(self, evaluator=None, chkpt_path=None, image_dir=None):
        if self.num_samples:
            assert self.num_samples % self.world_size == 0, "Number of samples should be divisible by WORLD_SIZE!"
        first, last = (0, 1) if self.dry_run else (self.start_epoch, self.epochs)
        self.start_epoch, self.epochs = first, last
        global_steps = 0
        for e in range(first, last):
            global_steps, results = self._epoch(e, global_steps)
            done = e + 1
            if image_dir and self.num_samples and done % self.image_intv == 0:
                self._write_samples(e, image_dir)
            if chkpt_path and done % self.chkpt_intv == 0:
                self.model.eval()
                if evaluator is not None:
                    results.update(evaluator.eval(self.sample_fn, is_leader=self.is_leader))
                if self.is_leader:
                    self.save_checkpoint(chkpt_path, epoch=done, **results)
            if self.distributed:
                dist.barrier()

    # ------------------------------------------------------------------ checkpoints (reference layout)
    def named_state_dicts(self):
        for name in self.trainees:
            yield name, getattr(self, name).state_dict()

    def save_checkpoint(self, chkpt_path, **extra_info):
        payload = {name: sd for name, sd in self.named_state_dicts()}
        payload.update(extra_info)
        if "epoch" in extra_info:                       # foo.pt / foo_12.pt -> foo_<epoch>.pt
            chkpt_path = re.sub(r"(_\d+)?\.pt", f"_{extra_info['epoch']}.pt", chkpt_path)
        torch.save(payload, chkpt_path)

    @staticmethod
    def _without_ddp_prefix(sd):
        """Keys written by a DistributedDataParallel-wrapped trainee carry ``module.``; accept either form."""
        return {(k[len("module."):] if isinstance(k, str) and k.startswith("module.") else k): v for k, v in sd.items()}

    def load_checkpoint(self, chkpt_path, map_location):
        chkpt = torch.load(chkpt_path, map_location=map_location)
        for name in self.trainees:
            target, saved = getattr(self, name), chkpt.get(name)
            if saved is None or not hasattr(target, "load_state_dict"):
                continue
            try:
                target.load_state_dict(saved)
            except RuntimeError:
                if name == "ema":
                    saved = dict(saved, shadow=self._without_ddp_prefix(saved["shadow"]))
                else:
                    saved = self._without_ddp_prefix(saved)
                target.load_state_dict(saved)
        self.start_epoch = chkpt["epoch"]
