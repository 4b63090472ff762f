"""Gaussian diffusion process with the reference's constructor and method contract
(``ddpm_torch/diffusion.py:13-243`` of tqch/ddpm-torch), re-built around device-resident tables.

The reference re-uploads an fp64 table from the host on every ``_extract`` call (two per ``q_sample``,
six or more per sampling step).  Here the fp64 tables are kept as attributes (same names; ``DDIM.from_ddpm``
and checkpoints of hyper-parameters read them), an fp32 copy is uploaded ONCE per device, and each of
``q_sample``, the eps-MSE loss and the ancestral step is a single fused HIP kernel that gathers its
per-sample coefficients by ``t`` on the device:

    x_t      = sqrt(ab_t) x_0 + sqrt(1 - ab_t) eps                                  (diffusion.py:92-97)
    loss_b   = mean_chw (eps - eps_hat)^2                                           (:236-239)
    x_{t-1}  = c1_t clamp(r_t x_t - m_t eps_hat, -1, 1) + c2_t x_t + 1[t>0] exp(logvar_t / 2) z   (:99-158)

Arithmetic is fp32 with the reference's operation order (no fused multiply-add contraction) so identical
(x_t, t, noise) give matching results.  No CPU path: CPU tensors raise.
"""
import contextlib
import os

import torch

from . import _hip
from ._graphs import SegmentedGraph

__all__ = ["get_beta_schedule", "GaussianDiffusion"]

_F64 = torch.float64


def get_beta_schedule(beta_schedule, beta_start, beta_end, timesteps, dtype=_F64):
    """Same schedules and names as diffusion.py:13-29."""
    def ramp(frac):
        out = torch.full((timesteps,), beta_end, dtype=dtype)
        k = int(timesteps * frac)
        out[:k] = torch.linspace(beta_start, beta_end, k, dtype=dtype)
        return out

    table = {
        "quad": lambda: torch.linspace(beta_start ** 0.5, beta_end ** 0.5, timesteps, dtype=dtype) ** 2,
        "linear": lambda: torch.linspace(beta_start, beta_end, timesteps, dtype=dtype),
        "warmup10": lambda: ramp(0.1),
        "warmup50": lambda: ramp(0.5),
        "const": lambda: torch.full((timesteps,), beta_end, dtype=dtype),
        "jsd": lambda: 1.0 / torch.linspace(timesteps, 1, timesteps, dtype=dtype),
    }
    if beta_schedule not in table:
        raise NotImplementedError(beta_schedule)
    betas = table[beta_schedule]()
    assert betas.shape == (timesteps,)
    return betas


_MEAN_CODE = {"eps": 0, "x_0": 1, "mean": 2}
_STEP_TABLES = ("sqrt_recip_alphas_bar", "sqrt_recip_m1_alphas_bar", "posterior_mean_coef1", "posterior_mean_coef2", "fixed_model_logvar")
_VLB_TABLES = ("sqrt_recip_alphas_bar", "sqrt_recip_m1_alphas_bar", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_logvar_clipped",
               "fixed_model_logvar")


class _AutogradMSE(torch.autograd.Function):
    """Per-sample eps-MSE with a fused backward (d/d pred only; the target is data)."""

    @staticmethod
    def forward(ctx, pred, target):
        B, n = pred.shape[0], pred[0].numel()
        loss = torch.empty(B, dtype=torch.float32, device=pred.device)
        _hip.call("ddpm_mse_fwd", pred.data_ptr(), target.data_ptr(), loss.data_ptr(), B, n, _hip.stream())
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, target = ctx.saved_tensors
        B, n = pred.shape[0], pred[0].numel()
        g = torch.empty_like(pred)
        _hip.call("ddpm_mse_bwd", pred.data_ptr(), target.data_ptr(), gloss.contiguous().float().data_ptr(), g.data_ptr(), B, n, _hip.stream())
        return g, None


class _AutogradVLB(torch.autograd.Function):
    """Per-sample variational-bound term (bits / dim) of the network output with a fused backward (d / d model_out only: x_0, x_t are data)."""

    @staticmethod
    def forward(ctx, model_out, x_0, x_t, t, tabs, mean_code, T):
        B, n = model_out.shape[0], model_out[0].numel()
        loss = torch.empty(B, dtype=torch.float32, device=model_out.device)
        _hip.call("ddpm_vlb_terms", x_0.data_ptr(), x_t.data_ptr(), model_out.data_ptr(), t.data_ptr(), *[tb.data_ptr() for tb in tabs],
                  loss.data_ptr(), 0, B, n, mean_code, 0, T, _hip.stream())
        ctx.save_for_backward(model_out, x_0, x_t, t)
        ctx.tabs, ctx.mean_code, ctx.T = tabs, mean_code, T
        return loss

    @staticmethod
    def backward(ctx, gloss):
        model_out, x_0, x_t, t = ctx.saved_tensors
        B, n = model_out.shape[0], model_out[0].numel()
        g = torch.empty_like(model_out)
        _hip.call("ddpm_vlb_terms_bwd", x_0.data_ptr(), x_t.data_ptr(), model_out.data_ptr(), t.data_ptr(), *[tb.data_ptr() for tb in ctx.tabs],
                  gloss.contiguous().float().data_ptr(), g.data_ptr(), B, n, ctx.mean_code, ctx.T, _hip.stream())
        return g, None, None, None, None, None, None


class GaussianDiffusion:
    def __init__(self, betas, model_mean_type, model_var_type, loss_type, **kwargs):
        assert isinstance(betas, torch.Tensor) and betas.dtype == _F64          # diffusion.py:42-43
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.timesteps = len(betas)
        self._build_tables(betas, torch.cumprod(1.0 - betas, dim=0), eta2=None)

    def _build_tables(self, betas, alphas_bar, eta2):
        """fp64 host tables under the reference's attribute names.

        eta2 is None : DDPM posterior (diffusion.py:49-73).
        eta2 given   : DDIM generalisation on a sub-sequence (ddim.py:61-92): variance scaled by eta^2,
                       logs floored at 1e-20, mean coefficients from Song et al. eq. 12.
        """
        one = torch.ones(1, dtype=_F64)
        ab, ab_prev = alphas_bar, torch.cat([one, alphas_bar[:-1]])
        ddim = eta2 is not None
        self.alphas_bar = ab
        self.sqrt_alphas_bar = ab.sqrt()
        self.sqrt_one_minus_alphas_bar = (1.0 - ab).sqrt()
        self.sqrt_recip_alphas_bar = (1.0 / ab).sqrt()
        self.sqrt_recip_m1_alphas_bar = (1.0 / ab - 1.0).sqrt()
        self.posterior_var = betas * (1.0 - ab_prev) / (1.0 - ab) * (eta2 if ddim else 1.0)

        def log_patched(head, tail):               # entry 0 takes the value of entry 1 (diffusion.py:65,71)
            v = torch.cat([head, tail])
            return torch.log(v.clip(min=1e-20) if ddim else v)

        self.posterior_logvar_clipped = log_patched(self.posterior_var[1:2], self.posterior_var[1:])
        if ddim:
            alphas = ab / ab_prev
            self.posterior_mean_coef2 = torch.sqrt(1 - ab - eta2 * betas) * torch.sqrt(1 - ab_prev) / (1.0 - ab)
            self.posterior_mean_coef1 = ab_prev.sqrt() * (1.0 - alphas.sqrt() * self.posterior_mean_coef2)
        else:
            self.posterior_mean_coef1 = betas * ab_prev.sqrt() / (1.0 - ab)
            self.posterior_mean_coef2 = (1.0 - betas).sqrt() * (1.0 - ab_prev) / (1.0 - ab)
        if self.model_var_type == "fixed-large":
            self.fixed_model_var = betas
            self.fixed_model_logvar = log_patched(self.posterior_var[1:2], betas[1:])
        elif self.model_var_type == "fixed-small":
            self.fixed_model_var = self.posterior_var
            self.fixed_model_logvar = self.posterior_logvar_clipped
        else:
            raise KeyError(self.model_var_type)        # "learned" is dead in the reference too (diffusion.py:70-73)
        self._dev = {}                                  # (table name, device) -> fp32 device copy
        return ab_prev

    # ------------------------------------------------------------------ device tables
    def _tab(self, name, device):
        key = (name, device)
        if key not in self._dev:
            self._dev[key] = getattr(self, name).to(torch.float32).to(device).contiguous()   # cast THEN gather (diffusion.py:83)
        return self._dev[key]

    @staticmethod
    def _extract(arr, t, x, dtype=torch.float32, device=torch.device("cpu"), ndim=4):
        """diffusion.py:75-84 (kept for callers that use it directly; the fused kernels do not)."""
        if x is not None:
            dtype, device, ndim = x.dtype, x.device, x.ndim
        out = torch.as_tensor(arr, dtype=dtype, device=device).gather(0, t)
        return out.reshape((-1,) + (1,) * (ndim - 1))

    @staticmethod
    def _prep(*tensors):
        _hip.require_cuda(*tensors)
        return [None if v is None else v.contiguous().float() for v in tensors]

    # ------------------------------------------------------------------ forward process / loss
    def q_sample(self, x_0, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_0)
        x_0, noise = self._prep(x_0, noise)
        t = t.contiguous()
        out = _hip.retain(torch.empty_like(x_0))
        B = x_0.shape[0]
        _hip.call("ddpm_q_sample", x_0.data_ptr(), noise.data_ptr(), t.data_ptr(), self._tab("sqrt_alphas_bar", x_0.device).data_ptr(),
             self._tab("sqrt_one_minus_alphas_bar", x_0.device).data_ptr(), out.data_ptr(), B, x_0[0].numel(), len(self.sqrt_alphas_bar), _hip.stream())
        return out

    def q_posterior_mean_var(self, x_0, x_t, t):
        e = self._extract
        mean = e(self.posterior_mean_coef1, t, x_0) * x_0 + e(self.posterior_mean_coef2, t, x_0) * x_t
        return mean, e(self.posterior_var, t, x_0), e(self.posterior_logvar_clipped, t, x_0)

    def _loss_term_bpd(self, denoise_fn, x_0, x_t, t, clip_denoised, return_pred):
        """diffusion.py:203-215: L_t in bits per dimension — KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) for t > 0, the discretized-Gaussian decoder
        NLL at t = 0 — per sample, one fused kernel behind the network (csrc/elementwise.hip `vlb_terms_kernel`).  With grad enabled and
        clip_denoised = False (how `train_losses` calls it) the result carries the fused backward to the network output."""
        model_out = denoise_fn(x_t, t)
        x_0, x_t, model_out = self._prep(x_0, x_t, model_out)
        t = t.contiguous()
        tabs = [self._tab(n_, x_0.device) for n_ in _VLB_TABLES]
        T, code = len(self.posterior_mean_coef1), _MEAN_CODE[self.model_mean_type]
        if torch.is_grad_enabled() and model_out.requires_grad:
            if clip_denoised or return_pred:
                raise NotImplementedError("the differentiable bound term is the training form: clip_denoised=False, return_pred=False")
            return _AutogradVLB.apply(model_out, x_0, x_t, t, tabs, code, T)
        B, n = x_0.shape[0], x_0[0].numel()
        loss = torch.empty(B, dtype=torch.float32, device=x_0.device)
        pred = torch.empty_like(x_0) if return_pred else None
        _hip.call("ddpm_vlb_terms", x_0.data_ptr(), x_t.data_ptr(), model_out.data_ptr(), t.data_ptr(), *[tb.data_ptr() for tb in tabs],
                  loss.data_ptr(), _hip.ptr(pred), B, n, code, int(bool(clip_denoised)), T, _hip.stream())
        return (loss, pred) if return_pred else loss

    def train_losses(self, denoise_fn, x_0, t, noise=None):
        """diffusion.py:217-243 -> per-sample losses [B]: "mse" (unweighted) or "kl" (the variational-bound term, :222-224)."""
        if self.loss_type not in ("mse", "kl"):
            raise NotImplementedError(self.loss_type)
        if noise is None:
            noise = torch.randn_like(x_0)
        x_t = self.q_sample(x_0, t, noise=noise)
        if self.loss_type == "kl":
            return self._loss_term_bpd(denoise_fn, x_0=x_0, x_t=x_t, t=t, clip_denoised=False, return_pred=False)
        target = self.loss_target(x_0, x_t, t, noise)
        model_out = denoise_fn(x_t, t)
        pred, target = self._prep(model_out, target)
        return _AutogradMSE.apply(pred, target)

    def loss_target(self, x_0, x_t, t, noise):
        """What the model output is regressed on (diffusion.py:225-236)."""
        if self.model_mean_type == "eps":
            return noise
        if self.model_mean_type == "x_0":
            return x_0
        if self.model_mean_type == "mean":
            # posterior mean from the device-resident tables (no host -> device upload: this runs inside captured steps)
            c1 = self._tab("posterior_mean_coef1", x_0.device).gather(0, t).reshape((-1,) + (1,) * (x_0.ndim - 1))
            c2 = self._tab("posterior_mean_coef2", x_0.device).gather(0, t).reshape((-1,) + (1,) * (x_0.ndim - 1))
            return c1 * x_0 + c2 * x_t
        raise NotImplementedError(self.model_mean_type)

    def supports_direct_step(self):
        """True when Trainer may run its autograd-free step: the loss is the plain MSE on a target that does not depend on
        the model (the reference's 'kl' loss would need the variance terms)."""
        return self.loss_type == "mse" and self.model_mean_type in _MEAN_CODE

    # ------------------------------------------------------------------ reverse process
    def _step(self, x_t, model_out, z, t, clip_denoised, want_pred):
        x_t, model_out, z = self._prep(x_t, model_out, z)
        dev = x_t.device
        out = torch.empty_like(x_t)
        pred = torch.empty_like(x_t) if want_pred else None
        tabs = [self._tab(n, dev).data_ptr() for n in _STEP_TABLES]
        _hip.call("ddpm_p_sample_step", x_t.data_ptr(), model_out.data_ptr(), z.data_ptr(), t.data_ptr(), *tabs, out.data_ptr(),
             _hip.ptr(pred), x_t.shape[0], x_t[0].numel(), _MEAN_CODE[self.model_mean_type], int(bool(clip_denoised)),
             len(self.posterior_mean_coef1), _hip.stream())
        return out, pred

    def p_mean_var(self, denoise_fn, x_t, t, clip_denoised, return_pred):
        """diffusion.py:107-138 (mean / variance are read off a zero-noise fused step)."""
        out = denoise_fn(x_t, t)
        mean, pred = self._step(x_t, out, torch.zeros_like(x_t), t, clip_denoised, True)
        var = self._tab("fixed_model_var", x_t.device).gather(0, t).reshape((-1,) + (1,) * (x_t.ndim - 1))
        logvar = self._tab("fixed_model_logvar", x_t.device).gather(0, t).reshape((-1,) + (1,) * (x_t.ndim - 1))
        return (mean, var, logvar, pred) if return_pred else (mean, var, logvar)

    def p_sample_step(self, denoise_fn, x_t, t, clip_denoised=True, return_pred=False, generator=None):
        """diffusion.py:152-158: one model call, one noise draw (also at t = 0, then masked), one fused update."""
        out = denoise_fn(x_t, t)
        noise = torch.empty_like(x_t).normal_(generator=generator)
        sample, pred = self._step(x_t, out, noise, t, clip_denoised, return_pred)
        return (sample, pred) if return_pred else sample

    def _model_t(self, t):
        return t

    def _sample_loop(self, denoise_fn, shape, device, noise, seed, on_step=None, z_stream=None):
        with self._time_tables(denoise_fn):
            return self._sample_loop_impl(denoise_fn, shape, device, noise, seed, on_step, z_stream)

    def _sample_loop_impl(self, denoise_fn, shape, device, noise, seed, on_step=None, z_stream=None):
        """The T-step (or S-step) loop.  ``z_stream`` (parity tests only) is an iterator of per-step noise tensors
        that replaces the generator draws, so an identical (x_T, z_1..z_T) stream can be injected.

        The step body (model call, noise draw, fused update, t -= 1) has static shapes, so it is captured ONCE into a
        hipGraph and replayed: at 32x32 the eager loop is bound by ~250 kernel launches per step, not by the GPU.
        The captured step is cached per (denoiser, shape) — later calls only re-seed and replay.
        RNG consumption (x_T first, then one z per step incl. t = 0) and results are identical to the eager loop;
        set DDPM_TORCH_AMD_GRAPH=0 to force eager."""
        device = torch.device(device)
        _hip.require_cuda(torch.empty(0, device=device))        # no CPU fallback: sampling runs on the GPU only
        B = (shape or noise.shape)[0]
        steps = self._num_steps()
        if (on_step is None and steps >= 4 and device.type == "cuda"
                and os.environ.get("DDPM_TORCH_AMD_GRAPH", "1") != "0" and not torch.cuda.is_current_stream_capturing()):
            done = self._graph_loop(denoise_fn, tuple(shape or noise.shape), device, noise, seed, steps, z_stream)
            if done is not None:
                return done
        rng = torch.Generator(device).manual_seed(seed) if seed is not None else None       # diffusion.py:164-166
        if noise is None:
            x_t = torch.empty(shape, device=device).normal_(generator=rng)                  # x_T first, then one z per step
        else:
            x_t = noise.to(device)
        t = torch.full((B,), steps - 1, dtype=torch.int64, device=device)
        x_t = x_t.contiguous().float().clone()
        for ti in range(steps - 1, -1, -1):
            t.fill_(ti)
            out = denoise_fn(x_t, self._model_t(t))
            z = torch.empty_like(x_t).normal_(generator=rng) if z_stream is None else next(z_stream).to(x_t)
            x_t, pred = self._step(x_t, out, z, t, True, on_step is not None)
            if on_step is not None:
                on_step(ti, pred)
        return x_t

    # ---- captured sampling step
    @contextlib.contextmanager
    def _time_tables(self, denoise_fn):
        """While a sampling loop (or the capture of its step) runs, the UNets under ``denoise_fn`` take their per-block time biases from
        the [T][sum Cout] table of all timesteps instead of running the embedding MLP every step (models/unet.py: time_table)."""
        engines = self._engines_of(denoise_fn) if os.environ.get("DDPM_TIME_TABLE", "1") != "0" else None
        for e in engines or ():
            e.enable_time_table(self.timesteps)
            e.tt_on = True
        try:
            yield
        finally:
            for e in engines or ():
                e.tt_on = False

    @staticmethod
    def _engines_of(denoise_fn):
        """Engines of every UNet of this package reachable from ``denoise_fn`` (their derived weight copies must be
        brought up to date eagerly before a captured step is replayed)."""
        from .models.unet import UNet
        if not isinstance(denoise_fn, torch.nn.Module):
            return None
        return [m.engine() for m in denoise_fn.modules() if isinstance(m, UNet)]

    def _graph_entry(self, denoise_fn, shape, device, default_rng, external_z=False):
        """Cached captured step for (denoiser, shape), capturing it on first use; None when capture is not possible.
        ``external_z`` (parity tests): the step does not draw its noise, the caller fills the captured ``z`` buffer before every replay."""
        cache = self.__dict__.setdefault("_sample_graphs", {})
        engines = self._engines_of(denoise_fn)
        # engine serials: model.to() / .float() / set_compute_dtype() re-create the engine (packed weights, workspaces) — a step
        # captured against the previous one points at freed memory and must never be replayed
        # ... and so is the time-bias table the step gathers from: a sampler with a longer schedule re-allocates it
        key = (id(denoise_fn), shape, str(device), (default_rng, external_z), getattr(denoise_fn, "training", None),
               tuple((e.serial, e.T, e.tt.data_ptr() if e.tt_on and e.tt is not None else 0) for e in engines) if engines else None)
        ent = cache.get(key)
        if ent is not None and ent["ref"]() is not denoise_fn:
            ent = None                                            # the id was recycled by another object
        if ent is None and engines:
            live = {e.serial for e in engines}
            for k in [k for k, v in cache.items() if v["ref"]() is denoise_fn and k[5] and {q[0] for q in k[5]} != live]:
                cache.pop(k)                                      # entries of this denoiser's earlier engines: dead weight (and dead pointers)
        if ent is None:
            ent = self._capture_sample_step(denoise_fn, shape, device, default_rng, external_z)
            if ent is not None and engines is not None:           # arbitrary callables are captured per call: nothing tells us when their weights change
                import weakref
                ent["ref"] = weakref.ref(denoise_fn)
                if len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                cache[key] = ent
        return ent

    def _graph_loop(self, denoise_fn, shape, device, noise, seed, steps, z_stream=None):
        """Replay the captured sampling step ``steps`` times; returns None if capture is not possible (the caller then
        runs the eager loop; no RNG state has been consumed).  ``z_stream`` (parity tests only): per-step noise tensors that are copied
        into the captured step's noise buffer instead of being drawn inside it."""
        ent = self._graph_entry(denoise_fn, shape, device, default_rng=seed is None, external_z=z_stream is not None)
        if ent is None:
            return None
        x_t, t, rng, graph = ent["x_t"], ent["t"], ent["rng"], ent["graph"]
        for e in self._engines_of(denoise_fn) or ():
            e.ensure_fresh()
        if rng is not None:
            rng.manual_seed(seed)
        if noise is None:
            x_t.normal_(generator=rng)                            # x_T first, then one z per step (diffusion.py:166-171)
        else:
            x_t.copy_(noise)
        t.fill_(steps - 1)
        for _ in range(steps):
            if z_stream is not None:
                ent["z"].copy_(next(z_stream))
            graph.replay()
        return x_t.clone()

    def prepare_sampler(self, denoise_fn, shape, device, seeded=True):
        """Capture (and cache) the hipGraph of one sampling step for ``(denoise_fn, shape)`` without running a chain, so that the
        first ``p_sample`` of a serving process does not pay for it.  Returns True when a captured step is ready.  Not in the
        reference (its loop is eager)."""
        device = torch.device(device)
        if device.type != "cuda" or os.environ.get("DDPM_TORCH_AMD_GRAPH", "1") == "0" or self._num_steps() < 4:
            return False
        with torch.inference_mode(), self._time_tables(denoise_fn):
            return self._graph_entry(denoise_fn, tuple(shape), device, default_rng=not seeded) is not None

    def _capture_sample_step(self, denoise_fn, shape, device, default_rng, external_z=False):
        dev = device
        x_t = torch.zeros(shape, dtype=torch.float32, device=dev)
        z = torch.empty_like(x_t)
        B, n = shape[0], x_t[0].numel()
        t = torch.full((B,), self._num_steps() - 1, dtype=torch.int64, device=dev)
        rng = None if default_rng else torch.Generator(dev)
        tabs = [self._tab(n_, dev) for n_ in _STEP_TABLES]
        mean_code = _MEAN_CODE[self.model_mean_type]
        T = len(self.posterior_mean_coef1)

        def body(cut=None):
            out = denoise_fn(x_t, self._model_t(t)).contiguous().float()
            if not external_z:
                z.normal_(generator=rng)
            _hip.call("ddpm_p_sample_step", x_t.data_ptr(), out.data_ptr(), z.data_ptr(), t.data_ptr(), *[tb.data_ptr() for tb in tabs],
                      x_t.data_ptr(), 0, B, n, mean_code, 1, T, _hip.stream())         # in place: each element is read once, then written
            _hip.call("ddpm_add_i64", t.data_ptr(), B, -1, _hip.stream())

        gen = rng if rng is not None else torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        snap_rng = gen.get_state()
        try:
            # warm-up on a side stream (lazy kernel / cache initialisation) on scratch state: nothing the caller sees is consumed
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            gen.set_state(snap_rng)
            graph = SegmentedGraph(dev)
            graph.register_generator(rng)
            graph.capture(body)
        except Exception as e:                                   # capture not possible for this denoise_fn: eager loop instead
            import warnings
            warnings.warn(f"hipGraph capture of the sampling step failed ({type(e).__name__}: {e}); running the eager loop")
            torch.cuda.synchronize(dev)
            gen.set_state(snap_rng)
            return None
        return {"x_t": x_t, "t": t, "z": z, "rng": rng, "graph": graph, "ref": lambda: denoise_fn}

    def _num_steps(self):
        return self.timesteps

    @torch.inference_mode()
    def p_sample(self, denoise_fn, shape=None, device=torch.device("cpu"), noise=None, seed=None):
        """diffusion.py:160-174."""
        return self._sample_loop(denoise_fn, shape, device, noise, seed)

    @torch.inference_mode()
    def p_sample_progressive(self, denoise_fn, shape, device=torch.device("cpu"), noise=None, pred_freq=10, seed=None, z_stream=None):
        """diffusion.py:176-198: the same loop, also returning pred_x0 of every ``pred_freq``-th step (on the host), filled from
        the back: ``preds[L-1]`` is the first one kept (t = T-1 when T is a multiple of pred_freq), ``preds[0]`` the one at
        t = pred_freq - 1.  The per-step read-back needs the eager loop (no graph replay).  ``z_stream`` (parity tests only): as in
        ``_sample_loop``."""
        shape = tuple(shape or noise.shape)
        B = shape[0]
        L = self._num_steps() // pred_freq
        preds = torch.zeros((L, B) + shape[1:], dtype=torch.float32)
        box = [L]

        def keep(ti, pred):
            if (ti + 1) % pred_freq == 0:
                box[0] -= 1
                preds[box[0]] = pred.cpu()

        x = self._sample_loop(denoise_fn, shape, device, noise, seed, on_step=keep, z_stream=z_stream)
        return x.cpu(), preds
