"""The op-level (G1), block-level (G2) and step-level (G5) fixtures written by the REFERENCE, driven straight through the
C-ABI entry points / the engine's block routines — no oracle and no emulator in between on the GPU
(tests/test_fixtures_gpu.py); the same cases run on the CPU with the ABI emulated (tests/test_host_emulated.py) so the
harness itself is exercised in the build container."""
import math

import torch

import ddpm_torch
from ddpm_torch import _hip
from ddpm_torch import _ops as ops
from ddpm_torch._ops import View
from ddpm_torch.functions import get_timestep_embedding
from tests.golden.recipes import check, rnd

TD = {0: torch.float32, 1: torch.bfloat16}


def nhwc(x, dev, dt=0, pad_to=None):
    """NCHW fp32 host tensor -> View over an NHWC device tensor (channels zero-padded to ``pad_to``)."""
    B, C, H, W = x.shape
    Cp = pad_to or C
    t = torch.zeros(B, H, W, Cp, dtype=TD[dt])
    t[..., :C] = x.permute(0, 2, 3, 1).to(TD[dt])
    return View(t.to(dev).contiguous(), B, H, W, Cp)


def pack_w(w, dev, dt=0, Cp=None):
    """[N][C][R][S] -> packed [N][R][S][Cp] in dtype."""
    N, C, R, S = w.shape
    Cp = Cp or C
    out = torch.zeros(N, R, S, Cp)
    out[..., :C] = w.permute(0, 2, 3, 1)
    return out.to(TD[dt]).to(dev).contiguous()


def conv(x, w, bias, dev, dt, Ho, Wo, stride=1, pad=(1, 1), upsample=0):
    N, C, R, S = w.shape
    vec = 16 // (2 if dt else 4)
    Cp = -(-C // vec) * vec
    xv = nhwc(x, dev, dt, Cp)
    wp = pack_w(w, dev, dt, Cp)
    y = torch.zeros(x.shape[0], N, Ho, Wo, device=dev)
    b = bias.to(dev)
    ops.conv2d(xv, wp.data_ptr(), y.data_ptr(), 0, N, R, S, Ho, Wo, stride=stride, pad_t=pad[0], pad_l=pad[1], upsample=upsample,
               bias=b.data_ptr(), out_mode=3)
    return y.cpu()


def run_g1(g, dev, dt=0):
    tol = 2e-5 if dt == 0 else 2.5e-2
    # GroupNorm(32, eps 1e-6) (+ SiLU)
    for k, rec in g.items():
        if not k.startswith("gn_"):
            continue
        rc = rec["x_recipe"]
        x = rnd(*rc["shape"], seed=rc["seed"], scale=rc["scale"], shift=rc["shift"])
        B, C, H, W = x.shape
        xv = nhwc(x, dev, dt)
        gamma, beta = rec["weight"].to(dev), rec["bias"].to(dev)
        ws = torch.zeros(ops.gn_workspace_floats(B, H * W, C, _hip.BF16 if dt else _hip.F32), device=dev)
        for silu in (False, True):
            yv = View.new(B, H, W, C, TD[dt], dev)
            ops.gn_fwd(xv, yv, gamma, beta, None, ws, silu=silu)
            if silu:
                check(yv.to_nchw().cpu(), rec["y_silu_digest"], tol * 2, name=k + ".silu")
            else:
                check(yv.to_nchw().cpu(), rec["y"], tol, name=k)
    # convolutions: 3x3 (incl. 3 input / 3 output channels), SAME-pad stride 2 on even and odd sizes, 1x1, upsample + conv
    for k in ("conv3_3_32", "conv3_32_3", "conv3_64_32"):
        r = g[k]
        check(conv(r["x"], r["weight"], r["bias"], dev, dt, 8, 8), r["y"], tol, name=k)
    for k, pad in (("down_hw8", (0, 0)), ("down_hw9", (1, 1))):
        r = g[k]
        check(conv(r["x"], r["weight"], r["bias"], dev, dt, r["y"].shape[2], r["y"].shape[3], stride=2, pad=pad), r["y"], tol, name=k)
    r = g["conv1x1"]
    check(conv(r["x"], r["weight"], r["bias"], dev, dt, 4, 4, pad=(0, 0)), r["y"], tol, name="conv1x1")
    r = g["up_conv"]
    check(conv(r["x"], r["weight"], r["bias"], dev, dt, 8, 8, upsample=1), r["y"], tol, name="up_conv")
    # attention core  softmax(q k^T / sqrt(C)) v : the three-launch path, and the fused kernel where it applies
    for k in ("qkv_C64_L16", "qkv_C256_L256"):
        r = g[k]
        q, kk, v = (rnd(*r["shape"], seed=s_) for s_ in r["seeds"])
        B, C, H, W = q.shape
        L = H * W
        qkv = torch.cat([q, kk, v], 1)
        qv = nhwc(qkv, dev, dt)
        es = qv.base.element_size()
        logits = torch.empty(B, L, L, device=dev)
        ops.gemm(qv.ptr, 3 * C, L * 3 * C, 0, qv.ptr + C * es, 3 * C, L * 3 * C, 0, logits.data_ptr(), L, L * L, L, L, C, qv.dtype, batch=B,
                 alpha=1.0 / math.sqrt(C), out_mode=1)
        prob = torch.empty(B, L, L, dtype=TD[dt], device=dev)
        _hip.call("ddpm_softmax_fwd", logits.data_ptr(), prob.data_ptr(), B * L, L, qv.dtype, _hip.stream())
        o = View.new(B, H, W, C, TD[dt], dev)
        ops.gemm(prob.data_ptr(), L, L * L, 0, qv.ptr + 2 * C * es, 3 * C, L * 3 * C, 1, o.ptr, C, L * C, L, C, L, qv.dtype, batch=B)
        check(o.to_nchw().cpu(), r["out"], tol, name=k)
        if dt == 1 and C in (128, 256) and L % 128 == 0:
            o2 = View.new(B, H, W, C, TD[dt], dev)
            _hip.call("ddpm_attention_fwd", qv.ptr, qv.ld, o2.ptr, o2.ld, B, L, C, 1.0 / math.sqrt(C), qv.dtype, _hip.stream())
            check(o2.to_nchw().cpu(), r["out"], tol, name=k + ".fused")
    if dt == 0:
        for dim in (128, 127):
            r = g[f"temb_{dim}"]
            check(get_timestep_embedding(r["t"].to(dev), dim).cpu(), r["emb"], 2e-6, name=f"temb{dim}")


TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1, apply_attn=[False, True], drop_rate=0.0)


def run_g2(g, dev, tol=2e-5, gtol=2e-4):
    """ResidualBlock(32 -> 64, embed 128) and AttentionBlock(64) of the reference (fwd, d/dx, d/dt_emb, every parameter
    gradient): the fixture's weights are loaded into the matching blocks of a tiny UNet and the ENGINE's block routines are
    run on the fixture's inputs (nothing else of the network executes)."""
    torch.manual_seed(0)
    m = ddpm_torch.UNet(**TINY).to(dev)
    m.train()
    pair = m.downsamples["level_1"][0]
    rb, ab = pair[0], pair[1]
    rb.load_state_dict(g["res"]["sd"])
    ab.load_state_dict(g["attn"]["sd"])
    eng = m.engine()
    eng.ensure_fresh(need_dgrad=True)
    E = eng.E

    def flat_grads(gflat, mod, prefix=""):
        return {prefix + k: gflat[eng.goff[id(p)]:eng.goff[id(p)] + p.numel()].view(p.shape).cpu() for k, p in mod.named_parameters()}

    # ---- residual block
    r = g["res"]
    B, _, H, W = r["x"].shape
    t_emb = r["t_emb"].to(dev).contiguous()
    s_t = torch.empty_like(t_emb)
    _hip.call("ddpm_silu_fwd", t_emb.data_ptr(), s_t.data_ptr(), B * E, _hip.stream())
    fc_w, fc_b = eng._fc_all()
    tb = eng._linear(s_t, fc_w, fc_b, B, eng.tb_total, E)
    st = dict(B=B, ws=eng._workspace(B, H, W), save=True, training=True, tape=[], drop_p=0.0, seed=0, seed_dev=0, tb=tb)
    zeros = torch.zeros(B, E, device=dev)
    st["temb_saved"] = (torch.zeros(B, eng.hid, device=dev), zeros, zeros, t_emb, s_t, fc_w)
    xv = nhwc(r["x"], dev)
    out = View.new(B, H, W, 64, torch.float32, dev)
    eng._res(st, rb, xv, out)
    check(out.to_nchw().cpu(), r["y"], tol, name="res.y")
    ctx = eng._open_backward(st)
    out.grad, out.ginit = nhwc(r["gy"], dev), True
    eng._res_bwd(ctx, st["tape"][-1])
    gflat = eng._close_backward(ctx, st)
    check(xv.grad.to_nchw().cpu(), r["gx"], gtol, name="res.gx")
    check(ctx["dt_emb"].cpu(), r["gt_emb"], gtol, name="res.gt_emb")
    got = flat_grads(gflat, rb)
    for k, v in r["grads"].items():
        check(got[k], v, gtol, atol=1e-6, name="res." + k)
    # ---- attention block
    r = g["attn"]
    B, C, H, W = r["x"].shape
    st = dict(B=B, ws=eng._workspace(B, H, W), save=True, training=True, tape=[], drop_p=0.0, seed=0, seed_dev=0, tb=tb)
    st["temb_saved"] = (torch.zeros(B, eng.hid, device=dev), zeros, zeros, t_emb, s_t, fc_w)
    xv = nhwc(r["x"], dev)
    out = View.new(B, H, W, C, torch.float32, dev)
    eng._attn(st, ab, xv, out)
    check(out.to_nchw().cpu(), r["y"], tol, name="attn.y")
    ctx = eng._open_backward(st)
    out.grad, out.ginit = nhwc(r["gy"], dev), True
    eng._attn_bwd(ctx, st["tape"][-1])
    gflat = eng._close_backward(ctx, st)
    check(xv.grad.to_nchw().cpu(), r["gx"], gtol, name="attn.gx")
    got = flat_grads(gflat, ab)
    for k, v in r["grads"].items():
        check(got[k], v, gtol, atol=1e-6, name="attn." + k)


def run_g5(g, g3, dev):
    """q_sample / eps-MSE / p_mean_var / p_sample_step on fixed (x_t, t, z) incl. t = 0 and t = T-1, with the closed-form
    denoiser and with the tiny UNet (reference outputs)."""
    from oracle import unet_ref as U
    betas = ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    torch.manual_seed(g3["tiny_init_seed"])
    m = ddpm_torch.UNet(**g3["tiny_cfg"])
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), g3["tiny_rand_seed"]))
    m.to(dev).eval()
    for vt in ("fixed-small", "fixed-large"):
        r = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in g[vt].items()}
        dif = ddpm_torch.GaussianDiffusion(betas, "eps", vt, "mse")
        check(dif.q_sample(r["x0"], r["t"], r["noise"]).cpu(), g[vt]["x_t"], 1e-6, name="x_t")
        lin = lambda x, t: 0.1 * x + 0.01 * t.reshape(-1, 1, 1, 1).to(x)
        check(dif.train_losses(lin, r["x0"], r["t"], noise=r["noise"]).cpu(), g[vt]["loss_lin"], 1e-5, name="loss_lin")
        xp, px0 = dif._step(r["x_t"], lin(r["x_t"], r["t"]), r["z"], r["t"], True, True)
        check(xp.cpu(), g[vt]["x_prev_lin"], 1e-5, name="x_prev_lin")
        check(px0.cpu(), g[vt]["pred_x0_lin"], 1e-5, name="pred_x0_lin")
        mean, var, logvar, _ = dif.p_mean_var(lin, r["x_t"], r["t"], True, True)
        check(mean.cpu(), g[vt]["mean_lin"], 1e-5, name="mean_lin")
        check(var.cpu(), g[vt]["var"], 1e-6, name="var")
        check(logvar.cpu(), g[vt]["logvar"], 1e-6, name="logvar")
        with torch.no_grad():
            check(dif.train_losses(m, r["x0"], r["t"], noise=r["noise"]).cpu(), g[vt]["loss_unet"], 2e-4, name="loss_unet")
            mean, _, _, px0 = dif.p_mean_var(m, r["x_t"], r["t"], True, True)
        check(mean.cpu(), g[vt]["mean_unet"], 1e-3, name="mean_unet")
        check(px0.cpu(), g[vt]["pred_x0_unet"], 1e-3, name="pred_x0_unet")
