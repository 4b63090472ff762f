"""TEST INFRASTRUCTURE: a host-memory emulator of the C ABI in include/ddpm_hip.h.

Purpose: exercise the product's HOST logic (engine orchestration, pitches, gradient fan-in, weight caches, the
autograd wiring, Trainer / EMA) in the CPU-only container, where the HIP kernels cannot run.  Each ``ddpm_*`` entry
point is re-stated with numpy/torch over the raw host pointers the engine passes (CPU tensors have real addresses).
It is installed by monkeypatching ``ddpm_torch._hip`` from a test fixture; the product never imports this file and
has no switch to enable it.  It says nothing about the kernels themselves — those are checked on the GPU (-m gpu).
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F

F32, BF16 = 0, 1


def _np(ptr, count, ctype, nptype):
    if count <= 0:
        return np.zeros(0, dtype=nptype)
    return np.frombuffer((ctype * count).from_address(ptr), dtype=nptype)


def f32(ptr, count):
    return _np(ptr, count, ctypes.c_float, np.float32)


def i64(ptr, count):
    return _np(ptr, count, ctypes.c_longlong, np.int64)


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _f32_to_bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = u + (0x7FFF + ((u >> 16) & 1))
    return (r >> 16).astype(np.uint16)


class Mat:
    """[rows][cols] matrix with a row pitch, in fp32 or bf16 host memory."""

    def __init__(self, ptr, rows, cols, ld, dcode):
        self.rows, self.cols, self.ld, self.dcode = rows, cols, ld, dcode
        n = (rows - 1) * ld + cols if rows > 0 else 0
        raw = _np(ptr, n, ctypes.c_uint16, np.uint16) if dcode == BF16 else f32(ptr, n)
        self.view = np.lib.stride_tricks.as_strided(raw, (rows, cols), (ld * raw.itemsize, raw.itemsize), writeable=True) if n else raw.reshape(0, cols)

    def get(self):
        return _bf16_to_f32(self.view) if self.dcode == BF16 else self.view.astype(np.float32)

    def set(self, values):
        values = np.asarray(values, dtype=np.float32).reshape(self.rows, self.cols)
        self.view[...] = _f32_to_bf16(values).reshape(self.rows, self.cols) if self.dcode == BF16 else values


def _keep_mask(seed, idx, thresh16):
    """numpy mirror of dropout_keep() in csrc/common.h: one hash word per pair of adjacent elements, 16 bits each."""
    def mix32(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16); x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
        x ^= x >> np.uint32(15); x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return x
    idx = idx.astype(np.uint64)
    pair = idx >> np.uint64(1)
    lo = (pair & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (pair >> np.uint64(32)).astype(np.uint32)
    s_lo, s_hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        w = mix32(lo ^ mix32((hi + s_lo).astype(np.uint32)) ^ np.uint32((int(s_hi) * 0x9E3779B9) & 0xFFFFFFFF))
    field = np.where((idx & np.uint64(1)) == 1, w >> np.uint32(16), w & np.uint32(0xFFFF))
    return field >= np.uint32(thresh16)


def _thresh(p):
    th = p * 65536.0
    return 0 if th <= 0 else (65536 if th >= 65536.0 else int(th + 0.5))


def _canvas(x_nchw, Ho, Wo, R, S, stride, pad_t, pad_l, upsample, dilate):
    """Zero canvas on which a plain VALID stride-`stride` correlation reproduces the virtual-grid gather."""
    B, C, H, W = x_nchw.shape
    if upsample:
        xv = x_nchw.repeat_interleave(2, 2).repeat_interleave(2, 3)
    elif dilate:
        xv = torch.zeros(B, C, 2 * H - 1, 2 * W - 1, dtype=x_nchw.dtype)
        xv[:, :, ::2, ::2] = x_nchw
    else:
        xv = x_nchw
    need_h, need_w = (Ho - 1) * stride + R, (Wo - 1) * stride + S
    can = torch.zeros(B, C, need_h, need_w, dtype=x_nchw.dtype)
    hv, wv = xv.shape[2], xv.shape[3]
    # canvas index = v + pad
    y0, x0 = pad_t, pad_l
    ys, xs = max(0, -y0), max(0, -x0)
    ye, xe = min(hv, need_h - y0), min(wv, need_w - x0)
    if ye > ys and xe > xs:
        can[:, :, y0 + ys:y0 + ye, x0 + xs:x0 + xe] = xv[:, :, ys:ye, xs:xe]
    return can


class Emulator:
    """Callable table name -> python implementation; ``call(name, *args)`` mimics ddpm_torch._hip.call."""

    def __init__(self, real_lib=None):
        self.real_lib = real_lib
        self.log = []

    def call(self, name, *args):
        self.log.append(name)
        getattr(self, name)(*args)

    # ------------------------------------------------------------------ conv / gemm
    def ddpm_conv2d_nhwc(self, x, x_ld, w, y, y_ld, bias, rowbias, rb_ld, res, res_ld, B, H, W, C, Ho, Wo, N, R, S,
                         stride, pad_t, pad_l, ups, dil, acc, mode, splits, skws, skcnt, dt, st):
        xin = torch.from_numpy(Mat(x, B * H * W, C, x_ld, dt).get()).reshape(B, H, W, C).permute(0, 3, 1, 2)
        wt = torch.from_numpy(Mat(w, N, R * S * C, R * S * C, dt).get()).reshape(N, R, S, C).permute(0, 3, 1, 2)
        can = _canvas(xin, Ho, Wo, R, S, stride, pad_t, pad_l, ups, dil)
        out = F.conv2d(can, wt, stride=stride)                      # [B, N, Ho, Wo]
        assert out.shape[2:] == (Ho, Wo), (out.shape, Ho, Wo)
        out = out.permute(0, 2, 3, 1).reshape(B * Ho * Wo, N).numpy().copy()
        if bias:
            out += f32(bias, N)[None, :]
        if rowbias:
            rb = Mat(rowbias, B, N, rb_ld, F32).get()
            out += np.repeat(rb, Ho * Wo, axis=0)
        if res:
            out += Mat(res, B * Ho * Wo, N, res_ld, dt).get()
        if mode == 0:
            dst = Mat(y, B * Ho * Wo, N, y_ld, dt)
        elif mode == 1:
            dst = Mat(y, B * Ho * Wo, N, y_ld, F32)
        else:
            arr = f32(y, B * N * Ho * Wo).reshape(B, N, Ho * Wo)
            val = out.reshape(B, Ho * Wo, N).transpose(0, 2, 1)
            arr[...] = arr + val if acc else val
            return
        dst.set(dst.get() + out if acc else out)

    def ddpm_conv2d_wgrad_nhwc(self, dy, dy_ld, x, x_ld, dw, slab_stride, B, H, W, C, Creal, Ho, Wo, N, Nreal, R, S, stride, pad_t, pad_l, ups,
                               splits, dt, st):
        xin = torch.from_numpy(Mat(x, B * H * W, C, x_ld, dt).get()).reshape(B, H, W, C).permute(0, 3, 1, 2)
        g = torch.from_numpy(Mat(dy, B * Ho * Wo, N, dy_ld, dt).get()).reshape(B, Ho, Wo, N).permute(0, 3, 1, 2)[:, :Nreal]
        can = _canvas(xin, Ho, Wo, R, S, stride, pad_t, pad_l, ups, 0)
        gw = torch.nn.grad.conv2d_weight(can.contiguous(), (Nreal, C, R, S), g.contiguous(), stride=stride)
        val = gw[:, :Creal].permute(0, 2, 3, 1).numpy()                            # packed layout [n][r][s][c]
        if slab_stride == 0:
            f32(dw, Nreal * Creal * R * S).reshape(Nreal, R, S, Creal)[...] += val
        else:                                                                       # any partition over the copies is a valid result
            for c in range(splits):
                f32(dw + 4 * c * slab_stride, Nreal * Creal * R * S).reshape(Nreal, R, S, Creal)[...] = val if c == 0 else 0.0

    def ddpm_conv3x3_wgrad_nhwc(self, dy, dy_ld, x, x_ld, dw, slab_stride, dbias, bias_stride, B, H, W, C, N, Nreal, splits, dt, st):
        copies = self.real_lib.ddpm_conv3x3_wgrad_splits(B, H, W, C, N, splits) if self.real_lib is not None else max(splits, 1)
        assert copies > 0
        self.ddpm_conv2d_wgrad_nhwc(dy, dy_ld, x, x_ld, dw, slab_stride, B, H, W, C, C, H, W, N, Nreal, 3, 3, 1, 1, 1, 0, copies, dt, st)
        if dbias:
            v = Mat(dy, B * H * W, N, dy_ld, dt).get()[:, :Nreal].sum(0)
            if slab_stride == 0:
                f32(dbias, Nreal)[...] += v
            else:
                for c in range(copies):
                    f32(dbias + 4 * c * bias_stride, Nreal)[...] = v if c == copies - 1 else 0.0

    def ddpm_conv3x3_wgrad_up_nhwc(self, dy, dy_ld, x, x_ld, dw, slab_stride, dbias, bias_stride, B, H, W, C, N, Nreal, splits, dt, st):
        """H, W: dy's image; x stored at H/2 x W/2 and up-sampled (nearest) on the fly."""
        copies = self.real_lib.ddpm_conv3x3_wgrad_splits(B, H, W, C, N, splits) if self.real_lib is not None else max(splits, 1)
        assert copies > 0 and H % 2 == 0 and W % 2 == 0
        self.ddpm_conv2d_wgrad_nhwc(dy, dy_ld, x, x_ld, dw, slab_stride, B, H // 2, W // 2, C, C, H, W, N, Nreal, 3, 3, 1, 1, 1, 1, copies, dt, st)
        if dbias:
            v = Mat(dy, B * H * W, N, dy_ld, dt).get()[:, :Nreal].sum(0)
            if slab_stride == 0:
                f32(dbias, Nreal)[...] += v
            else:
                for c in range(copies):
                    f32(dbias + 4 * c * bias_stride, Nreal)[...] = v if c == copies - 1 else 0.0

    def ddpm_conv1x1_wgrad_nhwc(self, dy, dy_ld, x, x_ld, dw, slab_stride, dbias, bias_stride, P, C, N, splits, dt, st):
        g = Mat(dy, P, N, dy_ld, dt).get().astype(np.float64)
        v = Mat(x, P, C, x_ld, dt).get().astype(np.float64)
        full = (g.T @ v).astype(np.float32).reshape(-1)                 # [N][C]; the slab copies sum to it (here: all in the last one)
        for c in range(splits):
            f32(dw + 4 * c * slab_stride, N * C)[...] = full if c == splits - 1 else 0.0
            if dbias:
                f32(dbias + 4 * c * bias_stride, N)[...] = g.sum(0).astype(np.float32) if c == splits - 1 else 0.0

    def ddpm_wgrad_reduce(self, table, n, st):
        for src, dst, length, copies, stride in i64(table, 5 * n).reshape(n, 5):
            acc = np.zeros(int(length), dtype=np.float32)
            for c in range(int(copies)):
                acc += f32(int(src) + 4 * c * int(stride), int(length))
            f32(int(dst), int(length))[...] = acc

    def ddpm_wgrad_unpack(self, gpack, gflat, descs, n, scale, st):
        d = i64(descs, 5 * n).reshape(n, 5)
        for src, dst, N, C, RS in d:
            v = f32(gpack + 4 * int(src), int(N * C * RS)).reshape(N, RS, C).transpose(0, 2, 1)
            f32(gflat + 4 * int(dst), int(N * C * RS)).reshape(N, C, RS)[...] = v * np.float32(scale)

    def ddpm_wgrad_unpack_sumsq(self, gpack, gflat, descs, n, scale, total, total_floats, st):
        assert total_floats >= 64 + 64 * n
        self.ddpm_wgrad_unpack(gpack, gflat, descs, n, scale, st)
        tot = 0.0
        for src, dst, N, C, RS in i64(descs, 5 * n).reshape(n, 5):
            g = f32(gflat + 4 * int(dst), int(N * C * RS)).astype(np.float64)
            tot += float((g * g).sum())
        bank = f32(total, 64)
        bank[...] = 0
        bank[0] = np.float32(tot)

    def _operand(self, p, ld, bs, trans, rows, K, batch, dt):
        es = 2 if dt == BF16 else 4
        mats = []
        for b in range(batch):
            base = p + b * bs * es
            m = Mat(base, K, rows, ld, dt).get().T if trans else Mat(base, rows, K, ld, dt).get()
            mats.append(m)
        return np.stack(mats)

    def ddpm_gemm(self, a, a_ld, a_bs, a_tr, b, b_ld, b_bs, b_tr, c, c_ld, c_bs, bias, res, res_ld, res_bs, M, N, K, batch, alpha,
                  acc, mode, splits, dt, st):
        A = self._operand(a, a_ld, a_bs, a_tr, M, K, batch, dt)
        Bm = self._operand(b, b_ld, b_bs, b_tr, N, K, batch, dt)
        out = alpha * np.einsum("bmk,bnk->bmn", A.astype(np.float64), Bm.astype(np.float64)).astype(np.float32)
        if bias:
            out += f32(bias, N)[None, None, :]
        odt = dt if mode == 0 else F32
        es_r = 2 if dt == BF16 else 4
        es_o = 2 if odt == BF16 else 4
        for i in range(batch):
            v = out[i]
            if res:
                v = v + Mat(res + i * res_bs * es_r, M, N, res_ld, dt).get()
            dst = Mat(c + i * c_bs * es_o, M, N, c_ld, odt)
            dst.set(dst.get() + v if (acc or mode == 2) else v)

    # ------------------------------------------------------------------ group norm
    def _gn(self, x, gamma, beta, G, eps, silu, drop_p, seed, B, HW, C):
        t = x.reshape(B, HW, C).permute(0, 2, 1).reshape(B, C, HW, 1)
        y = F.group_norm(t, G, gamma, beta, eps)
        if silu:
            y = F.silu(y)
        if drop_p > 0:
            idx = np.arange(B * HW * C, dtype=np.uint64)
            keep = torch.from_numpy(_keep_mask(seed, idx, _thresh(drop_p)).astype(np.float32)).reshape(B, HW, C).permute(0, 2, 1).reshape(B, C, HW, 1)
            y = y * keep / (1.0 - drop_p)
        return y.reshape(B, C, HW).permute(0, 2, 1).reshape(B * HW, C)

    def ddpm_groupnorm_silu_fwd(self, x, x_ld, y, y_ld, gamma, beta, stats, ws, B, HW, C, G, eps, silu, drop_p, seed, seed_dev, dt, st):
        if seed_dev and drop_p > 0:
            seed = (seed + int(_np(seed_dev, 1, ctypes.c_uint64, np.uint64)[0])) & ((1 << 64) - 1)
        xin = torch.from_numpy(Mat(x, B * HW, C, x_ld, dt).get())
        g, b = torch.from_numpy(f32(gamma, C).copy()), torch.from_numpy(f32(beta, C).copy())
        out = self._gn(xin, g, b, G, eps, silu, drop_p, seed, B, HW, C)
        Mat(y, B * HW, C, y_ld, dt).set(out.numpy())
        if stats:
            xg = xin.reshape(B, HW, G, C // G).permute(0, 2, 1, 3).reshape(B, G, -1).double()
            mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
            s = f32(stats, B * G * 2).reshape(B, G, 2)
            s[..., 0] = mean.float().numpy(); s[..., 1] = (1.0 / torch.sqrt(var + eps)).float().numpy()

    def ddpm_groupnorm_silu_bwd(self, x, x_ld, dy, dy_ld, dx, dx_ld, gamma, beta, stats, dgamma, dbeta, ws, B, HW, C, G, silu, drop_p,
                                seed, seed_dev, acc, dx_colsum, colsum_ld, add, add_ld, dt, st):
        if seed_dev and drop_p > 0:
            seed = (seed + int(_np(seed_dev, 1, ctypes.c_uint64, np.uint64)[0])) & ((1 << 64) - 1)
        xin = torch.from_numpy(Mat(x, B * HW, C, x_ld, dt).get()).requires_grad_(True)
        g = torch.from_numpy(f32(gamma, C).copy()).requires_grad_(True)
        b = torch.from_numpy(f32(beta, C).copy()).requires_grad_(True)
        gy = torch.from_numpy(Mat(dy, B * HW, C, dy_ld, dt).get())
        with torch.enable_grad():                  # we are called from inside an autograd.Function.backward
            self._gn(xin, g, b, G, 1e-6, silu, drop_p, seed, B, HW, C).backward(gy)
        dst = Mat(dx, B * HW, C, dx_ld, dt)
        val = xin.grad.numpy()
        if add:
            val = val + Mat(add, B * HW, C, add_ld, dt).get()
        dst.set(dst.get() + val if acc else val)
        if dx_colsum:
            cs = Mat(dx_colsum, B, C, colsum_ld, F32)
            cs.set(cs.get() + dst.get().reshape(B, HW, C).sum(1))
        if dgamma:
            f32(dgamma, C)[...] += g.grad.numpy()
        if dbeta:
            f32(dbeta, C)[...] += b.grad.numpy()

    # ------------------------------------------------------------------ elementwise
    def ddpm_timestep_embedding(self, t, freqs, out, B, dim, st):
        tt = i64(t, B).astype(np.float32)
        fr = f32(freqs, dim // 2)
        arg = (tt[:, None] * fr[None, :]).astype(np.float32)
        o = f32(out, B * dim).reshape(B, dim)
        o[...] = 0
        o[:, :dim // 2] = np.sin(arg); o[:, dim // 2:2 * (dim // 2)] = np.cos(arg)

    def ddpm_nchw_to_nhwc(self, x, y, B, C, HW, Cp, dt, st):
        src = f32(x, B * C * HW).reshape(B, C, HW).transpose(0, 2, 1)
        out = np.zeros((B, HW, Cp), dtype=np.float32)
        out[..., :C] = src
        Mat(y, B * HW, Cp, Cp, dt).set(out.reshape(B * HW, Cp))

    def ddpm_pack_weight(self, w, wf, wd, N, C, R, S, Cp, Np, dt, st):
        src = f32(w, N * C * R * S).reshape(N, C, R, S)
        if wf:
            out = np.zeros((N, R, S, Cp), dtype=np.float32)
            out[..., :C] = src.transpose(0, 2, 3, 1)
            Mat(wf, N, R * S * Cp, R * S * Cp, dt).set(out.reshape(N, -1))
        if wd:
            out = np.zeros((C, R, S, Np), dtype=np.float32)
            out[..., :N] = src[:, :, ::-1, ::-1].transpose(1, 2, 3, 0)
            Mat(wd, C, R * S * Np, R * S * Np, dt).set(out.reshape(C, -1))

    def ddpm_pack_weight_multi(self, descs, n, dt, st):
        for w, wf, wd, N, C, R, Cp, Np in i64(descs, 8 * n).reshape(n, 8):
            w, wf, wd, N, C, R, Cp, Np = int(w), int(wf), int(wd), int(N), int(C), int(R), int(Cp), int(Np)
            if R & 0x100 and wd:                  # upsample conv: wd = 4x4 / stride-2 effective dgrad kernel [C][4][4][Np]
                self.ddpm_pack_weight(w, wf, 0, N, C, 3, 3, Cp, Np, dt, st)
                W = f32(w, N * C * 9).reshape(N, C, 3, 3)
                D = W[:, :, ::-1, ::-1]                                     # flipped taps: D[r][s] = W[2-r][2-s]
                E = np.zeros((N, C, 4, 4), dtype=np.float32)
                for a in range(2):
                    for b in range(2):
                        E[:, :, a:a + 3, b:b + 3] += D
                out = np.zeros((C, 4, 4, Np), dtype=np.float32)
                out[..., :N] = E.transpose(1, 2, 3, 0)
                Mat(wd, C * 16, Np, Np, dt).set(out.reshape(C * 16, Np))
            else:
                self.ddpm_pack_weight(w, wf, wd, N, C, R & 0xff, R & 0xff, Cp, Np, dt, st)

    def ddpm_q_sample(self, x0, noise, t, ca, cb, xt, B, n, T, st):
        tt = i64(t, B)
        assert 0 <= int(tt.min()) and int(tt.max()) < T
        a, b = f32(ca, T)[tt], f32(cb, T)[tt]
        f32(xt, B * n).reshape(B, n)[...] = a[:, None] * f32(x0, B * n).reshape(B, n) + b[:, None] * f32(noise, B * n).reshape(B, n)

    def ddpm_mse_fwd(self, pred, target, loss, B, n, st):
        d = f32(target, B * n).reshape(B, n) - f32(pred, B * n).reshape(B, n)
        f32(loss, B)[...] = (d * d).mean(1)

    def ddpm_weighted_sum_f32(self, x, w, out, n, st):
        f32(out, 1)[0] = np.float32((f32(x, n).astype(np.float64) * f32(w, n)).sum())

    def ddpm_mse_bwd(self, pred, target, gloss, gpred, B, n, st):
        d = f32(pred, B * n).reshape(B, n) - f32(target, B * n).reshape(B, n)
        f32(gpred, B * n).reshape(B, n)[...] = d * (2.0 / n) * f32(gloss, B)[:, None]

    def ddpm_p_sample_step(self, x_t, out, z, t, recip, recip_m1, c1, c2, logvar, x_prev, pred, B, n, mean_type, clip, T, st):
        tt = i64(t, B)
        assert 0 <= int(tt.min()) and int(tt.max()) < T
        g = lambda p: f32(p, T)[tt][:, None]
        xt, o, zz = (f32(p, B * n).reshape(B, n) for p in (x_t, out, z))
        if mean_type == 0:
            x0 = g(recip) * xt - g(recip_m1) * o
        elif mean_type == 1:
            x0 = o.copy()
        else:
            x0 = o / g(c1) - g(c2) / g(c1) * xt
        if clip:
            x0 = np.clip(x0, -1.0, 1.0)
        mean = o if mean_type == 2 else g(c1) * x0 + g(c2) * xt
        mask = (tt > 0).astype(np.float32)[:, None]
        f32(x_prev, B * n).reshape(B, n)[...] = mean + mask * np.exp(0.5 * g(logvar)) * zz
        if pred:
            f32(pred, B * n).reshape(B, n)[...] = x0

    def _vlb(self, x_0, x_t, out, t, recip, recip_m1, c1, c2, lv1, lv2, B, n, mean_type, clip, T, out_requires_grad):
        """diffusion.py:203-215 in torch on host views (fp32), returning (per-sample bits/dim, pred_x0, the `out` leaf)."""
        tt = torch.from_numpy(i64(t, B).copy())
        assert 0 <= int(tt.min()) and int(tt.max()) < T
        g = lambda p: torch.from_numpy(f32(p, T).copy())[tt][:, None]
        x0, xt = (torch.from_numpy(f32(p, B * n).reshape(B, n).copy()) for p in (x_0, x_t))
        o = torch.from_numpy(f32(out, B * n).reshape(B, n).copy()).requires_grad_(out_requires_grad)
        if mean_type == 0:
            pred = g(recip) * xt - g(recip_m1) * o
        elif mean_type == 1:
            pred = o
        else:
            pred = o / g(c1) - g(c2) / g(c1) * xt
        if clip:
            pred = pred.clamp(-1.0, 1.0)
        mean = o if mean_type == 2 else g(c1) * pred + g(c2) * xt
        true_mean = g(c1) * x0 + g(c2) * xt
        d = g(lv1) - g(lv2)
        kl = 0.5 * ((-1.0 - d) + (true_mean - mean) ** 2 * torch.exp(-g(lv2)) + torch.exp(d))
        cdf = lambda z: 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (z + 0.044715 * z ** 3)))
        inv_std = torch.exp(-0.5 * g(lv2))
        cu = torch.where(x0 > 0.999, torch.ones_like(x0), cdf(inv_std * (x0 - mean + 1.0 / 255)))
        cl = torch.where(x0 < -0.999, torch.zeros_like(x0), cdf(inv_std * (x0 - mean - 1.0 / 255)))
        nll = -torch.log(torch.clamp(cu - cl - 1e-12, min=0) + 1e-12)
        term = torch.where((tt > 0)[:, None], kl, nll).mean(1) / math.log(2.0)
        return term, pred, o

    def ddpm_vlb_terms(self, x_0, x_t, out, t, recip, recip_m1, c1, c2, lv1, lv2, loss, pred_x0, B, n, mean_type, clip, T, st):
        with torch.no_grad():
            term, pred, _ = self._vlb(x_0, x_t, out, t, recip, recip_m1, c1, c2, lv1, lv2, B, n, mean_type, clip, T, False)
        f32(loss, B)[...] = term.numpy()
        if pred_x0:
            f32(pred_x0, B * n).reshape(B, n)[...] = pred.numpy()

    def ddpm_vlb_terms_bwd(self, x_0, x_t, out, t, recip, recip_m1, c1, c2, lv1, lv2, gloss, gout, B, n, mean_type, T, st):
        with torch.enable_grad():               # (called from inside an autograd backward, where grad mode is off)
            term, _, o = self._vlb(x_0, x_t, out, t, recip, recip_m1, c1, c2, lv1, lv2, B, n, mean_type, 0, T, True)
            (term * torch.from_numpy(f32(gloss, B).copy())).sum().backward()
        f32(gout, B * n).reshape(B, n)[...] = o.grad.numpy()

    def ddpm_mt_gather_f32(self, table, n, st):
        for a, b, dst, numel in i64(table, 4 * n).reshape(n, 4):
            v = f32(int(a), int(numel)).copy()
            if b:
                v += f32(int(b), int(numel))
            f32(int(dst), int(numel))[...] = v

    def ddpm_mt_grad_sumsq(self, table, n, total, total_floats, st):
        assert total_floats >= 64 + 64 * n
        tot = 0.0
        for row in i64(table, 6 * n).reshape(n, 6):
            g = f32(int(row[1]), int(row[5])).astype(np.float64)
            tot += float((g * g).sum())
        bank = f32(total, 64)
        bank[...] = 0
        bank[0] = np.float32(tot)

    def ddpm_mt_adam_ema(self, table, n, total, max_norm, lr, b1, b2, eps, bc1, bc2, ema_w, hyper, st):
        if hyper:
            lr, bc1, bc2, ema_w = (float(v) for v in f32(hyper, 4))
        clip = 1.0
        if total and max_norm > 0:
            clip = min(1.0, max_norm / (float(np.sqrt(f32(total, 64).astype(np.float64).sum())) + 1e-6))
        for row in i64(table, 6 * n).reshape(n, 6):
            p_, g_, m_, v_, sh_, numel = (int(v) for v in row)
            p, g, m, v = f32(p_, numel), f32(g_, numel) * np.float32(clip), f32(m_, numel), f32(v_, numel)
            m[...] = b1 * m + (1 - b1) * g
            v[...] = b2 * v + (1 - b2) * g * g
            p[...] = p - (lr / bc1) * (m / (np.sqrt(v) / np.sqrt(bc2) + eps))
            if sh_:
                sh = f32(sh_, numel)
                sh[...] = sh + ema_w * (p - sh)

    def ddpm_gather_i64(self, idx, mp, out, B, st):
        ii = i64(idx, B)
        i64(out, B)[...] = i64(mp, int(ii.max()) + 1)[ii]

    def ddpm_gather_rows_f32(self, table, idx, out, rows, row_len, table_rows, st):
        ii = i64(idx, rows)
        tab = f32(table, table_rows * row_len).reshape(table_rows, row_len)
        o = f32(out, rows * row_len).reshape(rows, row_len)
        ok = (ii >= 0) & (ii < table_rows)
        o[...] = np.where(ok[:, None], tab[np.clip(ii, 0, table_rows - 1)], np.nan)

    def ddpm_add_i64(self, t, B, delta, st):
        i64(t, B)[...] += delta

    def ddpm_silu_fwd(self, x, y, n, st):
        v = torch.from_numpy(f32(x, n).copy())
        f32(y, n)[...] = F.silu(v).numpy()

    def ddpm_silu_bwd(self, x, dy, dx, n, acc, st):
        v = torch.from_numpy(f32(x, n).copy()).requires_grad_(True)
        with torch.enable_grad():
            F.silu(v).backward(torch.from_numpy(f32(dy, n).copy()))
        d = f32(dx, n)
        d[...] = d + v.grad.numpy() if acc else v.grad.numpy()

    def ddpm_colsum(self, dy, ld, per_sample, ps_ld, total, B, HW, C, dt, st):
        v = Mat(dy, B * HW, C, ld, dt).get().reshape(B, HW, C).sum(1)
        if per_sample:
            dst = Mat(per_sample, B, C, ps_ld, F32)
            dst.set(dst.get() + v)
        if total:
            f32(total, C)[...] += v.sum(0)

    def ddpm_upsample2x_bwd(self, dyu, dx, dx_ld, B, H, W, C, acc, dt, st):
        v = Mat(dyu, B * 4 * H * W, C, C, dt).get().reshape(B, H, 2, W, 2, C).sum((2, 4)).reshape(B * H * W, C)
        dst = Mat(dx, B * H * W, C, dx_ld, dt)
        dst.set(dst.get() + v if acc else v)

    def ddpm_resample2x_nhwc(self, x, x_ld, y, y_ld, B, H, W, C, up, scale, acc, dt, st):
        if not up:
            v = Mat(x, B * 4 * H * W, C, x_ld, dt).get().reshape(B, H, 2, W, 2, C).sum((2, 4)).reshape(B * H * W, C) * np.float32(scale)
            dst = Mat(y, B * H * W, C, y_ld, dt)
        else:
            v = Mat(x, B * H * W, C, x_ld, dt).get().reshape(B, H, 1, W, 1, C) * np.float32(scale)
            v = np.broadcast_to(v, (B, H, 2, W, 2, C)).reshape(B * 4 * H * W, C)
            dst = Mat(y, B * 4 * H * W, C, y_ld, dt)
        dst.set(dst.get() + v if acc else v)

    def ddpm_add_rows(self, x, x_ld, y, y_ld, rows, C, acc, dt, st):
        v = Mat(x, rows, C, x_ld, dt).get()
        dst = Mat(y, rows, C, y_ld, dt)
        dst.set(dst.get() + v if acc else v)

    def ddpm_softmax_fwd(self, s, p, rows, L, dt, st):
        v = torch.softmax(torch.from_numpy(f32(s, rows * L).reshape(rows, L).copy()), -1)
        Mat(p, rows, L, L, dt).set(v.numpy())

    def ddpm_attention_fwd(self, qkv, ld, out, out_ld, B, L, C, scale, dt, st):
        x = Mat(qkv, B * L, 3 * C, ld, dt).get().reshape(B, L, 3 * C).astype(np.float64)
        q, k, v = x[..., :C], x[..., C:2 * C], x[..., 2 * C:]
        s = np.einsum("bic,bjc->bij", q, k) * scale
        s = np.exp(s - s.max(-1, keepdims=True))
        p_ = s / s.sum(-1, keepdims=True)
        Mat(out, B * L, C, out_ld, dt).set(np.einsum("bij,bjc->bic", p_, v).reshape(B * L, C).astype(np.float32))

    def ddpm_attention_fwd_lse(self, qkv, ld, out, out_ld, lse, B, L, C, scale, dt, st):
        x = Mat(qkv, B * L, 3 * C, ld, dt).get().reshape(B, L, 3 * C).astype(np.float64)
        q, k, v = x[..., :C], x[..., C:2 * C], x[..., 2 * C:]
        s = np.einsum("bic,bjc->bij", q, k) * scale
        m = s.max(-1, keepdims=True)
        e = np.exp(s - m)
        if lse:
            f32(lse, B * L)[...] = (m[..., 0] + np.log(e.sum(-1))).reshape(-1)
        p_ = e / e.sum(-1, keepdims=True)
        Mat(out, B * L, C, out_ld, dt).set(np.einsum("bij,bjc->bic", p_, v).reshape(B * L, C).astype(np.float32))

    def ddpm_attention_bwd(self, qkv, ld, o, o_ld, d_o, do_ld, lse, dvec, dqkv, dqkv_ld, B, L, C, scale, dt, st):
        x = Mat(qkv, B * L, 3 * C, ld, dt).get().reshape(B, L, 3 * C).astype(np.float64)
        q, k, v = x[..., :C], x[..., C:2 * C], x[..., 2 * C:]
        O = Mat(o, B * L, C, o_ld, dt).get().reshape(B, L, C).astype(np.float64)
        dO = Mat(d_o, B * L, C, do_ld, dt).get().reshape(B, L, C).astype(np.float64)
        p_ = np.exp(np.einsum("bic,bjc->bij", q, k) * scale - f32(lse, B * L).reshape(B, L, 1))
        D = (dO * O).sum(-1)
        f32(dvec, B * L)[...] = D.reshape(-1)
        ds = p_ * (np.einsum("bic,bjc->bij", dO, v) - D[..., None]) * scale
        g = np.concatenate([np.einsum("bij,bjc->bic", ds, k), np.einsum("bij,bic->bjc", ds, q), np.einsum("bij,bic->bjc", p_, dO)], -1)
        Mat(dqkv, B * L, 3 * C, dqkv_ld, dt).set(g.reshape(B * L, 3 * C).astype(np.float32))

    def ddpm_softmax_bwd(self, p, dp, ds, rows, L, dt, st):
        P = Mat(p, rows, L, L, dt).get()
        d = f32(dp, rows * L).reshape(rows, L)
        Mat(ds, rows, L, L, dt).set(P * (d - (d * P).sum(1, keepdims=True)))

    def ddpm_dropout_mask(self, mask, n, p, seed, st):
        f32(mask, n)[...] = _keep_mask(seed, np.arange(n, dtype=np.uint64), _thresh(p)).astype(np.float32)


    def ddpm_atb_f32(self, a, lda, b, ldb, c, ldc, M, N, K, st):
        A = np.lib.stride_tricks.as_strided(f32(a, (K - 1) * lda + M), (K, M), (lda * 4, 4))
        Bm = np.lib.stride_tricks.as_strided(f32(b, (K - 1) * ldb + N), (K, N), (ldb * 4, 4))
        C = np.lib.stride_tricks.as_strided(f32(c, (M - 1) * ldc + N), (M, N), (ldc * 4, 4), writeable=True)
        C[...] = (A.astype(np.float64).T @ Bm.astype(np.float64)).astype(np.float32)

    # ------------------------------------------------------------------ plan helpers (csrc/plan.hip)
    def ddpm_stream_order(self, waiter, signaller):
        pass                                             # host memory: everything is already in program order

    def ddpm_fill_zero(self, p, nbytes, st):
        if nbytes:
            ctypes.memset(p, 0, nbytes)


def install(monkeypatch, hip_module):
    """Route the product's ABI calls to the emulator for the duration of a test."""
    real_lib = hip_module.lib()                      # the real .so still answers host-side geometry queries
    emu = Emulator(real_lib)
    monkeypatch.setattr(hip_module, "_invoke", lambda name, args: emu.call(name, *args))
    monkeypatch.setattr(hip_module, "stream", lambda: 0)
    monkeypatch.setattr(hip_module, "require_cuda", lambda *a: None)
    monkeypatch.setattr(hip_module, "on_device", lambda t: True)
    return emu
