"""Oracle (test infrastructure): functional fp32/fp64 CPU restatement of the reference UNet.

The network is evaluated as a pure function of a ``state_dict`` (reference key
layout) so the same tensors can be fed to the HIP engine
and to this checker.  Reference citations are ``file:line`` in the upstream repo.

Not product code: see ``oracle/__init__.py`` for who may import this.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

GN_GROUPS = 32      # ddpm_torch/models/unet.py:19
GN_EPS = 1e-6       # ddpm_torch/models/unet.py:20


def normalize_cfg(cfg):
    """Fill constructor defaults the way UNet.__init__ does (unet.py:96-121)."""
    c = dict(cfg)
    levels = len(c["ch_multipliers"])
    attn = c["apply_attn"]
    if isinstance(attn, bool):
        attn = [attn] * levels                      # unet.py:115-116
    c["apply_attn"] = list(attn)
    c.setdefault("out_channels", c["in_channels"])
    c["time_embedding_dim"] = c.get("time_embedding_dim") or 4 * c["hid_channels"]  # unet.py:112
    c.setdefault("drop_rate", 0.0)
    c.setdefault("resample_with_conv", True)             # False: AvgPool2d(2) / bare nearest Upsample, no conv parameters (unet.py:163-170,196-199)
    return c


# --------------------------------------------------------------------------- parameter layout

def _res_keys(prefix, cin, cout, edim):
    """ResidualBlock parameter order (unet.py:75-81)."""
    ks = [
        (prefix + "norm1.weight", (cin,), "ones"), (prefix + "norm1.bias", (cin,), "zeros"),
        (prefix + "conv1.weight", (cout, cin, 3, 3), 1.0), (prefix + "conv1.bias", (cout,), "zeros"),
        (prefix + "fc.weight", (cout, edim), 1.0), (prefix + "fc.bias", (cout,), "zeros"),
        (prefix + "norm2.weight", (cout,), "ones"), (prefix + "norm2.bias", (cout,), "zeros"),
        (prefix + "conv2.weight", (cout, cout, 3, 3), 0.0), (prefix + "conv2.bias", (cout,), "zeros"),
    ]
    if cin != cout:
        ks += [(prefix + "skip.weight", (cout, cin, 1, 1), 1.0), (prefix + "skip.bias", (cout,), "zeros")]
    return ks


def _attn_keys(prefix, c):
    """AttentionBlock parameter order (unet.py:35-41)."""
    return [
        (prefix + "norm.weight", (c,), "ones"), (prefix + "norm.bias", (c,), "zeros"),
        (prefix + "project_in.weight", (3 * c, c, 1, 1), 1.0), (prefix + "project_in.bias", (3 * c,), "zeros"),
        (prefix + "project_out.weight", (c, c, 1, 1), 0.0), (prefix + "project_out.bias", (c,), "zeros"),
    ]


def _block_keys(prefix, cin, cout, edim, attn):
    """A level block is Res, or Sequential[Res, Attn] on attention levels (unet.py:144-154)."""
    if attn:
        return _res_keys(prefix + "0.", cin, cout, edim) + _attn_keys(prefix + "1.", cout)
    return _res_keys(prefix, cin, cout, edim)


def param_spec(cfg):
    """Ordered (key, shape, init) list in module-construction order (unet.py:122-142).

    ``init`` is "ones"/"zeros" or the variance-scaling ``init_scale`` (modules.py:11-18).
    """
    c = normalize_cfg(cfg)
    hid, edim, n = c["hid_channels"], c["time_embedding_dim"], c["num_res_blocks"]
    chs = [hid * m for m in c["ch_multipliers"]]
    L = len(chs)
    spec = [
        ("embed.0.weight", (edim, hid), 1.0), ("embed.0.bias", (edim,), "zeros"),
        ("embed.2.weight", (edim, edim), 1.0), ("embed.2.bias", (edim,), "zeros"),
        ("in_conv.weight", (hid, c["in_channels"], 3, 3), 1.0), ("in_conv.bias", (hid,), "zeros"),
    ]
    for i in range(L):                                             # unet.py:156-171
        prev = chs[i - 1] if i else hid
        p = f"downsamples.level_{i}."
        spec += _block_keys(p + "0.", prev, chs[i], edim, c["apply_attn"][i])
        for j in range(1, n):
            spec += _block_keys(p + f"{j}.", chs[i], chs[i], edim, c["apply_attn"][i])
        if i != L - 1 and c["resample_with_conv"]:
            spec += [(p + f"{n}.1.weight", (chs[i], chs[i], 3, 3), 1.0), (p + f"{n}.1.bias", (chs[i],), "zeros")]
    mid = chs[-1]
    spec += _res_keys("middle.0.", mid, mid, edim) + _attn_keys("middle.1.", mid) + _res_keys("middle.2.", mid, mid, edim)
    for i in range(L):                                             # unet.py:173-203 (ModuleDict built for i ascending)
        nxt = hid if i == 0 else chs[i - 1]
        prev = chs[-1] if i == L - 1 else chs[i + 1]
        p = f"upsamples.level_{i}."
        spec += _block_keys(p + "0.", prev + chs[i], chs[i], edim, c["apply_attn"][i])
        for j in range(1, n):
            spec += _block_keys(p + f"{j}.", 2 * chs[i], chs[i], edim, c["apply_attn"][i])
        spec += _block_keys(p + f"{n}.", nxt + chs[i], chs[i], edim, c["apply_attn"][i])
        if i != 0 and c["resample_with_conv"]:
            spec += [(p + f"{n + 1}.1.weight", (chs[i], chs[i], 3, 3), 1.0), (p + f"{n + 1}.1.bias", (chs[i],), "zeros")]
    spec += [
        ("out_conv.0.weight", (hid,), "ones"), ("out_conv.0.bias", (hid,), "zeros"),
        ("out_conv.2.weight", (c["out_channels"], hid, 3, 3), 0.0), ("out_conv.2.bias", (c["out_channels"],), "zeros"),
    ]
    return spec


def init_state_dict(cfg, dtype=torch.float32):
    """Reference initialisation under the current torch RNG state (modules.py:11-18,53-56,100-103).

    Draw order == construction order, so ``torch.manual_seed(s); init_state_dict(cfg)`` reproduces
    ``torch.manual_seed(s); UNet(**cfg).state_dict()`` bit for bit.
    """
    sd = OrderedDict()
    for key, shape, init in param_spec(cfg):
        w = torch.empty(shape, dtype=torch.float32)
        if init == "ones":
            w.fill_(1.0)
        elif init == "zeros":
            w.zero_()
        else:
            torch.nn.init.xavier_uniform_(w, gain=math.sqrt(init or 1e-10))   # modules.py:18
        sd[key] = w.to(dtype)
    return sd


def randomize_state_dict(sd, seed, scale_zero_init=True):
    """Re-randomise the ~zero-initialised layers and perturb biases / GN affine
    so parity checks are not vacuous.  Deterministic in ``seed``."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, v in sd.items():
        v = v.clone()
        if v.ndim >= 2:
            if scale_zero_init and float(v.abs().max()) < 1e-3:
                fan = v[0].numel() + v.shape[0] * (v[0].numel() // v.shape[1])
                a = math.sqrt(6.0 / fan)
                v = (torch.rand(v.shape, generator=g, dtype=torch.float64) * 2 - 1).mul(a).to(v.dtype)
        elif k.endswith("weight"):      # GN gamma
            v = v + 0.2 * (torch.rand(v.shape, generator=g, dtype=torch.float64) - 0.5).to(v.dtype)
        else:                            # biases, GN beta
            v = v + 0.1 * (torch.rand(v.shape, generator=g, dtype=torch.float64) - 0.5).to(v.dtype)
        out[k] = v
    return out


# --------------------------------------------------------------------------- ops

def timestep_embedding(t, dim, dtype=torch.float32):
    """functions.py:10-26 — sin/cos of t * exp(-i ln(1e4)/(half-1)); zero-pad odd dims."""
    half = dim // 2
    rate = math.log(10000) / (half - 1)
    freqs = torch.exp(-torch.arange(half, dtype=dtype) * rate)
    arg = torch.outer(t.reshape(-1).to(dtype), freqs)
    emb = torch.cat([arg.sin(), arg.cos()], dim=1)
    if dim % 2:
        emb = F.pad(emb, [0, 1])
    return emb


def group_norm(x, w, b):
    """unet.py:18-20 — GroupNorm(32, C, eps=1e-6), biased variance."""
    return F.group_norm(x, GN_GROUPS, w, b, eps=GN_EPS)


# ---- storage-rounding hooks (error-budget studies only: scripts/bf16_error_budget.py).  The MI355X engine's bf16 mode keeps fp32
# accumulators and rounds a tensor when it is STORED: "conv" = a convolution's output after its fused epilogue (bias, time bias,
# residual), "gn" = a GroupNorm(+SiLU) output, "attn" = the attention core's output and its probability tile, "input" = the image.
# None (the default) leaves the restatement exactly as the reference computes it.
ROUND = None


def _st(kind, x):
    return x if ROUND is None or kind not in ROUND else ROUND[kind](x)


def same_pad_s2(x):
    """modules.py:145-160 for kernel 3, stride 2 (TF SAME)."""
    h, w = x.shape[-2:]
    hp = 2 * math.ceil(h / 2 - 1) + 3 - h
    wp = 2 * math.ceil(w / 2 - 1) + 3 - w
    top, bot = (hp // 2, hp - hp // 2) if hp else (0, 0)
    lef, rig = (wp // 2, wp - wp // 2) if wp else (0, 0)
    return F.pad(x, (lef, rig, top, bot))


def attention_core(q, k, v):
    """unet.py:43-51 — single head, d = C, softmax over the H*W keys, scale 1/sqrt(C)."""
    B, C, H, W = q.shape
    qf, kf, vf = (z.reshape(B, C, H * W) for z in (q, k, v))
    logits = torch.einsum("bcq,bck->bqk", qf, kf) / math.sqrt(C)
    p = _st("attn", torch.softmax(logits, dim=-1))
    out = torch.einsum("bqk,bck->bcq", p, vf)
    return _st("attn", out.reshape(B, C, H, W))


def attention_block(sd, p, x):
    """unet.py:53-60 (skip is Identity: in == out everywhere in the UNet)."""
    h = _st("gn", group_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"]))
    qkv = _st("conv", F.conv2d(h, sd[p + "project_in.weight"], sd[p + "project_in.bias"]))
    q, k, v = qkv.chunk(3, dim=1)
    h = attention_core(q, k, v)
    h = F.conv2d(h, sd[p + "project_out.weight"], sd[p + "project_out.bias"])
    return _st("conv", h + x)


def residual_block(sd, p, x, t_emb, drop_p=0.0, training=False, mask=None):
    """unet.py:83-89.  ``mask`` (0/1, shape of the conv2 input) overrides torch's dropout RNG."""
    skip = x
    if p + "skip.weight" in sd:
        skip = _st("conv", F.conv2d(x, sd[p + "skip.weight"], sd[p + "skip.bias"]))
    h = F.conv2d(_st("gn", F.silu(group_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]))),
                 sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = _st("conv", h + F.linear(F.silu(t_emb), sd[p + "fc.weight"], sd[p + "fc.bias"])[:, :, None, None])
    h = _st("gn", F.silu(group_norm(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])))
    if training and drop_p > 0:
        if mask is not None:
            h = h * mask.to(h.dtype) / (1.0 - drop_p)
        else:
            h = F.dropout(h, drop_p, training=True)
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return _st("conv", h + skip)


def _block(sd, p, x, t_emb, attn, **kw):
    if attn:
        x = residual_block(sd, p + "0.", x, t_emb, mask=kw.pop("mask", None), **kw)
        return attention_block(sd, p + "1.", x)
    return residual_block(sd, p, x, t_emb, **kw)


def res_block_prefixes(cfg):
    """Prefixes of every ResidualBlock in forward-execution order (used to key dropout masks)."""
    c = normalize_cfg(cfg)
    n, L = c["num_res_blocks"], len(c["ch_multipliers"])
    out = []
    for i in range(L):
        for j in range(n):
            out.append(f"downsamples.level_{i}.{j}." + ("0." if c["apply_attn"][i] else ""))
    out += ["middle.0.", "middle.2."]
    for i in range(L - 1, -1, -1):
        for j in range(n + 1):
            out.append(f"upsamples.level_{i}.{j}." + ("0." if c["apply_attn"][i] else ""))
    return out


def unet_forward(sd, cfg, x, t, training=False, masks=None):
    """unet.py:205-233.  ``masks``: optional {res-block prefix: keep-mask} for dropout parity."""
    c = normalize_cfg(cfg)
    n, L = c["num_res_blocks"], len(c["ch_multipliers"])
    kw = dict(drop_p=c["drop_rate"], training=training)
    masks = masks or {}

    def mk(prefix, attn):
        return masks.get(prefix + ("0." if attn else ""))

    t_emb = timestep_embedding(t, c["hid_channels"]).to(x.dtype)
    t_emb = F.linear(t_emb, sd["embed.0.weight"], sd["embed.0.bias"])
    t_emb = F.linear(F.silu(t_emb), sd["embed.2.weight"], sd["embed.2.bias"])

    hs = [_st("conv", F.conv2d(_st("input", x), sd["in_conv.weight"], sd["in_conv.bias"], padding=1))]
    for i in range(L):
        p = f"downsamples.level_{i}."
        a = c["apply_attn"][i]
        for j in range(n):
            hs.append(_block(sd, p + f"{j}.", hs[-1], t_emb, a, mask=mk(p + f"{j}.", a), **kw))
        if i != L - 1:
            if c["resample_with_conv"]:
                hs.append(_st("conv", F.conv2d(same_pad_s2(hs[-1]), sd[p + f"{n}.1.weight"], sd[p + f"{n}.1.bias"], stride=2)))
            else:
                hs.append(F.avg_pool2d(hs[-1], 2))                            # unet.py:169 nn.AvgPool2d(2)

    h = residual_block(sd, "middle.0.", hs[-1], t_emb, mask=masks.get("middle.0."), **kw)
    h = attention_block(sd, "middle.1.", h)
    h = residual_block(sd, "middle.2.", h, t_emb, mask=masks.get("middle.2."), **kw)

    for i in range(L - 1, -1, -1):
        p = f"upsamples.level_{i}."
        a = c["apply_attn"][i]
        for j in range(n + 1):
            h = _block(sd, p + f"{j}.", torch.cat([h, hs.pop()], dim=1), t_emb, a, mask=mk(p + f"{j}.", a), **kw)
        if i != 0:
            h = F.interpolate(h, scale_factor=2, mode="nearest")              # unet.py:199
            if c["resample_with_conv"]:
                h = _st("conv", F.conv2d(h, sd[p + f"{n + 1}.1.weight"], sd[p + f"{n + 1}.1.bias"], padding=1))
    assert not hs
    h = _st("gn", F.silu(group_norm(h, sd["out_conv.0.weight"], sd["out_conv.0.bias"])))
    return F.conv2d(h, sd["out_conv.2.weight"], sd["out_conv.2.bias"], padding=1)


def forward_flops_per_sample(cfg, H, W):
    """Algorithmic forward FLOPs (2*MAC over convs, linears, attention matmuls)."""
    c = normalize_cfg(cfg)
    spec = {k: s for k, s, _ in param_spec(c)}
    n, L = c["num_res_blocks"], len(c["ch_multipliers"])
    total = 0

    def conv(key, hw):
        nonlocal total
        co, ci, kh, kw_ = spec[key]
        total += 2 * hw * co * ci * kh * kw_

    def lin(key):
        nonlocal total
        o, i = spec[key]
        total += 2 * o * i

    def res(p, hw):
        conv(p + "conv1.weight", hw); conv(p + "conv2.weight", hw); lin(p + "fc.weight")
        if p + "skip.weight" in spec:
            conv(p + "skip.weight", hw)

    def attn(p, hw):
        nonlocal total
        conv(p + "project_in.weight", hw); conv(p + "project_out.weight", hw)
        total += 4 * hw * hw * spec[p + "norm.weight"][0]

    def block(p, hw, a):
        if a:
            res(p + "0.", hw); attn(p + "1.", hw)
        else:
            res(p, hw)

    lin("embed.0.weight"); lin("embed.2.weight")
    h, w = H, W
    conv("in_conv.weight", h * w)
    for i in range(L):
        for j in range(n):
            block(f"downsamples.level_{i}.{j}.", h * w, c["apply_attn"][i])
        if i != L - 1:
            if c["resample_with_conv"]:
                h, w = (h + 1) // 2, (w + 1) // 2
                conv(f"downsamples.level_{i}.{n}.1.weight", h * w)
            else:
                h, w = h // 2, w // 2
    res("middle.0.", h * w); attn("middle.1.", h * w); res("middle.2.", h * w)
    sizes = [(H, W)]
    for _ in range(L - 1):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2) if c["resample_with_conv"] else (sizes[-1][0] // 2, sizes[-1][1] // 2))
    for i in range(L - 1, -1, -1):
        h, w = sizes[i]
        for j in range(n + 1):
            block(f"upsamples.level_{i}.{j}.", h * w, c["apply_attn"][i])
        if i != 0 and c["resample_with_conv"]:
            h2, w2 = sizes[i - 1]
            conv(f"upsamples.level_{i}.{n + 1}.1.weight", h2 * w2)
    conv("out_conv.2.weight", H * W)
    return total
