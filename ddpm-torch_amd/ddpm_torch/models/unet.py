"""MI355X-native UNet denoiser with the reference's constructor, ``forward(x, t)`` and state-dict layout.

Interface parity: ``ddpm_torch/models/unet.py:92-233`` of tqch/ddpm-torch (class ``UNet``; blocks at
``:23-60`` AttentionBlock and ``:63-89`` ResidualBlock).  The module tree below exists to own the
``nn.Parameter`` s under the reference's key names and to reproduce its
initialisation draw order; the arithmetic does NOT run through ``nn.Module.forward`` of the children.
``UNet.forward`` hands the whole network to ``_Engine``, which launches hand-written gfx950 kernels
through the C-ABI in ``csrc/`` (see ``include/ddpm_hip.h``):

* activations are NHWC (bf16 or fp32), every kernel takes an explicit pixel pitch, so the decoder's
  ``torch.cat([h, skip])`` is zero-copy — producers write straight into channel slices of the consumer's buffer;
* 3x3 / 1x1 convolutions, Linear layers and the attention matmuls are one MFMA implicit-GEMM kernel; SAME-pad
  stride-2, nearest-2x upsample, bias, per-sample time bias and the residual add are folded into its loader / epilogue;
* GroupNorm(32, eps=1e-6)+SiLU(+dropout) is one fused pair of launches; the 22 per-block time-bias Linears
  are ONE GEMM over concatenated weights and ``SiLU(t_emb)`` is computed once (the reference recomputes it 22x);
* training runs a hand-written backward (dgrad = the same kernel on flipped weights, wgrad = transposed-operand
  GEMM with fp32 atomics into the gradient) behind a single ``torch.autograd.Function``.

There is no CPU path: CPU tensors raise (the CPU restatement used for checking lives in ``oracle/``).
"""
import contextlib
import itertools
import math
import os

import torch
import torch.nn as nn

from .. import _hip
from .. import _ops as ops
from .._ops import View

__all__ = ["UNet", "ResidualBlock", "AttentionBlock"]

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def _vs_init_(w, scale):
    """TF variance-scaling(fan_avg, uniform) == xavier_uniform with gain sqrt(scale) (modules.py:11-18)."""
    return nn.init.xavier_uniform_(w, gain=math.sqrt(scale or 1e-10))


class _Conv(nn.Module):
    """Parameter holder for a Conv2d site (modules.py:66-123): weight [Cout, Cin, k, k] fp32, bias zeros."""

    def __init__(self, cin, cout, k, init_scale=1.0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        _vs_init_(self.weight, init_scale)
        nn.init.zeros_(self.bias)


class _Linear(nn.Module):
    """Parameter holder for a Linear site (modules.py:34-63)."""

    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        _vs_init_(self.weight, 1.0)
        nn.init.zeros_(self.bias)


class _Norm(nn.Module):
    """GroupNorm(32, C, eps=1e-6) affine parameters (unet.py:18-20)."""

    def __init__(self, c):
        super().__init__()
        if c % ops.GN_GROUPS:
            raise ValueError(f"GroupNorm(32) needs channels divisible by 32, got {c}")
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Slot(nn.Module):
    """Parameter-less position in a Sequential (SamePad2d / Upsample / SiLU in the reference)."""


class ResidualBlock(nn.Module):
    """unet.py:63-89 — norm1, conv1, fc, norm2, conv2 (init_scale 0), skip (1x1 when channels change)."""

    def __init__(self, in_channels, out_channels, embed_dim, drop_rate=0.0):
        super().__init__()
        self.in_channels, self.out_channels, self.drop_rate = in_channels, out_channels, drop_rate
        self.norm1 = _Norm(in_channels)
        self.conv1 = _Conv(in_channels, out_channels, 3)
        self.fc = _Linear(embed_dim, out_channels)
        self.norm2 = _Norm(out_channels)
        self.conv2 = _Conv(out_channels, out_channels, 3, init_scale=0.0)
        self.has_skip = in_channels != out_channels     # the reference holds nn.Identity otherwise: no keys either way
        if self.has_skip:
            self.skip = _Conv(in_channels, out_channels, 1)


class AttentionBlock(nn.Module):
    """unet.py:23-60 — norm, project_in (C -> 3C, 1x1), project_out (init_scale 0); single head, d = C."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = _Norm(in_channels)
        self.project_in = _Conv(in_channels, 3 * in_channels, 1)
        self.project_out = _Conv(in_channels, in_channels, 1, init_scale=0.0)


def _seq(*mods):
    return nn.Sequential(*mods)


class UNet(nn.Module):
    """Same constructor / forward contract as the reference (unet.py:96-107, :205)."""

    def __init__(self, in_channels, hid_channels, out_channels, ch_multipliers, num_res_blocks, apply_attn,
                 time_embedding_dim=None, drop_rate=0., resample_with_conv=True):
        super().__init__()
        self.in_channels, self.hid_channels, self.out_channels = in_channels, hid_channels, out_channels
        self.time_embedding_dim = time_embedding_dim or 4 * hid_channels
        self.levels = levels = len(ch_multipliers)
        self.ch_multipliers = ch_multipliers
        if isinstance(apply_attn, bool):
            apply_attn = [apply_attn] * levels
        self.apply_attn = apply_attn
        self.num_res_blocks, self.drop_rate, self.resample_with_conv = num_res_blocks, drop_rate, resample_with_conv
        E, n = self.time_embedding_dim, num_res_blocks
        chs = [hid_channels * m for m in ch_multipliers]

        def block(level, cin, cout):
            res = ResidualBlock(cin, cout, E, drop_rate)
            return _seq(res, AttentionBlock(cout)) if apply_attn[level] else res

        # construction order == the reference's, so seeded initialisation is bit-identical
        self.embed = _seq(_Linear(hid_channels, E), _Slot(), _Linear(E, E))
        self.in_conv = _Conv(in_channels, hid_channels, 3)
        self.downsamples = nn.ModuleDict()
        for i in range(levels):
            prev = chs[i - 1] if i else hid_channels
            mods = [block(i, prev, chs[i])] + [block(i, chs[i], chs[i]) for _ in range(n - 1)]
            if i != levels - 1:                   # unet.py:163-170: SamePad2d + stride-2 conv, or nn.AvgPool2d(2) (no parameters)
                mods.append(_seq(_Slot(), _Conv(chs[i], chs[i], 3)) if resample_with_conv else _Slot())
            self.downsamples[f"level_{i}"] = nn.ModuleList(mods)
        mid = chs[-1]
        self.middle = _seq(ResidualBlock(mid, mid, E, drop_rate), AttentionBlock(mid), ResidualBlock(mid, mid, E, drop_rate))
        self.upsamples = nn.ModuleDict()
        for i in range(levels):
            nxt = hid_channels if i == 0 else chs[i - 1]
            prev = chs[-1] if i == levels - 1 else chs[i + 1]
            mods = [block(i, prev + chs[i], chs[i])] + [block(i, 2 * chs[i], chs[i]) for _ in range(n - 1)]
            mods.append(block(i, nxt + chs[i], chs[i]))
            if i != 0:                            # unet.py:196-199: nearest-2x Upsample, followed by a 3x3 conv only when resample_with_conv
                mods.append(_seq(_Slot(), _Conv(chs[i], chs[i], 3)) if resample_with_conv else _seq(_Slot()))
            self.upsamples[f"level_{i}"] = nn.ModuleList(mods)
        self.out_conv = _seq(_Norm(hid_channels), _Slot(), _Conv(hid_channels, out_channels, 3, init_scale=0.0))

        self.compute_dtype = _DTYPES[os.environ.get("DDPM_TORCH_AMD_COMPUTE", "fp32").lower()]
        self._eng = None

    # ------------------------------------------------------------------ precision knob (not in the reference)
    def set_compute_dtype(self, dtype):
        """torch.float32 (default: exact-fp32 MFMA, meets the 1e-3 parity bar) or torch.bfloat16 (throughput)."""
        if isinstance(dtype, str):
            dtype = _DTYPES[dtype.lower()]
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(dtype)
        self.compute_dtype = dtype
        self._eng = None
        return self

    def set_process_group(self, group="default", broadcast=True):
        """Native data parallelism (instead of wrapping the model in DistributedDataParallel): the hand-written backward
        all-reduces its gradient staging buffer over ``group`` in a few large chunks, issued from inside the backward as
        soon as each chunk is final (RCCL runs them on its own stream under the remaining backward kernels), and divides
        by the world size in the unpack — the result equals DDP's averaged gradients.  ``broadcast`` copies rank 0's
        parameters to every rank first, as DDP's constructor does.  Pass ``None`` to switch it off."""
        import torch.distributed as dist
        self._pg = None if group is None else (dist.group.WORLD if group == "default" else group)
        if self._pg is not None and broadcast:
            with torch.no_grad():
                for p in self.parameters():
                    dist.broadcast(p.detach(), src=dist.get_global_rank(self._pg, 0), group=self._pg)   # detach() shares the version counter, .data does not
        if self._eng is not None:
            self._eng.pg = self._pg
            self._eng.apply_reserved_cus()
        return self

    def _apply(self, fn, *a, **k):
        self._eng = None                       # .to()/.cuda() re-create parameter storage: drop pointer caches
        return super()._apply(fn, *a, **k)

    def engine(self):
        if self._eng is None:
            self._eng = _Engine(self)
            self._eng.pg = getattr(self, "_pg", None)
            self._eng.apply_reserved_cus()
        return self._eng

    def forward(self, x, t):
        _hip.require_cuda(x, t)
        eng = self.engine()
        params = eng.params
        # (x.requires_grad: the reference returns d/dx through autograd, unet.py:205-233; here the backward's last step — in_conv's data
        #  gradient, otherwise skipped: training never needs it, x_t is data — is run when the autograd graph asks for it)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _UNetFn.apply(x, t, eng, self.training, *params)
        return eng.forward(x, t, self.training, None)


class _UNetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward runs the engine with a tape, backward replays it."""

    @staticmethod
    def forward(ctx, x, t, eng, training, *params):
        tape = []
        out = eng.forward(x, t, training, tape)
        ctx.eng, ctx.tape = eng, tape
        eng.last_tape = tape if eng.debug_keep_tape else None     # introspection hook for tests (dropout seeds, saved activations)
        return out

    @staticmethod
    def backward(ctx, gout):
        want_dx = ctx.needs_input_grad[0]
        grads = ctx.eng.backward(ctx.tape, gout, want_dx=want_dx)
        dx = ctx.eng.last_dx if want_dx else None
        ctx.eng.last_dx = None
        ctx.tape = None
        return (dx, None, None, None) + tuple(grads)


# ====================================================================================================== engine

class _ConvW:
    """Packed device copies of one conv weight, refreshed when the master parameter changes."""
    __slots__ = ("mod", "N", "C", "R", "Cp", "Np", "wf", "wd", "ver", "src", "up")

    def __init__(self, mod, vec):
        self.mod = mod
        self.N, self.C, self.R, _ = mod.weight.shape
        self.Cp = -(-self.C // vec) * vec
        self.Np = -(-self.N // vec) * vec
        self.wf = self.wd = None
        self.ver = self.src = None
        self.up = False                # conv behind a nearest-2x upsample: its dgrad pack is the 4x4 / stride-2 effective kernel


_WGRAD_TARGET_BLOCKS = int(os.environ.get("DDPM_WGRAD_BLOCKS", "512"))
_FUSED_ATTENTION = os.environ.get("DDPM_FUSED_ATTENTION", "1") != "0"
_FLASH_ATTENTION = os.environ.get("DDPM_FLASH_ATTENTION", "1") != "0"     # 0: the five-product path with L x L tensors (A/B only)
_FLASH_INFERENCE = os.environ.get("DDPM_FLASH_INFERENCE", "1") != "0"     # inference attention through the training forward kernel (no lse stored)
_UP_DGRAD_FUSED = os.environ.get("DDPM_UP_DGRAD_FUSED", "1") != "0"    # upsample convs: dgrad as one 4x4 stride-2 conv
_SIDE_STREAM = os.environ.get("DDPM_SIDE_STREAM", "1") != "0"      # weight / bias gradients on a second HIP stream
_SIDE_PRIORITY = int(os.environ.get("DDPM_SIDE_PRIORITY", "0"))    # its priority (lower number = dispatched first; clamped to the device's range)
_WGRAD_MINSTEPS = int(os.environ.get("DDPM_WGRAD_MINSTEPS", "20"))
_WGRAD_SLABS = os.environ.get("DDPM_WGRAD_SLABS", "0") != "0"      # deterministic slab reduction instead of atomics
_WGRAD3 = os.environ.get("DDPM_WGRAD3", "1") != "0"                # patch-stationary kernel for the 3x3 / stride-1 weight gradients
_WGRAD3_UP = os.environ.get("DDPM_WGRAD3_UP", "1") != "0"          # ... also for the Upsample blocks' conv (nearest-2x gather folded into the halo loads)
_SLAB_FLUSH_ROWS = int(os.environ.get("DDPM_SLAB_FLUSH_ROWS", "3"))   # slab reductions queued on the side stream every so many rows
# The backward ends at the network's first layers, and the step ends when the side stream has drained: the weight gradients of the LAST
# residual blocks of the backward (the first in execution order) find the main stream with little left to run — they take the whole chip
# instead of the usual half (twice the K slices).  Same-box sweeps on two boxes (profiles/r05_tail_boost_sweep.txt): 0 / 4 / 8 blocks ->
# 9.87 / 9.81 / 9.83 ms per step (every one of nine 0-vs-4 pairs in favour of 4); 0 = off.
_TAIL_BLOCKS = int(os.environ.get("DDPM_WGRAD3_TAIL_BLOCKS", "4"))
_TAIL_ANY = os.environ.get("DDPM_WGRAD3_TAIL_ANY", "0") != "0"
# Data parallel: 1 = chunk exchanges ordered behind the side stream, no main-stream joins (-7.5 % on the one-rank RCCL step, round 5).
# Default 0 = the round-4 form (join + issue on the main stream): a one-rank all-reduce is the identity and gloo stages through the host, so
# the ordering of the fast form has never been observable on this one-GPU pool.  Opt in after tests/test_multi_gpu.py passed on >= 2 GPUs.
_DP_ISSUE_ON_SIDE = os.environ.get("DDPM_DP_ISSUE_ON_SIDE", "0") != "0"
_TEMB_LEAVES = os.environ.get("DDPM_TEMB_LEAVES", "0") != "0"          # A/B switch: the embedding MLP's parameter gradients as side-stream leaves (see _temb_bwd)
_ABL_NO_LEAF_ORDER = os.environ.get("DDPM_ABL_NO_LEAF_ORDER", "0") != "0"   # TIMING-ONLY ablation (wrong results): leaves are not ordered behind the main stream
_WGRAD1 = os.environ.get("DDPM_WGRAD1", "1") != "0"                # slab kernel for the 1x1 weight gradients
_WGRAD3_ATOMIC = os.environ.get("DDPM_WGRAD3_ATOMIC", "0") != "0"  # ... with fp32 atomics instead of slab copies


class _Engine:
    _serials = itertools.count(1)

    def __init__(self, model):
        self.m = model
        self.serial = next(_Engine._serials)       # identity for caches of captured graphs (an id() can be recycled, this cannot)
        self.params = list(model.parameters())
        self.names = [k for k, _ in model.named_parameters()]
        _hip.require_cuda(self.params[0])          # no CPU path: move the model to the GPU first
        self.device = self.params[0].device
        self.T = model.compute_dtype
        self.dcode = _hip.BF16 if self.T == torch.bfloat16 else _hip.F32
        self.es = 2 if self.T == torch.bfloat16 else 4
        self.vec = 16 // self.es
        self.goff, off = {}, 0                     # parameter -> offset in the flat fp32 gradient buffer
        for p in self.params:
            self.goff[id(p)] = off
            off += (p.numel() + 3) // 4 * 4        # keep every slice 16-byte aligned
        self.gtotal = off
        self.wdesc = None                          # device table for ddpm_wgrad_unpack (built on first backward)
        self._gpack, self._slabs, self._slab_tables, self._eff_splits, self._side = None, {}, {}, {}, None
        self.last_dx = None                                # d/d(input image) of the last backward that was asked for it (UNet.forward, x.requires_grad)
        self._works = []                           # outstanding all-reduce handles of the native data-parallel backward
        m = model
        self.hid, self.E, self.L, self.n = m.hid_channels, m.time_embedding_dim, m.levels, m.num_res_blocks
        self.chs = [m.hid_channels * k for k in m.ch_multipliers]
        # ordered residual blocks (execution order) -> slice of the concatenated time-bias GEMM
        self.res_blocks = []
        self.convs = {}
        self._collect()
        self.tb_off, o = {}, 0
        for rb in self.res_blocks:
            self.tb_off[id(rb)] = o
            o += rb.out_channels
        self.tb_total = o
        half = self.hid // 2
        rate = math.log(10000) / (half - 1)
        self.freqs = torch.exp(-torch.arange(half, dtype=torch.float32) * rate).to(self.device)   # functions.py:19-20
        self.fc_w = self.fc_b = self.fc_table = None       # concatenated time-bias projection (persistent: graph replays read it)
        self.fc_ver = None
        self.tt = None                                    # [T][sum Cout] time biases of every timestep (sampling only; see time_table)
        self.tt_T, self.tt_key, self.tt_idx, self.tt_on = 0, None, None, False
        self._ws = None
        self._ws_need, self._ws_retired = {}, []
        self.drop_calls = 0
        self.last_tape = None
        self.debug_keep_tape = False               # tests set it to look at the tape after a step; costs the activations' lifetime
        self.splitk = ops.SplitK(self.device)
        self.pg = None                              # process group of the native data-parallel path (set_process_group)
        self.dp_standin = None                      # callable(chunk) run where an all-reduce is issued (bench.py's one-rank stand-in measurement)
        self.dp_trace = None                        # list -> (what, bytes, event) records of one backward's exchange (bench.py's config.dp)
        self.pack_table = self.pack_ptrs = self.pack_key = None
        self.pack_has_dgrad = False
        _hip.lib()

    def apply_reserved_cus(self):
        """Data-parallel runs: leave DDPM_DP_RESERVED_CUS compute units to the communicator's kernels (the persistent kernels otherwise
        take every CU whole and an all-reduce issued inside the backward finds one only at a block boundary).  Process-wide switch of the
        library; the slab counts of the weight-gradient kernels depend on it, so the cached plans are dropped when it changes.
        Default 0: on one GPU a stand-in copy kernel issued where the all-reduces are gets its CUs within one block of the
        persistent kernels either way (profiles/r05_dp_one_rank.json: `reserved_cus` sweep); the right value for 8 ranks over xGMI is
        to be swept on the node."""
        want = int(os.environ.get("DDPM_DP_RESERVED_CUS", "0")) if self.pg is not None else 0
        lib = _hip.lib()
        if int(lib.ddpm_get_reserved_cus()) != want:
            if lib.ddpm_set_reserved_cus(want) != 0:
                raise ValueError(f"DDPM_DP_RESERVED_CUS={want}: must be in [0, 192]")
            self._eff_splits.clear()
            self._slabs.clear()
            self._slab_tables.clear()

    # ---------------------------------------------------------------- topology helpers
    def _split(self, blk):
        return (blk[0], blk[1]) if isinstance(blk, nn.Sequential) else (blk, None)

    def _collect(self):
        m = self.m

        def reg(conv):
            self.convs[id(conv)] = _ConvW(conv, self.vec)

        def regblock(blk):
            res, att = self._split(blk)
            self.res_blocks.append(res)
            reg(res.conv1); reg(res.conv2)
            if res.has_skip:
                reg(res.skip)
            if att is not None:
                reg(att.project_in); reg(att.project_out)

        reg(m.in_conv)
        for i in range(self.L):
            mods = m.downsamples[f"level_{i}"]
            for j in range(self.n):
                regblock(mods[j])
            if i != self.L - 1 and m.resample_with_conv:
                reg(mods[self.n][1])
        self.res_blocks.append(m.middle[0]); reg(m.middle[0].conv1); reg(m.middle[0].conv2)
        reg(m.middle[1].project_in); reg(m.middle[1].project_out)
        self.res_blocks.append(m.middle[2]); reg(m.middle[2].conv1); reg(m.middle[2].conv2)
        for i in range(self.L - 1, -1, -1):
            mods = m.upsamples[f"level_{i}"]
            for j in range(self.n + 1):
                regblock(mods[j])
            if i != 0 and m.resample_with_conv:
                reg(mods[self.n + 1][1])
                self.convs[id(mods[self.n + 1][1])].up = _UP_DGRAD_FUSED
        reg(m.out_conv[2])

    # ---------------------------------------------------------------- derived weight caches
    def _refresh_packs(self, need_dgrad):
        """Re-derive every conv's packed copies in ONE launch when any master weight changed (optimizer step,
        load_state_dict, EMA swap: all bump ``_version``) or the dgrad copies are needed for the first time."""
        key = self._pack_versions()
        ptrs = tuple(cw.mod.weight.data_ptr() for cw in self.convs.values())
        if self.pack_table is None or ptrs != self.pack_ptrs or (need_dgrad and not self.pack_has_dgrad):
            rows = []
            for cw in self.convs.values():
                if cw.wf is None:
                    cw.wf = torch.empty(cw.N * cw.R * cw.R * cw.Cp, dtype=self.T, device=self.device)
                if need_dgrad and cw.wd is None:
                    cw.wd = torch.empty(cw.C * (16 if cw.up else cw.R * cw.R) * cw.Np, dtype=self.T, device=self.device)
                rows.append([cw.mod.weight.data_ptr(), cw.wf.data_ptr(), _hip.ptr(cw.wd), cw.N, cw.C, cw.R | (0x100 if cw.up else 0), cw.Cp, cw.Np])
            self.pack_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self.pack_ptrs, self.pack_has_dgrad, self.pack_key = ptrs, self.pack_has_dgrad or need_dgrad, None
        if key != self.pack_key:
            self._launch_pack()
            self.pack_key = key

    def _pack_versions(self):
        return tuple(cw.mod.weight._version for cw in self.convs.values())

    def _launch_pack(self):
        _hip.call("ddpm_pack_weight_multi", self.pack_table.data_ptr(), self.pack_table.shape[0], self.dcode, _hip.stream())

    def _packed(self, conv, need_dgrad):
        return self.convs[id(conv)]

    def _fc_versions(self):
        return tuple((rb.fc.weight._version, rb.fc.bias._version, rb.conv1.bias._version, rb.fc.weight.data_ptr()) for rb in self.res_blocks)

    def _fc_all(self):
        """Concatenated time-bias projection: rows = all ResidualBlock.fc weights; bias = fc.bias + conv1.bias
        (both are added at the same place, unet.py:85-86), so conv1's epilogue adds one per-sample vector.  The two
        buffers are persistent and re-filled by ONE multi-tensor launch when a member parameter changed."""
        key = self._fc_versions()
        if self.fc_table is None or key[0][3] != self.fc_ptr0 or any(k[3] != q for k, q in zip(key, self.fc_ptrs)):
            E = self.E
            if self.fc_w is None:
                self.fc_w = self._f32(self.tb_total, E)
                self.fc_b = self._f32(self.tb_total)
            rows = []
            for rb in self.res_blocks:
                o, c = self.tb_off[id(rb)], rb.out_channels
                rows.append([rb.fc.weight.data_ptr(), 0, self.fc_w.data_ptr() + 4 * o * E, c * E])
                rows.append([rb.fc.bias.data_ptr(), rb.conv1.bias.data_ptr(), self.fc_b.data_ptr() + 4 * o, c])
            self.fc_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self.fc_ptrs = tuple(k[3] for k in key)
            self.fc_ptr0 = key[0][3]
            self.fc_ver = None
        if key != self.fc_ver:
            self._launch_fc()
            self.fc_ver = key
        return self.fc_w, self.fc_b

    def _launch_fc(self):
        _hip.call("ddpm_mt_gather_f32", self.fc_table.data_ptr(), self.fc_table.shape[0], _hip.stream())

    # ---- time biases of ALL timesteps (sampling).  Every ResidualBlock's fc(act(t_emb)) depends on t alone (unet.py:86 behind UNet.embed
    # :122-126,207), and all samples of a sampling step share one t: the sampler asks for the [T][sum Cout] table once per weight version
    # and each step gathers its rows — one launch instead of the embedding MLP's six (sinusoid, 3 fp32 GEMMs, 2 SiLUs: ~60 us of a
    # 3-ms step).  Only forwards issued by a sampler (tt_on, set around its loop / capture) read it: t is known to lie inside the schedule
    # there; training and plain eval calls run the MLP.
    def enable_time_table(self, T):
        """Called by the samplers (GaussianDiffusion / DDIM) with the length of the schedule the model is evaluated on."""
        if T > self.tt_T:
            self.tt_T, self.tt_key = int(T), None
            if self.tt is not None:
                self._ws_retired.append(self.tt)          # a captured step may still point at it
            with torch.inference_mode(False):             # (the samplers run under inference_mode: a tensor born there could not be refilled outside it)
                self.tt = self._f32(self.tt_T, self.tb_total)
                self.tt_idx = torch.arange(self.tt_T, dtype=torch.int64, device=self.device)

    def _embed_versions(self):
        m = self.m
        return tuple((p._version, p.data_ptr()) for p in (m.embed[0].weight, m.embed[0].bias, m.embed[2].weight, m.embed[2].bias))

    def _time_biases(self, t, B):
        """[B][sum Cout] time biases (+ conv1 biases) of timesteps ``t``, and what the backward of the path needs."""
        m, E = self.m, self.E
        temb = self._f32(B, self.hid)
        _hip.call("ddpm_timestep_embedding", t.data_ptr(), self.freqs.data_ptr(), temb.data_ptr(), B, self.hid, _hip.stream())
        e1 = self._linear(temb, m.embed[0].weight, m.embed[0].bias, B, E, self.hid)
        s1 = self._f32(B, E)
        _hip.call("ddpm_silu_fwd", e1.data_ptr(), s1.data_ptr(), B * E, _hip.stream())
        t_emb = self._linear(s1, m.embed[2].weight, m.embed[2].bias, B, E, E)
        s_t = self._f32(B, E)                                  # SiLU(t_emb): shared by every block (unet.py:86)
        _hip.call("ddpm_silu_fwd", t_emb.data_ptr(), s_t.data_ptr(), B * E, _hip.stream())
        fc_w, fc_b = self._fc_all()
        tb = self._linear(s_t, fc_w, fc_b, B, self.tb_total, E)   # [B, sum Cout]: every block's time bias (+conv1 bias)
        return tb, (temb, e1, s1, t_emb, s_t, fc_w)

    def time_table(self):
        """The table, rebuilt (in place: captured steps hold its address) when a parameter of the path changed.

        Built as ONE M = T = 1000 pass of the embedding MLP, not in row blocks of the sampler's batch: at M = 1000 the dispatcher
        picks the 128 x 128-tile GEMM where the per-step MLP at M = B picks gemm64, which splits K and adds the halves in another
        order.  A sampler reading the table therefore differs from DDPM_TIME_TABLE=0 (and from a plain ``model(x, t)`` eval call) in
        the last bits of each fp32 time bias.  The bars: table vs no table <= 2e-5 of the sample's range after a full chain
        (tests/test_configs_gpu.py::test_sampler_time_table_follows_weights_and_schedule_length); table-fed sampler vs the fp32
        reference 3.9e-6 max abs after the 1000-step config-2 chain (tests/test_config2_parity_gpu.py), inside the 1e-4 parity bar."""
        self._fc_all()
        key = (self.fc_ver, self._embed_versions())
        if key != self.tt_key:
            tb, _ = self._time_biases(self.tt_idx, self.tt_T)
            self.tt.copy_(tb)
            self.tt_key = key
        return self.tt

    # ---- derived-copy management for replayed hipGraphs (the kernels of a captured step read the packed copies through
    # fixed addresses; nothing inside a graph can look at version counters)
    def ensure_fresh(self, need_dgrad=False):
        """Eagerly bring every derived copy (packed conv weights, concatenated time-bias projection) up to date with the
        master parameters — what forward() does on entry; callers replaying a captured graph do it before the replay."""
        self._refresh_packs(need_dgrad)
        self._fc_all()
        if self.tt_on:                                   # only a sampler reads the [T][sum Cout] table: a training step after the per-epoch
            self.time_table()                            # sample grid must not rebuild it (1000-row embedding MLP) before every replay

    def refresh_unconditionally(self):
        """Launch the derivation kernels regardless of version counters (the tail of a captured training step: the
        parameters it has just updated are re-packed inside the same graph)."""
        self._launch_pack()
        self._launch_fc()

    def mark_fresh(self, bumped=False):
        """The derived copies match the current parameter versions (after a step whose tail re-derived them).  ``bumped``: the caller
        knows that every parameter's version counter went up by exactly one since ``ensure_fresh`` verified the keys (the fused
        update's ``increment_version``) — the keys are advanced arithmetically instead of re-read (~350 attribute reads, 0.25 ms)."""
        if bumped and self.pack_key is not None and self.fc_ver is not None:
            pack_key = tuple(v + 1 for v in self.pack_key)
            fc_ver = tuple((a + 1, b + 1, c + 1, ptr) for a, b, c, ptr in self.fc_ver)
            first, last = next(iter(self.convs.values())), self.res_blocks[-1]
            if first.mod.weight._version == pack_key[0] and last.fc.weight._version == fc_ver[-1][0]:     # spot check; else re-read everything
                self.pack_key, self.fc_ver = pack_key, fc_ver
                return
        self.pack_key = self._pack_versions()
        self.fc_ver = self._fc_versions()

    def _workspace(self, B, H, W):
        """GroupNorm scratch for this input geometry.  Grow-only: captured graphs (training step, sampler) hold its address, and a
        forward at another batch size in between (the per-epoch sample grid) must not free what they point at — a buffer that is
        outgrown is retired, never released."""
        key = (B, H, W)
        need = self._ws_need.get(key)
        if need is None:
            need = 0
            for i in range(self.L):                      # every GroupNorm width that occurs at level i
                hw = (H >> i) * (W >> i)
                below = self.chs[i + 1] if i + 1 < self.L else self.chs[-1]
                above = self.chs[i - 1] if i else self.hid
                for c in {self.chs[i], above, self.chs[i] + below, 2 * self.chs[i], self.chs[i] + above}:
                    need = max(need, ops.gn_workspace_floats(B, hw, c, self.dcode))
            self._ws_need[key] = need
        if self._ws is None or self._ws.numel() < need:
            self._ws_retired.append(self._ws)
            self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
        return self._ws

    # ---------------------------------------------------------------- small helpers
    def _new(self, B, H, W, C):
        return View.new(B, H, W, C, self.T, self.device)

    def _f32(self, *shape):
        return _hip.retain(torch.empty(shape, dtype=torch.float32, device=self.device))

    def _zeros(self, *shape):
        """fp32 zeros by a stream-ordered C-ABI fill (a launch-plan records it; torch.zeros would be invisible to the plan)."""
        t = self._f32(*shape)
        _hip.call("ddpm_fill_zero", t.data_ptr(), t.numel() * 4, _hip.stream())
        return t

    def _gptr(self, gflat, p):
        return gflat.data_ptr() + 4 * self.goff[id(p)]

    def _wgrad_table(self):
        """Layout of the packed gradient scratch ``gpack`` and the descriptor table that ddpm_wgrad_unpack scatters from
        it into the flat parameter-order gradient: conv weights ([N][R*S][C] -> [N][C][R*S]) plus plain segment copies
        (R*S = 1) for everything that is produced once and owed to several parameters: the concatenated fc weight / bias
        gradients (fc.bias and conv1.bias share one vector) and the bias shared by conv2 and the 1x1 skip."""
        if self.wdesc is None:
            self.poff, rows, off = {}, [], 0

            def slot(key, n):
                nonlocal off
                self.poff[key] = off
                off += (n + 3) // 4 * 4
                return self.poff[key]

            # ---- conv weights in execution order, cut into the all-reduce chunks of the native data-parallel path as they are laid out:
            # a chunk = a run of conv weights + the rows of the time-projection gradient d(fc.weight) of the residual blocks whose conv1
            # lies in it (contiguous columns of the concatenated projection: ONE product per chunk).  Chunks become final one after another
            # as the backward walks the network in reverse; the chunk holding the network's FIRST layers is final LAST (the backward ends at
            # in_conv) and whatever travels after the backward has ended is exposed, so the first two chunks in execution order are capped
            # at 1/16 and 1/4 of the target (round 5; uniform chunks put 23 MB of the CIFAR net and 80 MB of the CelebA-HQ net — deep-
            # encoder weights that had been final for milliseconds — behind in_conv's gradient: profiles/r05_dp_one_rank.json), and the
            # 10-18 MB of d(fc.weight), which round 4 produced in one piece after the backward, travel with the chunks.
            # ~6 chunks of ~24 MB for the CIFAR / CelebA nets (143 MB of gradients); DDPM_DP_CHUNK_MB overrides the chunk size — xGMI rings
            # are per-link bound, so the right size is a property of the node and is to be swept there (bench.py prints config.dp);
            # DDPM_DP_CHUNK_RAMP=0: uniform chunks.
            E = self.E
            convs = list(self.convs.values())
            conv_floats = sum((cw.mod.weight.numel() + 3) // 4 * 4 for cw in convs) + self.tb_total * E
            target = max(conv_floats // 6, 1 << 20)
            if os.environ.get("DDPM_DP_CHUNK_MB"):
                target = max(int(float(os.environ["DDPM_DP_CHUNK_MB"]) * (1 << 18)), 1 << 16)
            ramp = [16, 4] if os.environ.get("DDPM_DP_CHUNK_RAMP", "1") != "0" else []
            rb_of_conv1 = {id(rb.conv1): rb for rb in self.res_blocks}
            self.chunks, self.chunk_fc, start, members, blocks = [], [], 0, [], []
            for i, cw in enumerate(convs):
                w = cw.mod.weight
                rows.append([slot(id(w), w.numel()), self.goff[id(w)], cw.N, cw.C, cw.R * cw.R])
                members.append(id(w))
                if id(cw.mod) in rb_of_conv1:
                    blocks.append(rb_of_conv1[id(cw.mod)])
                k = len(self.chunks)
                limit = max(target // ramp[k], 1 << 16) if k < len(ramp) else target
                pending_fc = sum(rb.out_channels for rb in blocks) * E
                if off + pending_fc - start >= limit or i + 1 == len(convs):
                    fc = None
                    if blocks:                                   # rows [o0, o0 + n) of the concatenated projection, stored at fbase
                        o0, n = self.tb_off[id(blocks[0])], sum(rb.out_channels for rb in blocks)
                        assert all(self.tb_off[id(rb)] == o0 + sum(q.out_channels for q in blocks[:j]) for j, rb in enumerate(blocks))
                        fbase = slot(("fc_w", k), n * E)
                        for rb in blocks:
                            rows.append([fbase + (self.tb_off[id(rb)] - o0) * E, self.goff[id(rb.fc.weight)], 1, rb.out_channels * E, 1])
                        fc = (o0, n, fbase)
                        members.append(("fc_w", k))
                    self.chunks.append((start, off, members))
                    self.chunk_fc.append(fc)
                    start, members, blocks = off, [], []
            conv_end = off
            fb = slot("fc_b", self.tb_total)
            for rb in self.res_blocks:
                o, c = self.tb_off[id(rb)], rb.out_channels
                rows.append([fb + o, self.goff[id(rb.fc.bias)], 1, c, 1])
                rows.append([fb + o, self.goff[id(rb.conv1.bias)], 1, c, 1])
                if rb.has_skip:
                    t = slot(("b2", id(rb)), c)
                    rows.append([t, self.goff[id(rb.conv2.bias)], 1, c, 1])
                    rows.append([t, self.goff[id(rb.skip.bias)], 1, c, 1])
            oc = self.m.out_conv[2]
            rows.append([slot("out_b", self.convs[id(oc)].Np), self.goff[id(oc.bias)], 1, self.m.out_channels, 1])
            # every remaining parameter (GroupNorm affine, biases with a single owner, the embed MLP) gets a plain slot:
            # the staging buffer is then the ONLY thing gradients are written to (and the only thing all-reduced).
            covered = {r[1] for r in rows}
            for prm in self.params:
                if self.goff[id(prm)] not in covered:
                    rows.append([slot(id(prm), prm.numel()), self.goff[id(prm)], 1, prm.numel(), 1])
            self.ptotal = off
            self.wdesc = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self.tail = (conv_end, self.ptotal)
        return self.wdesc

    @contextlib.contextmanager
    def _leaf(self, ctx, *views):
        """Weight / bias gradients are leaves of the backward graph: nothing downstream waits for them.  On the GPU they are
        enqueued on a SIDE stream (ordered after the producer of their inputs by an event), so the MFMA-bound wgrad kernels
        run next to the HBM-bound GroupNorm / softmax / element-wise kernels of the critical path instead of between them.
        The inputs are kept alive until the backward ends (the caching allocator must not hand them out while the side
        stream still reads them)."""
        side = ctx.get("side")
        if side is None:
            yield
            return
        ctx["keep"].extend(views)
        if not _ABL_NO_LEAF_ORDER:
            _hip.call("ddpm_stream_order", ctx["side_handle"], ctx["main_handle"])     # inputs are final on the main stream here
        # The kernels take their stream as an argument: routing _hip.stream() to the side stream's handle is all a leaf needs (no
        # torch.cuda.stream() context: two stream switches and several device look-ups per leaf, ~190 leaves per step).  Nothing
        # inside a leaf allocates temporaries (the slab copies are persistent), so the allocator's stream bookkeeping is not involved.
        prev = _hip.route_stream(ctx["side_handle"])
        try:
            yield
        finally:
            _hip.route_stream(prev)

    def join_forked_streams(self):
        """The CURRENT stream waits for the weight-gradient stream if that one is part of a stream capture (a capture being abandoned
        half-way: see SegmentedGraph.capture)."""
        side = self._side
        if side is None:
            return
        with torch.cuda.stream(side):
            forked = torch.cuda.is_current_stream_capturing()
        if forked:
            _hip._invoke("ddpm_stream_order", (_hip.stream(), side.cuda_stream))

    def _join_side(self, ctx):
        """Main stream waits for everything queued on the side stream so far."""
        if ctx.get("side") is not None:
            _hip.call("ddpm_stream_order", ctx["main_handle"], ctx["side_handle"])

    def _wgrad(self, ctx, weight, dy, x, Creal, Nreal, R, S, splits=1, bias=None, **kw):
        """Weight gradient into the staging buffer; then hand finished all-reduce chunks to the communicator.  ``bias`` = staging
        key of the bias gradient when it is the plain column sum of the same ``dy``: returns True when this call produced it too.

        3x3 / stride 1 / pad 1, bf16 (the bulk of the FLOPs): the patch-stationary kernel (csrc/wgrad.hip).  Its small output
        tiles need few reduction slices; every slice STORES its partial into its own slab copy (persistent per-weight workspace)
        and one multi-tensor launch sums the copies in a fixed order — bit-deterministic gradients (DDPM_WGRAD3_ATOMIC=1: fp32
        atomics instead).  It also accumulates the bias gradient from the dy fragments it reads.

        Everything else: the transposed-operand GEMM; its split-K slices add into the staging buffer with fp32 atomics, or
        (DDPM_WGRAD_SLABS=1) store slab copies as above."""
        did_bias = False
        patch = 0
        boost_req = kw.pop("boost", False)
        up = bool(kw.get("upsample"))                    # Upsample block: x stored at half of dy's size, gathered in place by the kernel
        if (_WGRAD3 and self.T == torch.bfloat16 and R == 3 and S == 3 and (not up or (_WGRAD3_UP and dy.H == 2 * x.H and dy.W == 2 * x.W))
                and kw.get("stride", 1) == 1 and kw.get("pad_t") == 1 and kw.get("pad_l") == 1 and Creal == x.C):
            boost = bool(boost_req)
            pkey = ("w3", x.B, dy.H, dy.W, x.C, dy.C, boost)
            patch = self._eff_splits.get(pkey)
            if patch is None:
                patch = ops.conv3x3_wgrad_splits(x.B, dy.H, dy.W, x.C, dy.C)
                if boost and patch:                     # the whole chip instead of half of it (see _TAIL_BLOCKS)
                    patch = ops.conv3x3_wgrad_splits(x.B, dy.H, dy.W, x.C, dy.C, 2 * patch)
                self._eff_splits[pkey] = patch
        point = 0
        if (not patch and _WGRAD1 and self.T == torch.bfloat16 and R == 1 and S == 1 and not kw.get("upsample") and kw.get("stride", 1) == 1
                and not kw.get("pad_t") and not kw.get("pad_l") and Creal == x.C and Nreal == dy.C):
            pkey = ("w1", dy.rows, x.C, dy.C)
            point = self._eff_splits.get(pkey)
            if point is None:
                point = self._eff_splits[pkey] = ops.conv1x1_wgrad_splits(dy.rows, x.C, dy.C)
        if point:
            # 1x1: slab kernel (csrc/wgrad1x1.hip) — one block per CU, partial tiles stored from registers, bias gradient folded in
            bias_ptr = self._pptr(ctx, bias) if bias is not None else 0
            with self._leaf(ctx, dy, x):
                n = Nreal * Creal
                stride = (n + 3) // 4 * 4
                bstride = (Nreal + 3) // 4 * 4
                slab = self._slabs.get(id(weight))
                if slab is None or slab.numel() < point * (stride + bstride):
                    slab = self._slabs[id(weight)] = torch.empty(point * (stride + bstride), dtype=torch.float32, device=self.device)
                bslab = slab.data_ptr() + 4 * point * stride
                ops.conv1x1_wgrad(dy, x, slab.data_ptr(), stride, bslab if bias is not None else 0, bstride, Nreal, point)
                ctx["slab_rows"].append((slab.data_ptr(), self._pptr(ctx, weight), n, point, stride))
                if bias is not None:
                    ctx["slab_rows"].append((bslab, bias_ptr, Nreal, point, bstride))
            did_bias = bias is not None
        elif patch:
            bias_ptr = self._pptr(ctx, bias) if bias is not None else 0
            with self._leaf(ctx, dy, x):
                if _WGRAD3_ATOMIC:
                    ops.conv3x3_wgrad(dy, x, self._pptr(ctx, weight), 0, bias_ptr, 0, Nreal, patch, upsample=up)
                else:
                    n = Nreal * 9 * Creal
                    stride = (n + 3) // 4 * 4
                    bstride = (Nreal + 3) // 4 * 4
                    slab = self._slabs.get(id(weight))
                    if slab is None or slab.numel() < patch * (stride + bstride):
                        slab = self._slabs[id(weight)] = torch.empty(patch * (stride + bstride), dtype=torch.float32, device=self.device)
                    bslab = slab.data_ptr() + 4 * patch * stride
                    ops.conv3x3_wgrad(dy, x, slab.data_ptr(), stride, bslab if bias is not None else 0, bstride, Nreal, patch, upsample=up)
                    ctx["slab_rows"].append((slab.data_ptr(), self._pptr(ctx, weight), n, patch, stride))
                    if bias is not None:
                        ctx["slab_rows"].append((bslab, bias_ptr, Nreal, patch, bstride))
            did_bias = bias is not None
        else:
            key = (dy.rows, splits, x.dtype)
            eff = self._eff_splits.get(key)
            if eff is None:
                eff = self._eff_splits[key] = ops.wgrad_effective_splits(dy.rows, splits, x.dtype)
            with self._leaf(ctx, dy, x):
                if _WGRAD_SLABS and eff > 1:
                    n = Nreal * R * S * Creal
                    stride = (n + 3) // 4 * 4
                    slab = self._slabs.get(id(weight))
                    if slab is None or slab.numel() < eff * stride:
                        slab = self._slabs[id(weight)] = torch.empty(eff * stride, dtype=torch.float32, device=self.device)
                    ops.conv2d_wgrad(dy, x, slab.data_ptr(), Creal, Nreal, R, S, splits=eff, slab_stride=stride, **kw)
                    ctx["slab_rows"].append((slab.data_ptr(), self._pptr(ctx, weight), n, eff, stride))
                else:
                    ops.conv2d_wgrad(dy, x, self._pptr(ctx, weight), Creal, Nreal, R, S, splits=eff, **kw)
        # chunk bookkeeping (also without a process group: a chunk whose convolutions are through their backward gets its time-projection
        # rows NOW, on the side stream, instead of on the tail of the step)
        for ch in ctx["pending"]:
            ch[2].discard(id(weight))
        while ctx["pending"]:
            a, b, mem, k = ctx["pending"][-1]
            if mem - {("fc_w", k)}:
                break                                        # conv gradients of this chunk are still to come
            if ("fc_w", k) in mem:                           # its residual blocks are through their backward: their time-projection rows
                self._fc_wgrad(ctx, k)
            ctx["pending"].pop()
            if ctx["dp"]:
                works, chunk, side = ctx["works"], ctx["gpack"][a:b], ctx.get("side")
                if side is not None and _DP_ISSUE_ON_SIDE and not torch.cuda.is_current_stream_capturing():
                    # Everything in a chunk is produced on the SIDE stream (weight gradients, their slab sums, the time-projection rows).
                    # torch's communicators order a collective behind the stream that is current when it is issued: issued with the side
                    # stream current, the exchange waits for exactly its producers and the MAIN stream — the critical path — waits for
                    # nothing.  (Round 5, found on the one-rank RCCL bench line: joining the side stream into the main stream at every
                    # chunk, as below, cost 1.0 ms per step — nine stalls of the chain that the exchange is supposed to hide under.)
                    self._flush_slabs(ctx, on_side=True)

                    def issue(works=works, chunk=chunk, side=side):
                        with torch.cuda.stream(side):
                            works.append(self._all_reduce(chunk))
                    self._comm(ctx, issue)
                else:
                    # (single-stream runs, and steps being captured as hipGraph segments: a segment ends with its side branch joined, and
                    #  at replay the whole segment is work of the stream the graph is launched on)
                    self._flush_slabs(ctx)                   # the chunk's conv gradients must be summed before they travel
                    self._join_side(ctx)                     # ... and produced: the communicator orders itself after the main stream
                    self._comm(ctx, lambda works=works, chunk=chunk: works.append(self._all_reduce(chunk)))
        return did_bias

    def _fc_wgrad(self, ctx, k):
        """d(fc.weight) of the residual blocks of chunk k: rows [o0, o0 + n) of dW = dtb^T s_t (all of them one product: the blocks'
        columns of the concatenated projection are contiguous), into the chunk's own region of the staging buffer.  A leaf."""
        if self.chunk_fc[k] is None:
            return
        o0, n, fbase = self.chunk_fc[k]
        dtb, s_t, E, Ct = ctx["dtb"], ctx["s_t"], self.E, self.tb_total
        with self._leaf(ctx, dtb, s_t):
            _hip.call("ddpm_atb_f32", dtb.data_ptr() + 4 * o0, Ct, s_t.data_ptr(), E, ctx["gpack"].data_ptr() + 4 * fbase, E, n, E, ctx["B"], _hip.stream())

    def _flush_slabs(self, ctx, on_side=False):
        """Sum the slab copies recorded since the last flush into the staging buffer (one launch).  ``on_side``: queue it on the side
        stream, behind the kernels that wrote the copies, without making the main stream wait — the backward flushes like this every
        few layers, so that the ~0.25 ms of slab traffic runs under the critical path instead of after it."""
        rows = tuple(ctx["slab_rows"])
        if not rows:
            return
        ctx["slab_rows"] = []
        table = self._slab_tables.get(rows)
        if table is None:                                    # addresses are stable (persistent buffers): built once per geometry
            table = self._slab_tables[rows] = torch.tensor(rows, dtype=torch.int64, device=self.device)
        if on_side and ctx.get("side") is not None:
            _hip.call("ddpm_wgrad_reduce", table.data_ptr(), len(rows), ctx["side_handle"])
            return
        self._join_side(ctx)
        _hip.call("ddpm_wgrad_reduce", table.data_ptr(), len(rows), _hip.stream())

    def _comm(self, ctx, fn):
        """Run a communicator call now, or — while the step is being captured — between two graph segments at replay time."""
        if ctx.get("cut") is None:
            fn()
        else:
            ctx["cut"](fn)

    def _all_reduce(self, t):
        import torch.distributed as dist
        if self.dp_trace is not None:                      # bench.py: when (on the compute stream's timeline) each exchange is issued
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.dp_trace.append(("all_reduce", t.numel() * 4, ev))
        if self.dp_standin is not None:                    # bench.py, one rank: a copy kernel of the chunk's size stands in for the ring step
            self.dp_standin(t)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _dp_mark(self, what):
        if self.dp_trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.dp_trace.append((what, 0, ev))

    def _pptr(self, ctx, p):
        return ctx["gpack"].data_ptr() + 4 * self.poff[p if isinstance(p, (str, tuple)) else id(p)]

    def _grad_target(self, v):
        """(view to write d/dv into, accumulate flag); the first writer stores, later writers accumulate."""
        if v.grad is None:
            v.grad = self._new(v.B, v.H, v.W, v.C)
        acc = 1 if v.ginit else 0
        v.ginit = True
        return v.grad, acc

    def _linear(self, a, w, bias, M, N, K):
        out = self._f32(M, N)
        ops.gemm(a.data_ptr(), K, 0, 0, w.data_ptr(), K, 0, 0, out.data_ptr(), N, 0, M, N, K, _hip.F32, bias=_hip.ptr(bias), out_mode=1)
        return out

    # ================================================================ forward
    def forward(self, x, t, training, tape, seed_dev=0):
        """``seed_dev``: address of a device uint64 holding the per-step part of the dropout seed (captured training step:
        the host rewrites that word before every replay); 0 = the whole seed travels by value."""
        m = self.m
        if x.dim() != 4 or x.shape[1] != m.in_channels:
            raise ValueError(f"expected x of shape [B, {m.in_channels}, H, W], got {tuple(x.shape)}")
        B, _, H, W = x.shape
        if t.shape != (B,):
            raise ValueError(f"expected t of shape [{B}], got {tuple(t.shape)}")
        if (H % (1 << (self.L - 1))) or (W % (1 << (self.L - 1))):
            raise ValueError("spatial size must be divisible by 2**(levels-1)")
        x = x.contiguous().float()
        t = t.contiguous().to(torch.int64)
        save = tape is not None
        self._refresh_packs(save)
        ws = self._workspace(B, H, W)
        st = dict(B=B, ws=ws, save=save, training=training, tape=tape)
        drop_p = float(m.drop_rate) if training else 0.0
        st["drop_p"] = drop_p
        st["seed_dev"] = seed_dev if drop_p > 0 else 0
        if drop_p > 0 and not seed_dev:
            st["seed"] = self.next_dropout_seed()
        else:
            st["seed"] = 0

        # ---- time embedding path (fp32; functions.py:10-26, unet.py:122-126,207)
        if not save and self.tt_on:                              # inside a sampler: the rows of the precomputed table (see time_table)
            tb = self._f32(B, self.tb_total)
            _hip.call("ddpm_gather_rows_f32", self.time_table().data_ptr(), t.data_ptr(), tb.data_ptr(), B, self.tb_total, self.tt_T, _hip.stream())
        else:
            tb, saved = self._time_biases(t, B)
            if save:
                st["temb_saved"] = saved
        st["tb"] = tb

        # ---- decoder concat buffers: encoder outputs are written straight into their slice
        n, L = self.n, self.L
        enc = [(self.hid, 0)]                                   # (channels, level) of every hs entry
        for i in range(L):
            enc += [(self.chs[i], i)] * n
            if i != L - 1:
                enc.append((self.chs[i], i + 1))
        sizes = [(H >> i, W >> i) for i in range(L)]
        cats, k = [], 0
        h_ch = self.chs[-1]
        for i in range(L - 1, -1, -1):
            for j in range(n + 1):
                cs, lvl = enc[len(enc) - 1 - k]
                assert lvl == i
                full = self._new(B, sizes[i][0], sizes[i][1], h_ch + cs)
                full_parts = (full.chan_slice(0, h_ch), full.chan_slice(h_ch, h_ch + cs))
                cats.append((full, full_parts))
                h_ch = self.chs[i]
                k += 1
        assert k == len(enc)

        def skip_slot(e):
            return cats[len(enc) - 1 - e][1][1]

        # ---- encoder
        xin = self._new(B, H, W, self.vec)                      # channel-padded NHWC copy of x
        _hip.call("ddpm_nchw_to_nhwc", x.data_ptr(), xin.ptr, B, m.in_channels, H * W, self.vec, self.dcode, _hip.stream())
        e = 0
        cur = skip_slot(e)
        self._conv(st, m.in_conv, xin, cur, 3)
        for i in range(L):
            mods = m.downsamples[f"level_{i}"]
            for j in range(n):
                e += 1
                nxt = skip_slot(e)
                self._block(st, mods[j], cur, nxt)
                cur = nxt
            if i != L - 1:
                e += 1
                nxt = skip_slot(e)
                if m.resample_with_conv:
                    self._conv(st, mods[n][1], cur, nxt, 3, stride=2)
                else:
                    self._resample(st, cur, nxt, up=0)           # nn.AvgPool2d(2)
                cur = nxt
        # ---- middle
        hh, ww = sizes[-1]
        a = self._new(B, hh, ww, self.chs[-1])
        self._res(st, m.middle[0], cur, a)
        b = self._new(B, hh, ww, self.chs[-1])
        self._attn(st, m.middle[1], a, b)
        self._res(st, m.middle[2], b, cats[0][1][0])
        # ---- decoder
        k = 0
        for i in range(L - 1, -1, -1):
            mods = m.upsamples[f"level_{i}"]
            for j in range(n + 1):
                full, parts = cats[k]
                last_of_level = j == n
                if not last_of_level:
                    dst = cats[k + 1][1][0]
                else:
                    dst = self._new(B, sizes[i][0], sizes[i][1], self.chs[i])
                self._block(st, mods[j], full, dst, parts=parts)
                k += 1
                cur = dst
            if i != 0:
                dst = cats[k][1][0]
                if m.resample_with_conv:
                    self._conv(st, mods[n + 1][1], cur, dst, 3, upsample=1)
                else:
                    self._resample(st, cur, dst, up=1)           # nn.Upsample(2, "nearest") alone
                cur = dst
        # ---- head: GN + SiLU + conv -> NCHW fp32
        out = self._f32(B, m.out_channels, H, W)
        act = self._new(B, H, W, self.hid)
        stats = self._f32(B, ops.GN_GROUPS, 2) if save else None
        norm, conv = m.out_conv[0], m.out_conv[2]
        ops.gn_fwd(cur, act, norm.weight, norm.bias, stats, ws, silu=True)
        cw = self._packed(conv, save)
        ops.conv2d(act, cw.wf.data_ptr(), out.data_ptr(), 0, cw.N, 3, 3, H, W, pad_t=1, pad_l=1, bias=conv.bias.data_ptr(), out_mode=3, splitk=self.splitk)
        if save:
            tape.append(("head", cur, act, stats, st))
        return out

    def next_dropout_seed(self):
        """Per-forward dropout seed (every block offsets it by its own constant): advances with each training forward."""
        self.drop_calls += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self.drop_calls * 0x632BE59BD9B4E019) & ((1 << 63) - 1)

    # ---- forward pieces (each appends what its backward needs)
    def _conv(self, st, conv, x, out, k, stride=1, upsample=0):
        cw = self._packed(conv, st["save"])
        if stride == 2:                       # TF-SAME for k=3, s=2: pad (0,1) on even sizes, (1,1) on odd (modules.py:145-160)
            pt = 0 if x.H % 2 == 0 else 1
            pl = 0 if x.W % 2 == 0 else 1
        else:
            pt = pl = k // 2
        xin = x
        if cw.Cp != x.C:
            raise RuntimeError("channel padding mismatch")
        ops.conv2d(xin, cw.wf.data_ptr(), out.ptr, out.ld, cw.N, k, k, out.H, out.W, stride=stride, pad_t=pt, pad_l=pl,
                   upsample=upsample, bias=conv.bias.data_ptr(), splitk=self.splitk)
        if st["save"]:
            st["tape"].append(("conv", conv, x, out, k, stride, pt, pl, upsample))

    def _resample(self, st, x, out, up):
        """resample_with_conv=False (unet.py:169, :196): AvgPool2d(2) (up = 0) or the bare nearest-2x Upsample (up = 1), one launch each way."""
        ops.resample2x(x, out, up, 1.0 if up else 0.25)
        if st["save"]:
            st["tape"].append(("resample", x, out, up))

    def _resample_bwd(self, ctx, rec):
        _, x, out, up = rec
        dy = out.grad
        assert dy is not None and out.ginit
        g, acc = self._grad_target(x)
        # d/dx of the pool = a quarter of dy replicated over the 2x2 block; of the replication = the 2x2 sum of dy
        ops.resample2x(dy, g, 0 if up else 1, 1.0 if up else 0.25, accumulate=acc)

    def _block(self, st, blk, x, out, parts=None):
        res, att = self._split(blk)
        if att is None:
            self._res(st, res, x, out, parts)
        else:
            mid = self._new(out.B, out.H, out.W, out.C)
            self._res(st, res, x, mid, parts)
            self._attn(st, att, mid, out)

    def _res(self, st, rb, x, out, parts=None):
        B, save, ws = st["B"], st["save"], st["ws"]
        Cin, Cout = rb.in_channels, rb.out_channels
        tb = st["tb"]
        rowbias = tb.data_ptr() + 4 * self.tb_off[id(rb)]
        c1 = self._packed(rb.conv1, save)
        c2 = self._packed(rb.conv2, save)
        a1 = stats1 = a2 = stats2 = None
        seed = 0
        h1 = self._new(B, x.H, x.W, Cout)
        a1 = self._new(B, x.H, x.W, Cin)
        stats1 = self._f32(B, ops.GN_GROUPS, 2) if save else None
        ops.gn_fwd(x, a1, rb.norm1.weight, rb.norm1.bias, stats1, ws, silu=True)
        ops.conv2d(a1, c1.wf.data_ptr(), h1.ptr, h1.ld, Cout, 3, 3, x.H, x.W, pad_t=1, pad_l=1,
                   rowbias=rowbias, rowbias_ld=self.tb_total, splitk=self.splitk)
        if rb.has_skip:
            cs = self._packed(rb.skip, save)
            ops.conv2d(x, cs.wf.data_ptr(), out.ptr, out.ld, Cout, 1, 1, x.H, x.W, bias=rb.skip.bias.data_ptr(), splitk=self.splitk)
            res_ptr, res_ld = out.ptr, out.ld          # conv2's epilogue adds the skip projection it finds in `out`
        else:
            res_ptr, res_ld = x.ptr, x.ld
        a2 = self._new(B, x.H, x.W, Cout)
        stats2 = self._f32(B, ops.GN_GROUPS, 2) if save else None
        seed = (st["seed"] + 0x51ED27 * (self.tb_off[id(rb)] + 1)) & ((1 << 63) - 1) if st["drop_p"] > 0 else 0
        ops.gn_fwd(h1, a2, rb.norm2.weight, rb.norm2.bias, stats2, ws, silu=True, drop_p=st["drop_p"], seed=seed, seed_dev=st["seed_dev"])
        ops.conv2d(a2, c2.wf.data_ptr(), out.ptr, out.ld, Cout, 3, 3, x.H, x.W, pad_t=1, pad_l=1,
                   bias=rb.conv2.bias.data_ptr(), res_ptr=res_ptr, res_ld=res_ld, splitk=self.splitk)
        if save:
            st["tape"].append(("res", rb, x, out, a1, stats1, h1, a2, stats2, seed, parts))

    def _attn(self, st, ab, x, out):
        B, save, ws = st["B"], st["save"], st["ws"]
        C, Lk = ab.in_channels, x.H * x.W
        hn = self._new(B, x.H, x.W, C)
        stats = self._f32(B, ops.GN_GROUPS, 2) if save else None
        ops.gn_fwd(x, hn, ab.norm.weight, ab.norm.bias, stats, ws, silu=False)
        qkv = self._new(B, x.H, x.W, 3 * C)
        ci = self._packed(ab.project_in, save)
        ops.conv2d(hn, ci.wf.data_ptr(), qkv.ptr, qkv.ld, 3 * C, 1, 1, x.H, x.W, bias=ab.project_in.bias.data_ptr(), splitk=self.splitk)
        es, bs = self.es, Lk * 3 * C
        o = self._new(B, x.H, x.W, C)
        prob = lse = None
        scale = 1.0 / math.sqrt(C)
        flash = _FLASH_ATTENTION and self.T == torch.bfloat16 and Lk <= 256 and Lk % 16 == 0 and C <= 512 and C % 32 == 0
        if not save and _FUSED_ATTENTION and not (flash and _FLASH_INFERENCE) and self.T == torch.bfloat16 and Lk % 128 == 0 and C in (128, 256):
            # inference: one kernel, the L x L logits / probabilities stay in LDS and registers (unet.py:41-52); since the training
            # forward got its DMA rings and the one-butterfly softmax it is the faster one (27 vs 32 us at B=128, L=256, C=256) and serves
            # inference too; DDPM_FLASH_INFERENCE=0 brings this kernel back
            _hip.call("ddpm_attention_fwd", qkv.ptr, qkv.ld, o.ptr, o.ld, B, Lk, C, scale, self.dcode, _hip.stream())
        elif flash:
            # training (and the geometries the kernel above does not serve): same, plus the row log-sum-exp the backward
            # kernels rebuild the probabilities from — no L x L tensor exists in memory
            lse = self._f32(B, Lk) if save else None
            _hip.call("ddpm_attention_fwd_lse", qkv.ptr, qkv.ld, o.ptr, o.ld, lse.data_ptr() if save else 0, B, Lk, C, scale,
                      self.dcode, _hip.stream())
        else:
            if Lk % self.vec:
                # (the probability matrix [B][L][L] is an operand of the second product: its rows must be 16-byte multiples.  No shipped
                #  configuration attends over fewer than 16 tokens; the reference itself has no such limit)
                raise NotImplementedError(f"attention over {Lk} tokens in the {'bf16' if self.T == torch.bfloat16 else 'fp32'} mode: the token "
                                          f"count must be a multiple of {self.vec} (use a larger image, or the fp32 mode for multiples of 4)")
            q, kk, v = qkv.ptr, qkv.ptr + C * es, qkv.ptr + 2 * C * es
            logits = self._f32(B, Lk, Lk)                       # S = Q K^T / sqrt(C)   (unet.py:46-48)
            ops.gemm(q, 3 * C, bs, 0, kk, 3 * C, bs, 0, logits.data_ptr(), Lk, Lk * Lk, Lk, Lk, C, self.dcode, batch=B,
                     alpha=scale, out_mode=1)
            prob = _hip.retain(torch.empty((B, Lk, Lk), dtype=self.T, device=self.device))
            _hip.call("ddpm_softmax_fwd", logits.data_ptr(), prob.data_ptr(), B * Lk, Lk, self.dcode, _hip.stream())
            ops.gemm(prob.data_ptr(), Lk, Lk * Lk, 0, v, 3 * C, bs, 1, o.ptr, C, Lk * C, Lk, C, Lk, self.dcode, batch=B)   # O = P V (unet.py:50)
        co = self._packed(ab.project_out, save)
        ops.conv2d(o, co.wf.data_ptr(), out.ptr, out.ld, C, 1, 1, x.H, x.W, bias=ab.project_out.bias.data_ptr(),
                   res_ptr=x.ptr, res_ld=x.ld, splitk=self.splitk)
        if save:
            st["tape"].append(("attn", ab, x, out, hn, stats, qkv, prob, o, lse))

    # ================================================================ backward
    def _open_backward(self, st, gflat=None, cut=None):
        """Gradient staging buffers + stream / communicator bookkeeping of one backward pass."""
        B, ws = st["B"], st["ws"]
        if gflat is None:
            gflat = self._f32(self.gtotal)   # written only by the final unpack
        dtb = self._zeros(B, self.tb_total)
        self._wgrad_table()
        if self._gpack is None or self._gpack.numel() != self.ptotal:
            self._gpack = torch.empty(self.ptotal, dtype=torch.float32, device=self.device)
        gpack = self._gpack
        _hip.call("ddpm_fill_zero", gpack.data_ptr(), gpack.numel() * 4, _hip.stream())
        # persistent (stable addresses for the slab-reduce tables): packed conv weight grads [N][RS][C] + tail
        ctx = dict(gflat=gflat, gpack=gpack, ws=ws, B=B, dtb=dtb, s_t=st["temb_saved"][4], pending=None, dp=self.pg is not None, works=self._works, slab_rows=[], side=None, keep=[], cut=cut,
                   seed_dev=st.get("seed_dev", 0), world=1, sumsq=0)
        if _SIDE_STREAM and gflat.is_cuda:
            if self._side is None:
                # Default priority 0 = the main stream's own: measured, a lower-priority side stream (DDPM_SIDE_PRIORITY=1..) did not move
                # the step (the weight-gradient stream has ~3.6 ms of work per ~9.6 ms step and is confined to half the CUs by the
                # kernels' own grids, DDPM_WGRAD3_CUS); the switch stays for experiments.
                self._side = torch.cuda.Stream(device=self.device, priority=_SIDE_PRIORITY)
            ctx["side"] = self._side
            ctx["side_handle"] = self._side.cuda_stream
            ctx["main"] = torch.cuda.current_stream()
            ctx["main_handle"] = _hip.stream()
            _hip.call("ddpm_stream_order", ctx["side_handle"], ctx["main_handle"])   # the staging buffer is zeroed on the main stream
        # chunks in execution order; the backward finishes them from the last one backwards
        ctx["pending"] = [(a, b, set(mem), k) for k, (a, b, mem) in enumerate(self.chunks)]
        if self.pg is not None:
            import torch.distributed as dist
            ctx["world"] = dist.get_world_size(self.pg)
        return ctx

    def _close_backward(self, ctx, st):
        """Time-embedding path, outstanding reductions / collectives, then ONE launch that rewrites the staging buffer into
        the flat parameter-order gradient (x 1/world)."""
        gpack, gflat = ctx["gpack"], ctx["gflat"]
        self._temb_bwd(ctx, st)
        self._flush_slabs(ctx)
        self._join_side(ctx)
        if self.pg is not None:
            assert not ctx["pending"], "a conv weight gradient was never produced"
            works, tail = ctx["works"], gpack[self.tail[0]:self.tail[1]]

            def finish():
                self._dp_mark("backward_compute_done")
                works.append(self._all_reduce(tail))
                for w in works:
                    w.wait()                               # the compute stream waits for the communicator; no host sync
                works.clear()
                self._dp_mark("exchange_done")
            self._comm(ctx, finish)
        wdesc = self._wgrad_table()
        if ctx["sumsq"]:           # the caller's squared-norm buffer (_hip.SUMSQ_FLOATS): the clip norm comes out of this pass (no second read of all gradients)
            if wdesc.shape[0] > _hip.SUMSQ_MAX_TENSORS:
                raise ValueError(f"{wdesc.shape[0]} gradient rows exceed the squared-norm buffer ({_hip.SUMSQ_MAX_TENSORS})")
            _hip.call("ddpm_wgrad_unpack_sumsq", gpack.data_ptr(), gflat.data_ptr(), wdesc.data_ptr(), wdesc.shape[0], 1.0 / ctx["world"], ctx["sumsq"], _hip.SUMSQ_FLOATS, _hip.stream())
        else:
            _hip.call("ddpm_wgrad_unpack", gpack.data_ptr(), gflat.data_ptr(), wdesc.data_ptr(), wdesc.shape[0], 1.0 / ctx["world"], _hip.stream())
        return gflat

    def backward(self, tape, gout, gflat=None, cut=None, want_views=True, sumsq=0, want_dx=False):
        """Replays the tape in reverse.  ``gflat``: caller-owned flat gradient buffer (stable address for captured steps).
        ``cut(fn)``: when the step is being captured as a sequence of hipGraphs, the communicator calls are not captured —
        ``cut`` ends the current graph segment, registers ``fn`` to run eagerly between the segments at replay time, and
        opens the next segment."""
        m = self.m
        head = tape[-1]
        st = head[4]
        B, ws = st["B"], st["ws"]
        gout = gout.contiguous().float()
        H, W = gout.shape[2], gout.shape[3]
        ctx = self._open_backward(st, gflat, cut)
        ctx["sumsq"] = sumsq             # device address of _hip.SUMSQ_FLOATS floats that receive ||grad||^2 in lane 0 (0: not wanted)
        ctx["want_dx"] = want_dx         # also d/d(input image) -> self.last_dx (NCHW fp32), see _conv_bwd
        # ---- head
        _, cur, act, stats, _ = head
        norm, conv = m.out_conv[0], m.out_conv[2]
        cw = self.convs[id(conv)]
        dy = self._new(B, H, W, cw.Np)                          # NCHW fp32 grad -> channel-padded NHWC
        _hip.call("ddpm_nchw_to_nhwc", gout.data_ptr(), dy.ptr, B, m.out_channels, H * W, cw.Np, self.dcode, _hip.stream())
        dact = self._new(B, H, W, self.hid)
        ops.conv2d(dy, cw.wd.data_ptr(), dact.ptr, dact.ld, self.hid, 3, 3, H, W, pad_t=1, pad_l=1, splitk=self.splitk)
        self._wgrad(ctx, conv.weight, dy, act, self.hid, cw.N, 3, 3, pad_t=1, pad_l=1, splits=self._splits(cw.N, 9 * self.hid, B * H * W))
        self._bias_grad(ctx, dy, [conv.bias], cw.N, slot="out_b")
        g, acc = self._grad_target(cur)
        ops.gn_bwd(cur, dact, g, norm.weight, norm.bias, stats, self._pptr(ctx, norm.weight), self._pptr(ctx, norm.bias), ws, silu=True, accumulate=acc)
        # ---- the rest of the tape in reverse
        for rec in reversed(tape[:-1]):
            kind = rec[0]
            # (the same grouping of rows whether the step runs eagerly, is being recorded or is being captured: the reduce tables are
            #  cached per group and built with a host -> device copy, which a stream capture does not allow)
            if len(ctx["slab_rows"]) >= _SLAB_FLUSH_ROWS and (not ctx["dp"] or (_DP_ISSUE_ON_SIDE and ctx["side"] is not None)):
                self._flush_slabs(ctx, on_side=True)
            if kind == "res":
                self._res_bwd(ctx, rec)
            elif kind == "attn":
                self._attn_bwd(ctx, rec)
            elif kind == "resample":
                self._resample_bwd(ctx, rec)
            else:
                self._conv_bwd(ctx, rec)
        gflat = self._close_backward(ctx, st)
        if not self.debug_keep_tape:
            # the head record holds `st` and `st` holds the tape: break the cycle so that the saved activations go back to the
            # allocator NOW (by reference count) instead of whenever the cyclic GC runs — with the cycle in place every few
            # steps paid ~75 ms of fresh hipMalloc calls
            st.pop("tape", None)
            tape.clear()
        if not want_views:                    # the direct training step addresses the flat buffer itself (604 tensor ops less per step)
            return None
        grads = []
        for p in self.params:
            o = self.goff[id(p)]
            grads.append(gflat[o:o + p.numel()].view(p.shape) if p.requires_grad else None)
        return grads

    def _splits(self, M, N, K):
        """Split the wgrad reduction so that tiles x splits is about one full wave of blocks (256 CUs x 2 resident blocks).
        Every split ends in 128x128 fp32 atomics (~20 us per block at the L2's atomic rate), so a split must keep enough
        K-steps to amortise them: >= 20 when K allows it, >= 8 on the short reductions (scripts/microbench.py sweeps)."""
        tiles = -(-M // 128) * -(-N // 128)
        ksteps = -(-K // (8 * self.vec))
        return max(1, min(_WGRAD_TARGET_BLOCKS // tiles, ksteps // (_WGRAD_MINSTEPS if ksteps >= 100 else min(8, _WGRAD_MINSTEPS))))

    def _bias_grad(self, ctx, dy, biases, creal, slot=None):
        """db[c] = sum over pixels and batch of dy.  One owner: atomics straight into its gradient.  Several owners or a
        channel-padded dy: reduce into a gpack slot that the unpack table fans out."""
        with self._leaf(ctx, dy):
            if slot is None:
                assert dy.C == creal and len(biases) == 1
                ops.colsum(dy, 0, 0, self._pptr(ctx, biases[0]))
            else:
                ops.colsum(dy, 0, 0, self._pptr(ctx, slot))

    def _conv_bwd(self, ctx, rec):
        _, conv, x, out, k, stride, pt, pl, upsample = rec
        gflat, B = ctx["gflat"], ctx["B"]
        cw = self.convs[id(conv)]
        dy = out.grad
        assert dy is not None and out.ginit
        first = conv is self.m.in_conv
        self._wgrad(ctx, conv.weight, dy, x, cw.C, cw.N, k, k, stride=stride, pad_t=pt, pad_l=pl, upsample=upsample,
                         splits=self._splits(cw.N, k * k * cw.Cp, dy.rows))
        self._bias_grad(ctx, dy, [conv.bias], cw.N)
        if first:
            # no gradient w.r.t. the input image — unless the autograd graph asked for it (UNet.forward with x.requires_grad): the 3-channel
            # data gradient straight into NCHW fp32 (the out_conv-shaped launch: few output channels, out_mode 3)
            if ctx.get("want_dx"):
                dx = torch.empty(B, cw.C, x.H, x.W, dtype=torch.float32, device=self.device)
                ops.conv2d(dy, cw.wd.data_ptr(), dx.data_ptr(), 0, cw.C, k, k, x.H, x.W, pad_t=k - 1 - pt, pad_l=k - 1 - pl, out_mode=3,
                           splitk=self.splitk)
                self.last_dx = dx
            return
        if upsample and cw.up:
            # d/dx of (nearest-2x upsample -> 3x3 conv) = one 4x4 / stride-2 / pad-1 conv over dy with the summed taps the pack
            # kernel prepared: 16 taps on a quarter of the pixels, no full-resolution intermediate, no 2x2 reduction pass
            g, acc = self._grad_target(x)
            ops.conv2d(dy, cw.wd.data_ptr(), g.ptr, g.ld, cw.C, 4, 4, x.H, x.W, stride=2, pad_t=1, pad_l=1, accumulate=acc, splitk=self.splitk)
        elif upsample:
            tmp = self._new(B, dy.H, dy.W, cw.C)            # dgrad at the upsampled resolution, then 2x2 sum
            ops.conv2d(dy, cw.wd.data_ptr(), tmp.ptr, tmp.ld, cw.C, k, k, dy.H, dy.W, pad_t=k - 1 - pt, pad_l=k - 1 - pl, splitk=self.splitk)
            g, acc = self._grad_target(x)
            _hip.call("ddpm_upsample2x_bwd", tmp.ptr, g.ptr, g.ld, B, x.H, x.W, cw.C, acc, self.dcode, _hip.stream())
        else:
            g, acc = self._grad_target(x)
            ops.conv2d(dy, cw.wd.data_ptr(), g.ptr, g.ld, cw.C, k, k, x.H, x.W, pad_t=k - 1 - pt, pad_l=k - 1 - pl,
                       dilate=1 if stride == 2 else 0, accumulate=acc, splitk=self.splitk)

    def _res_bwd(self, ctx, rec):
        _, rb, x, out, a1, stats1, h1, a2, stats2, seed, parts = rec
        gflat, ws, B = ctx["gflat"], ctx["ws"], ctx["B"]
        Cin, Cout = rb.in_channels, rb.out_channels
        dout = out.grad
        assert dout is not None and out.ginit
        drop_p = float(rb.drop_rate) if seed else 0.0
        c1, c2 = self.convs[id(rb.conv1)], self.convs[id(rb.conv2)]
        # conv2
        da2 = self._new(B, x.H, x.W, Cout)
        ops.conv2d(dout, c2.wd.data_ptr(), da2.ptr, da2.ld, Cout, 3, 3, x.H, x.W, pad_t=1, pad_l=1, splitk=self.splitk)
        b2 = ("b2", id(rb)) if rb.has_skip else rb.conv2.bias
        # (... on images up to 64 x 64 only: on the CelebA-HQ step the same boost COSTS 1.4 % — 11.71 against 11.55 ms — its first blocks sit
        #  at 256 x 256 / 128 x 128 with a long main-stream chain still to come; DDPM_WGRAD3_TAIL_ANY=1 lifts the limit)
        boost = _TAIL_BLOCKS > 0 and (x.H * x.W <= 4096 or _TAIL_ANY) and any(rb is q for q in self.res_blocks[:_TAIL_BLOCKS])

        if not self._wgrad(ctx, rb.conv2.weight, dout, a2, Cout, Cout, 3, 3, pad_t=1, pad_l=1, splits=self._splits(Cout, 9 * Cout, dout.rows), bias=b2, boost=boost):
            self._bias_grad(ctx, dout, [rb.conv2.bias], Cout, slot=("b2", id(rb)) if rb.has_skip else None)
        # GN2 + SiLU + dropout
        dh1 = self._new(B, x.H, x.W, Cout)
        # ... which also yields the gradient of the time bias (+ conv1 bias): the per-sample column sums of dh1 land in the
        # block's columns of the concatenated dtb
        ops.gn_bwd(h1, da2, dh1, rb.norm2.weight, rb.norm2.bias, stats2, self._pptr(ctx, rb.norm2.weight), self._pptr(ctx, rb.norm2.bias),
                   ws, silu=True, drop_p=drop_p, seed=seed, seed_dev=ctx["seed_dev"],
                   colsum_ptr=ctx["dtb"].data_ptr() + 4 * self.tb_off[id(rb)], colsum_ld=self.tb_total)
        # conv1
        da1 = self._new(B, x.H, x.W, Cin)
        ops.conv2d(dh1, c1.wd.data_ptr(), da1.ptr, da1.ld, Cin, 3, 3, x.H, x.W, pad_t=1, pad_l=1, splitk=self.splitk)
        self._wgrad(ctx, rb.conv1.weight, dh1, a1, Cin, Cout, 3, 3, pad_t=1, pad_l=1, splits=self._splits(Cout, 9 * Cin, dh1.rows), boost=boost)
        # GN1 + SiLU, then the skip path, into d(x)
        g, acc = self._grad_target(x)
        # (identity skip: the block's output gradient joins d(x) inside the same kernel)
        ops.gn_bwd(x, da1, g, rb.norm1.weight, rb.norm1.bias, stats1, self._pptr(ctx, rb.norm1.weight), self._pptr(ctx, rb.norm1.bias),
                   ws, silu=True, accumulate=acc, add=None if rb.has_skip else dout)
        if rb.has_skip:
            cs = self.convs[id(rb.skip)]
            ops.conv2d(dout, cs.wd.data_ptr(), g.ptr, g.ld, Cin, 1, 1, x.H, x.W, accumulate=1, splitk=self.splitk)
            self._wgrad(ctx, rb.skip.weight, dout, x, Cin, Cout, 1, 1, splits=self._splits(Cout, Cin, dout.rows))
        if parts is not None:                                 # x was a concat buffer: hand each producer its slice
            c0 = parts[0].C
            for pv, (a, b) in zip(parts, ((0, c0), (c0, x.C))):
                pv.grad, pv.ginit = g.chan_slice(a, b), True

    def _attn_bwd(self, ctx, rec):
        _, ab, x, out, hn, stats, qkv, prob, o, lse = rec
        gflat, ws, B = ctx["gflat"], ctx["ws"], ctx["B"]
        C, Lk = ab.in_channels, x.H * x.W
        dout = out.grad
        assert dout is not None and out.ginit
        ci, co = self.convs[id(ab.project_in)], self.convs[id(ab.project_out)]
        es, bs = self.es, Lk * 3 * C
        # project_out
        do = self._new(B, x.H, x.W, C)
        ops.conv2d(dout, co.wd.data_ptr(), do.ptr, do.ld, C, 1, 1, x.H, x.W, splitk=self.splitk)
        if not self._wgrad(ctx, ab.project_out.weight, dout, o, C, C, 1, 1, splits=self._splits(C, C, dout.rows), bias=ab.project_out.bias):
            self._bias_grad(ctx, dout, [ab.project_out.bias], C)
        # attention core
        q, kk, v = qkv.ptr, qkv.ptr + C * es, qkv.ptr + 2 * C * es
        dqkv = self._new(B, x.H, x.W, 3 * C)
        dq, dk, dv = dqkv.ptr, dqkv.ptr + C * es, dqkv.ptr + 2 * C * es
        scale = 1.0 / math.sqrt(C)
        if lse is not None:
            # flash-style: P is rebuilt from q, k and the saved log-sum-exp inside the kernels (dQ; then dK and dV)
            dvec = self._f32(B, Lk)
            _hip.call("ddpm_attention_bwd", qkv.ptr, qkv.ld, o.ptr, o.ld, do.ptr, do.ld, lse.data_ptr(), dvec.data_ptr(),
                      dqkv.ptr, dqkv.ld, B, Lk, C, scale, self.dcode, _hip.stream())
        else:
            dp = self._f32(B, Lk, Lk)                            # dP = dO V^T
            ops.gemm(do.ptr, C, Lk * C, 0, v, 3 * C, bs, 0, dp.data_ptr(), Lk, Lk * Lk, Lk, Lk, C, self.dcode, batch=B, out_mode=1)
            # dV = P^T dO
            ops.gemm(prob.data_ptr(), Lk, Lk * Lk, 1, do.ptr, C, Lk * C, 1, dv, 3 * C, bs, Lk, C, Lk, self.dcode, batch=B)
            ds = _hip.retain(torch.empty((B, Lk, Lk), dtype=self.T, device=self.device))
            _hip.call("ddpm_softmax_bwd", prob.data_ptr(), dp.data_ptr(), ds.data_ptr(), B * Lk, Lk, self.dcode, _hip.stream())
            ops.gemm(ds.data_ptr(), Lk, Lk * Lk, 0, kk, 3 * C, bs, 1, dq, 3 * C, bs, Lk, C, Lk, self.dcode, batch=B, alpha=scale)   # dQ = dS K
            ops.gemm(ds.data_ptr(), Lk, Lk * Lk, 1, q, 3 * C, bs, 1, dk, 3 * C, bs, Lk, C, Lk, self.dcode, batch=B, alpha=scale)    # dK = dS^T Q
        # project_in
        dhn = self._new(B, x.H, x.W, C)
        ops.conv2d(dqkv, ci.wd.data_ptr(), dhn.ptr, dhn.ld, C, 1, 1, x.H, x.W, splitk=self.splitk)
        if not self._wgrad(ctx, ab.project_in.weight, dqkv, hn, C, 3 * C, 1, 1, splits=self._splits(3 * C, C, dqkv.rows), bias=ab.project_in.bias):
            self._bias_grad(ctx, dqkv, [ab.project_in.bias], 3 * C)
        # GN (no SiLU) + identity residual
        g, acc = self._grad_target(x)
        ops.gn_bwd(x, dhn, g, ab.norm.weight, ab.norm.bias, stats, self._pptr(ctx, ab.norm.weight), self._pptr(ctx, ab.norm.bias),
                   ws, silu=False, accumulate=acc, add=dout)

    def _temb_bwd(self, ctx, st):
        m, gflat, B, dtb, E = self.m, ctx["gflat"], ctx["B"], ctx["dtb"], self.E
        temb, e1, s1, t_emb, s_t, fc_w = st["temb_saved"]
        Ct = self.tb_total
        F = _hip.F32
        # fc (all blocks at once): dW = dtb^T s_t ; db = colsum(dtb) ; d(s_t) = dtb W.  dW / db land in gpack slots that the
        # unpack table fans out to every fc.weight / fc.bias / conv1.bias.
        # (parameter gradients are leaves: on the side stream, next to the chain d(s_t) -> d(t_emb) -> d(s1) -> d(e1) that is the very
        #  end of the backward's critical path)
        # (the rows of d(fc.weight) were produced chunk by chunk inside the backward: _fc_wgrad.  A PARTIAL backward — a harness that
        #  runs one block's routines — leaves chunks open: their rows are produced here)
        for _, _, mem, k in ctx["pending"]:
            if ("fc_w", k) in mem:
                self._fc_wgrad(ctx, k)
                mem.discard(("fc_w", k))
        with self._leaf(ctx, dtb, s_t):
            ops.colsum(View(dtb, 1, B, 1, Ct), 0, 0, self._pptr(ctx, "fc_b"))
        # K = sum Cout (~5000): split-K with fp32 atomics (run-to-run summation order) — unless the deterministic-reduction mode is on
        # (DDPM_WGRAD_SLABS=1: bit-reproducible gradients), where this product and the lin2 one below run as single-pass GEMMs
        det = _WGRAD_SLABS
        ds_t = self._f32(B, E) if det else self._zeros(B, E)
        ops.gemm(dtb.data_ptr(), Ct, 0, 0, fc_w.data_ptr(), E, 0, 1, ds_t.data_ptr(), E, 0, B, E, Ct, F, out_mode=1 if det else 2,
                 splits=1 if det else max(1, Ct // 256))
        dt_emb = self._f32(B, E)
        _hip.call("ddpm_silu_bwd", t_emb.data_ptr(), ds_t.data_ptr(), dt_emb.data_ptr(), B * E, 0, _hip.stream())
        ctx["dt_emb"] = dt_emb                                  # d/d(t_emb): what the reference's ResidualBlock hands back to the embedding MLP
        lin2, lin1 = m.embed[2], m.embed[0]
        # (K = E = 512 over 4 output tiles would leave 252 CUs idle for ~75 us at the very end of the backward: split-K with fp32 atomics
        #  like the fc product above — output tiles x E / 64 K slices)
        ds1 = self._f32(B, E) if det else self._zeros(B, E)
        ops.gemm(dt_emb.data_ptr(), E, 0, 0, lin2.weight.data_ptr(), E, 0, 1, ds1.data_ptr(), E, 0, B, E, E, F, out_mode=1 if det else 2,
                 splits=1 if det else max(1, E // 64))
        de1 = self._f32(B, E)
        _hip.call("ddpm_silu_bwd", e1.data_ptr(), ds1.data_ptr(), de1.data_ptr(), B * E, 0, _hip.stream())
        # The embedding MLP's parameter gradients stay on THIS stream, behind the chain that produced their inputs: the chain ends here while the
        # side stream still has the first layers' weight gradients, the last chunk's time-projection rows and a run of bias sums queued — as
        # leaves these four launches sat at the end of that queue and the join below waited ~90 us longer for them (profiles/r06_step_tail.txt).
        # (DDPM_TEMB_LEAVES=1: as leaves on the side stream again, for A/B runs)
        with (self._leaf(ctx, dt_emb, s1, de1, temb) if _TEMB_LEAVES else contextlib.nullcontext()):
            _hip.call("ddpm_atb_f32", dt_emb.data_ptr(), E, s1.data_ptr(), E, self._pptr(ctx, lin2.weight), E, E, E, B, _hip.stream())
            ops.colsum(View(dt_emb, 1, B, 1, E), 0, 0, self._pptr(ctx, lin2.bias))
            _hip.call("ddpm_atb_f32", de1.data_ptr(), E, temb.data_ptr(), self.hid, self._pptr(ctx, lin1.weight), self.hid, E, self.hid, B, _hip.stream())
            ops.colsum(View(de1, 1, B, 1, E), 0, 0, self._pptr(ctx, lin1.bias))
