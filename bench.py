"""Headline benchmark (BASELINE.json): training imgs/s and 1000-step DDPM samples/s of the CIFAR-10 UNet on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU under torch.distributed.run — started bare, bench.py
                                                            launches the ranks itself; a WORLD_SIZE that differs from N is an error)

One "step" = one full reference training step (ddpm_torch/utils/train.py:148-170: forward, backward, global-norm clip,
Adam, LR schedule, EMA, loss reduce incl. loss.item()) on B=128 synthetic 32x32 images PER GPU (weak scaling), bf16
compute with fp32 master weights / gradients / optimizer state, dropout 0.1 active — configs/cifar10.json of the
reference (BASELINE config 2; config 3 at N > 1).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  Beyond the driver's contract it carries:
  roofline     — the MFMA kernel with the most GPU time per step (ranked with every kernel alone on the chip): algorithmic
                 FLOPs of its launches / their summed HIP-event durations in the product step (events recorded on the launch
                 stream — main or side — around every MFMA launch of one extra step), vs the dense bf16 peak; `isolated` = the
                 same with the two-stream overlap switched off; `runner_up` = the next kernel by in-product time; `traffic` =
                 HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/, see scripts/gpu_profile.sh)
  sampling     — eval-mode ancestral sampling (B=128, 1000 steps, hipGraph replay) + a batch sweep
  other_configs— N = 1 only: the fp32 (parity) mode of the same step / sampler, BASELINE config 4 (CelebA 64x64 UNet,
                 DDIM-50, B=128) and the per-GPU work of config 5 (CelebA-HQ 256x256 UNet, B=2 training step)
  cpu_baseline — the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CIFAR = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
             apply_attn=[False, True, False, False], drop_rate=0.1)
CELEBA = dict(CIFAR, apply_attn=[False, False, True, False], drop_rate=0.0)
CELEBAHQ = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 1, 2, 2, 4, 4], num_res_blocks=2,
                apply_attn=[False, False, False, False, True, False], drop_rate=0.0)
# forward GFLOP per sample (2*MAC over convs, linears, attention matmuls); a training step counts 3x
FWD_GFLOP = {"cifar": 12.443713536, "celeba": 46.741, "celebahq": 497.028}
PEAK = {"bf16": 2500.0, "fp32": 157.3}                    # dense MFMA TFLOP/s, MI355X_MICROARCH.md
B_PER_GPU = 128
VARIANT = {1: "gemm_kernel<4 waves,128x128>", 2: "gemm_kernel<8 waves,128x128>", 3: "gemm_kernel<deep ring,128x128>",
           4: "gemm64_kernel<64x64>", 5: "conv3x3_halo_kernel<256px x 128>", 6: "wgrad3x3_kernel<64x32 x 9 taps> (8x8 / 4x4 images)", 7: "pw_conv_kernel<persistent 1x1>", 8: "conv3x3_stream_kernel<16> (persistent, 256px x 128)",
           9: "wgrad1x1_kernel<128c x 128n slabs>", 10: "conv3x3_stream_kernel<8> (persistent, 64px x 128)",
           11: "conv3x3_few_out_kernel (out_conv)", 12: "conv3x3_few_in_kernel (in_conv)",
           13: "conv3x3_pc_kernel (persistent, 256px x 128, loader + consumer waves)",
           14: "wgrad3x3_ws_kernel (64x64 x 9 taps, 4 consumer + 4 loader waves)"}


HBM_KIND = {"hbm_gn_fwd": "GroupNorm(32)+SiLU(+dropout) forward (gn_lds_fwd / gn_apply family, norm.hip)",
            "hbm_gn_bwd": "GroupNorm(32)+SiLU(+dropout) backward (gn_lds_bwd / gn_bwd_apply family, norm.hip)",
            "hbm_pw": "1x1 conv forward / data gradient (pw_conv_kernel, pointwise.hip)",
            "hbm_wg1": "1x1 conv weight gradient (wgrad1x1_kernel, wgrad1x1.hip)"}
HBM_ACHIEVABLE = 6300.0                                   # GB/s: measured float4-copy rate, MI355X_MICROARCH.md (8000 spec)

# share of the 256 CUs a wgrad3x3 launch takes (csrc/wgrad.hip: block budget, DDPM_WGRAD3_CUS)
WGRAD3_CU_SHARE = min(int(os.environ.get("DDPM_WGRAD3_CUS", "128")), 256) / 256.0


def host_cpu():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    return model, len(phys) or (os.cpu_count() or 1)


def cpu_baseline(seconds_budget=24.0):
    """Oracle (fp32, torch CPU ops) on a bounded sample: the CIFAR training step and the eval forward at B=16.
    Thread count: a 3-point sweep (more threads than this workload can feed makes oneDNN SLOWER: 128 threads on B=16 ran at half
    the 8-core rate), one step each; the best count then runs the timed steps."""
    from oracle import diffusion_ref as D, train_ref, unet_ref as U
    torch.manual_seed(1234)
    sd = U.init_state_dict(CIFAR)
    st = train_ref.TrainState(sd, CIFAR, lr=2e-4, warmup=5000)
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-large")
    g = torch.Generator().manual_seed(1234)
    B = 16
    x = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 3, 32, 32, generator=g)
    model, phys = host_cpu()
    ncpu = os.cpu_count() or 1
    cands = sorted({max(1, min(n, ncpu)) for n in (16, 32, 64)} | {min(phys, ncpu)})      # the last point = every physical core
    threads_was = torch.get_num_threads()
    t_begin = time.perf_counter()
    sweep = {}
    torch.set_num_threads(cands[0])
    st.step(T, x, t, noise)                              # lazy initialisation (oneDNN primitive caches) outside every measurement
    for n in cands:
        torch.set_num_threads(n)
        st.step(T, x, t, noise)
        s0 = time.perf_counter()
        st.step(T, x, t, noise)
        sweep[n] = time.perf_counter() - s0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n_steps, t0 = 0, time.perf_counter()
    while n_steps < 5 and (time.perf_counter() - t_begin < seconds_budget or n_steps == 0):
        st.step(T, x, t, noise)
        n_steps += 1
    dt = time.perf_counter() - t0
    # eval forward (what one sampling step costs; the reference's sampler is 1000 of these per sample, diffusion.py:160-174)
    with torch.no_grad():
        U.unet_forward(sd, CIFAR, x, t, training=False)
        f0, n_fwd = time.perf_counter(), 0
        while n_fwd < 5 and (time.perf_counter() - f0 < 6.0 or n_fwd == 0):
            U.unet_forward(sd, CIFAR, x, t, training=False)
            n_fwd += 1
        fdt = (time.perf_counter() - f0) / n_fwd
    # 256 x 256 leg (north star: "32x32 and 256x256 ... next to the reference's CPU path"): the CelebA-HQ net's eval forward at B = 1,
    # bounded to ~8 s (one warm-up + up to two timed forwards); a training step there is ~3x this, a 1000-step chain 1000x
    hq = None
    try:
        torch.manual_seed(4321)
        sd_hq = U.init_state_dict(CELEBAHQ)
        xh, th_ = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1, torch.randint(0, 1000, (1,), generator=g)
        with torch.no_grad():
            U.unet_forward(sd_hq, CELEBAHQ, xh, th_, training=False)
            h0, n_hq = time.perf_counter(), 0
            while n_hq < 2 and (time.perf_counter() - h0 < 6.0 or n_hq == 0):
                U.unet_forward(sd_hq, CELEBAHQ, xh, th_, training=False)
                n_hq += 1
            hdt = (time.perf_counter() - h0) / n_hq
        hq = {"forward_imgs_per_s": round(1.0 / hdt, 3), "samples_per_s_1000_steps_extrapolated": round(1.0 / hdt / 1000, 6),
              "train_imgs_per_s_estimated": round(1.0 / (3.0 * hdt), 3), "threads": best,
              "sample": f"oracle CelebA-HQ UNet (113.7M parameters) fp32 eval forward at 256x256, B=1, x{n_hq} after one warm-up; "
                        "training estimated as forward / 3 (not run: one step is ~3 forwards)"}
        del sd_hq
    except MemoryError:
        pass
    torch.set_num_threads(threads_was)
    return {"value": round(B * n_steps / dt, 3), "unit": "imgs/s", "cores": best, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus": ncpu,
            "thread_sweep_s_per_step": {str(k): round(v, 3) for k, v in sweep.items()},
            "forward_imgs_per_s": round(B / fdt, 2), "samples_per_s_1000_steps_extrapolated": round(B / fdt / 1000, 5),
            "all_cores": {"threads": cands[-1], "imgs_per_s": round(B / sweep[cands[-1]], 2)} if cands[-1] in sweep else None,
            "celebahq_256x256": hq,
            "sample": f"oracle (CPU restatement of the reference) CIFAR UNet fp32, B={B}: training step x{n_steps} and eval forward x{n_fwd} "
                      f"after warm-up, {best} torch threads (best of a {sorted(sweep)} sweep); sampling rate = forward rate / 1000 steps"}


def make_trainer(ddpm_torch, cfg, dev, dtype, shape, var_type, distributed=False, rank=0, native=True, local=0, lr=2e-4):
    model = ddpm_torch.UNet(**cfg).to(dev).set_compute_dtype(dtype)
    net = model
    if distributed:
        if native:
            model.set_process_group()
        else:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local])
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", var_type, "mse")
    opt = torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: min((s + 1) / 5000, 1.0))
    tr = ddpm_torch.Trainer(net, opt, dif, epochs=1, trainloader=None, sampler=object() if distributed else None, scheduler=sched,
                            use_ema=True, grad_norm=1.0, shape=shape, device=dev, distributed=distributed, rank=rank)
    return model, net, dif, tr


def settle(tr, x0, first_step, limit=40):
    """Untimed steps until every direct-step object of the trainer has made its eager-vs-replay decision (auto mode probes 4 + 4
    steps and captures the graph in between): nothing of that may fall inside a timed region, whatever --warmup is.
    Returns the number of extra steps taken."""
    extra = 0
    if first_step == 1:                          # --warmup 0: the direct-step objects do not exist before the first step
        tr.step(x0, global_steps=1)
        extra = 1
    while extra < limit and tr._direct and any(not d.settled() for d in tr._direct.values()):
        tr.step(x0, global_steps=first_step + extra)
        extra += 1
    return extra


def timed_steps(tr, x0, steps, warmup, sync, blocks=1):
    """`blocks` consecutive timed regions of exactly `steps` steps each, every one bracketed by sync() (barrier + device sync) on both
    sides.  Returns the list of block times in seconds (the reported figure is the median block)."""
    for i in range(warmup):
        tr.step(x0, global_steps=i + 1)
    warmup += settle(tr, x0, warmup + 1)
    tr.current_stats
    sync()
    before = [(d.choice, id(d.graph), id(d.plan), d.captures) for d in tr._direct.values()]
    els = []
    for blk in range(blocks):
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(x0, global_steps=warmup + blk * steps + i + 1)
        tr.current_stats                 # every step's loss read-back is collected INSIDE the timed region (the last one is still pending)
        sync()
        els.append(time.perf_counter() - t0)
    after = [(d.choice, id(d.graph), id(d.plan), d.captures) for d in tr._direct.values()]
    assert before == after, f"the step changed form inside the timed region: {before} -> {after}"
    return els


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sample-steps", type=int, default=1000, help="length of the timed p_sample chain (1000 = the real thing; 0 = skip sampling and the other configs)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / CelebA / CelebA-HQ lines")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started bare (`python bench.py --gpus N`): become the launcher — one rank per GPU under torch.distributed.run, rendezvous
        # on the loopback address (the reference self-spawns too: train.py:286-301).  The driver's own torchrun command line sets
        # WORLD_SIZE and lands in the branch below.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would be mislabelled; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or run bench.py bare")
    distributed = world > 1
    if os.environ.get("BENCH_LAUNCH_PROBE"):
        # test hook (tests/test_bench_contract.py, no GPU): prove that N ranks were started and can talk, then stop before any GPU work
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if distributed:
            dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
        seen = torch.ones(1)
        if distributed:
            dist.all_reduce(seen)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_probe": True, "n_gpus": world, "ranks_in_collective": int(seen)}))
        return
    # The JSON line must be the only thing on stdout.  RCCL prints a version banner ("RCCL version : ... / Librccl path : ...") through C
    # stdio when a communicator comes up; on a pipe or a file that buffer is flushed at process exit, i.e. AFTER Python's own line — seen on
    # the one-rank RCCL run (profiles/r05h_bench_ddp1.json was followed by five banner lines), and with N ranks every rank would add its own.
    # So the process-level stdout (fd 1) of every rank is pointed at stderr from here on and Python keeps the original for its one print.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks_seen = 1
    # BENCH_DDP=native on ONE GPU: a world-size-1 RCCL communicator, so that the exchange path (chunked all-reduces issued from inside the
    # backward, loss reduce, barrier) runs and config.dp can be read before a node is available
    self_ddp = world == 1 and os.environ.get("BENCH_DDP") == "native"
    if self_ddp:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port1 = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port1}", world_size=1, rank=0)
        distributed = True
    elif distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)     # "nccl" is RCCL on ROCm
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                                                                 # every rank present on the RCCL communicator
        ranks_seen = int(probe)
        assert ranks_seen == world, f"RCCL communicator has {ranks_seen} ranks, expected {world}"

    import ddim
    import ddpm_torch
    import ddpm_torch.models.unet as unet_mod
    import ddpm_torch.utils.train as train_mod
    from ddpm_torch import _ops
    ddpm_torch.seed_all(1234)
    # Data parallelism: the engine's own reducer (parameters broadcast from rank 0; gradient staging buffer all-reduced over
    # RCCL in chunks from inside the hand-written backward, overlapped with the rest of it).  BENCH_DDP=torch falls back to
    # the reference's DistributedDataParallel wrapper (train.py:110), which also works but only starts reducing after backward.
    native = os.environ.get("BENCH_DDP", "native") != "torch"
    model, net, dif, tr = make_trainer(ddpm_torch, CIFAR, dev, args.dtype, (3, 32, 32), "fixed-large", distributed, rank, native, local)
    g = torch.Generator().manual_seed(1234 + rank)
    x0 = (torch.rand(B_PER_GPU, 3, 32, 32, generator=g) * 2 - 1).to(dev)            # resident before timing

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    net.train()
    # 5 consecutive blocks of exactly --steps steps, each bracketed by barrier + synchronize; per block the MAX over ranks; the line
    # reports the MEDIAN block (one 0.2-s block cannot resolve a 2 % change: box jitter is of that order)
    n_blocks = max(1, int(os.environ.get("BENCH_BLOCKS", "5")))
    block_s = timed_steps(tr, x0, args.steps, args.warmup, sync, n_blocks)
    if distributed:
        te = torch.tensor(block_s, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        block_s = [float(v) for v in te]
    elapsed = median(block_s)
    ms_per_step = elapsed / args.steps * 1e3
    imgs_per_s = B_PER_GPU * world * args.steps / elapsed
    loss = tr.current_stats["loss"]
    kinds = {d.last_kind for d in tr._direct.values()}
    form = "graph" if "graph" in kinds else ("plan" if "plan" in kinds else "eager")
    step_mode = "direct step, " + {"graph": "hipGraph replay", "plan": "launch plan (ddpm_plan_run: the step's C-ABI calls re-issued from C on both streams)",
                                   "eager": "eager launches"}[form] + (" (chosen by measurement)" if any(d.choice for d in tr._direct.values()) else "") \
        if tr._direct else "autograd step"
    # what the auto-probe measured for each form of the step on THIS box (ms per step, 4 pipelined steps each, before the timed region)
    step_probe = {}
    for d in tr._direct.values():
        step_probe = {k + "_ms": round(v * 1e3, 3) for k, v in d.times.items()}
        if d.plan is not None:
            step_probe["plan_launches"] = d.plan.launches

    # ---- roofline of the dominant kernel: per-launch HIP events (recorded on the stream each kernel is launched on) over
    # one more training step.  The product runs the weight-gradient kernels on a side stream NEXT to the critical path, so
    # a launch's duration includes the slowdown from sharing the chip; `isolated` repeats the measurement with that
    # overlap switched off (every kernel alone on the GPU), which is what says how good each kernel is by itself.
    peak = PEAK[args.dtype]

    # The eager step is host-bound (~7 ms of launch work for ~9 ms of GPU work): with the GPU keeping up with the host, a kernel's start event
    # is stamped when the queue is EMPTY and its launch arrives microseconds later — the bracket then measures the host.  So the whole
    # step is enqueued behind a gate (a spin kernel of ~20 ms on the main stream, which the side stream's first wait depends on too) and
    # runs back to back once the gate opens, exactly as the launches of the pipelined product steps do.
    torch.cuda.synchronize()
    t_g = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
    gate_cycles = int(20_000_000 * 20e-3 / max(time.perf_counter() - t_g, 1e-4))

    def profile_step(trainer, x, step_no):
        graph_was, train_mod._TRAIN_GRAPH = train_mod._TRAIN_GRAPH, False       # per-launch events need the eager form of the step
        _ops.PROFILE = []
        eng = model.engine()
        eng.dp_trace = [] if eng.pg is not None else None
        torch.cuda.synchronize()
        if os.environ.get("BENCH_NO_GATE") is None:
            torch.cuda._sleep(gate_cycles)
        trainer.step(x, global_steps=step_no)
        torch.cuda.synchronize()
        prof, _ops.PROFILE = _ops.PROFILE, None
        dp_trace, eng.dp_trace = eng.dp_trace, None
        if dp_trace:
            profile_step.dp = dp_trace
        train_mod._TRAIN_GRAPH = graph_was
        agg, shapes, hbm = {}, {}, {}
        for kind, flops, a, b, shape, variant in prof:
            dt_s = a.elapsed_time(b) * 1e-3
            if kind.startswith("hbm_"):                       # GroupNorm launches: `flops` holds algorithmic BYTES
                e = hbm.setdefault(HBM_KIND[kind], [0, 0.0, 0.0])
                e[0] += 1; e[1] += flops; e[2] += dt_s
                e2 = hbm.setdefault(HBM_KIND[kind] + " | " + shape, [0, 0.0, 0.0])
                e2[0] += 1; e2[1] += flops; e2[2] += dt_s
                continue
            name = VARIANT.get(variant, "other") + (" wgrad (both operands k-strided)" if kind == "gemm_tt" else "")
            if variant in (7, 9):
                # the 1x1 kernels are bandwidth kernels (~190 FLOP/B): priced against HBM as well.  Algorithmic bytes: conv = x + y + w
                # (M x K, M x N, N x K bf16); weight gradient = dy + x read once (K pixels x (M + N) channels, bf16) + the fp32 result
                mm = re.search(r"M=(\d+) N=(\d+) K=(\d+)", shape)
                Mv, Nv, Kv = (int(v) for v in mm.groups())
                nbytes = 2.0 * (Mv * Kv + Mv * Nv + Nv * Kv) if variant == 7 else 2.0 * Kv * (Mv + Nv) + 4.0 * Mv * Nv
                hk = HBM_KIND["hbm_pw" if variant == 7 else "hbm_wg1"]
                for key in (hk, hk + " | " + shape):
                    e = hbm.setdefault(key, [0, 0.0, 0.0])
                    e[0] += 1; e[1] += nbytes; e[2] += dt_s
            e = agg.setdefault(name, [0, 0.0, 0.0])
            e[0] += 1; e[1] += flops; e[2] += dt_s
            e2 = shapes.setdefault(f"{name} | {kind} {shape}", [0, 0.0, 0.0])
            e2[0] += 1; e2[1] += flops; e2[2] += dt_s
        return agg, shapes, hbm

    def table(agg, pk):
        return {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2] * 1e3, 3), "tflops": round(v[1] / v[2] / 1e12, 1),
                    "avg_launch_us": round(v[2] / v[0] * 1e6, 2), "frac": round(v[1] / v[2] / 1e12 / pk, 4)}
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}

    def total(agg, pk):
        f, t = sum(v[1] for v in agg.values()), sum(v[2] for v in agg.values())
        return {"tflops": round(f / t / 1e12, 1), "ms": round(t * 1e3, 3), "frac": round(f / t / 1e12 / pk, 4)}

    agg, shapes, hbm = profile_step(tr, x0, args.warmup + n_blocks * args.steps + 1)
    side_was = unet_mod._SIDE_STREAM
    unet_mod._SIDE_STREAM = False
    agg_iso, shapes_iso, hbm_iso = profile_step(tr, x0, args.warmup + n_blocks * args.steps + 2)
    unet_mod._SIDE_STREAM = side_was
    # (every rank ran the two extra steps above: with N > 1 they contain the gradient all-reduce and the loss reduce)
    # What the matrix pipe SUSTAINS on this box: an MFMA-only loop on register-resident random bf16 operands, ~0.3 s (the chip clocks to
    # its power budget; 2.5 PFLOP/s is reached on zero operands only — profiles/r04_probe_mfma_power.txt).  Context for `frac`, not a target.
    practical = None
    if args.dtype == "bf16" and rank == 0:
        from ddpm_torch import _hip as hip_
        sink = torch.zeros(256, device=dev)
        it_p = 200000
        st_p = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            hip_.call("ddpm_mfma_probe", sink.data_ptr(), it_p, 0, st_p)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(6):
            hip_.call("ddpm_mfma_probe", sink.data_ptr(), it_p, 0, st_p)
        eb.record(); torch.cuda.synchronize()
        practical = {"tflops": round(6 * it_p * 16 * 32768.0 * 1024 / (ea.elapsed_time(eb) * 1e-3) / 1e12, 1),
                     "what": "ddpm_mfma_probe: MFMA-only loop (v_mfma_f32_32x32x16_bf16, 8 accumulators per wave, one wave per SIMD, all 256 CUs) on random bf16 "
                             "operands held in registers, sustained ~0.3 s after warm-up on this box; zero operands reach the nominal peak"}
    out = None
    if rank == 0:
        if os.environ.get("BENCH_SHAPES"):
            with open(os.environ["BENCH_SHAPES"], "w") as f:
                for title, tab in (("in the product step (two streams)", shapes), ("isolated (one stream)", shapes_iso)):
                    f.write(f"# {title}\n")
                    for k, v in sorted(tab.items(), key=lambda kv: -kv[1][2]):
                        f.write(f"{v[2] * 1e3:8.3f} ms  n={v[0]:3d}  {v[2] / v[0] * 1e6:7.1f} us  {v[1] / v[2] / 1e12:7.1f} TF  {k}\n")
        # dominant kernel = the one with the most GPU time per step when every kernel has the chip to itself (the isolated pass:
        # stable from run to run, and the order rocprofv3's kernel-trace of the product step gives); its `achieved` below is the
        # in-product figure.  The event-bracketed durations of the side stream's weight-gradient kernels include queueing behind
        # main-stream workgroups, which is why the runner-up is listed with them.
        # (GPU time = launch duration x the share of the chip the launch occupies: the 3x3 weight-gradient kernel is launched on HALF the CUs
        #  by design — csrc/wgrad.hip — so that the main stream keeps the other half; its duration doubles, its CU time does not)
        share = {k: (WGRAD3_CU_SHARE if k.startswith("wgrad3x3") else 1.0) for k in agg}
        # headline kernel = the one with the most summed launch DURATION in the product step (what rocprofv3's kernel trace ranks by);
        # the pick by duration x CU share (a launch on half the chip counts half) is printed as `dominant_by_cu_time`
        dom_name = max(agg.items(), key=lambda kv: kv[1][2])[0]
        dom = agg[dom_name]
        achieved = dom[1] / dom[2] / 1e12
        ru_name, ru = max(((k, v) for k, v in agg.items() if k != dom_name), key=lambda kv: kv[1][2])
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
        tpath = tfiles[-1] if tfiles else ""                       # the latest committed counter passes (scripts/gpu_profile.sh)
        if tpath:
            recs = json.load(open(tpath)).get("kernels", [])
            exact = [r for r in recs if dom_name.split(" ")[0] in r["kernel"]]          # same template instance when the name carries it
            for rec in (exact or recs):
                if traffic is None and rec["match"] in dom_name:
                    traffic = {"hbm_bytes_per_launch": rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"],
                               "fetch_bytes_per_launch": rec["fetch_bytes_per_launch"], "write_bytes_per_launch": rec["write_bytes_per_launch"],
                               "source": rec.get("source", "profiles/" + os.path.basename(tpath) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not collected in this run; FETCH_SIZE x2 per the gfx950 guide)")}
        by_dur = max(agg.items(), key=lambda kv: kv[1][2])[0]
        by_cu = max(agg.items(), key=lambda kv: kv[1][2] * share.get(kv[0], 1.0))[0]

        def hbm_table(tab):
            out_t = {}
            for k, v in sorted(tab.items(), key=lambda kv: -kv[1][2]):
                if " | " in k and len(out_t) >= 16:
                    continue
                gbs = v[1] / v[2] / 1e9
                rec = {"launches": v[0], "algorithmic_mb": round(v[1] / 1e6, 1), "ms": round(v[2] * 1e3, 3), "gb_per_s": round(gbs, 1),
                       "frac_of_achievable_6300": round(gbs / HBM_ACHIEVABLE, 4), "frac_of_spec_8000": round(gbs / 8000.0, 4)}
                out_t[k] = rec
            return out_t
        hbm_counter = None
        if tpath:
            recs = json.load(open(tpath)).get("kernels", [])
            # (HBM bytes per launch from the committed FETCH_SIZE x 2 / WRITE_SIZE passes, per kernel instantiation; compare with
            #  algorithmic_mb / launches of the shapes that instantiation serves)
            hbm_counter = {r["kernel"].replace("void ", "")[:44]: {"fetch_mb": round(r["fetch_bytes_per_launch"] / 1e6, 1), "write_mb": round(r["write_bytes_per_launch"] / 1e6, 1)}
                           for r in recs if r.get("match", "").startswith("gn_")}
            hbm_counter["source"] = "profiles/" + os.path.basename(tpath)
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "dominant_by_duration": {"kernel": by_dur, "ms_in_step": round(agg[by_dur][2] * 1e3, 3),
                                             "frac": round(agg[by_dur][1] / agg[by_dur][2] / 1e12 / peak, 4)},
                    "dominant_by_cu_time": {"kernel": by_cu, "ms_in_step_x_cu_share": round(agg[by_cu][2] * share.get(by_cu, 1.0) * 1e3, 3),
                                            "frac": round(agg[by_cu][1] / agg[by_cu][2] / 1e12 / peak, 4)},
                    "practical_peak": practical,
                    "frac_of_practical_peak": round(achieved / practical["tflops"], 4) if practical else None,
                    "hbm_kernels": {"bound": "hbm", "peak_gb_per_s": 8000.0, "achievable_gb_per_s": HBM_ACHIEVABLE,
                                    "note": "algorithmic bytes (SURVEY 8d: forward x + y; backward x + dy + dx [+ the residual gradient it adds]) / event-bracketed duration",
                                    "in_step": hbm_table(hbm), "isolated": hbm_table(hbm_iso), "counters": hbm_counter},
                    "launches_per_step": dom[0], "avg_launch_us": round(dom[2] / dom[0] * 1e6, 2),
                    **({"cu_share": share[dom_name], "frac_of_occupied_cus": round(achieved / (peak * share[dom_name]), 4)} if share.get(dom_name, 1.0) != 1.0 else {}),
                    "note": "durations as they occur in the product step (two HIP streams share the GPU); `isolated` = same step, one stream; "
                            "kernel = most summed launch duration in the product step (frac is priced against the WHOLE chip even for a launch that "
                            "takes half the CUs by design: cu_share / frac_of_occupied_cus say so)",
                    "runner_up": {"kernel": ru_name, "achieved": round(ru[1] / ru[2] / 1e12, 1), "frac": round(ru[1] / ru[2] / 1e12 / peak, 4),
                                  "launches_per_step": ru[0], "avg_launch_us": round(ru[2] / ru[0] * 1e6, 2),
                                  **({"cu_share": share[ru_name], "frac_of_occupied_cus": round(ru[1] / ru[2] / 1e12 / (peak * share[ru_name]), 4),
                                      "note": "launched on half the CUs by design (the main stream keeps the other half): frac_of_occupied_cus prices it against the CUs it holds"}
                                     if share.get(ru_name, 1.0) != 1.0 else {})},
                    "all_mfma_kernels": total(agg, peak), "per_kernel": table(agg, peak),
                    "isolated": {"all_mfma_kernels": total(agg_iso, peak), "per_kernel": table(agg_iso, peak)}}
        dp = None
        tr_dp = getattr(profile_step, "dp", None)
        if tr_dp:
            done = [e for w, _, e in tr_dp if w == "backward_compute_done"][-1]
            fin = [e for w, _, e in tr_dp if w == "exchange_done"][-1]
            ars = [(nb, e) for w, nb, e in tr_dp if w == "all_reduce"]
            ars = ars[-(len(model.engine().chunks) + 1):]                 # the last profiled backward (the isolated pass)
            dp = {"collective": "RCCL all-reduce (sum) of the packed fp32 gradient staging buffer, in chunks issued from inside the backward; 1/world applied in the unpack kernel",
                  "world": world, "chunks": len(ars), "bytes_per_chunk": [nb for nb, _ in ars], "total_mb": round(sum(nb for nb, _ in ars) / 1e6, 1),
                  "issued_ms_before_end_of_backward": [round(e.elapsed_time(done), 3) for _, e in ars],
                  "exposed_wait_ms": round(done.elapsed_time(fin), 3),
                  "chunk_target": os.environ.get("DDPM_DP_CHUNK_MB", "conv-weight region / 6 (~24 MB)") ,
                  "note": "timestamps on the compute stream of the profiled eager step (side stream off); the last chunk is the small-tensor tail. "
                          "On one GPU the communicator has a single rank: the times show WHEN the exchanges are issued, not what xGMI does with them"}
        out = {"metric": "training imgs/s/GPU + 1000-step DDPM samples/s, CIFAR-10 UNet @1/2/4/8 MI355X",
               "value": round(imgs_per_s, 2), "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "ms_per_step_blocks": [round(v / args.steps * 1e3, 3) for v in block_s],
               "timing": f"median of {n_blocks} consecutive blocks of {args.steps} steps, each bracketed by barrier + device sync (max over ranks per block)",
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "configs/cifar10.json UNet (35.7M params), full Trainer.step, B=128 per GPU, 32x32, T=1000, dropout 0.1, Adam+clip+EMA",
                          "global_batch": B_PER_GPU * world, "rccl_ranks": ranks_seen, "dp": dp,
                          "parallelism": f"dp{world}" + ("" if world == 1 else (" (native chunked RCCL all-reduce inside backward)" if native else " (torch DDP)")),
                          "step_execution": step_mode, "step_probe": step_probe,
                          "hbm_gb": {"reserved": round(torch.cuda.memory_reserved() / 2 ** 30, 2), "allocated_peak": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                                     "of": 288, "note": "whole bench process at the end of the run (training state + the step form's private pool + the sampler's graphs + the other-configs legs)"},
                          "imgs_per_s_per_gpu": round(imgs_per_s / world, 2),
                          "train_model_tflops_per_gpu": round(imgs_per_s / world * 3 * FWD_GFLOP["cifar"] / 1e3, 1),
                          "train_model_frac_of_peak": round(imgs_per_s / world * 3 * FWD_GFLOP["cifar"] / 1e3 / peak, 4), "final_loss": round(loss, 4)},
               "roofline": roofline}

    # ---- sampling + the other BASELINE configurations (rank 0 only: independent work, no collectives)
    S = args.sample_steps
    if rank == 0 and S > 0:
        net.eval()
        # the reference's p_sample (EMA weights are what generate.py samples with; same cost), B=128, eval mode, fixed-large,
        # seed 131071.  --sample-steps 1000 (default) runs the real 1000-step chain end to end; a smaller S times an S-step
        # chain (identical per-step work) and scales to 1000 steps.
        def chain(process, mdl, shape, seed, steps):
            process.prepare_sampler(mdl, shape, dev)          # capture of the step graph for this shape (a serving process does it once)
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            xs = process.p_sample(mdl, shape=shape, device=dev, seed=seed)
            torch.cuda.synchronize()
            el = time.perf_counter() - s0
            assert xs.shape == shape and bool(torch.isfinite(xs).all())
            return el

        sdif = dif if S == 1000 else ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, S), "eps", "fixed-large", "mse")
        s_el = chain(sdif, model, (B_PER_GPU, 3, 32, 32), 131071, S)
        samp = {"batch": B_PER_GPU, "steps_timed": S, "seconds": round(s_el, 3), "ms_per_step": round(s_el / S * 1e3, 3),
                "samples_per_s_1000_steps": round(B_PER_GPU / (s_el / S * 1000), 4),
                "model_tflops": round(B_PER_GPU * FWD_GFLOP["cifar"] / (s_el / S) / 1e3, 1),
                "frac_of_peak": round(B_PER_GPU * FWD_GFLOP["cifar"] / (s_el / S) / 1e3 / peak, 4),
                "mode": "hipGraph replay of the captured step" if os.environ.get("DDPM_TORCH_AMD_GRAPH", "1") != "0" else "eager"}
        if S == 1000 and not os.environ.get("BENCH_NO_SWEEP"):
            # batch sweep (SURVEY.md §8d "M2: B=128 and a sweep"): 50-step chains (identical per-step work), scaled to 1000 steps
            swdif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 50), "eps", "fixed-large", "mse")
            sweep = {}
            for sb in (32, 256, 512, 2048):            # (2048: the largest power of two whose 384-channel tensors stay under the 2-GiB operand limit; 8 GiB of HBM)
                w_el = chain(swdif, model, (sb, 3, 32, 32), 4, 50) / 50
                sweep[str(sb)] = {"ms_per_step": round(w_el * 1e3, 3), "samples_per_s_1000_steps": round(sb / (w_el * 1000), 3)}
            samp["batch_sweep"] = sweep
        out["sampling"] = samp

        if world == 1 and not args.no_extras:
            extras = {}
            # (1) the exact-fp32 MFMA mode (the mode that meets the 1e-3 parity bar) on the same step and sampler
            if args.dtype == "bf16":
                m32, n32, d32, t32 = make_trainer(ddpm_torch, CIFAR, dev, "fp32", (3, 32, 32), "fixed-large")
                n32.train()
                el = timed_steps(t32, x0, 6, 2, torch.cuda.synchronize)[0]
                f32_train = B_PER_GPU * 6 / el
                n32.eval()
                d50 = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 50), "eps", "fixed-large", "mse")
                s32 = chain(d50, m32, (B_PER_GPU, 3, 32, 32), 131071, 50) / 50
                extras["cifar10_fp32_mode"] = {
                    "train_imgs_per_s": round(f32_train, 1), "train_ms_per_step": round(el / 6 * 1e3, 2),
                    "train_frac_of_fp32_peak": round(f32_train * 3 * FWD_GFLOP["cifar"] / 1e3 / PEAK["fp32"], 4),
                    "sampling_ms_per_step": round(s32 * 1e3, 3), "samples_per_s_1000_steps": round(B_PER_GPU / (s32 * 1000), 3),
                    "sampling_frac_of_fp32_peak": round(B_PER_GPU * FWD_GFLOP["cifar"] / s32 / 1e3 / PEAK["fp32"], 4),
                    "note": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), peak 157.3 TFLOP/s; forward error vs the fp32 oracle 4e-6 (bf16 mode: 1.7e-2)"}
                del m32, n32, t32
            # (2) BASELINE config 4: CelebA 64x64 UNet, DDIM 50 steps (linear sub-sequence, eta 0), B=128, seed 131071
            mc = ddpm_torch.UNet(**CELEBA).to(dev).set_compute_dtype(args.dtype).eval()
            dd = ddim.DDIM(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-small", "mse", eta=0.0,
                           subsequence=ddim.get_selection_schedule("linear", 50, 1000))
            c_el = chain(dd, mc, (B_PER_GPU, 3, 64, 64), 131071, 50)
            extras["celeba64_ddim50"] = {
                "batch": B_PER_GPU, "seconds_per_chain": round(c_el, 3), "samples_per_s": round(B_PER_GPU / c_el, 2), "ms_per_step": round(c_el / 50 * 1e3, 3),
                "model_tflops": round(B_PER_GPU * FWD_GFLOP["celeba"] / (c_el / 50) / 1e3, 1),
                "frac_of_peak": round(B_PER_GPU * FWD_GFLOP["celeba"] / (c_el / 50) / 1e3 / peak, 4), "dtype": args.dtype}
            del mc
            # (3) BASELINE config 5, per-GPU work: CelebA-HQ 256x256 UNet (113.7M params), B=2, full Trainer.step
            torch.cuda.empty_cache()
            mh, nh, dh, th = make_trainer(ddpm_torch, CELEBAHQ, dev, args.dtype, (3, 256, 256), "fixed-small", lr=2e-5)
            nh.train()
            xh = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)
            h_el = median(timed_steps(th, xh, 10, 3, torch.cuda.synchronize, 3))
            extras["celebahq256_train_b2"] = {
                "batch_per_gpu": 2, "imgs_per_s_per_gpu": round(2 * 10 / h_el, 2), "ms_per_step": round(h_el / 10 * 1e3, 2),
                "model_tflops": round(2 * 10 / h_el * 3 * FWD_GFLOP["celebahq"] / 1e3, 1),
                "frac_of_peak": round(2 * 10 / h_el * 3 * FWD_GFLOP["celebahq"] / 1e3 / peak, 4), "dtype": args.dtype,
                "final_loss": round(th.current_stats["loss"], 4)}
            # (4) 256 x 256 sampling (north star: "sampling throughput on 32x32 and 256x256 batches"): the same CelebA-HQ net, eval mode, B = 8,
            # ancestral steps through the captured graph; a 20-step chain is timed (identical per-step work) and scaled to 1000 steps
            nh.eval()
            d20 = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 20), "eps", "fixed-small", "mse")
            q_el = chain(d20, mh, (8, 3, 256, 256), 131071, 20) / 20
            extras["celebahq256_sampling_b8"] = {
                "batch": 8, "ms_per_step": round(q_el * 1e3, 2), "samples_per_s_1000_steps": round(8 / (q_el * 1000), 4),
                "model_tflops": round(8 * FWD_GFLOP["celebahq"] / q_el / 1e3, 1), "frac_of_peak": round(8 * FWD_GFLOP["celebahq"] / q_el / 1e3 / peak, 4),
                "dtype": args.dtype, "note": "20-step chain timed, scaled to 1000 steps"}
            del mh, nh, th
            out["other_configs"] = extras
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)


if __name__ == "__main__":
    main()
