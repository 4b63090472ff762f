"""Headline benchmark (BASELINE.json): training imgs/s and 1000-step DDPM samples/s of the CIFAR-10 UNet on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one full reference training step (ddpm_torch/utils/train.py:148-170: forward, backward, global-norm clip,
Adam, LR schedule, EMA, loss reduce) on B=128 synthetic 32x32 images PER GPU (weak scaling), bf16 compute with fp32
master weights / gradients / optimizer state, dropout 0.1 active — configs/cifar10.json of the reference.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line; extra objects:
  roofline     — the dominant kernel (MFMA implicit-GEMM, operands K-contiguous: all conv forward/dgrad + linears):
                 algorithmic FLOPs of its launches in one training step / their summed HIP-event durations, vs 2.5 PFLOP/s
  sampling     — eval-mode ancestral sampling throughput (B=128) from timed p_sample steps, scaled to 1000 steps
  cpu_baseline — the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CIFAR = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
             apply_attn=[False, True, False, False], drop_rate=0.1)
FWD_GFLOP_PER_SAMPLE = 12.443713536      # SURVEY.md §8d (2*MAC over convs, linears, attention matmuls)
PEAK_BF16_TFLOPS = 2500.0                # dense MFMA peak, MI355X_MICROARCH.md
B_PER_GPU = 128


def cpu_baseline(seconds_budget=25.0):
    """Oracle training step (fp32, torch CPU ops) on a bounded sample: B=16, warm-up 1, then steps until ~budget."""
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
    st.step(T, x, t, noise)
    n, t0 = 0, time.perf_counter()
    while n < 8 and (time.perf_counter() - t0 < seconds_budget or n == 0):
        st.step(T, x, t, noise)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "imgs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (CPU restatement of the reference) training step, CIFAR UNet fp32, B={B}, {n} steps after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sample-steps", type=int, default=1000, help="length of the timed p_sample chain (1000 = the real thing)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)     # "nccl" is RCCL on ROCm

    import ddpm_torch
    from ddpm_torch import _ops
    ddpm_torch.seed_all(1234)
    model = ddpm_torch.UNet(**CIFAR).to(dev).set_compute_dtype(args.dtype)
    # Data parallelism: the engine's own reducer (parameters broadcast from rank 0; gradient staging buffer all-reduced over
    # RCCL in chunks from inside the hand-written backward, overlapped with the rest of it).  BENCH_DDP=torch falls back to
    # the reference's DistributedDataParallel wrapper (train.py:110), which also works but only starts reducing after backward.
    net = model
    if distributed:
        if os.environ.get("BENCH_DDP", "native") == "torch":
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local])
        else:
            model.set_process_group()
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: min((s + 1) / 5000, 1.0))
    tr = ddpm_torch.Trainer(net, opt, dif, epochs=1, trainloader=None, sampler=object() if distributed else None, scheduler=sched,
                            use_ema=True, grad_norm=1.0, shape=(3, 32, 32), device=dev, distributed=distributed, rank=rank)
    g = torch.Generator().manual_seed(1234 + rank)
    x0 = (torch.rand(B_PER_GPU, 3, 32, 32, generator=g) * 2 - 1).to(dev)            # resident before timing

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    net.train()
    for i in range(args.warmup):
        tr.step(x0, global_steps=i + 1)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.step(x0, global_steps=args.warmup + i + 1)
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        te = torch.tensor([elapsed], device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te)
    ms_per_step = elapsed / args.steps * 1e3
    imgs_per_s = B_PER_GPU * world * args.steps / elapsed
    loss = tr.current_stats["loss"]

    out = None
    # ---- roofline of the dominant kernel: per-launch HIP events (recorded on the stream each kernel is launched on) over
    # one more training step.  The product runs the weight-gradient kernels on a side stream NEXT to the critical path, so
    # a launch's duration includes the slowdown from sharing the chip; `isolated` repeats the measurement with that
    # overlap switched off (every kernel alone on the GPU), which is what says how good each kernel is by itself.
    import ddpm_torch.models.unet as unet_mod
    VARIANT = {1: "gemm_kernel<4 waves,128x128>", 2: "gemm_kernel<8 waves,128x128>", 3: "gemm_kernel<deep ring,128x128>",
               4: "gemm64_kernel<64x64>", 5: "conv3x3_halo_kernel<256px x 128>"}
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3

    import ddpm_torch.utils.train as train_mod

    def profile_step(step_no):
        graph_was, train_mod._TRAIN_GRAPH = train_mod._TRAIN_GRAPH, False       # per-launch events need the eager form of the step
        _ops.PROFILE = []
        tr.step(x0, global_steps=step_no)
        torch.cuda.synchronize()
        prof, _ops.PROFILE = _ops.PROFILE, None
        train_mod._TRAIN_GRAPH = graph_was
        agg, shapes = {}, {}
        for kind, flops, a, b, shape, variant in prof:
            dt_s = a.elapsed_time(b) * 1e-3
            name = VARIANT.get(variant, "other") + (" wgrad (both operands k-strided)" if kind == "gemm_tt" else "")
            e = agg.setdefault(name, [0, 0.0, 0.0])
            e[0] += 1; e[1] += flops; e[2] += dt_s
            e2 = shapes.setdefault(f"{name} | {kind} {shape}", [0, 0.0, 0.0])
            e2[0] += 1; e2[1] += flops; e2[2] += dt_s
        return agg, shapes

    def table(agg):
        return {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2] * 1e3, 3), "tflops": round(v[1] / v[2] / 1e12, 1),
                    "avg_launch_us": round(v[2] / v[0] * 1e6, 2), "frac": round(v[1] / v[2] / 1e12 / peak, 4)}
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}

    agg, shapes = profile_step(args.warmup + args.steps + 1)
    side_was = unet_mod._SIDE_STREAM
    unet_mod._SIDE_STREAM = False
    agg_iso, shapes_iso = profile_step(args.warmup + args.steps + 2)
    unet_mod._SIDE_STREAM = side_was
    # (every rank ran the two extra steps above: with N > 1 they contain the gradient all-reduce and the loss reduce)
    if rank == 0:
        if os.environ.get("BENCH_SHAPES"):
            with open(os.environ["BENCH_SHAPES"], "w") as f:
                for k, v in sorted(shapes_iso.items(), key=lambda kv: -kv[1][2]):
                    f.write(f"{v[2] * 1e3:8.3f} ms  n={v[0]:3d}  {v[1] / v[2] / 1e12:7.1f} TF  {k}\n")
        # dominant kernel = the one with the most GPU time in the step
        dom_name, dom = max(agg.items(), key=lambda kv: kv[1][2])
        achieved = dom[1] / dom[2] / 1e12
        tot_f, tot_t = sum(v[1] for v in agg.values()), sum(v[2] for v in agg.values())
        iso_f, iso_t = sum(v[1] for v in agg_iso.values()), sum(v[2] for v in agg_iso.values())
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": None,
                    "launches_per_step": dom[0], "avg_launch_us": round(dom[2] / dom[0] * 1e6, 2),
                    "note": "durations as they occur in the product step (two HIP streams share the GPU); `isolated` = same step, one stream",
                    "all_mfma_kernels": {"tflops": round(tot_f / tot_t / 1e12, 1), "ms": round(tot_t * 1e3, 3), "frac": round(tot_f / tot_t / 1e12 / peak, 4)},
                    "per_kernel": table(agg),
                    "isolated": {"all_mfma_kernels": {"tflops": round(iso_f / iso_t / 1e12, 1), "ms": round(iso_t * 1e3, 3), "frac": round(iso_f / iso_t / 1e12 / peak, 4)},
                                 "per_kernel": table(agg_iso)}}
        # ---- sampling: the reference's p_sample (EMA weights are what generate.py samples with; same cost), B=128,
        # eval mode, fixed-large, seed 131071.  --sample-steps 1000 (default) runs the real 1000-step chain end to end;
        # a smaller S times an S-step chain (identical per-step work) and scales to 1000 steps.
        net.eval()
        S = args.sample_steps
        if S <= 0:
            raise SystemExit(json.dumps({"train_only_ms_per_step": ms_per_step, "roofline": roofline}))
        sdif = dif if S == 1000 else ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, S), "eps", "fixed-large", "mse")
        sdif.p_sample(model, shape=(8, 3, 32, 32), device=dev, seed=1)          # warm-up (small batch, same kernels)
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        xs = sdif.p_sample(model, shape=(B_PER_GPU, 3, 32, 32), device=dev, seed=131071)
        torch.cuda.synchronize()
        s_el = time.perf_counter() - s0
        assert xs.shape == (B_PER_GPU, 3, 32, 32) and bool(torch.isfinite(xs).all())
        samp = {"batch": B_PER_GPU, "steps_timed": S, "seconds": round(s_el, 3), "ms_per_step": round(s_el / S * 1e3, 3),
                "samples_per_s_1000_steps": round(B_PER_GPU / (s_el / S * 1000), 4),
                "model_tflops": round(B_PER_GPU * FWD_GFLOP_PER_SAMPLE / (s_el / S) / 1e3, 1),
                "mode": "hipGraph replay of the captured step" if os.environ.get("DDPM_TORCH_AMD_GRAPH", "1") != "0" else "eager"}
        # batch sweep (SURVEY.md §8d "M2: B=128 and a sweep"): 50-step chains (identical per-step work), scaled to 1000 steps
        sweep = {}
        if S == 1000 and not os.environ.get("BENCH_NO_SWEEP"):
            swdif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 50), "eps", "fixed-large", "mse")
            for sb in (32, 256, 512):
                swdif.p_sample(model, shape=(sb, 3, 32, 32), device=dev, seed=3)      # capture / warm-up for this batch size
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                swdif.p_sample(model, shape=(sb, 3, 32, 32), device=dev, seed=4)
                torch.cuda.synchronize()
                w_el = (time.perf_counter() - w0) / 50
                sweep[str(sb)] = {"ms_per_step": round(w_el * 1e3, 3), "samples_per_s_1000_steps": round(sb / (w_el * 1000), 3)}
            samp["batch_sweep"] = sweep
        out = {"metric": "training imgs/s/GPU + 1000-step DDPM samples/s, CIFAR-10 UNet @1/2/4/8 MI355X",
               "value": round(imgs_per_s, 2), "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "configs/cifar10.json UNet (35.7M params), full Trainer.step, B=128 per GPU, 32x32, T=1000, dropout 0.1, Adam+clip+EMA",
                          "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}" + ("" if world == 1 else (" (torch DDP)" if os.environ.get("BENCH_DDP") == "torch" else " (native chunked RCCL all-reduce inside backward)")), "imgs_per_s_per_gpu": round(imgs_per_s / world, 2),
                          "train_model_tflops_per_gpu": round(imgs_per_s / world * 3 * FWD_GFLOP_PER_SAMPLE / 1e3, 1), "final_loss": round(loss, 4)},
               "roofline": roofline, "sampling": samp}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
