"""Training CLI of the MI355X build — counterpart of the reference's ``train.py`` (tqch/ddpm-torch, ``train.py:16-305``):
same flags, same JSON-over-flags precedence (``get_param``), same checkpoint / hyper-parameter record layout.

    python train.py --dataset cifar10 --use-ema [--dry-run]                      # one GPU
    torchrun --nnodes 1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --dataset cifar10 --use-ema --distributed

Differences that follow from the platform, not from taste: there is no CPU path (the model lives on the GPU);
``--distributed`` uses the engine's own data parallelism (``UNet.set_process_group``: gradient all-reduce over RCCL issued
from inside the hand-written backward) unless ``--ddp-wrapper`` asks for ``DistributedDataParallel``; ``--compute``
chooses the arithmetic (bf16 throughput mode or exact-fp32 MFMA); images come from a tensor file or synthetic data
(``ddpm_torch/datasets.py``); ``--eval`` scores FID through ``ddpm_torch.metrics`` when a feature network is provisioned (``DDPM_TORCH_AMD_INCEPTION`` = TorchScript file, ``precomputed/fid_stats_<dataset>.npz``) and is refused otherwise.
"""
import argparse
import json
import os
import sys
import tempfile
from datetime import datetime

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import ddim  # noqa: E402
import ddpm_torch  # noqa: E402
from ddpm_torch import ConfigDict, get_param  # noqa: E402

TRAIN_KEYS = ("batch_size", "beta1", "beta2", "lr", "epochs", "grad_norm", "warmup", "chkpt_intv", "image_intv", "num_samples",
              "use_ema", "ema_decay")
DIFFUSION_KEYS = ("beta_schedule", "beta_start", "beta_end", "timesteps", "model_mean_type", "model_var_type", "loss_type")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    add = p.add_argument
    add("--config-path", type=str, help="JSON configuration (default: <config-dir>/<dataset>.json)")
    add("--exp-name", type=str)
    add("--dataset", choices=sorted(ddpm_torch.DATASET_DICT), default="cifar10")
    add("--root", default="~/datasets", type=str)
    add("--epochs", default=50, type=int)
    add("--lr", default=2e-4, type=float)
    add("--beta1", default=0.9, type=float)
    add("--beta2", default=0.999, type=float)
    add("--batch-size", default=128, type=int, help="GLOBAL batch (divided by the world size when --distributed)")
    add("--num-accum", default=1, type=int)
    add("--block-size", default=1, type=int, help="pixel-(un)shuffle factor around the UNet")
    add("--timesteps", default=1000, type=int)
    add("--beta-schedule", choices=["quad", "linear", "warmup10", "warmup50", "jsd"], default="linear")
    add("--beta-start", default=1e-4, type=float)
    add("--beta-end", default=0.02, type=float)
    add("--model-mean-type", choices=["mean", "x_0", "eps"], default="eps")
    add("--model-var-type", choices=["learned", "fixed-small", "fixed-large"], default="fixed-large")
    add("--loss-type", choices=["kl", "mse"], default="mse")
    add("--num-workers", default=4, type=int)
    add("--train-device", default="cuda:0", type=str)
    add("--eval-device", default="cuda:0", type=str)
    add("--image-dir", default="./images", type=str)
    add("--image-intv", default=10, type=int)
    add("--num-samples", default=64, type=int)
    add("--config-dir", default=os.path.join(HERE, "configs"), type=str)
    add("--chkpt-dir", default="./chkpts", type=str)
    add("--chkpt-name", default="", type=str)
    add("--chkpt-intv", default=120, type=int)
    add("--seed", default=1234, type=int)
    add("--resume", action="store_true")
    add("--chkpt-path", default="", type=str)
    add("--eval", action="store_true")
    add("--eval-total-size", default=50000, type=int)
    add("--eval-batch-size", default=256, type=int)
    add("--use-ema", action="store_true")
    add("--use-ddim", action="store_true")
    add("--skip-schedule", choices=["linear", "quadratic"], default="linear")
    add("--subseq-size", default=50, type=int)
    add("--ema-decay", default=0.9999, type=float)
    add("--distributed", action="store_true")
    add("--rigid-launch", action="store_true", help="spawn one process per GPU from here instead of torchrun / srun")
    add("--num-gpus", default=1, type=int)
    add("--dry-run", action="store_true", help="stop after the first parameter update; checkpoint and sample once")
    add("--compute", choices=["bf16", "fp32"], default="bf16", help="arithmetic of the UNet kernels (not in the reference)")
    add("--ddp-wrapper", action="store_true", help="wrap the model in DistributedDataParallel instead of the native reducer")
    return p.parse_args(argv)


def sections(args):
    """(meta config, experiment name, dataset, train / diffusion hyper-parameters): JSON values win over flags."""
    path = args.config_path or os.path.join(args.config_dir, args.dataset + ".json")
    with open(path) as f:
        meta = json.load(f)
    dataset = meta.get("dataset", args.dataset)
    train = ConfigDict(**{k: get_param(k, meta.get("train", {}), args) for k in TRAIN_KEYS})
    train.batch_size //= args.num_accum
    diffusion = ConfigDict(**{k: get_param(k, meta.get("diffusion", {}), args) for k in DIFFUSION_KEYS})
    return meta, os.path.splitext(os.path.basename(path))[0], dataset, train, diffusion


def rendezvous(args, rank, temp_dir):
    """Join the process group; returns (global rank, local rank).  torchrun / srun export the coordinates, --rigid-launch shares
    a file store between the processes it spawned (single node)."""
    assert dist.is_available() and torch.cuda.is_available(), "distributed training needs GPUs"
    env = os.environ
    if args.rigid_launch:
        assert temp_dir, "rigid launch needs the shared temporary directory"
        store = "file://" + os.path.join(os.path.abspath(temp_dir), ".torch_distributed_init")
        dist.init_process_group("nccl", init_method=store, rank=rank, world_size=args.num_gpus)        # "nccl" is RCCL on ROCm
        env["WORLD_SIZE"], env["LOCAL_RANK"] = str(args.num_gpus), str(rank)
        return rank, rank
    world = int(env.get("WORLD_SIZE", env.get("SLURM_NTASKS", "1")))
    rank = int(env.get("RANK", env.get("SLURM_PROCID", "0")))
    dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
    per_node = int(env.get("LOCAL_WORLD_SIZE", "0")) or int(env.get("SLURM_GPUS_ON_NODE", "0")) or torch.cuda.device_count()
    local = int(env.get("LOCAL_RANK", "0")) or rank % per_node
    args.num_gpus = world
    env.setdefault("WORLD_SIZE", str(world))
    return rank, local


def run(rank=0, args=None, temp_dir=""):
    meta, exp_name, dataset, tcfg, dcfg = sections(args)
    info = ddpm_torch.DATASET_INFO[dataset]
    in_channels, shape = info["channels"], (info["channels"],) + tuple(info["resolution"])
    seed = meta.get("seed", args.seed)
    ddpm_torch.seed_all(seed)

    betas = ddpm_torch.get_beta_schedule(dcfg.beta_schedule, beta_start=dcfg.beta_start, beta_end=dcfg.beta_end, timesteps=dcfg.timesteps)
    diffusion = ddpm_torch.GaussianDiffusion(betas=betas, **dcfg)

    mcfg = dict(meta["model"])
    block = mcfg.pop("block_size", args.block_size)
    out_channels = 2 * in_channels if dcfg.model_var_type == "learned" else in_channels
    mcfg.update(in_channels=in_channels * block ** 2, out_channels=out_channels * block ** 2)
    unet = ddpm_torch.UNet(**mcfg).set_compute_dtype(args.compute)
    net = unet if block == 1 else ddpm_torch.ModelWrapper(unet, torch.nn.PixelUnshuffle(block), torch.nn.PixelShuffle(block))

    device = torch.device(args.train_device)
    local = 0
    if args.distributed:
        rank, local = rendezvous(args, rank, temp_dir)
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        net.to(device)
        if args.ddp_wrapper or block != 1:
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local])
        else:
            model = unet.set_process_group()              # parameters broadcast from rank 0; gradients averaged inside the backward
    else:
        rank = 0
        model = net.to(device)
    leader = rank == 0

    def say(*a, **k):
        if leader:
            print(*a, **k)

    say(f"Dataset: {dataset}; effective batch {tcfg.batch_size} x {args.num_accum} accumulation step(s); compute {args.compute}")
    opt = torch.optim.Adam(model.parameters(), lr=tcfg.lr, betas=(tcfg.beta1, tcfg.beta2))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: min((t + 1) / tcfg.warmup, 1.0)) if tcfg.warmup > 0 else None
    loader, sampler = ddpm_torch.get_dataloader(dataset, batch_size=tcfg.batch_size, split="all" if dataset == "celeba" else "train", val_size=0.,
                                                random_seed=seed, root=os.path.expanduser(args.root), drop_last=True, pin_memory=True,
                                                num_workers=args.num_workers, distributed=args.distributed)
    if args.dry_run:
        say("This is a dry run.")
        args.chkpt_intv = tcfg.image_intv = 1
    chkpt_dir = os.path.join(args.chkpt_dir, exp_name)
    chkpt_path = os.path.join(chkpt_dir, args.chkpt_name or f"{exp_name}.pt")
    image_dir = os.path.join(args.image_dir, "train", exp_name)
    say(f"Checkpoints: {os.path.abspath(chkpt_path)} every {args.chkpt_intv} epoch(s); samples (x{tcfg.num_samples}): "
        f"{os.path.abspath(image_dir)} every {tcfg.image_intv} epoch(s)")
    if leader:
        os.makedirs(chkpt_dir, exist_ok=True)
        os.makedirs(image_dir, exist_ok=True)
        record = {"dataset": dataset, "seed": seed, "diffusion": dcfg, "model": dict(mcfg, block_size=block), "train": tcfg}
        with open(os.path.join(chkpt_dir, f"exp_{datetime.now():%Y-%m-%dT%H%M%S%f}.info"), "w") as f:
            json.dump(record, f, indent=2)

    trainer = ddpm_torch.Trainer(model=model, optimizer=opt, diffusion=diffusion, epochs=tcfg.epochs, trainloader=loader, sampler=sampler,
                                 scheduler=sched, num_accum=args.num_accum, use_ema=tcfg.use_ema, grad_norm=tcfg.grad_norm, shape=shape,
                                 device=device, chkpt_intv=args.chkpt_intv, image_intv=tcfg.image_intv, num_samples=tcfg.num_samples,
                                 ema_decay=args.ema_decay, rank=rank, distributed=args.distributed, dry_run=args.dry_run)
    if args.eval:
        sampler_proc = diffusion
        if args.use_ddim:
            sub = ddim.get_selection_schedule(args.skip_schedule, size=args.subseq_size, timesteps=dcfg.timesteps)
            sampler_proc = ddim.DDIM.from_ddpm(diffusion, eta=0., subsequence=sub)
        evaluator = ddpm_torch.Evaluator(dataset=dataset, diffusion=sampler_proc, eval_batch_size=args.eval_batch_size,
                                         eval_total_size=args.eval_total_size, device=torch.device(args.eval_device))
    else:
        evaluator = None
    if args.resume or args.distributed:                   # distributed runs always try to resume, like the reference
        try:
            trainer.load_checkpoint(args.chkpt_path or chkpt_path, map_location={"cuda:0": f"cuda:{local}"} if args.distributed else device)
            say(f"Resumed from epoch {trainer.start_epoch}.")
        except FileNotFoundError:
            say("Checkpoint file does not exist!\nStarting from scratch...")
    say("Training starts...", flush=True)
    trainer.train(evaluator, chkpt_path=chkpt_path, image_dir=image_dir)
    if args.distributed:
        dist.destroy_process_group()
    return trainer


def main(argv=None):
    args = parse_args(argv)
    if args.distributed and args.rigid_launch:
        import torch.multiprocessing as mp
        mp.set_start_method("spawn")
        with tempfile.TemporaryDirectory() as tmp:
            mp.spawn(run, args=(args, tmp), nprocs=args.num_gpus)
        return None
    return run(args=args)


if __name__ == "__main__":
    main()
