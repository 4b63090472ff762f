"""Sampling CLI of the MI355X build — counterpart of the reference's ``generate.py`` (tqch/ddpm-torch, ``generate.py:25-178``):
loads a training checkpoint (EMA shadow when the run used EMA, ``module.`` prefixes of DDP runs stripped; a bare state dict
is accepted too), freezes the model and writes ``--total-size`` samples as PNG files, ``--batch-size`` per sampling chain
(ancestral DDPM or ``--use-ddim``).  One process per GPU when ``--num-gpus`` > 1 (independent chains, no exchange).

    python generate.py --dataset cifar10 --chkpt-path chkpts/cifar10/cifar10_2040.pt --use-ddim --total-size 50000
"""
import argparse
import json
import math
import os
import sys
import uuid
from concurrent.futures import ThreadPoolExecutor

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import ddim  # noqa: E402
import ddpm_torch  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    add = p.add_argument
    add("--config-path", type=str)
    add("--dataset", choices=sorted(ddpm_torch.DATASET_DICT), default="cifar10")
    add("--batch-size", default=128, type=int)
    add("--total-size", default=50000, type=int)
    add("--config-dir", default=os.path.join(HERE, "configs"), type=str)
    add("--chkpt-dir", default="./chkpts", type=str)
    add("--chkpt-path", default="", type=str)
    add("--save-dir", default="./images", type=str)
    add("--device", default="cuda:0", type=str)
    add("--use-ema", action="store_true")
    add("--use-ddim", action="store_true")
    add("--eta", default=0., type=float)
    add("--skip-schedule", default="linear", type=str)
    add("--subseq-size", default=50, type=int)
    add("--suffix", default="", type=str)
    add("--max-workers", default=8, type=int)
    add("--num-gpus", default=1, type=int)
    add("--compute", choices=["bf16", "fp32"], default="bf16")
    add("--seed", default=None, type=int, help="seeds x_T AND the per-step noise of every chain (default: nondeterministic, like the reference)")
    return p.parse_args(argv)


def weights_from(chkpt, prefer_ema):
    """The state dict to sample with: EMA shadow / model section of a training checkpoint, or the file itself."""
    try:
        sd = chkpt["ema"]["shadow"] if prefer_ema else chkpt["model"]
    except (KeyError, TypeError):
        print("Not a training checkpoint: loading it directly as model weights...")
        sd = chkpt
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def build(args, rank=0):
    path = args.config_path or os.path.join(args.config_dir, args.dataset + ".json")
    with open(path) as f:
        meta = json.load(f)
    dataset = meta.get("dataset", args.dataset)
    info = ddpm_torch.DATASET_INFO[dataset]
    shape = (info["channels"], info["resolution"][0], info["resolution"][0])
    dk = dict(meta["diffusion"])
    betas = ddpm_torch.get_beta_schedule(dk.pop("beta_schedule"), dk.pop("beta_start"), dk.pop("beta_end"), dk.pop("timesteps"))
    if args.use_ddim:
        dk["model_var_type"] = "fixed-small"
        sub = ddim.get_selection_schedule(args.skip_schedule, size=args.subseq_size, timesteps=len(betas))
        process = ddim.DDIM(betas, **dk, eta=args.eta, subsequence=sub)
    else:
        process = ddpm_torch.GaussianDiffusion(betas, **dk)
    device = torch.device(f"cuda:{rank}" if args.num_gpus > 1 else args.device)
    mk = dict(meta["model"])
    block = mk.pop("block_size", 1)
    mk["in_channels"] = mk.get("in_channels", info["channels"]) * block ** 2
    model = ddpm_torch.UNet(out_channels=info["channels"] * block ** 2, **mk).set_compute_dtype(args.compute)
    if block > 1:
        model = ddpm_torch.ModelWrapper(model, torch.nn.PixelUnshuffle(block), torch.nn.PixelShuffle(block))
    model.to(device)
    chkpt_path = args.chkpt_path or os.path.join(args.chkpt_dir, f"ddpm_{dataset}.pt")
    use_ema = meta.get("train", {}).get("use_ema", args.use_ema)
    model.load_state_dict(weights_from(torch.load(chkpt_path, map_location=device), use_ema))
    model.eval()
    for prm in model.parameters():
        prm.requires_grad_(False)
    exp = os.path.splitext(os.path.basename(path))[0]
    out_dir = os.path.join(args.save_dir, "eval", exp, os.path.splitext(os.path.basename(chkpt_path))[0] + args.suffix)
    return process, model, device, shape, out_dir


def to_uint8(x):
    """[-1, 1] float NCHW -> uint8 NHWC (generate.py:129)."""
    return (x * 127.5 + 127.5).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()


def write_png(arr, out_dir):
    from PIL import Image
    Image.fromarray(arr.squeeze(-1) if arr.shape[-1] == 1 else arr).save(os.path.join(out_dir, f"{uuid.uuid4()}.png"))


def generate(rank, args):
    process, model, device, shape, out_dir = build(args, rank)
    os.makedirs(out_dir, exist_ok=True)
    world = max(args.num_gpus, 1)
    mine = args.total_size // world + (1 if rank < args.total_size % world else 0)
    done = chain = 0
    with ThreadPoolExecutor(max_workers=args.max_workers) as pool:
        while done < mine:
            n = min(args.batch_size, mine - done)
            if args.seed is None:                    # the reference's behaviour (generate.py:126-128): x_T and every z from the default RNG
                x = process.p_sample(model, shape=(n,) + shape, device=device, noise=torch.randn((n,) + shape, device=device))
            else:
                # reproducible runs: ONE seeded generator per chain feeds x_T and then every per-step z (diffusion.py:164-171), so the
                # whole chain — not only its starting point — repeats; chains and ranks get distinct streams
                x = process.p_sample(model, shape=(n,) + shape, device=device, seed=args.seed + 1000003 * rank + chain)
            chain += 1
            list(pool.map(lambda a: write_png(a, out_dir), list(to_uint8(x))))
            done += n
            if rank == 0:
                print(f"\r{done}/{mine}", end="", flush=True)
    if rank == 0:
        print()
    return out_dir


def main(argv=None):
    args = parse_args(argv)
    if args.num_gpus > 1:
        import torch.multiprocessing as mp
        mp.set_start_method("spawn")
        mp.spawn(generate, args=(args,), nprocs=args.num_gpus)
        return None
    return generate(0, args)


if __name__ == "__main__":
    main()
