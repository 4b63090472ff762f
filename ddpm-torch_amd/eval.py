"""Evaluation CLI of the MI355X build — counterpart of the reference's ``eval.py`` (tqch/ddpm-torch, ``eval.py:1-141``): scores a folder
of generated images against a dataset with FID and/or improved precision / recall and appends the result to ``metrics.txt`` next to
the folder.

    python eval.py --dataset cifar10 --sample-folder ./images/eval/cifar10/<run> --metrics fid pr

The arithmetic (streaming activation statistics, Fréchet distance, manifold radii, the precision / recall test) is this package's
``ddpm_torch.metrics``; the two pretrained feature networks the reference downloads cannot be fetched here, so they are FILES YOU
PROVIDE: ``--inception`` / ``DDPM_TORCH_AMD_INCEPTION`` (TorchScript: float images in [-1, 1] -> [N, 2048] pool3 activations) and
``--vgg`` / ``DDPM_TORCH_AMD_VGG`` (TorchScript: uint8 images -> [N, 4096] fc2 features).  Without them the metric is refused — no
number is produced from any other network.  Reference statistics: ``<precomputed-dir>/fid_stats_<...>.npz`` /
``pr_manifold_<dataset>.pt`` when present, otherwise computed from the dataset's raw images (``ddpm_torch.datasets``) and saved there.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import ddpm_torch  # noqa: E402
from ddpm_torch import datasets  # noqa: E402
from ddpm_torch.metrics import InceptionStatistics, ManifoldBuilder, calc_fd, calc_pr, get_precomputed, load_manifold  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    add = p.add_argument
    add("--root", default="~/datasets", type=str)
    add("--dataset", choices=sorted(ddpm_torch.DATASET_DICT), default="cifar10")
    add("--eval-batch-size", default=512, type=int)
    add("--eval-total-size", default=50000, type=int)
    add("--num-workers", default=4, type=int)
    add("--nhood-size", default=3, type=int)
    add("--row-batch-size", default=10000, type=int)
    add("--col-batch-size", default=10000, type=int)
    add("--device", default="cuda:0", type=str)
    add("--precomputed-dir", default="./precomputed", type=str)
    add("--metrics", nargs="+", default=["fid", "pr"], type=str)
    add("--seed", default=1234, type=int)
    add("--sample-folder", default="", type=str)
    add("--num-gpus", default=1, type=int)                  # accepted for command-line compatibility: the feature networks run on --device
    add("--inception", default=os.environ.get("DDPM_TORCH_AMD_INCEPTION", ""), type=str)
    add("--vgg", default=os.environ.get("DDPM_TORCH_AMD_VGG", ""), type=str)
    return p.parse_args(argv)


def _script(path, device, what):
    if not path:
        raise RuntimeError(f"{what}: no feature network given.  The pretrained weights cannot be downloaded here and nothing is substituted "
                           f"for them; pass a TorchScript file (see the module docstring).")
    net = torch.jit.load(os.path.expanduser(path), map_location=device).eval()
    return lambda x: net(x)


def main(argv=None, extractors=None):
    """``extractors`` = {"fid": callable, "pr": callable} replaces the TorchScript files (tests; embedding the CLI)."""
    args = parse_args(argv)
    extractors = dict(extractors or {})
    device = torch.device(args.device)
    root = os.path.expanduser(args.root)
    print(f"Dataset: {args.dataset}")
    folder = args.sample_folder.rstrip("/\\")
    ddpm_torch.seed_all(args.seed)

    samples = datasets.ImageFolder(folder)
    if len(samples) > args.eval_total_size:                  # the reference's subsample: numpy's seeded choice (eval.py:74-76)
        samples = Subset(samples, indices=torch.as_tensor(np.random.choice(len(samples), size=args.eval_total_size, replace=False)))
    loader = DataLoader(samples, batch_size=args.eval_batch_size, shuffle=False, num_workers=args.num_workers, drop_last=False)
    os.makedirs(args.precomputed_dir, exist_ok=True)

    def eval_fid():
        net = extractors.get("fid") or _script(args.inception, device, "fid")
        istats = InceptionStatistics(device=device, input_transform=lambda im: (im.float() - 127.5) / 127.5, feature_extractor=net)
        try:
            true_mean, true_var = get_precomputed(args.dataset, download_dir=args.precomputed_dir)
        except FileNotFoundError:
            print("Precomputed statistics cannot be loaded! Computing from raw data...")
            raw = datasets.get_dataloader(args.dataset, batch_size=args.eval_batch_size, split="all", val_size=0., root=root, drop_last=False,
                                          num_workers=args.num_workers, raw=True)[0]
            for x in raw:
                istats(x.to(device))
            true_mean, true_var = istats.get_statistics()
            np.savez(os.path.join(args.precomputed_dir, f"fid_stats_{args.dataset}.npz"), mu=true_mean, sigma=true_var)
        istats.reset()
        for x in loader:
            istats(x.to(device))
        gen_mean, gen_var = istats.get_statistics()
        return float(calc_fd(gen_mean, gen_var, true_mean, true_var))

    def eval_pr():
        net = extractors.get("pr") or _script(args.vgg, device, "pr")
        places = math.ceil(math.log(args.eval_total_size, 10))
        build = lambda data: ManifoldBuilder(                                                                      # noqa: E731
            data=data, extr_batch_size=args.eval_batch_size, max_sample_size=args.eval_total_size, row_batch_size=args.row_batch_size,
            col_batch_size=args.col_batch_size, nhood_size=args.nhood_size, num_workers=args.num_workers, device=device, extractor=net)
        path = os.path.join(args.precomputed_dir, f"pr_manifold_{args.dataset}.pt")
        if os.path.exists(path):
            true = load_manifold(path)
        else:
            builder = build(datasets._open(args.dataset, root, "all", raw=True))
            builder.save(path)
            true = builder.manifold
        gen = build(samples).manifold
        precision, recall = calc_pr(gen, true, row_batch_size=args.row_batch_size, col_batch_size=args.col_batch_size, device=device)
        return f"{float(precision):.{places}f}/{float(recall):.{places}f}"

    result = {"folder_name": os.path.basename(folder)}
    with open(os.path.join(os.path.dirname(folder), "metrics.txt"), "a") as f:
        for metric in args.metrics:
            fn = {"fid": eval_fid, "pr": eval_pr}.get(metric)
            if fn is None:
                print("Unsupported metric passed! Ignore.")
                continue
            result[metric] = fn()
            print(f"{metric.upper()}: {result[metric]}")
        f.write(str(result))
    return result


if __name__ == "__main__":
    main()
