"""Worker for tests/test_ddp_gloo.py (one rank of a world_size-2 gloo job on CPU: the engine runs unmodified, its C-ABI
calls go to tests/abi_emulator.py — the HIP kernels need a GPU, the data-parallel LOGIC does not) and for
tests/test_multi_gpu.py (device kind "cuda": one rank per visible GPU over the "nccl" = RCCL backend, real kernels)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]

TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1, apply_attn=[False, True], drop_rate=0.0)
CIFAR = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2, apply_attn=[False, True, False, False], drop_rate=0.0)
# DDP_WORKER_CFG=cifar: the configs/cifar10.json geometry at B = 4 per rank (tests/test_multi_gpu.py) — the chunk plan, the packed gradient
# staging buffer and the hot kernels of BASELINE config 3 instead of the 8 x 8 toy
# DDP_WORKER_CFG=celebahq: the configs/celebahq.json geometry (113.7 M parameters, 454.7 MB of gradients per exchange) at 256 x 256, B = 1 per rank
HQ = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 1, 2, 2, 4, 4], num_res_blocks=2,
          apply_attn=[False, False, False, False, True, False], drop_rate=0.0)
BIG = os.environ.get("DDP_WORKER_CFG")
CFG, SHAPE = {"cifar": (CIFAR, (4, 3, 32, 32)), "celebahq": (HQ, (1, 3, 256, 256))}.get(BIG, (TINY, (2, 3, 8, 8)))


def install_emulator():
    from ddpm_torch import _hip
    from tests.abi_emulator import Emulator
    emu = Emulator(_hip.lib())
    _hip._invoke = lambda name, args: emu.call(name, *args)
    _hip.stream = lambda: 0
    _hip.require_cuda = lambda *a: None
    _hip.on_device = lambda t: True


def run(rank, world, port, mode, out_dir, kind="cpu"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    if kind == "cuda":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)            # "nccl" is RCCL on ROCm
    elif kind == "cuda_shared":
        # every rank on GPU 0, collectives over gloo (RCCL refuses two ranks on one device): the REAL kernels, streams and launch plan
        # under a world size > 1 on a one-GPU box — the exchange logic on device, not the xGMI transport (tests/test_ddp_one_gpu.py)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        install_emulator()
    import ddpm_torch
    from oracle import unet_ref as U
    torch.manual_seed(100 + rank)                        # different initial weights per rank: the broadcast must fix that
    model = ddpm_torch.UNet(**CFG)
    if rank == 0:
        torch.manual_seed(7)
        model.load_state_dict(U.randomize_state_dict(ddpm_torch.UNet(**CFG).state_dict(), 17))
    g = torch.Generator().manual_seed(1000 + rank)       # each rank sees its own shard
    x, gy, t = torch.randn(*SHAPE, generator=g), torch.randn(*SHAPE, generator=g), torch.randint(0, 1000, (SHAPE[0],), generator=g)
    model.to(dev)
    native = mode.startswith("native")
    if native:
        model.set_process_group()                         # broadcast + chunked all-reduce inside the hand-written backward
        net = model
    else:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index] if dev.type == "cuda" else None)     # the reference's wrapper (train.py:110) must keep working
    net.train()
    y = net(x.to(dev), t.to(dev))
    (y * gy.to(dev)).sum().backward()
    cpu = lambda d: {k: v.detach().cpu().clone() for k, v in d.items()}
    grads = cpu({k: p.grad for k, p in model.named_parameters()})
    torch.save(dict(grads=grads, sd=cpu(model.state_dict()), x=x, gy=gy, t=t), os.path.join(out_dir, f"{mode}_{rank}.pt"))
    if native:
        # full distributed Trainer.steps on top (loss reduce to rank 0 included).  "native": the direct step, captured into
        # graph segments on a GPU (the all-reduces run between the segments); "native_eager": same step without capture
        from ddpm_torch.utils import train as train_mod
        # force the captured / the eager / the launch-plan form (the default picks by measurement)
        train_mod._TRAIN_GRAPH = {"native_eager": False, "native_eager4": False, "native_plan": "plan"}.get(mode, True)
        dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        tr = ddpm_torch.Trainer(model, opt, dif, epochs=1, trainloader=None, sampler=object(), use_ema=True, shape=SHAPE[1:],
                                device=dev, distributed=True, rank=rank)
        model.zero_grad(set_to_none=True)
        losses = []
        for i in range(4 if kind.startswith("cuda") or mode in ("native_plan", "native_eager4") else 1):
            tr.stats.reset()
            tr.step(x.clamp(-1, 1).to(dev), global_steps=i + 1)
            losses.append(tr.current_stats["loss"])
        ds = tr._direct.get((SHAPE, True))
        torch.save(dict(sd=cpu(model.state_dict()), shadow=cpu(tr.ema.shadow), losses=losses,
                        segments=None if ds is None or ds.graph is None else ds.graph.launches,
                        plan_segments=None if ds is None or ds.plan is None else len(ds.plan.segments),
                        plan_launches=None if ds is None or ds.plan is None else ds.plan.launches,
                        last_kind=None if ds is None else ds.last_kind,
                        direct=ds is not None), os.path.join(out_dir, f"after_step_{mode}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], *(sys.argv[6:7]))
