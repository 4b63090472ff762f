"""Worker for tests/test_ddp_gloo.py: one rank of a world_size-2 gloo job on CPU.  The engine runs unmodified; its C-ABI
calls go to tests/abi_emulator.py (the HIP kernels need a GPU, the data-parallel LOGIC does not)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]

TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1, apply_attn=[False, True], drop_rate=0.0)


def install_emulator():
    from ddpm_torch import _hip
    from tests.abi_emulator import Emulator
    emu = Emulator()
    _hip.lib()
    _hip.call = emu.call
    _hip.stream = lambda: 0
    _hip.require_cuda = lambda *a: None


def run(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    install_emulator()
    import ddpm_torch
    from oracle import unet_ref as U
    torch.manual_seed(100 + rank)                        # different initial weights per rank: the broadcast must fix that
    model = ddpm_torch.UNet(**TINY)
    if rank == 0:
        torch.manual_seed(7)
        model.load_state_dict(U.randomize_state_dict(ddpm_torch.UNet(**TINY).state_dict(), 17))
    g = torch.Generator().manual_seed(1000 + rank)       # each rank sees its own shard
    x, gy, t = torch.randn(2, 3, 8, 8, generator=g), torch.randn(2, 3, 8, 8, generator=g), torch.randint(0, 1000, (2,), generator=g)
    if mode == "native":
        model.set_process_group()                         # broadcast + chunked all-reduce inside the hand-written backward
        net = model
    else:
        net = torch.nn.parallel.DistributedDataParallel(model)     # the reference's wrapper (train.py:110) must keep working
    net.train()
    y = net(x, t)
    (y * gy).sum().backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    torch.save(dict(grads=grads, sd={k: v.clone() for k, v in model.state_dict().items()}, x=x, gy=gy, t=t), os.path.join(out_dir, f"{mode}_{rank}.pt"))
    if mode == "native":                                  # a full distributed Trainer.step on top (loss reduce to rank 0 included)
        dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        tr = ddpm_torch.Trainer(model, opt, dif, epochs=1, trainloader=None, sampler=object(), use_ema=True, shape=(3, 8, 8),
                                device=torch.device("cpu"), distributed=True, rank=rank)
        model.zero_grad(set_to_none=True)
        tr.step(x.clamp(-1, 1))
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, f"after_step_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
