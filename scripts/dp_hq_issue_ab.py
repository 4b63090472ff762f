"""CelebA-HQ 256x256, B = 2 (BASELINE config 5's per-GPU work) through the native data-parallel step on a one-rank RCCL communicator:
ms per step with the chunk exchanges ordered behind the main stream (DDPM_DP_ISSUE_ON_SIDE=0, the round-4 form) and behind the
weight-gradient stream (1, default).  usage: python scripts/dp_hq_issue_ab.py  -> two lines."""
import os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[1]}", world_size=1, rank=0)
    import ddpm_torch
    import ddpm_torch.utils.train as train_mod
    from bench import CELEBAHQ, make_trainer
    train_mod._TRAIN_GRAPH = "plan"
    ddpm_torch.seed_all(1234)
    model, net, dif, tr = make_trainer(ddpm_torch, CELEBAHQ, dev, "bf16", (3, 256, 256), "fixed-small", True, 0, True, 0)
    x0 = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
    net.train()
    for i in range(6):
        tr.step(x0, global_steps=i + 1)
    tr.current_stats; torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(12):
        tr.step(x0, global_steps=10 + i)
    tr.current_stats; torch.cuda.synchronize()
    ds = next(iter(tr._direct.values()))
    print(f"DDPM_DP_ISSUE_ON_SIDE={os.environ.get('DDPM_DP_ISSUE_ON_SIDE', '1')} celebahq 256x256 B=2, one-rank RCCL, {ds.last_kind} form: "
          f"{(time.perf_counter() - t0) / 12 * 1e3:.3f} ms/step", file=sys.stderr, flush=True)
    dist.destroy_process_group()
else:
    for v in ("0", "1"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        p = subprocess.run([sys.executable, __file__, str(port)], env=dict(os.environ, DDPM_DP_ISSUE_ON_SIDE=v), capture_output=True, text=True, timeout=100)
        print("\n".join(l for l in p.stderr.splitlines() if "ms/step" in l) or p.stderr[-400:])
