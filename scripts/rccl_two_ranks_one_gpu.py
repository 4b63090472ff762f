"""Does RCCL accept two ranks on ONE device?  (It does not: 'Duplicate GPU detected' — which is why tests/test_ddp_one_gpu.py exchanges over
gloo.)  usage: python scripts/rccl_two_ranks_one_gpu.py   -> prints each rank's outcome."""
import os
import socket
import subprocess
import sys


def worker(rank, port):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=2, rank=rank, device_id=torch.device("cuda:0"))
        t = torch.full((4,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok -> {t.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:                                  # noqa: BLE001
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:400]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        worker(int(sys.argv[1]), int(sys.argv[2]))
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        procs = [subprocess.Popen([sys.executable, __file__, str(r), str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        for p in procs:
            try:
                print(p.communicate(timeout=120)[0].decode()[-1500:])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                print("timeout")
