"""What ONE GPU can say about the data-parallel step (no multi-GPU node in this pool): the native gradient exchange on a one-rank RCCL
communicator, with a stand-in for the ring step's copy kernel.

For each workload (BASELINE config 3's per-GPU work: configs/cifar10.json B = 128; config 5's: configs/celebahq.json at 256x256, B = 2) and
each DDPM_DP_RESERVED_CUS in RESERVED:
  * ms per training step (launch-plan form: the all-reduce calls sit between plan segments);
  * where in the backward each gradient chunk is handed to the communicator (ms before the end of the backward), and how long the step
    waits for the last one — the engine's dp_trace;
  * a COPY KERNEL of the chunk's size (ddpm_copy_probe, 32 workgroups of 512 threads — the shape of a ring step) launched on a third
    stream at each of those points: its duration there vs alone on the idle GPU = what a collective's kernel pays for sharing the chip
    with the persistent MFMA kernels, and what reserving compute units buys.
A one-rank communicator moves nothing over xGMI: the traces show WHEN the exchanges are issued and WHAT a kernel issued there gets.
Output: profiles/r05_dp_one_rank.json / .txt (scripts/gpu_r5*.sh).
"""
import json, os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import torch.distributed as dist

RESERVED = [int(v) for v in os.environ.get("DP_RESERVED", "0,16,32").split(",")]
STEPS = int(os.environ.get("DP_STEPS", "12"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)

import ddpm_torch
from ddpm_torch import _hip
import ddpm_torch.utils.train as train_mod
from bench import CIFAR, CELEBAHQ, make_trainer

WORK = [("cifar10 32x32 B=128", CIFAR, (3, 32, 32), 128, "fixed-large"), ("celebahq 256x256 B=2", CELEBAHQ, (3, 256, 256), 2, "fixed-small")]
results = []
for name, cfg, shape, B, var in WORK:
    for reserved in RESERVED:
        os.environ["DDPM_DP_RESERVED_CUS"] = str(reserved)
        train_mod._TRAIN_GRAPH = "plan"
        ddpm_torch.seed_all(1234)
        model, net, dif, tr = make_trainer(ddpm_torch, cfg, dev, "bf16", shape, var, True, 0, True, 0)
        eng = model.engine()
        assert int(_hip.lib().ddpm_get_reserved_cus()) == reserved
        x0 = (torch.rand(B, *shape, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
        net.train()
        for i in range(6):
            tr.step(x0, global_steps=i + 1)
        tr.current_stats; torch.cuda.synchronize()
        ds = next(iter(tr._direct.values()))
        t0 = time.perf_counter()
        for i in range(STEPS):
            tr.step(x0, global_steps=10 + i)
        tr.current_stats; torch.cuda.synchronize()
        ms_plain = (time.perf_counter() - t0) / STEPS * 1e3
        # ---- stand-in copy kernels where the all-reduces are issued
        third = torch.cuda.Stream(device=dev)
        src = torch.empty(eng.ptotal, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        rec = []

        def standin(chunk):
            n = chunk.numel() * 4 // 16 * 16
            if n == 0:
                return
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            third.wait_stream(torch.cuda.current_stream())            # ordered behind the chunk's producers, like the communicator's stream
            a.record(third)
            _hip.call("ddpm_copy_probe", dst.data_ptr(), src.data_ptr(), n, 32, third.cuda_stream)
            b.record(third)
            rec.append((n, a, b))
        eng.dp_standin = standin
        eng.dp_trace = []
        t0 = time.perf_counter()
        for i in range(STEPS):
            tr.step(x0, global_steps=40 + i)
        tr.current_stats; torch.cuda.synchronize()
        ms_standin = (time.perf_counter() - t0) / STEPS * 1e3
        trace, eng.dp_trace, eng.dp_standin = eng.dp_trace, None, None
        per = len(eng.chunks) + 1
        in_step = {}
        for k, (n, a, b) in enumerate(rec):
            in_step.setdefault(k % per, []).append(a.elapsed_time(b) * 1e3)
        # the same copies alone
        alone = {}
        torch.cuda.synchronize()
        for k in range(per):
            n = rec[k][0]
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(third)
                _hip.call("ddpm_copy_probe", dst.data_ptr(), src.data_ptr(), n, 32, third.cuda_stream)
                b.record(third); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            alone[k] = min(ts[1:])
        done = [e for w, _, e in trace if w == "backward_compute_done"][-1]
        fin = [e for w, _, e in trace if w == "exchange_done"][-1]
        ars = [(nb, e) for w, nb, e in trace if w == "all_reduce"][-per:]
        r = dict(workload=name, reserved_cus=reserved, step_form=ds.last_kind, plan_segments=len(ds.plan.segments) if ds.plan else None,
                 ms_per_step=round(ms_plain, 3), ms_per_step_with_standin=round(ms_standin, 3),
                 chunks=per, chunk_mb=[round(rec[k][0] / 1e6, 1) for k in range(per)], total_mb=round(sum(rec[k][0] for k in range(per)) / 1e6, 1),
                 issued_ms_before_end_of_backward=[round(e.elapsed_time(done), 3) for _, e in ars],
                 exposed_wait_ms=round(done.elapsed_time(fin), 3),
                 standin_us_in_step=[round(sorted(v)[len(v) // 2], 1) for _, v in sorted(in_step.items())],
                 standin_us_alone=[round(alone[k], 1) for k in range(per)])
        r["standin_slowdown"] = [round(a / max(b, 1e-3), 2) for a, b in zip(r["standin_us_in_step"], r["standin_us_alone"])]
        results.append(r)
        print(json.dumps(r), flush=True)
        del tr, model, net, eng, src, dst
        torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(results, open(os.path.join(ROOT, "gpurun_out", "r05_dp_one_rank.json"), "w"), indent=1)
dist.destroy_process_group()
