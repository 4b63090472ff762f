"""CPU: the last committed bench.py output (profiles/*_bench_default.json) carries every field of the driver's contract."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_matches_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_default.json")))
    assert files, "no committed bench output under profiles/"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k, t in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float, higher_is_better=bool,
                     scaling=str, dtype=str, data=str, config=dict, roofline=dict, cpu_baseline=dict).items():
        assert isinstance(d[k], t), (k, type(d[k]))
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert abs(d["ms_per_step"] * d["value"] / 1e3 - d["config"]["global_batch"]) < 0.01 * d["config"]["global_batch"]
    # physically possible: no kernel row above the dense peak (a row at 1.8x once exposed events bracketing the wrong stream), the kernels' summed
    # isolated time below the step, the dominant kernel's traffic present and its in-product time consistent with launches x average
    for table in (r["per_kernel"], r["isolated"]["per_kernel"]):
        assert all(0 < v["frac"] < 1 for v in table.values()), {k: v["frac"] for k, v in table.items() if not 0 < v["frac"] < 1}
    assert r["isolated"]["all_mfma_kernels"]["ms"] < d["ms_per_step"]
    dom = r["per_kernel"][r["kernel"]]
    assert dom["launches"] == r["launches_per_step"] and abs(dom["avg_launch_us"] - r["avg_launch_us"]) < 0.05
    assert r["traffic"] and r["traffic"]["hbm_bytes_per_launch"] > 0 and r["runner_up"]["kernel"] != r["kernel"]
    # round 6: the timed region is 5 consecutive blocks of `steps`, the line reports the median block
    blocks = d["ms_per_step_blocks"]
    assert len(blocks) == 5 and sorted(blocks)[2] == d["ms_per_step"] and "median" in d["timing"]
    assert max(blocks) < 1.05 * min(blocks), blocks                      # the blocks of one run agree far better than boxes do
    # the headline kernel is the one with the most summed launch duration in the step; a launch that takes part of the chip says so
    assert r["kernel"] == max(r["per_kernel"].items(), key=lambda kv: kv[1]["ms"])[0] == r["dominant_by_duration"]["kernel"]
    if "cu_share" in r:
        assert 0 < r["cu_share"] < 1 and abs(r["frac_of_occupied_cus"] - r["frac"] / r["cu_share"]) < 2e-3
    # HBM rows: GroupNorm forward / backward and the 1x1 kernels, in the step and isolated, priced against 6.3 TB/s achievable
    for leg in ("in_step", "isolated"):
        fams = [k for k in r["hbm_kernels"][leg] if " | " not in k]
        assert len(fams) == 4 and all(0 < r["hbm_kernels"][leg][k]["frac_of_achievable_6300"] < 1 for k in fams), fams
    # CPU baseline: an all-physical-cores point and the 256 x 256 leg next to the CIFAR one
    assert c["all_cores"]["threads"] == c["physical_cores"] and c["all_cores"]["imgs_per_s"] > 0
    assert c["celebahq_256x256"]["forward_imgs_per_s"] > 0


def _bench(args, env_extra, timeout=300):
    env = dict(os.environ, BENCH_LAUNCH_PROBE="1", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bare_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE launches two ranks itself (gloo stand-in for the communicator: no GPU here)."""
    r = _bench(["--gpus", "2"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_probe": True, "n_gpus": 2, "ranks_in_collective": 2}


def test_world_size_mismatch_is_an_error():
    r = _bench(["--gpus", "4"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _bench(["--gpus", "1"], {})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
