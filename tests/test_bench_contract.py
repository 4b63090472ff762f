"""CPU: the last committed bench.py output (profiles/*_bench_default.json) carries every field of the driver's contract."""
import glob
import json
import os

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
