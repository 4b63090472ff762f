"""Turn a rocprofv3 rocpd database (gpurun_out/.../*_results.db) into the per-kernel summary committed under profiles/."""
import re
import sqlite3
import sys


def main(db_path, out_path, note=""):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
    rows = [dict(zip(cols, r)) for r in cur.execute("select * from top_kernels")]
    tot = sum(r["total_duration"] for r in rows)
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({note})\n# source db: {db_path}\n")
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for r in rows:
            name = re.sub(r"\s+", " ", r["name"]).replace(",", ";")[:160]
            f.write(f"\"{name}\",{r['total_calls']},{r['total_duration'] / 1e3:.3f},{r['average']:.2f},{100.0 * r['total_duration'] / tot:.2f}\n")
        f.write(f"# total kernel time {tot / 1e3:.3f} ms over {sum(r['total_calls'] for r in rows)} dispatches\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
