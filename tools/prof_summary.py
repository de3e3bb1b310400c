#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite output) -> plain-text per-kernel summary for profiles/.

usage: tools/prof_summary.py gpurun_out/<dir>/<name>_results.db [n_proofs] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# source: rocprofv3 --kernel-trace --stats ({db}); durations in microseconds; per_proof = total / {n:g}")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6} {'per_proof_us':>13}  kernel")
    for name, calls, total, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f} {total / n:13.1f}  {short}")
    # PMC counters, if this run collected any
    try:
        pmc = list(c.execute("select * from counters_collection limit 1"))
        if pmc:
            cur = c.execute("select * from counters_collection")
            cols = [d[0] for d in cur.description]
            print("\n# counters_collection:", cols)
            agg = {}
            for r in cur:
                d = dict(zip(cols, r))
                k = (str(d.get("name") or d.get("kernel_name") or d.get("kernel")), str(d.get("counter_name")))
                agg.setdefault(k, [0, 0.0])
                agg[k][0] += 1
                agg[k][1] += float(d.get("value") or d.get("counter_value") or 0)
            for (k, cn), (cnt, val) in sorted(agg.items()):
                print(f"{cn:>28} sum={val:.6g} n={cnt} avg={val / cnt:.6g}  {k.split('(')[0]}")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main()
