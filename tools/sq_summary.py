#!/usr/bin/env python3
"""SQ counters of one rocprofv3 --pmc pass -> per-kernel JSON for bench.py's roofline.valu (instruction-issue fractions of the
integer-VALU-bound kernels, SURVEY 8(d): "integer VALU throughput" is the binding roofline of the mask and hash phases).

usage: tools/sq_summary.py <sq_results.db> <n_proofs> <out.json>"""
import json
import sqlite3
import sys


def main():
    db, n, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    c = sqlite3.connect(db)
    rows = {}
    for name, counter, cnt, total in c.execute(
            "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        k = name.split("(")[0].replace("void ", "")
        rows.setdefault(k, {"launches_per_proof": cnt / n})[counter + "_per_proof"] = total / n
    for k, r in rows.items():
        wc, wi = r.get("SQ_WAVE_CYCLES_per_proof"), r.get("SQ_WAIT_INST_ANY_per_proof")
        if wc and wi is not None:
            r["wait_inst_any_frac_of_wave_cycles"] = wi / wc
    json.dump({"n_proofs": n, "counters": "SQ_INSTS_VALU = wavefront-level VALU instructions issued (all SIMDs, all XCDs)", "kernels": rows},
              open(out, "w"), indent=1)
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_per_proof", 0))[:8]:
        print("%14.4g VALU insts/proof  %s" % (r.get("SQ_INSTS_VALU_per_proof", 0), k))


if __name__ == "__main__":
    main()
