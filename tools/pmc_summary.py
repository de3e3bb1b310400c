#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md prescribes)
-> per-kernel HBM traffic per proof.

usage: tools/pmc_summary.py <fetch_db> <write_db> <n_proofs> <out_prefix>

Corrections applied (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a coalesced streaming read, so the read side is doubled
(checked here against k_b3_chunks, whose algorithmic read volume is exactly known: the whole
online transcript once).  WRITE_SIZE matched the algorithmic write volume of k_aes_gf2_masks and
k_interp_full to within 2 % uncorrected.
"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, n, total in c.execute(
            "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        out[name.split("(")[0].replace("void ", "")] = (n, total)
    return out


def main():
    fetch_db, write_db, n, prefix = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(f) | set(w)):
        fk = f.get(k, (0, 0.0))
        wk = w.get(k, (0, 0.0))
        rd = fk[1] * 1024 * 2 / n  # KiB -> B, gfx950 x2 correction
        wr = wk[1] * 1024 / n
        rows[k] = {"launches_per_proof": max(fk[0], wk[0]) / n, "read_bytes_per_proof": rd, "write_bytes_per_proof": wr,
                   "hbm_bytes_per_proof": rd + wr, "fetch_size_kib_raw_total": fk[1], "write_size_kib_raw_total": wk[1]}
    json.dump({"n_proofs": n, "kernels": rows}, open(prefix + ".json", "w"), indent=1)
    with open(prefix + ".txt", "w") as fh:
        fh.write(f"# rocprofv3 --pmc FETCH_SIZE ({fetch_db}) and --pmc WRITE_SIZE ({write_db}), separate passes, {n:g} proofs each\n")
        fh.write("# per proof; read = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), write = WRITE_SIZE KiB x 1024\n")
        fh.write(f"{'launches':>9} {'read_GB':>9} {'write_GB':>9} {'total_GB':>9}  kernel\n")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_proof"]):
            fh.write(f"{r['launches_per_proof']:9.1f} {r['read_bytes_per_proof'] / 1e9:9.3f} {r['write_bytes_per_proof'] / 1e9:9.3f} "
                     f"{r['hbm_bytes_per_proof'] / 1e9:9.3f}  {k}\n")
    print(open(prefix + ".txt").read())


if __name__ == "__main__":
    main()
