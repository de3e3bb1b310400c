"""Kernel timeline of the LAST proof of a run under rocprofv3 --kernel-trace: start / end (us, relative to the proof's first
kernel), queue, grid and registers of every kernel; short kernels of one name in a row on one queue are folded into one line.
usage: python tools/trace_proof.py <results.db> [first-kernel-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_expand_seeds"
rows = list(c.execute("select start, end, queue_id, name, grid_x, workgroup_x, vgpr_count from kernels order by start"))
idx = max(i for i, r in enumerate(rows) if first in r[3])
t0 = rows[idx][0]
out = []
for st, en, q, name, gx, wx, vg in rows[idx:]:
    short = name.split("(")[0].replace("void ", "").replace("rv::", "")
    if out and out[-1][3] == short and out[-1][2] == q and (en - st) < 30000:
        o = out[-1]
        o[1] = en
        o[4] += 1
        o[5] += en - st
        continue
    out.append([st, en, q, short, 1, en - st, gx // max(wx, 1), wx, vg])
for st, en, q, short, n, busy, blocks, wx, vg in out:
    print("%9.1f %9.1f  q%-2s x%-4d busy %8.1f  %5d x %-4d v%-3d %s" % ((st - t0) / 1e3, (en - t0) / 1e3, q, n, busy / 1e3, blocks, wx, vg, short[:70]))
