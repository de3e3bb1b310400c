"""Kernel timeline of the LAST proof of a short bench run under rocprofv3 --kernel-trace: start / end (us, relative) and queue of
every kernel from the first opening kernel on -- shows whether the copy kernels of the second stream overlap the extraction.
usage (GPU box): tools/trace_tail.sh ; python tools/trace_tail.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
rows = list(c.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
# last k_fs_challenge onwards
idx = max(i for i, r in enumerate(rows) if "k_fs_challenge" in r[3])
t0 = rows[idx][0]
for st, en, q, name in rows[idx:idx + 40]:
    print("%9.1f %9.1f  q%-3s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, q, name.split("(")[0][:60]))
