"""Timeline of the opening phase of the LAST proof of a short bench run under
rocprofv3 --kernel-trace --memory-copy-trace: kernels and copies from the last k_fs_challenge on (us, relative).
usage (GPU box): python tools/trace_open.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = [(st, en, "q%s" % q, name.split("(")[0][:60])
        for st, en, q, name in c.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id")]
mc = [t for t in tabs if "memory_copy" in t and "rocpd" in t]
if mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({mc[0]})")]
    size = "size" if "size" in cols else ("bytes" if "bytes" in cols else "0")
    for st, en, sz in c.execute(f"select start, end, {size} from {mc[0]}"):
        rows.append((st, en, "copy", "memcpy %d bytes (%.1f GB/s)" % (sz, sz / max(en - st, 1))))
else:
    print("no memory-copy table among", tabs)
rows.sort()
idx = max(i for i, r in enumerate(rows) if "k_fs_challenge" in r[3])
t0 = rows[idx][0]
for st, en, q, name in rows[idx - 2:idx + int(sys.argv[2]) if len(sys.argv) > 2 else idx + 40]:
    print("%9.1f %9.1f  %-5s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, q, name))
