"""Timeline of the LAST proof of a run under rocprofv3 --kernel-trace --memory-copy-trace: kernels that are not interpreter levels
(those are summarised per 20), and memory copies, with start / end in us relative to the proof's first mask kernel.
usage: python tools/early_trace.py <results.db>"""
import sqlite3, sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = [(st, en, "q%s" % q, name.split("(")[0][:50]) for st, en, q, name in c.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id")]
mc = [t for t in tabs if "memory_copy" in t and "rocpd" in t]
for t in mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
    if "start" in cols and "end" in cols:
        szc = "size" if "size" in cols else None
        for r in c.execute(f"select start, end{', ' + szc if szc else ''} from {t}"):
            rows.append((r[0], r[1], "copy", "memcpy %s B" % (r[2] if szc else "?")))
        break
rows.sort()
idx = max(i for i, r in enumerate(rows) if "k_aes_gf2_masks" in r[3])
t0 = rows[idx][0]
n_lvl, lvl_a, lvl_b = 0, None, None
for st, en, q, name in rows[idx:]:
    if "k_interp_full" in name:
        if n_lvl == 0:
            lvl_a = st
        n_lvl += 1
        lvl_b = en
        if n_lvl == 10:
            print("%9.1f %9.1f  %-5s 10 x k_interp_full  (%.1f us)" % ((lvl_a - t0) / 1e3, (lvl_b - t0) / 1e3, q, (lvl_b - lvl_a) / 1e3))
            n_lvl = 0
        continue
    if n_lvl:
        print("%9.1f %9.1f  %-5s %d x k_interp_full" % ((lvl_a - t0) / 1e3, (lvl_b - t0) / 1e3, q, n_lvl))
        n_lvl = 0
    print("%9.1f %9.1f  %-5s %s  (%.1f us)" % ((st - t0) / 1e3, (en - t0) / 1e3, q, name, (en - st) / 1e3))
