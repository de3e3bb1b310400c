#!/bin/bash
# PMC counters of a short bench run: tools/kpmc.sh <tag> "<counters>" [kernel filter]
tag=$1; ctrs=$2; filt=${3:-k_rep}
mkdir -p /root/repo/gpurun_out/r2/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs -d /root/repo/gpurun_out/r2/$tag -- python /root/repo/bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
cd /root/repo
db=$(ls gpurun_out/r2/$tag/*/*_results.db | head -1)
python - "$db" "$filt" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == 'counters_collection']
cur = c.execute("select * from counters_collection")
cols = [d[0] for d in cur.description]
agg = {}
for r in cur:
    d = dict(zip(cols, r))
    k = str(d.get("kernel_name") or d.get("name") or d.get("kernel"))
    if sys.argv[2] not in k: continue
    k = k.split("(")[0]
    cn = str(d.get("counter_name")); v = float(d.get("value") or d.get("counter_value") or 0)
    a = agg.setdefault((k, cn), [0, 0.0]); a[0] += 1; a[1] += v
for (k, cn), (n, v) in sorted(agg.items()):
    print(f"{k:40s} {cn:28s} per-launch {v/n:.4g}  (n={n})")
PY
