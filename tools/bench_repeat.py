"""N consecutive default `python bench.py` runs on one box; every numeric record's spread across the runs.
    python tools/bench_repeat.py [N]      (prints one line per record: min / max / spread %)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
recs = []
for i in range(n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
    recs.append(json.loads(out[-1]))
    open(os.path.join(ROOT, "gpurun_out", "bench_repeat_%d.json" % i), "w").write(out[-1] + "\n")


def flat(d, pre=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from flat(v, pre + k + ".")
        elif isinstance(v, (int, float)) and not isinstance(v, bool):
            yield pre + k, float(v)


keys = ["value", "ms_per_step"]
cfg = recs[0].get("config", {})
for k, v in flat(cfg, "config."):
    if any(t in k for t in ("ms", "per_s", "value")) and "bytes" not in k:
        keys.append(k)
tabs = [dict(flat(r)) for r in recs]
worst = 0.0
for k in keys:
    vals = [t[k] for t in tabs if k in t]
    if len(vals) < n or min(vals) <= 0:
        continue
    sp = (max(vals) - min(vals)) / (sum(vals) / len(vals)) * 100
    worst = max(worst, sp)
    if sp > 3 or k in ("value", "ms_per_step"):
        print("%-70s min %.4g max %.4g spread %.1f %%" % (k, min(vals), max(vals), sp))
print("records compared: %d, worst spread %.1f %%" % (len(keys), worst))
