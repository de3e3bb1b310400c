"""one rank's share of a sharded config-5 proof (10^6 Z64 MUL): commit + open of a shard of REPS repetitions, RV_Z64_FUSED=1 / 0"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, circuits, reverie_amd
from reverie_amd.dist import HipShardBackend
reps = int(os.environ.get("REPS", "32"))
prog, w64, wc, st = circuits.layered_z64(n_mul=int(os.environ.get("Z64_MULS", "1000000")))
seeds = np.random.default_rng(7).integers(0, 256, (256, 16), dtype=np.uint8)
omit = np.random.default_rng(8).integers(0, 9, 256).astype(np.uint8)
for fused in ("1", "0"):
    os.environ["RV_Z64_FUSED"] = fused
    c = reverie_amd.Circuit(prog, wc)
    be = HipShardBackend(c)
    ts = []
    for i in range(4):
        t0 = time.perf_counter()
        s = be.commit([], w64, seeds[:reps], 0, reps)
        d = be.digests(s)
        part = be.open(s, omit)
        ts.append(time.perf_counter() - t0)
        be.destroy(s)
    print("reps", reps, "RV_Z64_FUSED", fused, "ms", " ".join("%.2f" % (t * 1e3) for t in ts), file=sys.stderr)
    c.close()
