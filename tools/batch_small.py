import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import bench, reverie_amd
ctx = reverie_amd.Context(0)
seeds = bench.rule_seeds()
name = sys.argv[1] if len(sys.argv) > 1 else "aes128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prog, w2, wc, n_and = bench.bristol_case(name)
c = reverie_amd.Circuit(prog, wc, ctx)
rng = np.random.default_rng(B)
bs = rng.integers(0, 256, (B, 256, 16), dtype=np.uint8)
bw = np.tile(np.asarray(w2, np.uint8), (B, 1))
reverie_amd.Proof.new_batch(c, bw, seeds=bs)
t0 = time.perf_counter()
for _ in range(4):
    p = reverie_amd.Proof.new_batch(c, bw, seeds=bs)
dt = (time.perf_counter() - t0) / 4
print(name, B, "ms per call %.3f us per proof %.2f" % (dt * 1e3, dt / B * 1e6))
