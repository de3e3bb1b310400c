import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import reverie_amd as rv
import circuits
prog, wit, wc, st = circuits.layered_gf2()
c = rv.Circuit(prog, wc)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)  # (random, as bench.py)
for i in range(int(os.environ.get("N_PROOFS", "2"))):
    p = rv.Proof.new(c, wit, [], seeds=seeds)
print("done", len(p), file=sys.stderr)
