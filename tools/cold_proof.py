import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import reverie_amd as rv
import circuits
prog, wit, wc, st = circuits.layered_gf2()
seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
c = rv.Circuit(prog, wc, whole_prover=True)
for _ in range(3): rv.Proof.new(c, wit, [], seeds=seeds)
os.environ["RV_OPS_CACHE"] = "0"
for i in range(4):
    if i == 3: os.environ["RV_COMPILE_STATS"] = "1"
    t = time.perf_counter()
    p = rv.Proof.new(prog, wit, [], wc, seeds=seeds)
    print("cold Proof.new(ops): %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
    del p
