import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import reverie_amd as rv
import circuits
prog, wit, wc, st = circuits.layered_gf2()
seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
c = rv.Circuit(prog, wc, whole_prover=True)
for _ in range(3): rv.Proof.new(c, wit, [], seeds=seeds)
# COLD_CACHE=1: the default cache stays on and is emptied between the calls (a first call also pays the kept copy of the op list)
keep = os.environ.get("COLD_CACHE") == "1"
if not keep: os.environ["RV_OPS_CACHE"] = "0"
from reverie_amd import _lib
for i in range(4):
    if keep: _lib.lib().rv_ctx_ops_cache_clear(rv.Context.default().handle)
    if i == 3: os.environ["RV_COMPILE_STATS"] = "1"
    t = time.perf_counter()
    p = rv.Proof.new(prog, wit, [], wc, seeds=seeds)
    print("cold Proof.new(ops): %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
    del p
