"""Same-box A/B of rv_verify / rv_prove on the 10^7-gate circuit: python tools/ab_verify.py  (run with RV_LIB_PATH set / unset)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import circuits
import reverie_amd
seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
prog, wit, wc, st = circuits.layered_gf2()
c = reverie_amd.Circuit(prog, wc, whole_prover=os.environ.get("HINT", "0") == "1")  # HINT=1: the prover's gate stream (lazy linear forms)
proof = reverie_amd.Proof.new(c, wit, [], seeds=seeds)
for name, fn in (("verify", lambda: proof.verify(c)), ("prove", lambda: reverie_amd.Proof.new(c, wit, [], seeds=seeds))):
    ts = []
    for _ in range(25):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[5:])
    print(os.environ.get("RV_LIB_PATH", "new"), "hint", os.environ.get("HINT", "0"), "vc", os.environ.get("RV_VERIFY_VC", "1"), name, "median %.3f ms  min %.3f ms" % (1e3 * ts[len(ts) // 2], 1e3 * ts[0]))
