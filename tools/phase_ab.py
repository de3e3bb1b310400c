"""rv_prove on the headline circuit under two environments, alternating in one process: per-phase HIP-event times.
    python tools/phase_ab.py          (A: as shipped; B: RV_OVERLAP=0 -- does the hash phase depend on what ran before it?)"""
import os, sys, time
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, circuits, reverie_amd
from reverie_amd import _lib
prog, wit, wc, st = circuits.layered_gf2()
c = reverie_amd.Circuit(prog, wc, whole_prover=True) if "whole_prover" in reverie_amd.Circuit.__init__.__code__.co_varnames else reverie_amd.Circuit(prog, wc)
seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
L = _lib.lib()
envs = {"A as shipped": {}, "B RV_OVERLAP=0": {"RV_OVERLAP": "0"}, "C RV_OVERLAP=0 RV_EARLY=0": {"RV_OVERLAP": "0", "RV_EARLY": "0"}, "D RV_EARLY=0": {"RV_EARLY": "0"}}
for _ in range(3):
    reverie_amd.Proof.new(c, wit, [], seeds=seeds)
for rnd in range(2):
    for name, env in envs.items():
        for k in ("RV_OVERLAP", "RV_EARLY"):
            os.environ.pop(k, None)
        os.environ.update(env)
        reverie_amd.Proof.new(c, wit, [], seeds=seeds)
        L.rv_ctx_profile(c.ctx.handle, 1, 1, None)
        n = 6
        t0 = time.perf_counter()
        for _ in range(n):
            reverie_amd.Proof.new(c, wit, [], seeds=seeds)
        dt = (time.perf_counter() - t0) / n
        prof = _lib.Profile()
        L.rv_ctx_profile(c.ctx.handle, 0, 0, C.byref(prof))
        print("%-28s %.2f ms/proof | %s" % (name, dt * 1e3, " ".join("%s %.2f" % (nm, prof.ms[i] / n) for i, nm in enumerate(_lib.PHASES))), file=sys.stderr)
