"""rv_verify on the headline circuit, eight calls (for rocprofv3 / RV profile phases): python tools/verify_laps.py"""
import os, sys, time
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, circuits, reverie_amd
from reverie_amd import _lib
seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
prog, wit, wc, st = circuits.layered_gf2(p_and=float(os.environ.get("P_AND", "0.5")))
c = reverie_amd.Circuit(prog, wc)
proof = reverie_amd.Proof.new(c, wit, [], seeds=seeds)
for _ in range(2):
    assert proof.verify(c)
L = _lib.lib()
L.rv_ctx_profile(c.ctx.handle, 1, 1, None)
ts = []
n = int(os.environ.get("N", "8"))
for _ in range(n):
    t0 = time.perf_counter(); ok = proof.verify(c); ts.append(time.perf_counter() - t0)
prof = _lib.Profile()
L.rv_ctx_profile(c.ctx.handle, 0, 0, C.byref(prof))
print("verify ms", " ".join("%.2f" % (t * 1e3) for t in ts), "|", " ".join("%s %.2f" % (nm, prof.ms[i] / n) for i, nm in enumerate(_lib.PHASES)), file=sys.stderr)
