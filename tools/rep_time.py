"""GPU time of the rep-sliced kernels for the library variant in RV_LIB_PATH (phase 'interp' = cleartext pass + interpreter)"""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RV_REP"] = "1"
import numpy as np
import bench, circuits, reverie_amd
from reverie_amd import _lib
ctx = reverie_amd.Context(0)
prog, wit, wc, st = circuits.layered_gf2()
c = reverie_amd.Circuit(prog, wc, ctx)
hp = bench.HostProver(c, wit, [], bench.rule_seeds())
hp.run(2)
L = _lib.lib()
L.rv_ctx_profile(ctx.handle, 1, 1, None)
hp.run(4)
prof = _lib.Profile()
L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
print(os.path.basename(os.environ.get("RV_LIB_PATH", "default")), {n: round(prof.ms[i] / 4, 3) for i, n in enumerate(_lib.PHASES)})
