"""single-proof latency of the AES-128 / SHA-256 Bristol circuits (host to host) with the per-phase HIP-event times;
RV_LDS_RUN=0 / RV_LDS_QS=2|4 select the interpreter for the narrow stretches"""
import os, sys, time, statistics
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, reverie_amd
from reverie_amd import _lib
ctx = reverie_amd.Context(0)
L = _lib.lib()
seeds = bench.rule_seeds()
for name in (sys.argv[1:] or ("aes128", "sha256")):
    prog, w2, wc, n_and = bench.bristol_case(name)
    c = reverie_amd.Circuit(prog, wc, ctx)
    hp = bench.HostProver(c, w2, [], seeds)
    hp.run(5)
    lat = [hp.run(1)[0] for _ in range(40)]
    L.rv_ctx_profile(ctx.handle, 1, 1, None)
    hp.run(20)
    prof = _lib.Profile()
    L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
    ph = " ".join("%s %.3f" % (n, prof.ms[i] / 20) for i, n in enumerate(_lib.PHASES))
    print(name, "LDS_RUN=" + os.environ.get("RV_LDS_RUN", "1"), "QS=" + os.environ.get("RV_LDS_QS", "auto"),
          "median ms %.3f min %.3f |" % (statistics.median(lat) * 1e3, min(lat) * 1e3), ph)
