"""config 5 (10^6 Z64 MUL) host to host with the per-phase HIP-event times"""
import os, sys, time
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, reverie_amd, circuits
from reverie_amd import _lib
ctx = reverie_amd.Context(0)
L = _lib.lib()
seeds = bench.rule_seeds()
n_mul = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_in = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prog, w64, wc, st = circuits.layered_z64(n_mul=n_mul, n_in=n_in)
c = reverie_amd.Circuit(prog, wc, ctx)
hp = bench.HostProver(c, [], w64, seeds)
hp.run(2)  # (two: the second page-locked output buffer is mapped outside the timed proofs)
L.rv_ctx_profile(ctx.handle, 1, 1, None)
n = 3
dt, data = hp.run(n)
prof = _lib.Profile()
L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
print("z64 n_mul", st["mul"], "ms/proof %.2f" % (dt / n * 1e3), "proof MB %.1f" % (len(data) / 1e6), "|",
      " ".join("%s %.2f" % (nm, prof.ms[i] / n) for i, nm in enumerate(_lib.PHASES)))
