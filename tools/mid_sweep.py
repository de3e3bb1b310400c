"""Prover / verifier time of layered GF(2) circuits of growing size under two builds of the library (AB_BASE=path of the other
libreverie_amd.so, built e.g. with `make OUT=../_build_base` in a worktree of an older commit): a check that a change made for one size
did not cost another.  One process per (size, build); crc32 of the proofs must agree.   AB_BASE=... python tools/mid_sweep.py"""
import os, subprocess, sys
CHILD = r'''
import os, sys, time, statistics
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from reverie_amd import _lib
if os.environ.get("RV_LIB_PATH"):  # (an older build may lack the newest hooks)
    import ctypes
    have = ctypes.CDLL(os.environ["RV_LIB_PATH"])
    _lib.SYMBOLS = [s for s in _lib.SYMBOLS if hasattr(have, s)]
import reverie_amd as rv
import circuits
width, layers, p_and = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
prog, wit, wc, st = circuits.layered_gf2(n_in=256, width=width, layers=layers, p_and=p_and, fold_to=16)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
c = rv.Circuit(prog, wc, whole_prover=True)
for _ in range(4): p = rv.Proof.new(c, wit, [], seeds=seeds)
ts = []
for _ in range(30):
    t = time.perf_counter(); p = rv.Proof.new(c, wit, [], seeds=seeds); ts.append((time.perf_counter() - t) * 1e3)
ok = p.verify(c)
tv = []
for _ in range(10):
    t = time.perf_counter(); ok = p.verify(c); tv.append((time.perf_counter() - t) * 1e3)
import zlib
print("RESULT and %8d | prove %.3f ms | verify %.3f ms | ok %s crc %08x" % (st["and"], statistics.median(ts), statistics.median(tv), ok, zlib.crc32(bytes(p))))
'''
base = os.environ.get("AB_BASE", os.path.join(os.getcwd(), "reverie_amd", "_build_base", "libreverie_amd.so"))
for width, layers, p_and in ((2048, 24, 0.5), (8192, 32, 0.5), (16384, 60, 0.5), (32768, 64, 0.5), (65536, 40, 0.5), (65536, 80, 1.0), (131072, 40, 0.5)):
    for name, lib in (("base", base), ("head", "")):
        env = dict(os.environ)
        if lib: env["RV_LIB_PATH"] = lib
        out = subprocess.run([sys.executable, "-c", CHILD, str(width), str(layers), str(p_and)], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(name, line[0][7:] if line else "FAILED " + out.stderr[-400:], flush=True)
