"""Which GF(2) mask generator for a proof of n CTR blocks per slot when nothing runs beside it: the 128-plane kernel (k_aes_gf2_masks<16>, a
lane per 32 blocks) or the lane-distributed one (k_aes_gf2_masks_col4, a quad of lanes per state).  One process per (size, generator):
masks-phase time and host-to-host proof time of layered circuits of growing size.   python tools/gen_sweep.py"""
import os, subprocess, sys
CHILD = r'''
import os, sys, time, statistics
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
import reverie_amd as rv
from reverie_amd import _lib
import circuits
width, layers = int(sys.argv[1]), int(sys.argv[2])
prog, wit, wc, st = circuits.layered_gf2(n_in=256, width=width, layers=layers, fold_to=16)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
c = rv.Circuit(prog, wc, whole_prover=True)
for _ in range(4): p = rv.Proof.new(c, wit, [], seeds=seeds)
ctx = rv.Context.default(); L = _lib.lib()
ts = []
for _ in range(30):
    t = time.perf_counter(); p = rv.Proof.new(c, wit, [], seeds=seeds); ts.append((time.perf_counter() - t) * 1e3)
L.rv_ctx_profile(ctx.handle, 1, 1, None)
for _ in range(10): p = rv.Proof.new(c, wit, [], seeds=seeds)
prof = _lib.Profile(); L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
print("RESULT and %d blocks %d | proof %.3f ms | masks %.3f interp %.3f" % (st["and"], (256 + 2 * st["and"] + 127) // 128, statistics.median(ts), prof.ms[1] / 10, prof.ms[2] / 10))
'''
for width, layers in ((1024, 12), (2048, 24), (4096, 32), (8192, 32), (8192, 64), (16384, 60)):
    for gen in ("", "1"):
        env = dict(os.environ, RV_OVERLAP="0")
        if gen: env["RV_AES_COL4"] = gen
        out = subprocess.run([sys.executable, "-c", CHILD, str(width), str(layers)], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print("col4 " if gen else "plane", line[0][7:] if line else "FAILED " + out.stderr[-300:], flush=True)
