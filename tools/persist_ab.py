"""A/B of the persistent level kernel (k_interp_persist) on the 10^7-gate circuit (same box, same process): ms per host-to-host proof and per-phase GPU
times with RV_PERSIST=0 / 1, bytes compared with each other.  python tools/flat_ab.py [reps] [bands,...]   (AB_P_AND=1.0: all-AND)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import reverie_amd as rv
from reverie_amd import _lib
import circuits

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
band_list = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"]
p_and = float(os.environ.get("AB_P_AND", "0.5"))
prog, wit, wc, st = circuits.layered_gf2(p_and=p_and)
seeds = np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
L = _lib.lib()
PH = _lib.PHASES + ["early", "clear"]


def run(label):
    t = time.perf_counter()
    c = rv.Circuit(prog, wc, whole_prover=os.environ.get("AB_HINT", "1") != "0")
    t_compile = time.perf_counter() - t
    for _ in range(3):
        p = rv.Proof.new(c, wit, [], seeds=seeds)
    ts = []
    ctx = rv.Context.default()
    L.rv_ctx_profile(ctx.handle, 1, 1, None)
    for i in range(reps):
        t = time.perf_counter()
        p = rv.Proof.new(c, wit, [], seeds=seeds)
        ts.append((time.perf_counter() - t) * 1e3)
    prof = _lib.Profile()
    L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
    print("   phases (ms/proof):", "  ".join(f"{n} {prof.ms[i] / max(prof.calls, 1):.3f}" for i, n in enumerate(PH)),
          " launches:", " ".join(str(prof.launches[i] // max(prof.calls, 1)) for i in range(8)), flush=True)
    ts.sort()
    print(f"{label}: compile+upload {t_compile:.2f} s; median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}  -> {st['and'] / ts[len(ts)//2] * 1e3:.3e} AND/s", flush=True)
    b = bytes(p)
    ok = p.verify(c, strict=True)
    c.close()
    return b, ok


os.environ["RV_PERSIST"] = "0"
ref, ok = run("RV_PERSIST=0")
print("  verifies:", ok, flush=True)
for occ in band_list:
    os.environ["RV_PERSIST"] = os.environ.get("AB_PERSIST", "1")
    if occ != "0":
        os.environ["RV_PERSIST_OCC"] = occ
    got, ok = run(f"RV_PERSIST=1 occ={occ}")
    print("  bytes equal:", got == ref, " verifies:", ok, flush=True)
