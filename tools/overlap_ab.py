"""A/B of the mask generator beside the level launches (RV_OVERLAP) on the 10^7-gate circuit, same box, one process per variant:
ms per host-to-host proof (median), per-phase times, proof bytes compared through their BLAKE3-free hash (zlib.crc32 + length).
python tools/overlap_ab.py [reps]   (env AB_P_AND, AB_VARIANTS="name:ENV=V,ENV=V;...")"""
import os, subprocess, sys, json

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
CHILD = r'''
import os, sys, time, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
if os.environ.get("AB_TORCH"):
    import torch
    torch.cuda.synchronize()
    _buf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
import reverie_amd as rv
from reverie_amd import _lib
import circuits
reps = int(sys.argv[1])
p_and = float(os.environ.get("AB_P_AND", "0.5"))
prog, wit, wc, st = circuits.layered_gf2(p_and=p_and)
# (random seeds, as bench.py's: with the counting pattern bytes(range(256)) * 16 only 16 of the 256 repetitions' seeds differ, and the
# GPU then runs the same kernels 4 % faster -- masks || levels 3.17 against 3.27 ms, hashes 1.06 against 1.16 -- less switching, higher
# clocks: tools/mb/bench_bisect.py.  AB_SEEDS=count restores the old pattern)
seeds = (np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16) if os.environ.get("AB_SEEDS") == "count"
         else np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8))
L = _lib.lib()
c = rv.Circuit(prog, wc, whole_prover=os.environ.get("AB_HINT", "1") != "0")
for _ in range(3):
    p = rv.Proof.new(c, wit, [], seeds=seeds)
ctx = rv.Context.default()
L.rv_ctx_profile(ctx.handle, 1, 1, None)
ts = []
gap = float(os.environ.get("AB_GAP_MS", "0")) * 1e-3   # host idle between proofs (bench.py's loop has none)
for i in range(reps):
    if gap:
        time.sleep(gap)
    t = time.perf_counter()
    p = rv.Proof.new(c, wit, [], seeds=seeds)
    ts.append((time.perf_counter() - t) * 1e3)
prof = _lib.Profile()
L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
ph = "  ".join(f"{n} {prof.ms[i] / max(prof.calls, 1):.3f}" for i, n in enumerate(_lib.PHASES))
ts.sort()
b = bytes(p)
print(f"RESULT median {ts[len(ts)//2]:.3f} min {ts[0]:.3f} max {ts[-1]:.3f} and/s {st['and'] / ts[len(ts)//2] * 1e3:.3e} crc {zlib.crc32(b):08x} len {len(b)} | {ph}", flush=True)
'''
variants = os.environ.get("AB_VARIANTS", "base:;col4:RV_AES_COL4=1;overlap:RV_OVERLAP=1;overlap8:RV_OVERLAP=1,RV_OVERLAP_CHUNKS=8;overlap32:RV_OVERLAP=1,RV_OVERLAP_CHUNKS=32;base2:")
for v in variants.split(";"):
    name, _, envs = v.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, val = kv.partition("=")
        env[k] = val
    r = subprocess.run([sys.executable, "-c", CHILD, str(reps)], env=env, capture_output=True, text=True, timeout=600)
    out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(f"{name:12s} {out[0][7:] if out else 'FAILED rc=%d %s' % (r.returncode, r.stderr[-400:])}", flush=True)
