"""A/B on the 10^6-MUL Z64 circuit (config 5), one process per variant: ms per host-to-host proof (median), per-phase GPU times,
crc32 + length of the proof (variants must agree), strict verification of the last proof.
python tools/z64_ab.py [proofs]   (env AB_VARIANTS="name:ENV=V,ENV=V;...", default: the fused kernel against the two-kernel prover)"""
import os, subprocess, sys

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CHILD = r'''
import os, sys, time, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
import reverie_amd as rv
from reverie_amd import _lib
import circuits
reps = int(sys.argv[1])
prog, w64, wc, st = circuits.layered_z64()
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
L = _lib.lib()
c = rv.Circuit(prog, wc)
for _ in range(2):
    p = rv.Proof.new(c, [], w64, seeds=seeds)
ctx = rv.Context.default()
L.rv_ctx_profile(ctx.handle, 1, 1, None)
ts = []
for i in range(reps):
    del p
    t = time.perf_counter()
    p = rv.Proof.new(c, [], w64, seeds=seeds)
    ts.append((time.perf_counter() - t) * 1e3)
prof = _lib.Profile()
L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
ph = "  ".join(f"{n} {prof.ms[i] / max(prof.calls, 1):.3f}" for i, n in enumerate(_lib.PHASES))
ts.sort()
ok = bool(p.verify(c))
ptr, n = p._buffer()
b = (C.c_char * n).from_address(ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value)
crc = zlib.crc32(b)
print(f"RESULT median {ts[len(ts)//2]:.2f} min {ts[0]:.2f} max {ts[-1]:.2f} mul/s {st['mul'] / ts[len(ts)//2] * 1e3:.3e} crc {crc:08x} len {n} verifies {ok} | {ph}", flush=True)
'''
variants = os.environ.get("AB_VARIANTS", "fused:;two_kernels:RV_Z64_FUSED=0;fused2:")
for v in variants.split(";"):
    name, _, envs = v.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, val = kv.partition("=")
        env[k] = val
    out = subprocess.run([sys.executable, "-c", CHILD, str(reps)], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(f"{name:12s} {line[0][7:] if line else 'FAILED: ' + out.stderr[-600:]}", flush=True)
