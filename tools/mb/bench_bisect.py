"""Why is bench.py's loop 2.5 % slower per proof than tools/overlap_ab.py's on the same box?  One variant per process (argv[1]):
a = the A/B tool's way; b = + an explicit Context; c = + HostProver's direct ctypes loop; d = + torch loaded and a device tensor;
e = + HipShardBackend; f = rule seeds (random) instead of the counting seeds"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
v = sys.argv[1]
if v >= "d":
    import torch
    torch.cuda.synchronize()
import reverie_amd as rv
from reverie_amd import _lib
import circuits
import bench
prog, wit, wc, st = circuits.layered_gf2()
L = _lib.lib()
ctx = rv.Context(0) if v >= "b" else rv.Context.default()
c = rv.Circuit(prog, wc, ctx, whole_prover=True)
seeds = bench.rule_seeds() if v >= "f" else np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
if v >= "e":
    from reverie_amd.dist import HipShardBackend
    be = HipShardBackend(c)
if v >= "d":
    buf = torch.empty(60 << 20, dtype=torch.uint8, device="cuda")
hp = bench.HostProver(c, wit, [], seeds)
def one():
    if v >= "c":
        p, n = hp.prove()
        return p
    return rv.Proof.new(c, wit, [], seeds=seeds)
last = None
for _ in range(5):
    p = one()
    if v >= "c" and last is not None: hp.free(last)
    last = p
L.rv_ctx_profile(ctx.handle, 1, 1, None)
ts = []
for _ in range(25):
    t = time.perf_counter()
    p = one()
    ts.append((time.perf_counter() - t) * 1e3)
    if v >= "c" and last is not None: hp.free(last)
    last = p
prof = _lib.Profile()
L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
ts.sort()
print(v, "median %.3f mean %.3f" % (ts[len(ts) // 2], sum(ts) / len(ts)), " ".join(f"{n} {prof.ms[i] / max(prof.calls, 1):.3f}" for i, n in enumerate(_lib.PHASES)), flush=True)
