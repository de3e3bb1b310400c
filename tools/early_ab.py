"""A/B of rv_prove's early-corrections path on the 10^7-gate circuit (same box, same process): ms per host-to-host proof
with RV_EARLY=0 / 1 (median of `reps` after a warm-up), bytes compared.  python tools/early_ab.py [reps] [chunks,...]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import reverie_amd as rv
from reverie_amd import _lib
import circuits

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chunk_list = sys.argv[2].split(",") if len(sys.argv) > 2 else ["10"]
p_and = float(os.environ.get("AB_P_AND", "0.5"))
prog, wit, wc, st = circuits.layered_gf2(p_and=p_and)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)  # (random, as bench.py: the counting pattern has 16 distinct seeds and runs 4 % faster)
L = _lib.lib()


def run(label):
    c = rv.Circuit(prog, wc, whole_prover=True)
    for _ in range(3):
        p = rv.Proof.new(c, wit, [], seeds=seeds)
    ts = []
    n0 = L.rv_hook_early_proofs()
    import ctypes as C
    ctx = rv.Context.default()
    L.rv_ctx_profile(ctx.handle, 1, 1, None)
    for i in range(reps):
        if i == reps - 1 and os.environ.get("RV_EARLY") == "1":
            os.environ["RV_EARLY_STATS"] = "1"
        t = time.perf_counter()
        p = rv.Proof.new(c, wit, [], seeds=seeds)
        ts.append((time.perf_counter() - t) * 1e3)
    os.environ["RV_EARLY_STATS"] = "0"
    prof = _lib.Profile()
    L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
    print("   phases (ms/proof):", "  ".join(f"{n} {prof.ms[i] / max(prof.calls, 1):.3f}" for i, n in enumerate(_lib.PHASES)), flush=True)
    ts.sort()
    print(f"{label}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}  early proofs {L.rv_hook_early_proofs() - n0}/{reps}  -> {st['and'] / ts[len(ts)//2] * 1e3:.3e} AND/s", flush=True)
    b = bytes(p)
    c.close()
    return b


os.environ["RV_EARLY"] = "0"
ref = run("RV_EARLY=0")
for ch in chunk_list:
    os.environ["RV_EARLY"] = "1"
    os.environ["RV_EARLY_CHUNKS"] = ch
    got = run(f"RV_EARLY=1 chunks={ch}")
    print("  bytes equal:", got == ref, flush=True)
os.environ["RV_EARLY"] = "0"
run("RV_EARLY=0 again")
