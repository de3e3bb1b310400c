"""A/B of the staged broadcast-bit vectors of rv_prove's early path (RecStage) on the 10^7-gate circuit: ms per host-to-host proof with
RV_EARLY_REC=0 / 1 and several slice counts, bytes compared.  python tools/early_rec_ab.py [reps] [slices,...]   (AB_P_AND=1.0: all-AND)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import reverie_amd as rv
import circuits

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
slices = sys.argv[2].split(",") if len(sys.argv) > 2 else ["4"]
prog, wit, wc, st = circuits.layered_gf2(p_and=float(os.environ.get("AB_P_AND", "0.5")))
seeds = np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
c = rv.Circuit(prog, wc, whole_prover=True)


def run(label):
    for _ in range(3):
        p = rv.Proof.new(c, wit, [], seeds=seeds)
    ts = []
    for i in range(reps):
        t = time.perf_counter()
        p = rv.Proof.new(c, wit, [], seeds=seeds)
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    print(f"{label}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  -> {st['and'] / ts[len(ts)//2] * 1e3:.3e} AND/s", flush=True)
    return bytes(p)


os.environ["RV_EARLY_REC"] = "0"
ref = run("RV_EARLY_REC=0")
os.environ["RV_EARLY_REC"] = "1"
for sl in slices:
    os.environ["RV_EARLY_REC_SLICES"] = sl
    got = run(f"RV_EARLY_REC=1 slices={sl}")
    print("  bytes equal:", got == ref, flush=True)
os.environ["RV_EARLY_REC"] = "0"
run("RV_EARLY_REC=0 again")
