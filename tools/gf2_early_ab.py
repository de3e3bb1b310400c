"""Same-process A/B of rv_prove's GF(2) early-corrections plans on the 10^7-gate circuit (AB_P_AND=1.0: its all-AND variant): one
context and circuit per plan (the plan is read per circuit), proofs interleaved.
usage: python tools/gf2_early_ab.py "RV_EARLY=2 RV_EARLY_REPS=256 RV_EARLY_CHUNKS=12" ["..." ...] [rounds=N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import reverie_amd as rv
import circuits
p_and = float(os.environ.get("AB_P_AND", "0.5"))
prog, wit, wc, st = circuits.layered_gf2(p_and=p_and)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)  # (random, as bench.py: the counting pattern has 16 distinct seeds and runs 4 % faster)
args = [a for a in sys.argv[1:] if not a.startswith("rounds=")]
rounds = int(([a for a in sys.argv[1:] if a.startswith("rounds=")] or ["rounds=15"])[0].split("=")[1])
variants = [("default", {})] + [(a, dict(kv.split("=") for kv in a.split())) for a in args]
provers = []
for name, env in variants:
    for k in ("RV_EARLY", "RV_EARLY_REPS", "RV_EARLY_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = rv.Context(0)
    c = rv.Circuit(prog, wc, ctx, whole_prover=os.environ.get("AB_HINT", "1") != "0")
    for _ in range(3):
        p = rv.Proof.new(c, wit, [], seeds=seeds)
    provers.append((name, c, ctx, bytes(p)))
assert all(b == provers[0][3] for *_, b in provers)
ts = {name: [] for name, *_ in provers}
for r in range(rounds):
    for name, c, ctx, _ in provers:
        t = time.perf_counter()
        p = rv.Proof.new(c, wit, [], seeds=seeds)
        ts[name].append((time.perf_counter() - t) * 1e3)
        del p
for name in ts:
    v = sorted(ts[name])
    print("%-50s median %.3f  min %.3f  max %.3f ms -> %.3e AND/s" % (name, v[len(v) // 2], v[0], v[-1], st["and"] / v[len(v) // 2] * 1e3))
