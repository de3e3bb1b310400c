"""Same-process A/B of the Z64 early-corrections staging (config 5): two contexts, one circuit each, compiled under different
RV_EARLY / RV_EARLY_REPS / RV_EARLY_CHUNKS (the plan is read per circuit), proofs interleaved.
usage: python tools/z64_early_ab.py "RV_EARLY=2 RV_EARLY_REPS=256 RV_EARLY_CHUNKS=12" [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, reverie_amd, circuits
seeds = bench.rule_seeds()
prog, w64, wc, st = circuits.layered_z64()
variants = [("default", {}), (sys.argv[1], dict(kv.split("=") for kv in sys.argv[1].split()))]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
provers = []
for name, env in variants:
    for k in ("RV_EARLY", "RV_EARLY_REPS", "RV_EARLY_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = reverie_amd.Context(0)
    c = reverie_amd.Circuit(prog, wc, ctx)
    hp = bench.HostProver(c, [], w64, seeds)
    hp.run(2)  # (the plan and the staging buffer are made here, under this variant's environment)
    provers.append((name, hp, c, ctx))
ts = {name: [] for name, *_ in provers}
for r in range(rounds):
    for name, hp, c, ctx in provers:
        dt, data = hp.run(1)
        ts[name].append(dt * 1e3)
for name in ts:
    v = sorted(ts[name])
    print("%-50s median %.2f  min %.2f  max %.2f ms  (%d proofs)" % (name, v[len(v) // 2], v[0], v[-1], len(v)))
