"""single-proof latency of the AES-128 / SHA-256 Bristol circuits (host to host), rep-sliced path on/off via RV_REP"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, reverie_amd
ctx = reverie_amd.Context(0)
seeds = bench.rule_seeds()
for name in ("aes128", "sha256"):
    prog, w2, wc, n_and = bench.bristol_case(name)
    c = reverie_amd.Circuit(prog, wc, ctx)
    hp = bench.HostProver(c, w2, [], seeds)
    hp.run(5)
    lat = [hp.run(1)[0] for _ in range(40)]
    print(name, "RV_REP=" + os.environ.get("RV_REP", "1"), "median ms %.3f min %.3f" % (statistics.median(lat) * 1e3, min(lat) * 1e3))
