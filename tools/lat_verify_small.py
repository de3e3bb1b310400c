"""single-proof verification latency of the AES-128 / SHA-256 Bristol circuits (host proof bytes in)"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, reverie_amd
ctx = reverie_amd.Context(0)
seeds = bench.rule_seeds()
for name in (sys.argv[1:] or ("aes128", "sha256")):
    prog, w2, wc, n_and = bench.bristol_case(name)
    c = reverie_amd.Circuit(prog, wc, ctx)
    p = reverie_amd.Proof(bytes(reverie_amd.Proof.new(c, w2, [], seeds=seeds)))
    for _ in range(5):
        assert p.verify(c)
    lat = []
    for _ in range(40):
        t0 = time.perf_counter(); p.verify(c); lat.append(time.perf_counter() - t0)
    print(name, "verify median ms %.3f min %.3f" % (statistics.median(lat) * 1e3, min(lat) * 1e3))
    lat = []
    for _ in range(5):
        assert reverie_amd.verify_batch(c, [p]) == [True]
    for _ in range(40):
        t0 = time.perf_counter(); reverie_amd.verify_batch(c, [p]); lat.append(time.perf_counter() - t0)
    print(name, "verify_batch(1) median ms %.3f min %.3f" % (statistics.median(lat) * 1e3, min(lat) * 1e3))
