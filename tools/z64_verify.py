"""config 5 (10^6 Z64 MUL): rv_verify wall times (warm), for rocprofv3 runs of the verifier's kernels"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, reverie_amd, circuits
seeds = bench.rule_seeds()
n_mul = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prog, w64, wc, st = circuits.layered_z64(n_mul=n_mul)
c = reverie_amd.Circuit(prog, wc)
proof = reverie_amd.Proof.new(c, [], w64, seeds=seeds)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); ok = proof.verify(c); ts.append(time.perf_counter() - t0)
print("z64 verify", ok, " ".join("%.2f" % (1e3 * t) for t in ts), "ms", file=sys.stderr)
copy = reverie_amd.Proof(bytes(proof))  # ordinary pageable memory, as a proof read from disk would be
ts = []
for _ in range(4):
    t0 = time.perf_counter(); ok = copy.verify(c); ts.append(time.perf_counter() - t0)
print("z64 verify (pageable proof)", ok, " ".join("%.2f" % (1e3 * t) for t in ts), "ms", file=sys.stderr)
