"""Bit-exactness of whole proofs against the CPU oracle at sizes between the test suite's and the benchmark's (the transcript trees take
different shapes there: one shared reduction launch, tree tops of 257 ... 1 024 nodes).   python tools/parity_mid.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import circuits, oracle_lib, reverie_amd as rv
seeds = np.random.default_rng(7).integers(0, 256, (256, 16), dtype=np.uint8)
bad = 0
for width, layers, p_and in ((65536, 64, 0.5), (65536, 100, 0.5), (65536, 33, 1.0), (32768, 250, 0.5), (65536, 125, 1.0)):
    prog, wit, wc, st = circuits.layered_gf2(width=width, layers=layers, p_and=p_and)
    want = oracle_lib.prove(prog, wit, [], wc, seeds, threads=16)
    c = rv.Circuit(prog, wc, whole_prover=True)
    for k in range(2):  # (the second proof takes the early-corrections path)
        got = rv.Proof.new(c, wit, [], seeds=seeds)
        ok = bytes(got) == want
        bad += not ok
        print("and %8d  chunks %5d  proof %d: %s" % (st["and"], (st["and"] + st["inputs"] + 1023) // 1024, k, "bit-exact" if ok else "MISMATCH"), flush=True)
    assert got.verify(c)
    c.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
