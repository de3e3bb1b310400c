"""N proofs of the 10^7-gate circuit with the schedule the environment selects (for rocprofv3 --kernel-trace --stats):
python tools/flat_prof.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import reverie_amd as rv
import circuits

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prog, wit, wc, st = circuits.layered_gf2(p_and=float(os.environ.get("AB_P_AND", "0.5")))
seeds = np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
c = rv.Circuit(prog, wc, whole_prover=os.environ.get("AB_HINT", "1") != "0")
for _ in range(n):
    p = rv.Proof.new(c, wit, [], seeds=seeds)
print("done", len(bytes(p)))
