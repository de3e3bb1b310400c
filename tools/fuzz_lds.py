"""Randomised parity run for the LDS runs (csrc/ldsrun.*): many random GF(2) / mixed programs of random shapes, prover and
verifier against the oracle, both slice widths.   python tools/fuzz_lds.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import circuits, oracle_lib, reverie_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = reverie_amd.Context(0)
t0 = time.time()
bad = 0
for case in range(n_cases):
    os.environ["RV_LDS_QS"] = str(int(rng.choice([0, 1, 2, 4])))
    if rng.random() < 0.8:
        prog, w2, wc = circuits.random_gf2(rng, n_in=int(rng.integers(1, 300)), n_gates=int(rng.integers(20, 6000)),
                                           n_wires=int(rng.integers(4, 600)), p_assert=float(rng.choice([0.0, 0.05, 0.2])))
        w64 = []
    else:
        prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(50, 900)))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    try:
        want = oracle_lib.prove(prog, w2, w64, wc, seeds, threads=8)
    except oracle_lib.OracleError:
        continue
    c = reverie_amd.Circuit(prog, wc, ctx, whole_prover=bool(rng.random() < 0.5))
    proof = reverie_amd.Proof.new(c, w2, w64, seeds=seeds)
    ok = bytes(proof) == want and bool(proof.verify(c))
    if not ok:
        bad += 1
        print("MISMATCH case", case, "qs", os.environ["RV_LDS_QS"], "ops", len(prog), flush=True)
    c.close()
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
