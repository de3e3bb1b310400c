"""Randomised parity run for the fused Z64 prover (csrc/aes.hip: k_z64_fused): random Z64 programs of random shapes (every eligible op,
wire reuse, Input gates anywhere, even / odd Input counts -- odd ones must fall back), layered circuits of random widths, whole proofs
and 32 / 64 / 128-repetition shards, against the oracle.   python tools/fuzz_z64f.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import circuits, oracle_lib, reverie_amd
from test_gpu_z64_fused import random_z64
from reverie_amd.ops import Z64, program

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = reverie_amd.Context(0)
os.environ["RV_Z64_FUSED"] = "1"
t0 = time.time()
bad = 0
for case in range(n_cases):
    kind = rng.random()
    if kind < 0.5:
        prog, w64, wc = random_z64(rng, n_in=int(rng.integers(1, 40)), n_gates=int(rng.integers(5, 4000)), n_wires=int(rng.integers(3, 400)),
                                   p_assert=float(rng.choice([0.0, 0.04, 0.1])))
    elif kind < 0.8:
        prog, w64, wc, _ = circuits.layered_z64(n_in=2 * int(rng.integers(1, 200)), width=int(rng.integers(1, 700)), n_mul=int(rng.integers(1, 6000)),
                                                seed=int(rng.integers(1, 1 << 62)), fold_to=int(rng.integers(1, 17)), recycle=bool(rng.random() < 0.3))
    else:
        # Input pairs sprinkled between the other gates (several cipher-block runs generated the plain way)
        ops, w64, vals, nxt = [], [], [], 0
        for blk in range(int(rng.integers(1, 40))):
            for _ in range(2 * int(rng.integers(1, 5))):
                w = int(rng.integers(0, 1 << 62)); ops.append(Z64.Input(nxt)); w64.append(w); vals.append(w); nxt += 1
            for _ in range(int(rng.integers(0, 60))):
                a, b = int(rng.integers(0, nxt)), int(rng.integers(0, nxt))
                if rng.random() < 0.6:
                    ops.append(Z64.Mul(nxt, a, b)); vals.append((vals[a] * vals[b]) & ((1 << 64) - 1))
                else:
                    ops.append(Z64.Sub(nxt, a, b)); vals.append((vals[a] - vals[b]) & ((1 << 64) - 1))
                nxt += 1
        ops.append(Z64.SubConst(nxt, nxt - 1, vals[-1])); ops.append(Z64.AssertZero(nxt)); nxt += 1
        prog, wc = program(ops), (nxt, 0)
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    try:
        want = oracle_lib.prove(prog, [], w64, wc, seeds, threads=8)
    except oracle_lib.OracleError:
        continue
    c = reverie_amd.Circuit(prog, wc, ctx)
    ok = True
    if rng.random() < 0.7:
        proof = reverie_amd.Proof.new(c, [], w64, seeds=seeds)
        ok = bytes(proof) == want and bool(proof.verify(c))
    else:
        from reverie_amd.dist import HipShardBackend, assemble
        from reverie_amd.proof import challenge, combine_digests
        reps = int(rng.choice([32, 64, 128]))
        be = HipShardBackend(c)
        shards = [be.commit([], w64, seeds[b:b + reps], b, reps) for b in range(0, 256, reps)]
        try:
            comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
            parts = [be.open(s, challenge(comm))[:2] for s in shards]
        finally:
            for s in shards:
                be.destroy(s)
        ok = assemble(comm, parts) == want
    if not ok:
        bad += 1
        print("MISMATCH case", case, "kind %.2f" % kind, "ops", len(prog), flush=True)
    c.close()
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
