"""Randomised parity run for the verifier's round-4 paths -- MODE_VERIFY_C (GF(2): one u64 of corrections per row) and
k_z64_fused<VERIFY> with its quad-group split: random eligible programs, valid proofs, bit flips anywhere in the proof, both the
reference-compatible and the strict answer against the oracle's.   python tools/fuzz_verify.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import circuits, oracle_lib, reverie_amd
from reverie_amd import _lib
from test_gpu_z64_fused import random_z64

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = reverie_amd.Context(0)
L = _lib.lib()
t0 = time.time()
bad_cases = 0
vc0 = L.rv_hook_verify_vc_count()
for case in range(n_cases):
    kind = rng.random()
    if kind < 0.5:  # wide GF(2) levels, one base per wire: MODE_VERIFY_C
        width = int(rng.integers(300, 5000))
        prog, w2, wc, _ = circuits.layered_gf2(n_in=int(rng.integers(1, 600)), width=width, layers=int(rng.integers(2, 12)),
                                               p_and=float(rng.choice([0.3, 0.5, 1.0])), seed=int(rng.integers(1, 1 << 62)), fold_to=width)
        w64 = []
    elif kind < 0.75:
        prog, w64, wc = random_z64(rng, n_in=2 * int(rng.integers(1, 20)), n_gates=int(rng.integers(50, 3000)), n_wires=int(rng.integers(8, 300)))
        w2 = []
    else:  # Z64 proofs beyond 4 MB: the side-stream copy and the quad-group split
        prog, w64, wc, _ = circuits.layered_z64(n_in=64, width=int(rng.integers(500, 3000)), n_mul=int(rng.integers(7000, 16000)), seed=int(rng.integers(1, 1 << 62)))
        w2 = []
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    try:
        good = oracle_lib.prove(prog, w2, w64, wc, seeds, threads=8)
    except oracle_lib.OracleError:
        continue
    c = reverie_amd.Circuit(prog, wc, ctx)
    ok = bool(reverie_amd.Proof(good).verify(c))
    for pos in rng.integers(0, len(good), 6):
        bad = bytearray(good)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        for strict in (False, True):
            try:
                want = (oracle_lib.verify(prog, wc, bytes(bad), strict=strict), None)
            except oracle_lib.OracleError as e:
                want = (None, e.code)
            try:
                got = (bool(reverie_amd.Proof(bytes(bad)).verify(c, strict=strict)), None)
            except reverie_amd.ReverieError as e:
                got = (None, e.code)
            ok = ok and got == want
    if not ok:
        bad_cases += 1
        print("MISMATCH case", case, "kind %.2f" % kind, "ops", len(prog), flush=True)
    c.close()
print(f"{n_cases} cases, {bad_cases} mismatches, {L.rv_hook_verify_vc_count() - vc0} verifications took MODE_VERIFY_C, {time.time() - t0:.1f} s")
sys.exit(1 if bad_cases else 0)
