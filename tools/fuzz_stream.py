"""Randomised parity run for the streaming prover's multi-piece feeds (pieces compiled ahead on worker threads, ShareGen
phases from count_masks, transcript offsets by relocate_chunk): random GF(2) / mixed programs with B2A through
prove_streaming with small chunks, against the oracle.   python tools/fuzz_stream.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import circuits, oracle_lib, reverie_amd
from reverie_amd.stream import prove_streaming, verify_streaming

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = reverie_amd.Context(0)
t0 = time.time()
bad = done = 0
for case in range(n_cases):
    if rng.random() < 0.5:
        prog, w2, wc = circuits.random_gf2(rng, n_in=int(rng.integers(1, 300)), n_gates=int(rng.integers(1500, 9000)), n_wires=int(rng.integers(4, 600)))
        w64 = []
    else:
        prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(1500, 6000)))
        wc = (12, 90)  # (the program's SizeHint grows the wire vectors to this: a stream is sized when it begins)
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    try:
        want = oracle_lib.prove(prog, w2, w64, wc, seeds, threads=8)
    except oracle_lib.OracleError:
        continue
    os.environ["RV_STREAM_THREADS"] = str(int(rng.choice([1, 2, 3, 6])))
    proof, info = prove_streaming(prog, w2, w64, wc, seeds=seeds, max_chunk_ops=int(rng.integers(1024, 2500)), ctx=ctx)
    done += 1
    if bytes(proof) != want:
        bad += 1
        print("MISMATCH case", case, "ops", len(prog), "chunks", info["chunks"], "threads", os.environ["RV_STREAM_THREADS"], flush=True)
    # the streaming verifier (other chunk cuts than the prover's) against the resident one: the proof, and the proof with a flipped byte
    flip = bytearray(want)
    flip[int(rng.integers(0, len(flip)))] ^= 1 << int(rng.integers(0, 8))
    for pb in (want, bytes(flip)):
        def run(f):
            try:
                return f()
            except reverie_amd.ReverieError as e:
                return ("error", e.code)
        a = run(lambda: verify_streaming(prog, wc, pb, max_chunk_ops=int(rng.integers(700, 3000)), ctx=ctx)[0])
        b = run(lambda: reverie_amd.Proof(pb).verify(prog, wc, ctx=ctx))
        if a != b or (pb is want and a is not True):
            bad += 1
            print("VERIFY MISMATCH case", case, "streaming", a, "resident", b, flush=True)
print(f"{done} cases run, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
