#!/usr/bin/env python3
"""rv_prove_batch on the 10^7-gate circuit: the proofs-in-flight path (worker threads, default for large circuits) against the
fused path (every launch carries the whole batch; RV_BATCH_BIG_GATES=huge), host bytes in, host proof bytes out."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import circuits  # noqa: E402
import reverie_amd  # noqa: E402

prog, wit, wc, st = circuits.layered_gf2(layers=int(os.environ.get("LAYERS", 153)))
c = reverie_amd.Circuit(prog, wc, whole_prover=os.environ.get("HINT", "1") != "0")
rng = np.random.default_rng(3)
for B in [int(x) for x in os.environ.get("BATCHES", "2,4").split(",")]:
    seeds = rng.integers(0, 256, (B, 256, 16), dtype=np.uint8)
    wits = np.tile(np.asarray(wit, np.uint8), (B, 1))
    reverie_amd.Proof.new_batch(c, wits, seeds=seeds)
    n = 4
    t = time.perf_counter()
    got = None
    for _ in range(n):
        del got  # hand the page-locked buffers back before the next call needs them
        got = reverie_amd.Proof.new_batch(c, wits, seeds=seeds)
    dt = time.perf_counter() - t
    ok = bytes(got[B - 1]) == bytes(reverie_amd.Proof.new(c, wit, [], seeds=seeds[B - 1]))
    print(json.dumps({"batch": B, "ms_per_proof": dt / (n * B) * 1e3, "and_per_s": st["and"] * n * B / dt, "equals_single": ok,
                      "path": "fused" if os.environ.get("RV_BATCH_BIG_GATES") else "worker threads"}))
