#!/usr/bin/env python3
"""Aggregate prover throughput with B independent proofs in flight (one context + thread each),
device-resident openings, same circuit.  Tests whether the VALU-bound and HBM-bound phases of
different proofs overlap on the chip."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import circuits  # noqa: E402
import reverie_amd  # noqa: E402
from reverie_amd.dist import HipShardBackend, prove_sharded  # noqa: E402

torch.cuda.set_device(0)
prog, wit, wc, st = circuits.layered_gf2()
seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
for B in (1, 2, 3):
    ctxs = [reverie_amd.Context(0) for _ in range(B)]
    bes = [HipShardBackend(reverie_amd.Circuit(prog, wc, c)) for c in ctxs]

    def worker(i, n):
        torch.cuda.set_device(0)
        for _ in range(n):
            prove_sharded(bes[i], wit, [], seeds, device_resident=True)

    def run(n):
        th = [threading.Thread(target=worker, args=(i, n)) for i in range(B)]
        t = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        torch.cuda.synchronize()
        return time.perf_counter() - t

    run(2)
    dt = run(8)
    print(json.dumps({"in_flight": B, "and_per_s": st["and"] * 8 * B / dt, "ms_per_proof_aggregate": dt / (8 * B) * 1e3}))
