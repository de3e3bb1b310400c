#!/usr/bin/env python3
"""rv_prove_batch on the 10^7-gate circuit, call after call the way bench.py times it (the previous call's proofs held while the next one
runs): per-call time, so that the slow calls of VERDICT r5 weak #6 show up one by one.  RV_BATCH_STATS=1 prints every proof's span inside
its call, RV_PINNED_TRACE=1 every page-locked allocation / release with its duration."""
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

if os.environ.get("TORCH"):  # what bench.py has beside the library: PyTorch's streams on the same HIP runtime
    import torch

    torch.cuda.synchronize()
    _x = torch.zeros(1 << 20, device="cuda")
prog, wit, wc, st = circuits.layered_gf2(layers=int(os.environ.get("LAYERS", 153)))
c = reverie_amd.Circuit(prog, wc, whole_prover=True)
B = int(os.environ.get("B", 8))
N = int(os.environ.get("CALLS", 20))
rng = np.random.default_rng(3)
seeds = rng.integers(0, 256, (B, 256, 16), dtype=np.uint8)
wits = np.tile(np.asarray(wit, np.uint8), (B, 1))
one = reverie_amd.Proof.new(c, wit, [], seeds=seeds[0])
ts = []
if os.environ.get("PROFILE"):  # bench.py times its proofs with the context's phase timers on
    import ctypes as C

    from reverie_amd import _lib

    L = _lib.lib()
    ctx = reverie_amd.Context.default()
    L.rv_ctx_profile(ctx.handle, 1, 1, None)
    for _ in range(int(os.environ.get("SINGLES", 20))):
        one = reverie_amd.Proof.new(c, wit, [], seeds=seeds[0])
    prof = _lib.Profile()
    L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
for _ in range(int(os.environ.get("SINGLES", 5))):
    t = time.perf_counter()
    one = reverie_amd.Proof.new(c, wit, [], seeds=seeds[0])
    ts.append(time.perf_counter() - t)
print("single rv_prove ms:", " ".join(f"{x * 1e3:.2f}" for x in ts), flush=True)
for _ in range(int(os.environ.get("COMPILES_BEFORE", 0))):  # bench.py compiles and releases the circuit a few times right before its batch leg
    c2 = reverie_amd.Circuit(prog, wc, whole_prover=True)
    c2.close()
proofs = None
ms = []
for k in range(N):
    sys.stderr.write(f"--- call {k}\n")
    sys.stderr.flush()
    t = time.perf_counter()
    proofs = reverie_amd.Proof.new_batch(c, wits, seeds=seeds)
    ms.append((time.perf_counter() - t) / B * 1e3)
print("ms per proof, call by call:", " ".join(f"{x:.2f}" for x in ms))
s = sorted(ms[2:])
print(json.dumps({"batch": B, "calls": N, "median": s[len(s) // 2], "min": s[0], "max": s[-1], "equals_single": bytes(proofs[0]) == bytes(one)}))
