#!/usr/bin/env python3
"""Prover / verifier wall times through the host-bytes entry points (rv_prove / rv_verify, PCIe included)
for the headline circuit and its all-AND variant.  Prints one JSON line per variant."""
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

seeds = np.random.default_rng(1).integers(0, 256, (256, 16), dtype=np.uint8)
for p_and in (0.5, 1.0):
    prog, wit, wc, st = circuits.layered_gf2(p_and=p_and)
    c = reverie_amd.Circuit(prog, wc)
    proof = reverie_amd.Proof.new(c, wit, [], seeds=seeds)
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); proof = reverie_amd.Proof.new(c, wit, [], seeds=seeds); t.append(time.perf_counter() - t0)
    import ctypes as C

    from reverie_amd import _lib

    def phases(fn, n=3):
        """wall times + the library's own HIP-event phase times (device side only) over n calls"""
        L = _lib.lib()
        L.rv_ctx_profile(c.ctx.handle, 1, 1, None)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        prof = _lib.Profile()
        L.rv_ctx_profile(c.ctx.handle, 0, 0, C.byref(prof))
        return ts, r, {nm: round(prof.ms[i] / n, 3) for i, nm in enumerate(_lib.PHASES)}

    t, proof, prove_phases = phases(lambda: reverie_amd.Proof.new(c, wit, [], seeds=seeds))
    v, ok, verify_phases = phases(lambda: proof.verify(c))
    copy = reverie_amd.Proof(bytes(proof))  # ordinary pageable memory, as a proof read from disk would be
    vp, okp, _ = phases(lambda: copy.verify(c))
    ok = ok and okp
    print(json.dumps({"p_and": p_and, "and": st["and"], "gates": st["gates"], "proof_bytes": len(proof), "prove_ms_host": min(t) * 1e3,
                      "verify_ms_host": min(v) * 1e3, "verify_ms_host_pageable_input": min(vp) * 1e3, "prove_and_per_s_host": st["and"] / min(t), "verify_and_per_s_host": st["and"] / min(v),
                      "verify_ok": ok, "prove_device_phases_ms": prove_phases, "verify_device_phases_ms": verify_phases}))
    c.close()
