"""Run by tests/test_gpu_multirank.py in a FRESH process with RV_RCCL_PATH = the rccl test shim (tests/rccl_shim): the library's
own multi-rank path -- rv_comm_create_all + rv_prove_multi (a host thread per rank, rv_prove_sharded on each: all-gather of the
digests, device-side challenge, grouped send / recv of the openings to rank 0; csrc/comm.inc) -- with 2, 4 and 8 ranks that share
the one GPU.  Every sharded proof must equal rv_prove's bytes and the oracle's.  Prints one JSON line."""
import ctypes as C
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import circuits  # noqa: E402
import oracle_lib  # noqa: E402
import reverie_amd  # noqa: E402
from reverie_amd import _lib  # noqa: E402


def p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and len(a) else None


def main():
    assert os.environ.get("RV_RCCL_PATH"), "the worker must run against the shim"
    L = _lib.lib()
    oracle_lib.build()
    seeds = np.zeros((256, 16), np.uint8)  # seed[r] = BLAKE3("rv-seed" || LE32(r))[0..16] (tests/conftest.py: rule_seeds)
    buf = C.create_string_buffer(32)
    for r in range(256):
        d = b"rv-seed" + struct.pack("<I", r)
        oracle_lib.lib().rvo_blake3_hash(d, C.c_size_t(len(d)), buf)
        seeds[r] = np.frombuffer(buf.raw[:16], np.uint8)
    rng = np.random.default_rng(77)
    cases = []
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=400)
    cases.append(("mixed", prog, w2, w64, wc))
    prog, wit, wc, st = circuits.layered_gf2(n_in=300, width=4096, layers=6)
    cases.append(("layered", prog, wit, [], wc))
    full = bool(os.environ.get("MULTI_FULL"))
    if full:
        # BASELINE config 4 at full size (10 027 008 gates): every rank holds the replicated gate stream and a shard of the
        # repetitions; the sharded proof against the oracle's and rv_prove's bytes
        prog, wit, wc, st = circuits.layered_gf2()
        cases = [("config4", prog, wit, [], wc)]
    worlds = [int(x) for x in (sys.argv[1:] or ["2", "4", "8"])]
    res = {}
    for name, prog, w2, w64, wc in cases:
        want = oracle_lib.prove(prog, w2, w64, wc, seeds, threads=32)
        single = bytes(reverie_amd.Proof.new(prog, w2, w64, wc, seeds=seeds))
        assert single == want, name
        g = np.ascontiguousarray(np.asarray(w2, np.uint8))
        z = np.ascontiguousarray(np.asarray(w64, np.uint64))
        for n in worlds:
            ctxs = [reverie_amd.Context(0) for _ in range(n)]
            circs = [reverie_amd.Circuit(prog, wc, cx) for cx in ctxs]
            hc = (C.c_void_p * n)(*[cx.handle for cx in ctxs])
            cm = (C.c_void_p * n)()
            _lib.check(L.rv_comm_create_all(hc, C.c_int(n), cm))
            hcirc = (C.c_void_p * n)(*[c.handle for c in circs])
            for it in range(1 if full else 2):
                out, ln = C.c_void_p(), C.c_size_t()
                _lib.check(L.rv_prove_multi(cm, hcirc, C.c_int(n), p(g), C.c_size_t(len(g)), p(z), C.c_size_t(len(z)), p(seeds), C.byref(out),
                                            C.byref(ln)))
                got = C.string_at(out, ln.value)
                L.rv_free(out)
                res["%s/%d/%d" % (name, n, it)] = got == want
            if full:
                for i in range(n):
                    L.rv_comm_destroy(C.c_void_p(cm[i]))
                for c in circs:
                    c.close()
                for cx in ctxs:
                    cx.close()
                continue
            # seeds = NULL: drawn once for all ranks; the proof must verify
            out, ln = C.c_void_p(), C.c_size_t()
            _lib.check(L.rv_prove_multi(cm, hcirc, C.c_int(n), p(g), C.c_size_t(len(g)), p(z), C.c_size_t(len(z)), None, C.byref(out), C.byref(ln)))
            pr = reverie_amd.Proof(C.string_at(out, ln.value))
            L.rv_free(out)
            res["%s/%d/os-seeds-verify" % (name, n)] = bool(pr.verify(prog, wc)) and bool(oracle_lib.verify(prog, wc, bytes(pr)))
            # an invalid witness is reported by every rank, nobody hangs
            bad = g.copy()
            bad[:7] ^= 1
            out, ln = C.c_void_p(), C.c_size_t()
            rc = L.rv_prove_multi(cm, hcirc, C.c_int(n), p(bad), C.c_size_t(len(bad)), p(z), C.c_size_t(len(z)), p(seeds), C.byref(out), C.byref(ln))
            res["%s/%d/invalid-witness" % (name, n)] = rc == 1
            for i in range(n):
                L.rv_comm_destroy(C.c_void_p(cm[i]))
            for c in circs:
                c.close()
            for cx in ctxs:
                cx.close()
    print(json.dumps(res))
    return 0 if all(res.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
