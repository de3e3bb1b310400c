"""ctypes binding of the CPU oracle (oracle/_build/librv_oracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "librv_oracle.so")

E_NAMES = {0: "OK", 1: "WITNESS_INVALID", 2: "WITNESS_SHORT", 3: "WIRE_OOB", 4: "PROOF_MALFORMED", 5: "BAD_OP", 6: "NOMEM"}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code} ({E_NAMES.get(code, '?')})")
        self.code = code


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.rvo_gf2_reconstruct.restype = C.c_uint64
        _lib.rvo_gf2_reconstruct.argtypes = [C.c_uint64]
        _lib.rvo_gf2_recon_unpack.restype = C.c_size_t
        _lib.rvo_gf2_share_unpack_selected.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _wit(wit_gf2, wit_z64):
    g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
    z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
    return g, z


def prove(prog, wit_gf2, wit_z64, wire_counts, seeds, threads=8) -> bytes:
    """wire_counts = (z64, gf2) like the reference's Proof::new."""
    g, z = _wit(wit_gf2, wit_z64)
    seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(256, 16)
    out = C.c_void_p()
    n = C.c_size_t()
    rc = lib().rvo_prove(_p(prog), C.c_size_t(len(prog)), _p(g), C.c_size_t(len(g)), _p(z), C.c_size_t(len(z)),
                         C.c_size_t(wire_counts[0]), C.c_size_t(wire_counts[1]), _p(seeds), C.c_int(threads),
                         C.byref(out), C.byref(n))
    if rc:
        raise OracleError(rc)
    data = C.string_at(out, n.value)
    lib().rvo_free(out)
    return data


def verify(prog, wire_counts, proof: bytes, threads=8, strict=False) -> bool:
    ok = C.c_int()
    buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
    rc = lib().rvo_verify_ex(_p(prog), C.c_size_t(len(prog)), C.c_size_t(wire_counts[0]), C.c_size_t(wire_counts[1]),
                             buf, C.c_size_t(len(proof)), C.c_int(threads), C.c_int(1 if strict else 0), C.byref(ok))
    if rc:
        raise OracleError(rc)
    return bool(ok.value)


def commit(prog, wit_gf2, wit_z64, wire_counts, seeds, threads=8):
    """-> (h[256,32], streams[256,4,32], comm[32])"""
    g, z = _wit(wit_gf2, wit_z64)
    seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(256, 16)
    h = np.zeros((256, 32), np.uint8)
    st = np.zeros((256, 4, 32), np.uint8)
    comm = np.zeros(32, np.uint8)
    rc = lib().rvo_commit(_p(prog), C.c_size_t(len(prog)), _p(g), C.c_size_t(len(g)), _p(z), C.c_size_t(len(z)),
                          C.c_size_t(wire_counts[0]), C.c_size_t(wire_counts[1]), _p(seeds), C.c_int(threads),
                          _p(h), _p(st), _p(comm))
    if rc:
        raise OracleError(rc)
    return h, st, comm


def challenge(comm) -> np.ndarray:
    comm = np.ascontiguousarray(np.asarray(comm, dtype=np.uint8))
    omit = np.zeros(256, np.uint8)
    lib().rvo_challenge(_p(comm), _p(omit))
    return omit


def expand_seed(seed) -> np.ndarray:
    seed = np.ascontiguousarray(np.asarray(seed, dtype=np.uint8))
    keys = np.zeros((8, 16), np.uint8)
    lib().rvo_expand_seed(_p(seed), _p(keys))
    return keys


def sharegen_gf2(keys, omit, n) -> np.ndarray:
    keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint8)).reshape(8, 8, 16)
    omit = np.ascontiguousarray(np.asarray(omit, dtype=np.uint32))
    out = np.zeros(n, np.uint64)
    lib().rvo_sharegen_gf2(_p(keys), _p(omit), C.c_size_t(n), _p(out))
    return out


def sharegen_z64(keys, omit, n) -> np.ndarray:
    keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint8)).reshape(8, 8, 16)
    omit = np.ascontiguousarray(np.asarray(omit, dtype=np.uint32))
    out = np.zeros((n, 8, 8), np.uint64)
    lib().rvo_sharegen_z64(_p(keys), _p(omit), C.c_size_t(n), _p(out))
    return out


def gf2_reconstruct(x: int) -> int:
    return int(lib().rvo_gf2_reconstruct(C.c_uint64(x)))


def group_wire_values(prog, wit_gf2, wit_z64, wire_counts, seeds8, gf2_wire=None, z64_wire=None):
    g, z = _wit(wit_gf2, wit_z64)
    seeds8 = np.ascontiguousarray(np.asarray(seeds8, dtype=np.uint8)).reshape(8, 16)
    g_out = C.c_uint64()
    z_out = np.zeros(8, np.uint64)
    rc = lib().rvo_group_wire_values(_p(prog), C.c_size_t(len(prog)), _p(g), C.c_size_t(len(g)), _p(z), C.c_size_t(len(z)),
                                     C.c_size_t(wire_counts[0]), C.c_size_t(wire_counts[1]), _p(seeds8),
                                     C.c_uint32(gf2_wire or 0), C.byref(g_out) if gf2_wire is not None else None,
                                     C.c_uint32(z64_wire or 0), _p(z_out) if z64_wire is not None else None)
    if rc:
        raise OracleError(rc)
    return (g_out.value if gf2_wire is not None else None), (z_out if z64_wire is not None else None)


def gf2_recon_pack(src, selected):
    src = np.ascontiguousarray(np.asarray(src, dtype=np.uint64))
    sel = np.ascontiguousarray(np.asarray(selected, dtype=np.uint8))
    cap = len(src) // 8 + 2
    dst = np.zeros((8, cap), np.uint8)
    lens = (C.c_size_t * 8)()
    lib().rvo_gf2_recon_pack(_p(src), C.c_size_t(len(src)), _p(sel), _p(dst), C.c_size_t(cap), lens)
    return [bytes(dst[i, : lens[i]]) for i in range(8)]


def gf2_recon_unpack(vecs):
    L = len(vecs[0])
    src = np.zeros((8, max(L, 1)), np.uint8)
    for i in range(8):
        src[i, :L] = np.frombuffer(vecs[i], np.uint8)[:L]
    src = np.ascontiguousarray(src[:, :L]) if L else np.zeros((8, 0), np.uint8)
    dst = np.zeros(8 * L + 8, np.uint64)
    n = lib().rvo_gf2_recon_unpack(_p(src), C.c_size_t(L), _p(dst))
    return dst[:n]


def gf2_share_pack_selected(src, selected):
    src = np.ascontiguousarray(np.asarray(src, dtype=np.uint64))
    sel = np.ascontiguousarray(np.asarray(selected, dtype=np.uint32))
    cap = len(src) // 8 + 2
    dst = np.zeros((8, cap), np.uint8)
    lens = (C.c_size_t * 8)()
    lib().rvo_gf2_share_pack_selected(_p(src), C.c_size_t(len(src)), _p(sel), _p(dst), C.c_size_t(cap), lens)
    return [bytes(dst[i, : lens[i]]) for i in range(8)]


def gf2_share_unpack_selected(vecs, selected):
    L = len(vecs[0])
    src = np.zeros((8, L), np.uint8)
    for i in range(8):
        src[i] = np.frombuffer(vecs[i], np.uint8)
    src = np.ascontiguousarray(src)
    sel = np.ascontiguousarray(np.asarray(selected, dtype=np.uint32))
    dst = np.zeros(8 * L + 8, np.uint64)
    n = lib().rvo_gf2_share_unpack_selected(_p(src), C.c_size_t(L), _p(sel), _p(dst))
    return dst[:n]
