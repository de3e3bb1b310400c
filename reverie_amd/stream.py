"""Host-side mirror of the streaming prover (include/reverie_amd.h: rv_stream_*).

The reference's README promises a streaming interface (/root/reference/README.md:14); its `Proof::new`
(src/proof/mod.rs:119-222) takes the whole gate list at once.  `StreamingProver` takes the gate list in pieces, twice:

    sp = StreamingProver((z64_wires, gf2_wires), seeds=seeds)
    for ops, w2, w64 in pieces: sp.feed(ops, w2, w64)     # pass 1
    comm = sp.commit()
    for ops, w2, w64 in pieces: sp.feed(ops, w2, w64)     # pass 2 (the same ops again)
    proof = sp.finish()                                    # == Proof.new(all ops, all witness, seeds=seeds)

Device memory is bounded by the wire counts, one piece's working set and the proof; everything runs through the C-ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .ops import OP_DTYPE, TOTAL_REPS, program
from .proof import Context, Proof, _ptr


class StreamingProver:
    def __init__(self, wire_counts: Tuple[int, int], seeds=None, max_chunk_ops: int = 0, ctx: Optional[Context] = None):
        self.ctx = ctx or Context.default()
        self.handle = C.c_void_p()
        s = None
        if seeds is not None:
            s = np.ascontiguousarray(np.frombuffer(bytes(seeds), np.uint8) if isinstance(seeds, (bytes, bytearray))
                                     else np.asarray(seeds, dtype=np.uint8)).reshape(TOTAL_REPS, 16)
        _lib.check(_lib.lib().rv_stream_begin(self.ctx.handle, C.c_size_t(int(wire_counts[0])), C.c_size_t(int(wire_counts[1])), _ptr(s),
                                              C.c_size_t(max_chunk_ops), C.byref(self.handle)))

    def feed(self, ops, wit_gf2: Sequence[int] = (), wit_z64: Sequence[int] = ()):
        ops = program(ops) if len(ops) else np.zeros(0, OP_DTYPE)
        g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
        z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
        _lib.check(_lib.lib().rv_stream_feed(self.handle, _ptr(ops), C.c_size_t(len(ops)), _ptr(g), C.c_size_t(len(g)), _ptr(z), C.c_size_t(len(z))))

    def same_cuts(self):
        """rv_stream_same_cuts: pass 2 will be fed in pass 1's pieces -- pass 1 keeps the last chunks' transcripts within
        RV_STREAM_KEEP_MB and pass 2 takes their openings from them instead of running them again"""
        _lib.check(_lib.lib().rv_stream_same_cuts(self.handle))

    def commit(self) -> bytes:
        comm = np.zeros(32, np.uint8)
        _lib.check(_lib.lib().rv_stream_commit(self.handle, _ptr(comm)))
        return comm.tobytes()

    def finish(self) -> Proof:
        out, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().rv_stream_finish(self.handle, C.byref(out), C.byref(n)))
        return Proof(_owned=(C.c_void_p(out.value), n.value))

    @property
    def info(self) -> dict:
        si = _lib.StreamInfo()
        _lib.check(_lib.lib().rv_stream_get_info(self.handle, C.byref(si)))
        return {n: int(getattr(si, n)) for n, _ in si._fields_}

    def close(self):
        if self.handle:
            if self.ctx.handle:
                _lib.lib().rv_stream_abort(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_streaming(ops, wit_gf2, wit_z64, wire_counts: Tuple[int, int], seeds=None, max_chunk_ops: int = 0,
                    ctx: Optional[Context] = None) -> Tuple[Proof, dict]:
    """rv_prove_streaming: both passes over an op array in host memory -> (Proof, stream info)"""
    ctx = ctx or Context.default()
    ops = program(ops) if len(ops) else np.zeros(0, OP_DTYPE)
    g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
    z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
    s = None
    if seeds is not None:
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(TOTAL_REPS, 16)
    out, n = C.c_void_p(), C.c_size_t()
    si = _lib.StreamInfo()
    _lib.check(_lib.lib().rv_prove_streaming(ctx.handle, _ptr(ops), C.c_size_t(len(ops)), C.c_size_t(int(wire_counts[0])),
                                             C.c_size_t(int(wire_counts[1])), _ptr(g), C.c_size_t(len(g)), _ptr(z), C.c_size_t(len(z)), _ptr(s),
                                             C.c_size_t(max_chunk_ops), C.byref(out), C.byref(n), C.byref(si)))
    return Proof(_owned=(C.c_void_p(out.value), n.value)), {k: int(getattr(si, k)) for k, _ in si._fields_}


class StreamingVerifier:
    """Proof::verify with bounded device memory: the ops are fed in pieces, once (rv_stream_verify_begin / feed / finish).

        sv = StreamingVerifier((z64_wires, gf2_wires), proof)
        for ops in pieces: sv.feed(ops)
        ok = sv.finish()            # == proof.verify(all ops, (z64_wires, gf2_wires))
    """

    def __init__(self, wire_counts: Tuple[int, int], proof, max_chunk_ops: int = 0, ctx: Optional[Context] = None):
        self.ctx = ctx or Context.default()
        self.handle = C.c_void_p()
        self._proof = proof if isinstance(proof, Proof) else Proof(bytes(proof))  # (kept alive: the stream reads it until finish)
        buf, n = self._proof._buffer()
        _lib.check(_lib.lib().rv_stream_verify_begin(self.ctx.handle, C.c_size_t(int(wire_counts[0])), C.c_size_t(int(wire_counts[1])), buf,
                                                     C.c_size_t(n), C.c_size_t(max_chunk_ops), C.byref(self.handle)))

    def feed(self, ops):
        ops = program(ops) if len(ops) else np.zeros(0, OP_DTYPE)
        _lib.check(_lib.lib().rv_stream_feed(self.handle, _ptr(ops), C.c_size_t(len(ops)), None, C.c_size_t(0), None, C.c_size_t(0)))

    def finish(self, strict: bool = True) -> bool:
        ok = C.c_int()
        _lib.check(_lib.lib().rv_stream_verify_finish(self.handle, C.c_uint32(0 if strict else _lib.RV_VERIFY_REFERENCE_COMPAT), C.byref(ok)))
        return bool(ok.value)

    info = StreamingProver.info
    close = StreamingProver.close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def verify_streaming(ops, wire_counts: Tuple[int, int], proof, strict: bool = True, max_chunk_ops: int = 0,
                     ctx: Optional[Context] = None) -> Tuple[bool, dict]:
    """rv_verify_streaming: one pass over an op array in host memory -> (ok, stream info)"""
    ctx = ctx or Context.default()
    ops = program(ops) if len(ops) else np.zeros(0, OP_DTYPE)
    pr = proof if isinstance(proof, Proof) else Proof(bytes(proof))
    buf, n = pr._buffer()
    ok = C.c_int()
    si = _lib.StreamInfo()
    _lib.check(_lib.lib().rv_verify_streaming(ctx.handle, _ptr(ops), C.c_size_t(len(ops)), C.c_size_t(int(wire_counts[0])), C.c_size_t(int(wire_counts[1])),
                                              buf, C.c_size_t(n), C.c_uint32(0 if strict else _lib.RV_VERIFY_REFERENCE_COMPAT), C.c_size_t(max_chunk_ops),
                                              C.byref(ok), C.byref(si)))
    return bool(ok.value), {k: int(getattr(si, k)) for k, _ in si._fields_}
