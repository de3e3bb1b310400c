"""Host-side mirror of the reference's `Proof` API for this path.

    reference (src/proof/mod.rs)                         here
    Proof::new(circuit, wit_gf2, wit_z64, (z64, gf2))    Proof.new(circuit, wit_gf2, wit_z64, (z64, gf2))
    proof.verify(circuit, (z64, gf2)) -> bool            proof.verify(circuit, (z64, gf2)) -> bool
    bincode::serialize(&proof)                           bytes(proof)     (byte-identical layout)
    bincode::deserialize(bytes)                          Proof(bytes)

`circuit` is an rv_op array (reverie_amd.ops.program) or an already compiled `Circuit`.
Everything runs through the C-ABI (include/reverie_amd.h) on the GPU; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .ops import OP_DTYPE, TOTAL_REPS, program


def _ptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Context:
    """One GPU (rv_ctx).  A default context on device 0 (or LOCAL_RANK) is created lazily."""

    _default = None

    def __init__(self, device: int = 0):
        self.handle = C.c_void_p()
        _lib.check(_lib.lib().rv_ctx_create(C.c_int(device), C.byref(self.handle)))
        self.device = device

    @classmethod
    def default(cls) -> "Context":
        if cls._default is None:
            import os

            cls._default = Context(int(os.environ.get("LOCAL_RANK", "0")))
        return cls._default

    def sync(self):
        _lib.check(_lib.lib().rv_ctx_sync(self.handle))

    def close(self):
        if self.handle:
            _lib.lib().rv_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Circuit:
    """A gate stream compiled (levelised) and resident in HBM (rv_circuit)."""

    def __init__(self, ops, wire_counts: Tuple[int, int], ctx: Optional[Context] = None, whole_prover: bool = False):
        """whole_prover: the circuit will mostly serve whole proofs on one GPU (Proof.new / new_batch) -- the
        RV_COMPILE_WHOLE_PROVER hint of rv_circuit_compile_ex; any use of the circuit still gives identical bytes."""
        self.ctx = ctx or Context.default()
        self.ops = program(ops) if len(ops) else np.zeros(0, OP_DTYPE)
        self.wire_counts = (int(wire_counts[0]), int(wire_counts[1]))  # (z64, gf2), proof/mod.rs:125
        self.handle = C.c_void_p()
        _lib.check(_lib.lib().rv_circuit_compile_ex(self.ctx.handle, _ptr(self.ops), C.c_size_t(len(self.ops)),
                                                    C.c_size_t(self.wire_counts[0]), C.c_size_t(self.wire_counts[1]),
                                                    C.c_uint32(_lib.RV_COMPILE_WHOLE_PROVER if whole_prover else 0), C.byref(self.handle)))

    @property
    def info(self) -> dict:
        ci = _lib.CircuitInfo()
        _lib.check(_lib.lib().rv_circuit_get_info(self.handle, C.byref(ci)))
        d = {n: int(getattr(ci, n)) for n, _ in ci._fields_}
        esb = C.c_uint64()
        _lib.check(_lib.lib().rv_circuit_early_staging_bytes(self.handle, C.byref(esb)))
        d["early_staging_bytes"] = int(esb.value)
        return d

    def record_sizes(self) -> Tuple[int, int]:
        """bytes of one OpenOnline record in the gf2 / z64 section of a proof of this circuit"""
        a, b = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib().rv_circuit_record_sizes(self.handle, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def close(self):
        if self.handle:
            if self.ctx.handle:  # a circuit lives in its context's arena: once that is gone there is nothing left to free
                _lib.lib().rv_circuit_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _as_circuit(circuit, wire_counts, ctx=None, whole_prover=False) -> Circuit:
    if isinstance(circuit, Circuit):
        if wire_counts is not None and tuple(wire_counts) != circuit.wire_counts:
            raise ValueError("wire_counts differ from the compiled circuit's")
        return circuit
    return Circuit(circuit, wire_counts, ctx, whole_prover=whole_prover)


def _witness(wit_gf2, wit_z64):
    g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
    z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
    return g, z


class Proof:
    """bincode(Proof) bytes.  A proof that came out of rv_prove stays in the library's (page-locked) buffer
    until it is dropped — `bytes(proof)` copies it out, `verify` reads it in place."""

    def __init__(self, data: bytes = b"", _owned=None):
        self._bytes = None if _owned is not None else bytes(data)
        self._ptr, self._len = _owned if _owned is not None else (None, len(self._bytes))
        if _owned is not None:
            import weakref

            weakref.finalize(self, _lib.lib().rv_free, self._ptr)

    @property
    def data(self) -> bytes:
        if self._bytes is None:
            self._bytes = C.string_at(self._ptr, self._len)
        return self._bytes

    def __bytes__(self):
        return self.data

    def __len__(self):
        return self._len

    @property
    def comm(self) -> bytes:
        return C.string_at(self._ptr, 32) if self._bytes is None else self._bytes[:32]

    def _buffer(self):
        """(pointer, length) of the proof bytes without copying them"""
        if self._ptr is not None:
            return self._ptr, self._len
        return C.cast(C.c_char_p(self._bytes), C.c_void_p), self._len

    @staticmethod
    def new(circuit, wit_gf2: Sequence[int], wit_z64: Sequence[int], wire_counts: Optional[Tuple[int, int]] = None,
            seeds: Union[None, bytes, np.ndarray] = None, ctx: Optional[Context] = None) -> "Proof":
        """Proof::new.  `seeds` (256x16 bytes) injects the per-repetition seeds the reference
        draws from OsRng; None draws them from the OS."""
        g, z = _witness(wit_gf2, wit_z64)
        s = None
        if seeds is not None:
            s = np.ascontiguousarray(np.frombuffer(bytes(seeds), np.uint8) if isinstance(seeds, (bytes, bytearray))
                                     else np.asarray(seeds, dtype=np.uint8)).reshape(TOTAL_REPS, 16)
        out = C.c_void_p()
        n = C.c_size_t()
        if isinstance(circuit, Circuit):
            c = _as_circuit(circuit, wire_counts, ctx)
            _lib.check(_lib.lib().rv_prove(c.ctx.handle, c.handle, _ptr(g), C.c_size_t(len(g)), _ptr(z), C.c_size_t(len(z)),
                                           _ptr(s), C.byref(out), C.byref(n)))
        else:
            # the reference's own call shape (proof/mod.rs:119-125): the raw op list, compiled for this one proof
            ops = program(circuit) if len(circuit) else np.zeros(0, OP_DTYPE)
            cx = ctx or Context.default()
            _lib.check(_lib.lib().rv_prove_ops(cx.handle, ops.ctypes.data_as(C.c_void_p), C.c_size_t(len(ops)), _ptr(g), C.c_size_t(len(g)),
                                               _ptr(z), C.c_size_t(len(z)), C.c_size_t(int(wire_counts[0])), C.c_size_t(int(wire_counts[1])),
                                               _ptr(s), C.byref(out), C.byref(n)))
        return Proof(_owned=(C.c_void_p(out.value), n.value))

    @staticmethod
    def new_batch(circuit, wits_gf2, wits_z64=None, wire_counts: Optional[Tuple[int, int]] = None, seeds=None,
                  ctx: Optional[Context] = None) -> "list[Proof]":
        """`len(wits_gf2)` proofs of one circuit in one pass (rv_prove_batch): every dependency level is launched
        once for the whole batch.  wits_gf2: [B][n] bits; wits_z64: [B][m] words or None; seeds: [B][256][16] bytes or
        None (OS randomness).  Each proof equals Proof.new(circuit, wits_gf2[b], wits_z64[b], seeds=seeds[b])."""
        c = _as_circuit(circuit, wire_counts, ctx, whole_prover=True)
        g = np.ascontiguousarray(np.asarray(wits_gf2, dtype=np.uint8))
        if g.ndim != 2:
            raise ValueError("wits_gf2 must be [batch][n_bits]")
        batch = g.shape[0]
        z = np.ascontiguousarray(np.asarray(wits_z64 if wits_z64 is not None else np.zeros((batch, 0)), dtype=np.uint64))
        if z.ndim != 2 or z.shape[0] != batch:
            raise ValueError("wits_z64 must be [batch][n_words]")
        s = None
        if seeds is not None:
            s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(batch, TOTAL_REPS, 16)
        outs = (C.c_void_p * batch)()
        lens = (C.c_size_t * batch)()
        _lib.check(_lib.lib().rv_prove_batch(c.ctx.handle, c.handle, C.c_size_t(batch), _ptr(g), C.c_size_t(g.shape[1]), _ptr(z),
                                             C.c_size_t(z.shape[1]), _ptr(s), outs, lens))
        return [Proof(_owned=(C.c_void_p(outs[b]), int(lens[b]))) for b in range(batch)]

    def verify(self, circuit, wire_counts: Optional[Tuple[int, int]] = None, ctx: Optional[Context] = None,
               strict: bool = True) -> bool:
        """Proof::verify (/root/reference/src/proof/mod.rs:224-307), strict by default: the opened repetitions'
        AssertZero gates must hold and the records' `omit` must match the challenge -- both of which the reference
        leaves unchecked, so that it accepts proofs of false statements.  strict=False is
        RV_VERIFY_REFERENCE_COMPAT: exactly the reference's answer (compatibility tests only)."""
        ok = C.c_int()
        buf, n = self._buffer()
        flags = 0 if strict else _lib.RV_VERIFY_REFERENCE_COMPAT
        if isinstance(circuit, Circuit):
            c = _as_circuit(circuit, wire_counts, ctx)
            _lib.check(_lib.lib().rv_verify_ex(c.ctx.handle, c.handle, buf, C.c_size_t(n), C.c_uint32(flags), C.byref(ok)))
        else:
            ops = program(circuit) if len(circuit) else np.zeros(0, OP_DTYPE)
            cx = ctx or Context.default()
            _lib.check(_lib.lib().rv_verify_ops(cx.handle, ops.ctypes.data_as(C.c_void_p), C.c_size_t(len(ops)), C.c_size_t(int(wire_counts[0])),
                                                C.c_size_t(int(wire_counts[1])), buf, C.c_size_t(n), C.c_uint32(flags), C.byref(ok)))
        return bool(ok.value)


def verify_batch(circuit, proofs, wire_counts: Optional[Tuple[int, int]] = None, ctx: Optional[Context] = None,
                 strict: bool = True) -> "list[bool]":
    """rv_verify_batch: Proof.verify for many proofs of one circuit in one pass (`proofs`: Proof objects or bytes);
    -> one bool per proof, each what Proof.verify(circuit, strict=strict) would return -- except that a proof whose
    bytes cannot be parsed is simply False here (on its own it raises): one malformed proof does not keep the others
    from being verified."""
    c = _as_circuit(circuit, wire_counts, ctx)
    n = len(proofs)
    if n == 0:
        return []
    keep = []
    ptrs = (C.c_void_p * n)()
    lens = (C.c_size_t * n)()
    for i, p in enumerate(proofs):
        if isinstance(p, Proof):
            buf, ln = p._buffer()
            ptrs[i] = C.cast(buf, C.c_void_p).value
            keep.append(p)
        else:
            b = bytes(p)
            buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0")
            ptrs[i] = C.addressof(buf)
            ln = len(b)
        keep.append(buf)
        lens[i] = ln
    ok = (C.c_int * n)()
    flags = 0 if strict else _lib.RV_VERIFY_REFERENCE_COMPAT
    _lib.check(_lib.lib().rv_verify_batch(c.ctx.handle, c.handle, C.c_size_t(n), ptrs, lens, C.c_uint32(flags), ok))
    return [bool(x) for x in ok]


# ---- Fiat-Shamir helpers (host) ----
def combine_digests(h) -> bytes:
    h = np.ascontiguousarray(np.asarray(h, dtype=np.uint8)).reshape(TOTAL_REPS, 32)
    out = np.zeros(32, np.uint8)
    _lib.check(_lib.lib().rv_combine_digests(_ptr(h), _ptr(out)))
    return out.tobytes()


def challenge(comm: bytes) -> np.ndarray:
    c = np.frombuffer(bytes(comm), np.uint8).copy()
    out = np.zeros(TOTAL_REPS, np.uint8)
    _lib.check(_lib.lib().rv_challenge(_ptr(c), _ptr(out)))
    return out
