"""Gate-stream types: the host-side mirror of the reference's circuit IR.

The reference consumes `Vec<mcircuit::CombineOperation>` (re-exported at
/root/reference/src/lib.rs:5-7; variants used at src/interpreter/single.rs:106-156 and
src/interpreter/combine.rs:120-132).  Across the C-ABI the same information travels as a
flat array of 24-byte `rv_op` records (include/reverie_amd.h).  This module builds those
arrays with numpy; it has no GPU dependency.
"""
from __future__ import annotations

import numpy as np

# layout == struct rv_op in include/reverie_amd.h
OP_DTYPE = np.dtype(
    [
        ("domain", "u1"),
        ("opcode", "u1"),
        ("reserved", "<u2"),
        ("dst", "<u4"),
        ("a", "<u4"),
        ("b", "<u4"),
        ("imm", "<u8"),
    ]
)
assert OP_DTYPE.itemsize == 24

DOM_GF2, DOM_Z64, DOM_B2A, DOM_SIZEHINT = 0, 1, 2, 3

(
    OP_INPUT,
    OP_RANDOM,
    OP_ADD,
    OP_ADDCONST,
    OP_SUB,
    OP_SUBCONST,
    OP_MUL,
    OP_MULCONST,
    OP_ASSERTZERO,
    OP_CONST,
) = range(10)

# protocol constants (src/lib.rs:17-38)
PLAYERS = 8
PACKED = 8
BATCH_SIZE = 128
ONLINE_REPS = 40
TOTAL_REPS = 256
PREPROCESSING_REPS = TOTAL_REPS - ONLINE_REPS
PACKED_REPS = TOTAL_REPS // PACKED


class Operation:
    """`mcircuit::Operation<T>` constructors for one domain (0 = GF2, 1 = Z64)."""

    def __init__(self, domain: int):
        self.domain = domain

    def _mk(self, opcode, dst=0, a=0, b=0, imm=0):
        return (self.domain, opcode, 0, int(dst), int(a), int(b), int(imm) & 0xFFFFFFFFFFFFFFFF)

    def Input(self, dst):
        return self._mk(OP_INPUT, dst)

    def Random(self, dst):
        return self._mk(OP_RANDOM, dst)

    def Add(self, dst, a, b):
        return self._mk(OP_ADD, dst, a, b)

    def AddConst(self, dst, a, c):
        return self._mk(OP_ADDCONST, dst, a, 0, c)

    def Sub(self, dst, a, b):
        return self._mk(OP_SUB, dst, a, b)

    def SubConst(self, dst, a, c):
        return self._mk(OP_SUBCONST, dst, a, 0, c)

    def Mul(self, dst, a, b):
        return self._mk(OP_MUL, dst, a, b)

    def MulConst(self, dst, a, c):
        return self._mk(OP_MULCONST, dst, a, 0, c)

    def AssertZero(self, a):
        return self._mk(OP_ASSERTZERO, 0, a)

    def Const(self, dst, c):
        return self._mk(OP_CONST, dst, 0, 0, c)


GF2 = Operation(DOM_GF2)
Z64 = Operation(DOM_Z64)


def B2A(dst_z64, src_gf2_low):
    """`CombineOperation::B2A(dst, src)`: z64[dst] <- the 64 gf2 wires src..src+63, LSB first."""
    return (DOM_B2A, 0, 0, int(dst_z64), int(src_gf2_low), 0, 0)


def SizeHint(z64_wires, gf2_wires):
    """`CombineOperation::SizeHint(z64, gf2)` (tuple order as combine.rs:122)."""
    return (DOM_SIZEHINT, 0, 0, 0, int(z64_wires), int(gf2_wires), 0)


def program(ops) -> np.ndarray:
    """list of op tuples (or an OP_DTYPE array) -> contiguous rv_op array."""
    if isinstance(ops, np.ndarray):
        assert ops.dtype == OP_DTYPE
        return np.ascontiguousarray(ops)
    return np.array(list(ops), dtype=OP_DTYPE)


def largest_wires(prog: np.ndarray):
    """Equivalent of `mcircuit::largest_wires` (call site src/main.rs:73): the wire-vector
    sizes (z64, gf2) needed to run `prog`, i.e. 1 + the largest index touched per domain."""
    prog = program(prog)
    z64 = gf2 = 0
    for dom, store in ((DOM_GF2, "gf2"), (DOM_Z64, "z64")):
        sel = prog[prog["domain"] == dom]
        if len(sel) == 0:
            continue
        opc = sel["opcode"]
        hi = 0
        has_dst = opc != OP_ASSERTZERO
        if has_dst.any():
            hi = max(hi, int(sel["dst"][has_dst].max()) + 1)
        uses_a = ~np.isin(opc, (OP_INPUT, OP_RANDOM, OP_CONST))
        if uses_a.any():
            hi = max(hi, int(sel["a"][uses_a].max()) + 1)
        uses_b = np.isin(opc, (OP_ADD, OP_SUB, OP_MUL))
        if uses_b.any():
            hi = max(hi, int(sel["b"][uses_b].max()) + 1)
        if store == "gf2":
            gf2 = max(gf2, hi)
        else:
            z64 = max(z64, hi)
    b2a = prog[prog["domain"] == DOM_B2A]
    if len(b2a):
        z64 = max(z64, int(b2a["dst"].max()) + 1)
        gf2 = max(gf2, int(b2a["a"].max()) + 64)
    sh = prog[prog["domain"] == DOM_SIZEHINT]
    if len(sh):
        z64 = max(z64, int(sh["a"].max()))
        gf2 = max(gf2, int(sh["b"].max()))
    return z64, gf2
