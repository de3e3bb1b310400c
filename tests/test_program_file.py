"""bincode(Vec<CombineOperation>) program files (layout per Appendix A.7 — recalled, unpinned).
Host-only: no GPU needed.  A hand-assembled byte string pins the layout this reader implements."""
import struct

import numpy as np
import pytest

import circuits
from reverie_amd import program_file
from reverie_amd._lib import ReverieError
from reverie_amd.ops import B2A, GF2, OP_DTYPE, Z64, SizeHint, program


def u32(v):
    return struct.pack("<I", v)


def u64(v):
    return struct.pack("<Q", v)


def test_hand_assembled_layout():
    # [SizeHint(3, 70), GF2 Input(5), GF2 AddConst(6, 5, true), GF2 Mul(7, 5, 6), GF2 AssertZero(7),
    #  Z64 Const(1, 2^63+9), Z64 MulConst(2, 1, 77), B2A(0, 64)]
    raw = u64(8)
    raw += u32(3) + u64(3) + u64(70)
    raw += u32(0) + u32(0) + u64(5)
    raw += u32(0) + u32(3) + u64(6) + u64(5) + b"\x01"
    raw += u32(0) + u32(6) + u64(7) + u64(5) + u64(6)
    raw += u32(0) + u32(8) + u64(7)
    raw += u32(1) + u32(9) + u64(1) + u64((1 << 63) + 9)
    raw += u32(1) + u32(7) + u64(2) + u64(1) + u64(77)
    raw += u32(2) + u64(0) + u64(64)
    want = program([SizeHint(3, 70), GF2.Input(5), GF2.AddConst(6, 5, 1), GF2.Mul(7, 5, 6), GF2.AssertZero(7),
                    Z64.Const(1, (1 << 63) + 9), Z64.MulConst(2, 1, 77), B2A(0, 64)])
    got = program_file.loads(raw)
    assert got.dtype == OP_DTYPE and got.tobytes() == want.tobytes()
    assert program_file.dumps(want) == raw
    assert program_file.loads(raw + b"trailing").tobytes() == want.tobytes()  # deserialize_from semantics


def test_round_trip_random_programs():
    rng = np.random.default_rng(11)
    for n in (0, 1, 50, 2000):
        prog, _, _, _ = circuits.random_mixed(rng, n_gates=n) if n else (program([]), None, None, None)
        data = program_file.dumps(prog)
        back = program_file.loads(data)
        # GF(2) immediates are one bit in the file
        norm = prog.copy()
        gf2 = norm["domain"] == 0
        norm["imm"][gf2] &= 1
        uses_imm = np.isin(norm["opcode"], (3, 5, 7, 9)) & (norm["domain"] <= 1)
        norm["imm"][~uses_imm] = 0
        assert back.tobytes() == _canonical(norm).tobytes()


def _canonical(p):
    """fields a variant does not carry come back as zero"""
    q = p.copy()
    dom, opc = q["domain"], q["opcode"]
    plain = dom <= 1
    q["dst"][plain & (opc == 8)] = 0
    q["a"][plain & np.isin(opc, (0, 1, 9))] = 0
    q["b"][plain & ~np.isin(opc, (2, 4, 6))] = 0
    q["b"][dom == 2] = 0
    q["dst"][dom == 3] = 0
    q["opcode"][dom >= 2] = 0
    return q


def test_malformed():
    good = program_file.dumps(program([GF2.Input(0), GF2.Mul(1, 0, 0)]))
    with pytest.raises(ReverieError):
        program_file.loads(good[:-1])  # truncated
    with pytest.raises(ReverieError):
        program_file.loads(u64(1) + u32(9) + u64(0) + u64(0))  # unknown CombineOperation variant
    with pytest.raises(ReverieError):
        program_file.loads(u64(1) + u32(0) + u32(12) + u64(0))  # unknown Operation variant
    with pytest.raises(ReverieError):
        program_file.loads(u64(1) + u32(0) + u32(9) + u64(0) + b"\x02")  # bool that is neither 0 nor 1
    with pytest.raises(ReverieError):
        program_file.loads(u64(1 << 40) + good[8:])  # absurd length prefix
    with pytest.raises(ReverieError):
        program_file.loads(u64(1) + u32(0) + u32(0) + u64(1 << 32))  # wire index beyond rv_op's u32
