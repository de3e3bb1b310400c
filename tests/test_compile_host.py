"""The gate-stream compiler on the host alone (rv_hook_compile_info: no device): the counters that are pure functions of
the op list against an independent count in Python -- ShareGen::next() calls per repetition (generator/share.rs:54-65:
Input / Random 1, Mul 2; B2A 64 + 2 x 63 GF(2) and one Z64, combine.rs:65-76,143-146), transcript-bearing gates, inputs --
the effect of the whole-prover compile hint, the errors the reference raises while stepping, and the bookkeeping of the
streaming prover's independently compiled pieces (count_masks / relocate_chunk) for many cut sizes."""
import ctypes as C
import os

import numpy as np
import pytest

import circuits
from reverie_amd.ops import (DOM_B2A, DOM_GF2, DOM_Z64, GF2, OP_ASSERTZERO, OP_INPUT, OP_MUL, Z64, program)

OP_RANDOM = 1


@pytest.fixture(scope="module")
def L():
    from reverie_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.lib()


def compile_info(L, prog, wc, flags=0, chunk_ops=0):
    from reverie_amd import _lib

    prog = np.ascontiguousarray(prog)
    ci = _lib.CircuitInfo()
    rc = L.rv_hook_compile_info(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(int(wc[0])), C.c_size_t(int(wc[1])),
                                C.c_uint32(flags), C.c_size_t(chunk_ops), C.byref(ci))
    return rc, {n: int(getattr(ci, n)) for n, _ in ci._fields_}


def python_counts(prog):
    dom, opc = prog["domain"], prog["opcode"]
    g, z, b = dom == DOM_GF2, dom == DOM_Z64, dom == DOM_B2A
    n_b2a = int(b.sum())
    cnt = lambda m, o: int((m & (opc == o)).sum())  # noqa: E731
    return {
        "gf2_inputs": cnt(g, OP_INPUT), "gf2_muls": cnt(g, OP_MUL) + 63 * n_b2a, "z64_inputs": cnt(z, OP_INPUT), "z64_muls": cnt(z, OP_MUL),
        "gf2_masks": cnt(g, OP_INPUT) + cnt(g, OP_RANDOM) + 2 * cnt(g, OP_MUL) + 190 * n_b2a,
        "z64_masks": cnt(z, OP_INPUT) + cnt(z, OP_RANDOM) + 2 * cnt(z, OP_MUL) + n_b2a,
        "gf2_asserts": cnt(g, OP_ASSERTZERO) + 64 * n_b2a,  # (B2A's 64 recorded reconstructions count as reveals)
        "z64_asserts": cnt(z, OP_ASSERTZERO), "b2a": n_b2a, "n_ops": len(prog),
    }


@pytest.mark.parametrize("seed", range(6))
def test_counters_match_an_independent_count(L, seed):
    rng = np.random.default_rng(100 + seed)
    if seed % 2:
        prog, _, _, _ = circuits.random_mixed(rng, n_gates=int(rng.integers(100, 1500)))
        wc = (12, 90)
    else:
        prog, _, wc = circuits.random_gf2(rng, n_in=int(rng.integers(1, 100)), n_gates=int(rng.integers(50, 4000)), n_wires=int(rng.integers(4, 300)))
    rc, info = compile_info(L, prog, wc)
    assert rc == 0
    want = python_counts(prog)
    for k, v in want.items():
        assert info[k] == v, k
    assert info["device_bytes"] == 0 and info["levels"] >= 1


def test_layered_workload_shape(L):
    prog, wit, wc, st = circuits.layered_gf2(n_in=256, width=2048, layers=12)
    rc, info = compile_info(L, prog, wc)
    assert rc == 0
    assert info["gf2_muls"] == st["and"] and info["gf2_inputs"] == 256 and info["gf2_masks"] == 256 + 2 * st["and"]
    # at least one level per layer that holds a Mul; a materialised Xor between two Muls adds one, the asserted tail a few
    assert 12 <= info["levels"] <= 2 * 12 + 4
    # every Mul reads two operands of at least one row each; materialised Xor gates are at most the circuit's
    n_linear_ops = int(((prog["opcode"] != OP_MUL) & (prog["opcode"] != OP_INPUT) & (prog["opcode"] != OP_ASSERTZERO)).sum())
    assert info["gf2_operand_rows"] >= 2 * st["and"] and info["gf2_rows_written"] <= n_linear_ops


def test_whole_prover_hint_changes_the_gate_stream_not_the_protocol_counters(L):
    from reverie_amd import _lib

    prog, wit, wc, st = circuits.layered_gf2(n_in=300, width=4096, layers=9)
    rc0, plain = compile_info(L, prog, wc)
    rc1, hinted = compile_info(L, prog, wc, flags=_lib.RV_COMPILE_WHOLE_PROVER)
    assert rc0 == 0 and rc1 == 0
    for k in ("gf2_inputs", "gf2_muls", "gf2_asserts", "gf2_masks", "n_ops"):
        assert plain[k] == hinted[k]
    if not os.environ.get("RV_LAZY_K"):  # (the knob fixes the choice for both)
        assert hinted["gf2_rows_written"] < plain["gf2_rows_written"]        # fewer materialised Xor gates ...
        assert hinted["gf2_operand_rows"] / hinted["gf2_muls"] > 2.0        # ... read as lazy sums by the Mul gates
    assert plain["gf2_operand_rows"] <= 2 * plain["gf2_muls"] + 2 * plain["gf2_rows_written"] + plain["gf2_asserts"]
    assert L.rv_hook_compile_info(None, 0, 0, 0, 2, 0, C.byref(_lib.CircuitInfo())) == 9  # unknown flag: RV_E_ARG


def test_errors_the_reference_raises_while_stepping(L):
    ok = program([GF2.Input(0), GF2.Input(1), GF2.Mul(2, 0, 1)])
    assert compile_info(L, ok, (0, 3))[0] == 0
    assert compile_info(L, ok, (0, 2))[0] == 3                                     # wire index >= wire count
    assert compile_info(L, program([Z64.Input(0), Z64.Mul(1, 0, 5)]), (2, 0))[0] == 3
    bad = ok.copy()
    bad["opcode"][2] = 99
    assert compile_info(L, bad, (0, 3))[0] == 5                                    # unknown opcode
    bad = ok.copy()
    bad["reserved"][1] = 1
    assert compile_info(L, bad, (0, 3))[0] == 5


@pytest.mark.parametrize("seed", range(4))
def test_streaming_pieces_add_up(L, seed):
    """what stream_feed_impl relies on: a piece compiled on its own, with the ShareGen phases count_masks predicts from the
    ops before it, consumes exactly its share of masks and transcript events -- for cuts that fall everywhere, B2A included"""
    rng = np.random.default_rng(500 + seed)
    if seed % 2:
        prog, _, _, _ = circuits.random_mixed(rng, n_gates=int(rng.integers(300, 2500)))
        wc = (12, 90)
    else:
        prog, _, wc = circuits.random_gf2(rng, n_in=int(rng.integers(1, 100)), n_gates=int(rng.integers(500, 5000)), n_wires=int(rng.integers(4, 300)))
    for chunk in (1, 7, 64, 129, 1000, len(prog) - 1, len(prog) + 5):
        assert compile_info(L, prog, wc, chunk_ops=max(chunk, 1))[0] == 0, chunk
