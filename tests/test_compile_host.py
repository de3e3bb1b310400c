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


# ---- the parallel compiler (csrc/compile_par.cpp) against the sequential one: identical tables, field by field ----
def compile_compare(L, prog, wc, flags=0, threads=4):
    prog = np.ascontiguousarray(prog)
    d = C.c_int(-7)
    rc = L.rv_hook_compile_compare(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(int(wc[0])), C.c_size_t(int(wc[1])),
                                   C.c_uint32(flags), C.c_int(threads), C.byref(d))
    return rc, d.value


def same_tables(L, prog, wc, flags, threads):
    """identical tables -- or, without the prover's hint, a deep narrow circuit left to the sequential compiler (which then
    searches over the ways of splitting long sums, compile.cpp `balance`)"""
    got = compile_compare(L, prog, wc, flags, threads)
    return got == (0, 0) or (flags == 0 and got == (0, -1))


def long_random_program(rng, n_ops, n_wires2=400, n_wires64=60, old2=32, p_z64=0.25):
    """A long op list built with numpy (no B2A): recycled wires (a read usually sees a recent write), `old2` GF(2) wires
    written once at the start and read everywhere (reads that see a write made many thread ranges earlier), wires that are
    never written (they read as the zero wire, interpreter/single.rs:16), SizeHints in the middle that enlarge the wire
    vectors, and every opcode of both rings."""
    from reverie_amd.ops import OP_DTYPE

    DOM_SIZEHINT = 3
    prog = np.zeros(n_ops, OP_DTYPE)
    dom = (rng.random(n_ops) < p_z64).astype(np.uint8)  # 0 = GF2, 1 = Z64
    opc = rng.choice(np.arange(10, dtype=np.uint8), n_ops, p=[0.04, 0.03, 0.22, 0.06, 0.10, 0.04, 0.30, 0.05, 0.06, 0.10])
    prog["domain"], prog["opcode"] = dom, opc
    nw = np.where(dom == 0, n_wires2, n_wires64)
    prog["dst"] = (rng.integers(0, 1 << 30, n_ops) % (nw - (dom == 0) * old2)) + (dom == 0) * old2   # never overwrite the old wires
    prog["a"] = rng.integers(0, 1 << 30, n_ops) % (nw + 3)
    prog["b"] = rng.integers(0, 1 << 30, n_ops) % (nw + 3)
    prog["a"] = np.minimum(prog["a"], nw - 1)
    prog["b"] = np.minimum(prog["b"], nw - 1)
    far = (rng.random(n_ops) < 0.1) & (dom == 0)
    prog["a"][far] = rng.integers(0, old2, int(far.sum()))
    prog["imm"] = rng.integers(0, 1 << 62, n_ops, dtype=np.uint64)
    # the old wires: inputs at the very start
    prog["domain"][:old2], prog["opcode"][:old2], prog["dst"][:old2] = 0, OP_INPUT, np.arange(old2)
    # two SizeHints: the last quarter of the program may use 50 more wires of each ring
    q = 3 * n_ops // 4
    late = np.arange(q + 1, n_ops)
    grow = late[rng.random(len(late)) < 0.05]
    prog["dst"][grow] = np.where(prog["domain"][grow] == 0, n_wires2, n_wires64) + rng.integers(0, 50, len(grow))
    for at, (z, g) in ((n_ops // 3, (n_wires64 // 2, n_wires2 // 2)), (q, (n_wires64 + 50, n_wires2 + 50))):
        prog[at] = (DOM_SIZEHINT, 0, 0, 0, z, g, 0)
    return prog, (n_wires64, n_wires2)


@pytest.mark.parametrize("seed", range(6))
def test_parallel_compiler_random_programs(L, seed):
    rng = np.random.default_rng(900 + seed)
    prog, _, _, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(200, 3000)))
    with_b2a = compile_compare(L, prog, wc)
    assert with_b2a == (0, -1)                 # B2A gates: left to the sequential compiler
    prog = prog[prog["domain"] != DOM_B2A]
    for flags in (0, 1):
        for threads in (2, 3, 8):
            assert same_tables(L, prog, wc, flags, threads)


@pytest.mark.parametrize("seed,n_ops,threads", [(1, 70_000, 8), (2, 120_000, 16), (3, 33_000, 5), (4, 260_000, 8)])
def test_parallel_compiler_long_programs_with_far_reads(L, seed, n_ops, threads):
    rng = np.random.default_rng(seed)
    prog, wc = long_random_program(rng, n_ops)
    for flags in (0, 1):
        assert same_tables(L, prog, wc, flags, threads)
    # the same ops as ONE dependency chain per ring (every op reads the previous result): the data-flow pass degenerates to
    # the sequential order, blocks handed from thread to thread
    chain = prog.copy()
    body = np.arange(40, len(chain))
    body = body[chain["domain"][body] <= 1]
    chain["a"][body] = np.where(chain["domain"][body] == 0, 399, 59)
    chain["dst"][body] = np.where(chain["opcode"][body] != OP_ASSERTZERO, chain["a"][body], chain["dst"][body])
    assert compile_compare(L, chain, wc, 1, threads) == (0, 0)


def test_parallel_compiler_layered_and_recycled(L):
    for kw in (dict(n_in=256, width=2048, layers=12), dict(n_in=100, width=128, layers=300, p_and=0.3), dict(n_in=128, width=1024, layers=20, recycle=True)):
        prog, wit, wc, st = circuits.layered_gf2(**kw)
        for flags in (0, 1):
            assert same_tables(L, prog, wc, flags, 7)
    assert compile_compare(L, *circuits.layered_gf2(n_in=256, width=2048, layers=12)[0:3:2], 0, 7) == (0, 0)  # (wide: never left)
    for recycle in (False, True):
        prog, wit, wc, st = circuits.layered_z64(n_in=64, width=512, n_mul=20000, recycle=recycle)
        assert compile_compare(L, prog, wc, 0, 6) == (0, 0)


def test_parallel_compiler_leaves_errors_to_the_sequential_one(L):
    rng = np.random.default_rng(5)
    prog, wc = long_random_program(rng, 50_000)
    bad = prog.copy()
    bad["a"][40_000] = 10_000                  # wire out of range late in the program (a GF(2) / Z64 op with operands)
    bad["opcode"][40_000] = OP_MUL
    assert compile_compare(L, bad, wc) == (3, -1)
    bad = prog.copy()
    bad["opcode"][123] = 77
    assert compile_compare(L, bad, wc) == (5, -1)
    # a wire the SizeHint only allows LATER is out of range before it
    bad = prog.copy()
    q = 3 * len(prog) // 4
    first_late = int(np.nonzero((bad["dst"] >= wc[1]) & (bad["domain"] == 0))[0][0])
    bad[[q, first_late]] = bad[[first_late, q]]
    assert compile_compare(L, bad, wc)[0] == 3


def test_compile_dispatch_takes_the_parallel_compiler_for_large_programs(L, monkeypatch):
    prog, wit, wc, st = circuits.layered_gf2(n_in=512, width=16384, layers=14)
    assert len(prog) > 200_000
    monkeypatch.setenv("RV_COMPILE_THREADS", "4")
    rc, par = compile_info(L, prog, wc, flags=1)
    monkeypatch.setenv("RV_COMPILE_SEQ", "1")
    rc2, seq = compile_info(L, prog, wc, flags=1)
    assert rc == 0 and rc2 == 0 and par == seq


# ---- the early-corrections plan (rv_prove, csrc/api.hip: early_plan), host side only ----
def early_plan(L, prog, wc, flags=1):
    prog = np.ascontiguousarray(prog)
    out = (C.c_uint64 * 22)()
    rc = L.rv_hook_early_plan(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(int(wc[0])), C.c_size_t(int(wc[1])), C.c_uint32(flags), out)
    assert rc == 0
    o = [int(x) for x in out]
    return {"ok": o[0], "z64": o[1], "reps": o[2], "chunks": o[3], "bytes": o[4], "check": o[5], "ready": o[6:6 + min(o[3], 16)]}


def chains_program(n_chains, depth, chain_major):
    """n_chains independent chains of `depth` dependent Mul gates; program order chain by chain (every level then writes
    preprocessing rows spread over the whole transcript) or round by round (the rows complete in step with the levels)"""
    ops = [GF2.Input(i) for i in range(n_chains + 1)]
    wire = n_chains + 1
    cur = list(range(n_chains))
    order = [(c, r) for c in range(n_chains) for r in range(depth)] if chain_major else [(c, r) for r in range(depth) for c in range(n_chains)]
    for c, _ in order:
        ops.append(GF2.Mul(wire, cur[c], n_chains))
        cur[c] = wire
        wire += 1
    return program(ops), (0, wire)


def test_early_plan_host(L, monkeypatch):
    """The plan rv_prove's early-corrections path follows, on the host alone: taken for layered GF(2) and Z64 circuits with every
    chunk's ready level checked against the compiled gates (no later level writes its rows, its own level does), refused when the
    rows do not complete in step with the levels, for programs with both domains, below the size threshold, and when all
    repetitions' vectors would not fit through PCIe beside the kernels."""
    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    for chunks in ("1", "4", "9"):
        monkeypatch.setenv("RV_EARLY_CHUNKS", chunks)
        prog, _, wc, st = circuits.layered_gf2(n_in=64, width=4096, layers=24, p_and=0.5, fold_to=16)
        for flags in (0, 1):
            p = early_plan(L, prog, wc, flags)
            assert p["ok"] and not p["z64"] and p["check"] and p["reps"] == 256, p
            assert p["chunks"] == int(chunks) and p["ready"] == sorted(p["ready"]), p
            assert p["bytes"] >= 256 * (st["and"] // 8 + 1), p
    monkeypatch.setenv("RV_EARLY_CHUNKS", "4")
    # rows in step with the levels / spread over the transcript
    prog, wc = chains_program(64, 48, chain_major=False)
    p = early_plan(L, prog, wc)
    assert p["ok"] and p["check"], p
    prog, wc = chains_program(64, 48, chain_major=True)
    assert not early_plan(L, prog, wc)["ok"]
    # Z64: the staged repetitions follow the PCIe estimate, or RV_EARLY_REPS under RV_EARLY=2
    prog, _, wc, st = circuits.layered_z64(n_in=64, width=1024, n_mul=6000)
    p = early_plan(L, prog, wc)
    assert p["ok"] and p["z64"] and p["check"] and 64 <= p["reps"] <= 256 and p["reps"] % 8 == 0, p
    assert p["bytes"] >= p["reps"] * 8 * st["mul"], p
    monkeypatch.setenv("RV_EARLY", "2")
    monkeypatch.setenv("RV_EARLY_REPS", "200")
    p = early_plan(L, prog, wc)
    assert p["ok"] and p["reps"] == 200 and p["check"], p
    monkeypatch.delenv("RV_EARLY")
    # both domains in one program: no plan
    prog, _, _, _ = circuits.random_mixed(np.random.default_rng(5), n_gates=3000)
    assert not early_plan(L, prog, (12, 90))["ok"]
    # below the threshold
    monkeypatch.delenv("RV_EARLY_MIN")
    prog, _, wc, _ = circuits.layered_gf2(n_in=64, width=4096, layers=24, p_and=0.5, fold_to=16)
    assert not early_plan(L, prog, wc)["ok"]


def test_early_plan_pcie_window(L, monkeypatch):
    """A few very wide all-AND levels: 32 bytes per Mul for all repetitions do not fit through PCIe in the time the levels and the
    hashes are estimated to take -- the plan stages fewer repetitions (RV_EARLY=2 takes all, or RV_EARLY_REPS)."""
    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    prog, _, wc, st = circuits.layered_gf2(n_in=64, width=1 << 20, layers=4, p_and=1.0, fold_to=16)
    assert st["and"] == 4 << 20
    # (round 6: with the mask generator beside the levels the phase lasts as long as the cipher needs -- 0.6 ns per Mul, and 32 bytes per Mul
    # cross the link in 0.58: everything fits; with the masks before the first level the window is the levels' own time again)
    p = early_plan(L, prog, wc)
    assert p["ok"] and p["check"] and p["reps"] == 256, p
    monkeypatch.setenv("RV_OVERLAP", "0")
    p = early_plan(L, prog, wc)
    assert p["ok"] and p["check"] and 64 <= p["reps"] < 256 and p["reps"] % 8 == 0, p
    monkeypatch.setenv("RV_EARLY", "2")
    p = early_plan(L, prog, wc)
    assert p["ok"] and p["check"] and p["reps"] == 256, p
    monkeypatch.setenv("RV_EARLY_REPS", "96")
    assert early_plan(L, prog, wc)["reps"] == 96


def test_parallel_compiler_survives_a_failing_worker(monkeypatch):
    """a bad_alloc on a worker thread of the parallel compiler must come back as RV_E_NOMEM from the calling thread
    (after every worker has left the pass), not as std::terminate; the next compile works (compile_par.cpp: Pool)"""
    import ctypes as C

    from reverie_amd import _lib

    L = _lib.lib()
    prog, wit, wc, st = circuits.layered_gf2(n_in=256, width=4096, layers=60)
    assert len(prog) >= 200_000
    ci = _lib.CircuitInfo()

    def compile_once():
        return L.rv_hook_compile_info(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(int(wc[0])), C.c_size_t(int(wc[1])),
                                      C.c_uint32(1), C.c_size_t(0), C.byref(ci))

    monkeypatch.setenv("RV_COMPILE_THREADS", "4")
    for k in (1, 3, 6):
        monkeypatch.setenv("RV_TEST_POOL_THROW", str(k))
        assert compile_once() == 6  # RV_E_NOMEM
    monkeypatch.delenv("RV_TEST_POOL_THROW")
    assert compile_once() == 0 and ci.gf2_muls == st["and"]


def test_ops_cache_comparison_is_by_content(L):
    """The compiled-circuit cache of rv_prove_ops / rv_verify_ops finds a circuit by comparing the op arrays themselves (round 6; round 5's
    unkeyed 128-bit hash had constructible collisions: VERDICT r5 weak #2).  rv_hook_ops_same is that comparison: equal arrays, a copy, one
    differing byte anywhere in a 40 MB array (first / middle / last 4 MiB piece), and the verdict's colliding pair (op 2's imm = the hash's
    public constant, op 3 differs) must all come out right."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 40 * 1024 * 1024 + 12345, dtype=np.uint8)
    b = a.copy()
    same = lambda x, y: L.rv_hook_ops_same(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_size_t(x.nbytes))  # noqa: E731
    assert same(a, a) == 1 and same(a, b) == 1
    for pos in (0, 1, (4 << 20) - 1, 4 << 20, 17 << 20, len(a) // 2, len(a) - 2, len(a) - 1):
        b[pos] ^= 0x40
        assert same(a, b) == 0, pos
        b[pos] ^= 0x40
    assert same(a, b) == 1

    def circuit(assertion):
        ops = [GF2.Input(0), GF2.Input(1), Z64.AddConst(0, 0, 0xE7037ED1A0B428DB), GF2.AssertZero(0) if assertion else GF2.AddConst(2, 0, 1)]
        ops += [GF2.Mul(3 + i, 0, 1) for i in range(2044)]
        return program(ops)

    pa, pb = circuit(True), circuit(False)
    assert pa.nbytes == pb.nbytes == 2048 * 24 and same(pa, pb) == 0 and same(pa, pa.copy()) == 1
