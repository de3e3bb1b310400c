"""The Z64 prover with the mask generator inside the interpreter's level launches (reverie_amd/csrc/aes.hip: k_z64_fused,
internal.h: Z64FParams) against the CPU oracle and against the two-kernel path (RV_Z64_FUSED=0), byte for byte.
Reference: Instance::op_mul / step at Z64 (src/interpreter/single.rs:25-157), ShareGen::next (src/generator/share.rs:54-65),
DomainZ64::batches_to_shares (src/algebra/z64/domain.rs:64-83), ProverTranscript (src/transcript/prover.rs:181-232)."""
import numpy as np
import pytest

import circuits
from reverie_amd.ops import GF2, Z64, program

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def rv():
    import reverie_amd

    return reverie_amd


def random_z64(rng, n_in, n_gates, n_wires, p_assert=0.04):
    """random Z64 program over the ops the fused path takes (wire reuse, constants, every linear op); asserts on known values"""
    ops, wit, vals = [], [], {}
    n_in = max(n_in, 1)
    for i in range(n_in):
        w = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
        d = int(rng.integers(0, n_wires))
        ops.append(Z64.Input(d))
        wit.append(w)
        vals[d] = w
    live = lambda: list(vals.keys())
    for _ in range(n_gates):
        ks = live()
        a, b = (int(ks[int(rng.integers(0, len(ks)))]) for _ in range(2))
        d = int(rng.integers(0, n_wires))
        c = int(rng.integers(0, 1 << 63))
        kind = rng.choice(["mul", "add", "sub", "addc", "subc", "mulc", "const", "assert"], p=[0.35, 0.2, 0.1, 0.08, 0.08, 0.08, 0.11 - p_assert, p_assert])
        if kind == "mul":
            ops.append(Z64.Mul(d, a, b)); vals[d] = (vals[a] * vals[b]) & M64
        elif kind == "add":
            ops.append(Z64.Add(d, a, b)); vals[d] = (vals[a] + vals[b]) & M64
        elif kind == "sub":
            ops.append(Z64.Sub(d, a, b)); vals[d] = (vals[a] - vals[b]) & M64
        elif kind == "addc":
            ops.append(Z64.AddConst(d, a, c)); vals[d] = (vals[a] + c) & M64
        elif kind == "subc":
            ops.append(Z64.SubConst(d, a, c)); vals[d] = (vals[a] - c) & M64
        elif kind == "mulc":
            ops.append(Z64.MulConst(d, a, c)); vals[d] = (vals[a] * c) & M64
        elif kind == "const":
            ops.append(Z64.Const(d, c)); vals[d] = c
        else:
            ops.append(Z64.SubConst(d, a, vals[a])); vals[d] = 0
            ops.append(Z64.AssertZero(d))
    return program(ops), wit, (n_wires, 0)


def _prove(rv, prog, w64, wc, seeds, monkeypatch, fused, w2=()):
    monkeypatch.setenv("RV_Z64_FUSED", str(fused))
    c = rv.Circuit(prog, wc)
    try:
        p = rv.Proof.new(c, list(w2), w64, seeds=seeds)
        assert p.verify(c, strict=True)
        return bytes(p)
    finally:
        c.close()


@pytest.mark.parametrize("seed", range(8))
def test_fused_random_programs_vs_oracle(rv, oracle, monkeypatch, seed):
    """even and odd Input counts (an odd count makes a Mul's masks straddle cipher blocks: the library must fall back), wire reuse"""
    rng = np.random.default_rng(6400 + seed)
    prog, w64, wc = random_z64(rng, n_in=int(rng.integers(1, 12)) * 2 + (seed & 1), n_gates=int(rng.integers(20, 1500)), n_wires=int(rng.integers(4, 120)))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, [], w64, wc, seeds, threads=8)
    assert _prove(rv, prog, w64, wc, seeds, monkeypatch, 1) == want
    assert _prove(rv, prog, w64, wc, seeds, monkeypatch, 0) == want


@pytest.mark.parametrize("n_in,width,n_mul", [(2, 1, 40), (64, 31, 500), (64, 33, 500), (1024, 129, 3000), (64, 2048, 20000), (2, 5000, 9000)])
def test_fused_layered_shapes_vs_oracle(rv, oracle, rule_seeds, monkeypatch, n_in, width, n_mul):
    """level widths around the kernel's 4-gate wavefront steps and 32-gate workgroup steps, levels without Mul gates"""
    prog, w64, wc, st = circuits.layered_z64(n_in=n_in, width=width, n_mul=n_mul, fold_to=min(16, width))
    want = oracle.prove(prog, [], w64, wc, rule_seeds, threads=8)
    assert _prove(rv, prog, w64, wc, rule_seeds, monkeypatch, 1) == want


def test_fused_interleaved_inputs(rv, oracle, rule_seeds, monkeypatch):
    """Input gates (pairs, so every Mul stays block-aligned) between the Mul gates: several cipher-block runs generated the plain way"""
    rng = np.random.default_rng(77)
    ops, wit, vals, nxt = [], [], [], 0
    for blk in range(12):
        for _ in range(2 * int(rng.integers(1, 4))):
            w = int(rng.integers(0, 1 << 62))
            ops.append(Z64.Input(nxt)); wit.append(w); vals.append(w); nxt += 1
        for _ in range(int(rng.integers(1, 30))):
            a, b = int(rng.integers(0, nxt)), int(rng.integers(0, nxt))
            ops.append(Z64.Mul(nxt, a, b)); vals.append((vals[a] * vals[b]) & M64); nxt += 1
    ops.append(Z64.SubConst(nxt, nxt - 1, vals[-1])); ops.append(Z64.AssertZero(nxt)); nxt += 1
    prog = program(ops)
    want = oracle.prove(prog, [], wit, (nxt, 0), rule_seeds)
    assert _prove(rv, prog, wit, (nxt, 0), rule_seeds, monkeypatch, 1) == want


def test_fused_with_gf2_gates_beside(rv, oracle, rule_seeds, monkeypatch):
    """a circuit with both domains and no B2A: the GF(2) half runs the level interpreter, the Z64 half the fused launches"""
    p2, w2, wc2, _ = circuits.layered_gf2(n_in=64, width=256, layers=6, fold_to=16)
    p64, w64, wc64, _ = circuits.layered_z64(n_in=64, width=128, n_mul=700)
    from reverie_amd.ops import largest_wires

    prog = np.concatenate([p2, p64])
    wc = largest_wires(prog)
    want = oracle.prove(prog, w2, w64, wc, rule_seeds)
    assert _prove(rv, prog, w64, wc, rule_seeds, monkeypatch, 1, w2) == want


def test_fused_invalid_witness_and_reuse(rv, oracle, rule_seeds, monkeypatch):
    """an AssertZero on a non-zero value is reported from the cleartext values; the same circuit then proves a valid witness"""
    monkeypatch.setenv("RV_Z64_FUSED", "1")
    prog, w64, wc, st = circuits.layered_z64(n_in=64, width=256, n_mul=1500)
    c = rv.Circuit(prog, wc)
    bad = list(w64)
    bad[0] ^= 1
    hit = False
    for i in range(len(bad)):
        bad = list(w64)
        bad[i] ^= 1
        try:
            rv.Proof.new(c, [], bad, seeds=rule_seeds)
        except rv.ReverieError as e:
            assert e.code == 1
            hit = True
            break
    assert hit
    good = rv.Proof.new(c, [], w64, seeds=rule_seeds)
    assert bytes(good) == oracle.prove(prog, [], w64, wc, rule_seeds)


@pytest.mark.parametrize("reps", [32, 64, 128])
def test_fused_shards_vs_oracle(rv, oracle, monkeypatch, reps):
    """repetition shards: 64 and 128 repetitions run the fused launches in blocks of 16 quad words (one / two quad groups), 32 in a block of 8 (round 5)"""
    from reverie_amd.dist import HipShardBackend, assemble
    from reverie_amd.proof import challenge, combine_digests

    monkeypatch.setenv("RV_Z64_FUSED", "1")
    prog, w64, wc, st = circuits.layered_z64(n_in=64, width=300, n_mul=2500)
    seeds = np.random.default_rng(reps).integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, [], w64, wc, seeds, threads=8)
    c = rv.Circuit(prog, wc)
    be = HipShardBackend(c)
    shards = [be.commit([], w64, seeds[b:b + reps], b, reps) for b in range(0, 256, reps)]
    try:
        comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
        omit = challenge(comm)
        parts = [be.open(s, omit)[:2] for s in shards]
    finally:
        for s in shards:
            be.destroy(s)
    assert assemble(comm, parts) == want


@pytest.mark.parametrize("fused_verify", [1, 0])
def test_fused_verifier_vs_oracle(rv, oracle, rule_seeds, monkeypatch, fused_verify):
    """the verifier's Z64 half through k_z64_fused<VERIFY> (RV_Z64_FUSED_VERIFY=1; verifier/online.rs:122-183 and
    verifier/preprocess.rs:46-79 at Z64): valid proofs, bit flips all over the proof, a program whose AssertZero fails -- strict
    and reference-compatible answers equal the oracle's"""
    monkeypatch.setenv("RV_Z64_FUSED_VERIFY", str(fused_verify))
    rng = np.random.default_rng(5150)
    for case in range(4):
        if case == 0:
            prog, w64, wc, _ = circuits.layered_z64(n_in=64, width=300, n_mul=2500)
        elif case == 3:
            # a proof of more than 4 MB: it crosses PCIe on the side stream, and the fused verifier runs the quad groups without an
            # opened repetition before the supplied values have arrived (api.hip: split64)
            prog, w64, wc, _ = circuits.layered_z64(n_in=64, width=2048, n_mul=12000)
        else:
            prog, w64, wc = random_z64(rng, n_in=2 * int(rng.integers(1, 10)), n_gates=int(rng.integers(100, 1500)), n_wires=int(rng.integers(8, 100)))
        good = oracle.prove(prog, [], w64, wc, rule_seeds, threads=8)
        c = rv.Circuit(prog, wc)
        assert rv.Proof(good).verify(c) and rv.Proof(good).verify(c, strict=False)
        for pos in rng.integers(0, len(good), 16):
            bad = bytearray(good)
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            for strict in (False, True):
                try:
                    want = (oracle.verify(prog, wc, bytes(bad), strict=strict), None)
                except oracle.OracleError as e:
                    want = (None, e.code)
                try:
                    got = (rv.Proof(bytes(bad)).verify(c, strict=strict), None)
                except rv.ReverieError as e:
                    got = (None, e.code)
                assert got == want, (case, pos, strict)
        c.close()
        idx = np.flatnonzero((prog["domain"] == 1) & (prog["opcode"] == 5))  # SubConst feeding an AssertZero
        if len(idx):
            bad_prog = prog.copy()
            bad_prog[idx[-1]]["imm"] ^= 1
            want = (oracle.verify(bad_prog, wc, good), oracle.verify(bad_prog, wc, good, strict=True))
            assert (rv.Proof(good).verify(bad_prog, wc, strict=False), rv.Proof(good).verify(bad_prog, wc, strict=True)) == want
