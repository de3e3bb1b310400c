"""Streaming prover (rv_stream_*): byte-identical to rv_prove / the oracle whatever the pieces are, with
device memory that does not grow with the number of gates."""
import json
import os

import numpy as np
import pytest

import circuits
from conftest import GOLDEN
from reverie_amd.ops import GF2, OP_DTYPE, Z64, program

pytestmark = pytest.mark.gpu

META = json.load(open(os.path.join(GOLDEN, "proofs.json")))


@pytest.fixture(scope="module")
def rv():
    import reverie_amd

    reverie_amd.Context.default()
    return reverie_amd


def _pieces(prog, w2, w64, cuts):
    """split (prog, witness) at the op indices `cuts`: every piece gets the witness elements its Input gates consume"""
    out = []
    i2 = i64 = 0
    edges = [0] + sorted(set(int(c) for c in cuts if 0 < c < len(prog))) + [len(prog)]
    for a, b in zip(edges[:-1], edges[1:]):
        part = prog[a:b]
        n2 = int(((part["domain"] == 0) & (part["opcode"] == 0)).sum())
        n64 = int(((part["domain"] == 1) & (part["opcode"] == 0)).sum())
        out.append((part, list(w2[i2:i2 + n2]), list(w64[i64:i64 + n64])))
        i2 += n2
        i64 += n64
    return out


def _stream(rv, prog, w2, w64, wc, seeds, cuts1, cuts2=None, same_cuts=False):
    from reverie_amd.stream import StreamingProver

    sp = StreamingProver(wc, seeds=seeds)
    if same_cuts:
        sp.same_cuts()
    for part, a, b in _pieces(prog, w2, w64, cuts1):
        sp.feed(part, a, b)
    comm = sp.commit()
    for part, a, b in _pieces(prog, w2, w64, cuts1 if cuts2 is None else cuts2):
        sp.feed(part, a, b)
    proof = sp.finish()
    info = sp.info
    sp.close()
    assert proof.comm == comm
    return proof, info


SMALL_GOLDEN = sorted(n for n in META if not META[n].get("digest_only"))  # (the 70 000-gate case: test_stream_bench70k)


@pytest.mark.parametrize("name", SMALL_GOLDEN)
def test_stream_golden(rv, rule_seeds, name):
    m = META[name]
    prog = program([tuple(o) for o in m["ops"]]) if m["ops"] else np.zeros(0, OP_DTYPE)
    gold = open(os.path.join(GOLDEN, f"proof_{name}.bin"), "rb").read()
    w2, w64, wc = m["wit_gf2"], [int(x) for x in m["wit_z64"]], tuple(m["wire_counts"])
    hint = prog[prog["domain"] == 3]  # a stream's wire store is sized at begin: SizeHint ops must fit in it
    wc = (max([wc[0]] + [int(x) for x in hint["a"]]), max([wc[1]] + [int(x) for x in hint["b"]]))
    seeds = rule_seeds
    n = len(prog)
    for cuts in ([], [n // 2], list(range(1, n, 3)), list(range(7, n, 50))):
        proof, _ = _stream(rv, prog, w2, w64, wc, seeds, cuts)
        assert bytes(proof) == gold, (name, cuts[:4])


def test_stream_bench70k(rv, oracle, rule_seeds):
    """the 70 000-gate golden case (the reference's bench circuit: 69 BLAKE3 chunks per transcript, the tree, the 64 KiB flush of
    BufferedHasher) through the streaming prover, cut in halves and every 4 099 ops: length and BLAKE3 digest of the golden proof"""
    from conftest import golden_matches, golden_ops

    m = META["bench70k"]
    prog = program(golden_ops(m))
    w2, w64, wc = m["wit_gf2"], [int(x) for x in m["wit_z64"]], tuple(m["wire_counts"])
    n = len(prog)
    for cuts in ([n // 2], list(range(4099, n, 4099))):
        proof, _ = _stream(rv, prog, w2, w64, wc, rule_seeds, cuts)
        assert golden_matches(oracle, "bench70k", m, bytes(proof)), cuts[:3]


@pytest.mark.parametrize("seed", range(6))
def test_stream_random_mixed(rv, oracle, seed):
    """GF(2) + Z64 + B2A with wire reuse, cut at random places (different cuts in the two passes): every carry path
    (transcript tails across BLAKE3 chunks, items across opening bytes, the shared AES block, Z64 mask parity)"""
    rng = np.random.default_rng(4200 + seed)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(150, 700)))
    # SizeHint may only appear where it does not grow the stream's wire store: give the stream the hinted sizes
    hint = prog[prog["domain"] == 3]
    wc = (max([wc[0]] + [int(x) for x in hint["a"]]), max([wc[1]] + [int(x) for x in hint["b"]]))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, w2, w64, wc, seeds)
    assert bytes(rv.Proof.new(prog, w2, w64, wc, seeds=seeds)) == want
    n = len(prog)
    for k in (1, 2, 9):
        c1 = rng.integers(1, n, k)
        c2 = rng.integers(1, n, k + 1)
        proof, info = _stream(rv, prog, w2, w64, wc, seeds, c1, c2)
        assert bytes(proof) == want, (seed, k)
        assert proof.verify(prog, wc)
        assert info["chunks"] >= 1 and info["n_ops"] == n


def test_stream_long_transcripts(rv, oracle, rule_seeds):
    """transcripts of several BLAKE3 chunks and superblocks, cut so that chunk boundaries fall before, at and after
    the 1024-event marks; Z64 transcripts (72 bytes per Mul) next to them"""
    rng = np.random.default_rng(5)
    ops = [GF2.Input(i) for i in range(8)] + [Z64.Input(i) for i in range(3)]
    for i in range(5000):
        a, b = int(rng.integers(0, 24)), int(rng.integers(0, 24))
        d = int(rng.integers(8, 24))
        ops.append(GF2.Mul(d, a, b) if i % 3 else GF2.Add(d, a, b))
        if i % 11 == 0:
            ops.append(Z64.Mul(int(rng.integers(3, 8)), int(rng.integers(0, 8)), int(rng.integers(0, 8))))
    prog = program(ops)
    w2 = rng.integers(0, 2, 8).tolist()
    w64 = [int(x) for x in rng.integers(0, 1 << 63, 3, dtype=np.uint64)]
    wc = (8, 24)
    want = oracle.prove(prog, w2, w64, wc, rule_seeds)
    n = len(prog)
    for cuts in ([1536 + 11], [1023, 1024, 1025, 2048, 3071], list(range(100, n, 137))):
        proof, info = _stream(rv, prog, w2, w64, wc, rule_seeds, cuts)
        assert bytes(proof) == want, cuts[:3]
    # the one-call form
    from reverie_amd.stream import prove_streaming

    proof, info = prove_streaming(prog, w2, w64, wc, seeds=rule_seeds, max_chunk_ops=1024)
    assert bytes(proof) == want and info["chunks"] == (n + 1023) // 1024


def test_stream_errors(rv, rule_seeds):
    from reverie_amd.stream import StreamingProver

    prog = program([GF2.Input(0), GF2.Input(1), GF2.Mul(2, 0, 1), GF2.AddConst(3, 2, 1), GF2.AssertZero(3)])
    sp = StreamingProver((0, 4), seeds=rule_seeds)
    with pytest.raises(rv.ReverieError) as e:
        sp.feed(prog, [1], [])
    assert e.value.code == 2  # witness too short
    with pytest.raises(rv.ReverieError):
        sp.feed(prog, [1, 1], [])  # a failed stream stays failed
    sp.close()
    sp = StreamingProver((0, 4), seeds=rule_seeds)
    with pytest.raises(rv.ReverieError) as e:
        sp.feed(prog, [1, 0], [])
    assert e.value.code == 1  # AssertZero fails
    sp.close()
    sp = StreamingProver((0, 4), seeds=rule_seeds)
    sp.feed(prog, [1, 1], [])
    sp.commit()
    sp.feed(prog[:3], [1, 1], [])
    with pytest.raises(rv.ReverieError) as e:
        sp.finish()  # pass 2 saw fewer ops than pass 1
    assert e.value.code == 9
    sp.close()
    sp = StreamingProver((0, 3), seeds=rule_seeds)
    with pytest.raises(rv.ReverieError) as e:
        sp.feed(prog, [1, 1], [])
    assert e.value.code == 3  # wire out of range
    sp.close()


@pytest.mark.parametrize("threads", ["1", "4"])
def test_stream_pass2_must_feed_pass1s_ops_and_witness(rv, rule_seeds, monkeypatch, threads):
    """Pass 2 with the SAME cuts as pass 1 reuses pass 1's compiled chunks (the default cache): the ops actually fed are still
    digested and compared -- a different program of the same length is RV_E_ARG at finish, not a proof of pass 1's program --
    and so is a different witness (even one that satisfies the circuit)."""
    from reverie_amd.stream import StreamingProver

    monkeypatch.setenv("RV_STREAM_THREADS", threads)
    rng = np.random.default_rng(11)
    prog, wit, wc = circuits.random_gf2(rng, n_in=40, n_gates=3000, n_wires=120, p_assert=0.0)
    # an unconstrained second witness: no AssertZero in the program, so every witness is valid
    other_wit = [b ^ 1 if i % 3 == 0 else b for i, b in enumerate(wit)]
    other = prog.copy()
    k = int(np.nonzero(other["opcode"] == 6)[0][-1])        # the last Mul reads another operand: same length, same counters
    other["a"][k] = (int(other["a"][k]) + 1) % wc[1]

    def run(p2_prog, p2_wit):
        # one feed per pass, cut by the library every 700 ops (threads > 1: the pieces are compiled / checked on worker threads)
        sp = StreamingProver(wc, seeds=rule_seeds, max_chunk_ops=700)
        try:
            sp.feed(prog, wit, [])
            sp.commit()
            sp.feed(p2_prog, p2_wit, [])
            return sp.finish()
        finally:
            sp.close()

    good = run(prog, wit)
    assert bytes(good) == bytes(rv.Proof.new(prog, wit, [], wc, seeds=rule_seeds))
    with pytest.raises(rv.ReverieError) as e:
        run(other, wit)
    assert e.value.code == 9
    with pytest.raises(rv.ReverieError) as e:
        run(prog, other_wit)
    assert e.value.code == 9


def test_stream_layered_bounded_memory(rv, oracle, rule_seeds):
    """the layered workload at two depths with recycled wire indices: the same proof as rv_prove / the oracle, and a
    device footprint that does not move when the circuit gets four times longer"""
    from reverie_amd.stream import prove_streaming

    infos = []
    for layers in (6, 24):
        prog, wit, wc, st = circuits.layered_gf2(layers=layers, width=16384, n_in=512, recycle=True)
        ssa_prog, _, ssa_wc, _ = circuits.layered_gf2(layers=layers, width=16384, n_in=512)
        want = oracle.prove(ssa_prog, wit, [], ssa_wc, rule_seeds)
        assert oracle.prove(prog, wit, [], wc, rule_seeds) == want  # wire numbering does not reach the proof
        proof, info = prove_streaming(prog, wit, [], wc, seeds=rule_seeds, max_chunk_ops=40000)
        assert bytes(proof) == want
        assert bytes(rv.Proof.new(prog, wit, [], wc, seeds=rule_seeds)) == want
        infos.append(info)
    a, b = infos
    assert b["n_ops"] > 3 * a["n_ops"] and b["chunks"] > 3 * a["chunks"]
    assert b["peak_chunk_bytes"] <= 1.1 * a["peak_chunk_bytes"] and b["wire_store_bytes"] <= 1.1 * a["wire_store_bytes"]
    assert b["hash_state_bytes"] <= a["hash_state_bytes"] + 4 * 4 * 256 * 32  # a few more tree levels, nothing else


def test_stream_full_size_configs(rv, rule_seeds):
    """BASELINE configs 4 and 5 at full size through the streaming prover: byte-identical to rv_prove's proofs, with
    a device footprint of well under 1 GB (config 4; the resident prover keeps ~6.4 GB) and under 10 GB (config 5, 10^6
    Z64 MUL; the resident prover keeps ~88 GB)"""
    from reverie_amd.stream import prove_streaming

    prog, wit, wc, st = circuits.layered_gf2(recycle=True)
    assert st["gates"] == 10027008
    c = rv.Circuit(prog, wc)
    want = bytes(rv.Proof.new(c, wit, [], seeds=rule_seeds))
    c.close()
    proof, info = prove_streaming(prog, wit, [], wc, seeds=rule_seeds, max_chunk_ops=1 << 18)
    assert bytes(proof) == want
    assert info["wire_store_bytes"] + info["peak_chunk_bytes"] + info["hash_state_bytes"] + info["proof_bytes"] < 1 << 30
    del proof, want
    prog, w64, wc, st = circuits.layered_z64(n_mul=1_000_000, recycle=True)
    proof, info = prove_streaming(prog, [], w64, wc, seeds=rule_seeds, max_chunk_ops=1 << 16)
    assert info["wire_store_bytes"] + info["peak_chunk_bytes"] + info["hash_state_bytes"] + info["proof_bytes"] < 10 * (1 << 30)
    c = rv.Circuit(prog, wc)
    want = rv.Proof.new(c, [], w64, seeds=rule_seeds)
    assert bytes(proof) == bytes(want)
    assert proof.verify(c)
    c.close()


# ---- the streaming verifier (rv_stream_verify_*): Proof::verify with the ops fed in pieces, bounded device memory ----
def _verify_stream(proof, prog, wc, cuts, strict=True):
    from reverie_amd.stream import StreamingVerifier

    sv = StreamingVerifier(wc, proof)
    edges = [0] + sorted(set(int(c) for c in cuts if 0 < c < len(prog))) + [len(prog)]
    for a, b in zip(edges[:-1], edges[1:]):
        sv.feed(prog[a:b])
    try:
        return sv.finish(strict=strict)
    finally:
        sv.close()


@pytest.mark.parametrize("name", SMALL_GOLDEN)
def test_stream_verify_golden(rv, name):
    m = META[name]
    prog = program([tuple(o) for o in m["ops"]]) if m["ops"] else np.zeros(0, OP_DTYPE)
    gold = open(os.path.join(GOLDEN, f"proof_{name}.bin"), "rb").read()
    wc = tuple(m["wire_counts"])
    hint = prog[prog["domain"] == 3]
    wc = (max([wc[0]] + [int(x) for x in hint["a"]]), max([wc[1]] + [int(x) for x in hint["b"]]))
    n = len(prog)
    for cuts in ([], [n // 2], list(range(1, n, 3)), list(range(7, n, 50))):
        assert _verify_stream(gold, prog, wc, cuts) is True, (name, cuts[:4])
    if n:
        bad = bytearray(gold)
        bad[len(bad) // 2] ^= 4
        try:  # (a flipped length field is RV_E_PROOF_MALFORMED for both verifiers, anything else `false`)
            want = rv.Proof(bytes(bad)).verify(prog, wc)
        except rv.ReverieError as e:
            want = e.code
        try:
            got = _verify_stream(bytes(bad), prog, wc, [n // 3])
        except rv.ReverieError as e:
            got = e.code
        assert got == want and got is not True


@pytest.mark.parametrize("seed", range(5))
def test_stream_verify_matches_resident_verifier(rv, oracle, seed):
    """random GF(2) + Z64 + B2A programs cut at random places: the streaming verifier answers what rv_verify_ex answers -- for
    honest proofs, for proofs with a flipped byte anywhere (strict and reference-compatible), and for proofs whose supplied
    vectors are short (exhausted iterators read as zero, online.rs:124,162,170)"""
    rng = np.random.default_rng(7700 + seed)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(150, 600)))
    hint = prog[prog["domain"] == 3]
    wc = (max([wc[0]] + [int(x) for x in hint["a"]]), max([wc[1]] + [int(x) for x in hint["b"]]))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    proof = rv.Proof.new(prog, w2, w64, wc, seeds=seeds)
    n = len(prog)
    for k in (0, 1, 5):
        cuts = rng.integers(1, n, k) if k else []
        assert _verify_stream(proof, prog, wc, cuts) is True
        assert _verify_stream(proof, prog, wc, cuts, strict=False) is True
    data = bytes(proof)
    for _ in range(12):
        bad = bytearray(data)
        at = int(rng.integers(0, len(bad)))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        for strict in (True, False):
            try:
                want = rv.Proof(bytes(bad)).verify(prog, wc, strict=strict)
            except rv.ReverieError as e:
                want = ("err", e.code)
            try:
                got = _verify_stream(bytes(bad), prog, wc, rng.integers(1, n, 3), strict=strict)
            except rv.ReverieError as e:
                got = ("err", e.code)
            assert got == want, (seed, at, strict)
    # the one-call form
    from reverie_amd.stream import verify_streaming

    ok, info = verify_streaming(prog, wc, proof, max_chunk_ops=1024)
    assert ok and info["n_ops"] == n


def test_stream_verify_long_transcripts_and_strict_checks(rv, oracle, rule_seeds):
    """transcripts of several BLAKE3 chunks cut around the 1024-event marks and away from byte boundaries of the opening
    vectors (a chunk's supplied values start at any bit); a proof of a FALSE statement that the reference-compatible check
    lets through is refused by the strict streaming verifier exactly as by rv_verify"""
    rng = np.random.default_rng(9)
    ops = [GF2.Input(i) for i in range(8)] + [Z64.Input(i) for i in range(3)]
    for i in range(5000):
        a, b = int(rng.integers(0, 24)), int(rng.integers(0, 24))
        d = int(rng.integers(8, 24))
        ops.append(GF2.Mul(d, a, b) if i % 3 else GF2.Add(d, a, b))
        if i % 11 == 0:
            ops.append(Z64.Mul(int(rng.integers(3, 8)), int(rng.integers(0, 8)), int(rng.integers(0, 8))))
    prog = program(ops)
    w2 = rng.integers(0, 2, 8).tolist()
    w64 = [int(x) for x in rng.integers(0, 1 << 63, 3, dtype=np.uint64)]
    wc = (8, 24)
    proof = rv.Proof.new(prog, w2, w64, wc, seeds=rule_seeds)
    n = len(prog)
    for cuts in ([1536 + 11], [1023, 1024, 1025, 2048, 3071], list(range(100, n, 137))):
        assert _verify_stream(proof, prog, wc, cuts) is True
    # a proof of C1 checked against C2 (same transcripts, failing AssertZero gates: tests/circuits.py): the reference-compatible
    # check accepts, the strict one refuses -- the streaming verifier exactly as rv_verify
    c1, c2, a2, a64, cwc = circuits.assert_circuits()
    pf = rv.Proof.new(c1, a2, a64, cwc, seeds=rule_seeds)
    for cprog, want in ((c1, (True, True)), (c2, (True, False))):
        for cuts in ([], [3], [5, 8]):
            assert (_verify_stream(pf, cprog, cwc, cuts, strict=False), _verify_stream(pf, cprog, cwc, cuts, strict=True)) == want


def test_stream_verify_full_size_bounded_memory(rv, rule_seeds):
    """BASELINE config 4 (10^7 gates, wire indices recycled): the streaming verifier accepts the prover's proof and refuses a
    tampered one in well under a GB of device memory next to the proof (rv_verify keeps ~6 GB resident)"""
    from reverie_amd.stream import verify_streaming

    prog, wit, wc, st = circuits.layered_gf2(recycle=True)
    proof = rv.Proof.new(prog, wit, [], wc, seeds=rule_seeds)
    ok, info = verify_streaming(prog, wc, proof)
    assert ok
    assert info["wire_store_bytes"] + info["peak_chunk_bytes"] < (1 << 30)
    bad = bytearray(bytes(proof))
    bad[len(bad) // 3] ^= 0x10
    ok, _ = verify_streaming(prog, wc, bytes(bad))
    assert not ok


@pytest.mark.parametrize("keep_mb", ["0", "1", None])
def test_stream_kept_transcripts(rv, oracle, monkeypatch, keep_mb):
    """rv_stream_same_cuts: pass 1 keeps the LAST chunks' transcripts on the device within RV_STREAM_KEEP_MB and pass 2 takes their
    openings from them (stream.inc: rv_stream::Kept) -- budget 0: every chunk runs twice, as without the promise; 1 MiB: the first
    chunks run twice, the last ones are kept; default: all kept.  GF(2) + Z64 + B2A; a broken promise is RV_E_ARG."""
    from reverie_amd._lib import ReverieError

    if keep_mb is None:
        monkeypatch.delenv("RV_STREAM_KEEP_MB", raising=False)
    else:
        monkeypatch.setenv("RV_STREAM_KEEP_MB", keep_mb)
    rng = np.random.default_rng(808)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=2500)
    hint = prog[prog["domain"] == 3]
    wc = (max([wc[0]] + [int(x) for x in hint["a"]]), max([wc[1]] + [int(x) for x in hint["b"]]))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, w2, w64, wc, seeds)
    n = len(prog)
    cuts = sorted(int(x) for x in rng.integers(1, n, 12))
    proof, info = _stream(rv, prog, w2, w64, wc, seeds, cuts, same_cuts=True)
    assert bytes(proof) == want
    assert (info["kept_mib"] == 0) == (keep_mb == "0")
    other = sorted(int(x) for x in rng.integers(1, n, 5)) + cuts[:4]
    proof, info = _stream(rv, prog, w2, w64, wc, seeds, cuts, other)  # (no promise: nothing kept, any cuts)
    assert bytes(proof) == want and info["kept_mib"] == 0
    if keep_mb is None:  # a chunk of pass 2 that has to run after one that was served from kept transcripts
        with pytest.raises(ReverieError) as e:
            _stream(rv, prog, w2, w64, wc, seeds, cuts, cuts[:6] + [cuts[6] + 1] + cuts[7:], same_cuts=True)
        assert e.value.code == 9  # RV_E_ARG
    # a long GF(2) stream through the one-call form (which makes the promise): many chunks, 1 MiB holds the last few
    from reverie_amd.stream import prove_streaming

    prog, wit, wc, st = circuits.layered_gf2(layers=12, width=8192, n_in=512, recycle=True)
    want = bytes(rv.Proof.new(prog, wit, [], wc, seeds=seeds))
    proof, info = prove_streaming(prog, wit, [], wc, seeds=seeds, max_chunk_ops=20000)
    assert bytes(proof) == want and info["chunks"] >= 5
    if keep_mb == "1":
        assert info["kept_mib"] == 1
