"""GPU parity tests proper: every call goes through the C-ABI (libreverie_amd.so) on a real
MI355X and is compared bit-for-bit with the CPU oracle / committed golden vectors."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import circuits
from conftest import GOLDEN
from reverie_amd.ops import GF2, OP_DTYPE, Z64, program

pytestmark = pytest.mark.gpu

META = json.load(open(os.path.join(GOLDEN, "proofs.json")))
ALL_GOLDEN = sorted(META)


@pytest.fixture(scope="module")
def rv():
    import reverie_amd

    reverie_amd.Context.default()  # raises loudly if the HIP library or the GPU is missing
    return reverie_amd


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_case(name):
    from conftest import golden_ops

    m = META[name]
    ops = golden_ops(m)
    prog = program(ops) if ops else np.zeros(0, OP_DTYPE)
    gold = None if m.get("digest_only") else open(os.path.join(GOLDEN, f"proof_{name}.bin"), "rb").read()
    return m, prog, m["wit_gf2"], [int(x) for x in m["wit_z64"]], tuple(m["wire_counts"]), gold


# ---------------------------------------------------------------- primitives (rows a1, a2, a4, a16)
def test_prg_blocks(rv, oracle):
    from reverie_amd import _lib

    rng = np.random.default_rng(1)
    keys = rng.integers(0, 256, (37, 16), dtype=np.uint8)
    keys[0] = 0
    out = np.zeros((37, 9, 16), np.uint8)
    _lib.check(_lib.lib().rv_hook_prg_blocks(rv.Context.default().handle, _p(keys), C.c_size_t(37), C.c_uint64(5), C.c_size_t(9), _p(out)))
    blk = C.create_string_buffer(16)
    for k in range(37):
        for b in range(9):
            oracle.lib().rvo_prg_block(keys[k].tobytes(), C.c_uint64(5 + b), blk)
            assert out[k, b].tobytes() == blk.raw
    prim = json.load(open(os.path.join(GOLDEN, "primitives.json")))
    z = np.zeros((1, 2, 16), np.uint8)
    zk = np.zeros((1, 16), np.uint8)
    _lib.check(_lib.lib().rv_hook_prg_blocks(rv.Context.default().handle, _p(zk), C.c_size_t(1), C.c_uint64(0), C.c_size_t(2), _p(z)))
    assert z.tobytes().hex() == prim["aes_ctr"][0]["stream"][:64]


def test_expand_seed(rv, oracle, rule_seeds):
    from reverie_amd import _lib

    keys = np.zeros((256, 8, 16), np.uint8)
    _lib.check(_lib.lib().rv_hook_expand_seed(rv.Context.default().handle, _p(rule_seeds), C.c_size_t(256), _p(keys)))
    for r in (0, 1, 17, 255):
        assert (keys[r] == oracle.expand_seed(rule_seeds[r])).all()


def test_sharegen_gf2(rv, oracle):
    """bitsliced AES-CTR mask generator == ShareGen<GF2>::next() (incl. omitted players)"""
    from reverie_amd import _lib

    sg = json.load(open(os.path.join(GOLDEN, "sharegen.json")))
    keys = np.array([[list(bytes.fromhex(k)) for k in row] for row in sg["keys"]], np.uint8)
    for case in sg["cases"]:
        out = np.zeros(sg["n"], np.uint64)
        omit = np.array(case["omit"], np.uint32)
        _lib.check(_lib.lib().rv_hook_sharegen_gf2(rv.Context.default().handle, _p(keys), _p(omit), C.c_size_t(sg["n"]), _p(out)))
        assert ["%016x" % int(x) for x in out] == case["gf2"]
    rng = np.random.default_rng(2)
    for n in (1, 127, 128, 129, 5000):
        keys = rng.integers(0, 256, (8, 8, 16), dtype=np.uint8)
        omit = rng.integers(0, 9, 8).astype(np.uint32)
        out = np.zeros(n, np.uint64)
        _lib.check(_lib.lib().rv_hook_sharegen_gf2(rv.Context.default().handle, _p(keys), _p(omit), C.c_size_t(n), _p(out)))
        assert (out == oracle.sharegen_gf2(keys, omit, n)).all()


def test_reconstruct_hooks(rv, oracle):
    """rows a7 / a8 piecewise: DomainGF2::reconstruct (per-byte parity smeared to 0x00/0xFF) and DomainZ64::reconstruct
    (wrapping sum over the players) through the interpreters' own device functions"""
    from reverie_amd import _lib

    rng = np.random.default_rng(77)
    sh = rng.integers(0, 2**64, 5000, dtype=np.uint64)
    sh[:4] = [0, 2**64 - 1, 0x8000000000000000, 0x0101010101010101]
    out = np.zeros_like(sh)
    _lib.check(_lib.lib().rv_hook_gf2_reconstruct(rv.Context.default().handle, _p(sh), C.c_size_t(len(sh)), _p(out)))
    assert [int(x) for x in out] == [oracle.gf2_reconstruct(int(x)) for x in sh]
    # the reference's own statement of the rule (gf2/domain.rs:47-63): byte i of the result = parity of byte i
    par = np.unpackbits(sh.view(np.uint8).reshape(-1, 8), axis=1).reshape(-1, 8, 8).sum(axis=2) & 1
    assert (out.view(np.uint8).reshape(-1, 8) == par * 255).all()
    z = rng.integers(0, 2**64, (333, 8, 8), dtype=np.uint64)
    z[0] = 2**64 - 1  # wrapping
    zo = np.zeros((333, 8), np.uint64)
    _lib.check(_lib.lib().rv_hook_z64_reconstruct(rv.Context.default().handle, _p(z), C.c_size_t(333), _p(zo)))
    with np.errstate(over="ignore"):
        assert (zo == z.sum(axis=2, dtype=np.uint64)).all()


def test_sharegen_z64(rv, oracle):
    """bitsliced AES + bit-plane transpose == ShareGen<Z64>::next()"""
    import hashlib

    from reverie_amd import _lib

    sg = json.load(open(os.path.join(GOLDEN, "sharegen.json")))
    keys = np.array([[list(bytes.fromhex(k)) for k in row] for row in sg["keys"]], np.uint8)
    for case in sg["cases"]:
        out = np.zeros((sg["n"], 8, 8), np.uint64)
        omit = np.array(case["omit"], np.uint32)
        _lib.check(_lib.lib().rv_hook_sharegen_z64(rv.Context.default().handle, _p(keys), _p(omit), C.c_size_t(sg["n"]), _p(out)))
        zs = [["%016x" % int(v) for v in row.reshape(-1)] for row in out]
        assert zs[:4] == case["z64_first4"] and zs[-1] == case["z64_last"]
        assert hashlib.sha256(json.dumps(zs).encode()).hexdigest() == case["z64_sha256_json"]
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 1001):
        keys = rng.integers(0, 256, (8, 8, 16), dtype=np.uint8)
        omit = rng.integers(0, 9, 8).astype(np.uint32)
        out = np.zeros((n, 8, 8), np.uint64)
        _lib.check(_lib.lib().rv_hook_sharegen_z64(rv.Context.default().handle, _p(keys), _p(omit), C.c_size_t(n), _p(out)))
        assert (out == oracle.sharegen_z64(keys, omit, n)).all()


def test_blake3_streams(rv, oracle):
    from reverie_amd import _lib

    prim = json.load(open(os.path.join(GOLDEN, "primitives.json")))
    for kat in prim["blake3"]:
        n = kat["len"]
        d = np.frombuffer(bytes(i % 251 for i in range(n)), np.uint8)
        data = np.ascontiguousarray(np.stack([d, d[::-1]])) if n else np.zeros((2, 0), np.uint8)
        out = np.zeros((2, 32), np.uint8)
        _lib.check(_lib.lib().rv_hook_blake3(rv.Context.default().handle, _p(data) if n else None, C.c_size_t(2), C.c_size_t(n), _p(out)))
        assert out[0].tobytes().hex() == kat["hash"], n
        buf = C.create_string_buffer(32)
        oracle.lib().rvo_blake3_hash(data[1].tobytes(), C.c_size_t(n), buf)
        assert out[1].tobytes() == buf.raw
    rng = np.random.default_rng(3)
    for n in (1, 1000, 70001):
        data = rng.integers(0, 256, (21, n), dtype=np.uint8)
        out = np.zeros((21, 32), np.uint8)
        _lib.check(_lib.lib().rv_hook_blake3(rv.Context.default().handle, _p(data), C.c_size_t(21), C.c_size_t(n), _p(out)))
        buf = C.create_string_buffer(32)
        for r in range(21):
            oracle.lib().rvo_blake3_hash(data[r].tobytes(), C.c_size_t(n), buf)
            assert out[r].tobytes() == buf.raw


# ---------------------------------------------------------------- whole proofs
@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_golden_proofs(rv, oracle, rule_seeds, name):
    from conftest import golden_matches

    m, prog, w2, w64, wc, gold = load_case(name)
    proof = rv.Proof.new(prog, w2, w64, wc, seeds=rule_seeds)
    assert golden_matches(oracle, name, m, bytes(proof))  # (byte for byte; the 70 000-gate case: length and BLAKE3 digest)
    assert proof.verify(prog, wc)
    assert oracle.verify(prog, wc, bytes(proof))
    if gold is not None:
        assert rv.Proof(gold).verify(prog, wc)


def test_bench_circuit_vs_oracle(rv, oracle, rule_seeds):
    """the reference's bench circuit (proof/mod.rs:318-354): 2 inputs + N x Mul(2,0,1), wire reuse"""
    prog = program([GF2.Input(0), GF2.Input(1)] + [GF2.Mul(2, 0, 1)] * 20000)
    proof = rv.Proof.new(prog, [1, 1], [0], (128, 128), seeds=rule_seeds)
    assert bytes(proof) == oracle.prove(prog, [1, 1], [0], (128, 128), rule_seeds)
    assert proof.verify(prog, (128, 128))


@pytest.mark.parametrize("seed", range(6))
def test_random_programs_vs_oracle(rv, oracle, seed):
    rng = np.random.default_rng(100 + seed)
    prog, wit, wc = circuits.random_gf2(rng, n_in=int(rng.integers(1, 20)), n_gates=int(rng.integers(1, 1500)),
                                        n_wires=int(rng.integers(3, 60)))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, wit, [], wc, seeds)
    proof = rv.Proof.new(prog, wit, [], wc, seeds=seeds)
    assert bytes(proof) == want
    assert proof.verify(prog, wc) and oracle.verify(prog, wc, bytes(proof))


@pytest.mark.parametrize("seed", range(4))
def test_random_mixed_programs_vs_oracle(rv, oracle, seed):
    """GF(2) + Z64 + B2A + SizeHint, random, every op kind"""
    rng = np.random.default_rng(500 + seed)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(20, 400)))
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, w2, w64, wc, seeds)
    proof = rv.Proof.new(prog, w2, w64, wc, seeds=seeds)
    assert bytes(proof) == want
    assert proof.verify(prog, wc) and oracle.verify(prog, wc, bytes(proof))
    bad = bytearray(want)
    bad[len(bad) - 216 * 48 - 100] ^= 4  # inside the z64 online openings
    try:
        w = oracle.verify(prog, wc, bytes(bad))
    except oracle.OracleError:
        w = None
    try:
        g = rv.Proof(bytes(bad)).verify(prog, wc, strict=False)
    except rv.ReverieError:
        g = None
    assert g == w


def test_z64_layered_vs_oracle(rv, oracle, rule_seeds):
    """config-5 generator (Z64 Mul/Add layers) at a size the oracle finishes in seconds"""
    prog, wit, wc, st = circuits.layered_z64(n_in=64, width=256, n_mul=1500)
    want = oracle.prove(prog, [], wit, wc, rule_seeds)
    c = rv.Circuit(prog, wc)
    proof = rv.Proof.new(c, [], wit, seeds=rule_seeds)
    assert bytes(proof) == want
    assert proof.verify(c)
    assert c.info["z64_muls"] == st["mul"]


def test_z64_mid_size_vs_oracle_and_full_size_round_trip(rv, oracle, rule_seeds):
    """config 5: 10^5 Mul gates of the full-size generator byte for byte against the oracle (the aliased
    fresh mask rows, 16 384-wide levels), then the full 10^6-Mul circuit through prove -> strict verify, a second proof
    with the same seeds (deterministic), and a flipped byte in the Z64 openings (must be rejected)"""
    prog, wit, wc, st = circuits.layered_z64(n_mul=100_000)
    assert st["mul"] >= 100_000
    want = oracle.prove(prog, [], wit, wc, rule_seeds, threads=32)
    c = rv.Circuit(prog, wc)
    proof = rv.Proof.new(c, [], wit, seeds=rule_seeds)
    assert bytes(proof) == want
    assert proof.verify(c)
    del proof, want, c
    prog, wit, wc, st = circuits.layered_z64(n_mul=1_000_000)
    c = rv.Circuit(prog, wc)
    proof = rv.Proof.new(c, [], wit, seeds=rule_seeds)
    assert proof.verify(c)
    first = bytes(proof)
    del proof
    again = rv.Proof.new(c, [], wit, seeds=rule_seeds)
    assert bytes(again) == first
    del again
    n = len(first)
    bad = bytearray(first)
    del first
    bad[n - 216 * 48 - 4096] ^= 1
    try:
        ok = rv.Proof(bytes(bad)).verify(c, strict=False)
    except rv.ReverieError:
        ok = False
    assert not ok


def test_layered_circuit_vs_oracle(rv, oracle, rule_seeds):
    """config-4 generator at a size the oracle finishes in seconds"""
    prog, wit, wc, st = circuits.layered_gf2(n_in=256, width=2048, layers=12)
    want = oracle.prove(prog, wit, [], wc, rule_seeds)
    c = rv.Circuit(prog, wc)
    proof = rv.Proof.new(c, wit, [], seeds=rule_seeds)
    assert bytes(proof) == want
    assert proof.verify(c)
    info = c.info
    assert info["gf2_muls"] == st["and"] and info["gf2_inputs"] == 256


def test_commit_digests_vs_oracle(rv, oracle, rule_seeds):
    from reverie_amd import _lib

    prog, wit, wc, _ = circuits.layered_gf2(n_in=64, width=512, layers=5)
    h, st, comm = oracle.commit(prog, wit, [], wc, rule_seeds)
    c = rv.Circuit(prog, wc)
    L = _lib.lib()
    g = np.asarray(wit, np.uint8)
    for begin, count in ((0, 256), (64, 32), (248, 8)):
        sh = C.c_void_p()
        seeds = np.ascontiguousarray(rule_seeds[begin:begin + count])
        _lib.check(L.rv_shard_commit(c.ctx.handle, c.handle, _p(g), C.c_size_t(len(g)), None, C.c_size_t(0), _p(seeds),
                                     C.c_uint32(begin), C.c_uint32(count), C.byref(sh)))
        out = np.zeros((count, 32), np.uint8)
        _lib.check(L.rv_shard_digests(sh, _p(out)))
        sd = np.zeros((count, 4, 32), np.uint8)
        _lib.check(L.rv_hook_shard_stream_digests(sh, _p(sd)))
        L.rv_shard_destroy(sh)
        assert (out == h[begin:begin + count]).all()
        assert (sd == st[begin:begin + count]).all()


# ---------------------------------------------------------------- verifier behaviour
def test_tamper_and_cross_verify(rv, oracle, rule_seeds):
    m, prog, w2, w64, wc, gold = load_case("adder64")
    c = rv.Circuit(prog, wc)
    rng = np.random.default_rng(3)
    for pos in rng.integers(0, len(gold), 40):
        bad = bytearray(gold)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            want = oracle.verify(prog, wc, bytes(bad))
            want_err = None
        except oracle.OracleError as e:
            want, want_err = None, e.code
        try:
            got = rv.Proof(bytes(bad)).verify(c, strict=False)  # (the oracle call above is the reference's verifier)
            got_err = None
        except rv.ReverieError as e:
            got, got_err = None, e.code
        assert (got, got_err) == (want, want_err), pos


def test_verify_fuzz_matches_oracle(rv, oracle):
    """bit flips, length-field corruption and truncations of a mixed GF(2)/Z64/B2A proof: the HIP verifier
    must return exactly what the oracle returns (true / false / malformed) and never crash"""
    rng = np.random.default_rng(777)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=250)
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    good = oracle.prove(prog, w2, w64, wc, seeds)
    c = rv.Circuit(prog, wc)
    assert rv.Proof(good).verify(c)

    def both(data):
        # both verifiers in both modes: the reference's check and the strict one (the boundary's default)
        w, g = [], []
        for strict in (False, True):
            try:
                w.append(oracle.verify(prog, wc, data, strict=strict))
            except oracle.OracleError as e:
                w.append(("err", e.code))
            try:
                g.append(rv.Proof(data).verify(c, strict=strict))
            except rv.ReverieError as e:
                g.append(("err", e.code))
        assert w[1] == g[1], "strict verifiers disagree"
        return w[0], g[0]

    n_false = n_err = 0
    for trial in range(160):
        bad = bytearray(good)
        kind = trial % 4
        if kind == 0:  # single bit flip anywhere
            pos = int(rng.integers(0, len(bad)))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # corrupt a byte early in an online record (omit / keys / length fields)
            pos = 40 + int(rng.integers(0, 4000))
            bad[pos] = int(rng.integers(0, 256))
        elif kind == 2:  # truncate
            bad = bad[:int(rng.integers(0, len(bad)))]
        else:  # overwrite a u64 length field of the first gf2 online record with a small / huge value
            off = 32 + 8 + 1 + 128 + int(rng.integers(0, 2)) * 0
            val = int(rng.choice([0, 1, 7, 8, 9, 2**32, 2**63]))
            bad[off:off + 8] = val.to_bytes(8, "little")
        w, g = both(bytes(bad))
        assert w == g, (trial, kind, w, g)
        n_false += w is False
        n_err += isinstance(w, tuple)
    assert n_false > 20 and n_err > 20


def test_errors(rv, rule_seeds):
    m, prog, w2, w64, wc, gold = load_case("adder64")
    bad = list(w2)
    bad[3] ^= 1
    with pytest.raises(rv.ReverieError) as e:
        rv.Proof.new(prog, bad, [], wc, seeds=rule_seeds)
    assert e.value.code == 1  # invalid witness (prover.rs:221-228 panics)
    with pytest.raises(rv.ReverieError) as e:
        rv.Proof.new(prog, w2[:-1], [], wc, seeds=rule_seeds)
    assert e.value.code == 2  # witness too short
    with pytest.raises(rv.ReverieError) as e:
        rv.Circuit(prog, (0, 10))
    assert e.value.code == 3  # wire out of range
    with pytest.raises(rv.ReverieError) as e:
        rv.Proof(gold[:500]).verify(prog, wc)
    assert e.value.code == 4
    # wrong repetition count is `false`, not an error (proof/mod.rs:225-230)
    m, prog, w2, w64, wc, gold = load_case("empty")
    off = len(gold) - 216 * 48 - 8
    cut = gold[:off] + (215).to_bytes(8, "little") + gold[off + 8:-48]
    assert rv.Proof(cut).verify(prog, wc) is False


def test_os_seeds_prove_verify(rv, oracle):
    """seeds=NULL draws from the OS like the reference's OsRng; proof must verify on both sides"""
    m, prog, w2, w64, wc, gold = load_case("gf2_mix")
    p1 = rv.Proof.new(prog, w2, w64, wc)
    p2 = rv.Proof.new(prog, w2, w64, wc)
    assert bytes(p1) != bytes(p2)
    assert p1.verify(prog, wc) and oracle.verify(prog, wc, bytes(p2))


def test_full_size_bit_exact_vs_oracle(rv, oracle, rule_seeds):
    """BASELINE config 4 at FULL size (10^7 gates, 5.0e6 AND, 50 MB proof): the GPU proof equals the oracle's byte
    for byte (the oracle needs a few seconds of 8-32 threads and ~3.5 GB for it)."""
    prog, wit, wc, st = circuits.layered_gf2()
    c = rv.Circuit(prog, wc)
    got = rv.Proof.new(c, wit, [], seeds=rule_seeds)
    want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=min(32, os.cpu_count() or 1))
    assert len(got) == len(want) == 50196120
    assert bytes(got) == want


def test_full_size_all_and_bit_exact_vs_oracle(rv, oracle, rule_seeds):
    """the all-AND variant of config 4 (10 027 008 AND gates, 100 MB proof), bit-exact against the oracle, then verified"""
    prog, wit, wc, st = circuits.layered_gf2(p_and=1.0)
    assert st["and"] == 10027008
    c = rv.Circuit(prog, wc)
    got = rv.Proof.new(c, wit, [], seeds=rule_seeds)
    want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=min(32, os.cpu_count() or 1))
    assert bytes(got) == want
    assert got.verify(c)
    c.close()


@pytest.mark.parametrize("layers,p_and", [(64, 0.5), (33, 1.0), (100, 0.5)])
def test_mid_size_transcript_trees_bit_exact_vs_oracle(rv, oracle, rule_seeds, layers, p_and):
    """Between the small cases and config 4 the two transcript hashes take other shapes (csrc/kernels.hip, launch_b3_pair_big: one shared
    launch for the chunks of both streams, ONE shared reduction launch of three tree levels, a shared tree top of 257 ... 1 024 nodes per
    repetition): 2.1 - 3.3 * 10^6 AND gates, the oracle's bytes for the first proof (plain openings) and the second (early corrections,
    broadcast vectors written to the host by the extraction kernel), then verified."""
    prog, wit, wc, st = circuits.layered_gf2(layers=layers, p_and=p_and)
    assert 2048 <= (st["and"] + st["inputs"] + 1023) // 1024 <= 8192
    want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=min(32, os.cpu_count() or 1))
    c = rv.Circuit(prog, wc, whole_prover=True)
    for _ in range(2):
        got = rv.Proof.new(c, wit, [], seeds=rule_seeds)
        assert bytes(got) == want
    assert got.verify(c)
    c.close()


def test_full_size_properties(rv, rule_seeds):
    """BASELINE config 4 at full size (10^7 gates): size-independent properties only —
    prove -> verify accepts, a flipped transcript bit is rejected, proof length is as derived."""
    prog, wit, wc, st = circuits.layered_gf2()
    c = rv.Circuit(prog, wc)
    proof = rv.Proof.new(c, wit, [], seeds=rule_seeds)
    n_rec = st["and"] + 128
    expect = 33160 + 40 * ((n_rec // 8) + (st["and"] // 8) + (st["inputs"] // 8))
    assert len(proof) == expect
    assert proof.verify(c)
    bad = bytearray(bytes(proof))
    bad[len(bad) // 3] ^= 0x10
    assert not rv.Proof(bytes(bad)).verify(c)


def test_prove_batch_chunks_and_buffer_modes(rv, oracle, monkeypatch):
    """A batch larger than what one pass may hold runs as consecutive chunks (forced here with RV_BATCH_MAX), and the
    proofs come out either as slices of one page-locked buffer (default) or as separate buffers
    (RV_BATCH_COPY_OUT=1): the same bytes every time, freed in any order."""
    rng = np.random.default_rng(31)
    prog, wit, wc = circuits.random_gf2(rng, n_in=10, n_gates=500, n_wires=30)
    c = rv.Circuit(prog, wc)
    nb = 11
    seeds = rng.integers(0, 256, (nb, 256, 16), dtype=np.uint8)
    wits = np.tile(np.asarray(wit, np.uint8), (nb, 1))
    want = [bytes(rv.Proof.new(c, wit, [], seeds=seeds[b])) for b in range(nb)]
    assert want[3] == oracle.prove(prog, wit, [], wc, seeds[3], threads=2)
    for env in ({}, {"RV_BATCH_MAX": "4"}, {"RV_BATCH_COPY_OUT": "1"}, {"RV_BATCH_MAX": "1"}):
        for k in ("RV_BATCH_MAX", "RV_BATCH_COPY_OUT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = rv.Proof.new_batch(c, wits, seeds=seeds)
        keep = [bytes(g) for g in got]
        # release out of order, with a new batch in between (the shared buffer must survive until its last slice goes)
        del got[::2]
        again = rv.Proof.new_batch(c, wits[:3], seeds=seeds[:3])
        assert [bytes(g) for g in got] == keep[1::2]
        del got
        assert keep == want and [bytes(g) for g in again] == want[:3]


def test_prove_batch_witness_strides_and_no_inputs(rv, oracle):
    """Witness rows wider than the circuit's input count (the batch upload is a strided 2-D copy) and a circuit
    without any Input gate (constants only): each proof still equals the single-proof path."""
    rng = np.random.default_rng(41)
    prog, wit, wc = circuits.random_gf2(rng, n_in=9, n_gates=300, n_wires=30)
    c = rv.Circuit(prog, wc)
    nb = 6
    seeds = rng.integers(0, 256, (nb, 256, 16), dtype=np.uint8)
    wide = np.zeros((nb, len(wit) + 7), np.uint8)
    wide[:, :len(wit)] = np.asarray(wit, np.uint8)
    wide[:, len(wit):] = 1  # never read
    got = rv.Proof.new_batch(c, wide, seeds=seeds)
    for b in range(nb):
        assert bytes(got[b]) == bytes(rv.Proof.new(c, wit, [], seeds=seeds[b]))
    assert bytes(got[2]) == oracle.prove(prog, wit, [], wc, seeds[2], threads=2)
    # no inputs at all
    ops = [GF2.Const(0, 1), GF2.Const(1, 1), GF2.Mul(2, 0, 1), GF2.AddConst(3, 2, 1), GF2.AssertZero(3), GF2.Random(4),
           GF2.Mul(5, 4, 2), GF2.Add(6, 5, 5), GF2.AssertZero(6)]
    p0 = program(ops)
    c0 = rv.Circuit(p0, (0, 7))
    g0 = rv.Proof.new_batch(c0, np.zeros((3, 0), np.uint8), seeds=seeds[:3])
    for b in range(3):
        assert bytes(g0[b]) == oracle.prove(p0, [], [], (0, 7), seeds[b], threads=2)
        assert g0[b].verify(c0, strict=True)


def test_prove_batch_equals_single_proofs(rv, oracle, rule_seeds):
    """rv_prove_batch: B proofs of one circuit, different witnesses and seeds, must each equal the single-proof entry
    point and the oracle — on a circuit with narrow runs of both kinds and launched levels (AES-128: valid and
    invalid witnesses), and on a mixed GF(2)/Z64 circuit (fallback path)."""
    import bristol_gen
    from reverie_amd import bristol
    from reverie_amd._lib import ReverieError

    bits = lambda d: [(b >> (7 - k)) & 1 for b in d for k in range(8)]  # noqa: E731
    # statement: AES-128 of the witness (key || plaintext) -- no output assertion, so every witness is valid
    prog, info = bristol.parse(bristol_gen.aes128())
    wc = info["wire_counts"]
    rng = np.random.default_rng(2024)
    B = 5
    wits = rng.integers(0, 2, (B, 256), dtype=np.uint8)
    seeds = rng.integers(0, 256, (B, 256, 16), dtype=np.uint8)
    c = rv.Circuit(prog, wc)
    got = rv.Proof.new_batch(c, wits, seeds=seeds)
    assert len(got) == B
    for b in range(B):
        assert bytes(got[b]) == bytes(rv.Proof.new(c, wits[b], [], seeds=seeds[b])) == oracle.prove(prog, wits[b], [], wc, seeds[b], threads=2)
        assert got[b].verify(c)
    # a larger batch (more proofs than side streams) of a small random circuit: every proof against the single path
    rng2 = np.random.default_rng(77)
    progr, witr, wcr = circuits.random_gf2(rng2, n_in=12, n_gates=900, n_wires=40)
    cr = rv.Circuit(progr, wcr)
    nb = 21
    seedsr = rng2.integers(0, 256, (nb, 256, 16), dtype=np.uint8)
    gr = rv.Proof.new_batch(cr, np.tile(np.asarray(witr, np.uint8), (nb, 1)), seeds=seedsr)
    for b in range(nb):
        assert bytes(gr[b]) == bytes(rv.Proof.new(cr, witr, [], seeds=seedsr[b]))
    assert bytes(gr[20]) == oracle.prove(progr, witr, [], wcr, seedsr[20], threads=2)
    # one invalid witness fails the whole call, like a panic in one rayon task would
    key = bytes(range(16)); pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    prog2, info2 = bristol.parse(bristol_gen.aes128(), expected_outputs=bits(bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")))
    c2 = rv.Circuit(prog2, info2["wire_counts"])
    good = np.array(bits(key) + bits(pt), np.uint8)
    bad = good.copy(); bad[3] ^= 1
    ok2 = rv.Proof.new_batch(c2, np.stack([good, good]), seeds=seeds[:2])
    assert bytes(ok2[1]) == bytes(rv.Proof.new(c2, good, [], seeds=seeds[1]))
    with pytest.raises(ReverieError):
        rv.Proof.new_batch(c2, np.stack([good, bad, good]), seeds=seeds[:3])
    # a circuit of more than 2^20 gates takes the two-proofs-in-flight path (two host threads, worker contexts)
    progl, witl, wcl, stl = circuits.layered_gf2(layers=17)
    cl = rv.Circuit(progl, wcl)
    wl = np.tile(np.asarray(witl, np.uint8), (3, 1))
    try:
        gl = rv.Proof.new_batch(cl, wl[[0, 0, 0]], seeds=seeds[:3])
        for b in range(3):
            assert bytes(gl[b]) == bytes(rv.Proof.new(cl, witl, [], seeds=seeds[b]))
        assert bytes(gl[0]) == oracle.prove(progl, witl, [], wcl, seeds[0], threads=4)
    finally:
        cl.close()
    # mixed circuit: falls back to one rv_prove per proof, same results
    progm, w2, w64, wcm = circuits.random_mixed(np.random.default_rng(5), n_gates=200)
    cm = rv.Circuit(progm, wcm)
    gm = rv.Proof.new_batch(cm, np.stack([w2, w2]), np.stack([w64, w64]), seeds=seeds[:2])
    for b in range(2):
        assert bytes(gm[b]) == oracle.prove(progm, w2, w64, wcm, seeds[b], threads=2)


def test_mask_generator_beside_the_levels(rv, oracle, rule_seeds, monkeypatch):
    """RV_OVERLAP (round 5, on by default for circuits of >= 8192 cipher blocks): the lane-distributed mask generator
    (csrc/aes_col4.hip; replaces generator/share.rs:54-65 + gf2/domain.rs:66-378 like k_aes_gf2_masks) runs chunk by chunk on a
    stream of its own BESIDE the level launches that read earlier chunks.  Threshold lowered so that circuits the oracle can prove
    take the path: whole proofs (both compiler hints), 128- and 64-repetition shards, a mixed GF(2) / Z64 circuit beside; the
    bytes with RV_OVERLAP=0; and the path really taken (rv_hook_overlap_commits)."""
    from reverie_amd import _lib
    from reverie_amd.dist import HipShardBackend
    from reverie_amd.proof import challenge, combine_digests

    L = _lib.lib()
    monkeypatch.setenv("RV_OVERLAP_MIN", "1000")
    monkeypatch.setenv("RV_EARLY_MIN", "100000")  # (the early-corrections path beside it, as on the benchmark circuit)
    prog, wit, wc, st = circuits.layered_gf2(n_in=64, width=8192, layers=80, fold_to=16)
    want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=4)
    for hint in (True, False):
        c = rv.Circuit(prog, wc, whole_prover=hint)
        n0 = L.rv_hook_overlap_commits()
        for _ in range(2):
            proof = rv.Proof.new(c, wit, [], seeds=rule_seeds)
            assert bytes(proof) == want
        assert L.rv_hook_overlap_commits() == n0 + 2
        assert proof.verify(c)
        monkeypatch.setenv("RV_OVERLAP", "0")
        assert bytes(rv.Proof.new(c, wit, [], seeds=rule_seeds)) == want and L.rv_hook_overlap_commits() == n0 + 2
        monkeypatch.delenv("RV_OVERLAP")
        if not hint:
            # repetition shards: 128 and 64 repetitions take the path (rows of 32 / 16 quad words), 32 keep the 128-plane generator
            for per in (128, 64, 32):
                be = HipShardBackend(c)
                n1 = L.rv_hook_overlap_commits()
                shards = [be.commit(wit, [], rule_seeds[b:b + per], b, per) for b in range(0, 256, per)]
                assert L.rv_hook_overlap_commits() == n1 + (256 // per if per >= 64 else 0)
                comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
                omit = challenge(comm)
                parts = [be.open(s, omit)[:2] for s in shards]
                for s in shards:
                    be.destroy(s)
                from reverie_amd.dist import assemble

                assert assemble(comm, parts) == want
        c.close()
    progm, w2, w64, wcm = circuits.random_mixed(np.random.default_rng(31), n_gates=400)
    cm = rv.Circuit(progm, wcm)
    pm = rv.Proof.new(cm, w2, w64, seeds=rule_seeds)
    assert bytes(pm) == oracle.prove(progm, w2, w64, wcm, rule_seeds, threads=2) and pm.verify(cm)
    cm.close()


def test_prove_device_invalid_witness(rv, rule_seeds):
    """rv_prove_device defers the invalid-witness check to its single synchronisation: it must still report it"""
    import torch

    from reverie_amd._lib import ReverieError
    from reverie_amd.dist import HipShardBackend

    prog = program([GF2.Input(0), GF2.Input(1), GF2.Mul(2, 0, 1), GF2.AssertZero(2)])
    c = rv.Circuit(prog, (0, 3))
    be = HipShardBackend(c)
    buf = torch.empty(max(sum(be.single_shard_sizes()), 1), dtype=torch.uint8, device="cuda")
    comm, omit, lens = be.prove_device([1, 0], [], rule_seeds, buf)
    assert len(comm) == 32 and int((omit < 8).sum()) == 40 and lens == be.single_shard_sizes()
    with pytest.raises(ReverieError) as e:
        be.prove_device([1, 1], [], rule_seeds, buf)
    assert e.value.code == 1  # RV_E_WITNESS_INVALID


def test_single_shard_host_and_device_fiat_shamir(rv, oracle):
    """One shard holding all 256 repetitions: the host-side challenge path (digests -> rv_combine_digests ->
    rv_challenge -> open) and the device-side one (rv_shard_open_self) must give the oracle's proof."""
    from reverie_amd.dist import HipShardBackend, assemble_device_parts, prove_sharded
    from reverie_amd.proof import challenge, combine_digests

    rng = np.random.default_rng(777)
    for n_gates in (1, 300):
        prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=n_gates)
        seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
        want = oracle.prove(prog, w2, w64, wc, seeds, threads=2)
        c = rv.Circuit(prog, wc)
        be = HipShardBackend(c)
        assert prove_sharded(be, w2, w64, seeds) == want  # host challenge
        comm, bufs, lens = prove_sharded(be, w2, w64, seeds, device_resident=True)  # device challenge
        assert assemble_device_parts(comm, bufs, lens) == want
        # the opening map the device derived is the host's
        shard = be.commit(w2, w64, seeds, 0, 256)
        try:
            import torch

            h = be.digests(shard)
            buf = torch.empty(max(sum(lens[0]), 1), dtype=torch.uint8, device="cuda")
            comm2, omit2, _ = be.open_self(shard, buf)
            assert comm2 == combine_digests(h) == want[:32]
            assert (omit2 == challenge(comm2)).all()
        finally:
            be.destroy(shard)


@pytest.mark.parametrize("n_shards", [2, 4, 8])
def test_shards_in_one_process_bristol_circuits(rv, oracle, rule_seeds, n_shards):
    """Repetition shards of 128 / 64 / 32 (row widths 32 / 16 / 8 quads, the per-GPU shapes of 2 / 4 / 8 GPUs) run one
    after the other on this GPU: AES-128 (levels of 64-256 gates: the class-loop narrow kernel) and SHA-256 (levels of
    <=32 gates: the per-gate narrow kernel), host-side challenge, both the prover and the sharded verifier."""
    import bristol_gen
    from reverie_amd import _lib, bristol
    from reverie_amd.dist import HipShardBackend, assemble, shard_range
    from reverie_amd.proof import challenge, combine_digests

    bits = lambda d: [(b >> (7 - k)) & 1 for b in d for k in range(8)]  # noqa: E731
    key = bytes(range(16)); pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    cases = [(bristol_gen.aes128(), bits(bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")), bits(key) + bits(pt))]
    if n_shards == 4:
        import hashlib

        block = b"abc" + b"\x80" + bytes(52) + (24).to_bytes(8, "big")
        cases.append((bristol_gen.sha256_block(), bits(hashlib.sha256(b"abc").digest()), bits(block)))
    for text, expect, wit in cases:
        prog, info = bristol.parse(text, expected_outputs=expect)
        wc = info["wire_counts"]
        want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=4)
        c = rv.Circuit(prog, wc)
        be = HipShardBackend(c)
        shards = []
        try:
            for r in range(n_shards):
                b, n = shard_range(r, n_shards)
                shards.append(be.commit(wit, [], rule_seeds[b:b + n], b, n))
            h = np.concatenate([be.digests(s) for s in shards])
            comm = combine_digests(h)
            omit = challenge(comm)
            parts = [be.open(s, omit)[:2] for s in shards]
        finally:
            for s in shards:
                be.destroy(s)
        proof = assemble(comm, parts)
        assert proof == want
        # the same with the challenge derived on the device from the "gathered" digests (rv_shard_open_gathered: what the
        # RCCL path does after its all-gather), every shard into a worst-case sized buffer
        import torch

        allh = torch.from_numpy(h.reshape(-1).copy()).to("cuda")
        shards = []
        try:
            parts2 = []
            for r in range(n_shards):
                b, n = shard_range(r, n_shards)
                sh = be.commit(wit, [], rule_seeds[b:b + n], b, n)
                shards.append(sh)
                buf = torch.empty(be.gathered_capacity(n), dtype=torch.uint8, device="cuda")
                comm2, omit2, lens2 = be.open_gathered(sh, allh, buf)
                assert comm2 == comm and (omit2 == omit).all() and lens2 == parts[r][1]
                parts2.append((buf.cpu().numpy().tobytes()[:sum(lens2)], lens2))
        finally:
            for sh in shards:
                be.destroy(sh)
        assert assemble(comm, parts2) == want
        # sharded verifier (rv_verify_shard over slot ranges, then rv_verify_finish)
        L = _lib.lib()
        dig = np.zeros((256, 32), np.uint8)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        per = 256 // n_shards
        for r in range(n_shards):
            part = np.zeros((per, 32), np.uint8)
            _lib.check(L.rv_verify_shard(c.ctx.handle, c.handle, buf, C.c_size_t(len(proof)), C.c_uint32(r * per), C.c_uint32(per),
                                         part.ctypes.data_as(C.c_void_p)))
            dig[r * per:(r + 1) * per] = part
        ok = C.c_int()
        _lib.check(L.rv_verify_finish(buf, C.c_size_t(len(proof)), dig.ctypes.data_as(C.c_void_p), C.byref(ok)))
        assert ok.value == 1


# ---------------------------------------------------------------- sharded path on the GPU
def _two_rank_worker(rank, world, port, out_path):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    import circuits as cg
    import oracle_lib
    import reverie_amd
    from reverie_amd.dist import HipShardBackend, assemble_device_parts, prove_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(4242)
    prog, w2, w64, wc = cg.random_mixed(rng, n_gates=300)
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    c = reverie_amd.Circuit(prog, wc, reverie_amd.Context(0))  # both ranks share GPU 0 (gloo for the 8 KiB)
    be = HipShardBackend(c)
    proof = prove_sharded(be, w2, w64, seeds)
    comm, bufs, lens = prove_sharded(be, w2, w64, seeds, device_resident=True)
    if rank == 0:
        want = oracle_lib.prove(prog, w2, w64, wc, seeds, threads=2)
        ok = proof == want and assemble_device_parts(comm, bufs, lens) == want
        open(out_path, "w").write("ok" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,device_path", [(2, False), (4, False), (2, True), (4, True)])
def test_sharded_hip_backend_ranks_share_gpu(tmp_path, monkeypatch, world, device_path):
    """N>1 orchestration with the REAL HIP shard backend (R = 128 / 64 per rank: the generic-NQ kernels); the box has a
    single GPU, so the ranks share it and talk over gloo.  device_path: the branch the RCCL run takes -- digests
    all-gathered as device tensors, challenge on the GPU (rv_shard_open_gathered), openings sent to rank 0 as device
    tensors -- exercised here with gloo carrying the CUDA tensors (RV_DIST_DEVICE_PATH=1)."""
    import socket

    import torch.multiprocessing as mp

    if device_path:
        monkeypatch.setenv("RV_DIST_DEVICE_PATH", "1")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.txt")
    mp.spawn(_two_rank_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


# ---------------------------------------------------------------- RV_VERIFY_STRICT
def test_strict_verify_zero_checks(rv, oracle, rule_seeds):
    """A proof of C1 checked against C2 (same transcripts, failing AssertZero gates): the reference's verifier — and
    RV_VERIFY_REFERENCE_COMPAT — accept it, the default (strict) rv_verify rejects it; every answer equals the oracle's."""
    c1, c2, w2, w64, wc = circuits.assert_circuits()
    pf = rv.Proof.new(c1, w2, w64, wc, seeds=rule_seeds)
    assert bytes(pf) == oracle.prove(c1, w2, w64, wc, rule_seeds)
    c3 = c1.copy()
    c3[8]["imm"] = 41  # only the Z64 assertion fails
    c4 = c1.copy()
    c4[3]["imm"] = 0  # only the GF(2) one
    for prog, want in ((c1, (True, True)), (c2, (True, False)), (c3, (True, False)), (c4, (True, False))):
        got = (pf.verify(prog, wc, strict=False), pf.verify(prog, wc, strict=True))
        assert got == want
        assert pf.verify(prog, wc) == got[1]  # strict is the default
        assert got == (oracle.verify(prog, wc, bytes(pf)), oracle.verify(prog, wc, bytes(pf), strict=True))
    # wide levels go through the per-level kernels, deep ones through the single-workgroup kernel: a failing
    # assertion in either must be seen
    rng = np.random.default_rng(5)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=400)
    seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
    pf = rv.Proof.new(prog, w2, w64, wc, seeds=seeds)
    assert pf.verify(prog, wc, strict=True) and oracle.verify(prog, wc, bytes(pf), strict=True)
    flipped = 0
    for i in np.flatnonzero((prog["opcode"] == 3) | (prog["opcode"] == 5)):  # AddConst / SubConst feeding assertions
        bad = prog.copy()
        bad[i]["imm"] ^= 1
        want = (oracle.verify(bad, wc, bytes(pf)), oracle.verify(bad, wc, bytes(pf), strict=True))
        assert (pf.verify(bad, wc, strict=False), pf.verify(bad, wc, strict=True)) == want
        flipped += want == (True, False)
        if flipped >= 3:
            break


def test_strict_verify_omit_must_match_challenge(rv, oracle, rule_seeds):
    """A prover that opens the challenged repetitions but hides ANOTHER player than the challenge names: consistent
    transcripts, so the reference accepts (proof/mod.rs:292-302 only checks which repetitions are opened);
    RV_VERIFY_STRICT compares the records' `omit` with the challenge."""
    from reverie_amd.dist import HipShardBackend, assemble
    from reverie_amd.proof import challenge, combine_digests

    m, prog, w2, w64, wc, gold = load_case("adder64")
    c = rv.Circuit(prog, wc)
    be = HipShardBackend(c)
    shard = be.commit(w2, w64, rule_seeds, 0, 256)
    try:
        comm = combine_digests(be.digests(shard))
        omit = challenge(comm)
        blob, lens, _, _ = be.open(shard, omit)
        honest = assemble(comm, [(blob, lens)])
        assert honest == gold
        forged_omit = omit.copy()
        k = int(np.flatnonzero(omit < 8)[7])
        forged_omit[k] = (omit[k] + 3) % 8
        blob, lens, _, _ = be.open(shard, forged_omit)
        forged = assemble(comm, [(blob, lens)])
    finally:
        be.destroy(shard)
    assert forged != honest
    for pf, want in ((honest, (True, True)), (forged, (True, False))):
        assert (rv.Proof(pf).verify(c, strict=False), rv.Proof(pf).verify(c, strict=True)) == want
        assert rv.Proof(pf).verify(c) == want[1]
        assert (oracle.verify(prog, wc, pf), oracle.verify(prog, wc, pf, strict=True)) == want


# ---------------------------------------------------------------- boundary shapes
@pytest.mark.parametrize("n_in,width,layers,p_and", [
    (1, 1, 40, 0.5), (127, 63, 9, 0.5), (128, 64, 9, 1.0), (129, 65, 9, 0.5), (5, 255, 12, 0.5), (64, 256, 12, 0.7),
    (300, 257, 12, 0.5), (2, 513, 5, 0.5), (1000, 1025, 4, 0.3), (16, 4097, 3, 0.5), (4096, 8192, 2, 0.0)])
def test_level_width_boundaries_vs_oracle(rv, oracle, n_in, width, layers, p_and):
    """Level widths around the kernels' switch points (one gate per wavefront <-> unrolled class loops at 32 gates, the
    single-workgroup narrow runs <-> one launch per level at 256, the 1 024-record LDS window, the 4 096-workgroup cap)
    and input counts around the 128-mask AES block; an all-XOR circuit has no Mul at all.  Whole proofs against the
    oracle, as one shard and as 32-repetition shards."""
    from reverie_amd.dist import HipShardBackend, assemble
    from reverie_amd.proof import challenge, combine_digests

    prog, wit, wc, st = circuits.layered_gf2(n_in=n_in, width=width, layers=layers, p_and=p_and, seed=n_in * 7919 + width, fold_to=width)
    seeds = np.random.default_rng(width).integers(0, 256, (256, 16), dtype=np.uint8)
    want = oracle.prove(prog, wit, [], wc, seeds, threads=4)
    c = rv.Circuit(prog, wc)
    got = rv.Proof.new(c, wit, [], seeds=seeds)
    assert bytes(got) == want
    assert got.verify(c, strict=True)
    be = HipShardBackend(c)
    shards = [be.commit(wit, [], seeds[b:b + 32], b, 32) for b in range(0, 256, 32)]
    try:
        comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
        omit = challenge(comm)
        parts = [be.open(s, omit)[:2] for s in shards]
    finally:
        for s in shards:
            be.destroy(s)
    assert assemble(comm, parts) == want


@pytest.mark.parametrize("n_shards", [2, 8])
def test_mixed_circuit_shards_vs_oracle(rv, oracle, n_shards):
    """GF(2) + Z64 + B2A circuits as repetition shards (the Z64 interpreter keeps two players per lane, four lanes per
    repetition, whatever the shard width): prover shards against the oracle's proof, verifier shards against
    rv_verify, odd numbers of Z64 inputs in front of the 64-byte transcript events included."""
    from reverie_amd import _lib
    from reverie_amd.dist import HipShardBackend, assemble
    from reverie_amd.proof import challenge, combine_digests

    L = _lib.lib()
    for seed in (11, 12, 13):
        rng = np.random.default_rng(seed)
        prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=250)
        seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
        want = oracle.prove(prog, w2, w64, wc, seeds, threads=4)
        c = rv.Circuit(prog, wc)
        be = HipShardBackend(c)
        n = 256 // n_shards
        shards = [be.commit(w2, w64, seeds[b:b + n], b, n) for b in range(0, 256, n)]
        try:
            comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
            omit = challenge(comm)
            parts = [be.open(s, omit)[:2] for s in shards]
        finally:
            for s in shards:
                be.destroy(s)
        assert assemble(comm, parts) == want
        # sharded verifier: slot ranges of 256 / n_shards, then the final check
        buf = np.frombuffer(want, np.uint8)
        dig = np.zeros((256, 32), np.uint8)
        zc_all = 1
        for b in range(0, 256, n):
            zc = C.c_int(1)
            part = np.zeros((n, 32), np.uint8)
            _lib.check(L.rv_verify_shard_ex(c.ctx.handle, c.handle, _p(buf), C.c_size_t(len(buf)), C.c_uint32(b), C.c_uint32(n), _p(part),
                                            C.byref(zc)))
            dig[b:b + n] = part
            zc_all &= zc.value
        ok = C.c_int()
        _lib.check(L.rv_verify_finish_ex(_p(buf), C.c_size_t(len(buf)), _p(dig), C.c_uint32(1), C.c_int(zc_all), C.byref(ok)))
        assert ok.value == 1 and zc_all == 1


# ---------------------------------------------------------------- rv_verify_batch
def test_verify_batch_matches_single_verifier(rv, oracle, monkeypatch):
    """rv_verify_batch answers, proof by proof, what rv_verify_ex answers (and the oracle): valid proofs, flipped payload
    and commitment bytes, wrong repetition counts (`false`), the assertion gap with and without RV_VERIFY_STRICT, a
    mixed circuit (proof-after-proof fallback), chunked batches; unparsable bytes fail the call like they do alone."""
    from reverie_amd._lib import ReverieError

    for seed in range(3):
        rng = np.random.default_rng(100 + seed)
        prog, wit, wc = circuits.random_gf2(rng, n_in=12, n_gates=700, n_wires=40)
        c = rv.Circuit(prog, wc)
        nb = 9
        seeds = rng.integers(0, 256, (nb, 256, 16), dtype=np.uint8)
        proofs = rv.Proof.new_batch(c, np.tile(np.asarray(wit, np.uint8), (nb, 1)), seeds=seeds)
        assert rv.verify_batch(c, proofs, strict=True) == [True] * nb  # Proof objects, in place
        blobs = [bytes(p) for p in proofs]
        for k in (1, 4, 7):  # a payload byte of an online record
            b = bytearray(blobs[k])
            b[200 + 37 * k] ^= 1 << (k % 8)
            blobs[k] = bytes(b)
        b = bytearray(blobs[5]); b[3] ^= 1; blobs[5] = bytes(b)  # a commitment byte
        short = blobs[8]
        off = len(short) - 216 * 48 - 8
        blobs[8] = short[:off] + (215).to_bytes(8, "little") + short[off + 8:-48]  # 215 preprocessing records: `false`
        for env in ({}, {"RV_BATCH_MAX": "4"}):
            monkeypatch.delenv("RV_BATCH_MAX", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for strict in (False, True):
                # a proof that cannot be parsed raises on its own and is simply rejected inside a batch
                want = []
                for bl in blobs:
                    try:
                        want.append(rv.Proof(bl).verify(c, strict=strict))
                    except ReverieError as e:
                        assert e.code == 4
                        want.append(None)
                assert rv.verify_batch(c, blobs, strict=strict) == [bool(w) for w in want]
                good = [bl for bl, w in zip(blobs, want) if w is not None]
                assert rv.verify_batch(c, good, strict=strict) == [w for w in want if w is not None]
                assert want[8] is False and want[0] is True
                assert [oracle.verify(prog, wc, bl, strict=strict) for bl, w in zip(blobs, want) if w is not None] == [w for w in want if w is not None]
            # malformed proofs in the batch: truncated bytes and an out-of-range `omit`
            cut = blobs[0][:len(blobs[0]) // 2]
            bad_omit = bytearray(blobs[1])
            bad_omit[32 + 8] = 9
            mixed = [blobs[0], cut, blobs[1], bytes(bad_omit), blobs[2]]
            assert rv.verify_batch(c, mixed) == [rv.Proof(blobs[0]).verify(c), False, rv.Proof(blobs[1]).verify(c), False, rv.Proof(blobs[2]).verify(c)]
            assert rv.verify_batch(c, [cut, cut]) == [False, False] and rv.verify_batch(c, [cut]) == [False]
        monkeypatch.delenv("RV_BATCH_MAX", raising=False)
    # the reference's assertion gap, batched: a pure GF(2) statement checked against other constants
    prog1 = program([GF2.Input(0), GF2.Input(1), GF2.Mul(2, 0, 1), GF2.AddConst(3, 2, 1), GF2.AssertZero(3)] +
                    [GF2.Mul(4 + i, 2, i % 2) for i in range(40)])
    prog2 = prog1.copy()
    prog2[3]["imm"] = 0
    seeds = np.random.default_rng(9).integers(0, 256, (5, 256, 16), dtype=np.uint8)
    c1, c2 = rv.Circuit(prog1, (0, 44)), rv.Circuit(prog2, (0, 44))
    proofs = rv.Proof.new_batch(c1, np.ones((5, 2), np.uint8), seeds=seeds)
    assert rv.verify_batch(c1, proofs, strict=False) == [True] * 5 and rv.verify_batch(c1, proofs, strict=True) == [True] * 5
    assert rv.verify_batch(c2, proofs, strict=False) == [True] * 5 and rv.verify_batch(c2, proofs) == [False] * 5
    # a mixed circuit falls back to one proof after the other
    cm1, cm2, w2, w64, wcm = circuits.assert_circuits()
    pm = rv.Proof.new(cm1, w2, w64, wcm, seeds=seeds[0])
    assert rv.verify_batch(cm2, [pm, pm], wcm, strict=False) == [True, True] and rv.verify_batch(cm2, [pm, pm], wcm) == [False, False]
    assert rv.verify_batch(c1, []) == []


def test_library_communicator_world1(rv, oracle, rule_seeds):
    """rv_comm_* / rv_prove_sharded / rv_prove_multi with a communicator of one rank (all this box has): RCCL is found
    and initialised, the collective entry point produces rv_prove's bytes.  (More ranks need more GPUs: RCCL refuses two
    ranks on one device; the shard kernels themselves are covered by the sharded tests above.)"""
    from reverie_amd import _lib
    from reverie_amd.dist import LibComm

    rng = np.random.default_rng(31)
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=300)
    c = rv.Circuit(prog, wc)
    want = oracle.prove(prog, w2, w64, wc, rule_seeds)
    lc = LibComm(c)
    assert (lc.rank, lc.world) == (0, 1)
    for _ in range(2):
        p = rv.Proof(_owned=lc.prove(w2, w64, rule_seeds))
        assert bytes(p) == want
    # rv_prove_multi over the same communicator (one host thread per rank; here one)
    g = np.ascontiguousarray(np.asarray(w2, np.uint8))
    z = np.ascontiguousarray(np.asarray(w64, np.uint64))
    comms = (C.c_void_p * 1)(lc.handle)
    circs = (C.c_void_p * 1)(c.handle)
    out, n = C.c_void_p(), C.c_size_t()
    _lib.check(_lib.lib().rv_prove_multi(comms, circs, C.c_int(1), _p(g), C.c_size_t(len(g)), _p(z), C.c_size_t(len(z)), _p(rule_seeds),
                                         C.byref(out), C.byref(n)))
    assert bytes(rv.Proof(_owned=(C.c_void_p(out.value), n.value))) == want
    # an invalid witness is reported, not hung on
    bad = list(w2)
    bad[0] ^= 1
    try:
        r = lc.prove(bad, w64, rule_seeds)
        if r:
            _lib.lib().rv_free(r[0])
    except rv.ReverieError as e:
        assert e.code == 1
    lc.close()
    # rv_comm_create_all: the single-process form
    ctxs = (C.c_void_p * 1)(c.ctx.handle)
    cm = (C.c_void_p * 1)()
    _lib.check(_lib.lib().rv_comm_create_all(ctxs, C.c_int(1), cm))
    out, n = C.c_void_p(), C.c_size_t()
    _lib.check(_lib.lib().rv_prove_sharded(C.c_void_p(cm[0]), c.handle, _p(g), C.c_size_t(len(g)), _p(z), C.c_size_t(len(z)), _p(rule_seeds),
                                           C.byref(out), C.byref(n)))
    assert bytes(rv.Proof(_owned=(C.c_void_p(out.value), n.value))) == want
    _lib.lib().rv_comm_destroy(C.c_void_p(cm[0]))


def _hourglass(seed=5):
    """600 inputs (a wide level), 40 narrow layers of 32 gates reading them, then 1024 Mul gates (a wide level again) that
    read wires of the narrow part and inputs, folded by Add gates into asserts: the narrow stretch has live-in AND
    live-out wires"""
    prog, wit, wc, st = circuits.layered_gf2(n_in=600, width=32, layers=40, fold_to=32, seed=0x5EED0000000000AA + seed)
    rng = np.random.default_rng(seed)
    # clear values of every wire of the SSA part (inputs, then layer outputs), to make the final asserts valid
    vals = np.zeros(wc[1] + 4096, np.uint8)
    vals[:600] = wit
    for op in prog:
        oc, d, a, b = int(op["opcode"]), int(op["dst"]), int(op["a"]), int(op["b"])
        if oc == circuits.OP_MUL:
            vals[d] = vals[a] & vals[b]
        elif oc == circuits.OP_ADD:
            vals[d] = vals[a] ^ vals[b]
        elif oc == circuits.OP_ADDCONST:
            vals[d] = vals[a] ^ (int(op["imm"]) & 1)
    n0 = wc[1]
    wide = np.zeros(1024, circuits.OP_DTYPE)
    wide["domain"] = circuits.DOM_GF2
    wide["opcode"] = circuits.OP_MUL
    wide["dst"] = np.arange(n0, n0 + 1024, dtype=np.uint32)
    wide["a"] = rng.integers(600 + 32 * 30, 600 + 32 * 40, 1024).astype(np.uint32)   # the last ten narrow layers
    wide["b"] = np.where(rng.random(1024) < 0.5, rng.integers(0, 600, 1024), rng.integers(600, 600 + 32 * 40, 1024)).astype(np.uint32)
    vals[n0:n0 + 1024] = vals[wide["a"]] & vals[wide["b"]]
    tail = []
    cur = list(range(n0, n0 + 1024))
    nxt = n0 + 1024
    while len(cur) > 4:
        half = len(cur) // 2
        t = np.zeros(half, circuits.OP_DTYPE)
        t["domain"] = circuits.DOM_GF2
        t["opcode"] = circuits.OP_ADD
        t["dst"] = np.arange(nxt, nxt + half, dtype=np.uint32)
        t["a"] = np.array(cur[:half], np.uint32)
        t["b"] = np.array(cur[half:], np.uint32)
        vals[nxt:nxt + half] = vals[t["a"]] ^ vals[t["b"]]
        tail.append(t)
        cur = list(range(nxt, nxt + half))
        nxt += half
    t = np.zeros(2 * len(cur), circuits.OP_DTYPE)
    t["domain"] = circuits.DOM_GF2
    t["opcode"][0::2] = circuits.OP_ADDCONST
    t["dst"][0::2] = np.arange(nxt, nxt + len(cur), dtype=np.uint32)
    t["a"][0::2] = np.array(cur, np.uint32)
    t["imm"][0::2] = vals[cur].astype(np.uint64)
    t["opcode"][1::2] = circuits.OP_ASSERTZERO
    t["a"][1::2] = np.arange(nxt, nxt + len(cur), dtype=np.uint32)
    nxt += len(cur)
    return np.ascontiguousarray(np.concatenate([prog, wide] + tail + [t])), wit, (0, nxt)


@pytest.mark.parametrize("qs", ["0", "1", "2", "4"])
def test_lds_runs(rv, oracle, rule_seeds, monkeypatch, qs):
    """Narrow stretches with the live wires in LDS (csrc/ldsrun.*; slice width RV_LDS_QS): prover and verifier must agree
    with the oracle byte for byte -- random narrow programs with every op kind (Random gates, constants, wire reuse,
    asserts), mixed GF(2) / Z64 / B2A programs (narrow GF(2) stretches between Z64 levels), a circuit whose narrow stretch
    has live-in and live-out wires, a rejected witness, tampered proofs, and repetition shards of 128 / 64 / 32"""
    from reverie_amd.dist import HipShardBackend, assemble
    from reverie_amd.proof import challenge, combine_digests

    monkeypatch.setenv("RV_LDS_QS", qs)
    rng = np.random.default_rng(1234 + int(qs))
    for trial in range(5):
        prog, wit, wc = circuits.random_gf2(rng, n_in=int(rng.integers(1, 60)), n_gates=int(rng.integers(100, 4000)), n_wires=int(rng.integers(8, 300)))
        seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
        want = oracle.prove(prog, wit, [], wc, seeds)
        proof = rv.Proof.new(prog, wit, [], wc, seeds=seeds)
        assert bytes(proof) == want, trial
        assert proof.verify(prog, wc), trial
        bad = bytearray(want)
        bad[int(rng.integers(40, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        try:
            w = oracle.verify(prog, wc, bytes(bad))
        except oracle.OracleError:
            w = None
        try:
            g = rv.Proof(bytes(bad)).verify(prog, wc, strict=False)
        except rv.ReverieError:
            g = None
        assert g == w, trial
    for trial in range(3):
        prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=int(rng.integers(100, 600)))
        seeds = rng.integers(0, 256, (256, 16), dtype=np.uint8)
        want = oracle.prove(prog, w2, w64, wc, seeds)
        proof = rv.Proof.new(prog, w2, w64, wc, seeds=seeds)
        assert bytes(proof) == want, trial
        assert proof.verify(prog, wc), trial
    prog, wit, wc = _hourglass()
    want = oracle.prove(prog, wit, [], wc, rule_seeds)
    c = rv.Circuit(prog, wc)
    proof = rv.Proof.new(c, wit, [], seeds=rule_seeds)
    assert bytes(proof) == want
    assert proof.verify(c)
    monkeypatch.setenv("RV_LDS_RUN", "0")
    assert bytes(rv.Proof.new(prog, wit, [], wc, seeds=rule_seeds)) == want  # (the row path on the same circuit)
    monkeypatch.delenv("RV_LDS_RUN")
    bad = wit.copy()
    bad[3] ^= 1
    try:
        oracle.prove(prog, bad, [], wc, rule_seeds)
        rejected = False
    except oracle.OracleError:
        rejected = True
    if rejected:
        with pytest.raises(rv.ReverieError) as e:
            rv.Proof.new(c, bad, [], seeds=rule_seeds)
        assert e.value.code == 1
    # batched proofs: the LDS runs serve small batches, one workgroup per proof (k_interp_narrow_b) the large ones
    nb = 4
    bseeds = rng.integers(0, 256, (nb, 256, 16), dtype=np.uint8)
    singles = [bytes(rv.Proof.new(c, wit, [], seeds=bseeds[b])) for b in range(nb)]
    assert singles[0] == oracle.prove(prog, wit, [], wc, bseeds[0])
    for limit in ("100000", "0"):
        monkeypatch.setenv("RV_LDS_BATCH_WGS", limit)
        got = rv.Proof.new_batch(c, np.tile(np.asarray(wit, np.uint8), (nb, 1)), seeds=bseeds)
        assert [bytes(g) for g in got] == singles, limit
        assert rv.verify_batch(c, got, strict=True) == [True] * nb, limit
    monkeypatch.delenv("RV_LDS_BATCH_WGS")
    be = HipShardBackend(c)
    for per in (128, 64, 32):
        shards = [be.commit(wit, [], rule_seeds[b:b + per], b, per) for b in range(0, 256, per)]
        comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
        omit = challenge(comm)
        parts = [be.open(s, omit)[:2] for s in shards]
        for s in shards:
            be.destroy(s)
        assert assemble(comm, parts) == want, per


def test_compile_hint_whole_prover(rv, oracle, rule_seeds):
    """rv_circuit_compile_ex(RV_COMPILE_WHOLE_PROVER): a different gate stream (linear gates kept as lazy sums), the same
    bytes from every entry point -- whole proofs, batches, repetition shards, the verifier -- on a wide layered circuit,
    a mixed GF(2) / Z64 / B2A program and a Bristol-style narrow one"""
    from reverie_amd.dist import HipShardBackend, assemble
    from reverie_amd.proof import challenge, combine_digests

    rng = np.random.default_rng(99)
    cases = [circuits.layered_gf2(n_in=300, width=4096, layers=9)[:3]]
    prog, w2, w64, wc = circuits.random_mixed(rng, n_gates=500)
    cases.append((prog, w2, wc, w64))
    prog, w2, wc = circuits.random_gf2(rng, n_in=40, n_gates=3000, n_wires=120)
    cases.append((prog, w2, wc))
    for case in cases:
        prog, w2, wc = case[:3]
        w64 = case[3] if len(case) > 3 else []
        want = oracle.prove(prog, w2, w64, wc, rule_seeds)
        hinted = rv.Circuit(prog, wc, whole_prover=True)
        plain = rv.Circuit(prog, wc)
        assert hinted.info["gf2_rows_written"] <= plain.info["gf2_rows_written"]
        p1 = rv.Proof.new(hinted, w2, w64, seeds=rule_seeds)
        assert bytes(p1) == want
        assert bytes(rv.Proof.new(plain, w2, w64, seeds=rule_seeds)) == want
        assert p1.verify(hinted) and p1.verify(plain)
        if not len(w64):
            got = rv.Proof.new_batch(hinted, np.tile(np.asarray(w2, np.uint8), (3, 1)), seeds=np.tile(rule_seeds, (3, 1, 1)))
            assert all(bytes(g) == want for g in got)
        be = HipShardBackend(hinted)
        shards = [be.commit(w2, w64, rule_seeds[b:b + 64], b, 64) for b in range(0, 256, 64)]
        comm = combine_digests(np.concatenate([be.digests(s) for s in shards]))
        omit = challenge(comm)
        parts = [be.open(s, omit)[:2] for s in shards]
        for s in shards:
            be.destroy(s)
        assert assemble(comm, parts) == want
    # the wide circuit really compiles differently (unless RV_LAZY_K fixes the choice for both)
    prog, w2, wc = cases[0]
    if not os.environ.get("RV_LAZY_K"):
        assert rv.Circuit(prog, wc, whole_prover=True).info["gf2_rows_written"] < rv.Circuit(prog, wc).info["gf2_rows_written"]


def test_early_corrections_path(rv, oracle, rule_seeds, monkeypatch):
    """rv_prove's early-corrections path (the corrections vectors of ALL repetitions cross PCIe before the challenge
    exists, the host copies the 40 opened ones into the proof, a kernel writes the rest around them: csrc/api.hip) on
    circuits small enough for the oracle, with the size threshold lowered: the oracle's bytes for every chunk count,
    Mul counts that are and are not multiples of 8 (the always-present last byte), the plain path's bytes with
    RV_EARLY=0, an invalid witness reported, and the path really taken (rv_hook_early_proofs)."""
    from reverie_amd import _lib

    L = _lib.lib()
    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    # (the last case is mostly XOR: its outputs still depend on the inputs after 80 layers, so a flipped witness bit is caught)
    cases = [(64, 8192, 40, 0.5, "10", 16), (37, 4736, 70, 0.6, "3", 37), (64, 8192, 36, 1.0, "1", 16), (128, 16384, 24, 0.5, "16", 16),
             (64, 16384, 80, 0.1, "4", 16)]
    for case_no, (n_in, width, layers, p_and, chunks, fold_to) in enumerate(cases):
        monkeypatch.setenv("RV_EARLY_CHUNKS", chunks)
        # (cases 1 and 3: only the first 96 / 200 repetitions are staged, the opened ones beyond them take the plain way)
        if case_no in (1, 3):
            monkeypatch.setenv("RV_EARLY_REPS", "96" if case_no == 1 else "200")
        else:
            monkeypatch.delenv("RV_EARLY_REPS", raising=False)
        monkeypatch.setenv("RV_EARLY", "2")
        prog, wit, wc, st = circuits.layered_gf2(n_in=n_in, width=width, layers=layers, p_and=p_and, fold_to=fold_to)
        want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=4)
        assert len(want) > (1 << 20), "the case must be large enough for a page-locked proof buffer"
        c = rv.Circuit(prog, wc)
        n0 = L.rv_hook_early_proofs()
        got = rv.Proof.new(c, wit, [], seeds=rule_seeds)
        assert L.rv_hook_early_proofs() == n0 + 1, "the early-corrections path was not taken"
        assert bytes(got) == want, (width, layers, chunks, st["and"] % 8)
        # (rv_circuit_info reports the page-locked staging the path holds per context: every repetition's corrections vector)
        assert c.info["early_staging_bytes"] >= 256 * ((st["and"] + 7) // 8)
        again = rv.Proof.new(c, wit, [], seeds=rule_seeds)  # the staging buffers and the mailbox are reused
        assert bytes(again) == want
        assert got.verify(c)
        monkeypatch.setenv("RV_EARLY", "0")
        plain = rv.Proof.new(c, wit, [], seeds=rule_seeds)
        assert c.info["early_staging_bytes"] == 0  # (RV_EARLY=0)
        assert L.rv_hook_early_proofs() == n0 + 2
        assert bytes(plain) == want
        monkeypatch.setenv("RV_EARLY", "2")
        if p_and < 0.2:
            bad = wit.copy()
            bad[0] ^= 1
            with pytest.raises(rv.ReverieError) as e:
                rv.Proof.new(c, bad, [], seeds=rule_seeds)
            assert e.value.code == 1
            assert L.rv_hook_early_proofs() == n0 + 2
            assert bytes(rv.Proof.new(c, wit, [], seeds=rule_seeds)) == want  # ... and the context is fine afterwards
        c.close()


@pytest.mark.parametrize("direct", ["0/1", "1/1", "1/2", "1/7"])
def test_early_corrections_openings_direct(rv, oracle, rule_seeds, monkeypatch, direct):
    """The early path's second half (csrc/internal.h: OpenDirect): the opened repetitions' broadcast vectors written into the
    page-locked proof buffer by the extraction kernel itself -- whole 16-byte aligned words, each workgroup every word that STARTS in
    its tile -- and k_copy_gaps bringing the rest of the image: the oracle's bytes with none, all, half and a seventh of the tiles
    sent that way, with all and with only 96 of the repetitions staged (the records beyond them keep their corrections in the
    image), Mul counts that are and are not multiples of 8."""
    from reverie_amd import _lib

    L = _lib.lib()
    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    monkeypatch.setenv("RV_EARLY", "2")
    monkeypatch.setenv("RV_OPEN_DIRECT", direct)
    # (vectors of 32 KiB and more: tiles of 16 bytes, the smallest the path takes)
    for n_in, width, layers, p_and, reps, fold_to in [(64, 16384, 36, 0.5, None, 16), (37, 9472, 70, 0.6, "96", 37), (64, 16384, 30, 1.0, None, 16)]:
        if reps:
            monkeypatch.setenv("RV_EARLY_REPS", reps)
        else:
            monkeypatch.delenv("RV_EARLY_REPS", raising=False)
        prog, wit, wc, st = circuits.layered_gf2(n_in=n_in, width=width, layers=layers, p_and=p_and, fold_to=fold_to)
        want = oracle.prove(prog, wit, [], wc, rule_seeds, threads=4)
        c = rv.Circuit(prog, wc)
        n0 = L.rv_hook_open_direct_proofs()
        for _ in range(2):
            assert bytes(rv.Proof.new(c, wit, [], seeds=rule_seeds)) == want, (direct, width, layers, st["and"] % 8)
        assert L.rv_hook_open_direct_proofs() == n0 + (0 if direct == "0/1" else 2), "the direct openings were not (or should not have been) taken"
        c.close()


@pytest.mark.parametrize("reps", ["64", "128", "256"])
def test_early_corrections_path_z64(rv, oracle, rule_seeds, monkeypatch, reps):
    """The Z64 form of the early-corrections path (a repetition's corrections vector is its preprocessing transcript: word
    ranges of the first r_spec repetitions' rows cross PCIe as 2-D copies before the challenge; the opened repetitions
    beyond r_spec are extracted the plain way): the oracle's bytes with 64 / 128 / 256 staged repetitions, the plain
    path's bytes, an invalid witness reported."""
    from reverie_amd import _lib

    L = _lib.lib()
    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    monkeypatch.setenv("RV_EARLY", "2")
    monkeypatch.setenv("RV_EARLY_REPS", reps)
    monkeypatch.setenv("RV_EARLY_CHUNKS", "5")
    prog, w64, wc, st = circuits.layered_z64(n_in=64, width=2048, n_mul=24000)
    assert st["mul"] % 2 == 0
    want = oracle.prove(prog, [], w64, wc, rule_seeds, threads=4)
    assert len(want) > (1 << 20)
    c = rv.Circuit(prog, wc)
    n0 = L.rv_hook_early_proofs()
    got = rv.Proof.new(c, [], w64, seeds=rule_seeds)
    assert L.rv_hook_early_proofs() == n0 + 1, "the early-corrections path was not taken"
    assert bytes(got) == want
    assert bytes(rv.Proof.new(c, [], w64, seeds=rule_seeds)) == want
    assert got.verify(c)
    monkeypatch.setenv("RV_EARLY", "0")
    assert bytes(rv.Proof.new(c, [], w64, seeds=rule_seeds)) == want
    assert L.rv_hook_early_proofs() == n0 + 2
    monkeypatch.setenv("RV_EARLY", "2")
    bad = list(w64)
    bad[0] ^= 1
    with pytest.raises(rv.ReverieError) as e:
        rv.Proof.new(c, [], bad, seeds=rule_seeds)
    assert e.value.code == 1
    assert bytes(rv.Proof.new(c, [], w64, seeds=rule_seeds)) == want
    c.close()


def test_early_corrections_two_circuits_one_context(rv, oracle, rule_seeds, monkeypatch):
    """The early-corrections staging buffers belong to the context: proofs of two circuits of different sizes alternate on it
    (the buffers grow once), a GF(2) and a Z64 one, with OS-drawn seeds too (those proofs must verify)."""
    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    monkeypatch.setenv("RV_EARLY", "2")
    monkeypatch.setenv("RV_EARLY_CHUNKS", "4")
    pa, wa, wca, _ = circuits.layered_gf2(n_in=64, width=8192, layers=30, p_and=0.5, fold_to=16)
    pb, wb, wcb, _ = circuits.layered_gf2(n_in=64, width=16384, layers=40, p_and=0.6, fold_to=16)
    pz, wz, wcz, _ = circuits.layered_z64(n_in=64, width=2048, n_mul=20000)
    ca, cb, cz = rv.Circuit(pa, wca), rv.Circuit(pb, wcb), rv.Circuit(pz, wcz)
    want_a = oracle.prove(pa, wa, [], wca, rule_seeds, threads=4)
    want_b = oracle.prove(pb, wb, [], wcb, rule_seeds, threads=4)
    want_z = oracle.prove(pz, [], wz, wcz, rule_seeds, threads=4)
    for _ in range(2):
        assert bytes(rv.Proof.new(ca, wa, [], seeds=rule_seeds)) == want_a
        assert bytes(rv.Proof.new(cz, [], wz, seeds=rule_seeds)) == want_z
        assert bytes(rv.Proof.new(cb, wb, [], seeds=rule_seeds)) == want_b
    for c, w2, w64 in ((ca, wa, []), (cb, wb, []), (cz, [], wz)):
        p = rv.Proof.new(c, w2, w64)  # seeds from the OS
        assert p.verify(c)
        assert oracle.verify(pa if c is ca else pb if c is cb else pz, wca if c is ca else wcb if c is cb else wcz, bytes(p))
    for c in (ca, cb, cz):
        c.close()


def test_early_corrections_two_contexts_two_threads(rv, oracle, rule_seeds, monkeypatch):
    """two contexts, a host thread each, proving at the same time with the early-corrections path on (every
    context has its own staging buffers, mailbox and helper threads; the mailbox waits sleep through most of the expected wait
    instead of spinning: api.hip, mailbox_wait).  Both streams of proofs are the oracle's, run after run."""
    import threading

    from reverie_amd import _lib

    monkeypatch.setenv("RV_EARLY_MIN", "1000")
    monkeypatch.setenv("RV_EARLY", "2")
    progs = [circuits.layered_gf2(n_in=64, width=8192, layers=30, p_and=0.5, fold_to=16),
             circuits.layered_gf2(n_in=64, width=16384, layers=24, p_and=0.6, fold_to=16, seed=77)]
    wants = [oracle.prove(p, w, [], wc, rule_seeds, threads=4) for p, w, wc, _ in progs]
    ctxs = [rv.Context(0), rv.Context(0)]
    circs = [rv.Circuit(p, wc, ctx=ctx) for (p, w, wc, _), ctx in zip(progs, ctxs)]
    n0 = _lib.lib().rv_hook_early_proofs()
    bad = []

    def work(i):
        try:
            for _ in range(6):
                if bytes(rv.Proof.new(circs[i], progs[i][1], [], seeds=rule_seeds)) != wants[i]:
                    bad.append(i)
        except Exception as e:  # noqa: BLE001
            bad.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad
    assert _lib.lib().rv_hook_early_proofs() == n0 + 12
    for c in circs:
        c.close()
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("vc", [1, 0])
def test_verifier_compact_corrections_vs_oracle(rv, oracle, rule_seeds, monkeypatch, vc):
    """MODE_VERIFY_C (one u64 of public corrections per share row instead of corr rows; csrc/kernels.hip, internal.h) on circuits
    that take it -- pure GF(2), wide levels, one base per wire -- against the oracle's verifier (verifier/online.rs:122-183,
    verifier/preprocess.rs:46-79): valid proofs, bit flips all over the proof, a program whose AssertZero fails (strict and
    reference-compatible answers), random programs with constants, Random gates and wire reuse."""
    monkeypatch.setenv("RV_VERIFY_VC", str(vc))
    rng = np.random.default_rng(4242)
    prog, wit, wc, st = circuits.layered_gf2(n_in=300, width=1500, layers=9, fold_to=1500)
    good = oracle.prove(prog, wit, [], wc, rule_seeds, threads=8)
    c = rv.Circuit(prog, wc)
    assert bytes(rv.Proof.new(c, wit, [], seeds=rule_seeds)) == good
    from reverie_amd import _lib

    n0 = _lib.lib().rv_hook_verify_vc_count()
    assert rv.Proof(good).verify(c) and rv.Proof(good).verify(c, strict=False)
    assert _lib.lib().rv_hook_verify_vc_count() == n0 + (2 if vc else 0)  # (the path this test is about was taken / was not)
    for pos in rng.integers(0, len(good), 24):
        bad = bytearray(good)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        res = []
        for strict in (False, True):
            try:
                want = (oracle.verify(prog, wc, bytes(bad), strict=strict), None)
            except oracle.OracleError as e:
                want = (None, e.code)
            try:
                got = (rv.Proof(bytes(bad)).verify(c, strict=strict), None)
            except rv.ReverieError as e:
                got = (None, e.code)
            assert got == want, (pos, strict)
            res.append(got)
    c.close()
    # a failing assertion: the last AddConst before an AssertZero flipped
    idx = np.flatnonzero(prog["opcode"] == 3)
    if len(idx):
        bad_prog = prog.copy()
        bad_prog[idx[-1]]["imm"] ^= 1
        want = (oracle.verify(bad_prog, wc, good), oracle.verify(bad_prog, wc, good, strict=True))
        assert (rv.Proof(good).verify(bad_prog, wc, strict=False), rv.Proof(good).verify(bad_prog, wc, strict=True)) == want
    for seed in range(4):
        r2 = np.random.default_rng(900 + seed)
        p2, w2, wc2 = circuits.random_gf2(r2, n_in=int(r2.integers(1, 60)), n_gates=int(r2.integers(200, 5000)), n_wires=int(r2.integers(8, 300)))
        seeds = r2.integers(0, 256, (256, 16), dtype=np.uint8)
        pf = oracle.prove(p2, w2, [], wc2, seeds, threads=8)
        assert rv.Proof(pf).verify(p2, wc2) and oracle.verify(p2, wc2, pf, strict=True)
        bad = bytearray(pf)
        bad[len(bad) // 2] ^= 4
        try:
            want = (oracle.verify(p2, wc2, bytes(bad), strict=True), None)
        except oracle.OracleError as e:
            want = (None, e.code)
        try:
            got = (rv.Proof(bytes(bad)).verify(p2, wc2), None)
        except rv.ReverieError as e:
            got = (None, e.code)
        assert got == want


def test_ops_cache_repeat_proof_new_on_the_same_op_list(rv, rule_seeds, monkeypatch):
    """Proof::new takes the raw op list at every call (proof/mod.rs:119-124): rv_prove_ops keeps the compiled circuit by CONTENT in
    the context (round 5), so the second call on the same ops is a cache hit with the same bytes; another circuit misses; the cache
    holds RV_OPS_CACHE (default 2) circuits, least recently used out; RV_OPS_CACHE=0 keeps nothing; verify_ops shares the mechanism."""
    import oracle_lib
    from reverie_amd import _lib

    L = _lib.lib()
    ctx = rv.Context(0)

    def circuit(seed, n=3000):
        rng = np.random.default_rng(seed)
        ops = [GF2.Input(i) for i in range(32)]
        w = 32
        for _ in range(n):
            a, b = int(rng.integers(0, w)), int(rng.integers(0, w))
            ops.append(GF2.Mul(w, a, b) if rng.integers(0, 2) else GF2.Add(w, a, b))
            w += 1
        return program(ops), [int(x) for x in rng.integers(0, 2, 32)], (0, w)

    pa, wa, wca = circuit(1)
    pb, wb, wcb = circuit(2)
    pc, wc_, wcc = circuit(3)
    want_a = oracle_lib.prove(pa, wa, [], wca, rule_seeds)
    h0 = L.rv_hook_ops_cache_hits()
    p1 = rv.Proof.new(pa, wa, [], wca, seeds=rule_seeds, ctx=ctx)
    assert L.rv_hook_ops_cache_hits() == h0 and bytes(p1) == want_a
    p2 = rv.Proof.new(pa, wa, [], wca, seeds=rule_seeds, ctx=ctx)
    assert L.rv_hook_ops_cache_hits() == h0 + 1 and bytes(p2) == want_a
    # a copy of the array with the same content is the same circuit; one changed operand is not
    p3 = rv.Proof.new(pa.copy(), wa, [], wca, seeds=rule_seeds, ctx=ctx)
    assert L.rv_hook_ops_cache_hits() == h0 + 2 and bytes(p3) == want_a
    assert bytes(rv.Proof.new(pb, wb, [], wcb, seeds=rule_seeds, ctx=ctx)) == oracle_lib.prove(pb, wb, [], wcb, rule_seeds)
    assert L.rv_hook_ops_cache_hits() == h0 + 2
    # the verifier's compile (no prover hint) is an entry of its own: first call a miss, second a hit
    assert p1.verify(pa, wca, ctx=ctx) and L.rv_hook_ops_cache_hits() == h0 + 2  # (evicts A's prover circuit or B's: cap 2)
    assert p1.verify(pa, wca, ctx=ctx) and L.rv_hook_ops_cache_hits() == h0 + 3
    # least recently used leaves: C comes in, then A's prover form must be compiled again (miss), bytes unchanged
    assert bytes(rv.Proof.new(pc, wc_, [], wcc, seeds=rule_seeds, ctx=ctx)) == oracle_lib.prove(pc, wc_, [], wcc, rule_seeds)
    h1 = L.rv_hook_ops_cache_hits()
    assert bytes(rv.Proof.new(pa, wa, [], wca, seeds=rule_seeds, ctx=ctx)) == want_a
    # RV_OPS_CACHE=0: nothing kept, nothing found
    monkeypatch.setenv("RV_OPS_CACHE", "0")
    h2 = L.rv_hook_ops_cache_hits()
    assert bytes(rv.Proof.new(pa, wa, [], wca, seeds=rule_seeds, ctx=ctx)) == want_a
    assert bytes(rv.Proof.new(pa, wa, [], wca, seeds=rule_seeds, ctx=ctx)) == want_a
    assert L.rv_hook_ops_cache_hits() == h2
    monkeypatch.delenv("RV_OPS_CACHE")
    assert L.rv_ctx_ops_cache_clear(ctx.handle) == 0
    h3 = L.rv_hook_ops_cache_hits()
    assert bytes(rv.Proof.new(pa, wa, [], wca, seeds=rule_seeds, ctx=ctx)) == want_a and L.rv_hook_ops_cache_hits() == h3
    assert h1 >= h0 + 3
    ctx.close()


def test_ops_cache_is_by_content_colliding_op_lists(rv, rule_seeds):
    """VERDICT r5 weak #2 / ADVICE r5 high: round 5's cache named an op list by an unkeyed 128-bit multiply-fold hash whose mixing
    step is 0 whenever a data word equals the public constant 0xe7037ed1a0b428db -- op 2's `imm` here -- so that lists differing only in
    op 3 shared one compiled circuit and rv_verify_ops accepted a proof of B as a proof of A.  Proof::verify replays the CALLER's ops
    (proof/mod.rs:224-307).  The cache now compares the op arrays themselves: every answer must be the oracle's, in either order, on
    one context, prover and verifier."""
    import oracle_lib

    ctx = rv.Context(0)

    def circuit(assertion):
        ops = [GF2.Input(0), GF2.Input(1), Z64.AddConst(0, 0, 0xE7037ED1A0B428DB)]
        ops.append(GF2.AssertZero(0) if assertion else GF2.AddConst(2, 0, 1))
        w = 3
        rng = np.random.default_rng(77)
        for _ in range(2044):
            a, b = int(rng.integers(0, 2)) if w == 3 else int(rng.integers(3, w)), int(rng.integers(0, 2))
            ops.append(GF2.Mul(w, a, b) if rng.integers(0, 2) else GF2.Add(w, a, b))
            w += 1
        return program(ops), (1, w)

    pa, wca = circuit(True)
    pb, wcb = circuit(False)
    assert wca == wcb and len(pa) == len(pb) == 2048 and (pa != pb).sum() == 1
    wit = [0, 1]
    proof_b = oracle_lib.prove(pb, wit, [], wcb, rule_seeds)
    proof_a = oracle_lib.prove(pa, wit, [], wca, rule_seeds)
    assert oracle_lib.verify(pb, wcb, proof_b) and oracle_lib.verify(pa, wca, proof_a)
    assert not oracle_lib.verify(pa, wca, proof_b) and not oracle_lib.verify(pb, wcb, proof_a)
    for strict in (True, False):
        # B first (cached), then A's list with B's proof: the reference rejects (A has one more reconstruction in its transcript)
        assert rv.Proof(proof_b).verify(pb, wcb, ctx=ctx, strict=strict)
        assert not rv.Proof(proof_b).verify(pa, wca, ctx=ctx, strict=strict)
        assert rv.Proof(proof_a).verify(pa, wca, ctx=ctx, strict=strict)
        assert not rv.Proof(proof_a).verify(pb, wcb, ctx=ctx, strict=strict)
        assert rv.Proof(proof_b).verify(pb.copy(), wcb, ctx=ctx, strict=strict)
    # the prover side: each list proves ITS statement
    assert bytes(rv.Proof.new(pb, wit, [], wcb, seeds=rule_seeds, ctx=ctx)) == proof_b
    assert bytes(rv.Proof.new(pa, wit, [], wca, seeds=rule_seeds, ctx=ctx)) == proof_a
    assert bytes(rv.Proof.new(pb, wit, [], wcb, seeds=rule_seeds, ctx=ctx)) == proof_b
    ctx.close()
