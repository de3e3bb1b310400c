"""Oracle (packed C restatement) vs the committed golden proofs from the independent
per-repetition spec model (tests/golden/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_matches, golden_ops
from reverie_amd.ops import OP_DTYPE, program

META = json.load(open(os.path.join(GOLDEN, "proofs.json")))


def load_case(name):
    m = META[name]
    ops = golden_ops(m)
    prog = program(ops) if ops else np.zeros(0, OP_DTYPE)
    gold = None if m.get("digest_only") else open(os.path.join(GOLDEN, f"proof_{name}.bin"), "rb").read()
    return m, prog, m["wit_gf2"], [int(x) for x in m["wit_z64"]], tuple(m["wire_counts"]), gold


@pytest.mark.parametrize("name", sorted(META))
def test_golden_proof_bytes(oracle, rule_seeds, name):
    m, prog, w2, w64, wc, gold = load_case(name)
    pf = oracle.prove(prog, w2, w64, wc, rule_seeds, threads=4)
    assert len(pf) == m["proof_len"]
    assert golden_matches(oracle, name, m, pf)  # (byte for byte; the large case: its length and BLAKE3 digest)
    assert oracle.verify(prog, wc, pf, threads=4)
    h, st, comm = oracle.commit(prog, w2, w64, wc, rule_seeds, threads=4)
    assert comm.tobytes().hex() == m["comm"] == pf[:32].hex()
    assert [h[i].tobytes().hex() for i in range(8)] + [h[255].tobytes().hex()] == m["h"]
    assert [st[0, k].tobytes().hex() for k in range(4)] == m["streams_rep0"]
    assert oracle.challenge(comm).tolist() == m["omit"]


def test_empty_proof_size():
    assert META["empty"]["proof_len"] == 33160


def test_tamper_rejected(oracle, rule_seeds):
    m, prog, w2, w64, wc, gold = load_case("adder64")
    rng = np.random.default_rng(3)
    rejected = 0
    for pos in rng.integers(0, len(gold), 24):
        bad = bytearray(gold)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            ok = oracle.verify(prog, wc, bytes(bad), threads=4)
        except oracle.OracleError:
            ok = False
        rejected += not ok
    assert rejected >= 20  # a few bytes are don't-care in the reference too (zeroed key, pad bits, F9)
    bad = bytearray(gold)
    bad[0] ^= 0x80
    assert not oracle.verify(prog, wc, bytes(bad))


def test_invalid_witness_is_an_error(oracle, rule_seeds):
    m, prog, w2, w64, wc, gold = load_case("adder64")
    w2 = list(w2)
    w2[5] ^= 1
    with pytest.raises(oracle.OracleError) as e:
        oracle.prove(prog, w2, w64, wc, rule_seeds)
    assert e.value.code == 1


def test_wrong_rep_counts_is_false_not_error(oracle):
    m, prog, w2, w64, wc, gold = load_case("empty")
    # drop the last preprocessing record of the z64 section and fix its count: check_format -> false
    cut = gold[:-48]
    z64_pre_count_off = len(gold) - 216 * 48 - 8
    cut = cut[:z64_pre_count_off] + (215).to_bytes(8, "little") + cut[z64_pre_count_off + 8:]
    assert oracle.verify(prog, wc, cut) is False
    with pytest.raises(oracle.OracleError) as e:
        oracle.verify(prog, wc, gold[:1000])
    assert e.value.code == 4
