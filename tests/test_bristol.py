"""Bristol front end (C-ABI rv_bristol_parse, host only) + the config 1-3 circuits."""
import hashlib
import json
import os

import numpy as np
import pytest

import bristol_gen
from reverie_amd import bristol
from reverie_amd.ops import GF2, OP_ADD, OP_ADDCONST, OP_ASSERTZERO, OP_INPUT, OP_MUL, program


def bits_msb(data: bytes):
    return [(b >> (7 - k)) & 1 for b in data for k in range(8)]


def test_parse_small_fashion_and_old():
    txt = "3 7\n2 2 2\n1 1\n\n2 1 0 1 4 XOR\n2 1 2 3 5 AND\n2 1 4 5 6 XOR\n"
    prog, info = bristol.parse(txt)
    assert (info["n_inputs"], info["n_outputs"], info["n_and"], info["n_xor"]) == (4, 1, 1, 2)
    assert prog["opcode"].tolist() == [OP_INPUT] * 4 + [OP_ADD, OP_MUL, OP_ADD]
    old = "3 7\n2 2 1\n\n2 1 0 1 4 XOR\n2 1 2 3 5 AND\n2 1 4 5 6 XOR\n"
    prog2, info2 = bristol.parse(old, fmt=2)
    assert prog2.tobytes() == prog.tobytes()
    # with expected outputs: AddConst + AssertZero per output wire, temporaries after the circuit's wires
    prog3, info3 = bristol.parse(txt, expected_outputs=[1])
    assert prog3["opcode"].tolist()[-2:] == [OP_ADDCONST, OP_ASSERTZERO]
    assert info3["gf2_wires"] == 8 and info3["wire_counts"] == (0, 8)
    with pytest.raises(ValueError):  # (1^0) ^ (1&1) = 0, expected 1
        bristol_gen.evaluate(prog3, [1, 0, 1, 1])
    bristol_gen.evaluate(bristol.parse(txt, expected_outputs=[0])[0], [1, 0, 1, 1])
    inv = "2 4\n1 2\n1 1\n\n1 1 0 2 INV\n2 1 2 1 3 AND\n"
    p4, _ = bristol.parse(inv)
    assert p4[2]["opcode"] == OP_ADDCONST and p4[2]["imm"] == 1


def test_parse_errors():
    import reverie_amd

    for bad in ("", "1 2\n", "1 3\n1 2\n1 1\n\n2 1 0 1 9 XOR\n", "1 3\n1 2\n1 1\n\n2 1 0 1 2 NAND\n"):
        with pytest.raises(reverie_amd.ReverieError):
            bristol.parse(bad)


def test_wire_ids_beyond_u32_are_unsupported_not_malformed():
    """The reference's wire ids are usize (interpreter/single.rs:14,109-155); rv_op carries u32.  A file that names a wire or a
    count of 2^32 and more is answered RV_E_UNSUPPORTED (8): a stated ceiling, not a silent truncation and not "malformed" (5)."""
    import reverie_amd

    ok = "1 3\n2 1 1\n1 1\n\n2 1 0 1 2 XOR\n"
    bristol.parse(ok)
    for text in (ok.replace("2 1 0 1 2 XOR", "2 1 0 1 4294967296 XOR"), ok.replace("1 3\n", "1 4294967296\n"),
                 ok.replace("2 1 0 1 2 XOR", "2 1 0 99999999999999999999999 2 XOR")):
        with pytest.raises(reverie_amd.ReverieError) as e:
            bristol.parse(text)
        assert e.value.code == 8
    with pytest.raises(reverie_amd.ReverieError) as e:
        bristol.parse(ok.replace("2 1 0 1 2 XOR", "2 1 0 1 4294967295 XOR"))  # fits u32, but no such wire in a 3-wire circuit
    assert e.value.code != 8


def test_adder64_circuit():
    prog, info = bristol.parse(bristol_gen.adder64())
    assert info["n_and"] == 63 and info["n_inputs"] == 128
    a, b = 0x0123456789ABCDEF, 0xFEDCBA9876543210
    wit = [(a >> i) & 1 for i in range(64)] + [(b >> i) & 1 for i in range(64)]
    v = bristol_gen.evaluate(prog, wit)
    out = sum(v[info["n_wires"] - 64 + i] << i for i in range(64))
    assert out == (a + b) & ((1 << 64) - 1)


@pytest.fixture(scope="module")
def aes_text():
    return bristol_gen.aes128()


@pytest.fixture(scope="module")
def sha_text():
    return bristol_gen.sha256_block()


def test_aes128_circuit_fips197(aes_text):
    prog, info = bristol.parse(aes_text)
    assert info["n_and"] == 6400  # 200 S-boxes x 32 AND: the canonical AES-128 AND count
    key = bytes(range(16))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    v = bristol_gen.evaluate(prog, bits_msb(key) + bits_msb(pt))
    ct = v[info["n_wires"] - 128:info["n_wires"]]
    assert ct == bits_msb(bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a"))  # FIPS-197 C.1


def test_sha256_circuit_abc(sha_text):
    prog, info = bristol.parse(sha_text)
    assert 22000 < info["n_and"] < 24000
    block = b"abc" + b"\x80" + bytes(52) + (24).to_bytes(8, "big")
    v = bristol_gen.evaluate(prog, bits_msb(block))
    dig = v[info["n_wires"] - 256:info["n_wires"]]
    assert dig == bits_msb(hashlib.sha256(b"abc").digest())


def test_adder64_oracle_prove_verify(oracle, rule_seeds):
    """config 1: CPU reference prove + verify (plumbing, no GPU)"""
    a, b = 0x0123456789ABCDEF, 0xFEDCBA9876543210
    total = (a + b) & ((1 << 64) - 1)
    prog, info = bristol.parse(bristol_gen.adder64(), expected_outputs=[(total >> i) & 1 for i in range(64)])
    wit = [(a >> i) & 1 for i in range(64)] + [(b >> i) & 1 for i in range(64)]
    pf = oracle.prove(prog, wit, [], info["wire_counts"], rule_seeds)
    assert oracle.verify(prog, info["wire_counts"], pf)
    bad = bytearray(pf); bad[3] ^= 1
    assert not oracle.verify(prog, info["wire_counts"], bytes(bad))
    with pytest.raises(oracle.OracleError):
        oracle.prove(prog, [1 - wit[0]] + wit[1:], [], info["wire_counts"], rule_seeds)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["aes128", "sha256"])
def test_config2_3_gpu_vs_oracle(oracle, rule_seeds, aes_text, sha_text, which):
    """configs 2 and 3: AES-128 / SHA-256 Bristol circuits, full KKW parameters, GPU proof
    bit-exact vs the oracle, cross-verified"""
    import reverie_amd

    if which == "aes128":
        key = bytes(range(16)); pt = bytes.fromhex("00112233445566778899aabbccddeeff")
        wit = bits_msb(key) + bits_msb(pt)
        prog, info = bristol.parse(aes_text, expected_outputs=bits_msb(bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")))
    else:
        block = b"abc" + b"\x80" + bytes(52) + (24).to_bytes(8, "big")
        wit = bits_msb(block)
        prog, info = bristol.parse(sha_text, expected_outputs=bits_msb(hashlib.sha256(b"abc").digest()))
    wc = info["wire_counts"]
    want = oracle.prove(prog, wit, [], wc, rule_seeds)
    c = reverie_amd.Circuit(prog, wc)
    proof = reverie_amd.Proof.new(c, wit, [], seeds=rule_seeds)
    assert bytes(proof) == want
    assert proof.verify(c) and oracle.verify(prog, wc, bytes(proof))
    wrong = list(wit); wrong[5] ^= 1
    with pytest.raises(reverie_amd.ReverieError) as e:
        reverie_amd.Proof.new(c, wrong, [], seeds=rule_seeds)
    assert e.value.code == 1
