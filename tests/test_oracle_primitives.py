"""Pins the oracle's primitives against committed known answers generated from OpenSSL
(AES-128-CTR) and the official BLAKE3 C build (tests/golden/gen_golden.py)."""
import ctypes as C
import json
import os

import numpy as np

from conftest import GOLDEN

PRIM = json.load(open(os.path.join(GOLDEN, "primitives.json")))


def _b3(oracle, data, n=32, seek=0, pieces=True):
    h = C.create_string_buffer(8192)
    L = oracle.lib()
    L.rvo_blake3_init(h)
    i, step = 0, 1
    while i < len(data):
        chunk = data[i:i + step] if pieces else data
        L.rvo_blake3_update(h, chunk, C.c_size_t(len(chunk)))
        i += len(chunk)
        step = step * 3 + 1
    out = C.create_string_buffer(n)
    L.rvo_blake3_finalize_xof(h, C.c_uint64(seek), out, C.c_size_t(n))
    return out.raw


def test_aes_fips197(oracle):
    L = oracle.lib()
    ctx = C.create_string_buffer(512)
    L.rvo_aes128_init(ctx, bytes(range(16)))
    out = C.create_string_buffer(16)
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    L.rvo_aes128_encrypt(ctx, pt, out)
    assert out.raw.hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    L.rvo_aes128_encrypt_portable(ctx, pt, out)
    assert out.raw.hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"


def test_aes_ctr_prg(oracle):
    L = oracle.lib()
    for kat in PRIM["aes_ctr"]:
        key = bytes.fromhex(kat["key"])
        prg = C.create_string_buffer(1024)
        L.rvo_prg_init(prg, key)
        buf = C.create_string_buffer(80)
        L.rvo_prg_gen(prg, buf, C.c_size_t(48))
        L.rvo_prg_gen(prg, C.byref(buf, 48), C.c_size_t(32))  # counter carries across calls
        assert buf.raw.hex() == kat["stream"]
        blk = C.create_string_buffer(16)
        L.rvo_prg_block(key, C.c_uint64(3), blk)
        assert blk.raw.hex() == kat["stream"][96:128]
    # zero key, first two blocks (checked against `openssl enc -aes-128-ctr` in SURVEY §8c)
    assert PRIM["aes_ctr"][0]["stream"].startswith("66e94bd4ef8a2c3b884cfa59ca342b2e58e2fccefa7e3061367f1d57a4e7455a")


def test_blake3_kats(oracle):
    assert _b3(oracle, b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert _b3(oracle, b"abc").hex().startswith("6437b3ac")
    for kat in PRIM["blake3"]:
        d = bytes(i % 251 for i in range(kat["len"]))
        assert _b3(oracle, d).hex() == kat["hash"], kat["len"]
        assert _b3(oracle, d, 131, 7).hex() == kat["xof_seek7_len131"], kat["len"]
        assert _b3(oracle, d, pieces=False).hex() == kat["hash"]


def test_expand_seed_and_rule_seeds(oracle, rule_seeds):
    for k, v in PRIM["rep_seed"].items():
        assert rule_seeds[int(k)].tobytes().hex() == v
    for kat in PRIM["expand_seed"]:
        keys = oracle.expand_seed(np.frombuffer(bytes.fromhex(kat["seed"]), np.uint8))
        assert keys.tobytes().hex() == kat["keys"]


def test_challenge(oracle):
    for kat in PRIM["challenge"]:
        omit = oracle.challenge(np.frombuffer(bytes.fromhex(kat["comm"]), np.uint8))
        assert omit.tolist() == kat["omit"]
        assert (omit < 8).sum() == 40
    # SURVEY §8c spot value: RO(comm = 0^32) first draw -> rep 250, omit 0 unless redrawn
    assert PRIM["challenge"][0]["omit"][250] < 8
