import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.build()
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def rule_seeds(oracle):
    """seed[r] = BLAKE3("rv-seed" || LE32(r))[0..16]  (SURVEY §8d / tests/golden/gen_golden.py)"""
    import ctypes as C

    out = np.zeros((256, 16), np.uint8)
    buf = C.create_string_buffer(32)
    for r in range(256):
        d = b"rv-seed" + struct.pack("<I", r)
        oracle.lib().rvo_blake3_hash(d, C.c_size_t(len(d)), buf)
        out[r] = np.frombuffer(buf.raw[:16], np.uint8)
    return out


GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_ops(m):
    """op tuples of a case of tests/golden/proofs.json (plain list, or run-length encoded for the large digest-only cases)"""
    if "ops" in m:
        return [tuple(o) for o in m["ops"]]
    out = []
    for count, op in m["ops_rle"]:
        out += [tuple(op)] * count
    return out


def golden_matches(oracle_mod, name, m, proof: bytes) -> bool:
    """proof == the committed golden proof (byte for byte), or -- digest-only cases -- has its length and BLAKE3 digest"""
    import ctypes as C

    if not m.get("digest_only"):
        return proof == open(os.path.join(GOLDEN, f"proof_{name}.bin"), "rb").read()
    buf = C.create_string_buffer(32)
    oracle_mod.lib().rvo_blake3_hash(proof, C.c_size_t(len(proof)), buf)
    return len(proof) == m["proof_len"] and buf.raw.hex() == m["proof_blake3"]
