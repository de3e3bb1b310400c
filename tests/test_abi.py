"""CPU-side checks of the product library: it loads, exports every symbol the header
declares, refuses to compute without a GPU, and its host-only Fiat-Shamir functions agree
with the golden vectors.  (No GPU compute here.)"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def L():
    from reverie_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.lib()


def test_header_symbols_exported(L):
    from reverie_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "reverie_amd.h")).read()
    declared = set(re.findall(r"\b(rv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.rv_abi_version() == 8
    assert C.sizeof(_lib.ShardParts) == 4 * 8 + 4 * 8 + 8


def test_op_layout_matches_header():
    from reverie_amd.ops import OP_DTYPE

    assert OP_DTYPE.itemsize == 24
    assert [OP_DTYPE.fields[n][1] for n in ("domain", "opcode", "reserved", "dst", "a", "b", "imm")] == [0, 1, 2, 4, 8, 12, 16]


def test_host_fiat_shamir_golden(L):
    import reverie_amd

    prim = json.load(open(os.path.join(GOLDEN, "primitives.json")))
    for kat in prim["challenge"]:
        assert reverie_amd.challenge(bytes.fromhex(kat["comm"])).tolist() == kat["omit"]
    # combine_hashes = BLAKE3 over 8 KiB: compare with the i%251 KAT of the same length
    data = np.frombuffer(bytes(i % 251 for i in range(8192)), np.uint8)
    want = [k["hash"] for k in prim["blake3"] if k["len"] == 8192][0]
    assert reverie_amd.combine_digests(data).hex() == want


def test_host_fiat_shamir_vs_oracle(L, oracle):
    import reverie_amd

    rng = np.random.default_rng(11)
    for _ in range(5):
        h = rng.integers(0, 256, (256, 32), dtype=np.uint8)
        comm = reverie_amd.combine_digests(h)
        buf = C.create_string_buffer(32)
        oracle.lib().rvo_blake3_hash(h.tobytes(), C.c_size_t(8192), buf)
        assert comm == buf.raw
        assert (reverie_amd.challenge(comm) == oracle.challenge(np.frombuffer(comm, np.uint8))).all()


def test_no_cpu_fallback(L):
    """Without a GPU the product must fail loudly (never route through the oracle)."""
    import torch

    import reverie_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(reverie_amd.ReverieError) as e:
        reverie_amd.Context(0)
    assert e.value.code == 7


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "reverie_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in src and "librv_oracle" not in src and "rvo_" not in src, f
