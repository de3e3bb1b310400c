"""The recipe that would pin the oracle to the Rust reference (oracle/ref_pin) cannot RUN here -- no cargo -- but it must not
rot: the seed-injection patch still applies to the reference source when that is present (this container; not the GPU box),
and cases.txt is exactly what export_cases.py writes from the committed golden circuits.  Whole-proof parity with the Rust
binary stays "unpinned" until somebody with a Rust toolchain runs the recipe (oracle/ref_pin/README.md)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "oracle", "ref_pin")
REF = "/root/reference"


def test_cases_txt_is_what_the_golden_circuits_export(tmp_path):
    work = tmp_path / "ref_pin"
    work.mkdir()
    (tmp_path / "tests" / "golden").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "proofs.json"), tmp_path / "tests" / "golden" / "proofs.json")
    # export_cases.py reads ../../tests/golden/proofs.json relative to itself: run a copy two levels below tmp_path
    (tmp_path / "oracle").mkdir()
    dst = tmp_path / "oracle" / "ref_pin"
    dst.mkdir()
    shutil.copy(os.path.join(PIN, "export_cases.py"), dst / "export_cases.py")
    subprocess.check_call([sys.executable, str(dst / "export_cases.py")], stdout=subprocess.DEVNULL)
    assert (dst / "cases.txt").read_bytes() == open(os.path.join(PIN, "cases.txt"), "rb").read()
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))
    text = open(os.path.join(PIN, "cases.txt")).read()
    assert text.count("end\n") == len(meta) and all(("case %s " % n) in text for n in meta)
    # every golden proof the comparison would be made against is committed
    for n in meta:
        if meta[n].get("digest_only"):  # (too large to commit: its length and BLAKE3 digest are)
            assert len(meta[n]["proof_blake3"]) == 64 and meta[n]["proof_len"] > 33160
            continue
        assert os.path.getsize(os.path.join(ROOT, "tests", "golden", "proof_%s.bin" % n)) >= 33160
    # the pin set reaches past one BLAKE3 chunk, the tree and BufferedHasher's 64 KiB flush (crypto/hash.rs:5-6) at whole-proof level
    assert any(sum(c for c, op in meta[n].get("ops_rle", []) if op[1] == 6) >= 65536 for n in meta)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "proof")), reason="the reference source is only present in the build container")
def test_seed_injection_patch_still_applies_to_the_reference():
    if not shutil.which("patch"):
        pytest.skip("no patch(1)")
    r = subprocess.run(["patch", "-p1", "--dry-run", "--batch", "-d", REF, "-i", os.path.join(PIN, "0001-proof-new-with-seeds.patch")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # the harness is written against the entry point the patch adds
    assert "new_with_seeds" in open(os.path.join(PIN, "dump_golden.rs")).read()
    assert "new_with_seeds" in open(os.path.join(PIN, "0001-proof-new-with-seeds.patch")).read()
