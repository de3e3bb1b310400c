#!/usr/bin/env python3
"""Sets the proofs the patched Rust reference dumped (dump_golden.rs -> $RV_PIN_OUT/proof_<name>.bin) against the
committed golden vectors tests/golden/proof_<name>.bin (which the CPU oracle and the HIP path reproduce byte for byte,
tests/test_oracle_golden.py / tests/test_gpu_parity.py).  Equal files = the whole parity chain
Rust reference == golden == oracle == HIP is pinned; a difference is located down to the proof section.

    python oracle/ref_pin/compare.py /path/to/RV_PIN_OUT
"""
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "..", "tests", "golden")


def sections(b: bytes):
    """byte ranges of a bincode(Proof): comm, then per domain the online records and the preprocessing records"""
    out = [("comm", 0, 32)]
    o = 32
    for dom in ("gf2", "z64"):
        n = struct.unpack_from("<Q", b, o)[0]
        o += 8
        for i in range(n):
            s = o
            o += 1 + 128
            for _ in range(3):
                ln = struct.unpack_from("<Q", b, o)[0]
                o += 8 + ln
            out.append((f"{dom}.online[{i}]", s, o))
        n = struct.unpack_from("<Q", b, o)[0]
        o += 8
        out.append((f"{dom}.preprocessing[0..{n})", o, o + 48 * n))
        o += 48 * n
    return out


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    meta = json.load(open(os.path.join(GOLD, "proofs.json")))
    bad = 0
    for name in sorted(meta):
        ref_path = os.path.join(sys.argv[1], f"proof_{name}.bin")
        if not os.path.exists(ref_path):
            print(f"{name}: MISSING {ref_path}")
            bad += 1
            continue
        ref = open(ref_path, "rb").read()
        if meta[name].get("digest_only"):
            # too large to commit: length and BLAKE3 digest (dump_golden.rs writes the reference proof's digest to proof_<name>.b3)
            b3_path = os.path.join(sys.argv[1], f"proof_{name}.b3")
            got = open(b3_path).read().strip() if os.path.exists(b3_path) else "?"
            if len(ref) == meta[name]["proof_len"] and got == meta[name]["proof_blake3"]:
                print(f"{name}: identical ({len(ref)} bytes, BLAKE3 {got[:16]}...)")
            else:
                bad += 1
                print(f"{name}: DIFFERS: reference {len(ref)} bytes, BLAKE3 {got}; golden {meta[name]['proof_len']} bytes, BLAKE3 {meta[name]['proof_blake3']}"
                      f"; comm equal: {ref[:32].hex() == meta[name]['comm']}")
            continue
        gold = open(os.path.join(GOLD, f"proof_{name}.bin"), "rb").read()
        if ref == gold:
            print(f"{name}: identical ({len(gold)} bytes)")
            continue
        bad += 1
        if len(ref) != len(gold):
            print(f"{name}: DIFFERENT LENGTH reference {len(ref)} vs golden {len(gold)}")
        first = next((i for i in range(min(len(ref), len(gold))) if ref[i] != gold[i]), min(len(ref), len(gold)))
        where = "?"
        try:
            for label, a, b in sections(gold):
                if a <= first < b:
                    where = f"{label} + {first - a}"
        except struct.error:
            pass
        print(f"{name}: DIFFERS at byte {first} ({where}); comm equal: {ref[:32] == gold[:32]}")
    print("PINNED: the Rust reference reproduces every golden proof" if not bad else f"{bad} case(s) differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
