#!/usr/bin/env python3
"""tests/golden/proofs.json -> cases.txt (plain text the Rust harness dump_golden.rs parses without extra crates).

    case <name> <z64_wires> <gf2_wires>
    op <domain> <opcode> <dst> <a> <b> <imm>      (include/reverie_amd.h: rv_op, reserved dropped)
    opx <count> <domain> <opcode> <dst> <a> <b> <imm>   the same op `count` times in a row (the large digest-only cases)
    w2 <bits...>            w64 <words...>
    end
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "..", "..", "tests", "golden", "proofs.json")))

with open(os.path.join(HERE, "cases.txt"), "w") as f:
    for name in sorted(META):
        m = META[name]
        f.write(f"case {name} {m['wire_counts'][0]} {m['wire_counts'][1]}\n")
        for dom, opc, _res, dst, a, b, imm in m.get("ops", []):
            f.write(f"op {dom} {opc} {dst} {a} {b} {imm}\n")
        for count, (dom, opc, _res, dst, a, b, imm) in m.get("ops_rle", []):
            f.write(f"opx {count} {dom} {opc} {dst} {a} {b} {imm}\n" if count > 1 else f"op {dom} {opc} {dst} {a} {b} {imm}\n")
        for i in range(0, len(m["wit_gf2"]), 64):
            f.write("w2 " + " ".join(str(int(x)) for x in m["wit_gf2"][i:i + 64]) + "\n")
        if m["wit_z64"]:
            f.write("w64 " + " ".join(str(int(x)) for x in m["wit_z64"]) + "\n")
        f.write("end\n")
print("wrote", os.path.join(HERE, "cases.txt"))
