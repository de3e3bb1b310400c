// Reference-pin harness (TEST INFRASTRUCTURE, see README.md in this directory).
//
// Drop this file into the reference crate as `tests/dump_golden.rs` AFTER applying
// `0001-proof-new-with-seeds.patch`, then run (nightly toolchain, like the crate itself):
//
//     RV_PIN_CASES=/path/to/cases.txt RV_PIN_OUT=/path/to/out cargo test --release --test dump_golden -- --nocapture
//
// For every case of cases.txt it builds the gate list, proves it with the deterministic seeds
// seed[r] = BLAKE3("rv-seed" || LE32(r))[0..16], checks that the reference verifies its own proof, and writes
// bincode(Proof) to $RV_PIN_OUT/proof_<name>.bin -- the files `compare.py` sets against tests/golden/proof_*.bin.
use std::convert::TryInto;
use std::fs;
use std::sync::Arc;

use reverie::proof::Proof;
use reverie::{CombineOperation, Operation};

fn rep_seed(r: u32) -> [u8; 16] {
    let mut h = blake3::Hasher::new();
    h.update(b"rv-seed");
    h.update(&r.to_le_bytes());
    h.finalize().as_bytes()[..16].try_into().unwrap()
}

fn gf2_op(opcode: u64, dst: usize, a: usize, b: usize, imm: u64) -> Operation<bool> {
    let c = imm & 1 == 1;
    match opcode {
        0 => Operation::Input(dst),
        1 => Operation::Random(dst),
        2 => Operation::Add(dst, a, b),
        3 => Operation::AddConst(dst, a, c),
        4 => Operation::Sub(dst, a, b),
        5 => Operation::SubConst(dst, a, c),
        6 => Operation::Mul(dst, a, b),
        7 => Operation::MulConst(dst, a, c),
        8 => Operation::AssertZero(a),
        9 => Operation::Const(dst, c),
        _ => panic!("bad gf2 opcode {}", opcode),
    }
}

fn z64_op(opcode: u64, dst: usize, a: usize, b: usize, imm: u64) -> Operation<u64> {
    match opcode {
        0 => Operation::Input(dst),
        1 => Operation::Random(dst),
        2 => Operation::Add(dst, a, b),
        3 => Operation::AddConst(dst, a, imm),
        4 => Operation::Sub(dst, a, b),
        5 => Operation::SubConst(dst, a, imm),
        6 => Operation::Mul(dst, a, b),
        7 => Operation::MulConst(dst, a, imm),
        8 => Operation::AssertZero(a),
        9 => Operation::Const(dst, imm),
        _ => panic!("bad z64 opcode {}", opcode),
    }
}

#[test]
fn dump_golden() {
    let cases = std::env::var("RV_PIN_CASES").expect("RV_PIN_CASES = path of cases.txt");
    let out_dir = std::env::var("RV_PIN_OUT").expect("RV_PIN_OUT = output directory");
    fs::create_dir_all(&out_dir).unwrap();
    let mut seeds = [[0u8; 16]; 256];
    for (r, s) in seeds.iter_mut().enumerate() {
        *s = rep_seed(r as u32);
    }
    let text = fs::read_to_string(cases).unwrap();
    let mut name = String::new();
    let (mut z64_wires, mut gf2_wires) = (0usize, 0usize);
    let mut ops: Vec<CombineOperation> = vec![];
    let mut w2: Vec<bool> = vec![];
    let mut w64: Vec<u64> = vec![];
    for line in text.lines() {
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.is_empty() {
            continue;
        }
        match f[0] {
            "case" => {
                name = f[1].to_string();
                z64_wires = f[2].parse().unwrap();
                gf2_wires = f[3].parse().unwrap();
                ops.clear();
                w2.clear();
                w64.clear();
            }
            // "op d o dst a b imm" = one op; "opx n d o dst a b imm" = the same op n times in a row
            "op" | "opx" => {
                let skip = if f[0] == "opx" { 2 } else { 1 };
                let count: usize = if f[0] == "opx" { f[1].parse().unwrap() } else { 1 };
                let v: Vec<u64> = f[skip..].iter().map(|x| x.parse().unwrap()).collect();
                let (dom, opc, dst, a, b, imm) = (v[0], v[1], v[2] as usize, v[3] as usize, v[4] as usize, v[5]);
                for _ in 0..count {
                    ops.push(match dom {
                        0 => CombineOperation::GF2(gf2_op(opc, dst, a, b, imm)),
                        1 => CombineOperation::Z64(z64_op(opc, dst, a, b, imm)),
                        2 => CombineOperation::B2A(dst, a),
                        3 => CombineOperation::SizeHint(a, b),
                        _ => panic!("bad domain {}", dom),
                    });
                }
            }
            "w2" => w2.extend(f[1..].iter().map(|x| *x == "1")),
            "w64" => w64.extend(f[1..].iter().map(|x| x.parse::<u64>().unwrap())),
            "end" => {
                let circuit = Arc::new(ops.clone());
                let proof = Proof::new_with_seeds(
                    circuit.clone(),
                    Arc::new(w2.clone()),
                    Arc::new(w64.clone()),
                    (z64_wires, gf2_wires),
                    Some(&seeds),
                );
                assert!(proof.verify(circuit, (z64_wires, gf2_wires)), "{}: the reference rejects its own proof", name);
                let bytes = bincode::serialize(&proof).unwrap();
                let path = format!("{}/proof_{}.bin", out_dir, name);
                fs::write(&path, &bytes).unwrap();
                // ... and its BLAKE3 digest next to it (compare.py sets the large cases against the committed digest)
                fs::write(format!("{}/proof_{}.b3", out_dir, name), blake3::hash(&bytes).to_hex().as_str()).unwrap();
                println!("{}: {} ops -> {} bytes -> {}", name, ops.len(), bytes.len(), path);
            }
            other => panic!("cases.txt: unknown record {}", other),
        }
    }
}
