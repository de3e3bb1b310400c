"""Gate-level generators that emit Bristol Fashion text for the BASELINE configs 1-3
(64-bit adder, AES-128, SHA-256 compression) — there is no network to fetch the canonical
files, so the circuits are built here and checked in the clear against FIPS-197 / hashlib
(tests/test_bristol.py).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reverie_amd", "csrc", "gen"))


class Netlist:
    def __init__(self, n_inputs: int):
        self.n_in = n_inputs
        self.n = n_inputs
        self.gates = []  # (kind, ins, out)
        self._zero = None
        self._one = None

    def new(self):
        self.n += 1
        return self.n - 1

    def xor(self, a, b):
        o = self.new(); self.gates.append(("XOR", (a, b), o)); return o

    def and_(self, a, b):
        o = self.new(); self.gates.append(("AND", (a, b), o)); return o

    def inv(self, a):
        o = self.new(); self.gates.append(("INV", (a,), o)); return o

    def const(self, bit):
        o = self.new(); self.gates.append(("EQ", (bit,), o)); return o

    def zero(self):
        if self._zero is None:
            self._zero = self.const(0)
        return self._zero

    def one(self):
        if self._one is None:
            self._one = self.const(1)
        return self._one

    def finish(self, outputs, in_groups, out_groups) -> str:
        """copies `outputs` to the last wires (Bristol convention) and renders Fashion text"""
        outs = []
        for w in outputs:
            o = self.new(); self.gates.append(("EQW", (w,), o)); outs.append(o)
        assert outs == list(range(self.n - len(outs), self.n))
        lines = [f"{len(self.gates)} {self.n}", " ".join(map(str, [len(in_groups)] + in_groups)),
                 " ".join(map(str, [len(out_groups)] + out_groups)), ""]
        for kind, ins, o in self.gates:
            lines.append(f"{len(ins)} 1 {' '.join(map(str, ins))} {o} {kind}")
        return "\n".join(lines) + "\n"


# ---------------------------------------------------------------- config 1: 64-bit adder
def adder64():
    nl = Netlist(128)
    a = list(range(64)); b = list(range(64, 128))  # LSB first
    out = []
    carry = None
    for i in range(64):
        if carry is None:
            out.append(nl.xor(a[i], b[i])); carry = nl.and_(a[i], b[i])
        else:
            ac = nl.xor(a[i], carry); bc = nl.xor(b[i], carry)
            out.append(nl.xor(ac, b[i]))
            if i < 63:
                carry = nl.xor(nl.and_(ac, bc), carry)
    return nl.finish(out, [64, 64], [64])


# ---------------------------------------------------------------- config 2: AES-128
def _sbox_netlist():
    import gen_sbox

    return gen_sbox.parse()


def _sbox(nl: Netlist, byte):  # byte: 8 wires, MSB first
    env = {f"U{i}": byte[i] for i in range(8)}
    for dst, op, x, y, neg in _SBOX:
        if op == "&":
            env[dst] = nl.and_(env[x], env[y])
        else:
            w = nl.xor(env[x], env[y])
            env[dst] = nl.inv(w) if neg else w
    return [env[f"S{i}"] for i in range(8)]


_SBOX = None


def aes128():
    """inputs: key (16 bytes, MSB-first bits) then plaintext; outputs: ciphertext bytes"""
    global _SBOX
    _SBOX = _sbox_netlist()
    nl = Netlist(256)
    key = [[8 * i + k for k in range(8)] for i in range(16)]
    st = [[128 + 8 * i + k for k in range(8)] for i in range(16)]
    xb = lambda p, q: [nl.xor(u, v) for u, v in zip(p, q)]  # noqa: E731

    def xtime(b):  # b MSB first: b[0]=bit7
        bit = lambda k: b[7 - k]  # noqa: E731
        y = [None] * 8
        y[0] = bit(7); y[1] = nl.xor(bit(0), bit(7)); y[2] = bit(1); y[3] = nl.xor(bit(2), bit(7))
        y[4] = nl.xor(bit(3), bit(7)); y[5] = bit(4); y[6] = bit(5); y[7] = bit(6)
        return [y[7 - i] for i in range(8)]

    rk = key
    st = [xb(st[i], rk[i]) for i in range(16)]
    rcon = 1
    for rnd in range(1, 11):
        # key schedule
        t = [_sbox(nl, rk[13]), _sbox(nl, rk[14]), _sbox(nl, rk[15]), _sbox(nl, rk[12])]
        t[0] = [nl.inv(t[0][i]) if (rcon >> (7 - i)) & 1 else t[0][i] for i in range(8)]
        rcon = ((rcon << 1) ^ (0x11B if rcon & 0x80 else 0)) & 0xFF
        nk = [None] * 16
        for i in range(4):
            nk[i] = xb(rk[i], t[i])
        for i in range(4, 16):
            nk[i] = xb(rk[i], nk[i - 4])
        rk = nk
        # round
        sb = [_sbox(nl, st[i]) for i in range(16)]
        sr = [sb[4 * ((c + r) & 3) + r] for c in range(4) for r in range(4)]
        if rnd < 10:
            mc = [None] * 16
            for c in range(4):
                a = sr[4 * c:4 * c + 4]
                allx = xb(xb(a[0], a[1]), xb(a[2], a[3]))
                for r in range(4):
                    mc[4 * c + r] = xb(xb(a[r], allx), xtime(xb(a[r], a[(r + 1) & 3])))
            sr = mc
        st = [xb(sr[i], rk[i]) for i in range(16)]
    return nl.finish([w for byte in st for w in byte], [128, 128], [128])


# ---------------------------------------------------------------- config 3: SHA-256 compression
_K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
_IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def sha256_block():
    """inputs: one 512-bit block (bytes in order, MSB-first bits); state = the SHA-256 IV (constants);
    outputs: the 256-bit digest of that single (already padded) block"""
    nl = Netlist(512)
    # words LSB-first lists of 32 wires
    W = [[32 * i + 31 - k for k in range(32)] for i in range(16)]
    cw = lambda v: [nl.one() if (v >> k) & 1 else nl.zero() for k in range(32)]  # noqa: E731
    x3 = lambda a, b, c: [nl.xor(nl.xor(p, q), r) for p, q, r in zip(a, b, c)]  # noqa: E731
    rotr = lambda w, n: [w[(k + n) % 32] for k in range(32)]  # noqa: E731
    shr = lambda w, n: [w[k + n] if k + n < 32 else nl.zero() for k in range(32)]  # noqa: E731

    def add(a, b):
        out = []
        c = None
        for i in range(32):
            if c is None:
                out.append(nl.xor(a[i], b[i])); c = nl.and_(a[i], b[i])
            else:
                ac = nl.xor(a[i], c); bc = nl.xor(b[i], c)
                out.append(nl.xor(ac, b[i]))
                if i < 31:
                    c = nl.xor(nl.and_(ac, bc), c)
        return out

    for t in range(16, 64):
        s0 = x3(rotr(W[t - 15], 7), rotr(W[t - 15], 18), shr(W[t - 15], 3))
        s1 = x3(rotr(W[t - 2], 17), rotr(W[t - 2], 19), shr(W[t - 2], 10))
        W.append(add(add(W[t - 16], s0), add(W[t - 7], s1)))
    H = [cw(v) for v in _IV]
    a, b, c, d, e, f, g, h = H
    for t in range(64):
        S1 = x3(rotr(e, 6), rotr(e, 11), rotr(e, 25))
        ch = [nl.xor(nl.and_(nl.xor(fk, gk), ek), gk) for ek, fk, gk in zip(e, f, g)]
        t1 = add(add(add(h, S1), add(ch, cw(_K[t]))), W[t])
        S0 = x3(rotr(a, 2), rotr(a, 13), rotr(a, 22))
        mj = [nl.xor(nl.and_(nl.xor(ak, bk), nl.xor(ak, ck)), ak) for ak, bk, ck in zip(a, b, c)]
        t2 = add(S0, mj)
        h, g, f, e, d, c, b, a = g, f, e, add(d, t1), c, b, a, add(t1, t2)
    out = [add(x, y) for x, y in zip(H, [a, b, c, d, e, f, g, h])]
    return nl.finish([w[31 - k] for w in out for k in range(32)], [512], [256])


def evaluate(prog, witness):
    """clear evaluation of a GF(2) rv_op program (the reference CLI's `oneshot` does this through
    mcircuit::evaluate_composite_program, main.rs:115-132).  Returns the wire values; raises on a
    failing AssertZero."""
    n = int(max(prog["dst"].max(), prog["a"].max(), prog["b"].max())) + 1
    v = [0] * n
    it = iter(witness)
    for op in prog.tolist():
        dom, opc, _r, d, a, b, imm = op
        assert dom == 0
        if opc == 0: v[d] = next(it)
        elif opc in (2, 4): v[d] = v[a] ^ v[b]
        elif opc in (3, 5): v[d] = v[a] ^ (imm & 1)
        elif opc == 6: v[d] = v[a] & v[b]
        elif opc == 7: v[d] = v[a] & (imm & 1)
        elif opc == 8:
            if v[a]: raise ValueError("AssertZero failed")
        elif opc == 9: v[d] = imm & 1
    return v
