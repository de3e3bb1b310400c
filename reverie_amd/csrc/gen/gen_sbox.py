#!/usr/bin/env python3
"""Generates reverie_amd/csrc/aes_sbox.inc: a straight-line boolean circuit for the AES
S-box (Boyar-Peralta style: linear top layer, shared GF(2^4)-tower nonlinear middle,
linear bottom layer; 32 AND + 81 XOR/XNOR), used by the bitsliced AES-128-CTR kernel.

The circuit is VERIFIED here exhaustively (all 256 inputs) against the S-box computed from
its definition (multiplicative inverse in GF(2^8) mod x^8+x^4+x^3+x+1, then the affine
map, FIPS-197 §5.1.1) before anything is written.  Bit convention: U0 = MSB ... U7 = LSB
of the input byte; S0 = MSB ... S7 = LSB of the output byte.
"""
import os
import re

CIRCUIT = """
y14 = U3 ^ U5
y13 = U0 ^ U6
y9 = U0 ^ U3
y8 = U0 ^ U5
t0 = U1 ^ U2
y1 = t0 ^ U7
y4 = y1 ^ U3
y12 = y13 ^ y14
y2 = y1 ^ U0
y5 = y1 ^ U6
y3 = y5 ^ y8
t1 = U4 ^ y12
y15 = t1 ^ U5
y20 = t1 ^ U1
y6 = y15 ^ U7
y10 = y15 ^ t0
y11 = y20 ^ y9
y7 = U7 ^ y11
y17 = y10 ^ y11
y19 = y10 ^ y8
y16 = t0 ^ y11
y21 = y13 ^ y16
y18 = U0 ^ y16
t2 = y12 & y15
t3 = y3 & y6
t4 = t3 ^ t2
t5 = y4 & U7
t6 = t5 ^ t2
t7 = y13 & y16
t8 = y5 & y1
t9 = t8 ^ t7
t10 = y2 & y7
t11 = t10 ^ t7
t12 = y9 & y11
t13 = y14 & y17
t14 = t13 ^ t12
t15 = y8 & y10
t16 = t15 ^ t12
t17 = t4 ^ t14
t18 = t6 ^ t16
t19 = t9 ^ t14
t20 = t11 ^ t16
t21 = t17 ^ y20
t22 = t18 ^ y19
t23 = t19 ^ y21
t24 = t20 ^ y18
t25 = t21 ^ t22
t26 = t21 & t23
t27 = t24 ^ t26
t28 = t25 & t27
t29 = t28 ^ t22
t30 = t23 ^ t24
t31 = t22 ^ t26
t32 = t31 & t30
t33 = t32 ^ t24
t34 = t23 ^ t33
t35 = t27 ^ t33
t36 = t24 & t35
t37 = t36 ^ t34
t38 = t27 ^ t36
t39 = t29 & t38
t40 = t25 ^ t39
t41 = t40 ^ t37
t42 = t29 ^ t33
t43 = t29 ^ t40
t44 = t33 ^ t37
t45 = t42 ^ t41
z0 = t44 & y15
z1 = t37 & y6
z2 = t33 & U7
z3 = t43 & y16
z4 = t40 & y1
z5 = t29 & y7
z6 = t42 & y11
z7 = t45 & y17
z8 = t41 & y10
z9 = t44 & y12
z10 = t37 & y3
z11 = t33 & y4
z12 = t43 & y13
z13 = t40 & y5
z14 = t29 & y2
z15 = t42 & y9
z16 = t45 & y14
z17 = t41 & y8
t46 = z15 ^ z16
t47 = z10 ^ z11
t48 = z5 ^ z13
t49 = z9 ^ z10
t50 = z2 ^ z12
t51 = z2 ^ z5
t52 = z7 ^ z8
t53 = z0 ^ z3
t54 = z6 ^ z7
t55 = z16 ^ z17
t56 = z12 ^ t48
t57 = t50 ^ t53
t58 = z4 ^ t46
t59 = z3 ^ t54
t60 = t46 ^ t57
t61 = z14 ^ t57
t62 = t52 ^ t58
t63 = t49 ^ t58
t64 = z4 ^ t59
t65 = t61 ^ t62
t66 = z1 ^ t63
S0 = t59 ^ t63
S6 = t56 ^ ~t62
S7 = t48 ^ ~t60
t67 = t64 ^ t65
S3 = t53 ^ t66
S4 = t51 ^ t66
S5 = t47 ^ t65
S1 = t64 ^ ~S3
S2 = t55 ^ ~t67
"""


def gf_mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a = ((a << 1) ^ (0x11B if a & 0x80 else 0)) & 0xFF
        b >>= 1
    return r


def sbox_def(x):
    inv = 0
    if x:
        acc, base, e = 1, x, 254
        while e:
            if e & 1:
                acc = gf_mul(acc, base)
            base = gf_mul(base, base)
            e >>= 1
        inv = acc
    s = inv
    rot = inv
    for _ in range(4):
        rot = ((rot << 1) | (rot >> 7)) & 0xFF
        s ^= rot
    return s ^ 0x63


def evaluate(x):
    env = {f"U{i}": (x >> (7 - i)) & 1 for i in range(8)}
    for line in CIRCUIT.strip().splitlines():
        lhs, rhs = [s.strip() for s in line.split("=")]
        rhs = re.sub(r"~(\w+)", r"(1 ^ \1)", rhs)
        env[lhs] = eval(rhs, {}, env) & 1
    return sum(env[f"S{i}"] << (7 - i) for i in range(8))


def parse():
    """-> list of (dst, op, x, y, neg_y) in order"""
    gates = []
    for line in CIRCUIT.strip().splitlines():
        lhs, rhs = [t.strip() for t in line.split("=")]
        m = re.match(r"(\w+) ([\^&]) (~?)(\w+)", rhs)
        gates.append((lhs, m.group(2), m.group(1), m.group(4), bool(m.group(3))))
    return gates


LUT3_COST = float(os.environ.get("SBOX_LUT3_COST", "1.15"))  # a LUT3 in two-input ops (the cover is exact for this weight)


def lut3_synthesis():
    """Greedy packing of the 2-input XOR/AND/XNOR netlist into <=3-input LUTs (gfx950 has
    v_bitop3_b32: any 3-input boolean function in one VALU op).  A gate is inlined into its
    consumer when it has fan-out 1 and the merged node still has <= 3 distinct inputs."""
    nodes = {}  # name -> (inputs tuple, fn(dict)->bit)
    order = []
    for dst, op, x, y, neg in parse():
        if op == "^":
            fn = (lambda x, y, neg: lambda env: env[x] ^ env[y] ^ (1 if neg else 0))(x, y, neg)
        else:
            fn = (lambda x, y: lambda env: env[x] & env[y])(x, y)
        nodes[dst] = ((x, y), fn)
        order.append(dst)
    outputs = {f"S{i}" for i in range(8)}

    def fanout():
        fo = {}
        for n, (ins, _) in nodes.items():
            for i in set(ins):
                fo[i] = fo.get(i, 0) + 1
        return fo

    def compose(fn, wfn, w):
        def f(env):
            e = dict(env)
            e[w] = wfn(env)
            return fn(e)
        return f

    # Cost-aware rule.  Measured on gfx950 (tools/mb/valu_mb.hip, occ_mb.hip): a 3-source VOP3 op
    # (v_bitop3_b32) issues at 4 cycles per wavefront, a VOP2 logic op at ~2.4 with two wavefronts per SIMD,
    # so a LUT3 costs LUT3_COST two-input ops.  A gate is absorbed into ALL its consumers (duplicated when
    # it has several) only when that lowers the total cost; minimising the op count instead (every absorption
    # taken) gave 80 ops / 60 LUT3 = 119 two-input equivalents, worse than the plain 113-gate netlist.
    def cost(ins):
        return LUT3_COST if len(ins) == 3 else 1.0

    changed = True
    while changed:
        changed = False
        best = None
        for w in list(order):
            if w not in nodes or w in outputs:
                continue
            wins, wfn = nodes[w]
            cons = [n for n in order if n in nodes and w in nodes[n][0]]
            if not cons:
                continue
            merged = {}
            ok = True
            for n in cons:
                ins, fn = nodes[n]
                union = tuple(dict.fromkeys([i for i in ins if i != w] + list(wins)))
                if len(union) > 3:
                    ok = False
                    break
                merged[n] = (union, compose(fn, wfn, w))
            if not ok:
                continue
            gain = cost(wins) + sum(cost(nodes[n][0]) for n in cons) - sum(cost(m[0]) for m in merged.values())
            if gain > 1e-9 and (best is None or gain > best[0]):
                best = (gain, w, merged)
        if best:
            _, w, merged = best
            nodes.update(merged)
            del nodes[w]
            changed = True
    final = [n for n in order if n in nodes]
    luts = []
    for n in final:
        ins, fn = nodes[n]
        ins = tuple(ins)
        tt = 0
        for idx in range(1 << len(ins)):
            env = {name: (idx >> (len(ins) - 1 - k)) & 1 for k, name in enumerate(ins)}
            if fn(env):
                tt |= 1 << idx
        luts.append((n, ins, tt))
    return luts


def lut3_mapping_exact():
    """Minimum-cost cover of the netlist by <=3-input LUTs (classic cut-based technology mapping, solved
    exactly as a small 0/1 program with scipy's HiGHS): x[n,c] = node n is implemented as cut c;
    every output is implemented; an implemented cut needs its non-input leaves implemented."""
    import itertools

    import numpy as np
    from scipy.optimize import Bounds, LinearConstraint, milp

    gates = parse()
    fan = {dst: (x, y) for dst, _, x, y, _ in gates}
    fn = {}
    for dst, op, x, y, neg in gates:
        fn[dst] = (op, x, y, neg)
    order = [g[0] for g in gates]
    outputs = [f"S{i}" for i in range(8)]

    def value(n, env):
        if n in env:
            return env[n]
        op, x, y, neg = fn[n]
        a, b = value(x, env), value(y, env)
        v = (a ^ b ^ (1 if neg else 0)) if op == "^" else (a & b)
        env[n] = v
        return v

    cuts = {}  # node -> list of frozenset(leaves)
    for n in order:
        x, y = fan[n]
        cx = [frozenset([x])] + (cuts.get(x, []))
        cy = [frozenset([y])] + (cuts.get(y, []))
        cs = set()
        for a, b in itertools.product(cx, cy):
            u = a | b
            if len(u) <= 3:
                cs.add(u)
        cuts[n] = sorted(cs, key=lambda c: (len(c), sorted(c)))
    var = []  # (node, cut)
    for n in order:
        for c in cuts[n]:
            var.append((n, c))
    idx = {v: i for i, v in enumerate(var)}
    cost = np.array([LUT3_COST if len(c) == 3 else 1.0 for _, c in var])
    rows, lo, hi = [], [], []
    by_node = {n: [idx[(n, c)] for c in cuts[n]] for n in order}
    for o in outputs:  # outputs implemented exactly once
        r = np.zeros(len(var))
        r[by_node[o]] = 1
        rows.append(r); lo.append(1); hi.append(1)
    for n in order:  # at most one implementation per node
        r = np.zeros(len(var))
        r[by_node[n]] = 1
        rows.append(r); lo.append(0); hi.append(1)
    for (n, c), i in idx.items():  # leaves of a chosen cut must exist
        for leaf in c:
            if leaf in fan:
                r = np.zeros(len(var))
                r[by_node[leaf]] = 1
                r[i] -= 1
                rows.append(r); lo.append(0); hi.append(np.inf)
    res = milp(cost, constraints=LinearConstraint(np.array(rows), lo, hi), integrality=np.ones(len(var)), bounds=Bounds(0, 1),
               options={"time_limit": 120})
    assert res.success, res.message
    chosen = [var[i] for i in range(len(var)) if res.x[i] > 0.5]
    luts = []
    pos = {n: k for k, n in enumerate(order)}
    for n, c in sorted(chosen, key=lambda v: pos[v[0]]):
        ins = tuple(sorted(c, key=lambda w: (w in fan, pos.get(w, -1), w)))
        tt = 0
        for a in range(1 << len(ins)):
            env = {name: (a >> (len(ins) - 1 - k)) & 1 for k, name in enumerate(ins)}
            if value(n, env):
                tt |= 1 << a
        luts.append((n, ins, tt))
    return luts


def eval_luts(luts, x):
    env = {f"U{i}": (x >> (7 - i)) & 1 for i in range(8)}
    for n, ins, tt in luts:
        idx = 0
        for name in ins:
            idx = (idx << 1) | env[name]
        env[n] = (tt >> idx) & 1
    return sum(env[f"S{i}"] << (7 - i) for i in range(8))


def main():
    for x in range(256):
        assert evaluate(x) == sbox_def(x), f"S-box circuit wrong at {x:#x}"
    lines = CIRCUIT.strip().splitlines()
    n_and = sum("&" in l for l in lines)
    luts = lut3_mapping_exact()
    for x in range(256):
        assert eval_luts(luts, x) == sbox_def(x), f"LUT3 netlist wrong at {x:#x}"
    n3 = sum(len(ins) == 3 for _, ins, _ in luts)
    out = ["// GENERATED by gen/gen_sbox.py — exhaustively verified against the FIPS-197 S-box definition.",
           f"// {len(lines)}-gate Boyar-Peralta style circuit ({n_and} AND) packed into {len(luts)} ops, {n3} of them",
           "// 3-input v_bitop3_b32 LUTs (truth-table index = (S0<<2)|(S1<<1)|S2). U0/S0 = most significant bit.",
           "// Included inside sbox8(): inputs U0..U7, results S0..S7."]
    for n, ins, tt in luts:
        if len(ins) == 3:
            out.append(f"const uint32_t {n} = __builtin_amdgcn_bitop3_b32({ins[0]}, {ins[1]}, {ins[2]}, 0x{tt:02x});")
        else:
            a_, b_ = ins
            expr = {0x6: f"{a_} ^ {b_}", 0x8: f"{a_} & {b_}", 0x9: f"~({a_} ^ {b_})", 0xe: f"{a_} | {b_}"}.get(tt)
            if expr is None:  # some other 2-input function: a LUT3 that ignores its third operand
                tt3 = sum(((tt >> (i >> 1)) & 1) << i for i in range(8))
                expr = f"__builtin_amdgcn_bitop3_b32({a_}, {b_}, {b_}, 0x{tt3:02x})"
            out.append(f"const uint32_t {n} = {expr};")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ.get("SBOX_OUT", "aes_sbox.inc"))
    open(path, "w").write("\n".join(out) + "\n")
    tab = ", ".join("0x%02x" % sbox_def(x) for x in range(256))
    tpath = os.path.join(os.path.dirname(path), "aes_sbox_table.inc")
    open(tpath, "w").write("// GENERATED by gen/gen_sbox.py from the FIPS-197 S-box definition (used only by the tiny\n"
                           "// key-schedule / seed-expansion kernels; the mask generator uses the bitsliced circuit).\n" + tab + "\n")
    print("verified 256/256;", len(lines), "gates,", n_and, "AND ->", len(luts), "LUT ops (", n3, "three-input ) cost",
          n3 * LUT3_COST + (len(luts) - n3), "two-input equivalents ->", path)


if __name__ == "__main__":
    main()
