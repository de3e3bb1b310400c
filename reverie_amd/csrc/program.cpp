// Program files of the reference CLI: bincode 1.3 (fixint, little endian) of
// Vec<mcircuit::CombineOperation>, read at /root/reference/src/main.rs:66 (prove), :98 (verify) and
// :122 (oneshot) and consumed by Proof::new / Proof::verify as the gate stream.  The enum itself lives in
// the un-vendored `mcircuit` crate (Cargo.toml:35); the variant order used here is the one SURVEY
// Appendix A.7 recalls from the public source and CANNOT be checked in this container, so callers must
// ask for this format explicitly (no auto-detection) — "parity unpinned" for this file format only.
//
//   Vec<T>            u64 length, then the items
//   CombineOperation  u32 variant: 0 GF2(Operation<bool>) 1 Z64(Operation<u64>) 2 B2A(usize, usize)
//                     3 SizeHint(usize, usize)
//   Operation<T>      u32 variant: 0 Input(dst) 1 Random(dst) 2 Add(dst,a,b) 3 AddConst(dst,a,T)
//                     4 Sub(dst,a,b) 5 SubConst(dst,a,T) 6 Mul(dst,a,b) 7 MulConst(dst,a,T)
//                     8 AssertZero(a) 9 Const(dst,T)
//   usize -> u64, bool -> one byte (0 / 1), u64 -> 8 bytes
// The rv_op opcodes follow the same order (include/reverie_amd.h), so a record maps field by field.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/reverie_amd.h"

namespace {

struct Reader {
    const uint8_t* p;
    size_t n, o = 0;
    bool ok = true;
    uint64_t u(int bytes) {
        if (!ok || n - o < (size_t)bytes) {
            ok = false;
            return 0;
        }
        uint64_t v = 0;
        for (int i = 0; i < bytes; i++) v |= (uint64_t)p[o + i] << (8 * i);
        o += bytes;
        return v;
    }
};

void put(std::vector<uint8_t>& out, uint64_t v, int bytes) {
    for (int i = 0; i < bytes; i++) out.push_back((uint8_t)(v >> (8 * i)));
}

}  // namespace

extern "C" int rv_program_from_bincode(const uint8_t* data, size_t len, rv_op** ops, size_t* n_ops) {
    if (!data || !ops || !n_ops) return RV_E_ARG;
    *ops = nullptr;
    *n_ops = 0;
    try {
        Reader r{data, len};
        const uint64_t count = r.u(8);
        // the shortest record (GF2 Input) takes 4 + 4 + 8 bytes: a count beyond that is not a program file
        if (!r.ok || count > (len - 8) / 16) return RV_E_BAD_OP;
        std::vector<rv_op> out;
        out.reserve((size_t)count);
        for (uint64_t i = 0; i < count; i++) {
            rv_op g;
            memset(&g, 0, sizeof g);
            const uint64_t dom = r.u(4);
            uint64_t dst = 0, a = 0, b = 0, imm = 0;
            if (dom == RV_DOM_GF2 || dom == RV_DOM_Z64) {
                const int tb = dom == RV_DOM_GF2 ? 1 : 8;  // bool or u64 immediate
                const uint64_t opc = r.u(4);
                switch (opc) {
                    case RV_OP_INPUT:
                    case RV_OP_RANDOM: dst = r.u(8); break;
                    case RV_OP_ADD:
                    case RV_OP_SUB:
                    case RV_OP_MUL: dst = r.u(8), a = r.u(8), b = r.u(8); break;
                    case RV_OP_ADDCONST:
                    case RV_OP_SUBCONST:
                    case RV_OP_MULCONST: dst = r.u(8), a = r.u(8), imm = r.u(tb); break;
                    case RV_OP_ASSERTZERO: a = r.u(8); break;
                    case RV_OP_CONST: dst = r.u(8), imm = r.u(tb); break;
                    default: return RV_E_BAD_OP;
                }
                if (tb == 1 && imm > 1) return RV_E_BAD_OP;  // bincode rejects any other bool encoding
                g.opcode = (uint8_t)opc;
            } else if (dom == RV_DOM_B2A) {
                dst = r.u(8), a = r.u(8);
            } else if (dom == RV_DOM_SIZEHINT) {
                a = r.u(8), b = r.u(8);
            } else {
                return RV_E_BAD_OP;
            }
            if (!r.ok) return RV_E_BAD_OP;  // truncated
            if (dst > 0xFFFFFFFFull || a > 0xFFFFFFFFull || b > 0xFFFFFFFFull) return RV_E_UNSUPPORTED;  // rv_op wires are u32
            g.domain = (uint8_t)dom;
            g.dst = (uint32_t)dst;
            g.a = (uint32_t)a;
            g.b = (uint32_t)b;
            g.imm = imm;
            out.push_back(g);
        }
        // like bincode::deserialize_from on a reader, bytes after the vector are not looked at
        rv_op* res = (rv_op*)malloc(sizeof(rv_op) * (out.empty() ? 1 : out.size()));
        if (!res) return RV_E_NOMEM;
        if (!out.empty()) memcpy(res, out.data(), sizeof(rv_op) * out.size());
        *ops = res;
        *n_ops = out.size();
        return RV_OK;
    } catch (...) {
        return RV_E_NOMEM;
    }
}

extern "C" int rv_program_to_bincode(const rv_op* ops, size_t n_ops, uint8_t** data, size_t* len) {
    if ((!ops && n_ops) || !data || !len) return RV_E_ARG;
    *data = nullptr;
    *len = 0;
    try {
        std::vector<uint8_t> out;
        out.reserve(8 + n_ops * 32);
        put(out, n_ops, 8);
        for (size_t i = 0; i < n_ops; i++) {
            const rv_op& g = ops[i];
            if (g.reserved) return RV_E_BAD_OP;
            put(out, g.domain, 4);
            if (g.domain == RV_DOM_GF2 || g.domain == RV_DOM_Z64) {
                const int tb = g.domain == RV_DOM_GF2 ? 1 : 8;
                const uint64_t imm = tb == 1 ? (g.imm & 1) : g.imm;
                put(out, g.opcode, 4);
                switch (g.opcode) {
                    case RV_OP_INPUT:
                    case RV_OP_RANDOM: put(out, g.dst, 8); break;
                    case RV_OP_ADD:
                    case RV_OP_SUB:
                    case RV_OP_MUL: put(out, g.dst, 8), put(out, g.a, 8), put(out, g.b, 8); break;
                    case RV_OP_ADDCONST:
                    case RV_OP_SUBCONST:
                    case RV_OP_MULCONST: put(out, g.dst, 8), put(out, g.a, 8), put(out, imm, tb); break;
                    case RV_OP_ASSERTZERO: put(out, g.a, 8); break;
                    case RV_OP_CONST: put(out, g.dst, 8), put(out, imm, tb); break;
                    default: return RV_E_BAD_OP;
                }
            } else if (g.domain == RV_DOM_B2A) {
                put(out, g.dst, 8), put(out, g.a, 8);
            } else if (g.domain == RV_DOM_SIZEHINT) {
                put(out, g.a, 8), put(out, g.b, 8);
            } else {
                return RV_E_BAD_OP;
            }
        }
        uint8_t* res = (uint8_t*)malloc(out.empty() ? 1 : out.size());
        if (!res) return RV_E_NOMEM;
        memcpy(res, out.data(), out.size());
        *data = res;
        *len = out.size();
        return RV_OK;
    } catch (...) {
        return RV_E_NOMEM;
    }
}
