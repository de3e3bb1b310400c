// see compile.h
#include "compile.h"

#include <algorithm>
#include <limits>

namespace rv {

namespace {

struct Builder {
    Compiled& out;
    // GF(2)
    std::vector<Gate> gates;      // program order
    std::vector<uint32_t> level;  // per gate
    std::vector<int32_t> ssa_level;
    std::vector<uint32_t> ssa_row;  // share row of each SSA wire: PRG mask index, or COMP | computed-row index
    uint32_t n_comp = 1;            // computed rows; 0 = the all-zero row
    std::vector<uint32_t> cur;      // gf2 wire index -> current SSA id
    // Z64
    std::vector<Gate64> gates64;
    std::vector<uint32_t> level64;
    std::vector<int32_t> ssa_level64;
    std::vector<uint32_t> cur64;
    uint32_t max_level = 0;
    bool any = false;

    static constexpr uint32_t COMP = 0x80000000u;
    explicit Builder(Compiled& o) : out(o) {
        ssa_level.push_back(-1);
        ssa_row.push_back(COMP | 0);
        ssa_level64.push_back(-1);
    }

    uint32_t new_ssa(int32_t lvl, uint32_t row) {
        ssa_level.push_back(lvl);
        ssa_row.push_back(row);
        return (uint32_t)(ssa_level.size() - 1);
    }
    uint32_t new_ssa64(int32_t lvl) {
        ssa_level64.push_back(lvl);
        return (uint32_t)(ssa_level64.size() - 1);
    }
    void note(uint32_t lvl) {
        if (lvl > max_level) max_level = lvl;
        any = true;
    }
    void emit(const Gate& g, uint32_t lvl) {
        gates.push_back(g);
        level.push_back(lvl);
        note(lvl);
    }
    void emit64(const Gate64& g, uint32_t lvl) {
        gates64.push_back(g);
        level64.push_back(lvl);
        note(lvl);
    }

    // ---- GF(2) primitives on SSA ids (used by plain ops and by the B2A expansion) ----
    uint32_t g_xor(uint32_t a, uint32_t b) {
        Gate g{};
        g.op = G_XOR;
        g.a = a;
        g.b = b;
        g.am = ssa_row[a];
        g.bm = ssa_row[b];
        g.dm = COMP | n_comp++;
        const int32_t lvl = std::max(ssa_level[a], ssa_level[b]) + 1;
        g.dst = new_ssa(lvl, g.dm);
        emit(g, (uint32_t)lvl);
        out.info.gf2_linear++;
        return g.dst;
    }
    uint32_t g_mul(uint32_t a, uint32_t b) {  // interpreter/single.rs:25-69
        Gate g{};
        g.op = G_MUL;
        g.a = a;
        g.b = b;
        g.am = ssa_row[a];
        g.bm = ssa_row[b];
        g.m = (uint32_t)out.n_masks;
        out.n_masks += 2;
        g.eo = (uint32_t)out.n_on++;
        g.ep = (uint32_t)out.n_pre++;
        g.x = (uint32_t)out.n_rec++;
        out.rec_rows.push_back(g.eo);
        const int32_t lvl = std::max(ssa_level[a], ssa_level[b]) + 1;
        g.dst = new_ssa(lvl, g.m + 1);  // the output's mask IS the fresh mask lambda_new
        emit(g, (uint32_t)lvl);
        out.info.gf2_muls++;
        return g.dst;
    }
    uint32_t g_random() {
        Gate g{};
        g.op = G_RANDOM;
        g.m = (uint32_t)out.n_masks++;
        g.dst = new_ssa(0, g.m);
        emit(g, 0);
        out.info.gf2_linear++;
        return g.dst;
    }
};

template <class T>
void sort_by_level(const std::vector<T>& in, const std::vector<uint32_t>& lvl, uint32_t n_levels, std::vector<T>& out,
                   std::vector<uint32_t>& start) {
    start.assign(n_levels + 1, 0);
    for (uint32_t l : lvl) start[l + 1]++;
    for (uint32_t l = 0; l < n_levels; l++) start[l + 1] += start[l];
    out.resize(in.size());
    std::vector<uint32_t> pos(start.begin(), start.end());
    for (size_t i = 0; i < in.size(); i++) out[pos[lvl[i]]++] = in[i];
}

}  // namespace

int compile_ops(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out) {
    out = Compiled();
    Builder b(out);
    b.cur.assign(gf2_wires, 0);
    b.cur64.assign(z64_wires, 0);
    rv_circuit_info& info = out.info;
    info.n_ops = n_ops;
    const uint64_t LIM = std::numeric_limits<uint32_t>::max() - 512;

    for (size_t i = 0; i < n_ops; i++) {
        const rv_op& op = ops[i];
        if (op.reserved != 0) return RV_E_BAD_OP;
        switch (op.domain) {
        case RV_DOM_SIZEHINT:  // interpreter/combine.rs:122-129
            if (op.b > b.cur.size()) b.cur.resize(op.b, 0);
            if (op.a > b.cur64.size()) b.cur64.resize(op.a, 0);
            break;
        case RV_DOM_GF2: {
            const size_t nw = b.cur.size();
            const uint32_t cbit = (uint32_t)(op.imm & 1);
            Gate g{};
            switch (op.opcode) {
            case RV_OP_INPUT:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G_INPUT;
                g.m = (uint32_t)out.n_masks++;
                g.eo = (uint32_t)out.n_on++;
                g.x = (uint32_t)out.n_in++;
                out.in_rows.push_back(g.eo);
                g.dst = b.new_ssa(0, g.m);
                b.cur[op.dst] = g.dst;
                b.emit(g, 0);
                info.gf2_inputs++;
                break;
            case RV_OP_RANDOM:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_random();
                break;
            case RV_OP_CONST:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G_CONST;
                g.x = cbit;
                g.dst = b.new_ssa(0, Builder::COMP | 0);
                b.cur[op.dst] = g.dst;
                b.emit(g, 0);
                info.gf2_linear++;
                break;
            case RV_OP_ADD:
            case RV_OP_SUB:  // gf2/share.rs:220-238: Add and Sub are both XOR
                if (op.dst >= nw || op.a >= nw || op.b >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_xor(b.cur[op.a], b.cur[op.b]);
                break;
            case RV_OP_MUL:
                if (op.dst >= nw || op.a >= nw || op.b >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_mul(b.cur[op.a], b.cur[op.b]);
                break;
            case RV_OP_ADDCONST:
            case RV_OP_SUBCONST:
            case RV_OP_MULCONST: {
                if (op.dst >= nw || op.a >= nw) return RV_E_WIRE_OOB;
                g.a = b.cur[op.a];
                g.op = (op.opcode == RV_OP_MULCONST) ? G_ANDC : G_XORC;
                g.x = cbit;
                g.am = b.ssa_row[g.a];
                const int32_t lvl = b.ssa_level[g.a] + 1;
                // the mask is unchanged (AddConst, MulConst 1) or zero (MulConst 0): alias, no copy
                g.dst = b.new_ssa(lvl, (g.op == G_ANDC && !cbit) ? (Builder::COMP | 0) : g.am);
                b.cur[op.dst] = g.dst;
                b.emit(g, (uint32_t)lvl);
                info.gf2_linear++;
                break;
            }
            case RV_OP_ASSERTZERO: {
                if (op.a >= nw) return RV_E_WIRE_OOB;
                g.op = G_ASSERT;
                g.a = b.cur[op.a];
                g.am = b.ssa_row[g.a];
                g.eo = (uint32_t)out.n_on++;
                g.x = (uint32_t)out.n_rec++;
                out.rec_rows.push_back(g.eo);
                const int32_t lvl = b.ssa_level[g.a] + 1;
                b.emit(g, (uint32_t)lvl);
                info.gf2_asserts++;
                break;
            }
            default:
                return RV_E_BAD_OP;
            }
            break;
        }
        case RV_DOM_Z64: {
            const size_t nw = b.cur64.size();
            Gate64 g{};
            g.imm = op.imm;
            switch (op.opcode) {
            case RV_OP_INPUT:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G64_INPUT;
                g.m = (uint32_t)out.n_masks64++;
                g.eo = out.on_words64;
                out.in_offs64.push_back(g.eo);
                out.on_words64 += 1;  // z64/recon.rs:131-137: 8 bytes per rep
                g.x = (uint32_t)out.n_in64++;
                g.dst = b.new_ssa64(0);
                b.cur64[op.dst] = g.dst;
                b.emit64(g, 0);
                info.z64_inputs++;
                break;
            case RV_OP_RANDOM:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G64_RANDOM;
                g.m = (uint32_t)out.n_masks64++;
                g.dst = b.new_ssa64(0);
                b.cur64[op.dst] = g.dst;
                b.emit64(g, 0);
                info.z64_linear++;
                break;
            case RV_OP_CONST:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G64_CONST;
                g.dst = b.new_ssa64(0);
                b.cur64[op.dst] = g.dst;
                b.emit64(g, 0);
                info.z64_linear++;
                break;
            case RV_OP_ADD:
            case RV_OP_SUB:
            case RV_OP_MUL: {
                if (op.dst >= nw || op.a >= nw || op.b >= nw) return RV_E_WIRE_OOB;
                g.a = b.cur64[op.a];
                g.b = b.cur64[op.b];
                const int32_t lvl = std::max(b.ssa_level64[g.a], b.ssa_level64[g.b]) + 1;
                if (op.opcode == RV_OP_MUL) {
                    g.op = G64_MUL;
                    g.m = (uint32_t)out.n_masks64;
                    out.n_masks64 += 2;
                    g.ep = out.pre_words64;
                    out.pre_words64 += 1;
                    g.xc = (uint32_t)out.n_corr64++;
                    g.eo = out.on_words64;
                    out.rec_offs64.push_back(g.eo);
                    out.on_words64 += 8;  // z64/share.rs:100-108: 8 players x 8 bytes per rep
                    g.x = (uint32_t)out.n_rec64++;
                    info.z64_muls++;
                } else {
                    g.op = op.opcode == RV_OP_ADD ? G64_ADD : G64_SUB;
                    info.z64_linear++;
                }
                g.dst = b.new_ssa64(lvl);
                b.cur64[op.dst] = g.dst;
                b.emit64(g, (uint32_t)lvl);
                break;
            }
            case RV_OP_ADDCONST:
            case RV_OP_SUBCONST:
            case RV_OP_MULCONST: {
                if (op.dst >= nw || op.a >= nw) return RV_E_WIRE_OOB;
                g.a = b.cur64[op.a];
                g.op = op.opcode == RV_OP_ADDCONST ? G64_ADDC : op.opcode == RV_OP_SUBCONST ? G64_SUBC : G64_MULC;
                const int32_t lvl = b.ssa_level64[g.a] + 1;
                g.dst = b.new_ssa64(lvl);
                b.cur64[op.dst] = g.dst;
                b.emit64(g, (uint32_t)lvl);
                info.z64_linear++;
                break;
            }
            case RV_OP_ASSERTZERO: {
                if (op.a >= nw) return RV_E_WIRE_OOB;
                g.op = G64_ASSERT;
                g.a = b.cur64[op.a];
                g.eo = out.on_words64;
                out.rec_offs64.push_back(g.eo);
                out.on_words64 += 8;
                g.x = (uint32_t)out.n_rec64++;
                const int32_t lvl = b.ssa_level64[g.a] + 1;
                b.emit64(g, (uint32_t)lvl);
                info.z64_asserts++;
                break;
            }
            default:
                return RV_E_BAD_OP;
            }
            break;
        }
        case RV_DOM_B2A: {  // interpreter/combine.rs:132-219, expanded in the reference's execution order
            const size_t dst = op.dst, src = op.a;
            if (dst >= b.cur64.size()) return RV_E_WIRE_OOB;
            if (src + 64 > b.cur.size() || src + 64 < src) return RV_E_WIRE_OOB;
            // 1. 64 fresh gf2 sharings a_k = {mask, corr 0}
            const uint32_t m2_first = (uint32_t)out.n_masks;
            uint32_t a[64], res[64], bw[64];
            for (int k = 0; k < 64; k++) a[k] = b.g_random();
            for (int k = 0; k < 64; k++) bw[k] = b.cur[src + k];
            // 2./3. z64 mask + correction are consumed here in program order (counters only)
            Gate64 g{};
            g.op = G64_B2A;
            g.m = (uint32_t)out.n_masks64++;
            g.m2 = m2_first;
            g.ep = out.pre_words64;
            out.pre_words64 += 1;
            g.xc = (uint32_t)out.n_corr64++;
            // 4. add_64 (combine.rs:39-93): 63 op_mul in carry order
            uint32_t carry = b.g_mul(a[0], bw[0]);
            res[0] = b.g_xor(a[0], bw[0]);
            for (int k = 1; k < 63; k++) {
                const uint32_t ac = b.g_xor(a[k], carry);
                const uint32_t bc = b.g_xor(bw[k], carry);
                const uint32_t t = b.g_mul(ac, bc);
                res[k] = b.g_xor(ac, bw[k]);
                carry = b.g_xor(t, carry);
            }
            res[63] = b.g_xor(carry, b.g_xor(a[63], bw[63]));
            // 5. 64 recorded reconstructions; outputs get 64 CONSECUTIVE SSA ids
            int32_t lvl_max = 0;
            uint32_t first_out = 0;
            for (int k = 0; k < 64; k++) {
                Gate r{};
                r.op = G_RECON;
                r.a = res[k];
                r.am = b.ssa_row[r.a];
                r.eo = (uint32_t)out.n_on++;
                r.x = (uint32_t)out.n_rec++;
                out.rec_rows.push_back(r.eo);
                const int32_t lvl = b.ssa_level[r.a] + 1;
                r.dst = b.new_ssa(lvl, Builder::COMP | 0);
                if (k == 0) first_out = r.dst;
                b.emit(r, (uint32_t)lvl);
                lvl_max = std::max(lvl_max, lvl);
                info.gf2_asserts++;
            }
            // 6. z64 wire = {0 - mu, Z - kappa}
            g.a = first_out;
            const int32_t lvl = lvl_max + 1;
            g.dst = b.new_ssa64(lvl);
            b.cur64[dst] = g.dst;
            b.emit64(g, (uint32_t)lvl);
            info.b2a++;
            break;
        }
        default:
            return RV_E_BAD_OP;
        }
        if (out.n_masks > LIM || b.ssa_level.size() > LIM || out.n_masks64 > LIM || b.ssa_level64.size() > LIM || out.n_on > LIM)
            return RV_E_UNSUPPORTED;
    }

    const uint32_t n_levels = b.any ? b.max_level + 1 : 0;
    sort_by_level(b.gates, b.level, n_levels, out.gates, out.level_start);
    sort_by_level(b.gates64, b.level64, n_levels, out.gates64, out.level_start64);
    // group each level by kind (gates of one level are independent, so any order is valid):
    // G_MUL first, then G_XOR, then everything else — lets the kernel run tight per-kind loops
    out.level_mul_end.assign(n_levels, 0);
    out.level_xor_end.assign(n_levels, 0);
    {
        std::vector<Gate> tmp;
        for (uint32_t l = 0; l < n_levels; l++) {
            const uint32_t lo = out.level_start[l], hi = out.level_start[l + 1];
            tmp.assign(out.gates.begin() + lo, out.gates.begin() + hi);
            uint32_t w = lo;
            for (const Gate& g : tmp)
                if (g.op == G_MUL) out.gates[w++] = g;
            out.level_mul_end[l] = w;
            for (const Gate& g : tmp)
                if (g.op == G_XOR) out.gates[w++] = g;
            out.level_xor_end[l] = w;
            for (const Gate& g : tmp)
                if (g.op != G_MUL && g.op != G_XOR) out.gates[w++] = g;
        }
    }
    // pipelining tables
    out.level_need_blocks.assign(n_levels, 0);
    out.level_done_on.assign(n_levels, 0);
    {
        std::vector<uint32_t> row_level(out.n_on, 0);
        for (uint32_t l = 0; l < n_levels; l++) {
            uint32_t need = l ? out.level_need_blocks[l - 1] : 0;
            for (uint32_t i = out.level_start[l]; i < out.level_start[l + 1]; i++) {
                const Gate& g = out.gates[i];
                uint32_t last = 0;
                bool uses = true;
                if (g.op == G_MUL)
                    last = g.m + 1;
                else if (g.op == G_INPUT || g.op == G_RANDOM)
                    last = g.m;
                else
                    uses = false;
                if (uses) need = std::max(need, last / 128 + 1);
                if (g.op == G_MUL || g.op == G_INPUT || g.op == G_ASSERT || g.op == G_RECON) row_level[g.eo] = l;
            }
            out.level_need_blocks[l] = need;
        }
        // aliased PRG rows are read by later levels too, but a row a gate reads through am/bm was
        // consumed as a fresh mask by an EARLIER level's gate, so the prefix maximum covers it
        uint64_t e = 0;
        uint32_t run = 0;
        for (uint32_t l = 0; l < n_levels; l++) {
            while (e < out.n_on && std::max(run, row_level[e]) <= l) {
                run = std::max(run, row_level[e]);
                e++;
            }
            out.level_done_on[l] = (uint32_t)e;
        }
    }
    out.n_ssa = b.ssa_level.size();
    out.n_ssa64 = b.ssa_level64.size();
    // resolve share rows: PRG masks first (padded to whole AES blocks), computed rows after
    out.n_masks_pad = (out.n_masks + 127) / 128 * 128;
    out.n_rows = out.n_masks_pad + b.n_comp;
    if (out.n_rows > LIM) return RV_E_UNSUPPORTED;
    auto fix = [&](uint32_t& r) {
        if (r & Builder::COMP) r = (uint32_t)(out.n_masks_pad + (r & ~Builder::COMP));
    };
    for (Gate& g : out.gates) {
        fix(g.dm);
        fix(g.am);
        fix(g.bm);
    }
    info.gf2_masks = out.n_masks;
    info.z64_masks = out.n_masks64;
    info.levels = n_levels;
    return RV_OK;
}

}  // namespace rv
