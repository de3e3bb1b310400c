// see compile.h
#include "compile.h"

#include <algorithm>
#include <limits>

namespace rv {

namespace {

struct Builder {
    std::vector<Gate> gates;      // program order
    std::vector<uint32_t> level;  // per gate
    std::vector<int32_t> ssa_level;
    std::vector<uint32_t> cur;    // gf2 wire index -> current SSA id
    Compiled& out;
    uint32_t max_level = 0;
    bool any = false;

    explicit Builder(Compiled& o) : out(o) { ssa_level.push_back(-1); }

    uint32_t new_ssa(int32_t lvl) {
        ssa_level.push_back(lvl);
        return (uint32_t)(ssa_level.size() - 1);
    }
    void emit(const Gate& g, uint32_t lvl) {
        gates.push_back(g);
        level.push_back(lvl);
        if (lvl > max_level) max_level = lvl;
        any = true;
    }
};

}  // namespace

int compile_ops(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out) {
    out = Compiled();
    Builder b(out);
    b.cur.assign(gf2_wires, 0);
    size_t n64 = z64_wires;
    (void)n64;
    rv_circuit_info& info = out.info;
    info.n_ops = n_ops;

    for (size_t i = 0; i < n_ops; i++) {
        const rv_op& op = ops[i];
        if (op.reserved != 0) return RV_E_BAD_OP;
        switch (op.domain) {
        case RV_DOM_SIZEHINT:
            if (op.b > b.cur.size()) b.cur.resize(op.b, 0);
            if (op.a > n64) n64 = op.a;
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
                g.dst = b.new_ssa(0);
                b.cur[op.dst] = g.dst;
                b.emit(g, 0);
                info.gf2_inputs++;
                break;
            case RV_OP_RANDOM:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G_RANDOM;
                g.m = (uint32_t)out.n_masks++;
                g.dst = b.new_ssa(0);
                b.cur[op.dst] = g.dst;
                b.emit(g, 0);
                info.gf2_linear++;
                break;
            case RV_OP_CONST:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                g.op = G_CONST;
                g.x = cbit;
                g.dst = b.new_ssa(0);
                b.cur[op.dst] = g.dst;
                b.emit(g, 0);
                info.gf2_linear++;
                break;
            case RV_OP_ADD:
            case RV_OP_SUB:
            case RV_OP_MUL: {
                if (op.dst >= nw || op.a >= nw || op.b >= nw) return RV_E_WIRE_OOB;
                g.a = b.cur[op.a];
                g.b = b.cur[op.b];
                const int32_t lvl = std::max(b.ssa_level[g.a], b.ssa_level[g.b]) + 1;
                if (op.opcode == RV_OP_MUL) {
                    g.op = G_MUL;
                    g.m = (uint32_t)out.n_masks;
                    out.n_masks += 2;
                    g.eo = (uint32_t)out.n_on++;
                    g.ep = (uint32_t)out.n_pre++;
                    g.x = (uint32_t)out.n_rec++;
                    out.rec_rows.push_back(g.eo);
                    info.gf2_muls++;
                } else {
                    g.op = G_XOR;  // gf2/share.rs:220-238: Add and Sub are both XOR
                    info.gf2_linear++;
                }
                g.dst = b.new_ssa(lvl);
                b.cur[op.dst] = g.dst;
                b.emit(g, (uint32_t)lvl);
                break;
            }
            case RV_OP_ADDCONST:
            case RV_OP_SUBCONST:
            case RV_OP_MULCONST: {
                if (op.dst >= nw || op.a >= nw) return RV_E_WIRE_OOB;
                g.a = b.cur[op.a];
                g.op = (op.opcode == RV_OP_MULCONST) ? G_ANDC : G_XORC;
                g.x = cbit;
                const int32_t lvl = b.ssa_level[g.a] + 1;
                g.dst = b.new_ssa(lvl);
                b.cur[op.dst] = g.dst;
                b.emit(g, (uint32_t)lvl);
                info.gf2_linear++;
                break;
            }
            case RV_OP_ASSERTZERO: {
                if (op.a >= nw) return RV_E_WIRE_OOB;
                g.op = G_ASSERT;
                g.a = b.cur[op.a];
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
        case RV_DOM_Z64:
        case RV_DOM_B2A:
            return RV_E_UNSUPPORTED;  // TODO(round 1 later): Z64 ring + B2A
        default:
            return RV_E_BAD_OP;
        }
        if (out.n_masks > std::numeric_limits<uint32_t>::max() - 4 || b.ssa_level.size() > std::numeric_limits<uint32_t>::max() - 4)
            return RV_E_UNSUPPORTED;
    }

    // counting sort by level (stable: program order inside a level)
    const uint32_t n_levels = b.any ? b.max_level + 1 : 0;
    out.level_start.assign(n_levels + 1, 0);
    for (uint32_t l : b.level) out.level_start[l + 1]++;
    for (uint32_t l = 0; l < n_levels; l++) out.level_start[l + 1] += out.level_start[l];
    out.gates.resize(b.gates.size());
    {
        std::vector<uint32_t> pos(out.level_start.begin(), out.level_start.end() - (n_levels ? 1 : 0));
        if (!n_levels) pos.clear();
        for (size_t i = 0; i < b.gates.size(); i++) out.gates[pos[b.level[i]]++] = b.gates[i];
    }
    out.n_ssa = b.ssa_level.size();
    info.gf2_masks = out.n_masks;
    info.levels = n_levels;
    return RV_OK;
}

}  // namespace rv
