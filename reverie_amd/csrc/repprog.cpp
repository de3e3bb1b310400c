// see repprog.h
#include "repprog.h"

#include <stdlib.h>

#include <algorithm>
#include <map>

namespace rv {

namespace {

// free LDS quads (4 slots each) as disjoint intervals; first fit, coalescing on release
struct QuadPool {
    std::map<uint32_t, uint32_t> free;  // start quad -> length
    uint32_t peak = 0, total = 0, in_use = 0, high = 0;  // high: one past the highest quad ever handed out (= the LDS needed)
    explicit QuadPool(uint32_t n_quads) : total(n_quads) { free[0] = n_quads; }
    bool take(uint32_t n, uint32_t* start) {
        for (auto it = free.begin(); it != free.end(); ++it) {
            if (it->second < n) continue;
            *start = it->first;
            const uint32_t rest = it->second - n, at = it->first + n;
            free.erase(it);
            if (rest) free[at] = rest;
            in_use += n;
            peak = std::max(peak, in_use);
            high = std::max(high, *start + n);
            return true;
        }
        return false;
    }
    void give(uint32_t start, uint32_t n) {
        in_use -= n;
        auto nx = free.lower_bound(start);
        if (nx != free.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == start) {
                start = pv->first;
                n += pv->second;
                free.erase(pv);
            }
        }
        if (nx != free.end() && start + n == nx->first) {
            n += nx->second;
            free.erase(nx);
        }
        free[start] = n;
    }
};

}  // namespace

bool build_rep_program(const Compiled& cc, uint32_t lds_slots, RepProgram& out, const char** why) {
    auto no = [&](const char* w) {
        if (why) *why = w;
        out = RepProgram();
        return false;
    };
    if (!cc.gates64.empty() || cc.n_masks64 || cc.on_words64 || cc.pre_words64) return no("Z64 gates");
    if (cc.row_prg_base) return no("streaming chunk");
    const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
    const size_t n_gates = cc.gates.size();
    // ---- eligibility + last use of every row
    std::vector<int32_t> last_use(cc.n_rows, -1);
    std::vector<uint32_t> level_of(n_gates);
    for (size_t l = 0; l < n_levels; l++)
        for (uint32_t i = cc.level_start[l]; i < cc.level_start[l + 1]; i++) {
            const Gate& g = cc.gates[i];
            level_of[i] = (uint32_t)l;
            const uint32_t op = g_op(g), na = g_na(g), nb = g_nb(g);
            switch (op) {
            case G_MUL:
                if (na > 1 || nb > 1) return no("multi-base Mul operand (compile with one base per wire)");
                break;
            case G_XORK:
                if (na != 2 || nb != 0) return no("Xor of other than two rows");
                break;
            case G_ASSERT:
                if (na > 1) return no("multi-base AssertZero operand");
                break;
            case G_INPUT:
                break;
            default:
                return no("Random / B2A gate (wire values differ between repetitions)");
            }
            for (uint32_t k = 0; k < na; k++) last_use[g.a[k]] = std::max(last_use[g.a[k]], (int32_t)l);
            for (uint32_t k = 0; k < nb; k++) last_use[g.b[k]] = std::max(last_use[g.b[k]], (int32_t)l);
        }
    // ---- segments level by level, LDS quads by liveness
    const uint32_t n_quads = lds_slots / 4;
    if (n_quads < 2) return no("no LDS");
    QuadPool pool(n_quads);
    uint32_t zero_quad = 0;
    pool.take(1, &zero_quad);  // slot 0: the constant-zero row (mask 0, value 0)
    std::vector<uint32_t> slot(cc.n_rows, 0xFFFFFFFFu);
    slot[cc.zero_row] = 0;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> dies(n_levels + 1);  // level -> (start quad, n quads) freed after it
    out.levels.assign(n_levels, RepLevel{0, 0});
    auto operand = [&](const Gate& g, bool second, uint32_t* enc) -> bool {
        const uint32_t n = second ? g_nb(g) : g_na(g);
        const uint32_t c = second ? g_cb(g) : g_ca(g);
        uint32_t s = 0;  // an operand without a base row is the constant c: the zero slot plus the constant
        if (n) {
            s = slot[second ? g.b[0] : g.a[0]];
            if (s == 0xFFFFFFFFu) return false;
        }
        *enc = s | (c << 31);
        return true;
    };
    for (size_t l = 0; l < n_levels; l++) {
        out.levels[l].seg0 = (uint32_t)out.segs.size();
        const uint32_t lo = cc.level_start[l], hi = cc.level_start[l + 1];
        // the level's gates by kind, program order inside a kind (the compiler's stable sort keeps it)
        uint32_t i = lo;
        while (i < hi) {
            const Gate& g0 = cc.gates[i];
            const uint32_t op = g_op(g0);
            RepSeg s{};
            s.kind = op == G_MUL ? RS_MUL : op == G_XORK ? RS_XOR : op == G_INPUT ? RS_INPUT : RS_ASSERT;
            s.m0 = g0.m;
            s.eo0 = g0.eo;
            s.ep0 = g0.ep;
            s.x0 = g0.x;
            uint32_t n = 1;
            while (i + n < hi && n < REP_SEG_MAX) {
                const Gate& g = cc.gates[i + n];
                if (g_op(g) != op) break;
                if (op == G_MUL && (g.ep != s.ep0 + n || g.m != s.m0 + 2 * n || g.eo != s.eo0 + n)) break;
                if (op == G_INPUT && (g.m != s.m0 + n || g.eo != s.eo0 + n || g.x != s.x0 + n)) break;
                if (op == G_ASSERT && g.eo != s.eo0 + n) break;
                n++;
            }
            s.count = n;
            s.off = (op == G_MUL || op == G_INPUT) ? (s.eo0 & 3u) : 0u;
            // records: one {a, b} pair per gate behind `off` dummies, padded to a multiple of four per segment
            const bool has_recs = op != G_INPUT;
            if ((uint64_t)(out.segs.size() + 1) * REP_SEG_RECS > 0xFFFFFFFFull) return no("too many segments");
            s.first = (uint32_t)(out.segs.size() * REP_SEG_RECS);
            s.vb0 = (uint32_t)(out.segs.size() * 64);
            out.recs.resize((out.segs.size() + 1) * (size_t)REP_SEG_RECS, RepRec{0, 0});
            if (has_recs) {
                size_t at = s.first + s.off;
                for (uint32_t k = 0; k < n; k++, at++) {
                    const Gate& g = cc.gates[i + k];
                    RepRec r{0, 0};
                    bool ok = true;
                    if (op == G_XORK) {  // dst = a[0] ^ a[1] ^ const
                        const uint32_t sa = slot[g.a[0]], sb = slot[g.a[1]];
                        ok = sa != 0xFFFFFFFFu && sb != 0xFFFFFFFFu;
                        r.a = sa | (g_ca(g) << 31);
                        r.b = sb;
                    } else {
                        ok = operand(g, false, &r.a) && (op == G_ASSERT || operand(g, true, &r.b));
                    }
                    if (!ok) return no("internal: operand row without a slot");
                    out.recs[at] = r;
                }
            }
            // output slots: one contiguous run of quads per segment (AssertZero writes nothing)
            if (op != G_ASSERT) {
                const uint32_t nq = (s.off + n + 3) / 4;
                uint32_t q0 = 0;
                if (!pool.take(nq, &q0)) return no("live wires do not fit the LDS");
                s.dst0 = 4 * q0;
                int32_t dead = (int32_t)l;  // the run is released once its last reader has run
                for (uint32_t k = 0; k < n; k++) {
                    const Gate& g = cc.gates[i + k];
                    slot[g.dst] = s.dst0 + s.off + k;
                    dead = std::max(dead, last_use[g.dst]);
                }
                dies[(size_t)dead].emplace_back(q0, nq);
            }
            out.segs.push_back(s);
            i += n;
        }
        out.levels[l].seg1 = (uint32_t)out.segs.size();
        for (const auto& d : dies[l]) pool.give(d.first, d.second);
        dies[l].clear();
        dies[l].shrink_to_fit();
    }
    out.n_vb_words = (uint32_t)(out.segs.size() * 64);
    out.lds_slots = std::max<uint32_t>(4 * pool.high, 4);
    out.n_levels = (uint32_t)n_levels;
    if (why) *why = "";
    return true;
}

}  // namespace rv
