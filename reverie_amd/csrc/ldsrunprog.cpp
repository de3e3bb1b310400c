// see ldsrun.h
#include "ldsrun.h"

#include <algorithm>
#include <functional>
#include <queue>

namespace rv {

namespace {
constexpr uint32_t UNSET = 0xFFFFFFFFu, DEFINED = 0xFFFFFFFEu, LIVE_IN = 0xFFFFFFFDu;

uint16_t op_word(uint32_t kind, uint32_t flags) {
    const uint32_t hot = kind == G_MUL ? LB_MUL : kind == G_XORK ? LB_XOR : kind == G_RECON ? LB_RECON : kind == G_INPUT ? LB_IN : LB_OTHER;
    return (uint16_t)(kind | flags | (1u << hot) | (kind == G_ASSERT ? 1u << LB_ASSERT : 0u));
}
LdsRec nop_rec() {
    LdsRec r{};
    r.dst = (uint16_t)LR_NONE;
    r.op = op_word(LK_NOP, 0);
    return r;
}
}  // namespace

// first_level: the first level of the circuit's first narrow stretch -- a row's readers before that level cannot be "later
// readers" of any run, so the scan starts there (for a wide circuit with a narrow tail that is a few thousand of 10^7 gates)
void LdsRunScratch::init(const Compiled& cc, uint32_t first_level) {
    last_use_level.init(cc.n_rows);
    slot_of.init(cc.n_rows);
    last_step.init(cc.n_rows);
    const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
    for (size_t l = first_level; l < n_levels; l++)
        for (uint32_t i = cc.level_start[l]; i < cc.level_start[l + 1]; i++) {
            const Gate& g = cc.gates[i];
            for (int k = 0; k < RV_LIN_K; k++) {
                last_use_level.set(g.a[k], (uint32_t)l);
                last_use_level.set(g.b[k], (uint32_t)l);
            }
        }
    // B2A reads 64 consecutive GF(2) wires' rows from the Z64 interpreter: those rows must stay in global memory
    for (size_t l = first_level; l + 1 < cc.level_start64.size(); l++)
        for (uint32_t i = cc.level_start64[l]; i < cc.level_start64[l + 1]; i++) {
            const Gate64& g = cc.gates64[i];
            if (g.op != G64_B2A) continue;
            for (uint32_t k = 0; k < 64; k++)
                if ((uint64_t)g.a + k < cc.n_rows) last_use_level.set(g.a + k, (uint32_t)std::max(last_use_level.geti(g.a + k), (int32_t)l));
        }
}

bool build_lds_run(const Compiled& cc, uint32_t l0, uint32_t l1, uint32_t QS, uint32_t max_slots, LdsRunScratch& S,
                   std::vector<LdsRec>& recs, LdsRun& run) {
    const uint32_t GPS = 64 / QS;
    const uint32_t zero = (uint32_t)cc.zero_row;
    std::vector<uint32_t> touched, live_in;
    auto reset = [&] {
        for (uint32_t r : touched) {
            S.slot_of.set(r, UNSET);
            S.last_step.set(r, UNSET);
        }
    };
    auto touch = [&](uint32_t r, uint32_t mark) {
        if (S.slot_of.get(r) == UNSET) {
            S.slot_of.set(r, mark);
            touched.push_back(r);
        }
    };
    // ---- pass 1: live-in rows (read before any definition inside the run)
    for (uint32_t i = cc.level_start[l0]; i < cc.level_start[l1]; i++) {
        const Gate& g = cc.gates[i];
        for (int k = 0; k < 2 * RV_LIN_K; k++) {
            const uint32_t r = k < RV_LIN_K ? g.a[k] : g.b[k - RV_LIN_K];
            if (r == zero || S.slot_of.get(r) != UNSET) continue;
            touch(r, LIVE_IN);
            live_in.push_back(r);
        }
        if (g_op(g) != G_ASSERT) touch(g.dst, DEFINED);
    }
    // ---- step numbering: load steps, then every level cut into steps of GPS gates
    const uint32_t n_load_steps = (uint32_t)((live_in.size() + GPS - 1) / GPS);
    std::vector<uint32_t> level_step0(l1 - l0 + 1);
    uint32_t n_steps = n_load_steps;
    for (uint32_t l = l0; l < l1; l++) {
        level_step0[l - l0] = n_steps;
        n_steps += (cc.level_start[l + 1] - cc.level_start[l] + GPS - 1) / GPS;
    }
    level_step0[l1 - l0] = n_steps;
    const uint32_t n_steps_pad = (n_steps + LR_CHUNK - 1) / LR_CHUNK * LR_CHUNK;
    // ---- pass 2: last step that reads each row
    for (uint32_t l = l0; l < l1; l++)
        for (uint32_t i = cc.level_start[l]; i < cc.level_start[l + 1]; i++) {
            const Gate& g = cc.gates[i];
            const int32_t st = (int32_t)(level_step0[l - l0] + (i - cc.level_start[l]) / GPS);
            for (int k = 0; k < RV_LIN_K; k++) {
                if (g.a[k] != zero) S.last_step.set(g.a[k], (uint32_t)st);
                if (g.b[k] != zero) S.last_step.set(g.b[k], (uint32_t)st);
            }
        }
    // ---- pass 3: slots by liveness (lowest free slot first, so the high-water mark stays small) and the records
    std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> free_slots;
    uint32_t fresh = 1;  // slot 0: the zero wire
    bool fits = true;
    auto take = [&]() -> uint32_t {
        if (!free_slots.empty()) {
            const uint32_t s = free_slots.top();
            free_slots.pop();
            return s;
        }
        if (fresh >= max_slots || fresh >= LR_NONE) {
            fits = false;
            return 0;
        }
        return fresh++;
    };
    std::vector<std::vector<uint32_t>> dies(n_steps_pad + 1);
    const size_t rec_base = recs.size();
    recs.resize(rec_base + (size_t)n_steps_pad * GPS, nop_rec());
    auto define = [&](uint32_t row, uint32_t step) -> uint16_t {  // slot for a row written at `step` (LR_NONE if nothing here reads it)
        if (S.last_step.geti(row) < 0) {
            S.slot_of.set(row, DEFINED);
            return (uint16_t)LR_NONE;
        }
        const uint32_t s = take();
        S.slot_of.set(row, s);
        dies[(size_t)S.last_step.geti(row)].push_back(s);
        (void)step;
        return (uint16_t)s;
    };
    auto release = [&](uint32_t step) {
        for (uint32_t s : dies[step]) free_slots.push(s);
        std::vector<uint32_t>().swap(dies[step]);
    };
    uint32_t step = 0;
    uint32_t eo_lo = 0xFFFFFFFFu, eo_hi = 0, ep_lo = 0xFFFFFFFFu, ep_hi = 0;
    for (size_t i = 0; i < live_in.size() && fits; i++) {
        step = (uint32_t)(i / GPS);
        LdsRec& r = recs[rec_base + (size_t)step * GPS + i % GPS];
        r.op = op_word(LK_LOAD, 0);
        r.m = live_in[i];
        r.dst = define(live_in[i], step);
        if (i % GPS == GPS - 1 || i + 1 == live_in.size()) release(step);
    }
    for (uint32_t l = l0; l < l1 && fits; l++) {
        const uint32_t lo = cc.level_start[l], hi = cc.level_start[l + 1];
        for (uint32_t i = lo; i < hi && fits; i++) {
            const Gate& g = cc.gates[i];
            step = level_step0[l - l0] + (i - lo) / GPS;
            LdsRec& r = recs[rec_base + (size_t)step * GPS + (i - lo) % GPS];
            const uint32_t op = g_op(g);
            for (int k = 0; k < RV_LIN_K; k++) {
                r.a[k] = g.a[k] == zero ? 0 : (uint16_t)S.slot_of.get(g.a[k]);
                r.b[k] = g.b[k] == zero ? 0 : (uint16_t)S.slot_of.get(g.b[k]);
            }
            uint32_t flags = (g_ca(g) ? LF_CA : 0u) | (g_cb(g) ? LF_CB : 0u);
            if (op == G_INPUT || op == G_MUL || op == G_ASSERT || op == G_RECON) {
                flags |= LF_ON;
                eo_lo = std::min(eo_lo, g.eo);
                eo_hi = std::max(eo_hi, g.eo);
            }
            if (op == G_MUL) {
                ep_lo = std::min(ep_lo, g.ep);
                ep_hi = std::max(ep_hi, g.ep);
            }
            r.eo = g.eo;
            r.ep = g.ep;
            r.x = g.x;
            r.m = (op == G_XORK || op == G_RECON) ? g.dst : g.m;
            if (op != G_ASSERT) {
                r.dst = define(g.dst, step);
                if (S.last_use_level.geti(g.dst) >= (int32_t)l1) flags |= LF_OUT;
            }
            r.op = op_word(op, flags);
            if ((i - lo) % GPS == GPS - 1 || i + 1 == hi) release(step);
        }
    }
    reset();
    if (fits && (fresh >= max_slots || fresh >= LR_NONE)) fits = false;  // (the scratch slot)
    // transcript stores use 32-bit byte offsets from the run's lowest row (rows are at most 256 bytes)
    if (eo_lo > eo_hi) eo_lo = eo_hi = 0;
    if (ep_lo > ep_hi) ep_lo = ep_hi = 0;
    if ((uint64_t)(eo_hi - eo_lo) >= (1ull << 24) - 1 || (uint64_t)(ep_hi - ep_lo) >= (1ull << 24) - 1) fits = false;
    if (!fits) {
        recs.resize(rec_base);
        return false;
    }
    for (size_t i = rec_base; i < recs.size(); i++)
        if (recs[i].dst == LR_NONE) recs[i].dst = (uint16_t)fresh;  // results nothing here reads, no-ops
    fresh++;
    run.l0 = l0;
    run.l1 = l1;
    run.n_steps = n_steps_pad;
    run.n_slots = fresh;
    run.rec0 = rec_base;
    run.eo0 = eo_lo;
    run.ep0 = ep_lo;
    return true;
}

}  // namespace rv
