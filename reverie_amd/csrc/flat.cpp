// see flat.h
#include "flat.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>

namespace rv {

bool build_flat_plan(const Compiled& cc, FlatPlan& P, uint32_t want_bands) {
    P.ok = false;
    P.xgates.clear();
    P.xlevels.clear();
    P.muls.clear();
    P.others.clear();
    P.bands.clear();
    P.clear_s.clear();
    P.clear_k.clear();
    P.clear_levels.clear();
    P.lite_s.clear();
    P.lite_k.clear();
    P.lite_levels.clear();
    if (!cc.gates64.empty() || cc.row_prg_base || cc.n_random_or_recon || cc.gates.empty()) return false;
    if (cc.n_on >= (1ull << MULREC_EO_BITS) || cc.n_rows >= 0xFFFFFFF0ull || cc.level_start.size() < 2) return false;
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t zero_row = cc.zero_row;
    const uint64_t n_comp = cc.n_rows - zero_row;
    const size_t n = cc.gates.size();
    P.n_clear_levels = cc.level_start.size() - 1;

    // ---- bands: equal ranges of the program's Mul gates (multiples of 1024: the early-corrections chunks are whole bytes) ----
    uint64_t n_mul = 0, n_other = 0, n_xor = 0;
    for (size_t i = 0; i < n; i++) {
        const Gate& g = cc.gates[i];
        switch (g_op(g)) {
        case G_MUL:
            if (g.ep >= cc.n_pre) return false;
            n_mul++;
            break;
        case G_XORK:
            if (g.dst <= zero_row) return false;  // (cannot happen: an XOR writes a computed row)
            n_xor++;
            break;
        case G_INPUT:
        case G_ASSERT: n_other++; break;
        default: return false;  // Random / Recon: values that differ between repetitions
        }
    }
    if (n_mul != cc.n_pre) return false;
    std::vector<uint32_t> band_mul0{0};  // first Mul of every band
    if (want_bands > 1)
        for (uint32_t k = 1; k < want_bands; k++) {
            const uint64_t at = (n_mul * k / want_bands + 1023) & ~1023ull;
            if (at > band_mul0.back() && at < n_mul) band_mul0.push_back((uint32_t)at);
        }
    const size_t n_bands = band_mul0.size();
    auto band_of_mul = [&](uint32_t ep) { return (uint32_t)(std::upper_bound(band_mul0.begin(), band_mul0.end(), ep) - band_mul0.begin()) - 1; };
    // The band an XOR row is computed in = the first band that reads it, directly (a Mul of the band) or through another XOR row
    // of the band; AssertZero gates read theirs behind the last band.  Backwards over the topological order.
    std::vector<uint32_t> need(n_comp, (uint32_t)n_bands - 1);
    for (size_t i = n; i-- > 0;) {
        const Gate& g = cc.gates[i];
        const uint32_t op = g_op(g);
        uint32_t b;
        if (op == G_MUL)
            b = band_of_mul(g.ep);
        else if (op == G_XORK)
            b = need[g.dst - zero_row];
        else
            continue;
        for (int k = 0; k < RV_LIN_K; k++)
            for (uint32_t row : {g.a[k], g.b[k]})
                if (row > zero_row && need[row - zero_row] > b) need[row - zero_row] = b;
    }
    // ---- x-levels: XOR -> XOR depth inside a band (rows of earlier bands are complete: depth 0, like the PRG rows) ----
    // The level-sorted gate array is a topological order; one pass over it.
    std::vector<uint32_t> xl(n_comp, 0);      // per computed row: x-level inside its band
    std::vector<uint32_t> gxl(n, 0);          // per gate (G_XORK): x-level, 1-based
    std::vector<uint32_t> gband(n, 0);        // per gate (G_XORK): its band
    std::vector<uint32_t> band_depth(n_bands, 0);
    for (size_t i = 0; i < n; i++) {
        const Gate& g = cc.gates[i];
        if (g_op(g) != G_XORK) continue;
        const uint32_t b = need[g.dst - zero_row];
        uint32_t d = 0;
        for (int k = 0; k < RV_LIN_K; k++)
            for (uint32_t row : {g.a[k], g.b[k]})
                if (row > zero_row && need[row - zero_row] == b) d = std::max(d, xl[row - zero_row]);
        d++;
        xl[g.dst - zero_row] = d;
        gxl[i] = d;
        gband[i] = b;
        band_depth[b] = std::max(band_depth[b], d);
    }
    // counting sort of the XOR gates by (band, x-level, class)
    std::vector<uint32_t> xbase(n_bands + 1, 0);  // first x-level slot of every band
    for (size_t b = 0; b < n_bands; b++) xbase[b + 1] = xbase[b] + band_depth[b];
    const size_t n_xl = xbase[n_bands];
    std::vector<uint32_t> pos(n_xl * 2 + 1, 0);
    auto cls = [](const Gate& g) -> uint32_t { return (g_na(g) == 2 && g_nb(g) == 0) ? 0u : 1u; };
    for (size_t i = 0; i < n; i++)
        if (gxl[i]) pos[(size_t)(xbase[gband[i]] + gxl[i] - 1) * 2 + cls(cc.gates[i]) + 1]++;
    for (size_t k = 0; k < n_xl * 2; k++) pos[k + 1] += pos[k];
    P.xlevels.resize(n_xl);
    for (size_t k = 0; k < n_xl; k++) {
        const uint32_t* e = &pos[k * 2];
        P.xlevels[k] = LevelRange{e[0], e[0], e[0], e[1], e[2], e[2]};
    }
    P.xgates.resize(n_xor);
    P.muls.resize(n_mul);
    P.others.reserve(n_other);
    for (size_t i = 0; i < n; i++) {
            const Gate& g = cc.gates[i];
            const uint32_t op = g_op(g);
            if (op == G_XORK) {
                P.xgates[pos[(size_t)(xbase[gband[i]] + gxl[i] - 1) * 2 + cls(g)]++] = g;
            } else if (op == G_MUL) {
                MulRec r;
                for (int k = 0; k < RV_LIN_K; k++) r.a[k] = g.a[k], r.b[k] = g.b[k];
                r.m = g.m;
                const uint32_t na = std::max(g_na(g), 1u) - 1, nb = std::max(g_nb(g), 1u) - 1;
                r.eo_flags = g.eo | (na << 26) | (nb << 28) | (g_ca(g) << 30) | (g_cb(g) << 31);
                P.muls[g.ep] = r;
            } else if (op == G_INPUT) {
                P.others.push_back(g);
            }
        }
    P.n_other_inputs = (uint32_t)P.others.size();
    for (size_t i = 0; i < n; i++)
        if (g_op(cc.gates[i]) == G_ASSERT) P.others.push_back(cc.gates[i]);
    // ---- value records: the level-sorted stream again, 16 or 32 bytes per gate (every gate for the cleartext pass of the flat
    // schedule; everything but the XOR gates for the split schedule's level chain) ----
    std::vector<uint32_t> mul_level(n_bands, 0);  // per band: 1 + the highest dependency level of its Mul gates
    {
        const size_t n_levels = cc.level_start.size() - 1;
        // XOR / AssertZero: one list of bases (the sum is symmetric): two of them still make a 16-byte record
        auto simple = [](const Gate& g) {
            const uint32_t op = g_op(g);
            return (op == G_XORK || op == G_ASSERT) ? g_na(g) + g_nb(g) <= 2 : (g_na(g) <= 1 && g_nb(g) <= 1);
        };
        auto build = [&](bool with_xor, decltype(P.clear_s)& S, decltype(P.clear_k)& K, std::vector<ClearLevel>& LV) {
            size_t ns = 0, nk = 0;
            for (size_t i = 0; i < n; i++)
                if (with_xor || g_op(cc.gates[i]) != G_XORK) (simple(cc.gates[i]) ? ns : nk)++;
            S.resize(ns);
            K.resize(nk);
            LV.resize(n_levels);
            size_t is = 0, ik = 0;
            for (size_t l = 0; l < n_levels; l++) {
                ClearLevel L{(uint32_t)is, 0, (uint32_t)ik, 0};
                for (uint32_t i = cc.level_start[l]; i < cc.level_start[l + 1]; i++) {
                    const Gate& g = cc.gates[i];
                    const uint32_t op = g_op(g);
                    if (!with_xor && op == G_XORK) continue;
                    uint32_t na = g_na(g), nb = g_nb(g);
                    if (simple(g)) {
                        uint32_t a0 = g.a[0], b0 = g.b[0];
                        if (op == G_INPUT) {
                            a0 = g.x, na = nb = 0;
                        } else if (op == G_XORK || op == G_ASSERT) {
                            uint32_t rows[2] = {0, 0}, cnt = 0;
                            for (uint32_t k = 0; k < na; k++) rows[cnt++] = g.a[k];
                            for (uint32_t k = 0; k < nb; k++) rows[cnt++] = g.b[k];
                            a0 = rows[0], b0 = rows[1], na = cnt >= 1, nb = cnt >= 2;
                        }
                        S[is++] = ClearRec{g.dst, a0, b0, op | (g_ca(g) << 3) | (g_cb(g) << 4) | (na << 8) | (nb << 10)};
                    } else {
                        ClearRecK r;
                        r.dst = g.dst, r.meta = op | (g_ca(g) << 3) | (g_cb(g) << 4) | (na << 8) | (nb << 10);
                        for (int k = 0; k < RV_LIN_K; k++) r.a[k] = g.a[k], r.b[k] = g.b[k];
                        K[ik++] = r;
                    }
                }
                L.s1 = (uint32_t)is, L.g1 = (uint32_t)ik;
                LV[l] = L;
            }
        };
        build(true, P.clear_s, P.clear_k, P.clear_levels);
        build(false, P.lite_s, P.lite_k, P.lite_levels);
        for (size_t l = 0; l < n_levels; l++)
            for (uint32_t i = cc.level_start[l]; i < cc.level_start[l + 1]; i++)
                if (g_op(cc.gates[i]) == G_MUL) {
                    const uint32_t b = band_of_mul(cc.gates[i].ep);
                    mul_level[b] = std::max(mul_level[b], (uint32_t)l + 1);
                }
        for (size_t b = 1; b < n_bands; b++) mul_level[b] = std::max(mul_level[b], mul_level[b - 1]);
    }
    // leading online rows final after band b: everything before the first row that is still to come -- the first Mul of the
    // next band, or the first AssertZero (their rows are written behind the last band)
    std::vector<uint32_t> on_end(n_bands, (uint32_t)cc.n_on);
    {
        uint64_t first_assert = cc.n_on;
        std::vector<uint64_t> first_mul_eo(n_bands + 1, cc.n_on);  // smallest eo among the Mul gates of band b
        for (size_t i = 0; i < n; i++) {
            const Gate& g = cc.gates[i];
            if (g_op(g) == G_ASSERT) first_assert = std::min<uint64_t>(first_assert, g.eo);
            if (g_op(g) == G_MUL) {
                const uint32_t b = band_of_mul(g.ep);
                first_mul_eo[b] = std::min<uint64_t>(first_mul_eo[b], g.eo);
            }
        }
        uint64_t later = cc.n_on;  // smallest eo among the Mul gates of the bands behind b
        for (size_t b = n_bands; b-- > 0;) {
            on_end[b] = (uint32_t)std::min(later, first_assert);
            later = std::min(later, first_mul_eo[b]);
        }
    }
    P.bands.resize(n_bands);
    for (size_t b = 0; b < n_bands; b++)
        P.bands[b] = FlatPlan::Band{xbase[b], xbase[b + 1], band_mul0[b], b + 1 < n_bands ? band_mul0[b + 1] : (uint32_t)n_mul, mul_level[b], on_end[b]};
    P.ok = true;
    if (getenv("RV_COMPILE_STATS")) {
        fprintf(stderr, "[rv flat] %zu gates -> %llu Mul (program order), %llu XOR rows, %llu others in %zu bands, %zu x-levels; %.1f ms\n", n,
                (unsigned long long)n_mul, (unsigned long long)n_xor, (unsigned long long)n_other, n_bands, n_xl,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        for (size_t b = 0; b < n_bands; b++) {
            const auto& B = P.bands[b];
            fprintf(stderr, "[rv flat]   band %zu: Mul [%u, %u), %u x-levels:", b, B.mul0, B.mul1, B.x1 - B.x0);
            for (uint32_t k = B.x0; k < B.x1 && k < B.x0 + 12; k++) fprintf(stderr, " %u", P.xlevels[k].hi - P.xlevels[k].lo);
            fprintf(stderr, "%s\n", B.x1 - B.x0 > 12 ? " ..." : "");
        }
    }
    return true;
}

}  // namespace rv
