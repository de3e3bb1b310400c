// see compile.h
#include "compile.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <limits>

namespace rv {

namespace {

constexpr int K = RV_LIN_K;
constexpr uint32_t COMP = 0x80000000u;  // computed-row flag (resolved to n_masks_pad + index at the end)
constexpr uint32_t CARRY = 0x40000000u; // carried wire row of a streaming chunk (resolved to the wire index)
constexpr uint32_t ZERO_ROW = COMP | 0;

// a wire as a linear form over base rows: XOR of b[0..n) (sorted, distinct) plus the constant c
struct Lin {
    uint8_t n = 0, c = 0;
    uint32_t b[K] = {0, 0, 0};
};

struct Builder {
    Compiled& out;
    bool counting;  // pass 1: only number the SSA wires and count how often each is read
    bool have_uses = true;  // false: there was no pass 1 (streaming chunks) -- every wire counts as read
    int lazy_k = 1; // largest base set a wire may keep symbolically (1 = aliases and constants only)
    uint32_t lazy_slack = 1;  // extra row reads a symbolic wire may cost over materialising it (f readers x (n - 1) rows vs n + 1)
    // deep narrow circuits: when a sum outgrows lazy_k rows, up to `balance` of its LATEST rows stay symbolic and only the strictly
    // earlier ones are summed into a computed row -- that row is ready before the late ones are, so the sum costs its consumer
    // no extra dependency level (0: the whole sum becomes one row, a level behind its latest base)
    int balance = 0;
    // GF(2)
    std::vector<Gate> gates;      // program order
    std::vector<uint32_t> level;  // per gate
    std::vector<Lin> lin;         // per SSA wire (pass 2)
    std::vector<uint32_t>& uses;  // per SSA wire: reads (filled by pass 1, read by pass 2)
    uint32_t n_ssa = 1;
    std::vector<int32_t> lvl_prg, lvl_comp;  // level at which a base row becomes available
    uint32_t n_comp = 1;                     // computed rows; 0 = the all-zero row
    std::vector<uint32_t> cur;               // gf2 wire index -> current SSA id
    // Z64
    std::vector<Gate64> gates64;
    std::vector<uint32_t> level64;
    std::vector<int32_t> ssa_level64;
    std::vector<uint32_t> ssa_row64;  // z64 SSA id -> where its mask row lives (wmask row id, or G64_MASK_ROW | PRG mask row)
    std::vector<uint32_t> cur64;
    uint32_t max_level = 0;
    bool any = false;

    Builder(Compiled& o, bool counting_, std::vector<uint32_t>& uses_, size_t size_hint = 0) : out(o), counting(counting_), uses(uses_) {
        if (!counting) {
            // pass 1 left one entry per SSA wire in `uses`: size the big vectors once (10^7 gates x 48 B would otherwise
            // be copied several times over while the vector grows)
            const size_t n = std::max(uses.size(), size_hint);
            lin.reserve(n + 1);
            gates.reserve(n);
            level.reserve(n);
            lvl_comp.reserve(n / 2 + 16);
            lvl_prg.reserve(n + 16);
        }
        lin.emplace_back();  // SSA 0: the default wire = constant 0
        lvl_comp.push_back(-1);
        ssa_level64.push_back(-1);
        ssa_row64.push_back(0);
        if (counting) uses.assign(1, 0);
    }

    int32_t row_level(uint32_t r) const { return (r & COMP) ? lvl_comp[r & ~COMP] : (r & CARRY) ? -1 : lvl_prg[r]; }
    int32_t lin_level(const Lin& L) const {
        int32_t l = -1;
        for (int i = 0; i < L.n; i++) l = std::max(l, row_level(L.b[i]));
        return l;
    }
    void set_prg_level(uint32_t m, int32_t l) {
        if (lvl_prg.size() <= m) lvl_prg.resize((size_t)m + 1, -1);  // grows by one or two per gate: amortised by the reserve / doubling
        lvl_prg[m] = l;
    }
    uint32_t new_ssa(const Lin& L) {
        if (counting)
            uses.push_back(0);
        else
            lin.push_back(L);
        return n_ssa++;
    }
    uint32_t new_ssa64(int32_t lvl) {
        ssa_level64.push_back(lvl);
        ssa_row64.push_back((uint32_t)(ssa_level64.size() - 1));  // its own wmask row, unless a fresh mask row IS the wire's mask
        return (uint32_t)(ssa_level64.size() - 1);
    }
    void use(uint32_t ssa) {
        if (counting) uses[ssa]++;
    }
    void note(uint32_t lvl) {
        if (lvl > max_level) max_level = lvl;
        any = true;
    }
    void emit(const Gate& g, uint32_t lvl) {
        if (counting) return;
        gates.push_back(g);
        level.push_back(lvl);
        note(lvl);
    }
    void emit64(const Gate64& g0, uint32_t lvl) {
        Gate64 g = g0;
        if (g.op != G64_B2A) {  // (B2A's a is a GF(2) row)
            g.am = ssa_row64[g.a];
            g.bm = ssa_row64[g.b];
        }
        // Input / Random / Mul: the result's mask is the fresh mask itself (z64 share.rs: the new sharing) -- later
        // gates read that PRG row instead of a copy
        if (g.op == G64_INPUT || g.op == G64_RANDOM) ssa_row64[g.dst] = G64_MASK_ROW | g.m;
        if (g.op == G64_MUL) ssa_row64[g.dst] = G64_MASK_ROW | (g.m + 1);
        if (counting) return;
        gates64.push_back(g);
        level64.push_back(lvl);
        note(lvl);
    }
    static Lin base(uint32_t row) {
        Lin L;
        L.n = 1;
        L.b[0] = row;
        return L;
    }
    static void fill(Gate& g, const Lin& A, const Lin* B) {
        for (int i = 0; i < K; i++) {
            g.a[i] = i < A.n ? A.b[i] : ZERO_ROW;
            g.b[i] = (B && i < B->n) ? B->b[i] : ZERO_ROW;
        }
        g.op |= (uint32_t)A.n << 8 | (uint32_t)(B ? B->n : 0) << 12 | (uint32_t)A.c << 16 | (uint32_t)(B ? B->c : 0) << 17;
    }

    // materialise the XOR of up to 2K base rows (+ constant) into a computed row
    uint32_t materialise(const uint32_t* rows, int n, uint8_t c) {
        Gate g{};
        g.op = G_XORK;
        int32_t lvl = -1;
        for (int i = 0; i < n; i++) lvl = std::max(lvl, row_level(rows[i]));
        lvl += 1;
        const int na = std::min(n, K);
        for (int i = 0; i < K; i++) {
            g.a[i] = i < na ? rows[i] : ZERO_ROW;
            g.b[i] = (K + i < n) ? rows[K + i] : ZERO_ROW;
        }
        g.op |= (uint32_t)na << 8 | (uint32_t)(n - na) << 12 | (uint32_t)c << 16;
        g.dst = COMP | n_comp++;
        lvl_comp.push_back(lvl);
        emit(g, (uint32_t)lvl);
        out.info.gf2_linear++;
        return g.dst;
    }

    // streaming chunk: the chunk's last level writes a wire's final value (a linear form) into its carried row
    void write_back(uint32_t wire, const Lin& L, uint32_t lvl) {
        Gate g{};
        g.op = G_XORK;
        for (int i = 0; i < K; i++) {
            g.a[i] = i < L.n ? L.b[i] : ZERO_ROW;
            g.b[i] = ZERO_ROW;
        }
        g.op |= (uint32_t)L.n << 8 | (uint32_t)L.c << 16;
        g.dst = CARRY | wire;
        emit(g, lvl);
        out.info.gf2_linear++;
    }

    // ---- GF(2) primitives on SSA ids (used by plain ops and by the B2A expansion) ----
    uint32_t g_xor(uint32_t a, uint32_t b) {  // gf2/share.rs:220-238: Add and Sub are both XOR
        use(a);
        use(b);
        if (counting) return new_ssa(Lin());
        const Lin &A = lin[a], &B = lin[b];
        uint32_t rows[2 * K];
        int n = 0, i = 0, j = 0;  // symmetric difference of two sorted lists (x ^ x = 0)
        while (i < A.n || j < B.n) {
            if (j >= B.n || (i < A.n && A.b[i] < B.b[j]))
                rows[n++] = A.b[i++];
            else if (i >= A.n || B.b[j] < A.b[i])
                rows[n++] = B.b[j++];
            else {
                i++;
                j++;
            }
        }
        const uint8_t c = A.c ^ B.c;
        const uint32_t f = have_uses ? uses[n_ssa] : 1u;  // how often the result will be read
        // keep it symbolic when that costs no more row traffic than materialising it:
        // f readers x (n - 1) extra rows  vs  n reads + 1 write
        // (lazy_slack: no such limit for the circuits that are bound by their dependency levels, see compile_ops_seq)
        const bool lazy = n <= 1 || (n <= lazy_k && (uint64_t)f * (uint32_t)(n - 1) <= (uint64_t)n + lazy_slack);
        // a linear gate nobody reads has no effect on the proof (no transcript entry, no mask consumed): drop it
        // (13.5 % of the XOR gates of the random layered workload have fan-out zero)
        if (f == 0) return new_ssa(Lin());
        Lin L;
        if (lazy) {
            L.n = (uint8_t)n;
            L.c = c;
            for (int k = 0; k < n; k++) L.b[k] = rows[k];
        } else if (!balanced_sum(rows, n, c, L)) {
            L = base(materialise(rows, n, c));
        }
        return new_ssa(L);
    }
    // see `balance`; false: no split with strictly earlier rows exists (or the mode is off)
    bool balanced_sum(const uint32_t* rows, int n, uint8_t c, Lin& L) {
        if (balance <= 0 || n <= lazy_k || lazy_k < 2) return false;
        uint32_t srt[2 * K];
        for (int k = 0; k < n; k++) srt[k] = rows[k];
        std::stable_sort(srt, srt + n, [&](uint32_t x, uint32_t y) { return row_level(x) < row_level(y); });
        for (int keep = std::min(lazy_k - 1, balance); keep >= 1; keep--) {
            const int ne = n - keep;
            if (ne < 1 || row_level(srt[ne - 1]) >= row_level(srt[ne])) continue;
            uint32_t early[2 * K];
            for (int k = 0; k < ne; k++) early[k] = srt[k];
            std::sort(early, early + ne);
            uint32_t all[K];
            all[0] = ne == 1 ? early[0] : materialise(early, ne, 0);  // (the constant stays with the form)
            for (int k = 0; k < keep; k++) all[1 + k] = srt[ne + k];
            std::sort(all, all + keep + 1);
            L.n = (uint8_t)(keep + 1);
            L.c = c;
            for (int k = 0; k <= keep; k++) L.b[k] = all[k];
            return true;
        }
        return false;
    }
    uint32_t g_xorc(uint32_t a, uint32_t cbit) {  // AddConst / SubConst: free
        use(a);
        if (counting) return new_ssa(Lin());
        Lin L = lin[a];
        L.c ^= (uint8_t)cbit;
        return new_ssa(L);
    }
    uint32_t g_andc(uint32_t a, uint32_t cbit) {  // MulConst: identity or the zero wire
        use(a);
        if (counting) return new_ssa(Lin());
        return new_ssa(cbit ? lin[a] : Lin());
    }
    uint32_t g_const(uint32_t cbit) {
        Lin L;
        L.c = (uint8_t)cbit;
        return new_ssa(L);
    }
    uint32_t g_mul(uint32_t a, uint32_t b) {  // interpreter/single.rs:25-69
        use(a);
        use(b);
        const uint32_t m = (uint32_t)out.n_masks;
        out.n_masks += 2;
        const uint32_t eo = (uint32_t)out.n_on++, ep = (uint32_t)out.n_pre++, x = (uint32_t)out.n_rec++;
        if (counting) return new_ssa(Lin());
        Gate g{};
        g.op = G_MUL;
        fill(g, lin[a], &lin[b]);
        g.m = m;
        g.eo = eo;
        g.ep = ep;
        g.x = x;
        out.rec_rows.push_back(eo);
        const int32_t lvl = std::max(lin_level(lin[a]), lin_level(lin[b])) + 1;
        g.dst = m + 1;  // the output's mask IS the fresh mask lambda_new
        set_prg_level(m, lvl);
        set_prg_level(m + 1, lvl);
        emit(g, (uint32_t)lvl);
        out.info.gf2_muls++;
        return new_ssa(base(m + 1));
    }
    uint32_t g_random() {
        const uint32_t m = (uint32_t)out.n_masks++;
        if (counting) return new_ssa(Lin());
        Gate g{};
        g.op = G_RANDOM;
        fill(g, Lin(), nullptr);
        g.m = m;
        g.dst = m;
        set_prg_level(m, 0);
        emit(g, 0);
        out.info.gf2_linear++;
        out.n_random_or_recon++;
        return new_ssa(base(m));
    }
    uint32_t g_input() {
        const uint32_t m = (uint32_t)out.n_masks++, eo = (uint32_t)out.n_on++, x = (uint32_t)out.n_in++;
        if (counting) return new_ssa(Lin());
        Gate g{};
        g.op = G_INPUT;
        fill(g, Lin(), nullptr);
        g.m = m;
        g.dst = m;
        g.eo = eo;
        g.x = x;
        out.in_rows.push_back(eo);
        set_prg_level(m, 0);
        emit(g, 0);
        out.info.gf2_inputs++;
        return new_ssa(base(m));
    }
    // AssertZero (recon = false) or B2A's recorded reconstruction (recon = true, returns the result wire)
    uint32_t g_reveal(uint32_t a, bool recon) {
        use(a);
        const uint32_t eo = (uint32_t)out.n_on++, x = (uint32_t)out.n_rec++;
        if (counting) return recon ? new_ssa(Lin()) : 0;
        Gate g{};
        g.op = recon ? G_RECON : G_ASSERT;
        fill(g, lin[a], nullptr);
        g.eo = eo;
        g.x = x;
        out.rec_rows.push_back(eo);
        const int32_t lvl = lin_level(lin[a]) + 1;
        uint32_t res = 0;
        if (recon) {
            g.dst = COMP | n_comp++;  // {mask 0, corr = revealed value}
            out.n_random_or_recon++;
            lvl_comp.push_back(lvl);
            res = new_ssa(base(g.dst));
        }
        emit(g, (uint32_t)lvl);
        out.info.gf2_asserts++;
        return res;
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

static int run_pass(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Builder& b, const ChunkStart* chunk) {
    Compiled& out = b.out;
    b.cur.assign(gf2_wires, 0);
    b.cur64.assign(z64_wires, 0);
    if (chunk) {
        // every wire starts as its carried row / slot (SSA 1 + w), available before level 0
        for (size_t w = 0; w < gf2_wires; w++) b.cur[w] = b.new_ssa(Builder::base(CARRY | (uint32_t)w));
        for (size_t w = 0; w < z64_wires; w++) b.cur64[w] = b.new_ssa64(-1);
        out.n_masks = chunk->mask_phase;
        out.n_masks64 = chunk->mask64_phase;
        out.n_on = chunk->on0;
        out.n_pre = chunk->pre0;
        out.on_words64 = chunk->on_words64_0;
        out.pre_words64 = chunk->pre_words64_0;
    }
    rv_circuit_info& info = out.info;
    const uint64_t LIM = std::numeric_limits<uint32_t>::max() - 512;

    for (size_t i = 0; i < n_ops; i++) {
        const rv_op& op = ops[i];
        if (op.reserved != 0) return RV_E_BAD_OP;
        switch (op.domain) {
        case RV_DOM_SIZEHINT:  // interpreter/combine.rs:122-129
            // (a streaming chunk's wire store was sized when the stream began: growing it mid-stream is not supported)
            if (chunk && (op.b > b.cur.size() || op.a > b.cur64.size())) return RV_E_UNSUPPORTED;
            if (op.b > b.cur.size()) b.cur.resize(op.b, 0);
            if (op.a > b.cur64.size()) b.cur64.resize(op.a, 0);
            break;
        case RV_DOM_GF2: {
            const size_t nw = b.cur.size();
            const uint32_t cbit = (uint32_t)(op.imm & 1);
            switch (op.opcode) {
            case RV_OP_INPUT:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_input();
                break;
            case RV_OP_RANDOM:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_random();
                break;
            case RV_OP_CONST:
                if (op.dst >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_const(cbit);
                break;
            case RV_OP_ADD:
            case RV_OP_SUB:
                if (op.dst >= nw || op.a >= nw || op.b >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_xor(b.cur[op.a], b.cur[op.b]);
                break;
            case RV_OP_MUL:
                if (op.dst >= nw || op.a >= nw || op.b >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_mul(b.cur[op.a], b.cur[op.b]);
                break;
            case RV_OP_ADDCONST:
            case RV_OP_SUBCONST:
                if (op.dst >= nw || op.a >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_xorc(b.cur[op.a], cbit);
                break;
            case RV_OP_MULCONST:
                if (op.dst >= nw || op.a >= nw) return RV_E_WIRE_OOB;
                b.cur[op.dst] = b.g_andc(b.cur[op.a], cbit);
                break;
            case RV_OP_ASSERTZERO:
                if (op.a >= nw) return RV_E_WIRE_OOB;
                b.g_reveal(b.cur[op.a], false);
                break;
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
            // 5. 64 recorded reconstructions; their results occupy 64 CONSECUTIVE computed rows
            int32_t lvl_max = 0;
            const uint32_t first_out = COMP | b.n_comp;
            for (int k = 0; k < 64; k++) {
                b.g_reveal(res[k], true);
                if (!b.counting) lvl_max = std::max(lvl_max, b.lvl_comp.back());
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
        if (out.n_masks > LIM || b.n_ssa > LIM || b.n_comp > LIM / 2 || out.n_masks64 > LIM || b.ssa_level64.size() > LIM ||
            out.n_on > LIM)
            return RV_E_UNSUPPORTED;
    }
    if (chunk) {
        // Write-back level.  A final form that still reads carried rows is materialised first: its write-back would
        // otherwise race with the write-back of the rows it reads (two wires swapped by the chunk).
        std::vector<std::pair<uint32_t, Lin>> wb;
        for (size_t w = 0; w < gf2_wires; w++) {
            const uint32_t ssa = b.cur[w];
            if (ssa == 1 + (uint32_t)w) continue;  // never written in this chunk
            b.use(ssa);                            // (pass 1: a wire that is live out is not dead)
            if (b.counting) continue;
            Lin L = b.lin[ssa];
            bool reads_carry = false;
            for (int i = 0; i < L.n; i++) reads_carry |= (L.b[i] & CARRY) != 0;
            if (reads_carry) L = Builder::base(b.materialise(L.b, L.n, L.c));
            wb.emplace_back((uint32_t)w, L);
        }
        if (!b.counting) {
            const uint32_t last = b.any ? b.max_level + 1 : 0;
            for (const auto& e : wb) b.write_back(e.first, e.second, last);
            for (size_t w = 0; w < z64_wires; w++) {
                if (b.cur64[w] == 1 + (uint32_t)w) continue;
                Gate64 g{};
                g.op = G64_ADDC;  // a copy: dst slot = the wire's carried slot
                g.dst = 1 + (uint32_t)w;
                g.a = b.cur64[w];
                b.emit64(g, last);
                info.z64_linear++;
            }
        }
    }
    return RV_OK;
}

void count_masks(const rv_op* ops, size_t n_ops, uint64_t* gf2_masks, uint64_t* z64_masks) {
    uint64_t m2 = 0, m64 = 0;
    for (size_t i = 0; i < n_ops; i++) {
        const rv_op& op = ops[i];
        const bool one = op.opcode == RV_OP_INPUT || op.opcode == RV_OP_RANDOM;
        if (op.domain == RV_DOM_GF2)
            m2 += one ? 1 : (op.opcode == RV_OP_MUL ? 2 : 0);
        else if (op.domain == RV_DOM_Z64)
            m64 += one ? 1 : (op.opcode == RV_OP_MUL ? 2 : 0);
        else if (op.domain == RV_DOM_B2A) {  // run_pass: 64 g_random + 63 g_mul, one Z64 mask
            m2 += 64 + 63 * 2;
            m64 += 1;
        }
    }
    *gf2_masks = m2;
    *z64_masks = m64;
}

// transcript events the ops make (per repetition): GF(2) online rows = inputs + reconstructions (Mul, AssertZero; B2A: 63 + 64),
// preprocessing rows = Mul (B2A: 63); Z64 online words = inputs + 8 per Mul / AssertZero, preprocessing words = Mul + B2A
void count_events(const rv_op* ops, size_t n_ops, StreamEvents* ev) {
    StreamEvents e;
    for (size_t i = 0; i < n_ops; i++) {
        const rv_op& op = ops[i];
        if (op.domain == RV_DOM_GF2) {
            if (op.opcode == RV_OP_INPUT) e.in2++;
            else if (op.opcode == RV_OP_MUL) e.rec2++, e.pre2++;
            else if (op.opcode == RV_OP_ASSERTZERO) e.rec2++;
        } else if (op.domain == RV_DOM_Z64) {
            if (op.opcode == RV_OP_INPUT) e.on64 += 1;
            else if (op.opcode == RV_OP_MUL) e.on64 += 8, e.pre64 += 1;
            else if (op.opcode == RV_OP_ASSERTZERO) e.on64 += 8;
        } else if (op.domain == RV_DOM_B2A) {
            e.rec2 += 63 + 64, e.pre2 += 63, e.pre64 += 1;
        }
    }
    *ev = e;
}

void relocate_chunk(Compiled& cc, uint64_t on0, uint64_t pre0, uint64_t on_words64_0, uint64_t pre_words64_0) {
    if (on0 || pre0) {
        for (Gate& g : cc.gates) {  // (eo / ep of gates without a transcript row are never read)
            g.eo += (uint32_t)on0;
            g.ep += (uint32_t)pre0;
        }
        for (uint32_t& r : cc.rec_rows) r += (uint32_t)on0;
        for (uint32_t& r : cc.in_rows) r += (uint32_t)on0;
        for (uint32_t& r : cc.level_done_on) r += (uint32_t)on0;
        cc.n_on += on0;
        cc.n_pre += pre0;
    }
    if (on_words64_0 || pre_words64_0) {
        for (Gate64& g : cc.gates64) {
            g.eo += on_words64_0;
            g.ep += pre_words64_0;
        }
        for (uint64_t& o : cc.rec_offs64) o += on_words64_0;
        for (uint64_t& o : cc.in_offs64) o += on_words64_0;
        cc.on_words64 += on_words64_0;
        cc.pre_words64 += pre_words64_0;
    }
}

int compile_ops(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out, const ChunkStart* chunk, int force_lazy_k) {
    const size_t par_min = getenv("RV_COMPILE_PAR_MIN") ? (size_t)atoll(getenv("RV_COMPILE_PAR_MIN")) : 200000;
    const bool seq = getenv("RV_COMPILE_SEQ") && atoi(getenv("RV_COMPILE_SEQ")) != 0;
    if (!chunk && !seq && n_ops >= par_min) {
        const int nt = compile_threads();
        if (nt > 1) {
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = compile_ops_par(ops, n_ops, z64_wires, gf2_wires, out, force_lazy_k, nt);
            if (getenv("RV_COMPILE_STATS"))
                fprintf(stderr, "[rv compile] parallel compiler (%d threads) returned %d after %.3f s\n", nt, rc,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            if (rc != RV_COMPILE_FALLBACK) return rc;
        }
    }
    return compile_ops_seq(ops, n_ops, z64_wires, gf2_wires, out, chunk, force_lazy_k);
}

int compile_ops_seq(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out, const ChunkStart* chunk, int force_lazy_k) {
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (getenv("RV_COMPILE_STATS"))
            fprintf(stderr, "[rv compile] %-28s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    std::vector<uint32_t> uses;
    // A streaming chunk is compiled once with every XOR materialised (lazy_k = 1 below): the read counts would only serve to drop the
    // XOR gates nobody reads (13.5 % of them on the layered workload -- the GPU runs them in passing), and the counting pass is
    // 3 of a piece's 18 - 22 ms on a worker thread, where the streaming prover's first pass is bound
    const bool count_reads = !chunk || force_lazy_k || getenv("RV_LAZY_K");
    if (count_reads) {
        // pass 1: SSA numbering + read counts (the materialisation rule needs each wire's fan-out)
        Compiled scratch;
        Builder b1(scratch, true, uses);
        int rc = run_pass(ops, n_ops, z64_wires, gf2_wires, b1, chunk);
        if (rc) return rc;
    }
    lap("pass 1 (SSA, read counts) done");
    // Wide circuits run fastest with every XOR materialised (exact two-row gates, HBM-bound); deep,
    // narrow ones (ripple-carry adders, hash rounds) are bound by the number of dependency levels, and
    // keeping XORs of up to RV_LIN_K rows symbolic shortens the chains (SHA-256: 5 386 -> 4 291 levels,
    // 11.2 -> 6.9 ms per proof).  Decide from the K = 1 compile; RV_LAZY_K overrides.
    int lazy_k = 1;
    bool forced = false;
    if (force_lazy_k) {
        lazy_k = std::min(std::max(force_lazy_k, 1), K);
        forced = true;
    } else if (const char* e = getenv("RV_LAZY_K")) {
        lazy_k = std::min(std::max(atoi(e), 1), K);
        forced = true;
    }
    Builder* bp = nullptr;
    // a deep narrow circuit runs as steps of up to 64 gates of one level (LDS runs, ldsrun.h): what counts is their number
    auto steps64 = [](const Builder& b) {
        std::vector<uint32_t> per((size_t)b.max_level + 1, 0);
        for (uint32_t l : b.level) per[l]++;
        uint64_t steps = 0;
        for (uint32_t g : per) steps += (g + 63) / 64;
        return steps;
    };
    static const int balance_env = getenv("RV_LAZY_BALANCE") ? atoi(getenv("RV_LAZY_BALANCE")) : -1;
    int balance = 0, best_balance = 0, phase = 0;  // phase 0: K = 1; 1: K rows, balance 0 .. K - 1 tried in turn; 2: the best one again
    uint64_t best_steps = ~0ull;
    for (;;) {
        out = Compiled();
        delete bp;
        bp = new Builder(out, false, uses, count_reads ? 0 : n_ops + gf2_wires + z64_wires + 64);
        bp->have_uses = count_reads;
        bp->lazy_k = lazy_k;
        bp->lazy_slack = lazy_slack_for(lazy_k, forced);
        bp->balance = balance;
        int rc = run_pass(ops, n_ops, z64_wires, gf2_wires, *bp, chunk);
        if (rc) {
            delete bp;
            return rc;
        }
        if (phase == 0) {
            const uint64_t n_levels_now = bp->any ? (uint64_t)bp->max_level + 1 : 0;
            const bool deep_narrow = n_levels_now && lazy_forms_pay(n_levels_now, bp->gates.size());
            if (forced || lazy_k != 1 || !deep_narrow) break;
            if (chunk) break;  // (a chunk's counters were seeded from ChunkStart; one attempt)
            lazy_k = K;
            phase = 1;
            if (balance_env >= 0) {  // (an experiment knob: no search)
                balance = std::min(balance_env, K - 1);
                phase = 2;
            }
            continue;
        }
        if (phase == 2) break;
        // phase 1: how many steps does this variant take?  (the search costs three more passes: small circuits only)
        const uint64_t st = steps64(*bp);
        if (st < best_steps) best_steps = st, best_balance = balance;
        if (balance + 1 <= K - 1 && bp->gates.size() < 1000000) {
            balance++;
            continue;
        }
        if (best_balance == balance) break;  // the variant just built is the one to keep
        balance = best_balance;
        phase = 2;
    }
    lap("pass 2 (gates) done");
    Builder& b = *bp;
    struct Guard {
        Builder* p;
        ~Guard() { delete p; }
    } guard{bp};
    rv_circuit_info& info = out.info;
    info.n_ops = n_ops;
    const uint64_t LIM = std::numeric_limits<uint32_t>::max() - 512;
    const uint32_t n_levels = b.any ? b.max_level + 1 : 0;
    // GF(2) gates: one stable counting sort by (level, class) -- gates of one level are independent, so grouping them by
    // class inside the level is free; see LevelRange for the classes
    out.level_range.assign(n_levels, LevelRange{});
    {
        auto cls = [](const Gate& g) -> uint32_t {
            const uint32_t op = g_op(g);
            if (op == G_MUL) return (g_na(g) == 1 && g_nb(g) == 1) ? 0u : 1u;
            if (op == G_XORK) return (g_na(g) == 2 && g_nb(g) == 0) ? 2u : 3u;
            return 4u;
        };
        const size_t n = b.gates.size();
        std::vector<uint32_t> pos((size_t)n_levels * 5 + 1, 0);
        std::vector<uint8_t> kc(n);
        for (size_t i = 0; i < n; i++) {
            kc[i] = (uint8_t)cls(b.gates[i]);
            pos[(size_t)b.level[i] * 5 + kc[i] + 1]++;
        }
        for (size_t k = 0; k < (size_t)n_levels * 5; k++) pos[k + 1] += pos[k];
        out.level_start.assign(n_levels + 1, 0);
        for (uint32_t l = 0; l < n_levels; l++) {
            const uint32_t* e = &pos[(size_t)l * 5];
            out.level_start[l] = e[0];
            out.level_range[l] = LevelRange{e[0], e[1], e[2], e[3], e[4], e[5]};
        }
        out.level_start[n_levels] = (uint32_t)n;
        out.gates.resize(n);
        for (size_t i = 0; i < n; i++) out.gates[pos[(size_t)b.level[i] * 5 + kc[i]]++] = b.gates[i];
    }
    sort_by_level(b.gates64, b.level64, n_levels, out.gates64, out.level_start64);
    lap("sorted by level and class");
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
                const uint32_t op = g_op(g);
                if (op == G_MUL)
                    last = g.m + 1;
                else if (op == G_INPUT || op == G_RANDOM)
                    last = g.m;
                else
                    uses = false;
                if (uses) need = std::max(need, last / 128 + 1);
                if (op == G_MUL || op == G_INPUT || op == G_ASSERT || op == G_RECON) row_level[g.eo] = l;
            }
            out.level_need_blocks[l] = need;
        }
        // PRG rows are also read as operand bases by later levels, but such a row was consumed as a
        // fresh mask by an EARLIER level's gate, so the prefix maximum covers it
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
    out.n_ssa = b.n_ssa;
    out.n_ssa64 = b.ssa_level64.size();
    // resolve share rows: PRG masks first (padded to whole AES blocks), computed rows after
    out.n_masks_pad = (out.n_masks + 127) / 128 * 128;
    out.row_prg_base = chunk ? gf2_wires : 0;
    out.zero_row = out.row_prg_base + out.n_masks_pad;
    out.n_rows = out.row_prg_base + out.n_masks_pad + b.n_comp;
    if (out.n_rows > LIM) return RV_E_UNSUPPORTED;
    // the mask kernels take CTR block indices below 2^24 (first-round constants, internal.h); more would not fit HBM anyway
    if (out.n_masks_pad / 128 > RV_MAX_CTR_BLOCKS || (out.n_masks64 + 1) / 2 > RV_MAX_CTR_BLOCKS) return RV_E_UNSUPPORTED;
    const uint32_t base = (uint32_t)out.row_prg_base;
    auto fix = [&](uint32_t& r) {
        if (r & COMP)
            r = (uint32_t)(base + out.n_masks_pad + (r & ~COMP));
        else if (r & CARRY)
            r &= ~CARRY;
        else
            r += base;  // a PRG mask row
    };
    for (Gate& g : out.gates) {
        fix(g.dst);
        for (int i = 0; i < K; i++) {
            fix(g.a[i]);
            fix(g.b[i]);
        }
        g.m += base;  // the kernels address a gate's fresh masks as share rows m, m + 1
    }
    for (Gate64& g : out.gates64)
        if (g.op == G64_B2A) {
            fix(g.a);
            g.m2 += base;
        }
    info.gf2_masks = out.n_masks;
    info.z64_masks = out.n_masks64;
    info.levels = n_levels;
    for (const Gate& g : out.gates) {  // what the interpreter moves per repetition quad (bench.py prices its launches with these)
        const uint32_t op = g_op(g);
        if (op == G_MUL || op == G_XORK) info.gf2_operand_rows += g_na(g) + g_nb(g);
        if (op == G_ASSERT || op == G_RECON) info.gf2_operand_rows += g_na(g);
        if (op == G_XORK || op == G_RECON) info.gf2_rows_written++;
    }
    lap("tables and row fix-up done");
    if (getenv("RV_COMPILE_STATS")) {  // interpreter HBM traffic model per 4-repetition quad column (x NQ x 4 B per row)
        uint64_t rd = 0, wr = 0, crd = 0, n_mul = 0, n_xor = 0, n_mul11 = 0;
        for (const Gate& g : out.gates) {
            const uint32_t op = g_op(g), n = g_na(g) + g_nb(g);
            if (op == G_MUL) {
                rd += n + 2, crd += n, wr += 1, n_mul++;
                n_mul11 += (g_na(g) == 1 && g_nb(g) == 1);
            } else if (op == G_XORK) {
                rd += n, crd += n, wr += 1, n_xor++;
            }
        }
        uint32_t hist[7] = {0, 0, 0, 0, 0, 0, 0};
        for (uint32_t l = 0; l < n_levels; l++) {
            const uint32_t w = out.level_start[l + 1] - out.level_start[l];
            hist[w <= 16 ? 0 : w <= 32 ? 1 : w <= 64 ? 2 : w <= 128 ? 3 : w <= 256 ? 4 : w <= 1024 ? 5 : 6]++;
        }
        fprintf(stderr, "[rv compile] level widths: <=16:%u <=32:%u <=64:%u <=128:%u <=256:%u <=1024:%u more:%u\n", hist[0], hist[1], hist[2],
                hist[3], hist[4], hist[5], hist[6]);
        {
            // cipher blocks (128 GF(2) mask rows = 64 Mul gates in program order) a level's Mul gates touch, summed over the levels,
            // against the blocks there are: what a mask generator inside the level launches would have to run
            uint64_t touched = 0, odd = 0;
            std::vector<uint64_t> blk;
            for (uint32_t l = 0; l < n_levels; l++) {
                blk.clear();
                for (uint64_t i = out.level_start[l]; i < out.level_start[l + 1]; i++)
                    if (g_op(out.gates[i]) == G_MUL) {
                        const uint64_t m = out.gates[i].m - out.row_prg_base;
                        blk.push_back(m >> 7);
                        odd += m & 1;
                    }
                std::sort(blk.begin(), blk.end());
                touched += (uint64_t)(std::unique(blk.begin(), blk.end()) - blk.begin());
            }
            fprintf(stderr, "[rv compile] Mul cipher blocks: %llu touched level by level, %llu dense (mul / 64), %llu Mul gates with an odd mask index\n",
                    (unsigned long long)touched, (unsigned long long)((n_mul + 63) / 64), (unsigned long long)odd);
        }
        fprintf(stderr, "[rv compile] lazy_k=%d levels=%u mul=%llu (one-base %llu) xork=%llu row_reads=%llu row_writes=%llu corr_reads=%llu\n",
                b.lazy_k, n_levels, (unsigned long long)n_mul, (unsigned long long)n_mul11, (unsigned long long)n_xor,
                (unsigned long long)rd, (unsigned long long)wr, (unsigned long long)crd);
    }
    return RV_OK;
}

}  // namespace rv
