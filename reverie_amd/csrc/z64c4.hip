// The Z64 prover's dependency level with the LANE-DISTRIBUTED cipher inside (round 5).  EXPERIMENT BUILDS ONLY (make
// EXTRA=-DRV_EXPERIMENTS, RV_Z64_C4=1): byte-identical proofs (tests/test_gpu_z64_fused.py) and, on the 10^6-Mul circuit, a TIE with
// k_z64_fused (aes.hip), which stays the library's kernel: 35.7 ms of level launches against 35.8 - 36.1, the hash phase behind it 6.25
// against 6.04 (the chip comes out of this kernel at a lower clock), 48.7 - 48.8 ms per proof either way.  What this file found and
// k_z64_fused took over: the transcript stores (z64_dev.h: whole 64-byte segments per four lanes; 44.3 -> 36.9 ms here, 38.4 -> 36.2
// there); its own preprocessing words leave in runs of eight through LDS (a wavefront takes consecutive gates: 37.9 -> 35.7).  Both
// kernels sit where the cipher's instruction stream (27 - 31 ms alone) and 120 GB of row and transcript traffic per proof (3.4 TB/s of
// scattered 4 KiB - 16 KiB pieces, reads = the algorithmic 65 GB exactly) meet; measured here and not kept: nontemporal transcript
// stores (no difference), the linear gate's rows requested with the Mul's (167 registers, 12 wavefronts: 39.7), 12 wavefronts (37.1:
// what a co-resident hash kernel would need), 8 wavefronts (41.0), a priority that falls with the trips done (42.9 against 44.9 before
// the store fix).
//
// Replaces, for one dependency level (all under /root/reference/src/): generator/share.rs:54-65 + algebra/z64/domain.rs:64-83
// (the two fresh masks of every Mul: one AES-128-CTR block per (repetition, player) stream), interpreter/single.rs:25-157
// instantiated at Z64 (op_mul, the linear ops, Input, AssertZero) and transcript/prover.rs:181-232's records.
//
// A Z64 Mul draws exactly one cipher block per stream, so the wavefront that runs the cipher for counter m / 2 ends with
// lambda_ab and lambda_new of ITS gate in registers: lambda_ab never reaches HBM.  k_z64_fused does this with a whole bitsliced state
// per lane (256 registers): nothing is left for row traffic in flight during the cipher, the rows need a piece-major layout to
// coalesce, and the SIMD idles through the row phases (38.5 ms per 10^6 Mul against a cipher floor of 27).  Here:
//   * cipher: a quad of lanes per state (aes_col4_dev.h), a wavefront = ONE gate x 16 quad words; 32 + 48 registers;
//   * the gate's operand pieces (2 x 64 bytes per lane) are REQUESTED BEFORE the cipher and used ~4 000 instructions later;
//   * after the last round lane c of a quad holds bit planes of keystream bytes 4c .. 4c+3 for the quad's 32 slots; ONE 32 x 32 bit
//     transpose turns them into the 32-bit quarter of every slot's block, and a 4 x 4 exchange of 8-register groups inside the quad
//     (two butterfly stages of DPP moves and bit selects) gives lane r all four quarters of repetition r's eight players:
//     lambda_ab = bytes 0..7, lambda_new = bytes 8..15 (z64/batch.rs:26-29: little-endian u64 pairs);
//   * lane = one repetition, a wavefront = the 64 repetitions of its 16 quad words = a 4 KiB block of every share row, kept
//     piece-major inside the block (Z4_PIECE) so that every row instruction is 1 KiB contiguous across the wavefront;
//   * the sum over a repetition's players is eight lane-local adds (z64/domain.rs:53-61).
#include <stdlib.h>

#include <algorithm>

#include "aes_col4_dev.h"
#include "z64_dev.h"
#include "internal.h"

namespace rv {

// 32x32 bit-matrix transpose (Hacker's Delight 7-3), registers only
__device__ __forceinline__ void z4_transpose32(uint32_t* A) {
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = (A[k] ^ (A[k + j] >> j)) & m;
            A[k] ^= t;
            A[k + j] ^= (t << j);
        }
        m ^= (m << (j >> 1));
    }
}
__device__ __forceinline__ void z4_ld16(const uint64_t* p, uint64_t& a, uint64_t& b) {
    const ulonglong2 v = *(const ulonglong2*)p;
    a = v.x;
    b = v.y;
}
__device__ __forceinline__ void z4_st16(uint64_t* p, uint64_t a, uint64_t b) { *(ulonglong2*)p = make_ulonglong2(a, b); }
__device__ __forceinline__ const uint64_t* z4_row(const Z64FParams& p, uint32_t ref, uint64_t S) {
    return (ref & G64_MASK_ROW) ? p.masks + (size_t)(ref & ~G64_MASK_ROW) * S : p.wmask + (size_t)ref * S;
}
// one repetition's eight transcript words (64 bytes; the stream is only 8-byte aligned)
__device__ __forceinline__ void z4_store_on(uint64_t* op, const uint64_t* w) {
    if (((uintptr_t)op & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) z4_st16(op + 2 * i, w[2 * i], w[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) op[i] = w[i];
    }
}
// Row layout of this path: inside a workgroup's block of 64 repetitions (4 KiB) the 16-byte piece i (players 2i, 2i + 1) of the block's
// repetition L sits at u64 offset i * 128 + 2 * L -- piece-major, so that each of a lane's four load instructions reads 1 KiB that is
// contiguous across the wavefront (a lane's 64 bytes in the natural order are 64 bytes apart from its neighbour's: four instructions
// that each touch all 64 lines of the block).  Every row this path reads it also wrote, except the Input gates' mask rows (z4_oth).
constexpr uint32_t Z4_PIECE = 128;

// Mul (interpreter/single.rs:25-69 with the prover's transcript, prover.rs:181-219): lambda_ab, lambda_new = the gate's cipher block
__device__ __forceinline__ uint64_t z4_mul(const Gate64& g, const Z64FParams& p, const uint4* rkl, uint32_t c, uint32_t rep, uint32_t zo, bool writer) {
    const uint64_t S = (uint64_t)p.NQ * 32;
    // the operand pieces first: they land while the cipher runs
    const uint64_t* ap = z4_row(p, g.am, S) + zo;
    const uint64_t* bp = z4_row(p, g.bm, S) + zo;
    uint64_t lx[8], ly[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        z4_ld16(ap + Z4_PIECE * i, lx[2 * i], lx[2 * i + 1]);
        z4_ld16(bp + Z4_PIECE * i, ly[2 * i], ly[2 * i + 1]);
    }
    const uint64_t va = p.v[g.a], vb = p.v[g.b];
    uint32_t A[32];
    {
        uint32_t s[32];
        c4_rounds_0_1((uint32_t)(p.first_block + (g.m >> 1)), c, rkl, s);
#pragma unroll 1
        for (int r = 2; r < 10; r++) c4_round(s, rkl + r * 8 * 64);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            c4_sbox8(s[8 * r + 7], s[8 * r + 6], s[8 * r + 5], s[8 * r + 4], s[8 * r + 3], s[8 * r + 2], s[8 * r + 1], s[8 * r + 0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // plane 8*r + k = bit k of keystream byte 4*c + r = bit 8*r + k of the column's little-endian 32-bit word
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 kv = rkl[(10 * 8 + k) * 64];
            A[31 - k] = s[k] ^ kv.x;
            A[31 - (8 + k)] = s[8 + k] ^ kv.y;
            A[31 - (16 + k)] = s[16 + k] ^ kv.z;
            A[31 - (24 + k)] = s[24 + k] ^ kv.w;
        }
    }
    z4_transpose32(A);  // A[slot] = the slot's 32-bit quarter (slot = 8 * repetition-in-quad + player at bit 31 - slot before)
    // 4 x 4 exchange of the 8-register groups inside the quad: lane r ends with group r of every lane (= column) of the quad
    uint32_t co = c;
    asm volatile("" : "+v"(co));  // (selects on c itself compile to branches)
    const uint32_t mh = (co & 2u) ? ~0u : 0u, ml = (co & 1u) ? ~0u : 0u;
    uint32_t K[2][8], Y[2][8];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t g_lo = A[8 * j + i], g_hi = A[8 * (2 + j) + i];  // groups j and 2 + j
            K[j][i] = z4_sel(mh, g_hi, g_lo);                               // the half this lane's repetition is in stays
            Y[j][i] = z4_dpp<0x4E>(z4_sel(mh, g_lo, g_hi));                 // the other half goes to lane c ^ 2, its half comes back
        }
    uint64_t lab[8], lnw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t own = z4_sel(ml, K[1][i], K[0][i]), rk_ = z4_dpp<0xB1>(z4_sel(ml, K[0][i], K[1][i]));  // column c, column c ^ 1
        const uint32_t yk = z4_sel(ml, Y[1][i], Y[0][i]), ry = z4_dpp<0xB1>(z4_sel(ml, Y[0][i], Y[1][i]));    // column c ^ 2, column c ^ 3
        const uint32_t p_lo = z4_sel(ml, rk_, own), p_hi = z4_sel(ml, own, rk_);  // the pair of columns this lane is in, low column first
        const uint32_t q_lo = z4_sel(ml, ry, yk), q_hi = z4_sel(ml, yk, ry);      // the other pair
        lab[i] = ((uint64_t)z4_sel(mh, q_hi, p_hi) << 32) | z4_sel(mh, q_lo, p_lo);  // columns 0, 1
        lnw[i] = ((uint64_t)z4_sel(mh, p_hi, q_hi) << 32) | z4_sel(mh, p_lo, q_lo);  // columns 2, 3
    }
    uint64_t a = 0, b = 0, cs = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a += lx[i];
        b += ly[i];
        cs += lab[i];
    }
    // corr = value - reconstruct(mask) (prover.rs:181-199 with the cleartext value known)
    const uint64_t cx = va - a, cy = vb - b;
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = ly[i] * cx + lx[i] * cy + lab[i] - lnw[i];
    uint64_t* lnp = p.masks + (size_t)(g.m + 1) * S + zo;
#pragma unroll
    for (int i = 0; i < 4; i++) z4_st16(lnp + Z4_PIECE * i, lnw[2 * i], lnw[2 * i + 1]);
    z4_store_on_quad(p.on + (size_t)(rep & ~3u) * p.on_words + g.eo, p.on_words, w, co, mh, ml);
    if (writer) p.v[g.dst] = va * vb;
    return a * b - cs;  // the preprocessing transcript's word of (this gate, this repetition): k_z64_c4 collects a run of them
}

// Add / Sub / AddConst / SubConst / MulConst (z64/share.rs:110-136 player by player) and the value
__device__ __forceinline__ void z4_lin(const Gate64& g, const Z64FParams& p, uint32_t zo, bool writer) {
    const uint64_t S = (uint64_t)p.NQ * 32;
    const uint64_t* ap = z4_row(p, g.am, S) + zo;
    uint64_t* dp = p.wmask + (size_t)g.dst * S + zo;
    const uint64_t va = p.v[g.a];
    uint64_t x[8];
#pragma unroll
    for (int i = 0; i < 4; i++) z4_ld16(ap + Z4_PIECE * i, x[2 * i], x[2 * i + 1]);
    if (g.op == G64_ADD || g.op == G64_SUB) {
        const uint64_t* bp = z4_row(p, g.bm, S) + zo;
        const uint64_t vb = p.v[g.b];
        const bool sub = g.op == G64_SUB;
        uint64_t y[8];
#pragma unroll
        for (int i = 0; i < 4; i++) z4_ld16(bp + Z4_PIECE * i, y[2 * i], y[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 4; i++)
            z4_st16(dp + Z4_PIECE * i, sub ? x[2 * i] - y[2 * i] : x[2 * i] + y[2 * i], sub ? x[2 * i + 1] - y[2 * i + 1] : x[2 * i + 1] + y[2 * i + 1]);
        if (writer) p.v[g.dst] = sub ? va - vb : va + vb;
    } else {
        const uint64_t f = g.op == G64_MULC ? g.imm : 1;
#pragma unroll
        for (int i = 0; i < 4; i++) z4_st16(dp + Z4_PIECE * i, x[2 * i] * f, x[2 * i + 1] * f);
        if (writer) p.v[g.dst] = g.op == G64_MULC ? va * g.imm : (g.op == G64_ADDC ? va + g.imm : va - g.imm);
    }
}

// Input (masked input = witness - reconstruct(fresh mask), prover.rs:181-199), AssertZero (the wire's mask shares go into the online
// transcript; the VALUE must be zero, prover.rs:221-228), Const
__device__ __forceinline__ void z4_oth(const Gate64& g, const Z64FParams& p, uint32_t rep, uint32_t zo, bool writer) {
    const uint64_t S = (uint64_t)p.NQ * 32;
    if (g.op == G64_INPUT) {
        // the mask row comes from k_aes_z64_masks in the natural order [rep][player]; it is rewritten in place in this path's layout.
        // The wavefront's 64 lanes read and write the SAME 4 KiB block: every lane has its 64 bytes before any lane overwrites them
        uint64_t* row = p.masks + (size_t)g.m * S;
        const uint64_t* lp = row + (size_t)rep * 8;
        const uint64_t wv = p.wit[g.x];
        uint64_t l[8], a = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) z4_ld16(lp + 2 * i, l[2 * i], l[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 8; i++) a += l[i];
        p.on[(size_t)rep * p.on_words + g.eo] = wv - a;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; i++) z4_st16(row + zo + Z4_PIECE * i, l[2 * i], l[2 * i + 1]);
        if (writer) p.v[g.dst] = wv;
    } else if (g.op == G64_ASSERT) {
        const uint64_t* ap = z4_row(p, g.am, S) + zo;
        uint64_t l[8];
#pragma unroll
        for (int i = 0; i < 4; i++) z4_ld16(ap + Z4_PIECE * i, l[2 * i], l[2 * i + 1]);
        z4_store_on(p.on + (size_t)rep * p.on_words + g.eo, l);
        if (writer && p.v[g.a] != 0) atomicOr(p.err, RV_E_WITNESS_INVALID);
    } else if (g.op == G64_CONST) {
        uint64_t* dp = p.wmask + (size_t)g.dst * S + zo;
#pragma unroll
        for (int i = 0; i < 4; i++) z4_st16(dp + Z4_PIECE * i, 0, 0);
        if (writer) p.v[g.dst] = g.imm;
    }
}

// A workgroup = 16 quad words (64 repetitions: their key image in LDS) x a share of the level's gates; a wavefront = ONE gate per
// step.  Mul steps and linear steps alternate inside a wavefront, so that the level's row traffic runs beside other wavefronts' ciphers.
// (the linear gate's operand rows requested together with the Mul's, its result stored after the cipher -- no wavefront ever waits for a
// row it has just asked for -- needs 167 registers = 12 wavefronts: 39.7 ms per 10^6 Mul against 36.8; 8 wavefronts 41.0)
#ifndef Z4_NW
#define Z4_NW 16
#endif
constexpr int Z4_WAVES = Z4_NW;
constexpr uint32_t Z4_PRE_RUN = 8;  // words per repetition a wavefront collects before it stores them (16 wavefronts x 4 KiB of LDS)
constexpr size_t Z4_LDS_BYTES = C4_LDS_BYTES + (size_t)Z4_WAVES * Z4_PRE_RUN * 64 * 8;
__global__ __launch_bounds__(Z4_WAVES * 64) void k_z64_c4(const uint4* __restrict__ img, const Gate64* __restrict__ gates, Z64FLevel lv, uint32_t mul_per,
                                                        uint32_t lin_per, uint32_t oth_per, Z64FParams p) {
    extern __shared__ uint4 z4_lds[];
    const uint32_t n_qg = p.qgn;
    const uint32_t qg = p.qg0 + blockIdx.x % n_qg, chunk = blockIdx.x / n_qg;
    const uint32_t m_lo = min(lv.mul0 + chunk * mul_per, lv.mul1), m_hi = min(m_lo + mul_per, lv.mul1);
    const uint32_t l_lo = min(lv.mul1 + chunk * lin_per, lv.lin1), l_hi = min(l_lo + lin_per, lv.lin1);
    const uint32_t o_lo = min(lv.lin1 + chunk * oth_per, lv.oth1), o_hi = min(o_lo + oth_per, lv.oth1);
    if (m_hi > m_lo) {  // (uniform over the workgroup)
        const uint4* src = img + (size_t)qg * C4_IMG_U4;
        constexpr uint32_t T = Z4_WAVES * 64, FULL = C4_IMG_U4 / T, REST = C4_IMG_U4 % T;
        uint4 v[FULL];
#pragma unroll
        for (uint32_t i = 0; i < FULL; i++) v[i] = src[threadIdx.x + i * T];
#pragma unroll
        for (uint32_t i = 0; i < FULL; i++) z4_lds[threadIdx.x + i * T] = v[i];
        if (REST != 0 && threadIdx.x < REST) z4_lds[threadIdx.x + FULL * T] = src[threadIdx.x + FULL * T];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t c = lane & 3, ql = lane >> 2;
    const uint32_t rep = (qg * 16 + ql) * 4 + c;  // = 64 * qg + lane
    const uint32_t zo = qg * 512 + 2 * lane;      // the lane's piece 0 inside a row (u64 units)
    const bool writer = rep == 0;
    const uint4* rkl = z4_lds + lane;
    const uint32_t n_m = m_hi - m_lo, n_l = l_hi - l_lo;
    // A wavefront takes CONSECUTIVE Mul gates: their words of the preprocessing transcript are then consecutive in every repetition's
    // stream, and a run of up to eight of them goes through LDS (Z4_PRE_RUN words x 64 lanes per wavefront, behind the key image) and
    // leaves as whole 64-byte segments per four lanes -- instead of 64 lanes storing 8 bytes each into 64 streams per gate
    const uint32_t base = n_m / Z4_WAVES, rem = n_m % Z4_WAVES;
    const uint32_t MI = base + (wave < rem ? 1u : 0u), m_first = m_lo + wave * base + min(wave, rem);
    const uint32_t LI = n_l > wave ? (n_l - wave + Z4_WAVES - 1) / Z4_WAVES : 0u;
    uint64_t* run = (uint64_t*)(z4_lds + C4_IMG_U4) + (size_t)wave * Z4_PRE_RUN * 64;
    uint32_t run_n = 0;
    uint64_t run_ep = 0;
    auto flush = [&]() {
        if (run_n == Z4_PRE_RUN) {
#pragma unroll
            for (int i = 0; i < 4; i++) {  // lane -> repetition 16 i + lane / 4 of the wavefront's 64, words 2 (lane % 4), + 1 of the run
                const uint32_t r = 16 * i + (lane >> 2), w2 = 2 * (lane & 3);
                z4_st16_stream(p.pre + (size_t)(64 * qg + r) * p.pre_words + run_ep + w2, run[(size_t)w2 * 64 + r], run[(size_t)(w2 + 1) * 64 + r]);
            }
        } else {
            for (uint32_t t = 0; t < run_n; t++) __builtin_nontemporal_store(run[(size_t)t * 64 + lane], &p.pre[(size_t)rep * p.pre_words + run_ep + t]);
        }
        run_n = 0;
    };
    uint32_t ld = 0;
    for (uint32_t it = 0; it < MI; it++) {
        const Gate64& g = gates[m_first + it];
        const uint64_t ep = g.ep;
        if (run_n && (run_n == Z4_PRE_RUN || ep != run_ep + run_n)) flush();
        if (!run_n) run_ep = ep;
        const uint64_t w = z4_mul(g, p, rkl, c, rep, zo, writer);
        run[(size_t)run_n * 64 + lane] = w;
        run_n++;
        const uint32_t lend = (uint32_t)(((uint64_t)(it + 1) * LI) / MI);
        for (; ld < lend; ld++) z4_lin(gates[l_lo + wave + Z4_WAVES * ld], p, zo, writer);
    }
    if (run_n) flush();
    for (; ld < LI; ld++) z4_lin(gates[l_lo + wave + Z4_WAVES * ld], p, zo, writer);
    for (uint32_t go = o_lo + wave; go < o_hi; go += Z4_WAVES) z4_oth(gates[go], p, rep, zo, writer);
}

bool z64_c4_supports(uint32_t NQ) { return NQ >= 16 && NQ % 16 == 0; }

void launch_z64_c4(hipStream_t st, const uint32_t* d_img, const Gate64* d_gates, const Z64FLevel& lv, const Z64FParams& p) {
    static const uint32_t cus = [] {
        if (const char* e = getenv("RV_Z64F_WGS")) return (uint32_t)std::max(atoi(e), 1);
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return (uint32_t)n;
    }();
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute((const void*)k_z64_c4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z4_LDS_BYTES);
        raised[dev] = true;
    }
    const uint32_t n_qg = p.qgn;
    const uint32_t n_mul = lv.mul1 - lv.mul0, n_lin = lv.lin1 - lv.mul1, n_oth = lv.oth1 - lv.lin1;
    if (!(n_mul + n_lin + n_oth) || !n_qg) return;
    // one generation of workgroups, one per compute unit (88 KiB of LDS each), every one with an equal share of the level; a level with
    // very little of anything only gets as many as have a wavefront step to do
    const uint64_t per_qg = std::max<uint32_t>(cus / n_qg, 1);
    const uint64_t chunks = std::max<uint64_t>(std::min<uint64_t>(per_qg, ((uint64_t)n_mul + n_lin + n_oth + Z4_WAVES - 1) / Z4_WAVES), 1);
    const uint32_t mul_per = (uint32_t)((n_mul + chunks - 1) / chunks), lin_per = (uint32_t)((n_lin + chunks - 1) / chunks),
                   oth_per = (uint32_t)((n_oth + chunks - 1) / chunks);
    hipLaunchKernelGGL(k_z64_c4, dim3((unsigned)(chunks * n_qg)), dim3(Z4_WAVES * 64), Z4_LDS_BYTES, st, (const uint4*)d_img, d_gates, lv, mul_per, lin_per,
                       oth_per, p);
}

}  // namespace rv
