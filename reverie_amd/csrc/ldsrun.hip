// LDS runs: narrow / deep stretches of a GF(2) circuit with the live wires in LDS (ldsrun.h has the design).
//
// Replaces, for those stretches, the same reference code as the row interpreter (kernels.hip):
//   interpreter/single.rs:25-157      Instance::step / op_mul over the GF(2) ring
//   algebra/gf2/domain.rs:10-63       reconstruct (per-byte parity)
//   transcript/prover.rs:181-232, verifier/online.rs:122-183, verifier/preprocess.rs:46-79
#include <stdlib.h>

#include "gf2dev.h"
#include "internal.h"
#include "ldsrun.h"

namespace rv {

struct LdsRunParams {
    const LdsRec* recs;  // the run's first record
    uint32_t n_steps;    // multiple of LR_CHUNK
    uint32_t n_slots;
};

// workgroup barrier that orders LDS only (__syncthreads() would also drain the consumer's outstanding transcript stores)
__device__ __forceinline__ void lr_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- producers: step records and their global operands into the ring --------------------------------------------------
// ring_val[step][lane] = {m0, m1, v0, v1} (zero where a kind has no such operand):
//   Input:  m0 = mask row;            v0 = prover: witness bit smeared / verifier: supplied masked input
//   Random: m0 = mask row
//   Mul:    m0, m1 = the two fresh mask rows;   verifier: v0 = supplied correction, v1 = the omitted player's broadcast
//   AssertZero / Recon (verifier): v1 = the omitted player's broadcast
//   Load:   m0 = the wire's row, v0 = its corr bits smeared
// SB consecutive steps per call, all their loads in flight together (two dependent round trips per call).
template <int MODE, int QS, int SB>
__device__ __forceinline__ void lr_fill(const LdsRec* __restrict__ recs, LdsRec* ring_rec, uint4* ring_val, const InterpParams& p,
                                        uint32_t NQ, uint32_t q, uint32_t lane) {
    constexpr uint32_t GPS = 64 / QS;
    const uint32_t k = lane / QS, ql = lane % QS;
    const uint32_t* const safe32 = p.rows + q;  // always readable
    const uint8_t* const safe8 = (const uint8_t*)p.rows;
    uint4 r0[SB], r1[SB];
#pragma unroll
    for (int s = 0; s < SB; s++) {
        const uint4* g = (const uint4*)(recs + (size_t)s * GPS + k);
        r0[s] = g[0];
        r1[s] = g[1];
    }
    uint32_t m0[SB], m1[SB], v0[SB], v1[SB], cb[SB];
#pragma unroll
    for (int s = 0; s < SB; s++) {
        const uint32_t kind = (r0[s].w >> 16) & 15u;
        const uint32_t ep = r1[s].y, m = r1[s].z, x = r1[s].w;
        const bool has_m = kind == G_INPUT || kind == G_RANDOM || kind == G_MUL || kind == LK_LOAD;
        m0[s] = *(has_m ? p.rows + (size_t)m * NQ + q : safe32);
        m1[s] = *(kind == G_MUL ? p.rows + (size_t)(m + 1) * NQ + q : safe32);
        cb[s] = *(kind == LK_LOAD ? p.corr + (size_t)m * (NQ >> 1) + (q >> 1) : safe8);
        if (MODE != MODE_VERIFY) {
            v0[s] = *(kind == G_INPUT ? p.wit + x : safe8);
            v1[s] = 0;
        } else {
            v0[s] = *(kind == G_INPUT ? p.sup_in + (size_t)x * NQ + q : kind == G_MUL ? p.sup_corr + (size_t)ep * NQ + q : safe32);
            v1[s] = *((kind == G_MUL || kind == G_ASSERT || kind == G_RECON) ? p.sup_rec + (size_t)x * NQ + q : safe32);
        }
    }
#pragma unroll
    for (int s = 0; s < SB; s++) {
        const uint32_t kind = (r0[s].w >> 16) & 15u;
        const bool has_m = kind == G_INPUT || kind == G_RANDOM || kind == G_MUL || kind == LK_LOAD;
        uint32_t a, b = 0;
        if (MODE != MODE_VERIFY) {
            a = (kind == G_INPUT && v0[s]) ? 0xFFFFFFFFu : 0u;
        } else {
            a = (kind == G_INPUT || kind == G_MUL) ? v0[s] : 0u;
            b = (kind == G_MUL || kind == G_ASSERT || kind == G_RECON) ? v1[s] : 0u;
        }
        if (kind == LK_LOAD) a = expand4((cb[s] >> (4 * (q & 1))) & 0xFu);
        ring_val[s * 64 + lane] = make_uint4(has_m ? m0[s] : 0u, kind == G_MUL ? m1[s] : 0u, a, b);
        LdsRec* dst = ring_rec + s * GPS + k;
        const uint4 x0 = r0[s], x1 = r1[s];
        if (QS == 4) {
            const bool hi = (ql & 2) != 0, odd = (ql & 1) != 0;
            const uint32_t e0 = hi ? x1.x : x0.x, e1 = hi ? x1.y : x0.y, e2 = hi ? x1.z : x0.z, e3 = hi ? x1.w : x0.w;
            ((uint2*)dst)[ql] = make_uint2(odd ? e2 : e0, odd ? e3 : e1);
        } else {
            ((uint4*)dst)[ql] = make_uint4(ql ? x1.x : x0.x, ql ? x1.y : x0.y, ql ? x1.z : x0.z, ql ? x1.w : x0.w);
        }
    }
}

// ---- consumer: one step (lane = gate k of the step, quad word ql of the slice) --------------------------------------
// W[slot][ql] = {share word, corr bits smeared to bytes}.  Every kind runs through the same straight-line code and picks
// its results at the end: a wavefront step usually holds Mul and Xor gates side by side, and a divergent switch costs
// more in branches than the few operations it saves.
template <int MODE, int QS>
__device__ __forceinline__ void lr_step(const uint4 r0, const uint4 r1, const uint4 v, uint2* W, const InterpParams& p, uint32_t NQ,
                                        uint32_t q, uint32_t ql, uint32_t onm) {
    const uint32_t a0 = r0.x & 0xFFFFu, a1 = r0.x >> 16, a2 = r0.y & 0xFFFFu, b0 = r0.y >> 16, b1 = r0.z & 0xFFFFu, b2 = r0.z >> 16;
    const uint32_t dst = r0.w & 0xFFFFu, op = r0.w >> 16;
    const uint32_t eo = r1.x, ep = r1.y, m = r1.z;
    const uint2 A0 = W[a0 * QS + ql], A1 = W[a1 * QS + ql], A2 = W[a2 * QS + ql];
    const uint2 B0 = W[b0 * QS + ql], B1 = W[b1 * QS + ql], B2 = W[b2 * QS + ql];
    const uint32_t ca = (op & LF_CA) ? 0xFFFFFFFFu : 0u, cb = (op & LF_CB) ? 0xFFFFFFFFu : 0u;
    const uint32_t lx = A0.x ^ A1.x ^ A2.x, ly = B0.x ^ B1.x ^ B2.x;
    const uint32_t cx = A0.y ^ A1.y ^ A2.y ^ ca, cy = B0.y ^ B1.y ^ B2.y ^ cb;
    // 0 / ~0 masks of the gate's kind (one v_bfe_i32 each)
    const uint32_t mm = (uint32_t)((int32_t)(op << (31 - LB_MUL)) >> 31), mx = (uint32_t)((int32_t)(op << (31 - LB_XOR)) >> 31);
    const uint32_t mr = (uint32_t)((int32_t)(op << (31 - LB_RECON)) >> 31), mi = (uint32_t)((int32_t)(op << (31 - LB_IN)) >> 31);
    const uint32_t mo = (uint32_t)((int32_t)(op << (31 - LB_OTHER)) >> 31);
    // Mul (single.rs:25-69) -- for an Input, c = reconstruct(its fresh mask)
    const uint32_t a = recon32(lx), b = recon32(ly), c = recon32(v.x);
    uint32_t delta = (a & b) ^ c;
    const uint32_t s = (ly & cx) ^ (lx & cy) ^ v.x ^ v.y;
    uint32_t t = (s & mm) | (lx & ~mm);  // what goes on the online transcript (AssertZero / Recon: the operand's shares)
    uint32_t corr_in;
    if (MODE != MODE_VERIFY) {
        corr_in = v.z ^ c;
    } else {
        corr_in = v.z & onm;
        delta = (v.z & onm) | (delta & ~onm);  // online-verified repetitions: the supplied correction
        t ^= onm ? v.w : 0u;                   // ... and the omitted player's broadcast
    }
    const uint32_t r_raw = recon32(t);
    const uint32_t r = MODE == MODE_VERIFY ? (r_raw & onm) : r_raw;
    if ((op & LF_ON) && (MODE != MODE_VERIFY || onm)) p.on[(size_t)eo * NQ + q] = (corr_in & mi) | (t & ~mi);
    if (mm) store_bits(p.pre, ep, NQ, q, delta);
    const uint32_t drow = (v.y & mm) | ((lx ^ ly) & mx) | (v.x & (mi | mo));
    const uint32_t dcorr = ((r ^ delta ^ (cx & cy)) & mm) | ((cx ^ cy ^ cb) & mx) | ((r ^ cx) & mr) | (corr_in & mi) | (v.z & mo);
    W[dst * QS + ql] = make_uint2(drow, dcorr);
    if (op & (LF_OUT | (1u << LB_ASSERT))) {  // the rare ones
        if (op & (1u << LB_ASSERT)) {
            // prover.rs:221-228 / online.rs:175-177
            const uint32_t bad = (r_raw ^ cx) & (MODE == MODE_VERIFY ? onm : 0xFFFFFFFFu);
            if (bad) atomicOr(p.err, MODE == MODE_VERIFY ? RV_DEV_ZERO_CHECK : RV_E_WITNESS_INVALID);
        }
        if (op & LF_OUT) {  // read again after the run: the row interpreter's layout in global memory
            const uint32_t grow = m + (mm & 1u);
            if (mx | mr) p.rows[(size_t)grow * NQ + q] = drow;
            store_bits(p.corr, grow, NQ, q, dcorr);
        }
    }
}

constexpr int LR_PRODUCERS = 4;  // producer wavefronts per workgroup (each stages LR_CHUNK / LR_PRODUCERS steps of a chunk)

template <int MODE, int QS, bool BATCH>
__global__ __launch_bounds__(64 * (1 + LR_PRODUCERS)) void k_interp_lds(LdsRunParams rp, InterpParams p1, const InterpParams* __restrict__ pp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lr_smem[];
    constexpr uint32_t GPS = 64 / QS;
    constexpr int SB = LR_CHUNK / LR_PRODUCERS;
    const InterpParams p = BATCH ? pp[blockIdx.y] : p1;
    LdsRec* ring_rec = (LdsRec*)lr_smem;                                        // [2][LR_CHUNK][GPS]
    uint4* ring_val = (uint4*)(lr_smem + 2 * LR_CHUNK * GPS * sizeof(LdsRec));  // [2][LR_CHUNK][64]
    uint2* W = (uint2*)(ring_val + 2 * LR_CHUNK * 64);                          // [n_slots][QS]
    const uint32_t NQ = p.NQ;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t k = lane / QS, ql = lane % QS;
    const uint32_t q = blockIdx.x * QS + ql;
    const uint32_t n_chunks = rp.n_steps / LR_CHUNK;
    const uint32_t s_first = (wave - 1) * SB;  // producers: first step of the chunk this wavefront stages
    uint32_t onm = 0;
    if (wave == 0) {
        if (MODE == MODE_VERIFY) onm = p.on_mask[q];
        if (lane < QS) W[lane] = make_uint2(0u, 0u);
    } else {
        lr_fill<MODE, QS, SB>(rp.recs + (size_t)s_first * GPS, ring_rec + s_first * GPS, ring_val + s_first * 64, p, NQ, q, lane);
    }
    lr_barrier();
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint32_t buf = c & 1u;
        if (wave == 0) {
            const LdsRec* rr = ring_rec + buf * LR_CHUNK * GPS + k;
            const uint4* vv = ring_val + buf * LR_CHUNK * 64 + lane;
            uint4 n0 = ((const uint4*)rr)[0], n1 = ((const uint4*)rr)[1], nv = vv[0];
#pragma unroll
            for (uint32_t s = 0; s < LR_CHUNK; s++) {
                const uint4 c0 = n0, c1 = n1, cv = nv;
                if (s + 1 < LR_CHUNK) {  // the next step's records are requested ahead of this step's LDS writes
                    n0 = ((const uint4*)(rr + (s + 1) * GPS))[0];
                    n1 = ((const uint4*)(rr + (s + 1) * GPS))[1];
                    nv = vv[(s + 1) * 64];
                }
                lr_step<MODE, QS>(c0, c1, cv, W, p, NQ, q, ql, onm);
            }
        } else if (c + 1 < n_chunks) {
            const uint32_t st = (c + 1) * LR_CHUNK + s_first, nb = (buf ^ 1u) * LR_CHUNK + s_first;
            lr_fill<MODE, QS, SB>(rp.recs + (size_t)st * GPS, ring_rec + nb * GPS, ring_val + nb * 64, p, NQ, q, lane);
        }
        lr_barrier();
    }
}

template <int MODE, int QS>
static void launch_lds_mq(hipStream_t st, const LdsRunParams& rp, size_t lds, uint32_t NQ, const InterpParams& p, const InterpParams* d_pp,
                          uint32_t batch) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)k_interp_lds<MODE, QS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        (void)hipFuncSetAttribute((const void*)k_interp_lds<MODE, QS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        return true;
    }();
    (void)attr;
    if (d_pp)
        hipLaunchKernelGGL((k_interp_lds<MODE, QS, true>), dim3(NQ / QS, batch), dim3(64 * (1 + LR_PRODUCERS)), lds, st, rp, InterpParams{}, d_pp);
    else
        hipLaunchKernelGGL((k_interp_lds<MODE, QS, false>), dim3(NQ / QS), dim3(64 * (1 + LR_PRODUCERS)), lds, st, rp, p, (const InterpParams*)nullptr);
}

void launch_interp_lds(hipStream_t st, int mode, uint32_t QS, uint32_t NQ, const LdsRec* d_recs, uint32_t n_steps, uint32_t n_slots,
                       const InterpParams& p, const InterpParams* d_pp, uint32_t batch) {
    LdsRunParams rp{d_recs, n_steps, n_slots};
    const size_t lds = lds_run_bytes(QS, n_slots);
    if (QS == 4) {
        if (mode == MODE_VERIFY)
            launch_lds_mq<MODE_VERIFY, 4>(st, rp, lds, NQ, p, d_pp, batch);
        else
            launch_lds_mq<MODE_PROVE, 4>(st, rp, lds, NQ, p, d_pp, batch);
    } else {
        if (mode == MODE_VERIFY)
            launch_lds_mq<MODE_VERIFY, 2>(st, rp, lds, NQ, p, d_pp, batch);
        else
            launch_lds_mq<MODE_PROVE, 2>(st, rp, lds, NQ, p, d_pp, batch);
    }
}

}  // namespace rv
