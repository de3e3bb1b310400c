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

// ---- producer: one chunk of step records and their global operands into the ring ----------------------------------
// ring_val[step][lane] = {m0, m1, v0, v1}:
//   Input:  m0 = mask row;            v0 = prover: witness bit smeared / verifier: supplied masked input
//   Random: m0 = mask row
//   Mul:    m0, m1 = the two fresh mask rows;   verifier: v0 = supplied correction, v1 = the omitted player's broadcast
//   AssertZero / Recon (verifier): v1 = the omitted player's broadcast
//   Load:   m0 = the wire's row, v0 = its corr nibble
template <int MODE, int QS>
__device__ __forceinline__ void lr_fill(const LdsRec* __restrict__ recs, LdsRec* ring_rec, uint4* ring_val, const InterpParams& p,
                                        uint32_t NQ, uint32_t q, uint32_t lane) {
    constexpr uint32_t GPS = 64 / QS;
    constexpr int SB = 8;  // steps in flight together
    const uint32_t k = lane / QS, ql = lane % QS;
    const uint32_t* const safe32 = p.rows + q;  // always readable
    const uint8_t* const safe8 = (const uint8_t*)p.rows;
#pragma unroll 1
    for (uint32_t s0 = 0; s0 < LR_CHUNK; s0 += SB) {
        uint4 r0[SB], r1[SB];
#pragma unroll
        for (int s = 0; s < SB; s++) {
            const uint4* g = (const uint4*)(recs + (size_t)(s0 + s) * GPS + k);
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
            uint32_t a = v0[s];
            if (MODE != MODE_VERIFY) a = a ? 0xFFFFFFFFu : 0u;
            if (kind == LK_LOAD) a = (cb[s] >> (4 * (q & 1))) & 0xFu;
            ring_val[(s0 + s) * 64 + lane] = make_uint4(m0[s], m1[s], a, v1[s]);
            LdsRec* dst = ring_rec + (s0 + s) * GPS + k;
            if (QS == 4) {
                const uint4 h = (ql & 2) ? r1[s] : r0[s];
                ((uint2*)dst)[ql] = (ql & 1) ? make_uint2(h.z, h.w) : make_uint2(h.x, h.y);
            } else {
                const uint4 x0 = r0[s], x1 = r1[s];
                ((uint4*)dst)[ql] = make_uint4(ql ? x1.x : x0.x, ql ? x1.y : x0.y, ql ? x1.z : x0.z, ql ? x1.w : x0.w);
            }
        }
    }
}

// ---- consumer: one step (lane = gate k of the step, quad word ql of the slice) --------------------------------------
template <int MODE, int QS>
__device__ __forceinline__ void lr_step(const uint4 r0, const uint4 r1, const uint4 v, uint32_t* R, uint8_t* C, const InterpParams& p,
                                        uint32_t NQ, uint32_t q, uint32_t ql, uint32_t onm) {
    const uint32_t a0 = r0.x & 0xFFFFu, a1 = r0.x >> 16, a2 = r0.y & 0xFFFFu, b0 = r0.y >> 16, b1 = r0.z & 0xFFFFu, b2 = r0.z >> 16;
    const uint32_t dst = r0.w & 0xFFFFu, op = r0.w >> 16, kind = op & 15u;
    const uint32_t eo = r1.x, ep = r1.y, m = r1.z;
    const uint32_t lx = R[a0 * QS + ql] ^ R[a1 * QS + ql] ^ R[a2 * QS + ql];
    const uint32_t ly = R[b0 * QS + ql] ^ R[b1 * QS + ql] ^ R[b2 * QS + ql];
    const uint32_t nx = (uint32_t)C[a0 * QS + ql] ^ C[a1 * QS + ql] ^ C[a2 * QS + ql];
    const uint32_t ny = (uint32_t)C[b0 * QS + ql] ^ C[b1 * QS + ql] ^ C[b2 * QS + ql];
    const uint32_t ca = (op & LF_CA) ? 0xFFFFFFFFu : 0u, cb = (op & LF_CB) ? 0xFFFFFFFFu : 0u;
    uint32_t drow = 0, dn = 0;
    switch (kind) {
    case G_INPUT: {
        const uint32_t lam = v.x;
        uint32_t corr;
        if (MODE != MODE_VERIFY)
            corr = v.z ^ recon32(lam);
        else
            corr = onm ? (v.z & onm) : 0u;
        if (MODE != MODE_VERIFY || onm) p.on[(size_t)eo * NQ + q] = corr;
        drow = lam;
        dn = compress4(corr);
        break;
    }
    case G_XORK:
        drow = lx ^ ly;
        dn = nx ^ ny ^ (ca & 0xFu);
        break;
    case G_RANDOM:
        drow = v.x;
        break;
    case G_MUL: {
        const uint32_t lab = v.x, lnew = v.y;
        const uint32_t a = recon32(lx), b = recon32(ly), c = recon32(lab);
        const uint32_t cx = expand4(nx) ^ ca, cy = expand4(ny) ^ cb;
        uint32_t delta = (a & b) ^ c;
        uint32_t s = (ly & cx) ^ (lx & cy) ^ lab ^ lnew;
        uint32_t r;
        if (MODE != MODE_VERIFY) {
            r = recon32(s);
        } else {
            if (onm) {
                delta = (v.z & onm) | (delta & ~onm);
                s ^= v.w;
            }
            r = recon32(s) & onm;
        }
        if (MODE != MODE_VERIFY || onm) p.on[(size_t)eo * NQ + q] = s;
        store_bits(p.pre, ep, NQ, q, delta);
        drow = lnew;
        dn = compress4(r ^ delta ^ (cx & cy));
        break;
    }
    case G_RECON: {
        uint32_t mm = lx;
        if (MODE == MODE_VERIFY && onm) mm ^= v.w;
        if (MODE != MODE_VERIFY || onm) p.on[(size_t)eo * NQ + q] = mm;
        uint32_t r = recon32(mm);
        if (MODE == MODE_VERIFY) r &= onm;
        dn = compress4(r ^ expand4(nx) ^ ca);
        break;
    }
    case G_ASSERT: {
        uint32_t mm = lx;
        if (MODE == MODE_VERIFY && onm) mm ^= v.w;
        if (MODE != MODE_VERIFY || onm) p.on[(size_t)eo * NQ + q] = mm;
        const uint32_t bad = recon32(mm) ^ expand4(nx) ^ ca;
        if (MODE != MODE_VERIFY) {
            if (bad) atomicOr(p.err, RV_E_WITNESS_INVALID);
        } else {
            if (bad & onm) atomicOr(p.err, RV_DEV_ZERO_CHECK);
        }
        break;
    }
    case LK_LOAD:
        drow = v.x;
        dn = v.z;
        break;
    default:
        break;
    }
    if (dst != LR_NONE) {
        R[dst * QS + ql] = drow;
        C[dst * QS + ql] = (uint8_t)dn;
    }
    if (op & LF_OUT) {  // read again after the run: the row interpreter's layout in global memory
        const uint32_t grow = kind == G_MUL ? m + 1 : m;
        if (kind == G_XORK || kind == G_RECON) p.rows[(size_t)grow * NQ + q] = drow;
        const uint32_t other = pair_swap(dn);
        if (!(q & 1)) p.corr[(size_t)grow * (NQ >> 1) + (q >> 1)] = (uint8_t)(dn | (other << 4));
    }
}

template <int MODE, int QS, bool BATCH>
__global__ __launch_bounds__(128) void k_interp_lds(LdsRunParams rp, InterpParams p1, const InterpParams* __restrict__ pp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lr_smem[];
    constexpr uint32_t GPS = 64 / QS;
    const InterpParams p = BATCH ? pp[blockIdx.y] : p1;
    LdsRec* ring_rec = (LdsRec*)lr_smem;                                             // [2][LR_CHUNK][GPS]
    uint4* ring_val = (uint4*)(lr_smem + 2 * LR_CHUNK * GPS * sizeof(LdsRec));       // [2][LR_CHUNK][64]
    uint32_t* R = (uint32_t*)(ring_val + 2 * LR_CHUNK * 64);                         // [n_slots][QS]
    uint8_t* C = (uint8_t*)(R + (size_t)rp.n_slots * QS);                            // [n_slots][QS]
    const uint32_t NQ = p.NQ;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t k = lane / QS, ql = lane % QS;
    const uint32_t q = blockIdx.x * QS + ql;
    const uint32_t n_chunks = rp.n_steps / LR_CHUNK;
    uint32_t onm = 0;
    if (wave == 0) {
        if (MODE == MODE_VERIFY) onm = p.on_mask[q];
        if (lane < QS) {
            R[lane] = 0;
            C[lane] = 0;
        }
    } else {
        lr_fill<MODE, QS>(rp.recs, ring_rec, ring_val, p, NQ, q, lane);
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
                lr_step<MODE, QS>(c0, c1, cv, R, C, p, NQ, q, ql, onm);
            }
        } else if (c + 1 < n_chunks) {
            lr_fill<MODE, QS>(rp.recs + (size_t)(c + 1) * LR_CHUNK * GPS, ring_rec + (buf ^ 1u) * LR_CHUNK * GPS,
                              ring_val + (buf ^ 1u) * LR_CHUNK * 64, p, NQ, q, lane);
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
        hipLaunchKernelGGL((k_interp_lds<MODE, QS, true>), dim3(NQ / QS, batch), dim3(128), lds, st, rp, InterpParams{}, d_pp);
    else
        hipLaunchKernelGGL((k_interp_lds<MODE, QS, false>), dim3(NQ / QS), dim3(128), lds, st, rp, p, (const InterpParams*)nullptr);
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
