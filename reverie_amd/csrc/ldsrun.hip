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
    uint32_t eo0, ep0;   // lowest online / preprocessing transcript row the run writes (store offsets are relative, 32 bits)
};

// workgroup barrier that orders LDS only (__syncthreads() would also drain the outstanding global loads and stores)
__device__ __forceinline__ void lr_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A lone wavefront issues one instruction every four cycles whatever its kind, so the consumer's step time is its
// instruction count.  Everything that does not depend on wire values is therefore done by the producers and handed over
// per LANE, ready to use: ring[buffer][step][field][lane], four 16-byte fields
//   F0 = LDS byte addresses of operand slots a0 a1 a2 b0          F1 = b1 b2, result slot, op word
//   F2 = m0 m1 v0 v1 (zero where a kind has no such operand):
//          Input:  m0 = mask row;   v0 = prover: witness bit smeared / verifier: supplied masked input
//          Random: m0 = mask row
//          Mul:    m0, m1 = the two fresh mask rows;   verifier: v0 = supplied correction, v1 = the omitted player's broadcast
//          AssertZero / Recon (verifier): v1 = the omitted player's broadcast
//          Load:   m0 = the wire's row, v0 = its corr bits smeared
//   F3 = reconstruct(m0), byte offset of the online-transcript word, byte offset of the preprocessing byte, global row of
//        the result (live-out wires)
constexpr int LR_PRODUCERS = 4;                 // producer wavefronts per workgroup
constexpr int LR_SB = LR_CHUNK / LR_PRODUCERS;  // steps of a chunk each of them stages
constexpr uint32_t LR_RING_BYTES = 2 * LR_CHUNK * 4 * 64 * 16;

extern __shared__ __attribute__((aligned(16))) uint8_t lr_smem[];
__device__ __forceinline__ uint8_t* lr_smem_base() { return lr_smem; }

struct LrRecs {  // a producer's records of one chunk (registers)
    uint4 r0[LR_SB], r1[LR_SB];
};
struct LrVals {  // ... and the global operands loaded for them
    uint32_t m0[LR_SB], m1[LR_SB], v0[LR_SB], v1[LR_SB], cb[LR_SB];
};

template <int QS>
__device__ __forceinline__ void lr_load_recs(const LdsRec* __restrict__ recs, uint32_t k, LrRecs& R) {
    constexpr uint32_t GPS = 64 / QS;
#pragma unroll
    for (int s = 0; s < LR_SB; s++) {
        const uint4* g = (const uint4*)(recs + (size_t)s * GPS + k);
        R.r0[s] = g[0];
        R.r1[s] = g[1];
    }
}

template <int MODE>
__device__ __forceinline__ void lr_load_vals(const LrRecs& R, const InterpParams& p, uint32_t NQ, uint32_t q, LrVals& V) {
    const uint32_t* const safe32 = p.rows + q;  // always readable
    const uint8_t* const safe8 = (const uint8_t*)p.rows;
#pragma unroll
    for (int s = 0; s < LR_SB; s++) {
        const uint32_t kind = (R.r0[s].w >> 16) & 15u;
        const uint32_t ep = R.r1[s].y, m = R.r1[s].z, x = R.r1[s].w;
        const bool has_m = kind == G_INPUT || kind == G_RANDOM || kind == G_MUL || kind == LK_LOAD;
        V.m0[s] = *(has_m ? p.rows + (size_t)m * NQ + q : safe32);
        V.m1[s] = *(kind == G_MUL ? p.rows + (size_t)(m + 1) * NQ + q : safe32);
        V.cb[s] = *(kind == LK_LOAD ? p.corr + (size_t)m * (NQ >> 1) + (q >> 1) : safe8);
        if (MODE != MODE_VERIFY) {
            V.v0[s] = *(kind == G_INPUT ? p.wit + x : safe8);
            V.v1[s] = 0;
        } else {
            // (supplied rows hold sup_nq quad words -- the ones with opened repetitions; the others' values are masked out anyway)
            const bool sq = q < p.sup_nq;
            V.v0[s] = *(sq && kind == G_INPUT ? p.sup_in + (size_t)x * p.sup_nq + q : sq && kind == G_MUL ? p.sup_corr + (size_t)ep * p.sup_nq + q : safe32);
            V.v1[s] = *((sq && (kind == G_MUL || kind == G_ASSERT || kind == G_RECON)) ? p.sup_rec + (size_t)x * p.sup_nq + q : safe32);
        }
    }
}

// ring_step0: field 0 of this producer's first step in the destination buffer, at this lane
template <int MODE, int QS>
__device__ __forceinline__ void lr_stage(const LrRecs& R, const LrVals& V, uint4* ring_step0, const LdsRunParams& rp, uint32_t NQ, uint32_t q,
                                         uint32_t ql) {
    // W[slot][ql] sits at wbase + slot * QS * 8 (an absolute LDS address)
    const uint32_t wbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lr_smem_base() + LR_RING_BYTES + ql * 8;
#pragma unroll
    for (int s = 0; s < LR_SB; s++) {
        const uint4 r0 = R.r0[s], r1 = R.r1[s];
        const uint32_t op = r0.w >> 16, kind = op & 15u;
        const bool has_m = kind == G_INPUT || kind == G_RANDOM || kind == G_MUL || kind == LK_LOAD;
        uint32_t a, b = 0;
        if (MODE != MODE_VERIFY) {
            a = (kind == G_INPUT && V.v0[s]) ? 0xFFFFFFFFu : 0u;
        } else {
            a = (kind == G_INPUT || kind == G_MUL) ? V.v0[s] : 0u;
            b = (kind == G_MUL || kind == G_ASSERT || kind == G_RECON) ? V.v1[s] : 0u;
        }
        if (kind == LK_LOAD) a = expand4((V.cb[s] >> (4 * (q & 1))) & 0xFu);
        const uint32_t m0 = has_m ? V.m0[s] : 0u, m1 = kind == G_MUL ? V.m1[s] : 0u;
        auto addr = [&](uint32_t slot) { return wbase + slot * (QS * 8); };
        uint4* f = ring_step0 + s * 4 * 64;
        f[0] = make_uint4(addr(r0.x & 0xFFFFu), addr(r0.x >> 16), addr(r0.y & 0xFFFFu), addr(r0.y >> 16));
        f[64] = make_uint4(addr(r0.z & 0xFFFFu), addr(r0.z >> 16), addr(r0.w & 0xFFFFu), op);
        // (the prover has no use for the fourth word: it carries reconstruct(m1), which takes a reconstruct off the
        // consumer's dependency chain -- see lr_step)
        f[128] = make_uint4(m0, m1, a, MODE != MODE_VERIFY ? recon32(m1) : b);
        f[192] = make_uint4(recon32(m0), ((r1.x - rp.eo0) * NQ + q) * 4u, (r1.y - rp.ep0) * (NQ >> 1) + (q >> 1), r1.z + (kind == G_MUL ? 1u : 0u));
    }
}

// ---- consumer: one step (lane = gate k of the step, quad word ql of the slice) --------------------------------------
// W[slot][ql] = {share word, corr bits smeared to bytes}.  Every kind runs through the same straight-line code and picks
// its results with masks: a wavefront step usually holds Mul and Xor gates side by side, and a divergent switch costs
// more in branches than the few operations it saves.
typedef uint32_t lr_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lr_u32x2 lds_u32x2;
__device__ __forceinline__ uint2 lds_get(uint32_t addr) {
    const lr_u32x2 v = *(const lds_u32x2*)(uintptr_t)addr;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void lds_put(uint32_t addr, uint32_t a, uint32_t b) {
    lr_u32x2 v;
    v.x = a;
    v.y = b;
    *(lds_u32x2*)(uintptr_t)addr = v;
}

// what a step leaves for later: its transcript stores and rare paths run AFTER the next step's LDS gathers have been
// issued, i.e. inside the latency that step would otherwise sit out (nothing here touches LDS)
struct LrPending {
    uint32_t op = 0, on_val = 0, delta = 0, on_off = 0, pre_off = 0, drow = 0, dcorr = 0, bad = 0, grow = 0;
};

// QS = 1 (a slice is ONE quad word, 64 gates per step): the quad's nibble of a bit-packed row shares its byte with the
// neighbouring slice's, i.e. another workgroup's -- the nibble is cleared and set with two fire-and-forget atomics on the
// aligned word (same lane, same address: applied in program order) instead of one byte store per pair of lanes
__device__ __forceinline__ void lr_put_nibble(uint8_t* base, size_t byte_off, uint32_t q, uint32_t n) {
    uint32_t* wp = (uint32_t*)(base + (byte_off & ~(size_t)3));
    const uint32_t sh = (((uint32_t)byte_off & 3u) << 3) | ((q & 1u) << 2);
    (void)__hip_atomic_fetch_and(wp, ~(0xFu << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_or(wp, n << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE, int QS>
__device__ __forceinline__ void lr_flush(const LrPending& w, uint8_t* on_base, uint8_t* pre_base, const InterpParams& p, uint32_t NQ,
                                         uint32_t q, uint32_t onm) {
    const uint32_t op = w.op;
    if ((op & LF_ON) && (MODE != MODE_VERIFY || onm)) *(uint32_t*)(on_base + (size_t)w.on_off) = w.on_val;
    if (op & (1u << LB_MUL)) {
        const uint32_t n = compress4(w.delta);
        if (QS == 1) {
            lr_put_nibble(pre_base, (size_t)w.pre_off, q, n);
        } else {
            const uint32_t other = pair_swap(n);
            if (!(q & 1)) pre_base[(size_t)w.pre_off] = (uint8_t)(n | (other << 4));
        }
    }
    if (op & (LF_OUT | (1u << LB_ASSERT))) {  // the rare ones
        if (op & (1u << LB_ASSERT)) {
            // prover.rs:221-228 / online.rs:175-177
            if (w.bad) atomicOr(p.err, MODE == MODE_VERIFY ? RV_DEV_ZERO_CHECK : RV_E_WITNESS_INVALID);
        }
        if (op & LF_OUT) {  // read again after the run: the row interpreter's layout in global memory
            if (op & ((1u << LB_XOR) | (1u << LB_RECON))) p.rows[(size_t)w.grow * NQ + q] = w.drow;
            if (QS == 1)
                lr_put_nibble(p.corr, (size_t)w.grow * (NQ >> 1) + (q >> 1), q, compress4(w.dcorr));
            else
                store_bits(p.corr, w.grow, NQ, q, w.dcorr);
        }
    }
}

template <int MODE, int QS>
__device__ __forceinline__ void lr_step(const uint4 f0, const uint4 f1, const uint4 v, const uint4 f3, LrPending& pend, uint8_t* on_base,
                                        uint8_t* pre_base, const InterpParams& p, uint32_t NQ, uint32_t q, uint32_t onm) {
    const uint32_t op = f1.w;
    // (the fields hold absolute LDS addresses: no base register to add)
    const uint2 A0 = lds_get(f0.x), A1 = lds_get(f0.y), A2 = lds_get(f0.z);
    const uint2 B0 = lds_get(f0.w), B1 = lds_get(f1.x), B2 = lds_get(f1.y);
    __builtin_amdgcn_sched_barrier(0);
    lr_flush<MODE, QS>(pend, on_base, pre_base, p, NQ, q, onm);  // the previous step's stores, while the gathers are in flight
    __builtin_amdgcn_sched_barrier(0);
    // 0 / ~0 masks (one v_bfe_i32 each): operand constants, the gate's kind
    const uint32_t ca = (uint32_t)((int32_t)(op << 27) >> 31), cb = (uint32_t)((int32_t)(op << 26) >> 31);
    const uint32_t mm = (uint32_t)((int32_t)(op << (31 - LB_MUL)) >> 31), mx = (uint32_t)((int32_t)(op << (31 - LB_XOR)) >> 31);
    const uint32_t mr = (uint32_t)((int32_t)(op << (31 - LB_RECON)) >> 31), mi = (uint32_t)((int32_t)(op << (31 - LB_IN)) >> 31);
    const uint32_t mo = (uint32_t)((int32_t)(op << (31 - LB_OTHER)) >> 31);
    const uint32_t lx = A0.x ^ A1.x ^ A2.x, ly = B0.x ^ B1.x ^ B2.x;
    const uint32_t cx = A0.y ^ A1.y ^ A2.y ^ ca, cy = B0.y ^ B1.y ^ B2.y ^ cb;
    // Mul (single.rs:25-69) -- for an Input, c = reconstruct(its fresh mask)
    const uint32_t a = recon32(lx), b = recon32(ly), c = f3.x;
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
    // reconstruct(t) without a second reconstruct behind the gathers (a lone wavefront pays ~10 cycles per DEPENDENT
    // instruction: the step is bound by its longest chain, not by its instruction count): reconstruct is linear, and a
    // share word ANDed with a per-repetition 0x00 / 0xFF mask reconstructs to the AND of its reconstruction with that mask,
    // so for a Mul  reconstruct(s) = (b & cx) ^ (a & cy) ^ reconstruct(m0) ^ reconstruct(m1);  AssertZero / Recon: a
    // (prover only: the verifier's fourth word is the omitted player's broadcast, and two extra reconstructs next to the
    // chain cost it more than the one on the chain -- measured)
    uint32_t r_raw;
    if (MODE != MODE_VERIFY) {
        const uint32_t r_mul = (b & cx) ^ (a & cy) ^ c ^ v.w;
        r_raw = (r_mul & mm) | (a & ~mm);
    } else {
        r_raw = recon32(t);
    }
    const uint32_t r = MODE == MODE_VERIFY ? (r_raw & onm) : r_raw;
    const uint32_t drow = (v.y & mm) | ((lx ^ ly) & mx) | (v.x & (mi | mo));
    const uint32_t dcorr = ((r ^ delta ^ (cx & cy)) & mm) | ((cx ^ cy ^ cb) & mx) | ((r ^ cx) & mr) | (corr_in & mi) | (v.z & mo);
    lds_put(f1.z, drow, dcorr);
    pend.op = op;
    pend.on_val = (corr_in & mi) | (t & ~mi);
    pend.delta = delta;
    pend.on_off = f3.y;
    pend.pre_off = f3.z;
    pend.drow = drow;
    pend.dcorr = dcorr;
    pend.bad = (r_raw ^ cx) & (MODE == MODE_VERIFY ? onm : 0xFFFFFFFFu);
    pend.grow = f3.w;
}

template <int MODE, int QS, bool BATCH>
__global__ __launch_bounds__(64 * (1 + LR_PRODUCERS)) void k_interp_lds(LdsRunParams rp, InterpParams p1, const InterpParams* __restrict__ pp) {
    constexpr uint32_t GPS = 64 / QS;
    const InterpParams p = BATCH ? pp[blockIdx.y] : p1;
    uint4* ring = (uint4*)lr_smem;  // [2][LR_CHUNK][4][64], then W[n_slots][QS] uint2
    const uint32_t NQ = p.NQ;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t k = lane / QS, ql = lane % QS;
    const uint32_t q = blockIdx.x * QS + ql;
    const uint32_t n_chunks = rp.n_steps / LR_CHUNK;
    if (wave == 0) {
        // ---- consumer
        const uint32_t onm = MODE == MODE_VERIFY ? p.on_mask[q] : 0u;
        uint8_t* on_base = (uint8_t*)(p.on + (size_t)rp.eo0 * NQ);
        uint8_t* pre_base = p.pre + (size_t)rp.ep0 * (NQ >> 1);
        if (lane < QS) ((uint2*)(lr_smem + LR_RING_BYTES))[lane] = make_uint2(0u, 0u);  // the zero wire
        lr_barrier();
        LrPending pend;
        for (uint32_t c = 0; c < n_chunks; c++) {
            const uint4* f = ring + (c & 1u) * (LR_CHUNK * 4 * 64) + lane;
            uint4 n0 = f[0], n1 = f[64], n2 = f[128], n3 = f[192];
#pragma unroll
            for (uint32_t s = 0; s < LR_CHUNK; s++) {
                const uint4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
                if (s + 1 < LR_CHUNK) {  // the next step's fields are requested ahead of this step's LDS write
                    n0 = f[(s + 1) * 256];
                    n1 = f[(s + 1) * 256 + 64];
                    n2 = f[(s + 1) * 256 + 128];
                    n3 = f[(s + 1) * 256 + 192];
                }
                lr_step<MODE, QS>(c0, c1, c2, c3, pend, on_base, pre_base, p, NQ, q, onm);
            }
            lr_barrier();
        }
        lr_flush<MODE, QS>(pend, on_base, pre_base, p, NQ, q, onm);
    } else {
        // ---- producers, three chunks deep: while the consumer works on chunk c a producer writes chunk c + 1 into the
        // other buffer from operands it requested one iteration ago, requests the operands of chunk c + 2 with records
        // it requested one iteration ago, and requests the records of chunk c + 3 -- every load has a whole consumer
        // chunk to arrive, and none of them is waited for at the barrier (chunk numbers past the end are clamped:
        // those buffers are never read)
        const uint32_t s_first = (wave - 1) * LR_SB;
        auto chunk_recs = [&](uint32_t c) { return rp.recs + ((size_t)(c < n_chunks ? c : n_chunks - 1) * LR_CHUNK + s_first) * GPS; };
        uint4* my = ring + s_first * 4 * 64 + lane;
        LrRecs Ra, Rb;
        LrVals Vb;
        lr_load_recs<QS>(chunk_recs(0), k, Rb);
        lr_load_recs<QS>(chunk_recs(1), k, Ra);
        lr_load_vals<MODE>(Rb, p, NQ, q, Vb);
        lr_stage<MODE, QS>(Rb, Vb, my, rp, NQ, q, ql);  // chunk 0
        Rb = Ra;
        lr_load_vals<MODE>(Rb, p, NQ, q, Vb);           // chunk 1 in flight
        lr_load_recs<QS>(chunk_recs(2), k, Ra);         // chunk 2's records in flight
        lr_barrier();
#pragma unroll 1
        for (uint32_t c = 0; c < n_chunks; c++) {
            lr_stage<MODE, QS>(Rb, Vb, my + ((c + 1) & 1u) * (LR_CHUNK * 4 * 64), rp, NQ, q, ql);  // chunk c + 1
            Rb = Ra;
            lr_load_vals<MODE>(Rb, p, NQ, q, Vb);        // chunk c + 2
            lr_load_recs<QS>(chunk_recs(c + 3), k, Ra);  // chunk c + 3
            lr_barrier();
        }
    }
}

template <int MODE, int QS>
static void launch_lds_mq(hipStream_t st, const LdsRunParams& rp, size_t lds, uint32_t NQ, const InterpParams& p, const InterpParams* d_pp,
                          uint32_t batch) {
    static const bool attr = [] {
        const int want = (int)std::min<size_t>(160 * 1024, device_lds_limit()) - 1024;
        const hipError_t a = hipFuncSetAttribute((const void*)k_interp_lds<MODE, QS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
        const hipError_t b = hipFuncSetAttribute((const void*)k_interp_lds<MODE, QS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
        if (a != hipSuccess || b != hipSuccess) {
            (void)hipGetLastError();
            fprintf(stderr, "[reverie_amd] LDS runs: the device refused %d bytes of dynamic LDS per workgroup (%s)\n", want, hipGetErrorString(a != hipSuccess ? a : b));
        }
        return a == hipSuccess && b == hipSuccess;
    }();
    (void)attr;  // (circuit_upload planned the runs within the device's limit; a refusal shows up as a failed launch, reported by the caller's hipGetLastError)
    if (d_pp)
        hipLaunchKernelGGL((k_interp_lds<MODE, QS, true>), dim3(NQ / QS, batch), dim3(64 * (1 + LR_PRODUCERS)), lds, st, rp, InterpParams{}, d_pp);
    else
        hipLaunchKernelGGL((k_interp_lds<MODE, QS, false>), dim3(NQ / QS), dim3(64 * (1 + LR_PRODUCERS)), lds, st, rp, p, (const InterpParams*)nullptr);
}

void launch_interp_lds(hipStream_t st, int mode, uint32_t QS, uint32_t NQ, const LdsRec* d_recs, uint32_t n_steps, uint32_t n_slots,
                       uint32_t eo0, uint32_t ep0, const InterpParams& p, const InterpParams* d_pp, uint32_t batch) {
    LdsRunParams rp{d_recs, n_steps, n_slots, eo0, ep0};
    const size_t lds = lds_run_bytes(QS, n_slots);
    if (QS == 4) {
        if (mode == MODE_VERIFY)
            launch_lds_mq<MODE_VERIFY, 4>(st, rp, lds, NQ, p, d_pp, batch);
        else
            launch_lds_mq<MODE_PROVE, 4>(st, rp, lds, NQ, p, d_pp, batch);
    } else if (QS == 1) {
        if (mode == MODE_VERIFY)
            launch_lds_mq<MODE_VERIFY, 1>(st, rp, lds, NQ, p, d_pp, batch);
        else
            launch_lds_mq<MODE_PROVE, 1>(st, rp, lds, NQ, p, d_pp, batch);
    } else {
        if (mode == MODE_VERIFY)
            launch_lds_mq<MODE_VERIFY, 2>(st, rp, lds, NQ, p, d_pp, batch);
        else
            launch_lds_mq<MODE_PROVE, 2>(st, rp, lds, NQ, p, d_pp, batch);
    }
}

}  // namespace rv
