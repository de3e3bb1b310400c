// Rep-sliced prover kernels for gfx950: one workgroup = ONE repetition, live wires in LDS (repprog.h).
//
// STATUS: OPT-IN (RV_REP=1 / 2), NOT on the default path.  This is the layout BASELINE.json's north_star prescribes ("one lane
// = one (repetition, player) slot, wire values staged in LDS"); it was built, is byte-identical to the oracle
// (tests/test_gpu_parity.py::test_rep_sliced_path) and was REFUTED by measurement as a replacement for the row interpreter:
// 9.9 ms of GPU time per proof of the 10^7-gate circuit against 5.4 ms ("Rep-sliced path").  It stays in
// the tree as the evidence for that choice and as the layout a fused mask-generator + interpreter kernel would start from.
//
// Replaces, for the prover of a pure GF(2) circuit whose live wires fit the LDS (all under /root/reference/src/):
//   interpreter/single.rs:25-157      Instance::step / op_mul
//   transcript/prover.rs:181-232      ProverTranscript::{input, reconstruct, correction, zero_check}
//   algebra/gf2/domain.rs:10-63       Share*Recon, reconstruct
//   transcript/prover.rs:57-175 + algebra/gf2/{share,recon}.rs Pack / PackSelected   (the openings)
//
// Layout (all REP-MAJOR, one byte = the 8 players of one repetition, player p at bit 7 - p -- the byte the
// reference hashes for a share, gf2/share.rs:211-218):
//   masks  [R][mask_stride]   byte m = the m-th ShareGen::next() of that repetition (written by k_aes_rep_masks)
//   on     [R][on_stride]     online transcript, one byte per event (a share byte, or 0x00/0xFF for a masked input)
//   pre    [R][pre_stride]    preprocessing transcript, 0x00/0xFF per correction (gf2/recon.rs:314-321)
//   LDS    [slot]             mask byte of every live share row (repprog.h); slot 0 = the zero row
// A wire's public correction is never stored: the prover knows every wire's cleartext value v (k_rep_clear evaluates
// the circuit once per proof -- it is the same in all 256 repetitions) and  corr = v XOR parity(mask).
// Integer VALU + LDS only; no MFMA (this is XOR/AND work).
#include "internal.h"
#include "repprog.h"

namespace rv {

// ---- SWAR helpers: four gates of a lane side by side in the bytes of a dword ----
// parity of every byte in bit 0 of that byte (DomainGF2::reconstruct before the smear, gf2/domain.rs:47-63)
__device__ __forceinline__ uint32_t par4(uint32_t x) {
    x ^= x >> 4;
    x ^= x >> 2;
    x ^= x >> 1;
    return x & 0x01010101u;
}
// 0/1 per byte -> 0x00/0xFF per byte
__device__ __forceinline__ uint32_t smear4(uint32_t x) { return (x << 8) - x; }
__device__ __forceinline__ uint32_t pack4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    return (b0 | (b1 << 8)) | ((b2 | (b3 << 8)) << 16);
}
// 8 / 4 bytes starting at an arbitrarily aligned address: aligned dwords and funnel shifts
__device__ __forceinline__ uint2 load8_unaligned(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
    return make_uint2(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh));
}
__device__ __forceinline__ uint32_t load4_unaligned(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    return __builtin_amdgcn_alignbyte(q[1], q[0], sh);
}
// bytes [lo, hi) of the dword w to p + lo .. p + hi (p dword-aligned): one dword store when all four are wanted
__device__ __forceinline__ void store_part(uint8_t* p, uint32_t w, int lo, int hi) {
    if (lo <= 0 && hi >= 4) {
        *(uint32_t*)p = w;
    } else {
        // (a fixed number of predicated stores: a loop of stores with a run-time trip count leaves the compiler's
        // s_waitcnt bookkeeping without a bound and every later wait becomes vmcnt(0) -- the end of all prefetching)
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k >= lo && k < hi) p[k] = (uint8_t)(w >> (8 * k));
    }
}
// workgroup barrier that orders LDS only: __syncthreads() also drains every outstanding global load (vmcnt(0)), i.e. the
// prefetch ring, once per level
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ------------------------------------------------------------------------------------------------
// Both kernels walk a level's segments wavefront by wavefront (wave w takes REP_BATCH consecutive segments starting at
// seg0 + w * REP_BATCH, then 16 * REP_BATCH further on, ...), one LDS-only workgroup barrier per level.  A segment step is
// a chain  header -> operand records / masks -> LDS gathers -> arithmetic -> stores, and with one workgroup per CU (the
// LDS is full of wires) there are only four wavefronts per SIMD to hide it.  Measured on the 10^7-gate circuit
// (tools/rep_time.py; DESIGN.md "Rep-sliced path"):
//   one segment per trip                                      3.9 ms interpreter, 3.1 ms cleartext pass (SQ_WAIT_ANY 69 %)
//   software prefetch into a register ring, 2 and 4 deep      no gain: with memory operations inside the step's branches
//                                                             the compiler puts s_waitcnt vmcnt(0) in front of every use
//   a prefetcher wavefront pulling the next level into L2     slower (4.7 ms): one consumer less, L2 hits no faster
//   (fire-and-forget LDS-DMA loads into a sink)
//   REP_BATCH = 4 segments per trip (headers, then records /  3.1 ms + 3.1 ms -- what is built here; 2: the same, 8: spills
//   masks, then gathers of the whole batch issued together)
//   the same without stores, mask loads and random gathers    2.0 ms: the skeleton itself is the cost
// Records and operand-value words have a fixed stride per segment (REP_SEG_RECS, 64): their addresses need no header.
// ------------------------------------------------------------------------------------------------
#ifndef REP_BATCH
#define REP_BATCH 4
#endif
constexpr uint32_t REP_WAVES = 16;

struct SegData {
    uint4 r0, r1;         // the lane's four operand records
    uint32_t w0, w1, w2;  // Mul / Input: the aligned dwords around the lane's mask bytes
    uint32_t vw;          // Mul: operand values (k_rep_clear)
};
// byte offset of the lane's first mask inside the repetition's mask array (may be a few bytes negative: front slack)
__device__ __forceinline__ int64_t seg_mask_off(const RepSeg& s, uint32_t lane) {
    const int i0 = (int)(4 * lane) - (int)s.off;
    return s.kind == RS_MUL ? (int64_t)s.m0 + 2 * (int64_t)i0 : (int64_t)s.m0 + i0;
}
template <bool CLEAR>
__device__ __forceinline__ void seg_fetch(const RepRec* __restrict__ recs, const uint32_t* __restrict__ vbits, const uint8_t* masks, const RepSeg& s,
                                          uint32_t si, uint32_t lane, SegData& d) {
    if (s.kind == RS_NONE) return;
    if (s.kind != RS_INPUT) {
        const uint4* rp = (const uint4*)(recs + (size_t)si * REP_SEG_RECS + 4 * lane);
        d.r0 = rp[0];
        d.r1 = rp[1];
    }
    if (!CLEAR) {
        if (s.kind == RS_MUL || s.kind == RS_INPUT) {
            const uintptr_t a = (uintptr_t)(masks + seg_mask_off(s, lane));
            const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
            d.w0 = q[0];
            d.w1 = q[1];
            d.w2 = q[2];
        }
        if (s.kind == RS_MUL) d.vw = vbits[(size_t)si * 64 + lane];
    }
}

// One segment.  CLEAR: cleartext values (one byte per slot), the operand-value words of the Mul segments, the AssertZero
// check; otherwise the interpreter proper on the repetition's mask bytes.
// the LDS gathers of a step, the same for every kind (a segment without records reads the zero slot): issued for
// all segments of a wave's batch before any of them is used
__device__ __forceinline__ void seg_gather(const uint8_t* lds, const SegData& d, uint32_t ga[4], uint32_t gb[4]) {
    const uint32_t ia[4] = {d.r0.x, d.r0.z, d.r1.x, d.r1.z}, ib[4] = {d.r0.y, d.r0.w, d.r1.y, d.r1.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ga[k] = lds[ia[k] & 0x7FFFFFFFu];
        gb[k] = lds[ib[k] & 0x7FFFFFFFu];
    }
}

template <bool CLEAR>
__device__ __forceinline__ void seg_process(uint8_t* lds, const RepSeg& s, uint32_t si, const SegData& d, const uint32_t ga[4], const uint32_t gb[4],
                                            uint32_t lane, const uint8_t* masks, uint8_t* on, uint8_t* pre, const uint8_t* __restrict__ wit,
                                            uint32_t* __restrict__ vbits_out, int* __restrict__ err) {
    if (s.kind == RS_NONE) return;
    const int i0 = (int)(4 * lane) - (int)s.off;  // gate index of the lane's byte 0; bytes k with 0 <= i0 + k < count are real
    if (i0 >= (int)s.count) return;
    const int lo = -i0, hi = (int)s.count - i0;  // the real bytes of this lane: [max(lo, 0), min(hi, 4))
    const uint32_t ia[4] = {d.r0.x, d.r0.z, d.r1.x, d.r1.z}, ib[4] = {d.r0.y, d.r0.w, d.r1.y, d.r1.w};
    uint32_t* dst = (uint32_t*)(lds + s.dst0 + 4 * lane);
    if (CLEAR) {
        if (s.kind == RS_INPUT) {
            uint32_t out4 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k >= lo && k < hi) out4 |= (wit[s.x0 + i0 + k] ? 1u : 0u) << (8 * k);
            *dst = out4;
            return;
        }
        const uint32_t* va = ga;
        const uint32_t* vb = gb;  // (AssertZero records have b = the zero slot)
        const uint32_t A = pack4(va[0], va[1], va[2], va[3]) ^ pack4(ia[0] >> 31, ia[1] >> 31, ia[2] >> 31, ia[3] >> 31);
        const uint32_t B = pack4(vb[0], vb[1], vb[2], vb[3]) ^ pack4(ib[0] >> 31, ib[1] >> 31, ib[2] >> 31, ib[3] >> 31);
        if (s.kind == RS_MUL) {
            vbits_out[(size_t)si * 64 + lane] = A | (B << 1);
            *dst = A & B;
        } else if (s.kind == RS_XOR) {
            *dst = A ^ B;
        } else {
            // AssertZero on a wire that is not zero (transcript/prover.rs:221-228 panics)
            uint32_t bad = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k >= lo && k < hi) bad |= (A >> (8 * k)) & 1u;
            if (bad) atomicOr(err, RV_E_WITNESS_INVALID);
        }
        return;
    }
    const uint32_t msh = (uint32_t)((uintptr_t)(masks + seg_mask_off(s, lane)) & 3);
    if (s.kind == RS_MUL) {
        const uint32_t* ma = ga;
        const uint32_t* mb = gb;
        // lambda_ab, lambda_new of the lane's four gates: 8 mask bytes from m0 + 2 i0
        const uint32_t mk0 = __builtin_amdgcn_alignbyte(d.w1, d.w0, msh), mk1 = __builtin_amdgcn_alignbyte(d.w2, d.w1, msh);
        const uint32_t MA = pack4(ma[0], ma[1], ma[2], ma[3]), MB = pack4(mb[0], mb[1], mb[2], mb[3]);
        const uint32_t LAB = __builtin_amdgcn_perm(mk1, mk0, 0x06040200u), LNEW = __builtin_amdgcn_perm(mk1, mk0, 0x07050301u);
        const uint32_t RA = par4(MA), RB = par4(MB), RAB = par4(LAB);
        const uint32_t CA = (d.vw & 0x01010101u) ^ RA, CB = ((d.vw >> 1) & 0x01010101u) ^ RB;  // corr = value ^ recon(mask)
        const uint32_t D = (RA & RB) ^ RAB;                                                    // single.rs:38-45
        const uint32_t S = (MB & smear4(CA)) ^ (MA & smear4(CB)) ^ LAB ^ LNEW;                 // single.rs:56-61
        *dst = LNEW;
        store_part(on + (s.eo0 - s.off) + 4 * lane, S, lo, hi);
        // the preprocessing bytes sit at ep0 + i: the same dword grid as the online bytes only if ep0 = eo0 (mod 4)
        const uint32_t D4 = smear4(D);
        if (((s.ep0 - s.eo0) & 3u) == 0) {
            store_part(pre + (s.ep0 - s.off) + 4 * lane, D4, lo, hi);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k >= lo && k < hi) pre[(int64_t)s.ep0 + i0 + k] = (uint8_t)(D4 >> (8 * k));
        }
    } else if (s.kind == RS_XOR) {
        *dst = pack4(ga[0] ^ gb[0], ga[1] ^ gb[1], ga[2] ^ gb[2], ga[3] ^ gb[3]);
    } else if (s.kind == RS_INPUT) {
        // prover.rs:181-199: mask = next(), corr = witness - recon(mask), hashed (and recorded) as a 0x00/0xFF byte
        const uint32_t lam4 = __builtin_amdgcn_alignbyte(d.w1, d.w0, msh);
        uint32_t w4 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k >= lo && k < hi) w4 |= (wit[s.x0 + i0 + k] ? 1u : 0u) << (8 * k);
        *dst = lam4;
        store_part(on + (s.eo0 - s.off) + 4 * lane, smear4(w4 ^ par4(lam4)), lo, hi);
    } else {
        // AssertZero: transcript.reconstruct(mask) hashes and records the share (prover.rs:221-228); the value check
        // itself is the cleartext pass's
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k >= lo && k < hi) on[s.eo0 + i0 + k] = (uint8_t)ga[k];
    }
}

// CLEAR = true: the cleartext pass (one workgroup; P.vbits is written).  It only depends on the witness, so it runs
// next to the mask generator.  CLEAR = false: the interpreter, blockIdx.x = repetition.
template <bool CLEAR>
__global__ __launch_bounds__(1024) void k_rep(RepParams P, uint32_t* __restrict__ vbits_out, int* __restrict__ err) {
    extern __shared__ uint8_t lds[];
    const uint32_t rep = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* masks = CLEAR ? nullptr : P.masks + (size_t)rep * P.mask_stride;
    uint8_t* on = CLEAR ? nullptr : P.on + (size_t)rep * P.on_stride;
    uint8_t* pre = CLEAR ? nullptr : P.pre + (size_t)rep * P.pre_stride;
    if (threadIdx.x == 0) *(uint32_t*)lds = 0;  // the zero row
    __syncthreads();
    for (uint32_t l = 0; l < P.n_levels; l++) {
        const RepLevel lv = P.levels[l];
        {
            // REP_BATCH consecutive segments per trip: their headers, then their records / masks, then their LDS gathers are
            // each issued together, so a trip pays the header -> data -> gather chain once for the whole batch (there are only
            // four wavefronts per SIMD to hide it otherwise)
            for (uint32_t s0 = lv.seg0 + wave * REP_BATCH; s0 < lv.seg1; s0 += REP_WAVES * REP_BATCH) {
                RepSeg h[REP_BATCH];
                SegData d[REP_BATCH];
                uint32_t ga[REP_BATCH][4], gb[REP_BATCH][4];
#pragma unroll
                for (uint32_t j = 0; j < REP_BATCH; j++) {
                    if (s0 + j < lv.seg1)
                        h[j] = P.segs[s0 + j];
                    else
                        h[j].kind = RS_NONE, h[j].count = 0, h[j].off = 0;
                }
#pragma unroll
                for (uint32_t j = 0; j < REP_BATCH; j++) {
                    d[j] = SegData{};
                    seg_fetch<CLEAR>(P.recs, P.vbits, masks, h[j], s0 + j, lane, d[j]);
                }
#pragma unroll
                for (uint32_t j = 0; j < REP_BATCH; j++) seg_gather(lds, d[j], ga[j], gb[j]);
#pragma unroll
                for (uint32_t j = 0; j < REP_BATCH; j++)
                    seg_process<CLEAR>(lds, h[j], s0 + j, d[j], ga[j], gb[j], lane, masks, on, pre, P.wit, vbits_out, err);
            }
        }
        lds_barrier();
    }
}

void launch_rep_clear(hipStream_t st, const RepLevel* d_levels, uint32_t n_levels, const RepSeg* d_segs, const RepRec* d_recs, const uint8_t* d_wit,
                      uint32_t* d_vbits, int* d_err, uint32_t lds_slots) {
    static bool attr = [] {
        const int want = (int)std::min<size_t>(160 * 1024, device_lds_limit()) - 1024;
        (void)hipFuncSetAttribute((const void*)k_rep<true>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
        (void)hipFuncSetAttribute((const void*)k_rep<false>, hipFuncAttributeMaxDynamicSharedMemorySize, want);
        return true;
    }();
    (void)attr;
    RepParams P{};
    P.levels = d_levels;
    P.segs = d_segs;
    P.recs = d_recs;
    P.wit = d_wit;
    P.n_levels = n_levels;
    hipLaunchKernelGGL(k_rep<true>, dim3(1), dim3(1024), lds_slots, st, P, d_vbits, d_err);
}

void launch_rep_interp(hipStream_t st, const RepParams& P, uint32_t R, uint32_t lds_slots) {
    hipLaunchKernelGGL(k_rep<false>, dim3(R), dim3(1024), lds_slots, st, P, (uint32_t*)nullptr, (int*)nullptr);
}

// ------------------------------------------------------------------------------------------------
// Openings from rep-major transcripts (Pack / PackSelected, gf2/share.rs:87-149, gf2/recon.rs:189-239): 8 items per
// byte MSB-first, n_items / 8 + 1 bytes.  kind 0: the omitted player's bit of a share byte; 1: a 0x00/0xFF byte.
// blockIdx.y = opened repetition (OnlineList), thread = output byte.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rep_open(const uint8_t* __restrict__ stream, uint64_t stride, const uint32_t* __restrict__ rows,
                                                  uint64_t n_items, int kind, const OnlineList* __restrict__ ol, const uint8_t* __restrict__ omit,
                                                  const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ out) {
    const uint32_t k = blockIdx.y;
    if (k >= ol->n) return;
    const uint32_t r = ol->rep[k];
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_bytes = n_items / 8 + 1;
    if (j >= n_bytes) return;
    const uint32_t sh = kind == 0 ? 7u - (omit[r] & 7u) : 0u;
    const uint8_t* src = stream + (size_t)r * stride;
    uint32_t acc = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint64_t it = 8 * j + t;
        if (it < n_items) {
            const uint64_t e = rows ? rows[it] : it;
            acc |= (((uint32_t)src[e] >> sh) & 1u) << (7 - t);
        }
    }
    out[dst_off[r] + j] = (uint8_t)acc;
}

void launch_rep_open(hipStream_t st, const uint8_t* d_stream, uint64_t stride, const uint32_t* d_rows, uint64_t n_items, int kind,
                     const OnlineList* d_ol, const uint8_t* d_omit, const uint64_t* d_dst_off, uint8_t* d_out) {
    const uint64_t n_bytes = n_items / 8 + 1;
    hipLaunchKernelGGL(k_rep_open, dim3((unsigned)((n_bytes + 255) / 256), RV_ONLINE_REPS), dim3(256), 0, st, d_stream, stride, d_rows, n_items, kind,
                       d_ol, d_omit, d_dst_off, d_out);
}

}  // namespace rv
