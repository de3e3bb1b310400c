// Rep-sliced prover kernels for gfx950: one workgroup = ONE repetition, live wires in LDS (repprog.h).
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
        for (int k = lo < 0 ? 0 : lo; k < (hi > 4 ? 4 : hi); k++) p[k] = (uint8_t)(w >> (8 * k));
    }
}

// ------------------------------------------------------------------------------------------------
// Cleartext evaluation (one workgroup, one byte per slot).  Produces, for every Mul segment, one word per lane: bit
// 8k = value of operand a of the lane's gate k, bit 8k + 1 = value of operand b (constants included) -- what
// k_rep_interp needs to rebuild the public corrections -- and checks the AssertZero gates.  It only depends on the
// witness, so it runs next to the mask generator.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rep_clear(const RepLevel* __restrict__ levels, uint32_t n_levels, const RepSeg* __restrict__ segs,
                                                    const RepRec* __restrict__ recs, const uint8_t* __restrict__ wit, uint32_t* __restrict__ vbits,
                                                    int* __restrict__ err) {
    extern __shared__ uint8_t v[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *(uint32_t*)v = 0;
    __syncthreads();
    for (uint32_t l = 0; l < n_levels; l++) {
        const RepLevel lv = levels[l];
        for (uint32_t si = lv.seg0 + wave; si < lv.seg1; si += 16) {
            const RepSeg s = segs[si];
            const int i0 = (int)(4 * lane) - (int)s.off;  // gate index of the lane's byte 0
            if (i0 >= (int)s.count) continue;
            uint32_t out4 = 0;
            if (s.kind == RS_INPUT) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i0 + k >= 0 && i0 + k < (int)s.count) out4 |= (wit[s.x0 + i0 + k] ? 1u : 0u) << (8 * k);
                *(uint32_t*)(v + s.dst0 + 4 * lane) = out4;
                continue;
            }
            const uint4* rp = (const uint4*)(recs + s.first + 4 * lane);
            const uint4 r0 = rp[0], r1 = rp[1];
            const uint32_t ia[4] = {r0.x, r0.z, r1.x, r1.z}, ib[4] = {r0.y, r0.w, r1.y, r1.w};
            uint32_t va[4], vb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                va[k] = v[ia[k] & 0x7FFFFFFFu];
                vb[k] = s.kind == RS_ASSERT ? 0u : (uint32_t)v[ib[k] & 0x7FFFFFFFu];
            }
            const uint32_t A = pack4(va[0], va[1], va[2], va[3]) ^ pack4(ia[0] >> 31, ia[1] >> 31, ia[2] >> 31, ia[3] >> 31);
            const uint32_t B = pack4(vb[0], vb[1], vb[2], vb[3]) ^ pack4(ib[0] >> 31, ib[1] >> 31, ib[2] >> 31, ib[3] >> 31);
            if (s.kind == RS_MUL) {
                vbits[s.vb0 + lane] = A | (B << 1);
                *(uint32_t*)(v + s.dst0 + 4 * lane) = A & B;
            } else if (s.kind == RS_XOR) {
                *(uint32_t*)(v + s.dst0 + 4 * lane) = A ^ B;
            } else {
                // AssertZero on a wire that is not zero (transcript/prover.rs:221-228 panics)
                uint32_t bad = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i0 + k >= 0 && i0 + k < (int)s.count) bad |= (A >> (8 * k)) & 1u;
                if (bad) atomicOr(err, RV_E_WITNESS_INVALID);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The interpreter: blockIdx.x = repetition, 16 wavefronts deal a level's segments among themselves
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rep_interp(RepParams P) {
    extern __shared__ uint8_t lds[];
    const uint32_t rep = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* masks = P.masks + (size_t)rep * P.mask_stride;
    uint8_t* on = P.on + (size_t)rep * P.on_stride;
    uint8_t* pre = P.pre + (size_t)rep * P.pre_stride;
    if (threadIdx.x == 0) *(uint32_t*)lds = 0;  // the zero row
    __syncthreads();
    for (uint32_t l = 0; l < P.n_levels; l++) {
        const RepLevel lv = P.levels[l];
        for (uint32_t si = lv.seg0 + wave; si < lv.seg1; si += 16) {
            const RepSeg s = P.segs[si];
            const int i0 = (int)(4 * lane) - (int)s.off;  // gate index of the lane's byte 0; bytes k with 0 <= i0 + k < count are real
            if (i0 >= (int)s.count) continue;
            const int lo = -i0, hi = (int)s.count - i0;  // the real bytes of this lane: [max(lo, 0), min(hi, 4))
            if (s.kind == RS_MUL) {
                const uint4* rp = (const uint4*)(P.recs + s.first + 4 * lane);
                const uint4 r0 = rp[0], r1 = rp[1];
                const uint32_t vw = P.vbits[s.vb0 + lane];
                // lambda_ab, lambda_new of the lane's four gates: 8 mask bytes from m0 + 2 i0
                const uint2 mk = load8_unaligned(masks + (int64_t)s.m0 + 2 * (int64_t)i0);
                const uint32_t ia[4] = {r0.x, r0.z, r1.x, r1.z}, ib[4] = {r0.y, r0.w, r1.y, r1.w};
                uint32_t ma[4], mb[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    ma[k] = lds[ia[k] & 0x7FFFFFFFu];
                    mb[k] = lds[ib[k] & 0x7FFFFFFFu];
                }
                const uint32_t MA = pack4(ma[0], ma[1], ma[2], ma[3]), MB = pack4(mb[0], mb[1], mb[2], mb[3]);
                const uint32_t LAB = __builtin_amdgcn_perm(mk.y, mk.x, 0x06040200u), LNEW = __builtin_amdgcn_perm(mk.y, mk.x, 0x07050301u);
                const uint32_t RA = par4(MA), RB = par4(MB), RAB = par4(LAB);
                const uint32_t CA = (vw & 0x01010101u) ^ RA, CB = ((vw >> 1) & 0x01010101u) ^ RB;  // corr = value ^ recon(mask)
                const uint32_t D = (RA & RB) ^ RAB;                                                // single.rs:38-45
                const uint32_t S = (MB & smear4(CA)) ^ (MA & smear4(CB)) ^ LAB ^ LNEW;             // single.rs:56-61
                *(uint32_t*)(lds + s.dst0 + 4 * lane) = LNEW;
                store_part(on + (s.eo0 - s.off) + 4 * lane, S, lo, hi);
                // the preprocessing bytes sit at ep0 + i: the same dword grid as the online bytes only if ep0 = eo0 (mod 4)
                const uint32_t D4 = smear4(D);
                const uint32_t sp = (s.ep0 - s.eo0) & 3u;
                if (sp == 0) {
                    store_part(pre + (s.ep0 - s.off) + 4 * lane, D4, lo, hi);
                } else {
                    for (int k = lo < 0 ? 0 : lo; k < (hi > 4 ? 4 : hi); k++) pre[(int64_t)s.ep0 + i0 + k] = (uint8_t)(D4 >> (8 * k));
                }
            } else if (s.kind == RS_XOR) {
                const uint4* rp = (const uint4*)(P.recs + s.first + 4 * lane);
                const uint4 r0 = rp[0], r1 = rp[1];
                const uint32_t ia[4] = {r0.x, r0.z, r1.x, r1.z}, ib[4] = {r0.y, r0.w, r1.y, r1.w};
                uint32_t x[4];
#pragma unroll
                for (int k = 0; k < 4; k++) x[k] = (uint32_t)(lds[ia[k] & 0x7FFFFFFFu] ^ lds[ib[k] & 0x7FFFFFFFu]);
                *(uint32_t*)(lds + s.dst0 + 4 * lane) = pack4(x[0], x[1], x[2], x[3]);
            } else if (s.kind == RS_INPUT) {
                // prover.rs:181-199: mask = next(), corr = witness - recon(mask), hashed (and recorded) as a 0x00/0xFF byte
                const uint32_t lam4 = load4_unaligned(masks + (int64_t)s.m0 + i0);
                uint32_t w4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k >= lo && k < hi) w4 |= (P.wit[s.x0 + i0 + k] ? 1u : 0u) << (8 * k);
                *(uint32_t*)(lds + s.dst0 + 4 * lane) = lam4;
                store_part(on + (s.eo0 - s.off) + 4 * lane, smear4(w4 ^ par4(lam4)), lo, hi);
            } else {
                // AssertZero: transcript.reconstruct(mask) hashes and records the share (prover.rs:221-228); the value
                // check itself is k_rep_clear's
                for (int k = lo < 0 ? 0 : lo; k < (hi > 4 ? 4 : hi); k++) on[s.eo0 + i0 + k] = lds[P.recs[s.first + 4 * lane + k].a & 0x7FFFFFFFu];
            }
        }
        __syncthreads();
    }
}

void launch_rep_clear(hipStream_t st, const RepLevel* d_levels, uint32_t n_levels, const RepSeg* d_segs, const RepRec* d_recs, const uint8_t* d_wit,
                      uint32_t* d_vbits, int* d_err, uint32_t lds_slots) {
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void*)k_rep_clear, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_rep_interp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr;
    hipLaunchKernelGGL(k_rep_clear, dim3(1), dim3(1024), lds_slots, st, d_levels, n_levels, d_segs, d_recs, d_wit, d_vbits, d_err);
}

void launch_rep_interp(hipStream_t st, const RepParams& P, uint32_t R, uint32_t lds_slots) {
    hipLaunchKernelGGL(k_rep_interp, dim3(R), dim3(1024), lds_slots, st, P);
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
