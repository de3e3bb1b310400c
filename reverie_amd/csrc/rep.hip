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

__device__ __forceinline__ uint32_t par8(uint32_t x) { return __builtin_popcount(x & 0xFFu) & 1u; }

// ------------------------------------------------------------------------------------------------
// Cleartext evaluation (one workgroup, one byte per slot): the value bits of every Mul's operands, the witness check
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rep_clear(const RepLevel* __restrict__ levels, uint32_t n_levels, const RepSeg* __restrict__ segs,
                                                    const RepRec* __restrict__ recs, const uint8_t* __restrict__ wit, uint8_t* __restrict__ vbits,
                                                    int* __restrict__ err) {
    extern __shared__ uint8_t v[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) v[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t l = 0; l < n_levels; l++) {
        const RepLevel lv = levels[l];
        for (uint32_t si = lv.seg0 + wave; si < lv.seg1; si += 16) {
            const RepSeg s = segs[si];
            const uint32_t k0 = 4 * lane;
            if (k0 >= s.count) continue;
            const uint32_t nk = s.count - k0 < 4 ? s.count - k0 : 4;
            uint32_t out4 = 0, vb = 0;
            for (uint32_t k = 0; k < nk; k++) {
                uint32_t val = 0;
                if (s.kind == RS_INPUT) {
                    val = wit[s.x0 + k0 + k] ? 1u : 0u;
                } else {
                    const RepRec r = recs[s.first + k0 + k];
                    const uint32_t va = v[r.a & 0x7FFFFFFFu] ^ (r.a >> 31);
                    if (s.kind == RS_MUL) {
                        const uint32_t vbb = v[r.b & 0x7FFFFFFFu] ^ (r.b >> 31);
                        val = va & vbb;
                        vb |= (va | (vbb << 1)) << (2 * k);
                    } else if (s.kind == RS_XOR) {
                        val = va ^ v[r.b & 0x7FFFFFFFu];
                    } else if (va) {  // AssertZero on a wire that is not zero (transcript/prover.rs:221-228 panics)
                        atomicOr(err, RV_E_WITNESS_INVALID);
                        if (atomicAdd(err + 1, 1) == 0) { err[2] = (int)l; err[3] = (int)si; err[4] = (int)(k0 + k); err[5] = (int)r.a; err[6] = (int)v[r.a & 0x7FFFFFFFu]; }
                    }
                }
                out4 |= val << (8 * k);
            }
            if (s.kind == RS_MUL) vbits[s.vb0 + lane] = (uint8_t)vb;
            if (s.kind != RS_ASSERT) *(uint32_t*)(v + s.dst0 + k0) = out4;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The interpreter: blockIdx.x = repetition
// ------------------------------------------------------------------------------------------------
// 8 bytes starting at an arbitrarily aligned address: three aligned dwords, two funnel shifts
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
// up to four bytes to an arbitrarily aligned address
__device__ __forceinline__ void store_bytes(uint8_t* p, uint32_t w, uint32_t n) {
    if (n == 4 && ((uintptr_t)p & 3) == 0) {
        *(uint32_t*)p = w;
    } else {
        for (uint32_t k = 0; k < n; k++) p[k] = (uint8_t)(w >> (8 * k));
    }
}


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
            const uint32_t k0 = 4 * lane;
            if (k0 >= s.count) continue;
            const uint32_t nk = s.count - k0 < 4 ? s.count - k0 : 4;
            if (s.kind == RS_MUL) {
                const uint4* rp = (const uint4*)(P.recs + s.first + k0);  // records are padded to whole groups of four
                const uint4 r0 = rp[0], r1 = rp[1];
                const uint32_t vb = P.vbits[s.vb0 + lane];
                const uint2 mk = load8_unaligned(masks + s.m0 + 2 * k0);  // lambda_ab, lambda_new of the lane's four gates
                const uint32_t ia[4] = {r0.x, r0.z, r1.x, r1.z}, ib[4] = {r0.y, r0.w, r1.y, r1.w};
                uint32_t ma[4], mb[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    ma[k] = lds[ia[k] & 0x7FFFFFFFu];
                    mb[k] = lds[ib[k] & 0x7FFFFFFFu];
                }
                uint32_t s4 = 0, d4 = 0, new4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t pair = (k < 2 ? mk.x >> (16 * k) : mk.y >> (16 * (k - 2))) & 0xFFFFu;
                    const uint32_t lab = pair & 0xFFu, lnew = pair >> 8;
                    const uint32_t ra = par8(ma[k]), rb = par8(mb[k]), rab = par8(lab);
                    const uint32_t ca = ((vb >> (2 * k)) & 1u) ^ ra, cb = ((vb >> (2 * k + 1)) & 1u) ^ rb;  // corr = value ^ recon(mask)
                    const uint32_t delta = (ra & rb) ^ rab;                                                // single.rs:38-45
                    const uint32_t sh = ((mb[k] & (0u - ca)) ^ (ma[k] & (0u - cb)) ^ lab ^ lnew) & 0xFFu;    // single.rs:56-61
                    s4 |= sh << (8 * k);
                    d4 |= (delta ? 0xFFu : 0u) << (8 * k);
                    new4 |= lnew << (8 * k);
                }
                *(uint32_t*)(lds + s.dst0 + k0) = new4;
                store_bytes(on + s.eo0 + k0, s4, nk);
                store_bytes(pre + s.ep0 + k0, d4, nk);
            } else if (s.kind == RS_XOR) {
                const uint4* rp = (const uint4*)(P.recs + s.first + k0);
                const uint4 r0 = rp[0], r1 = rp[1];
                const uint32_t ia[4] = {r0.x, r0.z, r1.x, r1.z}, ib[4] = {r0.y, r0.w, r1.y, r1.w};
                uint32_t new4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) new4 |= (uint32_t)(lds[ia[k] & 0x7FFFFFFFu] ^ lds[ib[k] & 0x7FFFFFFFu]) << (8 * k);
                *(uint32_t*)(lds + s.dst0 + k0) = new4;
            } else if (s.kind == RS_INPUT) {
                // prover.rs:181-199: mask = next(), corr = witness - recon(mask), hashed (and recorded) as a 0x00/0xFF byte
                const uint32_t lam4 = load4_unaligned(masks + s.m0 + k0);
                uint32_t c4 = 0;
                for (uint32_t k = 0; k < nk; k++) {
                    const uint32_t w = P.wit[s.x0 + k0 + k] ? 1u : 0u;
                    c4 |= ((w ^ par8(lam4 >> (8 * k))) ? 0xFFu : 0u) << (8 * k);
                }
                *(uint32_t*)(lds + s.dst0 + k0) = lam4;
                store_bytes(on + s.eo0 + k0, c4, nk);
            } else {
                // AssertZero: transcript.reconstruct(mask) hashes and records the share (prover.rs:221-228); the value
                // check itself is k_rep_clear's
                uint32_t m4 = 0;
                for (uint32_t k = 0; k < nk; k++) m4 |= (uint32_t)lds[P.recs[s.first + k0 + k].a & 0x7FFFFFFFu] << (8 * k);
                store_bytes(on + s.eo0 + k0, m4, nk);
            }
        }
        __syncthreads();
    }
}

void launch_rep_clear(hipStream_t st, const RepLevel* d_levels, uint32_t n_levels, const RepSeg* d_segs, const RepRec* d_recs, const uint8_t* d_wit,
                      uint8_t* d_vbits, int* d_err, uint32_t lds_slots) {
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
