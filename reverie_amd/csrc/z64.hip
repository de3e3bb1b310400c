// Z64 ring interpreter, transcript hashing and openings for gfx950.
//
// Replaces (all under /root/reference/src/):
//   algebra/z64/{share,recon,domain}.rs       wrapping u64 arithmetic per (rep, player)
//   interpreter/single.rs:25-157              Instance::step / op_mul instantiated at Z64
//   interpreter/combine.rs:19-36,132-219      recon_gf2_to_z64 and the Z64 half of B2A
//   transcript/{prover,verifier/*}.rs         the same transcript rules as GF(2)
//
// Lane mapping: one lane = two players of one repetition (16-byte accesses), 4 adjacent lanes = one
// repetition; reconstruct = a local add + 2-step shuffle-add inside the 4-lane group.  A gate occupies
// R*4 lanes (16 wavefronts at R = 256).
#include "b3.h"
#include "internal.h"

namespace rv {

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    const uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t recon32_(uint32_t t) {
    t ^= t >> 4;
    t ^= t >> 2;
    t ^= t >> 1;
    t &= 0x01010101u;
    return (t << 8) - t;
}

// Two players per lane: every row access is a 16-byte load/store (twice the bytes in flight per wavefront of the
// first, one-u64-per-lane version, which ran at 2.7 TB/s), a gate occupies R*4 lanes and the player sum is one local
// add plus two shuffle steps inside the 4-lane group of a repetition.
struct U2 {
    uint64_t x, y;
};
__device__ __forceinline__ U2 ld2(const uint64_t* p) {
    const ulonglong2 v = *(const ulonglong2*)p;
    return U2{v.x, v.y};
}
__device__ __forceinline__ void st2(uint64_t* p, U2 v) { *(ulonglong2*)p = make_ulonglong2(v.x, v.y); }
// transcript words are only 8-byte aligned (an odd number of 8-byte events may precede a 64-byte one)
__device__ __forceinline__ void st2_unaligned(uint64_t* p, U2 v) {
    p[0] = v.x;
    p[1] = v.y;
}
// DomainZ64::reconstruct (z64/domain.rs:53-61): wrapping sum over the 8 players
__device__ __forceinline__ uint64_t sum8(U2 v) {
    uint64_t t = v.x + v.y;
    t += shfl_xor64(t, 1);
    t += shfl_xor64(t, 2);
    return t;
}

// parity hook for DomainZ64::reconstruct (z64/domain.rs:53-61) on ShareZ64 values ([8 reps][8 players] u64 each): the
// interpreter's own lane mapping (two players per lane, four lanes per repetition) and its sum8
__global__ void k_hook_recon_z64(const uint64_t* __restrict__ shares, uint64_t n_reps_total, uint64_t* __restrict__ out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = tid >> 2;
    const U2 v = r < n_reps_total ? ld2(shares + 2 * tid) : U2{0, 0};
    const uint64_t s = sum8(v);
    if (r < n_reps_total && (tid & 3) == 0) out[r] = s;
}
void launch_hook_recon_z64(hipStream_t st, const uint64_t* d_shares, uint64_t n, uint64_t* d_out) {
    const uint64_t lanes = n * 8 * 4;
    if (n) hipLaunchKernelGGL(k_hook_recon_z64, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, d_shares, n * 8, d_out);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_interp64(const Gate64* __restrict__ gates, uint32_t lo, uint32_t hi, Interp64Params p) {
    const uint32_t S = p.R * 8;   // u64 per row
    const uint32_t S2 = p.R * 4;  // lanes per gate
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t l = tid % S2;
    const uint32_t r = l >> 2, pk = l & 3;  // this lane holds players 2*pk and 2*pk + 1 of repetition r
    const uint32_t worker = tid / S2, n_workers = (gridDim.x * blockDim.x) / S2;
    const uint32_t om = (MODE == MODE_VERIFY) ? p.omit[r] : 8u;
    const bool online = om < 8;  // online-verified repetition (MODE_VERIFY only)
    const bool mine = (om >> 1) == pk;  // the omitted player sits in this lane (slot om & 1)
    for (uint32_t gi = lo + worker; gi < hi; gi += n_workers) {
        const Gate64 g = gates[gi];
        uint64_t* dm = p.wmask + (size_t)g.dst * S + 2 * l;
        uint64_t* dc = p.wcorr + (size_t)g.dst * p.R + r;
        // an operand's mask row: its own wmask row, or the fresh PRG mask row that IS the wire's mask (Input / Random / Mul results)
        const uint64_t* am = ((g.am & G64_MASK_ROW) ? p.masks + (size_t)(g.am & ~G64_MASK_ROW) * S : p.wmask + (size_t)g.am * S) + 2 * l;
        const uint64_t* ac = p.wcorr + (size_t)g.a * p.R + r;
        const uint64_t* bm = ((g.bm & G64_MASK_ROW) ? p.masks + (size_t)(g.bm & ~G64_MASK_ROW) * S : p.wmask + (size_t)g.bm * S) + 2 * l;
        const uint64_t* bc = p.wcorr + (size_t)g.b * p.R + r;
        switch (g.op) {
        case G64_INPUT: {
            const U2 lam = ld2(p.masks + (size_t)g.m * S + 2 * l);
            uint64_t corr;
            if (MODE == MODE_PROVE)
                corr = p.wit[g.x] - sum8(lam);
            else
                corr = online ? p.sup_in[(size_t)g.x * p.sup_r + r] : 0;
            if (pk == 0) {
                *dc = corr;
                p.on[(size_t)r * p.on_words + g.eo] = corr;
            }
            break;
        }
        case G64_ADD: {
            const U2 x = ld2(am), y = ld2(bm);
            st2(dm, U2{x.x + y.x, x.y + y.y});
            if (pk == 0) *dc = *ac + *bc;
            break;
        }
        case G64_SUB: {
            const U2 x = ld2(am), y = ld2(bm);
            st2(dm, U2{x.x - y.x, x.y - y.y});
            if (pk == 0) *dc = *ac - *bc;
            break;
        }
        case G64_ADDC:
            st2(dm, ld2(am));
            if (pk == 0) *dc = *ac + g.imm;
            break;
        case G64_SUBC:
            st2(dm, ld2(am));
            if (pk == 0) *dc = *ac - g.imm;
            break;
        case G64_MULC: {
            const U2 x = ld2(am);
            st2(dm, U2{x.x * g.imm, x.y * g.imm});
            if (pk == 0) *dc = *ac * g.imm;
            break;
        }
        case G64_CONST:
            st2(dm, U2{0, 0});
            if (pk == 0) *dc = g.imm;
            break;
        case G64_RANDOM:
            if (pk == 0) *dc = 0;
            break;
        case G64_MUL: {
            const U2 lx = ld2(am), ly = ld2(bm);
            const U2 lab = ld2(p.masks + (size_t)g.m * S + 2 * l), lnew = ld2(p.masks + (size_t)(g.m + 1) * S + 2 * l);
            const uint64_t cx = *ac, cy = *bc;
            const uint64_t a = sum8(lx), b = sum8(ly), c = sum8(lab);
            uint64_t delta = a * b - c;
            U2 s{ly.x * cx + lx.x * cy + lab.x - lnew.x, ly.y * cx + lx.y * cy + lab.y - lnew.y};
            if (MODE == MODE_VERIFY && online) {
                delta = p.sup_corr[(size_t)g.xc * p.sup_r + r];
                if (mine) {
                    const uint64_t sup = p.sup_rec[(size_t)g.x * p.sup_r + r];
                    if (om & 1) s.y += sup; else s.x += sup;
                }
            }
            st2_unaligned(p.on + (size_t)r * p.on_words + g.eo + 2 * pk, s);
            uint64_t rec = sum8(s);
            if (MODE == MODE_VERIFY && !online) rec = 0;
            if (pk == 0) {
                p.pre[(size_t)r * p.pre_words + g.ep] = delta;
                *dc = rec + delta + cx * cy;
            }
            break;
        }
        case G64_ASSERT: {
            U2 m = ld2(am);
            if (MODE == MODE_VERIFY && online && mine) {
                const uint64_t sup = p.sup_rec[(size_t)g.x * p.sup_r + r];
                if (om & 1) m.y += sup; else m.x += sup;
            }
            st2_unaligned(p.on + (size_t)r * p.on_words + g.eo + 2 * pk, m);
            {
                const uint64_t v = sum8(m) + *ac;
                if (MODE == MODE_PROVE) {
                    if (v != 0 && pk == 0) atomicOr(p.err, RV_E_WITNESS_INVALID);
                } else if (online && v != 0 && pk == 0) {
                    atomicOr(p.err, RV_DEV_ZERO_CHECK);  // online.rs:175-177 (read by RV_VERIFY_STRICT only)
                }
            }
            break;
        }
        case G64_B2A: {
            // random 64-bit value shared bitwise in GF(2): bit k = recon(fresh gf2 mask m2+k)
            const uint32_t qw = r >> 2, sh = 24 - 8 * (r & 3);
            uint64_t zval = 0, zrec = 0;
            for (int k = 0; k < 64; k++) {
                const uint32_t w = recon32_(p.masks2[(size_t)(g.m2 + k) * p.NQ + qw]);
                zval |= (uint64_t)((w >> sh) & 1u) << k;
                // revealed sum bit k: bit-per-rep corr row of the k-th G_RECON output
                const uint32_t v = p.corr2[(size_t)(g.a + k) * (p.NQ >> 1) + (qw >> 1)];
                zrec |= (uint64_t)((v >> (4 * (qw & 1) + 3 - (r & 3))) & 1u) << k;
            }
            const U2 mu = ld2(p.masks + (size_t)g.m * S + 2 * l);
            uint64_t kappa = zval - sum8(mu);
            if (MODE == MODE_VERIFY && online) kappa = p.sup_corr[(size_t)g.xc * p.sup_r + r];
            st2(dm, U2{0 - mu.x, 0 - mu.y});
            if (pk == 0) {
                p.pre[(size_t)r * p.pre_words + g.ep] = kappa;
                *dc = zrec - kappa;
            }
            break;
        }
        default:
            break;
        }
    }
}

void launch_interp64(hipStream_t st, int mode, const Gate64* d_gates, uint32_t lo, uint32_t hi, const Interp64Params& p) {
    if (hi <= lo) return;
    const uint64_t S2 = (uint64_t)p.R * 4;  // lanes per gate
    const uint64_t want = (uint64_t)(hi - lo) * S2;
    uint64_t blocks = (want + 255) / 256;
    const uint64_t cap = ((uint64_t)8192 * 256 / S2) * S2 / 256;  // whole workers only
    if (blocks > cap) blocks = cap;
    if (mode == MODE_PROVE)
        hipLaunchKernelGGL(k_interp64<MODE_PROVE>, dim3((unsigned)blocks), dim3(256), 0, st, d_gates, lo, hi, p);
    else
        hipLaunchKernelGGL(k_interp64<MODE_VERIFY>, dim3((unsigned)blocks), dim3(256), 0, st, d_gates, lo, hi, p);
}

// ---- BLAKE3 over R contiguous little-endian streams: thread = (chunk, rep), chunk fastest ----
// Adjacent lanes hash adjacent 1 KiB chunks of the SAME stream, so a wavefront walks one contiguous 64 KiB region
// (the first version put the repetitions of one chunk in adjacent lanes: 64 MB apart, every 16-byte load a different
// DRAM page, 1.2 TB/s).  A lane fetches a whole 128-byte line (two blocks, eight 16-byte loads issued together) and
// then runs the two dependent compressions, so each line crosses L2 -> L1 once instead of eight times.
// stride_bytes: distance between two repetitions' streams (>= n_bytes; the streaming prover hashes a prefix of each);
// chunk_base / root_ok: see B_k_b3_chunks
__global__ __launch_bounds__(256) void k_b3_chunks_contig(const uint32_t* __restrict__ streams, uint64_t stride_bytes, uint64_t n_bytes, uint32_t R,
                                                          uint64_t n_chunks, uint32_t* __restrict__ cvs, uint64_t chunk_base, uint32_t root_ok) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t c = tid % n_chunks;
    const uint32_t r = (uint32_t)(tid / n_chunks);
    if (r >= R) return;
    const uint64_t b0 = c * 1024;
    const uint64_t len = (n_bytes - b0 < 1024) ? (n_bytes - b0) : 1024;
    const uint32_t nblk = len == 0 ? 1 : (uint32_t)((len + 63) / 64);
    const uint32_t* src = streams + ((size_t)r * stride_bytes + b0) / 4;
    uint32_t cv[8];
    b3::iv(cv);
    uint32_t b = 0;
    if (len == 1024 && n_chunks > 1) {  // the common case: 16 full blocks, two per step
        for (; b < 16; b += 2) {
            const uint4* s4 = (const uint4*)(src + 16 * b);
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = s4[k];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t m[16], o[8];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    m[4 * k] = v[4 * h + k].x;
                    m[4 * k + 1] = v[4 * h + k].y;
                    m[4 * k + 2] = v[4 * h + k].z;
                    m[4 * k + 3] = v[4 * h + k].w;
                }
                const uint32_t flags = (b + h == 0 ? b3::CHUNK_START : 0u) | (b + h == 15 ? b3::CHUNK_END : 0u);
                b3::compress<false>(cv, m, c + chunk_base, 64, flags, o);
#pragma unroll
                for (int k = 0; k < 8; k++) cv[k] = o[k];
            }
        }
    }
    for (; b < nblk; b++) {
        const uint32_t blen = (b + 1 < nblk) ? 64u : (uint32_t)(len - 64ull * b);
        uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) | (b + 1 == nblk ? b3::CHUNK_END : 0u);
        if (b + 1 == nblk && n_chunks == 1 && root_ok) flags |= b3::ROOT;
        uint32_t m[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t w = (4u * k < blen) ? src[16 * b + k] : 0u;
            if (4u * k < blen && 4u * k + 4 > blen) w &= (1u << (8 * (blen - 4u * k))) - 1u;  // byte streams: the last word may be partial
            m[k] = w;
        }
        uint32_t o[8];
        b3::compress<false>(cv, m, c + chunk_base, blen, flags, o);
#pragma unroll
        for (int k = 0; k < 8; k++) cv[k] = o[k];
    }
    uint32_t* dst = cvs + ((size_t)c * R + r) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = cv[k];
}

void launch_b3_contig_chunks(hipStream_t st, const uint64_t* d_streams, uint64_t stride_words, uint64_t n_words, uint32_t R, uint32_t* d_cv,
                             uint64_t chunk_base, uint32_t root_ok) {
    const uint64_t n_bytes = n_words * 8;
    const uint64_t n = n_bytes == 0 ? 1 : (n_bytes + 1023) / 1024;
    const uint64_t threads = n * R;
    hipLaunchKernelGGL(k_b3_chunks_contig, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_streams,
                       stride_words * 8, n_bytes, R, n, d_cv, chunk_base, root_ok);
}

uint32_t launch_b3_contig(hipStream_t st, const uint64_t* d_streams, uint64_t n_words, uint32_t R, uint32_t* d_cv_a, uint32_t* d_cv_b,
                      uint32_t* d_digest) {
    const uint64_t n_bytes = n_words * 8;
    const uint64_t n = n_bytes == 0 ? 1 : (n_bytes + 1023) / 1024;
    launch_b3_contig_chunks(st, d_streams, n_words, n_words, R, d_cv_a, 0, 1);
    return 1 + b3_reduce_tree(st, d_cv_a, d_cv_b, n, R, d_digest);  // launches
}

// ---- openings: 8 bytes LE per item (z64/share.rs:36-49, z64/recon.rs:45-66) ----
__global__ void k_extract64(const uint64_t* __restrict__ stream, uint64_t stride_words, const uint64_t* __restrict__ offs,
                            uint64_t n_items, int add_omit, uint32_t R, const uint8_t* __restrict__ omit,
                            const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t it = tid % n_items;
    const uint32_t r = (uint32_t)(tid / n_items);
    if (r >= R) return;
    const uint32_t om = omit[r];
    if (om >= 8) return;
    const uint64_t off = (offs ? offs[it] : it) + (add_omit ? om : 0);
    const uint64_t v = stream[(size_t)r * stride_words + off];
    uint8_t* d = out + dst_off[r] + 8 * it;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = (uint8_t)(v >> (8 * i));
}

// the same over the opened repetitions only (ol: the shard's OnlineList in device memory): thread = (rank among the opened, item) --
// 40 x n_items threads instead of R x n_items of which 216 in 256 returned at once (10^6-MUL circuit: opening phase 1.06 -> 0.75 ms)
__global__ void k_extract64_ol(const uint64_t* __restrict__ stream, uint64_t stride_words, const uint64_t* __restrict__ offs,
                               uint64_t n_items, int add_omit, const OnlineList* __restrict__ ol, const uint8_t* __restrict__ omit,
                               const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ out, uint32_t rep_min) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t it = tid % n_items;
    const uint32_t k = (uint32_t)(tid / n_items);
    if (k >= ol->n || k >= RV_ONLINE_REPS) return;
    const uint32_t r = ol->rep[k];
    if (r < rep_min) return;  // (Z64 early corrections: the host has this repetition's vector already)
    const uint32_t om = omit[r];
    const uint64_t off = (offs ? offs[it] : it) + (add_omit ? om : 0);
    const uint64_t v = stream[(size_t)r * stride_words + off];
    uint8_t* d = out + dst_off[r] + 8 * it;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = (uint8_t)(v >> (8 * i));
}

void launch_extract64(hipStream_t st, const uint64_t* d_stream, uint64_t stride_words, const uint64_t* d_offs, uint64_t n_items,
                      int add_omit, uint32_t R, const uint8_t* d_omit, const uint64_t* d_dst_off, uint8_t* d_out, const OnlineList* d_ol, uint32_t rep_min) {
    if (!n_items) return;
    if (d_ol) {
        const uint64_t threads = n_items * RV_ONLINE_REPS;
        hipLaunchKernelGGL(k_extract64_ol, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_stream, stride_words, d_offs,
                           n_items, add_omit, d_ol, d_omit, d_dst_off, d_out, rep_min);
        return;
    }
    const uint64_t threads = n_items * R;
    hipLaunchKernelGGL(k_extract64, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_stream, stride_words, d_offs,
                       n_items, add_omit, R, d_omit, d_dst_off, d_out);
}

// verifier: proof vectors -> dense [item][out_r] u64 (out_r = R, or the first 64 repetitions when no other is opened -- the
// verifier's slot order: a quarter of the bytes); items past a vector's end read as zero
// (z64/recon.rs:96-104, z64/share.rs:78-88 `unwrap_or([0u8; 8])`)
__global__ void k_unpack64(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ src_off,
                           const uint64_t* __restrict__ src_len, const uint8_t* __restrict__ omit, uint64_t n_items, uint32_t out_r,
                           uint64_t* __restrict__ out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t it = tid / out_r;
    const uint32_t r = (uint32_t)(tid % out_r);
    if (it >= n_items) return;
    uint64_t v = 0;
    if (omit[r] < 8 && (it + 1) * 8 <= src_len[r]) {
        const uint8_t* s = blob + src_off[r] + 8 * it;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= (uint64_t)s[i] << (8 * i);
    }
    out[it * out_r + r] = v;
}

void launch_unpack64(hipStream_t st, const uint8_t* d_blob, const uint64_t* d_src_off, const uint64_t* d_src_len,
                     const uint8_t* d_omit, uint64_t n_items, uint32_t R, uint64_t* d_out, uint32_t out_r) {
    if (!n_items) return;
    (void)R;
    const uint64_t threads = n_items * out_r;
    hipLaunchKernelGGL(k_unpack64, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_blob, d_src_off, d_src_len, d_omit,
                       n_items, out_r, d_out);
}

}  // namespace rv
