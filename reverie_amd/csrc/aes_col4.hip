// AES-128-CTR share expansion, lane-distributed form ("col4") for gfx950.
//
// Replaces the same reference code as k_aes_gf2_masks (aes.hip): crypto/prg.rs:16-37 (PRG::gen), generator/batch.rs:13-40,
// generator/share.rs:54-65 (the ShareGen refill) and algebra/gf2/domain.rs:66-378 (batches_to_shares), all under
// /root/reference/src/ -- and writes the identical masks[(j*128 + b)*NQ + q] rows.
//
// k_aes_gf2_masks keeps a whole bitsliced state in ONE lane: 128 planes + a column of temporaries = 256 VGPRs, two wavefronts per
// SIMD, a compute unit per workgroup -- nothing else fits beside it, which is why every attempt to run the VALU-bound cipher
// next to the latency-bound interpreter failed (rounds 2 - 4).  Here a QUAD of lanes holds one state:
//   lane c of the quad = state column c = 4 bytes x 8 bit planes = 32 VGPRs (32 slots per word, as before)
//   SubBytes    lane-local: the 74-op cover of aes_sbox.inc, four times
//   MixColumns  lane-local: a column is a lane; plane by plane over the four rows, in place
//   ShiftRows   folded into AddRoundKey: the state is kept SHIFTED (y_i = ShiftRows(x_i); SubBytes commutes with it), so a round
//               ends with y[row r] = quad_perm_r(t[row r]) ^ srk[row r] -- ONE v_xor_b32_dpp per word, round keys stored pre-shifted
// 417 VALU instructions per lane and round = 104 per byte (the 128-plane form: 101), 80 VGPRs instead of 256.
// Output: the wavefront's lanes are (quad ql, column c) = 4*ql + c; a ds_bpermute per word regroups them as 16*c + ql so that
// sixteen consecutive lanes store 64 contiguous bytes of one mask row, like k_aes_gf2_masks (without it the stores cost 20 %).
#include <stdlib.h>

#include <algorithm>

#include "aes_col4_dev.h"
#include "internal.h"
#include "launch.h"

namespace rv {

// the key image of one group of 16 quad words, from the plane-major round keys of k_bitslice_rk (areas 0..10 = rk0..rk10,
// 11 / 12 = the first-round constants: internal.h RK_BYTES):
// img[((qg*11 + area)*8 + k)*64 + lane].row = plane k of a key byte of quad 16*qg + ql, lane = 4*ql + c:
//   areas 2..9  round key byte (row, (c + row) & 3): pre-shifted, the state is kept shifted
//   area 10     round key byte (row, c): the last AddRoundKey meets the state unshifted
//   area 1      K1 byte (row, (c + row) & 3): round-1 output with S(state bytes 13..15) taken as zero, shifted
//   area 0      .x only: rk0 byte 15 - c, the ONE byte of the lane's column that meets the counter (c < 3)
__global__ void k_rk_col4(const uint32_t* __restrict__ rk, uint32_t NQ, uint32_t* __restrict__ img) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (NQ / 16) * C4_IMG_U4 * 4;
    if (t >= n) return;
    const uint32_t row = t & 3, lane = (t >> 2) & 63, k = (t >> 8) & 7, area = (t >> 11) % C4_AREAS, qg = (t >> 11) / C4_AREAS;
    const uint32_t ql = lane >> 2, c = lane & 3;
    uint32_t ga = area, byte = 4 * ((c + row) & 3) + row;
    if (area == 10) byte = 4 * c + row;
    if (area == 1) ga = 12;
    if (area == 0) {
        ga = 11;
        byte = 15 - c;
        if (row != 0 || c == 3) {
            img[t] = 0;
            return;
        }
    }
    img[t] = rk[(size_t)(ga * 128 + 8 * byte + k) * NQ + 16 * qg + ql];
}

// A workgroup = 16 quad words (their key image in LDS for its lifetime) x a range of CTR blocks; a wavefront = one CTR block of
// the 16 quads per trip.  WAVES wavefronts of 80 registers: with 8 the workgroup leaves three quarters of every SIMD's register
// file and all of its other wavefront slots to whatever else is resident (the interpreter's level launches: api.hip, RV_OVERLAP).
constexpr int C4_WAVES = 8;
__global__ __launch_bounds__(C4_WAVES * 64) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_aes_gf2_masks_col4(
    const uint4* __restrict__ img, const uint32_t* __restrict__ keep, uint32_t NQ, uint64_t first_block, uint64_t n_blocks, uint32_t blocks_per_wg,
    uint32_t* __restrict__ masks) {
    constexpr int WAVES = C4_WAVES;
    extern __shared__ uint4 c4_lds[];  // C4_IMG_U4 (dynamic: a static 88 KiB would make the compiler size the register budget for 2 waves)
    const uint32_t n_qg = NQ / 16;
    // (the n_qg workgroups that write the four 64-byte pieces of the same rows placed on ONE XCD, so that the pieces meet in one L2:
    // measured, no difference -- the plain mapping stays)
    const uint32_t qg = blockIdx.x % n_qg;
    const uint64_t chunk = blockIdx.x / n_qg;
    {
        const uint4* src = img + (size_t)qg * C4_IMG_U4;
        constexpr uint32_t T = WAVES * 64, FULL = C4_IMG_U4 / T, REST = C4_IMG_U4 % T;
        uint4 v[FULL];
#pragma unroll
        for (uint32_t i = 0; i < FULL; i++) v[i] = src[threadIdx.x + i * T];
#pragma unroll
        for (uint32_t i = 0; i < FULL; i++) c4_lds[threadIdx.x + i * T] = v[i];
        if (REST != 0 && threadIdx.x < REST) c4_lds[threadIdx.x + FULL * T] = src[threadIdx.x + FULL * T];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = lane & 3;
    const uint4* rkl = c4_lds + lane;
    // after the regrouping a lane stores for (column cs, quad qs)
    const uint32_t cs = lane >> 4, qs = qg * 16 + (lane & 15);
    const uint32_t from = 4 * (4 * (lane & 15) + cs);  // ds_bpermute address of source lane 4*ql + c
    const uint32_t kp = keep ? keep[qs] : 0xFFFFFFFFu;
    const uint64_t j_lo = chunk * blocks_per_wg;
    const uint64_t j_hi = (j_lo + blocks_per_wg < n_blocks) ? j_lo + blocks_per_wg : n_blocks;
    for (uint64_t jl = j_lo + wave; jl < j_hi; jl += WAVES) {
        const uint32_t j = (uint32_t)(first_block + jl);
        uint32_t s[32];
        c4_rounds_0_1(j, c, rkl, s);
#pragma unroll 1
        for (int r = 2; r < 10; r++) c4_round(s, rkl + r * 8 * 64);
        // last round: SubBytes (the state is shifted already), AddRoundKey
#pragma unroll
        for (int r = 0; r < 4; r++) {
            c4_sbox8(s[8 * r + 7], s[8 * r + 6], s[8 * r + 5], s[8 * r + 4], s[8 * r + 3], s[8 * r + 2], s[8 * r + 1], s[8 * r + 0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // keystream bit order is MSB-first inside each byte (gf2/domain.rs: share 8i+j <- bit 7-j of byte i): byte 4*cs + r, plane k
        // -> mask index 8*(4*cs + r) + 7 - k
        uint32_t* out = masks + ((size_t)jl * 128 + 32 * cs) * NQ + qs;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 kv = rkl[(10 * 8 + k) * 64];
            const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)(s[8 * r + k] ^ kw[r]));
                // nontemporal: the rows are read once, levels later (beside the level launches: 5.20 -> 5.13 ms per proof)
                __builtin_nontemporal_store(o & kp, &out[(size_t)(8 * r + (7 - k)) * NQ]);
            }
        }
    }
}

bool aes_col4_supports(uint32_t NQ) { return NQ % 16 == 0; }
size_t aes_col4_image_bytes(uint32_t NQ) { return (size_t)(NQ / 16) * C4_LDS_BYTES; }

void launch_rk_col4(hipStream_t st, const uint32_t* d_rk, uint32_t NQ, uint32_t* d_img) {
    const uint32_t n = (NQ / 16) * C4_IMG_U4 * 4;
    hipLaunchKernelGGL(k_rk_col4, dim3((n + 255) / 256), dim3(256), 0, st, d_rk, NQ, d_img);
}

void launch_aes_gf2_masks_col4(hipStream_t st, const uint32_t* d_img, const uint32_t* d_keep, uint32_t NQ, uint64_t first_block, uint64_t n_blocks,
                               uint32_t* d_masks) {
    if (!n_blocks) return;
    static const uint64_t target_wgs = [] {
        if (const char* e = getenv("RV_AES_WGS")) return (uint64_t)std::max(atoi(e), 1);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint64_t)cus;
    }();
    // (per device: the attribute belongs to the function ON the current device)
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute((const void*)k_aes_gf2_masks_col4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4_LDS_BYTES);
        raised[dev] = true;
    }
    const uint32_t n_qg = NQ / 16;
    uint64_t per = (n_blocks * n_qg + target_wgs - 1) / target_wgs;
    per = (per + C4_WAVES - 1) / C4_WAVES * C4_WAVES;
    const uint64_t chunks = (n_blocks + per - 1) / per;
    hipLaunchKernelGGL(k_aes_gf2_masks_col4, dim3((unsigned)(chunks * n_qg)), dim3(C4_WAVES * 64), C4_LDS_BYTES, st, (const uint4*)d_img, d_keep, NQ, first_block,
                       n_blocks, (uint32_t)per, d_masks);
}

}  // namespace rv
