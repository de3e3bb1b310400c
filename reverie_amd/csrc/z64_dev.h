// Device helpers shared by the two Z64 prover level kernels (z64c4.hip: k_z64_c4, aes.hip: k_z64_fused).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rv {

// m ? a : b, bit by bit (m = all ones or zero per lane: selects on a lane-dependent bool compile to branches)
__device__ __forceinline__ uint32_t z4_sel(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xca); }
template <int CTRL>
__device__ __forceinline__ uint32_t z4_dpp(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }


// 16 bytes of a transcript stream: the streams are only 8-byte aligned (global_store_dwordx4 takes any dword-aligned address:
// tools/mb/unaligned16_mb.hip).  Nontemporal -- the transcripts are read again in the hash phase only (measured: no difference, 36.86 /
// 36.82 ms of level launches per 10^6 Mul in k_z64_c4; lambda_new, an operand a level later, stored nontemporal as well: 37.0)
typedef uint64_t z4_u64x2 __attribute__((ext_vector_type(2)));
typedef z4_u64x2 z4_u64x2_a8 __attribute__((aligned(8)));
__device__ __forceinline__ void z4_st16_stream(uint64_t* p, uint64_t a, uint64_t b) {
    const z4_u64x2 v = {a, b};
    __builtin_nontemporal_store(v, (z4_u64x2_a8*)p);
}

// A Mul's eight transcript words per repetition are 64 bytes of the repetition's stream, the wavefront's 64 repetitions 64 streams: with
// every lane storing its own 64 bytes in four pieces each store instruction touches 64 lines a quarter each (measured: 9.6 ms of a
// 44 ms proof against 1.6 ms for the same bytes stored contiguously).  A 4 x 4 transpose of the pieces inside the quad first: lane c
// then holds piece c of the quad's k-th stream in e[k], and an instruction stores WHOLE 64-byte segments, four lanes each.
// oq = the first lane's stream position, stride = u64 words between the four lanes' streams; mh / ml = all ones where c & 2 / c & 1.
__device__ __forceinline__ void z4_store_on_quad(uint64_t* oq, uint64_t stride, const uint64_t* w, uint32_t c, uint32_t mh, uint32_t ml) {
    uint32_t e[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        e[k][0] = (uint32_t)w[2 * k];
        e[k][1] = (uint32_t)(w[2 * k] >> 32);
        e[k][2] = (uint32_t)w[2 * k + 1];
        e[k][3] = (uint32_t)(w[2 * k + 1] >> 32);
    }
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {  // lanes c, c ^ 2: the low lane's e[k + 2] <-> the high lane's e[k]
            const uint32_t recv = z4_dpp<0x4E>(z4_sel(mh, e[k][d], e[k + 2][d]));
            e[k][d] = z4_sel(mh, recv, e[k][d]);
            e[k + 2][d] = z4_sel(mh, e[k + 2][d], recv);
        }
#pragma unroll
        for (int k = 0; k < 4; k += 2) {  // lanes c, c ^ 1: the low lane's e[k + 1] <-> the high lane's e[k]
            const uint32_t recv = z4_dpp<0xB1>(z4_sel(ml, e[k][d], e[k + 1][d]));
            e[k][d] = z4_sel(ml, recv, e[k][d]);
            e[k + 1][d] = z4_sel(ml, e[k + 1][d], recv);
        }
    }
    uint64_t* op = oq + 2 * c;
#pragma unroll
    for (int k = 0; k < 4; k++) z4_st16_stream(op + k * stride, ((uint64_t)e[k][1] << 32) | e[k][0], ((uint64_t)e[k][3] << 32) | e[k][2]);
}

// v of the four lanes (row 0..3, l) -- the wavefront's four rows of 16 lanes -- in every one of them (gfx950's v_permlane16_swap /
// v_permlane32_swap: registers only)
__device__ __forceinline__ void z4_row_gather(uint64_t v, uint64_t* out) {
    const uint32_t w[2] = {(uint32_t)v, (uint32_t)(v >> 32)};
    uint32_t r[4][2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const auto s16 = __builtin_amdgcn_permlane16_swap(w[h], w[h], false, false);    // [v0 v0 v2 v2], [v1 v1 v3 v3] by rows
        const auto e = __builtin_amdgcn_permlane32_swap(s16[0], s16[0], false, false);  // v0 everywhere, v2 everywhere
        const auto o = __builtin_amdgcn_permlane32_swap(s16[1], s16[1], false, false);  // v1, v3
        r[0][h] = e[0], r[2][h] = e[1], r[1][h] = o[0], r[3][h] = o[1];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = ((uint64_t)r[k][1] << 32) | r[k][0];
}

}  // namespace rv
