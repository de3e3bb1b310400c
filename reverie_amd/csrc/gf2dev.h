// Device helpers shared by the GF(2) interpreter kernels (kernels.hip, ldsrun.hip): the quad-word arithmetic of
// algebra/gf2/domain.rs:10-63 (reconstruct = per-byte parity) and the one-bit-per-repetition packing of corr /
// preprocessing bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rv {

// 0x01 bytes -> 0xFF bytes: (t << 8) - t.  Written out because the compiler turns the expression into a multiplication by
// 255, and v_mul_lo_u32 runs at a quarter of the rate of the shift and the subtraction
__device__ __forceinline__ uint32_t smear01(uint32_t t) {
    uint32_t r;
    asm("v_lshlrev_b32 %0, 8, %1\n\tv_sub_u32 %0, %0, %1" : "=&v"(r) : "v"(t));
    return r;
}
// DomainGF2::reconstruct on a quad word: per-byte parity, smeared to 0x00/0xFF
__device__ __forceinline__ uint32_t recon32(uint32_t t) {
    t ^= t >> 4;
    t ^= t >> 2;
    t ^= t >> 1;
    t &= 0x01010101u;
    return smear01(t);
}

// corr / preprocessing bits are stored one bit per repetition: nibble bit k <-> byte k of the
// smeared word (LSB-first), i.e. repetition 4q + 3 - k
__device__ __forceinline__ uint32_t expand4(uint32_t n) {
    const uint32_t t = (n | (n << 7) | (n << 14) | (n << 21)) & 0x01010101u;
    return smear01(t);
}
__device__ __forceinline__ uint32_t compress4(uint32_t x) {
    const uint32_t y = x & 0x08040201u;
    return (y | (y >> 8) | (y >> 16) | (y >> 24)) & 0xFu;
}
__device__ __forceinline__ uint32_t load_bits(const uint8_t* base, size_t row, uint32_t NQ, uint32_t q) {
    return expand4(((uint32_t)base[row * (NQ >> 1) + (q >> 1)] >> (4 * (q & 1))) & 0xFu);
}
// the value of lane ^ 1 (DPP quad_perm [1,0,3,2]: one VALU instruction, no trip through the LDS crossbar)
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
// the two quads sharing a byte are adjacent lanes of the same gate
__device__ __forceinline__ void store_bits(uint8_t* base, size_t row, uint32_t NQ, uint32_t q, uint32_t smeared) {
    const uint32_t n = compress4(smeared);
    const uint32_t other = pair_swap(n);
    if (!(q & 1)) base[row * (NQ >> 1) + (q >> 1)] = (uint8_t)(n | (other << 4));
}

}  // namespace rv
