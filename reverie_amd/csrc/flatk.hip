// Kernels of the flat prover schedule (flat.h) for gfx950: the cleartext pass and the program-order Mul kernel.
//
// Replaces, for the prover of a pure GF(2) circuit (all under /root/reference/src/):
//   interpreter/single.rs:25-69      Instance::op_mul               -> k_mul_flat (every Mul of the circuit, program order)
//   transcript/prover.rs:181-232     ProverTranscript::{input,reconstruct,correction}  (the transcript rows it writes)
//   the wire VALUES the reference carries as corr = value - reconstruct(mask) (interpreter/mod.rs:10-19) -> k_clear
// The XOR rows and the Input / AssertZero transcript rows run through kernels.hip's level kernels in MODE_PROVE_F.
#include <stdlib.h>

#include <algorithm>

#include "flat.h"
#include "gf2dev.h"
#include "internal.h"

namespace rv {

// ------------------------------------------------------------------------------------
// k_clear: the circuit evaluated in the clear, once per proof -- one bit per share row, the same in every repetition.
// The only part of a proof that still walks the dependency levels; it moves a byte per row, so a handful of workgroups
// do it beside the mask generator (which is VALU-bound and leaves them the memory system).
//
// G workgroups of 1024 threads; level l: thread (wg, t) takes gates lo + wg * 1024 + t, + G * 1024, ...; then a barrier over
// the G workgroups.  Values cross workgroups (other CUs, other XCDs' L2s) as write-through byte stores and L1-bypassing
// loads (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 on both sides, every storing wave drains vmcnt before the
// arrival is published).  Every spin is bounded: a workgroup that waits longer than CLEAR_SPIN_TICKS gives up, sets the
// abort word and the proof fails with RV_E_DEVICE instead of hanging the queue.
// ------------------------------------------------------------------------------------
constexpr long long CLEAR_SPIN_TICKS = 200000000ll;  // wall_clock64 runs at 100 MHz: 2 s

#define RV_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RV_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

struct ClearParams {
    const ClearRec* recs;          // 16-byte records (at most one base row per operand)
    const ClearRecK* recs_k;       // 32-byte records
    const ClearLevel* levels;      // [n_levels]
    uint32_t n_levels;
    const uint8_t* wit;
    uint8_t* v;                    // [n_rows] bit 0: cleartext value of the share row's wire; a Mul's output row: bit 1 / 2 = its operands' values
    int* err;                      // RV_E_WITNESS_INVALID when an AssertZero wire is not zero (prover.rs:221-228)
    uint32_t* sync;                // [0] arrivals, [1] abort
};

__device__ __forceinline__ void clear_finish(const ClearParams& p, uint32_t meta, uint32_t dst, uint32_t xa, uint32_t xb) {
    const uint32_t op = meta & 7u, ca = (meta >> 3) & 1u, cb = (meta >> 4) & 1u;
    if (op == G_MUL) {
        const uint32_t vx = (xa ^ ca) & 1u, vy = (xb ^ cb) & 1u;
        RV_AGENT_STORE(p.v + dst, (uint8_t)((vx & vy) | (vx << 1) | (vy << 2)));
    } else if (op == G_XORK) {
        RV_AGENT_STORE(p.v + dst, (uint8_t)((xa ^ xb ^ ca) & 1u));
    } else if (op == G_INPUT) {
        RV_AGENT_STORE(p.v + dst, (uint8_t)(xa ? 1 : 0));
    } else if (op == G_ASSERT) {
        if (((xa ^ xb ^ ca) & 1u) != 0) atomicOr(p.err, RV_E_WITNESS_INVALID);
    }
}

// One level's records of one kind, U per thread and pass: every record load of the pass is in flight before the first value
// load, every value load before the first store (a level is two dependent memory round trips, not two per gate).
template <int U>
__device__ __forceinline__ void clear_simple(const ClearParams& p, uint32_t lo, uint32_t hi, uint32_t first, uint32_t stride) {
    for (uint32_t i0 = lo + first; i0 < hi; i0 += U * stride) {
        uint4 r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = i0 + u * stride;
            r[u] = make_uint4(0, 0, 0, 0xFFu);  // (op 7: nothing)
            if (i < hi) r[u] = *(const uint4*)(p.recs + i);
        }
        uint32_t xa[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t meta = r[u].w, op = meta & 7u;
            xa[u] = 0, xb[u] = 0;
            if (op == G_INPUT) {
                xa[u] = p.wit[r[u].y];
            } else if (op != 7u) {
                if ((meta >> 8) & 3u) xa[u] = RV_AGENT_LOAD(p.v + r[u].y);
                if ((meta >> 10) & 3u) xb[u] = RV_AGENT_LOAD(p.v + r[u].z);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) clear_finish(p, r[u].w, r[u].x, xa[u], xb[u]);
    }
}
template <int U>
__device__ __forceinline__ void clear_general(const ClearParams& p, uint32_t lo, uint32_t hi, uint32_t first, uint32_t stride) {
    for (uint32_t i0 = lo + first; i0 < hi; i0 += U * stride) {
        uint4 r0[U], r1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = i0 + u * stride;
            r0[u] = make_uint4(0, 0xFFu, 0, 0), r1[u] = make_uint4(0, 0, 0, 0);
            if (i < hi) {
                const uint4* q = (const uint4*)(p.recs_k + i);
                r0[u] = q[0], r1[u] = q[1];
            }
        }
        // ClearRecK: dst meta a0 a1 | a2 b0 b1 b2
        uint32_t xa[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t meta = r0[u].y, na = (meta >> 8) & 3u, nb = (meta >> 10) & 3u;
            uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
            if ((meta & 7u) != 7u) {
                if (na > 0) t0 = RV_AGENT_LOAD(p.v + r0[u].z);
                if (na > 1) t1 = RV_AGENT_LOAD(p.v + r0[u].w);
                if (na > 2) t2 = RV_AGENT_LOAD(p.v + r1[u].x);
                if (nb > 0) t3 = RV_AGENT_LOAD(p.v + r1[u].y);
                if (nb > 1) t4 = RV_AGENT_LOAD(p.v + r1[u].z);
                if (nb > 2) t5 = RV_AGENT_LOAD(p.v + r1[u].w);
            }
            xa[u] = t0 ^ t1 ^ t2, xb[u] = t3 ^ t4 ^ t5;
        }
#pragma unroll
        for (int u = 0; u < U; u++) clear_finish(p, r0[u].y, r0[u].x, xa[u], xb[u]);
    }
}

__global__ __launch_bounds__(1024) void k_clear(ClearParams p) {
    __shared__ uint32_t s_abort;
    const uint32_t G = gridDim.x;
    const uint32_t stride = G * 1024u, first = blockIdx.x * 1024u + threadIdx.x;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    for (uint32_t l = 0; l < p.n_levels; l++) {
        const ClearLevel L = p.levels[l];
        clear_simple<4>(p, L.s0, L.s1, first, stride);
        clear_general<2>(p, L.g0, L.g1, first, stride);
        if (l + 1 == p.n_levels) break;
        // barrier over the G workgroups: every wave's write-through stores have left before the arrival is counted
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = G * (l + 1);
            const long long t0 = wall_clock64();
            uint32_t spins = 0;
            while (RV_AGENT_LOAD(p.sync) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0) {
                    if (RV_AGENT_LOAD(p.sync + 1) != 0 || wall_clock64() - t0 > CLEAR_SPIN_TICKS) {
                        RV_AGENT_STORE(p.sync + 1, 1u);
                        s_abort = 1;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (s_abort) {
            if (threadIdx.x == 0) atomicOr(p.err, RV_DEV_CLEAR_ABORT);  // reported as RV_E_DEVICE by the host
            return;
        }
    }
}

__global__ void k_or_word(int* __restrict__ dst, const int* __restrict__ src) {
    const int v = *src;
    if (v) atomicOr(dst, v);
}
void launch_or_word(hipStream_t st, int* d_dst, const int* d_src) { hipLaunchKernelGGL(k_or_word, dim3(1), dim3(1), 0, st, d_dst, d_src); }

void launch_clear(hipStream_t st, uint32_t n_wgs, const ClearRec* d_recs, const ClearRecK* d_recs_k, const ClearLevel* d_levels, uint32_t n_levels,
                  const uint8_t* d_wit, uint8_t* d_v, int* d_err, uint32_t* d_sync) {
    ClearParams p{d_recs, d_recs_k, d_levels, n_levels, d_wit, d_v, d_err, d_sync};
    hipLaunchKernelGGL(k_clear, dim3(n_wgs), dim3(1024), 0, st, p);
}

// ------------------------------------------------------------------------------------
// k_mul_flat: Mul gates [i0, i1) of the program (index = preprocessing row), a persistent grid: wavefront w takes the
// 4-gate steps w, w + n_waves, ...  No gate depends on another one here -- operand rows are PRG masks or XOR rows that the
// launches before this one completed, the operands' cleartext values come from k_clear -- so the fresh masks, the online
// rows and the preprocessing bits are sequential streams and only the operand rows are gathered.
// ------------------------------------------------------------------------------------
struct MulFlatParams {
    const uint32_t* rows;
    uint32_t* on;
    uint8_t* pre;
    const uint8_t* v;  // k_clear's value bytes, indexed by share row
};

template <int NQ, int U>
__device__ __forceinline__ void mul_flat_step(const MulRec* __restrict__ recs, uint32_t g0, const MulFlatParams& p, uint32_t sub, uint32_t q) {
    constexpr uint32_t GPW = 64 / NQ;
    MulRec r[U];
    uint32_t vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        r[u] = recs[g0 + u * GPW + sub];
        vb[u] = p.v[r[u].m + 1];  // k_clear: bit 1 / 2 = the operands' cleartext values (constants applied)
    }
    uint32_t ra[U][RV_LIN_K], rb[U][RV_LIN_K], lab[U], lnew[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t na = (r[u].eo_flags >> 26) & 3u, nb = (r[u].eo_flags >> 28) & 3u;  // bases - 1
#pragma unroll
        for (int i = 0; i < RV_LIN_K; i++) {
            ra[u][i] = 0;
            rb[u][i] = 0;
            if (i == 0 || (uint32_t)i <= na) ra[u][i] = p.rows[(size_t)r[u].a[i] * NQ + q];
            if (i == 0 || (uint32_t)i <= nb) rb[u][i] = p.rows[(size_t)r[u].b[i] * NQ + q];
        }
        lab[u] = __builtin_nontemporal_load(&p.rows[(size_t)r[u].m * NQ + q]);
        lnew[u] = p.rows[(size_t)(r[u].m + 1) * NQ + q];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t lx = ra[u][0], ly = rb[u][0];
#pragma unroll
        for (int i = 1; i < RV_LIN_K; i++) lx ^= ra[u][i], ly ^= rb[u][i];
        const uint32_t a = recon32(lx), b = recon32(ly), c = recon32(lab[u]);
        const uint32_t cx = a ^ ((vb[u] & 2u) ? 0xFFFFFFFFu : 0u);  // corr = value - reconstruct(mask)
        const uint32_t cy = b ^ ((vb[u] & 4u) ? 0xFFFFFFFFu : 0u);
        const uint32_t delta = (a & b) ^ c;
        const uint32_t s = (ly & cx) ^ (lx & cy) ^ lab[u] ^ lnew[u];
        __builtin_nontemporal_store(s, &p.on[(size_t)(r[u].eo_flags & MULREC_EO_MASK) * NQ + q]);
        store_bits(p.pre, g0 + u * GPW + sub, NQ, q, delta);
    }
}

template <int NQ>
__global__ __launch_bounds__(256) void k_mul_flat(const MulRec* __restrict__ recs, uint32_t i0, uint32_t i1, MulFlatParams p) {
    constexpr int U = 4;
    constexpr uint32_t GPW = 64 / NQ, STEP = U * GPW;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t q = lane % NQ, sub = lane / NQ;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t n_full = (i1 - i0) / STEP;
    for (uint32_t t = wave; t < n_full; t += n_waves) mul_flat_step<NQ, U>(recs, i0 + t * STEP, p, sub, q);
    // the last < STEP gates, GPW at a time
    const uint32_t rest0 = i0 + n_full * STEP;
    const uint32_t n_rest = (i1 - rest0 + GPW - 1) / GPW;
    if (wave >= n_rest) return;
    const uint32_t g = rest0 + wave * GPW + sub;
    if (g < i1) {
        // (one gate per lane group: the same arithmetic, scalar form)
        const MulRec r = recs[g];
        const uint32_t vb = p.v[r.m + 1];
        uint32_t lx = 0, ly = 0;
#pragma unroll
        for (int i = 0; i < RV_LIN_K; i++) lx ^= p.rows[(size_t)r.a[i] * NQ + q], ly ^= p.rows[(size_t)r.b[i] * NQ + q];
        const uint32_t lab = p.rows[(size_t)r.m * NQ + q], lnew = p.rows[(size_t)(r.m + 1) * NQ + q];
        const uint32_t a = recon32(lx), b = recon32(ly), c = recon32(lab);
        const uint32_t cx = a ^ ((vb & 2u) ? 0xFFFFFFFFu : 0u), cy = b ^ ((vb & 4u) ? 0xFFFFFFFFu : 0u);
        p.on[(size_t)(r.eo_flags & MULREC_EO_MASK) * NQ + q] = (ly & cx) ^ (lx & cy) ^ lab ^ lnew;
        store_bits(p.pre, g, NQ, q, (a & b) ^ c);
    }
}

template <int NQ>
static void launch_mul_flat_nq(hipStream_t st, const MulRec* d_recs, uint32_t i0, uint32_t i1, const MulFlatParams& p) {
    constexpr uint32_t STEP = 4 * (64 / NQ);
    const uint64_t steps = ((uint64_t)(i1 - i0) + STEP - 1) / STEP;
    static const uint32_t max_blocks = [] {
        if (const char* e = getenv("RV_FLAT_BLOCKS")) return (uint32_t)std::max(atoi(e), 1);
        return 2048u;  // 8 per CU: every wavefront slot of the chip
    }();
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((steps + 3) / 4, 1), max_blocks);
    hipLaunchKernelGGL((k_mul_flat<NQ>), dim3(blocks), dim3(256), 0, st, d_recs, i0, i1, p);
}

bool mul_flat_supports(uint32_t NQ) { return NQ == 64 || NQ == 32 || NQ == 16 || NQ == 8; }

void launch_mul_flat(hipStream_t st, uint32_t NQ, const MulRec* d_recs, uint32_t i0, uint32_t i1, const uint32_t* d_rows, uint32_t* d_on, uint8_t* d_pre,
                     const uint8_t* d_v) {
    if (i1 <= i0) return;
    const MulFlatParams p{d_rows, d_on, d_pre, d_v};
    switch (NQ) {
    case 64: return launch_mul_flat_nq<64>(st, d_recs, i0, i1, p);
    case 32: return launch_mul_flat_nq<32>(st, d_recs, i0, i1, p);
    case 16: return launch_mul_flat_nq<16>(st, d_recs, i0, i1, p);
    case 8: return launch_mul_flat_nq<8>(st, d_recs, i0, i1, p);
    default: break;
    }
}

}  // namespace rv
