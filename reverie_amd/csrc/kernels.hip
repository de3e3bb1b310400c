// Interpreter, transcript hashing and opening kernels for gfx950.
//
// Replaces (all under /root/reference/src/):
//   interpreter/single.rs:25-157      Instance::step / op_mul over the GF(2) ring
//   algebra/gf2/domain.rs:10-63       Share*Recon, reconstruct (per-byte parity)
//   transcript/prover.rs:181-232      ProverTranscript::{input,reconstruct,correction,zero_check}
//   transcript/verifier/online.rs:122-183, verifier/preprocess.rs:46-79
//   crypto/hash.rs:17-104             BufferedHasher / PackedHasher (per-rep BLAKE3 streams)
//   transcript/mod.rs:77-96, interpreter/combine.rs:104-118   digest joins
//   transcript/prover.rs:57-175 + algebra/gf2/{share,recon}.rs Pack/PackSelected  (openings)
//
// Lane mapping: one lane = one quad word = 4 repetitions x 8 players (see internal.h);
// NQ consecutive lanes cover every repetition of the shard for one gate, so a wavefront
// reads/writes whole 256-byte rows.  Gates of one dependency level are independent and
// are spread over the grid; levels are separate launches.
#include <stdlib.h>

#include <algorithm>

#include "b3.h"
#include "gf2dev.h"
#include "internal.h"
#include "launch.h"

namespace rv {

static size_t g_device_lds_limit = 160 * 1024;
void set_device_lds_limit(size_t bytes) { g_device_lds_limit = bytes; }
size_t device_lds_limit() { return g_device_lds_limit; }


// XOR of the n listed base rows / their corr bits.  Unused slots hold the zero row, so all
// RV_LIN_K slots are loaded unconditionally with STATIC indices (a runtime-indexed id array would
// push the gate record into scratch memory; a per-slot branch would serialise the loads).
__device__ __forceinline__ uint32_t gather_rows(const uint32_t* rows, const uint32_t* ids, uint32_t NQ, uint32_t q) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < RV_LIN_K; i++) v ^= rows[(size_t)ids[i] * NQ + q];
    return v;
}
__device__ __forceinline__ uint32_t gather_corr_byte(const uint8_t* corr, const uint32_t* ids, uint32_t NQ, uint32_t o) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < RV_LIN_K; i++) v ^= corr[(size_t)ids[i] * (NQ >> 1) + o];
    return v;
}
__device__ __forceinline__ uint32_t gather_corr(const uint8_t* corr, const uint32_t* ids, uint32_t NQ, uint32_t q) {
    return (gather_corr_byte(corr, ids, NQ, q >> 1) >> (4 * (q & 1))) & 0xFu;
}

// MODE_PROVE_V: cleartext value of an operand = XOR of its base rows' values (unused slots hold the zero row, value 0)
__device__ __forceinline__ uint32_t gather_vclr(const uint8_t* vclr, const uint32_t* ids) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < RV_LIN_K; i++) v ^= vclr[ids[i]];
    return v;
}

// MODE_VERIFY_C (round 4): the verifier of a whole proof without corr rows.  Only the opened repetitions need public
// corrections, and in the verifier's slot order they sit in the first sixteen quad words of a row: ONE u64 per row (nibble q =
// the four corr bits of quad word q, the 32-byte row's own bit order) instead of a 32-byte row that every lane gathers a byte
// of.  An operand's corrections are then one 8-byte access at a wave-uniform address per base row, an XOR gate's are a u64 XOR
// by one lane, and lazy linear forms stop costing the verifier a row access per base.
__device__ __forceinline__ uint32_t vc_nib(uint64_t c, uint32_t q) {
    const uint32_t w = (q & 8) ? (uint32_t)(c >> 32) : (uint32_t)c;
    return q < 16 ? (w >> (4 * (q & 7))) & 0xFu : 0u;
}
__device__ __forceinline__ uint64_t gather_vc(const uint64_t* vc, const uint32_t* ids) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < RV_LIN_K; i++) v ^= vc[ids[i]];
    return v;
}
// the lanes q = 0 .. 15 of a full-width row (one DPP row) put their nibbles together: lanes 7 and 15 end up with the low and
// the high word (OR over the eight lanes before them) and store it
__device__ __forceinline__ void vc_store(uint64_t* vc, size_t row, uint32_t q, uint32_t smeared) {
    uint32_t v = q < 16 ? compress4(smeared) << (4 * (q & 7)) : 0u;
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    if (q == 7 || q == 15) ((uint32_t*)(vc + row))[q >> 3] = v;
}
constexpr bool is_verify(int mode) { return mode == MODE_VERIFY || mode == MODE_VERIFY_C; }

// Accesses that must be seen across workgroups INSIDE one launch (k_interp_persist: a level reads what other compute units,
// on other XCDs, wrote a few microseconds earlier in the same kernel): agent-scope relaxed atomics = `sc1` loads that bypass the
// reader's L1 and write-through `sc1` stores (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 on both sides needs no fence).
// COH = 0: the plain accesses of the one-launch-per-level kernels.  COH = 2 (k_chain: every workgroup of the launch sits on ONE
// XCD, i.e. behind one L2): loads bypass the L1 the same way, stores are plain -- they stay in the shared L2, where the other
// compute units of the XCD find them at L2 latency instead of memory latency.
// (experiment switches: which of the three kinds of access take the coherent form)
#ifndef RV_COH_LDROW
#define RV_COH_LDROW 1
#endif
#ifndef RV_COH_STROW
#define RV_COH_STROW 1
#endif
#ifndef RV_COH_V
#define RV_COH_V 1
#endif
template <int COH>
__device__ __forceinline__ uint32_t ld_row(const uint32_t* p) {
    if (COH && RV_COH_LDROW) return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int COH>
__device__ __forceinline__ void st_row(uint32_t* p, uint32_t v) {
    if (COH == 1 && RV_COH_STROW)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <int COH>
__device__ __forceinline__ uint32_t ld_v(const uint8_t* p) {
    if (COH && RV_COH_V) return __hip_atomic_load(const_cast<uint8_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int COH>
__device__ __forceinline__ void st_v(uint8_t* p, uint8_t v) {
    if (COH == 1 && RV_COH_V)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <int COH>
__device__ __forceinline__ uint32_t gather_rows_c(const uint32_t* rows, const uint32_t* ids, uint32_t NQ, uint32_t q) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < RV_LIN_K; i++) v ^= ld_row<COH>(&rows[(size_t)ids[i] * NQ + q]);
    return v;
}
template <int COH>
__device__ __forceinline__ uint32_t gather_vclr_c(const uint8_t* vclr, const uint32_t* ids) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < RV_LIN_K; i++) v ^= ld_v<COH>(&vclr[ids[i]]);
    return v;
}

template <int MODE, int COH = 0>
__device__ __forceinline__ void interp_one_impl(const Gate& g, const InterpParams& p, uint32_t NQ, uint32_t q, uint32_t onm) {
    switch (g_op(g)) {
    case G_INPUT: {
        const uint32_t lam = p.rows[(size_t)g.m * NQ + q];
        uint32_t corr;
        if (!is_verify(MODE)) {
            const uint32_t w = p.wit[g.x] ? 0xFFFFFFFFu : 0u;
            corr = w ^ recon32(lam);
        } else {
            corr = onm ? (p.sup_in[(size_t)g.x * p.sup_nq + q] & onm) : 0u;  // (rows of quads without an opened repetition are never written)
        }
        if (!is_verify(MODE) || onm) p.on[(size_t)g.eo * NQ + q] = corr;
        if (MODE == MODE_PROVE_V) {
            if (q == 0) st_v<COH>(&p.vclr[g.dst], p.wit[g.x] ? 1 : 0);
        } else if (MODE == MODE_VERIFY_C) {
            vc_store(p.vc, g.dst, q, corr);
        } else {
            store_bits(p.corr, g.dst, NQ, q, corr);
        }
        break;
    }
    case G_XORK: {
        st_row<COH>(&p.rows[(size_t)g.dst * NQ + q], gather_rows_c<COH>(p.rows, g.a, NQ, q) ^ gather_rows_c<COH>(p.rows, g.b, NQ, q));
        if (MODE == MODE_PROVE_V) {
            if (q == 0) st_v<COH>(&p.vclr[g.dst], (uint8_t)((g_ca(g) ^ gather_vclr_c<COH>(p.vclr, g.a) ^ gather_vclr_c<COH>(p.vclr, g.b)) & 1u));
            break;
        }
        if (MODE == MODE_VERIFY_C) {
            if (q == 0) p.vc[g.dst] = gather_vc(p.vc, g.a) ^ gather_vc(p.vc, g.b) ^ (g_ca(g) ? ~0ull : 0ull);
            break;
        }
        // corr bits: plain byte XOR, no expansion needed
        if (!(q & 1)) {
            const size_t h = NQ >> 1, o = q >> 1;
            const uint32_t c = (g_ca(g) ? 0xFFu : 0u) ^ gather_corr_byte(p.corr, g.a, NQ, o) ^ gather_corr_byte(p.corr, g.b, NQ, o);
            p.corr[(size_t)g.dst * h + o] = (uint8_t)c;
        }
        break;
    }
    case G_RANDOM: {
        if (MODE == MODE_VERIFY_C) {
            if (q == 0) p.vc[g.dst] = 0;
            break;
        }
        if (!(q & 1)) p.corr[(size_t)g.dst * (NQ >> 1) + (q >> 1)] = 0;
        break;
    }
    case G_MUL: {
        const uint32_t lx = gather_rows_c<COH>(p.rows, g.a, NQ, q), ly = gather_rows_c<COH>(p.rows, g.b, NQ, q);
        const uint32_t lab = p.rows[(size_t)g.m * NQ + q], lnew = p.rows[(size_t)(g.m + 1) * NQ + q];
        const uint32_t a = recon32(lx), b = recon32(ly), c = recon32(lab);
        uint32_t cx, cy, vx = 0, vy = 0;
        if (MODE == MODE_PROVE_V) {
            vx = (gather_vclr_c<COH>(p.vclr, g.a) ^ g_ca(g)) & 1u;
            vy = (gather_vclr_c<COH>(p.vclr, g.b) ^ g_cb(g)) & 1u;
            cx = a ^ (vx ? 0xFFFFFFFFu : 0u);  // corr = value - reconstruct(mask)
            cy = b ^ (vy ? 0xFFFFFFFFu : 0u);
        } else if (MODE == MODE_VERIFY_C) {
            cx = expand4(vc_nib(gather_vc(p.vc, g.a), q)) ^ (g_ca(g) ? 0xFFFFFFFFu : 0u);
            cy = expand4(vc_nib(gather_vc(p.vc, g.b), q)) ^ (g_cb(g) ? 0xFFFFFFFFu : 0u);
        } else {
            cx = expand4(gather_corr(p.corr, g.a, NQ, q)) ^ (g_ca(g) ? 0xFFFFFFFFu : 0u);
            cy = expand4(gather_corr(p.corr, g.b, NQ, q)) ^ (g_cb(g) ? 0xFFFFFFFFu : 0u);
        }
        uint32_t delta = (a & b) ^ c;
        uint32_t s = (ly & cx) ^ (lx & cy) ^ lab ^ lnew;
        uint32_t r;
        if (!is_verify(MODE)) {
            r = recon32(s);
        } else {
            // online-verified reps: supplied correction, add the unopened player's broadcast
            if (onm) {
                delta = (p.sup_corr[(size_t)g.ep * p.sup_nq + q] & onm) | (delta & ~onm);
                s ^= p.sup_rec[(size_t)g.x * p.sup_nq + q];
            }
            r = recon32(s) & onm;  // preprocessing-verified reps: reconstruct() returns zero
        }
        // verifier: the online transcript is only hashed for quads that hold an opened repetition (the other
        // repetitions' online digests come from the proof), so only those lanes store -- in the verifier's slot order
        // they are the first ten quads of a row, two 32-byte sectors instead of eight
        if (!is_verify(MODE) || onm) p.on[(size_t)g.eo * NQ + q] = s;
        store_bits(p.pre, g.ep, NQ, q, delta);
        if (MODE == MODE_PROVE_V) {
            if (q == 0) st_v<COH>(&p.vclr[g.dst], (uint8_t)(vx & vy));
        } else if (MODE == MODE_VERIFY_C) {
            vc_store(p.vc, g.dst, q, r ^ delta ^ (cx & cy));
        } else {
            store_bits(p.corr, g.dst, NQ, q, r ^ delta ^ (cx & cy));
        }
        break;
    }
    case G_RECON: {
        // B2A's recorded reconstruction (combine.rs:181-183): value = reconstruct(mask) + corr
        uint32_t m = gather_rows(p.rows, g.a, NQ, q);
        if (is_verify(MODE) && onm) m ^= p.sup_rec[(size_t)g.x * p.sup_nq + q];
        if (MODE == MODE_PROVE || onm) p.on[(size_t)g.eo * NQ + q] = m;
        uint32_t r = recon32(m);
        if (is_verify(MODE)) r &= onm;
        const uint32_t cx = expand4(gather_corr(p.corr, g.a, NQ, q)) ^ (g_ca(g) ? 0xFFFFFFFFu : 0u);
        p.rows[(size_t)g.dst * NQ + q] = 0;
        store_bits(p.corr, g.dst, NQ, q, r ^ cx);
        break;
    }
    case G_ASSERT: {
        uint32_t m = gather_rows_c<COH>(p.rows, g.a, NQ, q);
        if (is_verify(MODE) && onm) m ^= p.sup_rec[(size_t)g.x * p.sup_nq + q];
        if (!is_verify(MODE) || onm) p.on[(size_t)g.eo * NQ + q] = m;
        if (MODE == MODE_PROVE_V) {
            // the wire's value itself must be zero (prover.rs:221-228), the same in every repetition
            if (q == 0 && ((gather_vclr_c<COH>(p.vclr, g.a) ^ g_ca(g)) & 1u) != 0) atomicOr(p.err, RV_E_WITNESS_INVALID);
        } else {
            const uint32_t cx = expand4(MODE == MODE_VERIFY_C ? vc_nib(gather_vc(p.vc, g.a), q) : gather_corr(p.corr, g.a, NQ, q)) ^ (g_ca(g) ? 0xFFFFFFFFu : 0u);
            if (MODE == MODE_PROVE) {
                if ((recon32(m) ^ cx) != 0) atomicOr(p.err, RV_E_WITNESS_INVALID);
            } else {
                // online.rs:175-177: okay &= recon.is_zero() -- the reference never reads it; RV_VERIFY_STRICT does
                if (((recon32(m) ^ cx) & onm) != 0) atomicOr(p.err, RV_DEV_ZERO_CHECK);
            }
        }
        break;
    }
    default:
        break;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_interp(const Gate* __restrict__ gates, uint32_t lo, uint32_t hi, InterpParams p) {
    const uint32_t NQ = p.NQ;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = tid % NQ;
    const uint32_t worker = tid / NQ;
    const uint32_t n_workers = (gridDim.x * blockDim.x) / NQ;
    const uint32_t onm = (is_verify(MODE)) ? p.on_mask[q] : 0u;
    for (uint32_t gi = lo + worker; gi < hi; gi += n_workers) {
        const Gate g = gates[gi];
        interp_one_impl<MODE>(g, p, NQ, q, onm);
    }
}

#ifndef RV_B3_RPL
#define RV_B3_RPL 4
#endif
#ifndef RV_INTERP_UNROLL
#define RV_INTERP_UNROLL 4
#endif
#ifndef RV_INTERP_UNROLL_SMALL
#define RV_INTERP_UNROLL_SMALL 2
#endif
// gates a wavefront keeps in flight per step and gate group: narrow rows (small repetition shards, several gates per
// wavefront already) want fewer -- measured per rank on the 10^7-gate circuit: 128 repetitions (NQ = 32) 1.75 ms
// with 2, 1.64 with 4; 64 repetitions the same either way; 32 repetitions 1.02 with 2, 1.09 with 4
#ifndef RV_INTERP_UNROLL_MID
#define RV_INTERP_UNROLL_MID 4
#endif
#ifndef RV_INTERP_UNROLL_FAST
#define RV_INTERP_UNROLL_FAST 4
#endif
// two-row Xor steps of the full-width variant without the multi-base loops when they differ from its Mul steps (0 = the same).
// The point of unequal steps: a level's wave-steps against the wavefronts the chip holds at once -- a level that needs 1.4
// generations of wavefronts takes two rounds of memory latency, one that fits takes one
#ifndef RV_INTERP_UXOR_FAST
#define RV_INTERP_UXOR_FAST 0
#endif
// `general` = the kernel variant that also carries the multi-base Mul / Xor loops (more registers)
__host__ __device__ constexpr int interp_unroll(int NQ, bool general = true) {
    return NQ >= 64 ? (general ? RV_INTERP_UNROLL : RV_INTERP_UNROLL_FAST) : NQ >= 32 ? RV_INTERP_UNROLL_MID : RV_INTERP_UNROLL_SMALL;
}

// Gate-record prefetch.  A wavefront of a level launch lives for three dependent memory round trips: its gate records
// -> the operand rows they name -> the stores.  The records are static and contiguous (sorted by level, class), so
// the wavefront that runs unrolled step t also touches the records of step t + dist (one load instruction, a lane per
// 128-byte line, result unused), issued right behind its own row loads: by the time a later wavefront asks for them
// they sit in L2 and the first round trip is an L2 hit instead of an HBM miss.  Steps past the end of this level map
// onto the first steps of the next one (the gate array is contiguous across levels).  L2 is per XCD and workgroups
// are dealt to the XCDs round-robin, so producer and consumer must agree modulo 8 workgroups = 32 wavefronts: step t
// runs on wavefront t mod n_waves, and dist and the wrap-around are kept multiples of 32.
struct PfPlan {
    uint32_t dist;         // 0 = off
    uint32_t rem;          // steps of this level modulo 32 (added back after the wrap so that t' = t + dist - 32k)
    uint32_t n[2][4];      // [0] this level, [1] the next one: full unrolled steps of classes 0..3
    uint32_t start[2][4];  // first gate of each class
};
template <uint32_t STEP>
__device__ __forceinline__ const Gate* pf_target(const Gate* __restrict__ gates, const PfPlan& pf, uint32_t t) {
    uint32_t tt = t + pf.dist;
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (tt < pf.n[k][c]) return gates + pf.start[k][c] + tt * STEP;
            tt -= pf.n[k][c];
        }
        tt += pf.rem;
    }
    return nullptr;
}
// one lane per 128-byte line of the STEP records at `g` (+ one for the unaligned tail); wave-uniform `g`.  The value
// is a plain load that pf_sink() "uses" after the wavefront's last store, so the compiler's own vmcnt bookkeeping
// covers it and its destination register stays reserved until it has landed.
template <uint32_t STEP>
__device__ __forceinline__ uint32_t pf_touch(const Gate* g, uint32_t lane) {
    constexpr uint32_t BYTES = STEP * (uint32_t)sizeof(Gate), NL = (BYTES + 127) / 128;
    uint32_t v = 0;
    if (g && lane <= NL) v = *(const uint32_t*)((const char*)g + (lane < NL ? lane * 128 : BYTES - 4));
    return v;
}
__device__ __forceinline__ void pf_sink(uint32_t v) { asm volatile("" ::"v"(v)); }

// Fast path (NQ = 64, 32, 16 or 8, i.e. R = 256 .. 32): a wavefront covers 64/NQ gates at a time and the
// per-class ranges run as 4-way unrolled loops that put every operand row of 4 x 64/NQ gates in flight
// before the first use — the generic kernel above is latency-bound on the dependent
// gate-record -> operand-row chain (2 HBM round trips per gate).  With NQ = 64 the gate index is
// wave-uniform and the records come through scalar loads.  KA / KB = operand base rows actually
// loaded per gate: exact for the common one-base-per-operand class, RV_LIN_K (unused slots point at
// the L1-hot zero row) for the rest.
template <int MODE, int NQ, int U, int KA, int KB, int COH = 0>
__device__ __forceinline__ void mulU(const Gate* __restrict__ gates, uint32_t g0, const InterpParams& p, uint32_t sub, uint32_t q,
                                     uint32_t onm, const Gate* pf = nullptr) {
    constexpr uint32_t GPW = 64 / NQ, H = NQ / 2;
    // verifier: the online rows are stored in whole 32-byte sectors (the quads of the opened repetitions are a sector and a
    // quarter in its slot order, and a partially written sector is a read-modify-write at the memory side; the digests read
    // the opened quads only, so what the others hold does not matter)
    const bool on_wr = !is_verify(MODE) || ((__ballot(onm != 0) >> ((sub * NQ + q) & ~7u)) & 0xFFull) != 0;
    Gate g[U];
#pragma unroll
    for (int u = 0; u < U; u++) g[u] = gates[g0 + u * GPW + sub];
    uint32_t lx[U], ly[U], lab[U], lnew[U], bx[U], by[U], sc[U], sr[U];
    // slots >= 1 are loaded only when the operand really has that many bases (a wave-uniform branch at
    // NQ = 64); every load is issued before any value is used
    uint32_t ra[U][KA], ca[U][KA], rb[U][KB], cb[U][KB];
    // MODE_VERIFY_C: a lane reads the 32-bit half of a row's corrections word that holds its quad word's nibble (lanes 8 .. 15 the
    // high one; lanes >= 16 read the low one and use nothing of it)
    const uint32_t* const vc32 = (const uint32_t*)p.vc + ((q >> 3) & 1u);
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int na = (int)g_na(g[u]), nb = (int)g_nb(g[u]);
#pragma unroll
        for (int i = 0; i < KA; i++) {
            ra[u][i] = 0;
            ca[u][i] = 0;
            if (i == 0 || i < na) {
                ra[u][i] = ld_row<COH>(&p.rows[(size_t)g[u].a[i] * NQ + q]);
                ca[u][i] = MODE == MODE_VERIFY_C ? vc32[2 * (size_t)g[u].a[i]]
                                                 : MODE == MODE_PROVE_V ? ld_v<COH>(&p.vclr[g[u].a[i]]) : p.corr[(size_t)g[u].a[i] * H + (q >> 1)];
            }
        }
#pragma unroll
        for (int i = 0; i < KB; i++) {
            rb[u][i] = 0;
            cb[u][i] = 0;
            if (i == 0 || i < nb) {
                rb[u][i] = ld_row<COH>(&p.rows[(size_t)g[u].b[i] * NQ + q]);
                cb[u][i] = MODE == MODE_VERIFY_C ? vc32[2 * (size_t)g[u].b[i]]
                                                 : MODE == MODE_PROVE_V ? ld_v<COH>(&p.vclr[g[u].b[i]]) : p.corr[(size_t)g[u].b[i] * H + (q >> 1)];
            }
        }
        // lambda_ab is read exactly once and the online row is not read again before the hash phase: nontemporal, so
        // they do not displace operand rows from L2 (interpreter 2.35 -> 2.30 ms; lambda_new -- the output wire's mask,
        // an operand of the next level -- and the XOR outputs are better left as plain accesses: 2.35 / 2.38)
        lab[u] = __builtin_nontemporal_load(&p.rows[(size_t)g[u].m * NQ + q]);
        lnew[u] = p.rows[(size_t)(g[u].m + 1) * NQ + q];
        if (is_verify(MODE)) {
            sc[u] = sr[u] = 0;
            if (onm) {  // supplied values exist (and are stored) only for quads with an opened repetition
                sc[u] = p.sup_corr[(size_t)g[u].ep * p.sup_nq + q];
                sr[u] = p.sup_rec[(size_t)g[u].x * p.sup_nq + q];
            }
        }
    }
    const uint32_t pfv = pf_touch<U * GPW>(pf, sub * NQ + q);
#pragma unroll
    for (int u = 0; u < U; u++) {
        lx[u] = ra[u][0];
        bx[u] = ca[u][0];
        ly[u] = rb[u][0];
        by[u] = cb[u][0];
#pragma unroll
        for (int i = 1; i < KA; i++) {
            lx[u] ^= ra[u][i];
            bx[u] ^= ca[u][i];
        }
#pragma unroll
        for (int i = 1; i < KB; i++) {
            ly[u] ^= rb[u][i];
            by[u] ^= cb[u][i];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t a = recon32(lx[u]), b = recon32(ly[u]), c = recon32(lab[u]);
        uint32_t cx, cy;
        const uint32_t vx = (bx[u] ^ g_ca(g[u])) & 1u, vy = (by[u] ^ g_cb(g[u])) & 1u;  // MODE_PROVE_V: the operands' cleartext values
        if (MODE == MODE_PROVE_V) {
            cx = a ^ (vx ? 0xFFFFFFFFu : 0u);  // corr = value - reconstruct(mask)
            cy = b ^ (vy ? 0xFFFFFFFFu : 0u);
        } else if (MODE == MODE_VERIFY_C) {
            cx = expand4(q < 16 ? (bx[u] >> (4 * (q & 7))) & 0xFu : 0u) ^ (g_ca(g[u]) ? 0xFFFFFFFFu : 0u);
            cy = expand4(q < 16 ? (by[u] >> (4 * (q & 7))) & 0xFu : 0u) ^ (g_cb(g[u]) ? 0xFFFFFFFFu : 0u);
        } else {
            cx = expand4((bx[u] >> (4 * (q & 1))) & 0xFu) ^ (g_ca(g[u]) ? 0xFFFFFFFFu : 0u);
            cy = expand4((by[u] >> (4 * (q & 1))) & 0xFu) ^ (g_cb(g[u]) ? 0xFFFFFFFFu : 0u);
        }
        uint32_t delta = (a & b) ^ c;
        uint32_t s = (ly[u] & cx) ^ (lx[u] & cy) ^ lab[u] ^ lnew[u];
        uint32_t r = 0;
        if (MODE == MODE_PROVE) {
            r = recon32(s);
        } else if (is_verify(MODE)) {
            delta = (sc[u] & onm) | (delta & ~onm);
            s ^= sr[u];
            r = recon32(s) & onm;
        }
        if (!is_verify(MODE) || on_wr) __builtin_nontemporal_store(s, &p.on[(size_t)g[u].eo * NQ + q]);
        store_bits(p.pre, g[u].ep, NQ, q, delta);
        if (MODE == MODE_PROVE_V) {
            if (q == 0) st_v<COH>(&p.vclr[g[u].dst], (uint8_t)(vx & vy));
        } else if (MODE == MODE_VERIFY_C) {
            vc_store(p.vc, g[u].dst, q, r ^ delta ^ (cx & cy));
        } else {
            store_bits(p.corr, g[u].dst, NQ, q, r ^ delta ^ (cx & cy));
        }
    }
    pf_sink(pfv);
}

// G_XORK: N = base rows loaded per gate (2: a[0], a[1]; 6: a[0..2], b[0..2] with zero-row padding)
template <int MODE, int NQ, int U, int N, int COH = 0>
__device__ __forceinline__ void xorU(const Gate* __restrict__ gates, uint32_t g0, const InterpParams& p, uint32_t sub, uint32_t q,
                                     const Gate* pf = nullptr) {
    constexpr uint32_t GPW = 64 / NQ, H = NQ / 2;
    Gate g[U];
#pragma unroll
    for (int u = 0; u < U; u++) g[u] = gates[g0 + u * GPW + sub];
    uint32_t x[U], bx[U], rr[U][N], cc[U][N];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int na = (int)g_na(g[u]), nb = (int)g_nb(g[u]);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t id = (N == 2) ? g[u].a[i] : (i < RV_LIN_K ? g[u].a[i] : g[u].b[i - RV_LIN_K]);
            rr[u][i] = 0;
            cc[u][i] = 0;
            // only the slots the gate uses (N == 2: both by construction)
            if (N == 2 || (i < RV_LIN_K ? i < na : i - RV_LIN_K < nb)) {
                rr[u][i] = ld_row<COH>(&p.rows[(size_t)id * NQ + q]);
                // H corr bytes per row: the first H lanes of the gate's lane group carry them (MODE_PROVE_V: one value byte)
                if (MODE == MODE_PROVE_V) {
                    if (q == 0) cc[u][i] = ld_v<COH>(&p.vclr[id]);
                } else if (MODE == MODE_VERIFY_C) {
                    if (q < 2) cc[u][i] = ((const uint32_t*)p.vc)[2 * (size_t)id + q];  // (lanes 0, 1: the word's two halves)
                } else if (q < H) {
                    cc[u][i] = p.corr[(size_t)id * H + q];
                }
            }
        }
    }
    const uint32_t pfv = pf_touch<U * GPW>(pf, sub * NQ + q);
#pragma unroll
    for (int u = 0; u < U; u++) {
        x[u] = 0;
        bx[u] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            x[u] ^= rr[u][i];
            bx[u] ^= cc[u][i];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        st_row<COH>(&p.rows[(size_t)g[u].dst * NQ + q], x[u]);
        if (MODE == MODE_PROVE_V) {
            if (q == 0) st_v<COH>(&p.vclr[g[u].dst], (uint8_t)((bx[u] ^ g_ca(g[u])) & 1u));
        } else if (MODE == MODE_VERIFY_C) {
            if (q < 2) ((uint32_t*)p.vc)[2 * (size_t)g[u].dst + q] = bx[u] ^ (g_ca(g[u]) ? 0xFFFFFFFFu : 0u);
        } else if (q < H) {
            p.corr[(size_t)g[u].dst * H + q] = (uint8_t)(bx[u] ^ (g_ca(g[u]) ? 0xFFu : 0u));
        }
    }
    pf_sink(pfv);
}

template <int MODE>
__device__ __forceinline__ void interp_one(const Gate& g, const InterpParams& p, uint32_t NQ, uint32_t q, uint32_t onm) {
    interp_one_impl<MODE>(g, p, NQ, q, onm);
}

// One dependency level, class by class (LevelRange), executed by wavefronts `wave` of `n_waves`: shared by the
// one-launch-per-level kernel (all wavefronts of the grid) and the narrow-run kernel (the 16 wavefronts of one
// workgroup, gate records in LDS).  ROTATE (narrow runs): work is dealt to the wavefronts round-robin ACROSS the
// classes (`slot` = wave-steps handed out so far) — a level of five gates in three classes must land on five
// different wavefronts, not three times on wave 0.  All gates that do not fill a 4-way unrolled step go through ONE
// loop at the end, so the big per-gate switch exists once in the instruction stream.
// GENERAL = false: the level has (next to) no multi-base Mul / Xor gates (LevelRange classes 1 and 3 — the case for
// a circuit compiled with one base per wire, e.g. the wide layered workload): their unrolled loops are compiled
// out, which keeps the kernel at 45 registers = 8 wavefronts per SIMD instead of 6; stray gates of those classes
// take the common per-gate loop.
// UXOR: unroll depth of the Xor classes when it differs from the Mul classes' (0 = the same) -- the single-workgroup
// kernel runs 8-gate Xor steps on circuits without multi-base gates
template <int MODE, int NQ, bool ROTATE, bool GENERAL = true, bool PF = false, int UXOR = 0, int COH = 0>
__device__ __forceinline__ void run_level(const Gate* __restrict__ gates, const LevelRange& r, const InterpParams& p, uint32_t wave,
                                          uint32_t n_waves, uint32_t lane, uint32_t onm, const Gate* pf_gates = nullptr,
                                          const PfPlan* pf = nullptr) {
    constexpr uint32_t GPW = 64 / NQ;  // gates per wavefront per step
    const uint32_t q = lane % NQ, sub = lane / NQ;
    constexpr int U = interp_unroll(NQ, GENERAL);
    constexpr int UX = UXOR ? UXOR : U;
    static_assert(!PF || UX == U, "the prefetch plan assumes one step size");
    uint32_t slot = 0;
    auto my = [&](uint32_t used) { return ROTATE ? (wave + n_waves - used % n_waves) % n_waves : wave; };
    const uint32_t begin[5] = {r.lo, r.mul11, r.mul, r.xor2, r.xork}, end[5] = {r.mul11, r.mul, r.xor2, r.xork, r.hi};
    uint32_t rest[5];  // first gate of each class that is left to the common loop
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // without the multi-base loops (GENERAL = false) the few gates of those classes all go to the common loop
        const uint32_t STEP = (uint32_t)(c >= 2 ? UX : U) * GPW;
        const uint32_t n_full = (!GENERAL && (c == 1 || c == 3)) ? 0u : (end[c] - begin[c]) / STEP;
        rest[c] = begin[c] + n_full * STEP;
        for (uint32_t g0 = begin[c] + my(slot) * STEP; g0 < rest[c]; g0 += n_waves * STEP) {
            const Gate* t = nullptr;
            if (PF && pf->dist) t = pf_target<U * GPW>(pf_gates, *pf, slot + (g0 - begin[c]) / STEP);
            if (c == 0) mulU<MODE, NQ, U, 1, 1, COH>(gates, g0, p, sub, q, onm, t);               // G_MUL, one base per operand
            if (c == 1 && GENERAL) mulU<MODE, NQ, U, RV_LIN_K, RV_LIN_K, COH>(gates, g0, p, sub, q, onm, t); // other G_MUL
            if (c == 2) xorU<MODE, NQ, UX, 2, COH>(gates, g0, p, sub, q, t);                            // G_XORK of two bases
            if (c == 3 && GENERAL) xorU<MODE, NQ, UX, 2 * RV_LIN_K, COH>(gates, g0, p, sub, q, t);                 // other G_XORK
        }
        slot += n_full;
    }
    rest[4] = begin[4];
    uint32_t cum[6];  // wave-steps (GPW gates each) of the common loop, per class
    cum[0] = 0;
#pragma unroll
    for (int c = 0; c < 5; c++) cum[c + 1] = cum[c] + (end[c] - rest[c] + GPW - 1) / GPW;
    for (uint32_t t = my(slot); t < cum[5]; t += n_waves) {
        uint32_t c0 = rest[0], e0 = end[0], base = 0;
#pragma unroll
        for (int c = 1; c < 5; c++)
            if (t >= cum[c]) c0 = rest[c], e0 = end[c], base = cum[c];
        const uint32_t gi = c0 + (t - base) * GPW + sub;
        if (gi < e0) interp_one_impl<MODE, COH>(gates[gi], p, NQ, q, onm);
    }
}

// (the full-width variant without the multi-base loops must fit eight wavefronts per SIMD: its verify-mode instance
// took 70 registers = seven; with the bound it is 61, without scratch.  Narrower rows keep the default: they would spill)
template <int MODE, int NQ, bool GENERAL>
__global__ __launch_bounds__(256, MODE == MODE_VERIFY_C ? (GENERAL ? 4 : 7) : (GENERAL || NQ != 64) ? 1 : 8) void k_interp_full(const Gate* __restrict__ gates, LevelRange r, InterpParams p, PfPlan pf) {
    // a level's wavefronts are short-lived and wait on memory most of the time; when the lane-distributed mask generator shares the
    // SIMD (api.hip: RV_OVERLAP) its two long-lived, always-ready wavefronts are the OLDEST and win every issue slot -- the level ran
    // 3.3x slower beside it until its own wavefronts asked for priority
    __builtin_amdgcn_s_setprio(1);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t onm = (is_verify(MODE)) ? p.on_mask[lane % NQ] : 0u;
    // ROTATE here too: a wavefront then runs ONE step of one class instead of a Mul step followed by an Xor step
    // (two generations of short-lived wavefronts beat one generation of twice-as-long ones: 2.52 -> 2.40 ms;
    // interleaving the two classes wave by wave instead of class after class is worse again, 2.56)
    constexpr int UXL = (!GENERAL && NQ == 64) ? RV_INTERP_UXOR_FAST : 0;
    constexpr bool PFL = UXL == 0 || UXL == interp_unroll(NQ, GENERAL);  // (the prefetch plan assumes one step size)
    run_level<MODE, NQ, true, GENERAL, PFL, UXL>(gates, r, p, wave, n_waves, lane, onm, gates, &pf);
}

// Batched proofs of one circuit (rv_prove_batch): blockIdx.y selects the proof; its buffers come from a device array
// of InterpParams.  The gate stream is shared, so one launch per level serves every proof in the batch.
template <int MODE, int NQ>
__global__ __launch_bounds__(256) void k_interp_full_b(const Gate* __restrict__ gates, LevelRange r, const InterpParams* __restrict__ pp) {
    const InterpParams p = pp[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t onm = (is_verify(MODE)) ? p.on_mask[lane % NQ] : 0u;
    run_level<MODE, NQ, true>(gates, r, p, wave, n_waves, lane, onm);
}

// enough multi-base Mul / Xor gates in a level to be worth the variant with their unrolled loops?  (a handful
// -- constant operands in an otherwise one-base circuit -- run through the common per-gate loop instead)
static bool level_is_general(const LevelRange& r) { return (r.mul - r.mul11) + (r.xork - r.xor2) >= 64; }

// wavefronts that prefetch gate records look this many unrolled steps ahead: half a generation of resident
// wavefronts (8 per SIMD x 4 x 256 CUs = 8 192) measured best on full-width rows (interpreter 2.53 -> 2.42 ms on the
// 10^7-gate circuit; 2 048: 2.47, 8 192: 2.49, 16 384: 2.52).  Narrower rows (repetition shards) read their records
// through vector loads and gain nothing, so the default there is off.  RV_PF_DIST overrides (0 = off).
static uint32_t pf_dist(int NQ) {
    static const int env = [] {
        const char* e = getenv("RV_PF_DIST");
        return e ? atoi(e) : -1;
    }();
    const uint32_t v = env >= 0 ? (uint32_t)env : (NQ == 64 ? 4096u : 0u);
    return v & ~31u;
}

template <int NQ>
static PfPlan make_pf_plan(const LevelRange& r, const LevelRange* next) {
    constexpr uint32_t GPW = 64 / NQ;
    PfPlan pf{};
    pf.dist = pf_dist(NQ);
    if (!pf.dist) return pf;
    const LevelRange* lr[2] = {&r, next};
    uint32_t total = 0;
    for (int k = 0; k < 2; k++) {
        if (!lr[k]) break;
        const LevelRange& x = *lr[k];
        const bool general = level_is_general(x);
        const uint32_t step = (uint32_t)interp_unroll(NQ, general) * GPW;
        const uint32_t begin[4] = {x.lo, x.mul11, x.mul, x.xor2}, end[4] = {x.mul11, x.mul, x.xor2, x.xork};
        for (int c = 0; c < 4; c++) {
            pf.start[k][c] = begin[c];
            pf.n[k][c] = (!general && (c == 1 || c == 3)) ? 0u : (end[c] - begin[c]) / step;
            if (k == 0) total += pf.n[k][c];
        }
    }
    pf.rem = total & 31u;
    return pf;
}

template <int NQ>
static void launch_interp_full(hipStream_t st, int mode, const Gate* d_gates, const LevelRange& r, const InterpParams& p,
                               const LevelRange* next) {
    constexpr uint32_t GPW = 64 / NQ;
    const bool general = level_is_general(r);
    const PfPlan pf = make_pf_plan<NQ>(r, next);
    const uint32_t u = (uint32_t)interp_unroll(NQ, general);
    uint64_t waves = ((uint64_t)(r.hi - r.lo) + u * GPW - 1) / (u * GPW);
    uint64_t blocks = (waves + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (mode == MODE_PROVE_V) {
        if (general)
            hipLaunchKernelGGL((k_interp_full<MODE_PROVE_V, NQ, true>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
        else
            hipLaunchKernelGGL((k_interp_full<MODE_PROVE_V, NQ, false>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
    } else if (mode == MODE_PROVE) {
        if (general)
            hipLaunchKernelGGL((k_interp_full<MODE_PROVE, NQ, true>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
        else
            hipLaunchKernelGGL((k_interp_full<MODE_PROVE, NQ, false>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
    } else if (mode == MODE_VERIFY_C && NQ == 64) {
        if (general)
            hipLaunchKernelGGL((k_interp_full<NQ == 64 ? MODE_VERIFY_C : MODE_VERIFY, NQ, true>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
        else
            hipLaunchKernelGGL((k_interp_full<NQ == 64 ? MODE_VERIFY_C : MODE_VERIFY, NQ, false>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
    } else {
        if (general)
            hipLaunchKernelGGL((k_interp_full<MODE_VERIFY, NQ, true>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
        else
            hipLaunchKernelGGL((k_interp_full<MODE_VERIFY, NQ, false>), dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r, p, pf);
    }
}

// (Rounds 2 and 4 built four more launch structures for the GF(2) prover -- persistent level kernels, the flat, split and chained
// schedules.  All byte-identical, all measured slower: DESIGN.md Appendix A; their last version is commit 56a26f2.)

// whether any level of the gate stream has enough multi-base gates for the kernel variants with their loops (the verifier's
// choice of MODE_VERIFY_C looks at it too)
bool persist_general(const LevelRange* lr, size_t n_levels) {
    for (size_t l = 0; l < n_levels; l++)
        if (level_is_general(lr[l])) return true;
    return false;
}

// Narrow levels (deep circuits: ripple-carry adders, AES/SHA rounds) would be launch-bound at one
// kernel per level (~4.6 us each).  A run of consecutive narrow levels is executed by ONE 1024-thread
// workgroup instead: level -> __syncthreads() -> level ...; all waves share the CU's L1, so the
// workgroup-scope barrier is all the ordering the row/corr hand-off between levels needs.
// Per level the dependent chain used to be level_start[l+1] -> gate record -> operand rows (three L2 round trips,
// 1.57 us per level on SHA-256); the level table of the run and a rolling window of gate records now sit in LDS
// (filled by coalesced loads, one refill per NARROW_WIN gates), and a level runs through the same 4-way unrolled
// class loops as a full launch, so it costs one round trip per 64 gates plus the barrier.
// NQ = 0: generic row width (one gate per NQ lanes, no unrolling).
constexpr uint32_t NARROW_MAX_LEVELS = 1024;  // levels per launch (longer runs are split)
constexpr uint32_t NARROW_WIN = 1024;         // gate records resident in LDS (48 KiB) >= 2 x the widest narrow level
// LEAN: the run holds no multi-base Mul / Xor gates (a circuit compiled with one base per wire, e.g. AES-128): their
// unrolled loops are left out and the Xor steps take 8 gates -- a level of ~20 Mul + ~75 Xor gates is then 5 + 10 steps,
// one round of the 16 wavefronts instead of two (AES-128: 2.7 -> 2.6 us per level; a level moves ~100 KB through ONE CU,
// which at 64 B/clk is 0.7 us of the 2.6)
template <int MODE, int NQT, bool LEAN = false>
__device__ __forceinline__ void interp_narrow_body(const Gate* __restrict__ gates, const LevelRange* __restrict__ level_range,
                                                   uint32_t l0, uint32_t l1, const InterpParams& p) {
    __shared__ LevelRange s_lr[NARROW_MAX_LEVELS];
    __shared__ __attribute__((aligned(16))) Gate s_g[NARROW_WIN];
    const uint32_t NQ = NQT ? (uint32_t)NQT : p.NQ;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t onm = (is_verify(MODE)) ? p.on_mask[NQT ? lane % NQ : threadIdx.x % NQ] : 0u;
    const uint32_t n_lv = l1 - l0;
    {
        const uint32_t* src = (const uint32_t*)(level_range + l0);
        uint32_t* dst = (uint32_t*)s_lr;
        for (uint32_t i = threadIdx.x; i < n_lv * (uint32_t)(sizeof(LevelRange) / 4); i += 1024) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t g_end = s_lr[n_lv - 1].hi;
    uint32_t win_lo = s_lr[0].lo, win_hi = win_lo;  // gates [win_lo, win_hi) are in s_g
    for (uint32_t l = 0; l < n_lv; l++) {
        const LevelRange r = s_lr[l];
        if (r.hi > win_hi) {  // workgroup-uniform: the previous level's barrier has retired every reader of the old window
            win_lo = r.lo;
            win_hi = (r.lo + NARROW_WIN < g_end) ? r.lo + NARROW_WIN : g_end;
            const uint4* src = (const uint4*)(gates + win_lo);
            uint4* dst = (uint4*)s_g;
            for (uint32_t i = threadIdx.x; i < (win_hi - win_lo) * (uint32_t)(sizeof(Gate) / 16); i += 1024) dst[i] = src[i];
            __syncthreads();
        }
        const Gate* g = s_g - win_lo;  // indexed by absolute gate number
        if (NQT) {
            if (LEAN)
                run_level<MODE, NQT ? NQT : 64, true, false, false, 8>(g, r, p, wave, 16, lane, onm);
            else
                run_level<MODE, NQT ? NQT : 64, true>(g, r, p, wave, 16, lane, onm);
        } else {
            const uint32_t q = threadIdx.x % NQ, worker = threadIdx.x / NQ, n_workers = 1024 / NQ;
            for (uint32_t gi = r.lo + worker; gi < r.hi; gi += n_workers) interp_one_impl<MODE>(g[gi], p, NQ, q, onm);
        }
        __syncthreads();
    }
}

template <int MODE, int NQT, bool LEAN = false>
__global__ __launch_bounds__(1024) void k_interp_narrow(const Gate* __restrict__ gates, const LevelRange* __restrict__ level_range,
                                                        uint32_t l0, uint32_t l1, InterpParams p) {
    interp_narrow_body<MODE, NQT, LEAN>(gates, level_range, l0, l1, p);
}
// batched proofs: one workgroup per proof (blockIdx.x), see k_interp_full_b
template <int MODE, int NQT>
__global__ __launch_bounds__(1024) void k_interp_narrow_b(const Gate* __restrict__ gates, const LevelRange* __restrict__ level_range,
                                                          uint32_t l0, uint32_t l1, const InterpParams* __restrict__ pp) {
    const InterpParams p = pp[blockIdx.x];
    interp_narrow_body<MODE, NQT>(gates, level_range, l0, l1, p);
}

template <int NQT, bool LEAN = false>
static void launch_narrow_nq(hipStream_t st, int mode, const Gate* d_gates, const LevelRange* d_lr, uint32_t a, uint32_t b,
                             const InterpParams& p) {
    if (mode == MODE_PROVE)
        hipLaunchKernelGGL((k_interp_narrow<MODE_PROVE, NQT, LEAN>), dim3(1), dim3(1024), 0, st, d_gates, d_lr, a, b, p);
    else
        hipLaunchKernelGGL((k_interp_narrow<MODE_VERIFY, NQT, LEAN>), dim3(1), dim3(1024), 0, st, d_gates, d_lr, a, b, p);
}

void launch_interp_narrow(hipStream_t st, int mode, const Gate* d_gates, const LevelRange* d_level_range, uint32_t l0, uint32_t l1,
                          int tiny, const InterpParams& p) {
    for (uint32_t a = l0; a < l1; a += NARROW_MAX_LEVELS) {
        const uint32_t b = (a + NARROW_MAX_LEVELS < l1) ? a + NARROW_MAX_LEVELS : l1;
        // NQT = 0 is the plain per-gate loop (also the fallback for row widths without a class-loop instantiation)
        switch (tiny == 1 ? 0u : p.NQ) {
        case 64:
            if (tiny == 2)
                launch_narrow_nq<64, true>(st, mode, d_gates, d_level_range, a, b, p);
            else
                launch_narrow_nq<64>(st, mode, d_gates, d_level_range, a, b, p);
            break;
        case 32: launch_narrow_nq<32>(st, mode, d_gates, d_level_range, a, b, p); break;
        case 16: launch_narrow_nq<16>(st, mode, d_gates, d_level_range, a, b, p); break;
        case 8: launch_narrow_nq<8>(st, mode, d_gates, d_level_range, a, b, p); break;
        default: launch_narrow_nq<0>(st, mode, d_gates, d_level_range, a, b, p); break;
        }
    }
}

void launch_interp(hipStream_t st, int mode, const Gate* d_gates, const LevelRange& r, const InterpParams& p, const LevelRange* next) {
    if (r.hi <= r.lo) return;
    switch (p.NQ) {
    case 64: return launch_interp_full<64>(st, mode, d_gates, r, p, next);
    case 32: return launch_interp_full<32>(st, mode, d_gates, r, p, next);
    case 16: return launch_interp_full<16>(st, mode, d_gates, r, p, next);
    case 8: return launch_interp_full<8>(st, mode, d_gates, r, p, next);
    default: break;
    }
    const uint64_t want = (uint64_t)(r.hi - r.lo) * p.NQ;
    uint64_t blocks = (want + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (mode == MODE_PROVE)
        hipLaunchKernelGGL(k_interp<MODE_PROVE>, dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r.lo, r.hi, p);
    else
        hipLaunchKernelGGL(k_interp<MODE_VERIFY>, dim3((unsigned)blocks), dim3(256), 0, st, d_gates, r.lo, r.hi, p);
}

// rv_prove_batch / rv_verify_batch: `batch` full proofs (256 repetitions, NQ = 64) of one circuit
void launch_interp_batched(hipStream_t st, const Gate* d_gates, const LevelRange& r, const InterpParams* d_pp, uint32_t batch, int mode) {
    if (r.hi <= r.lo || !batch) return;
    uint64_t waves = ((uint64_t)(r.hi - r.lo) + RV_INTERP_UNROLL - 1) / RV_INTERP_UNROLL;
    uint64_t blocks = (waves + 3) / 4;
    const uint64_t cap = std::max<uint64_t>(4096 / batch, 1);
    if (blocks > cap) blocks = cap;
    if (mode == MODE_PROVE)
        hipLaunchKernelGGL((k_interp_full_b<MODE_PROVE, 64>), dim3((unsigned)blocks, batch), dim3(256), 0, st, d_gates, r, d_pp);
    else
        hipLaunchKernelGGL((k_interp_full_b<MODE_VERIFY, 64>), dim3((unsigned)blocks, batch), dim3(256), 0, st, d_gates, r, d_pp);
}

void launch_interp_narrow_batched(hipStream_t st, const Gate* d_gates, const LevelRange* d_level_range, uint32_t l0, uint32_t l1,
                                  int tiny, const InterpParams* d_pp, uint32_t batch, int mode) {
    for (uint32_t a = l0; a < l1 && batch; a += NARROW_MAX_LEVELS) {
        const uint32_t b = (a + NARROW_MAX_LEVELS < l1) ? a + NARROW_MAX_LEVELS : l1;
        if (mode == MODE_PROVE) {
            if (tiny == 1)
                hipLaunchKernelGGL((k_interp_narrow_b<MODE_PROVE, 0>), dim3(batch), dim3(1024), 0, st, d_gates, d_level_range, a, b, d_pp);
            else
                hipLaunchKernelGGL((k_interp_narrow_b<MODE_PROVE, 64>), dim3(batch), dim3(1024), 0, st, d_gates, d_level_range, a, b, d_pp);
        } else {
            if (tiny == 1)
                hipLaunchKernelGGL((k_interp_narrow_b<MODE_VERIFY, 0>), dim3(batch), dim3(1024), 0, st, d_gates, d_level_range, a, b, d_pp);
            else
                hipLaunchKernelGGL((k_interp_narrow_b<MODE_VERIFY, 64>), dim3(batch), dim3(1024), 0, st, d_gates, d_level_range, a, b, d_pp);
        }
    }
}

// ------------------------------------------------------------------------------------
// BLAKE3 over row-format transcripts.  Thread = (chunk, quad): reads 64 rows per block
// (coalesced across the quads of a row), de-interleaves the 4 repetitions of its quad
// word into 4 x 16 message words and runs the 4 compressions back to back.
// ------------------------------------------------------------------------------------
// RPL = repetitions per lane (4: one lane per quad word; 1: four lanes share a quad word).  Fewer
// repetitions per lane = more, lighter wavefronts: 4 900 chunks x 64 lanes is only 1.6 rounds of the
// chip at 3 waves/SIMD (40 % of the time is tail), RPL = 1 gives 19 600 waves at 7+ waves/SIMD.
// UNI (RPL = 4, full-width rows, no quad list: the prover's whole proofs): a chunk per wavefront, lane = quad word.  The chunk index
// is then wave-uniform BY CONSTRUCTION, so a block's 64 row loads take a scalar base and one shared 32-bit lane offset instead of
// 64 vector address computations (128 of ~3 200 VALU instructions per block).
template <int RPL, bool UNI = false>
struct B_k_b3_chunks {
    // quads (nullable) / n_quads: only these quad words are hashed -- the verifier needs the online digest of the 40
    // opened repetitions alone (the other 216 carry theirs in the proof), i.e. of at most 40 of the 64 quads
    // chunk_base / root_ok (streaming prover): the stream handed in is a piece of a longer one -- its first chunk has
    // BLAKE3 chunk counter chunk_base, and a lone chunk only takes the ROOT flag when the caller knows it is the whole stream
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs /*[n_chunks][R][8]*/, const uint32_t* __restrict__ quads, uint32_t n_quads, uint64_t chunk_base, uint32_t root_ok) const {
    run((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stream, n_events, NQ, n_chunks, cvs, quads, n_quads, chunk_base, root_ok);
    }
    static __device__ __forceinline__ void run(uint64_t tid, const uint32_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs, const uint32_t* __restrict__ quads, uint32_t n_quads, uint64_t chunk_base, uint32_t root_ok) {
    constexpr uint32_t SUBS = 4 / RPL;
    static_assert(!UNI || RPL == 4, "a chunk per wavefront needs one lane per quad word");
    const uint32_t lanes_per_chunk = UNI ? 64u : (quads ? n_quads : NQ) * SUBS;
    uint64_t c = tid / lanes_per_chunk;
    if (UNI) c = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)c) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32);
    const uint32_t ql = (uint32_t)(tid % lanes_per_chunk);
    const uint32_t q = UNI ? ql : (quads ? quads[ql / SUBS] : ql / SUBS), sub = ql % SUBS;
    if (c >= n_chunks) return;
    const uint64_t ev0 = c * 1024;
    const uint64_t len = (n_events - ev0 < 1024) ? (n_events - ev0) : 1024;
    const uint32_t nblk = len == 0 ? 1 : (uint32_t)((len + 63) / 64);
    uint32_t cv[RPL][8];
#pragma unroll
    for (int i = 0; i < RPL; i++) b3::iv(cv[i]);
    for (uint32_t b = 0; b < nblk; b++) {
        const uint64_t e0 = ev0 + 64ull * b;
        const uint32_t blen = (b + 1 < nblk) ? 64u : (uint32_t)(len - 64ull * b);
        uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) | (b + 1 == nblk ? b3::CHUNK_END : 0u);
        if (b + 1 == nblk && n_chunks == 1 && root_ok) flags |= b3::ROOT;
        uint32_t w[64];
        if (blen == 64) {
            // unguarded: a per-element "load or zero" select makes hipcc branch around every load and
            // wait for it (64 dependent round trips per block)
            if (UNI) {
                const char* rb = (const char*)(stream + e0 * 64);
                const uint32_t qoff = q * 4u;
#pragma unroll
                for (int e = 0; e < 64; e++) w[e] = *(const uint32_t*)(rb + e * 256 + qoff);
            } else {
#pragma unroll
                for (int e = 0; e < 64; e++) w[e] = stream[(e0 + e) * NQ + q];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 64; e++) w[e] = (e0 + e < n_events) ? stream[(e0 + e) * NQ + q] : 0u;
        }
        uint32_t m[RPL][16];
#pragma unroll
        for (int i = 0; i < RPL; i++) {
            const uint32_t i4 = sub * RPL + i;  // repetition inside the quad word; its byte counts from the MSB
            const uint32_t sel = 3 - i4;        // byte index for v_perm (0 = LSB)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                // m = byte(w[4k]) | byte(w[4k+1]) << 8 | byte(w[4k+2]) << 16 | byte(w[4k+3]) << 24: two v_perm_b32
                const uint32_t lo = __builtin_amdgcn_perm(w[4 * k + 1], w[4 * k], 0x0c0c0400u + sel * 0x0101u);
                const uint32_t hi = __builtin_amdgcn_perm(w[4 * k + 3], w[4 * k + 2], 0x0c0c0400u + sel * 0x0101u);
                m[i][k] = lo | (hi << 16);
            }
        }
        b3::compress_n<RPL>(cv, m, c + chunk_base, blen, flags);  // the lane's repetitions in lockstep
    }
    const uint32_t R = NQ * 4;
#pragma unroll
    for (int i = 0; i < RPL; i++) {
        uint32_t* dst = cvs + ((size_t)c * R + 4 * q + sub * RPL + i) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = cv[i][k];
    }
}
};
template <int RPL>
__global__ __launch_bounds__(256) void k_b3_chunks(const uint32_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs /*[n_chunks][R][8]*/, const uint32_t* __restrict__ quads, uint32_t n_quads, uint64_t chunk_base, uint32_t root_ok) {
    B_k_b3_chunks<RPL>{}(stream, n_events, NQ, n_chunks, cvs, quads, n_quads, chunk_base, root_ok);
}
__global__ __launch_bounds__(256) void k_b3_chunks_uni(const uint32_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs /*[n_chunks][R][8]*/, const uint32_t* __restrict__ quads, uint32_t n_quads, uint64_t chunk_base, uint32_t root_ok) {
    B_k_b3_chunks<4, true>{}(stream, n_events, NQ, n_chunks, cvs, quads, n_quads, chunk_base, root_ok);
}

// Same for a bit-per-rep transcript (the preprocessing stream): every bit is hashed as the
// 0x00/0xFF byte the reference feeds its hasher (gf2/recon.rs:314-321).
// RPL as in k_b3_chunks: 4 = one lane per quad word, 1 = four lanes share it (short transcripts: more, lighter wavefronts)
template <int RPL, bool UNI = false>
struct B_k_b3_chunks_bits {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs, uint64_t chunk_base, uint32_t root_ok) const {
    run((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stream, n_events, NQ, n_chunks, cvs, chunk_base, root_ok);
    }
    static __device__ __forceinline__ void run(uint64_t tid, const uint8_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs, uint64_t chunk_base, uint32_t root_ok) {
    constexpr uint32_t SUBS = 4 / RPL;
    static_assert(!UNI || RPL == 4, "a chunk per wavefront needs one lane per quad word");
    const uint32_t lanes_per_chunk = UNI ? 64u : NQ * SUBS;  // (UNI: NQ = 64, see B_k_b3_chunks)
    uint64_t c = tid / lanes_per_chunk;
    if (UNI) c = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)c) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32);
    const uint32_t ql = (uint32_t)(tid % lanes_per_chunk);
    const uint32_t q = ql / SUBS, sub = ql % SUBS;
    if (c >= n_chunks) return;
    const uint64_t ev0 = c * 1024;
    const uint64_t len = (n_events - ev0 < 1024) ? (n_events - ev0) : 1024;
    const uint32_t nblk = len == 0 ? 1 : (uint32_t)((len + 63) / 64);
    const uint32_t h = NQ >> 1, o = q >> 1, sh = 4 * (q & 1);
    uint32_t cv[RPL][8];
#pragma unroll
    for (int i = 0; i < RPL; i++) b3::iv(cv[i]);
    for (uint32_t b = 0; b < nblk; b++) {
        const uint64_t e0 = ev0 + 64ull * b;
        const uint32_t blen = (b + 1 < nblk) ? 64u : (uint32_t)(len - 64ull * b);
        uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) | (b + 1 == nblk ? b3::CHUNK_END : 0u);
        if (b + 1 == nblk && n_chunks == 1 && root_ok) flags |= b3::ROOT;
        // P = the nibbles of events 4k..4k+3, one per byte; repetition i4 owns nibble bit 3-i4
        uint32_t m[RPL][16];
        uint32_t nbs[64];
        if (blen == 64) {
            if (UNI) {
                const uint8_t* rb = stream + e0 * 32;
#pragma unroll
                for (int e = 0; e < 64; e++) nbs[e] = *(rb + e * 32 + o);
            } else {
#pragma unroll
                for (int e = 0; e < 64; e++) nbs[e] = stream[(e0 + e) * h + o];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 64; e++) nbs[e] = (e0 + e < n_events) ? (uint32_t)stream[(e0 + e) * h + o] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t P = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) P |= ((nbs[4 * k + j] >> sh) & 0xFu) << (8 * j);
#pragma unroll
            for (int i = 0; i < RPL; i++) {
                const uint32_t i4 = sub * RPL + i;
                const uint32_t t = (P >> (3 - i4)) & 0x01010101u;
                m[i][k] = (t << 8) - t;
            }
        }
        b3::compress_n<RPL>(cv, m, c + chunk_base, blen, flags);
    }
    const uint32_t R = NQ * 4;
#pragma unroll
    for (int i = 0; i < RPL; i++) {
        uint32_t* dst = cvs + ((size_t)c * R + 4 * q + sub * RPL + i) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = cv[i][k];
    }
}
};
__global__ __launch_bounds__(256) void k_b3_chunks_bits(const uint8_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs, uint64_t chunk_base, uint32_t root_ok) {
    B_k_b3_chunks_bits<4>{}(stream, n_events, NQ, n_chunks, cvs, chunk_base, root_ok);
}
__global__ __launch_bounds__(256) void k_b3_chunks_bits_uni(const uint8_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs, uint64_t chunk_base, uint32_t root_ok) {
    B_k_b3_chunks_bits<4, true>{}(stream, n_events, NQ, n_chunks, cvs, chunk_base, root_ok);
}
__global__ __launch_bounds__(256) void k_b3_chunks_bits1(const uint8_t* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs, uint64_t chunk_base, uint32_t root_ok) {
    B_k_b3_chunks_bits<1>{}(stream, n_events, NQ, n_chunks, cvs, chunk_base, root_ok);
}

// LG tree levels per launch: thread = (group of G = 2^LG consecutive nodes, repetition).  One level is
// out[i] = parent(in[2i], in[2i+1]) with an odd last node promoted unchanged; groups are aligned to G, so
// reducing a group locally level by level gives exactly the nodes LG global levels would (the ragged
// last group follows the same promote rule).  The ROOT flag belongs to the merge of the last two nodes of
// the whole tree, which can only happen inside the only group of a launch.
template <int LG>
struct B_k_b3_reduce {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ in, uint64_t n_in, uint32_t R, uint32_t* __restrict__ out) const {
    run((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, in, n_in, R, out);
    }
    static __device__ __forceinline__ void run(uint64_t tid, const uint32_t* __restrict__ in, uint64_t n_in, uint32_t R, uint32_t* __restrict__ out) {
    constexpr int G = 1 << LG;
    const uint64_t n_out = (n_in + G - 1) / G;
    const uint64_t g = tid / R;
    const uint32_t r = (uint32_t)(tid % R);
    if (g >= n_out) return;
    uint32_t cnt = (uint32_t)((n_in - G * g < (uint64_t)G) ? n_in - G * g : G);
    uint32_t cv[G][8];
#pragma unroll
    for (int i = 0; i < G; i++) {
        if ((uint32_t)i < cnt) {
            const uint4* src = (const uint4*)(in + ((size_t)(G * g + i) * R + r) * 8);
            const uint4 lo = src[0], hi = src[1];
            cv[i][0] = lo.x; cv[i][1] = lo.y; cv[i][2] = lo.z; cv[i][3] = lo.w;
            cv[i][4] = hi.x; cv[i][5] = hi.y; cv[i][6] = hi.z; cv[i][7] = hi.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) cv[i][k] = 0;
        }
    }
#pragma unroll
    for (int lvl = 0; lvl < LG; lvl++) {
        const uint32_t flags = (n_out == 1 && cnt == 2) ? b3::ROOT : 0u;
#pragma unroll
        for (int i = 0; i < (G >> (lvl + 1)); i++) {
            if ((uint32_t)(2 * i + 1) < cnt) {
                uint32_t o[8];
                b3::parent(cv[2 * i], cv[2 * i + 1], flags, o);
#pragma unroll
                for (int k = 0; k < 8; k++) cv[i][k] = o[k];
            } else if ((uint32_t)(2 * i) < cnt) {
#pragma unroll
                for (int k = 0; k < 8; k++) cv[i][k] = cv[2 * i][k];
            }
        }
        cnt = (cnt + 1) / 2;
    }
    uint4* d = (uint4*)(out + ((size_t)g * R + r) * 8);
    d[0] = make_uint4(cv[0][0], cv[0][1], cv[0][2], cv[0][3]);
    d[1] = make_uint4(cv[0][4], cv[0][5], cv[0][6], cv[0][7]);
}
};
template <int LG>
__global__ __launch_bounds__(256) void k_b3_reduce(const uint32_t* __restrict__ in, uint64_t n_in, uint32_t R, uint32_t* __restrict__ out) {
    B_k_b3_reduce<LG>{}(in, n_in, R, out);
}

// The top of the tree (at most B3_TAIL nodes per repetition): one workgroup per repetition walks the
// remaining levels through LDS, a barrier per level instead of a launch per level.
constexpr uint32_t B3_TAIL = 512;
// CAP = most nodes the workgroup takes (its LDS footprint): B3_TAIL with 256 threads, or 64 with one wavefront for the
// short transcripts of small circuits -- 2.3 KB instead of 18 KB of LDS, so a batch of proofs gets four times the
// workgroups per CU
template <int CAP>
struct B_k_b3_tree_tail {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ in, uint32_t n_in, uint32_t R, uint32_t* __restrict__ digest) const {
    run(blockIdx.x, in, n_in, R, digest);
    }
    static __device__ __forceinline__ void run(uint32_t r, const uint32_t* __restrict__ in, uint32_t n_in, uint32_t R, uint32_t* __restrict__ digest) {
    __shared__ uint32_t cv[CAP][8 + 1];  // +1: odd row stride, no bank conflicts on the strided pair reads
    for (uint32_t i = threadIdx.x; i < n_in * 8; i += blockDim.x) cv[i >> 3][i & 7] = in[((size_t)(i >> 3) * R + r) * 8 + (i & 7)];
    __syncthreads();
    uint32_t cnt = n_in;
    while (cnt > 1) {
        const uint32_t half = (cnt + 1) / 2;
        const uint32_t i = threadIdx.x;
        uint32_t o[8];
        if (i < half) {
            if (2 * i + 1 < cnt) {
                uint32_t l[8], rr[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    l[k] = cv[2 * i][k];
                    rr[k] = cv[2 * i + 1][k];
                }
                b3::parent(l, rr, cnt == 2 ? b3::ROOT : 0u, o);
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) o[k] = cv[2 * i][k];
            }
        }
        __syncthreads();
        if (i < half) {
#pragma unroll
            for (int k = 0; k < 8; k++) cv[i][k] = o[k];
        }
        __syncthreads();
        cnt = half;
    }
    if (threadIdx.x < 8) digest[(size_t)r * 8 + threadIdx.x] = cv[0][threadIdx.x];
}
};
__global__ __launch_bounds__(256) void k_b3_tree_tail(const uint32_t* __restrict__ in, uint32_t n_in, uint32_t R, uint32_t* __restrict__ digest) {
    B_k_b3_tree_tail<(int)B3_TAIL>{}(in, n_in, R, digest);
}
__global__ __launch_bounds__(64) void k_b3_tree_tail_small(const uint32_t* __restrict__ in, uint32_t n_in, uint32_t R, uint32_t* __restrict__ digest) {
    B_k_b3_tree_tail<64>{}(in, n_in, R, digest);
}

// The same top of the tree with ONE LANE per repetition (at most 64 chaining values): the lane folds its values the way
// the incremental hasher does -- complete subtrees of 2^k chunks wait in slot k, the last value closes them from the
// smallest up and the last parent carries ROOT -- n - 1 dependent compressions, but 64 repetitions per wavefront instead
// of one.  For a batch of proofs that is the difference between 65 536 workgroups of one mostly idle wavefront each
// (rv_prove_batch of 256 AES-128 proofs: 2 x 261 us) and 1 024 full wavefronts.
struct B_k_b3_tree_lane {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ in, uint32_t n_in, uint32_t R, uint32_t* __restrict__ digest) const {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    auto load = [&](uint32_t i, uint32_t* cv) {
        const uint4* p = (const uint4*)(in + ((size_t)i * R + r) * 8);
        const uint4 a = p[0], b = p[1];
        cv[0] = a.x, cv[1] = a.y, cv[2] = a.z, cv[3] = a.w, cv[4] = b.x, cv[5] = b.y, cv[6] = b.z, cv[7] = b.w;
    };
    uint32_t st[6][8], cur[8], o[8];
    for (uint32_t i = 0; i + 1 < n_in; i++) {  // (uniform: every lane walks the same tree shape)
        load(i, cur);
        bool placed = false;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (placed) continue;
            if ((i >> k) & 1u) {
                b3::parent(st[k], cur, 0u, o);
#pragma unroll
                for (int w = 0; w < 8; w++) cur[w] = o[w];
            } else {
#pragma unroll
                for (int w = 0; w < 8; w++) st[k][w] = cur[w];
                placed = true;
            }
        }
    }
    const uint32_t last = n_in - 1;
    load(last, cur);  // a single chunk is already its own root (the chunk kernels applied the ROOT flag)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if ((last >> k) & 1u) {
            b3::parent(st[k], cur, (last >> (k + 1)) == 0 ? b3::ROOT : 0u, o);
#pragma unroll
            for (int w = 0; w < 8; w++) cur[w] = o[w];
        }
    }
    uint4* d = (uint4*)(digest + (size_t)r * 8);
    d[0] = make_uint4(cur[0], cur[1], cur[2], cur[3]);
    d[1] = make_uint4(cur[4], cur[5], cur[6], cur[7]);
}
};
__global__ __launch_bounds__(64) void k_b3_tree_lane(const uint32_t* __restrict__ in, uint32_t n_in, uint32_t R, uint32_t* __restrict__ digest) {
    B_k_b3_tree_lane{}(in, n_in, R, digest);
}

// Both transcripts of a small proof in the same two launches.  With a handful of chunks per stream the chunk kernels are
// one dependent chain of 16 compressions per lane (~37 us) whatever the number of lanes, and the tree tops a few more:
// run back to back, preprocessing then online, they are ~90 us of a 0.55 ms AES-128 proof.  The first blocks of a paired
// launch take the preprocessing stream, the rest the online one (one repetition per lane in both).
struct B_k_b3_chunks_pair {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ pre, uint64_t n_pre, uint32_t* __restrict__ cv_pre, const uint32_t* __restrict__ on,
                                               uint64_t n_on, uint32_t* __restrict__ cv_on, uint32_t NQ, uint32_t blocks_pre, const uint32_t* __restrict__ quads,
                                               uint32_t n_quads) const {
    const uint64_t c_pre = n_pre == 0 ? 1 : (n_pre + 1023) / 1024, c_on = n_on == 0 ? 1 : (n_on + 1023) / 1024;
    if (blockIdx.x < blocks_pre)
        B_k_b3_chunks_bits<1>::run((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, pre, n_pre, NQ, c_pre, cv_pre, 0, 1);
    else
        B_k_b3_chunks<1>::run((uint64_t)(blockIdx.x - blocks_pre) * blockDim.x + threadIdx.x, on, n_on, NQ, c_on, cv_on, quads, n_quads, 0, 1);
    }
};
__global__ __launch_bounds__(256) void k_b3_chunks_pair(const uint8_t* __restrict__ pre, uint64_t n_pre, uint32_t* __restrict__ cv_pre, const uint32_t* __restrict__ on,
                                                        uint64_t n_on, uint32_t* __restrict__ cv_on, uint32_t NQ, uint32_t blocks_pre,
                                                        const uint32_t* __restrict__ quads, uint32_t n_quads) {
    B_k_b3_chunks_pair{}(pre, n_pre, cv_pre, on, n_on, cv_on, NQ, blocks_pre, quads, n_quads);
}
// ONE small proof (no batch to supply wavefronts): a chunk's 16 chained compressions are the whole duration of the chunk launch, so
// each repetition's chain runs on a QUAD of lanes (b3.h: compress_q) -- lane = (chunk, repetition, column).  A lane assembles the four
// message words of its column's share of the block (16 of the block's 64 rows), the quad exchanges them through LDS.  ~37 -> ~15 us.
// BITS: the preprocessing stream (a bit per repetition and event); else the online stream (a byte).  quads / n_quads as in
// B_k_b3_chunks; a quad of lanes never straddles two chunks, so its four lanes always run the same number of blocks.
template <bool BITS>
struct B_k_b3_chunks_q {
    static __device__ __forceinline__ void run(uint64_t tid, const void* __restrict__ stream, uint64_t n_events, uint32_t NQ, uint64_t n_chunks, uint32_t* __restrict__ cvs,
                                               const uint32_t* __restrict__ quads, uint32_t n_quads, uint32_t* s_msg /* [threads / 4][16] */) {
    const uint32_t qc = (uint32_t)(tid & 3);
    const uint32_t reps_per_chunk = (quads ? n_quads : NQ) * 4;
    const uint64_t gi = tid >> 2;  // (chunk, repetition slot)
    const uint64_t c = gi / reps_per_chunk;
    const uint32_t rs = (uint32_t)(gi % reps_per_chunk);
    if (c >= n_chunks) return;  // (whole quads leave together)
    const uint32_t q = quads ? quads[rs >> 2] : rs >> 2, i4 = rs & 3;
    uint32_t* const msg = s_msg + (threadIdx.x >> 2) * 16;
    const b3::QuadSchedule qs = b3::quad_schedule(qc);
    const uint64_t ev0 = c * 1024;
    const uint64_t len = (n_events - ev0 < 1024) ? (n_events - ev0) : 1024;
    const uint32_t nblk = len == 0 ? 1 : (uint32_t)((len + 63) / 64);
    uint32_t cva = qc == 0 ? B3_IV0 : qc == 1 ? B3_IV1 : qc == 2 ? B3_IV2 : B3_IV3;
    uint32_t cvb = qc == 0 ? B3_IV4 : qc == 1 ? B3_IV5 : qc == 2 ? B3_IV6 : B3_IV7;
    // this lane's 16 rows of block b (message words 4 qc .. 4 qc + 3), one word each (a byte of the bit rows); the rows of block b + 1
    // are requested before block b is compressed -- a memory round trip per block was most of the chain
    const uint32_t h = NQ >> 1, o = q >> 1, sh = 4 * (q & 1);
    auto load_rows = [&](uint32_t b, uint32_t (&w)[16]) {
        const uint64_t e0 = ev0 + 64ull * b + 16ull * qc;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            if (BITS)
                w[e] = (e0 + e < n_events) ? (uint32_t)((const uint8_t*)stream)[(e0 + e) * h + o] : 0u;
            else
                w[e] = (e0 + e < n_events) ? ((const uint32_t*)stream)[(e0 + e) * NQ + q] : 0u;
        }
    };
    uint32_t w[16];
    load_rows(0, w);
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t blen = (b + 1 < nblk) ? 64u : (uint32_t)(len - 64ull * b);
        uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) | (b + 1 == nblk ? b3::CHUNK_END : 0u);
        if (b + 1 == nblk && n_chunks == 1) flags |= b3::ROOT;
        uint32_t m4[4];
        if (BITS) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                uint32_t P = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) P |= ((w[4 * kk + j] >> sh) & 0xFu) << (8 * j);
                const uint32_t t = (P >> (3 - i4)) & 0x01010101u;
                m4[kk] = (t << 8) - t;
            }
        } else {
            const uint32_t sel = 3 - i4;  // byte index for v_perm (0 = LSB): the repetition's byte counts from the MSB
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint32_t lo = __builtin_amdgcn_perm(w[4 * kk + 1], w[4 * kk], 0x0c0c0400u + sel * 0x0101u);
                const uint32_t hi = __builtin_amdgcn_perm(w[4 * kk + 3], w[4 * kk + 2], 0x0c0c0400u + sel * 0x0101u);
                m4[kk] = lo | (hi << 16);
            }
        }
        if (b + 1 < nblk) load_rows(b + 1, w);
        // (the quad's lanes sit in one wavefront, whose LDS accesses execute in order: the block's reads of the step before are done)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) msg[4 * qc + kk] = m4[kk];
        __builtin_amdgcn_wave_barrier();
        b3::compress_q<false>(cva, cvb, msg, qs, qc, c, blen, flags);
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t R = NQ * 4;
    uint32_t* dst = cvs + ((size_t)c * R + 4 * q + i4) * 8;
    dst[qc] = cva;
    dst[4 + qc] = cvb;
    }
};
struct B_k_b3_chunks_pair_q {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ pre, uint64_t n_pre, uint32_t* __restrict__ cv_pre, const uint32_t* __restrict__ on,
                                               uint64_t n_on, uint32_t* __restrict__ cv_on, uint32_t NQ, uint32_t blocks_pre, const uint32_t* __restrict__ quads,
                                               uint32_t n_quads) const {
    __shared__ uint32_t s_msg[64 * 16];
    const uint64_t c_pre = n_pre == 0 ? 1 : (n_pre + 1023) / 1024, c_on = n_on == 0 ? 1 : (n_on + 1023) / 1024;
    if (blockIdx.x < blocks_pre)
        B_k_b3_chunks_q<true>::run((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, pre, n_pre, NQ, c_pre, cv_pre, nullptr, 0, s_msg);
    else
        B_k_b3_chunks_q<false>::run((uint64_t)(blockIdx.x - blocks_pre) * blockDim.x + threadIdx.x, on, n_on, NQ, c_on, cv_on, quads, n_quads, s_msg);
    }
};
__global__ __launch_bounds__(256) void k_b3_chunks_pair_q(const uint8_t* __restrict__ pre, uint64_t n_pre, uint32_t* __restrict__ cv_pre, const uint32_t* __restrict__ on,
                                                          uint64_t n_on, uint32_t* __restrict__ cv_on, uint32_t NQ, uint32_t blocks_pre,
                                                          const uint32_t* __restrict__ quads, uint32_t n_quads) {
    B_k_b3_chunks_pair_q{}(pre, n_pre, cv_pre, on, n_on, cv_on, NQ, blocks_pre, quads, n_quads);
}
// tree tops of both: workgroups [0, R) the preprocessing stream, [R, 2R) the online one
struct B_k_b3_tree_tail_pair {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ in_a, uint32_t n_a, uint32_t* __restrict__ dig_a, const uint32_t* __restrict__ in_b, uint32_t n_b,
                                               uint32_t* __restrict__ dig_b, uint32_t R) const {
    if (blockIdx.x < R)
        B_k_b3_tree_tail<64>::run(blockIdx.x, in_a, n_a, R, dig_a);
    else
        B_k_b3_tree_tail<64>::run(blockIdx.x - R, in_b, n_b, R, dig_b);
    }
};
__global__ __launch_bounds__(64) void k_b3_tree_tail_pair(const uint32_t* __restrict__ in_a, uint32_t n_a, uint32_t* __restrict__ dig_a, const uint32_t* __restrict__ in_b,
                                                          uint32_t n_b, uint32_t* __restrict__ dig_b, uint32_t R) {
    B_k_b3_tree_tail_pair{}(in_a, n_a, dig_a, in_b, n_b, dig_b, R);
}
// true (and two launches issued) when both streams are short enough for the paired kernels; d_cv_a / d_cv_b each hold one
// stream's chunk chaining values (the tree tops need no second buffer at this size).  d_quads / n_quads as in
// launch_b3_stream (the verifier hashes the online stream of the opened quads only; n_quads = 0 with a list: not paired)
bool launch_b3_pair_small(hipStream_t st, const uint8_t* d_pre, uint64_t n_pre, const uint32_t* d_on, uint64_t n_on, uint32_t NQ, uint32_t* d_cv_a,
                          uint32_t* d_cv_b, uint32_t* d_dig_pre, uint32_t* d_dig_on, const uint32_t* d_quads, uint32_t n_quads) {
    if (d_quads && !n_quads) return false;
    const uint64_t c_pre = n_pre == 0 ? 1 : (n_pre + 1023) / 1024, c_on = n_on == 0 ? 1 : (n_on + 1023) / 1024;
    const uint32_t batch = g_recorder ? g_recorder->batch : 1u;
    // (the same "few chunks" rule as the separate launchers' one-repetition-per-lane choice, and trees the small tail kernel takes)
    if (c_pre > 64 || c_on > 64 || std::max(c_pre, c_on) * NQ * batch >= 64 * 1024) return false;
    const uint32_t R = NQ * 4;
    if (!g_recorder) {
        // (one proof: a quad of lanes per repetition's chain)
        const uint32_t b_pre = (uint32_t)((c_pre * NQ * 16 + 255) / 256), b_on = (uint32_t)((c_on * (d_quads ? n_quads : NQ) * 16 + 255) / 256);
        hipLaunchKernelGGL(k_b3_chunks_pair_q, dim3(b_pre + b_on), dim3(256), 0, st, d_pre, n_pre, d_cv_a, d_on, n_on, d_cv_b, NQ, b_pre, d_quads, n_quads);
    } else {
    const uint32_t b_pre = (uint32_t)((c_pre * NQ * 4 + 255) / 256), b_on = (uint32_t)((c_on * (d_quads ? n_quads : NQ) * 4 + 255) / 256);
    launch<B_k_b3_chunks_pair, 256>(k_b3_chunks_pair, st, dim3(b_pre + b_on), dim3(256), d_pre, n_pre, d_cv_a, d_on, n_on, d_cv_b, NQ, b_pre, d_quads, n_quads);
    }
    launch<B_k_b3_tree_tail_pair, 64>(k_b3_tree_tail_pair, st, dim3(2 * R), dim3(64), (const uint32_t*)d_cv_a, (uint32_t)c_pre, d_dig_pre,
                                      (const uint32_t*)d_cv_b, (uint32_t)c_on, d_dig_on, R);
    return true;
}

// The trees of BOTH transcripts of a large proof in shared launches (blockIdx.y = the stream): after the two chunk kernels a whole
// proof of the 10^7-gate circuit ran two reduction launches and a tree top per stream, six dependent launches of ~20 us that each
// occupy a fraction of the chip -- three of them now.  A stream that is already at the tree top's size sits a reduction out.
// ... and their chunk kernels as ONE launch: the first workgroups hash the preprocessing stream (a bit per repetition), the others the online
// stream (a byte), a chunk per wavefront in both (B_k_b3_chunks<4, true>) -- the ragged last generation of the first fills with
// wavefronts of the second (on two streams that cost more in events than it gave: DESIGN.md section 4)
// (QUADS: the verifier -- the online stream of the quad words with an opened repetition only, a lane per listed quad word)
// (QUADS = 2: few listed quad words -- a quarter of the row or less --: a lane per REPETITION of them, as launch_b3_stream_chunks chooses)
template <int QUADS>
struct B_k_b3_chunks_pair_uni {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ pre, uint64_t n_pre, uint32_t* __restrict__ cv_pre, const uint32_t* __restrict__ on,
                                               uint64_t n_on, uint32_t* __restrict__ cv_on, uint32_t blocks_pre, const uint32_t* __restrict__ quads, uint32_t n_quads) const {
    const uint64_t c_pre = n_pre == 0 ? 1 : (n_pre + 1023) / 1024, c_on = n_on == 0 ? 1 : (n_on + 1023) / 1024;
    if (blockIdx.x < blocks_pre)
        B_k_b3_chunks_bits<4, true>::run((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, pre, n_pre, 64, c_pre, cv_pre, 0, 1);
    else if (QUADS == 2)
        B_k_b3_chunks<1, false>::run((uint64_t)(blockIdx.x - blocks_pre) * blockDim.x + threadIdx.x, on, n_on, 64, c_on, cv_on, quads, n_quads, 0, 1);
    else if (QUADS)
        B_k_b3_chunks<4, false>::run((uint64_t)(blockIdx.x - blocks_pre) * blockDim.x + threadIdx.x, on, n_on, 64, c_on, cv_on, quads, n_quads, 0, 1);
    else
        B_k_b3_chunks<4, true>::run((uint64_t)(blockIdx.x - blocks_pre) * blockDim.x + threadIdx.x, on, n_on, 64, c_on, cv_on, nullptr, 0, 0, 1);
    }
};
template <int QUADS>
__global__ __launch_bounds__(256) void k_b3_chunks_pair_uni(const uint8_t* __restrict__ pre, uint64_t n_pre, uint32_t* __restrict__ cv_pre, const uint32_t* __restrict__ on,
                                                            uint64_t n_on, uint32_t* __restrict__ cv_on, uint32_t blocks_pre, const uint32_t* __restrict__ quads,
                                                            uint32_t n_quads) {
    B_k_b3_chunks_pair_uni<QUADS>{}(pre, n_pre, cv_pre, on, n_on, cv_on, blocks_pre, quads, n_quads);
}
// (B3_TAIL_PAIR: the shared tree top takes up to 1 024 nodes per repetition with 512 threads, and a shared reduction launch folds THREE
// levels -- 4 900 chunks are 613 nodes after one launch, where two levels per launch and a 512-node top needed two launches)
constexpr uint32_t B3_TAIL_PAIR = 1024;
struct B_k_b3_reduce_pair {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ in_a, uint64_t n_a, uint32_t* __restrict__ out_a, const uint32_t* __restrict__ in_b,
                                               uint64_t n_b, uint32_t* __restrict__ out_b, uint32_t R) const {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {
        if (n_a > B3_TAIL_PAIR) B_k_b3_reduce<3>::run(tid, in_a, n_a, R, out_a);
    } else {
        if (n_b > B3_TAIL_PAIR) B_k_b3_reduce<3>::run(tid, in_b, n_b, R, out_b);
    }
    }
};
__global__ __launch_bounds__(256) void k_b3_reduce_pair(const uint32_t* __restrict__ in_a, uint64_t n_a, uint32_t* __restrict__ out_a, const uint32_t* __restrict__ in_b,
                                                        uint64_t n_b, uint32_t* __restrict__ out_b, uint32_t R) {
    B_k_b3_reduce_pair{}(in_a, n_a, out_a, in_b, n_b, out_b, R);
}
struct B_k_b3_tree_tail_pair_big {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ in_a, uint32_t n_a, uint32_t* __restrict__ dig_a, const uint32_t* __restrict__ in_b, uint32_t n_b,
                                               uint32_t* __restrict__ dig_b, uint32_t R) const {
    if (blockIdx.x < R)
        B_k_b3_tree_tail<(int)B3_TAIL_PAIR>::run(blockIdx.x, in_a, n_a, R, dig_a);
    else
        B_k_b3_tree_tail<(int)B3_TAIL_PAIR>::run(blockIdx.x - R, in_b, n_b, R, dig_b);
    }
};
__global__ __launch_bounds__(512) void k_b3_tree_tail_pair_big(const uint32_t* __restrict__ in_a, uint32_t n_a, uint32_t* __restrict__ dig_a, const uint32_t* __restrict__ in_b,
                                                               uint32_t n_b, uint32_t* __restrict__ dig_b, uint32_t R) {
    B_k_b3_tree_tail_pair_big{}(in_a, n_a, dig_a, in_b, n_b, dig_b, R);
}
static uint64_t b3_rpl1_lanes();
// does launch_b3_pair_big take these two transcripts?  (both trees must end in the 256-thread tree top: more than 64 nodes left)
bool b3_pair_big_ok(uint64_t n_pre, uint64_t n_on, uint32_t NQ, const uint32_t* d_quads, uint32_t n_quads) {
    if (NQ != 64 || g_recorder || RV_B3_RPL != 4) return false;
    if (d_quads && !n_quads) return false;
    for (int i = 0; i < 2; i++) {
        const uint64_t n_ev = i == 0 ? n_pre : n_on;
        const bool listed = i == 1 && d_quads;  // (the online stream of the listed quad words only)
        uint64_t n = n_ev == 0 ? 1 : (n_ev + 1023) / 1024;
        // (short transcripts: the separate launchers pick other chunk kernels; few listed quad words are hashed a repetition per lane anyway)
        if (!(listed && n_quads * 4 <= NQ) && n * (listed ? std::min(n_quads, NQ) : NQ) < b3_rpl1_lanes()) return false;
        while (n > B3_TAIL_PAIR) n = (n + 7) / 8;
        if (n <= 64) return false;
    }
    return true;
}
// cv_a0 / cv_a1 and cv_b0 / cv_b1: ping-pong buffers of the preprocessing and the online stream (b3_stream_scratch_words each);
// -> launches
uint32_t launch_b3_pair_big(hipStream_t st, const uint8_t* d_pre, uint64_t n_pre, const uint32_t* d_on, uint64_t n_on, uint32_t NQ, uint32_t* cv_a0,
                            uint32_t* cv_a1, uint32_t* cv_b0, uint32_t* cv_b1, uint32_t* d_dig_pre, uint32_t* d_dig_on, const uint32_t* d_quads,
                            uint32_t n_quads) {
    const uint32_t R = NQ * 4;
    uint64_t n_a = n_pre == 0 ? 1 : (n_pre + 1023) / 1024, n_b = n_on == 0 ? 1 : (n_on + 1023) / 1024;
    uint32_t launches = 1;
    const uint32_t b_pre = (uint32_t)((n_a * 64 + 255) / 256);  // (a chunk per wavefront)
    if (d_quads && n_quads * 4 <= NQ) {
        // (the chaining values of skipped quad words stay whatever the buffer held: the tree above them runs on garbage and the caller
        // replaces those digests, as with launch_b3_stream)
        const uint32_t b_on = (uint32_t)((n_b * n_quads * 4 + 255) / 256);
        launch<B_k_b3_chunks_pair_uni<2>, 256>(k_b3_chunks_pair_uni<2>, st, dim3(b_pre + b_on), dim3(256), d_pre, n_pre, cv_a0, d_on, n_on, cv_b0, b_pre, d_quads,
                                              n_quads);
    } else if (d_quads) {
        const uint32_t b_on = (uint32_t)((n_b * n_quads + 255) / 256);
        launch<B_k_b3_chunks_pair_uni<1>, 256>(k_b3_chunks_pair_uni<1>, st, dim3(b_pre + b_on), dim3(256), d_pre, n_pre, cv_a0, d_on, n_on, cv_b0, b_pre, d_quads,
                                              n_quads);
    } else {
        const uint32_t b_on = (uint32_t)((n_b * 64 + 255) / 256);
        launch<B_k_b3_chunks_pair_uni<0>, 256>(k_b3_chunks_pair_uni<0>, st, dim3(b_pre + b_on), dim3(256), d_pre, n_pre, cv_a0, d_on, n_on, cv_b0, b_pre,
                                              (const uint32_t*)nullptr, 0u);
    }
    while (n_a > B3_TAIL_PAIR || n_b > B3_TAIL_PAIR) {
        const uint64_t out_a = (n_a + 7) / 8, out_b = (n_b + 7) / 8;
        const uint64_t threads = std::max(n_a > B3_TAIL_PAIR ? out_a : 0, n_b > B3_TAIL_PAIR ? out_b : 0) * R;
        launch<B_k_b3_reduce_pair, 256>(k_b3_reduce_pair, st, dim3((unsigned)((threads + 255) / 256), 2), dim3(256), (const uint32_t*)cv_a0, n_a, cv_a1,
                                        (const uint32_t*)cv_b0, n_b, cv_b1, R);
        if (n_a > B3_TAIL_PAIR) std::swap(cv_a0, cv_a1), n_a = out_a;
        if (n_b > B3_TAIL_PAIR) std::swap(cv_b0, cv_b1), n_b = out_b;
        launches++;
    }
    launch<B_k_b3_tree_tail_pair_big, 512>(k_b3_tree_tail_pair_big, st, dim3(2 * R), dim3(512), (const uint32_t*)cv_a0, (uint32_t)n_a, d_dig_pre,
                                           (const uint32_t*)cv_b0, (uint32_t)n_b, d_dig_on, R);
    return launches + 1;
}

// tree reduction of n chunk chaining values per repetition; the roots land in d_digest ([R][8] words)
uint32_t b3_reduce_tree(hipStream_t st, uint32_t* cur, uint32_t* nxt, uint64_t n, uint32_t R, uint32_t* d_digest) {
    uint32_t launches = 1;
    while (n > B3_TAIL) {  // two levels per launch while the level is wide
        const uint64_t n_out = (n + 3) / 4;
        const uint64_t threads = n_out * R;
        launch<B_k_b3_reduce<2>, 256>(k_b3_reduce<2>, st, dim3((unsigned)((threads + 255) / 256)), dim3(256), cur, n, R, nxt);
        uint32_t* t = cur;
        cur = nxt;
        nxt = t;
        n = n_out;
        launches++;
    }
    // a single chunk is already its own root (the chunk kernels applied the ROOT flag): cnt == 1 just copies
    // lane per repetition: always for a batch of proofs (gridDim.y supplies the parallelism), for a single proof only while
    // its n - 1 dependent compressions (~1.2 us each) beat the workgroup version's log2(n) levels with their barriers
    if (n <= 64 && ((g_recorder && g_recorder->batch >= 8) || n <= 4))
        launch<B_k_b3_tree_lane, 64>(k_b3_tree_lane, st, dim3((R + 63) / 64), dim3(64), cur, (uint32_t)n, R, d_digest);
    else if (n <= 64)
        launch<B_k_b3_tree_tail<64>, 64>(k_b3_tree_tail_small, st, dim3(R), dim3(64), cur, (uint32_t)n, R, d_digest);
    else
        launch<B_k_b3_tree_tail<(int)B3_TAIL>, 256>(k_b3_tree_tail, st, dim3(R), dim3(256), cur, (uint32_t)n, R, d_digest);
    return launches;
}

size_t b3_stream_scratch_words(uint64_t n_events, uint32_t R) {
    const uint64_t n_chunks = n_events == 0 ? 1 : (n_events + 1023) / 1024;
    return (size_t)n_chunks * R * 8;  // per ping-pong buffer
}

// (chunk, quad word) lanes below which a lane takes ONE repetition instead of four: four times the wavefronts, each a quarter as
// long -- for transcripts that would not fill the chip's wavefront slots otherwise
static uint64_t b3_rpl1_lanes() { return (uint64_t)128 * 1024; }  // (64-repetition shards of the 10^7-gate circuit: digests 0.47 -> 0.38 ms)

// chunk chaining values only ([n_chunks][R][8] into d_cv); chunk_base / root_ok: see B_k_b3_chunks
void launch_b3_stream_chunks(hipStream_t st, const uint32_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv, const uint32_t* d_quads,
                             uint32_t n_quads, uint64_t chunk_base, uint32_t root_ok) {
    const uint64_t n = n_events == 0 ? 1 : (n_events + 1023) / 1024;
    // (the chaining values of skipped quads stay whatever the scratch buffer held: the tree above them runs on
    // garbage and the caller replaces those digests)
    const uint64_t threads = n * (d_quads ? n_quads : NQ);
    // few lanes (a quarter of the row or less in the verifier; a transcript of a few chunks, i.e. a small circuit):
    // one repetition per lane gives four times the wavefronts, each a quarter as long
    if ((d_quads && n_quads * 4 <= NQ) || threads * (g_recorder ? g_recorder->batch : 1u) < b3_rpl1_lanes())
        launch<B_k_b3_chunks<1>, 256>(k_b3_chunks<1>, st, dim3((unsigned)((threads * 4 + 255) / 256)), dim3(256), d_stream, n_events, NQ, n, d_cv, d_quads, n_quads, chunk_base, root_ok);
    else if (RV_B3_RPL == 4 && NQ == 64 && !d_quads)
        launch<B_k_b3_chunks<4, true>, 256>(k_b3_chunks_uni, st, dim3((unsigned)((threads + 255) / 256)), dim3(256), d_stream, n_events, NQ, n, d_cv, d_quads, n_quads, chunk_base, root_ok);
    else
        launch<B_k_b3_chunks<RV_B3_RPL>, 256>(k_b3_chunks<RV_B3_RPL>, st, dim3((unsigned)((threads * (4 / RV_B3_RPL) + 255) / 256)), dim3(256), d_stream, n_events, NQ, n, d_cv, d_quads, n_quads, chunk_base, root_ok);
}

uint32_t launch_b3_stream(hipStream_t st, const uint32_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv_a,
                      uint32_t* d_cv_b, uint32_t* d_digest, const uint32_t* d_quads, uint32_t n_quads) {
    const uint32_t R = NQ * 4;
    uint64_t n = n_events == 0 ? 1 : (n_events + 1023) / 1024;
    if (d_quads && !n_quads) return 0;  // a verifier shard without opened repetitions: every online digest comes from the proof
    launch_b3_stream_chunks(st, d_stream, n_events, NQ, d_cv_a, d_quads, n_quads, 0, 1);
    return 1 + b3_reduce_tree(st, d_cv_a, d_cv_b, n, R, d_digest);  // launches
}

void launch_b3_stream_bits_chunks(hipStream_t st, const uint8_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv, uint64_t chunk_base,
                                  uint32_t root_ok) {
    const uint64_t n = n_events == 0 ? 1 : (n_events + 1023) / 1024;
    const uint64_t threads = n * NQ;
    // a transcript of a few chunks (small circuit, and no batch to supply the wavefronts): one repetition per lane
    if (threads * (g_recorder ? g_recorder->batch : 1u) < b3_rpl1_lanes())
        launch<B_k_b3_chunks_bits<1>, 256>(k_b3_chunks_bits1, st, dim3((unsigned)((threads * 4 + 255) / 256)), dim3(256), d_stream, n_events, NQ, n,
                                           d_cv, chunk_base, root_ok);
    else if (NQ == 64)
        launch<B_k_b3_chunks_bits<4, true>, 256>(k_b3_chunks_bits_uni, st, dim3((unsigned)((threads + 255) / 256)), dim3(256), d_stream, n_events, NQ, n,
                                                 d_cv, chunk_base, root_ok);
    else
        launch<B_k_b3_chunks_bits<4>, 256>(k_b3_chunks_bits, st, dim3((unsigned)((threads + 255) / 256)), dim3(256), d_stream, n_events, NQ, n,
                                           d_cv, chunk_base, root_ok);
}

uint32_t launch_b3_stream_bits(hipStream_t st, const uint8_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv_a,
                           uint32_t* d_cv_b, uint32_t* d_digest) {
    const uint32_t R = NQ * 4;
    const uint64_t n = n_events == 0 ? 1 : (n_events + 1023) / 1024;
    launch_b3_stream_bits_chunks(st, d_stream, n_events, NQ, d_cv_a, 0, 1);
    return 1 + b3_reduce_tree(st, d_cv_a, d_cv_b, n, R, d_digest);  // launches
}

// ---- incremental BLAKE3 tree (streaming prover): the chunk chaining values of a stream arrive in batches ----
// one tree level over a batch: seq = [pending?] ++ in[0 .. n_in); out[i] = parent(seq[2i], seq[2i+1]) for i < n_pairs
// (never ROOT: whether a merge is the root is only known when the stream ends, see k_b3_fold)
__global__ __launch_bounds__(256) void k_b3_pairs(const uint32_t* __restrict__ pending /* [R][8] or null */, const uint32_t* __restrict__ in,
                                                  uint64_t n_pairs, uint32_t R, uint32_t* __restrict__ out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = tid / R;
    const uint32_t r = (uint32_t)(tid % R);
    if (i >= n_pairs) return;
    const uint64_t shift = pending ? 1 : 0;
    const uint32_t* lp = (pending && i == 0) ? pending + (size_t)r * 8 : in + ((size_t)(2 * i - shift) * R + r) * 8;
    const uint32_t* rp = in + ((size_t)(2 * i + 1 - shift) * R + r) * 8;
    uint32_t l[8], rr[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        l[k] = lp[k];
        rr[k] = rp[k];
    }
    b3::parent(l, rr, 0, o);
    uint32_t* d = out + ((size_t)i * R + r) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = o[k];
}
void launch_b3_pairs(hipStream_t st, const uint32_t* d_pending, const uint32_t* d_in, uint64_t n_pairs, uint32_t R, uint32_t* d_out) {
    if (!n_pairs) return;
    const uint64_t threads = n_pairs * R;
    hipLaunchKernelGGL(k_b3_pairs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_pending, d_in, n_pairs, R, d_out);
}
// end of a stream: the last chunk's chaining value folded into the pending subtree roots, smallest first; the last
// merge is the root (a lone last chunk was hashed with ROOT already and n = 0 just copies it)
__global__ void k_b3_fold(B3FoldList L, const uint32_t* __restrict__ last, uint32_t R, uint32_t* __restrict__ digest) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    uint32_t cv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cv[k] = last[(size_t)r * 8 + k];
    for (uint32_t i = 0; i < L.n; i++) {
        uint32_t l[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) l[k] = L.p[i][(size_t)r * 8 + k];
        b3::parent(l, cv, i + 1 == L.n ? b3::ROOT : 0u, o);
#pragma unroll
        for (int k = 0; k < 8; k++) cv[k] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) digest[(size_t)r * 8 + k] = cv[k];
}
void launch_b3_fold(hipStream_t st, const B3FoldList& L, const uint32_t* d_last, uint32_t R, uint32_t* d_digest) {
    hipLaunchKernelGGL(k_b3_fold, dim3((R + 63) / 64), dim3(64), 0, st, L, d_last, R, d_digest);
}

// Transcript::hash + CombineInstance::hash: h = B3(B3(pre2||on2) || B3(pre64||on64))
struct B_k_join {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ pre2, const uint32_t* __restrict__ on2, const uint32_t* __restrict__ pre64, const uint32_t* __restrict__ on64, uint32_t R, uint8_t* __restrict__ h) const {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    uint32_t m[16], h2[8], h64[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m[k] = pre2[r * 8 + k];
        m[8 + k] = on2[r * 8 + k];
    }
    b3::hash64(m, h2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m[k] = pre64[r * 8 + k];
        m[8 + k] = on64[r * 8 + k];
    }
    b3::hash64(m, h64);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m[k] = h2[k];
        m[8 + k] = h64[k];
    }
    b3::hash64(m, o);
    uint32_t* d = (uint32_t*)(h + 32 * (size_t)r);
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = o[k];
}
};
__global__ void k_join(const uint32_t* __restrict__ pre2, const uint32_t* __restrict__ on2, const uint32_t* __restrict__ pre64, const uint32_t* __restrict__ on64, uint32_t R, uint8_t* __restrict__ h) {
    B_k_join{}(pre2, on2, pre64, on64, R, h);
}

// verifier set-up: rows of `src` replace those of `dst` for the repetitions with (omit[r] < 8) == want_online
// (opened player keys and carried-over online commitments arrive in ONE staging copy instead of one tiny
// host-to-device copy per repetition)
struct B_k_overlay_rows {
    __device__ __forceinline__ void operator()(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const uint8_t* __restrict__ omit, uint32_t R, uint32_t row_words, int want_online) const {
    const uint32_t r = blockIdx.x;
    if (r >= R || (int)(omit[r] < RV_PLAYERS) != want_online) return;
    for (uint32_t t = threadIdx.x; t < row_words; t += blockDim.x) dst[(size_t)r * row_words + t] = src[(size_t)r * row_words + t];
}
};
__global__ void k_overlay_rows(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const uint8_t* __restrict__ omit, uint32_t R, uint32_t row_words, int want_online) {
    B_k_overlay_rows{}(dst, src, omit, R, row_words, want_online);
}

// parity hook for DomainGF2::reconstruct (gf2/domain.rs:47-63): the reference's packed u64 share is two quad words (hi, lo)
__global__ void k_hook_recon_gf2(const uint64_t* __restrict__ shares, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = shares[i];
    out[i] = ((uint64_t)recon32((uint32_t)(v >> 32)) << 32) | recon32((uint32_t)v);
}
void launch_hook_recon_gf2(hipStream_t st, const uint64_t* d_shares, uint64_t n, uint64_t* d_out) {
    if (n) hipLaunchKernelGGL(k_hook_recon_gf2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_shares, n, d_out);
}

void launch_overlay_rows(hipStream_t st, uint32_t* d_dst, const uint32_t* d_src, const uint8_t* d_omit, uint32_t R,
                         uint32_t row_words, int want_online) {
    launch<B_k_overlay_rows, 32>(k_overlay_rows, st, dim3(R), dim3(32), d_dst, d_src, d_omit, R, row_words, want_online);
}

// n_rows copies of one 32-byte digest (the Z64 transcripts of a pure GF(2) circuit are empty: BLAKE3(""))
struct Digest8 {
    uint32_t w[8];
};
struct B_k_fill_digests {
    __device__ __forceinline__ void operator()(uint32_t* __restrict__ dst, uint32_t n_rows, Digest8 d) const {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows * 8) dst[i] = d.w[i & 7];
}
};
__global__ void k_fill_digests(uint32_t* __restrict__ dst, uint32_t n_rows, Digest8 d) {
    B_k_fill_digests{}(dst, n_rows, d);
}
void launch_fill_digests(hipStream_t st, uint32_t* d_dst, uint32_t n_rows, const uint32_t digest[8]) {
    Digest8 d;
    for (int k = 0; k < 8; k++) d.w[k] = digest[k];
    launch<B_k_fill_digests, 256>(k_fill_digests, st, dim3((n_rows * 8 + 255) / 256), dim3(256), d_dst, n_rows, d);
}

// start of the interpreter phase: clear the invalid-witness flag and the all-zero row (mask words, corr-bit words)
// (fill / n_fill_rows / d: also n_fill_rows copies of the digest d -- the Z64 transcripts' digests of a pure GF(2) circuit, BLAKE3(""),
// which otherwise cost a launch of their own between the hashes and the commitment)
struct B_k_shard_init {
    __device__ __forceinline__ void operator()(int* __restrict__ err, uint32_t* __restrict__ zero_mask, uint32_t n_mask_words, uint8_t* __restrict__ zero_corr, uint32_t n_corr_bytes,
                                               uint32_t* __restrict__ fill, uint32_t n_fill_rows, Digest8 d, uint8_t* __restrict__ zero_byte) const {
    const uint32_t i = threadIdx.x;
    if (i == 0) *err = 0;
    if (i == 1 && zero_byte) *zero_byte = 0;  // (MODE_PROVE_V: the zero row's cleartext value -- a memset launch of its own before)
    if (i < n_mask_words) zero_mask[i] = 0;
    if (i < n_corr_bytes) zero_corr[i] = 0;
    if (fill)
        for (uint32_t j = i; j < n_fill_rows * 8; j += blockDim.x) fill[j] = d.w[j & 7];
}
};
__global__ void k_shard_init(int* __restrict__ err, uint32_t* __restrict__ zero_mask, uint32_t n_mask_words, uint8_t* __restrict__ zero_corr, uint32_t n_corr_bytes,
                             uint32_t* __restrict__ fill, uint32_t n_fill_rows, Digest8 d, uint8_t* __restrict__ zero_byte) {
    B_k_shard_init{}(err, zero_mask, n_mask_words, zero_corr, n_corr_bytes, fill, n_fill_rows, d, zero_byte);
}
void launch_shard_init(hipStream_t st, int* d_err, uint32_t* d_zero_mask, uint32_t n_mask_words, uint8_t* d_zero_corr,
                       uint32_t n_corr_bytes, uint32_t* d_fill, uint32_t n_fill_rows, const uint32_t* digest, uint8_t* d_zero_byte) {
    Digest8 d{};
    if (d_fill)
        for (int k = 0; k < 8; k++) d.w[k] = digest[k];
    launch<B_k_shard_init, 64>(k_shard_init, st, dim3(1), dim3(64), d_err, d_zero_mask, n_mask_words, d_zero_corr, n_corr_bytes, d_fill, n_fill_rows, d, d_zero_byte);
}

void launch_join(hipStream_t st, const uint32_t* d_pre2, const uint32_t* d_on2, const uint32_t* d_pre64, const uint32_t* d_on64,
                 uint32_t R, uint8_t* d_h) {
    launch<B_k_join, 64>(k_join, st, dim3((R + 63) / 64), dim3(64), d_pre2, d_on2, d_pre64, d_on64, R, d_h);
}

// ------------------------------------------------------------------------------------
// Openings.  kind 0: omitted player's bit of a recorded broadcast share (PackSelected,
// gf2/share.rs:87-149); kind 1: a 0x00/0xFF recon byte (Pack, gf2/recon.rs:189-239).
// Items are packed 8 per byte MSB-first; the output vector has n_items/8 + 1 bytes (the
// reference always emits one more chunk).
//
// A workgroup produces EX_TB consecutive output bytes of EVERY opened repetition: the packed
// bytes are first collected in LDS ([slot][byte]) and then written out as contiguous runs.
// (Writing each byte straight from the lane that computed it cost 40 single-byte partial-line
// writes per 16 lines read: the kernel was bound by write transactions, not by HBM bytes.)
// ------------------------------------------------------------------------------------
constexpr uint32_t EX_TB = 128;
// (k_extract_rows: its own tile, for A/B builds)
#ifndef RV_EXR_TB
#define RV_EXR_TB 256
#endif
constexpr uint32_t EXR_TB = RV_EXR_TB;

// slot of every opened repetition (rank among the opened ones) and its output offset, into LDS
// stage_pitch != 0: the output is a dense staging block, slot k's bytes at k * stage_pitch (dst_off is not read)
__device__ __forceinline__ uint32_t ex_slots(const uint8_t* __restrict__ omit, const uint64_t* __restrict__ dst_off, uint32_t R,
                                             uint8_t* s_slot /*[256]*/, uint64_t* s_dst /*[RV_ONLINE_REPS]*/, uint32_t* s_cnt /*[5]*/, uint64_t stage_pitch = 0) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool on = tid < R && omit[tid] < 8;
    const unsigned long long bal = __ballot(on);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += s_cnt[w];
    const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    s_slot[tid] = (on && slot < RV_ONLINE_REPS) ? (uint8_t)slot : (uint8_t)0xFF;
    if (on && slot < RV_ONLINE_REPS) s_dst[slot] = stage_pitch ? (uint64_t)slot * stage_pitch : dst_off[tid];
    uint32_t n = 0;
    for (uint32_t w = 0; w < 4; w++) n += s_cnt[w];
    __syncthreads();
    return n < RV_ONLINE_REPS ? n : RV_ONLINE_REPS;
}

// contiguous write-out of the collected bytes: s_buf[slot][0 .. nb)
template <uint32_t TB = EX_TB>
__device__ __forceinline__ void ex_flush(const uint8_t* s_buf, const uint64_t* s_dst, uint32_t n_slots, uint64_t t0, uint32_t nb,
                                         uint8_t* __restrict__ out, uint32_t k0 = 0) {
    for (uint32_t idx = threadIdx.x + k0 * TB; idx < n_slots * TB; idx += blockDim.x) {
        const uint32_t k = idx / TB, i = idx % TB;
        if (i < nb && s_dst[k] != ~0ull) out[s_dst[k] + t0 + i] = s_buf[k * TB + i];  // (~0: a slot that is left out)
    }
}

// the same to page-locked HOST memory (internal.h: OpenDirect), in whole 16-byte aligned words: every word that STARTS inside the
// slot's run of nb bytes and ends inside the nbx >= nb bytes the workgroup has extracted (a run starts at an odd offset of the proof;
// byte stores cross the link as partial writes one by one, and with them the proof was SLOWER than without the direct path).  What is
// left -- the partial words at a vector's two ends -- k_copy_gaps copies from the image.  `out` is 16-byte aligned; TB = the stride
// of s_buf.
template <uint32_t TB>
__device__ __forceinline__ void ex_flush_host(const uint8_t* s_buf, const uint64_t* s_dst, uint32_t n_slots, uint64_t t0, uint32_t nb, uint32_t nbx,
                                              uint8_t* __restrict__ out) {
    constexpr uint32_t W = TB / 16 + 1;
    for (uint32_t idx = threadIdx.x; idx < n_slots * W; idx += blockDim.x) {
        const uint32_t k = idx / W, w = idx % W;
        if (s_dst[k] == ~0ull) continue;
        const uint64_t d0 = s_dst[k] + t0;
        const uint64_t wa = (d0 & ~15ull) + 16ull * w;
        if (wa >= d0 + nb) continue;
        const uint8_t* sp = s_buf + k * TB;
        if (wa >= d0 && wa + 16 <= d0 + nbx) {
            const uint32_t o = (uint32_t)(wa - d0);
            uint32_t x[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                x[q] = (uint32_t)sp[o + 4 * q] | ((uint32_t)sp[o + 4 * q + 1] << 8) | ((uint32_t)sp[o + 4 * q + 2] << 16) | ((uint32_t)sp[o + 4 * q + 3] << 24);
            *(uint4*)(out + wa) = make_uint4(x[0], x[1], x[2], x[3]);
        }
    }
}

template <int KIND>
struct B_k_extract_rows {
    // block0: added to blockIdx.x modulo 2^32 = this launch's workgroup 0 is workgroup block0 of the vectors (k_open_small runs the
    // extraction as one range of its grid and passes minus the range's first workgroup).  stage_pitch != 0: into a dense staging block
    // [slot][stage_pitch] instead of the proof image (unused since round 6's pruning; ex_slots keeps the form)
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ stream, const uint32_t* __restrict__ rows, uint64_t n_items, uint32_t NQ, uint32_t tb /* <= EXR_TB */, const uint8_t* __restrict__ omit /*[R]*/, const uint64_t* __restrict__ dst_off /*[R]*/, uint8_t* __restrict__ out, uint64_t stage_pitch, uint32_t block0, uint8_t* __restrict__ out2 = nullptr, uint32_t n_direct = 0) const {
    // (internal.h: OpenDirect) a workgroup that also writes to the proof buffer on the host sends every 16-byte aligned word that
    // STARTS in its tile, so it extracts up to LA bytes of the next tile as well: no word is left for two workgroups to share
    constexpr uint32_t LA = 16, SB = EXR_TB + LA;
    __shared__ uint8_t s_buf[RV_ONLINE_REPS * SB];
    __shared__ uint32_t s_rows[8 * SB];
    __shared__ uint8_t s_slot[256];
    __shared__ uint64_t s_dst[RV_ONLINE_REPS];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint8_t s_aq[64];
    __shared__ uint32_t s_naq;
    const uint64_t n_bytes = n_items / 8 + 1;
    const uint64_t t0 = (uint64_t)(uint32_t)(blockIdx.x + block0) * tb;  // (modulo 2^32: k_open_small passes minus its range's first block)
    const uint32_t nb = (uint32_t)((n_bytes - t0 < tb) ? n_bytes - t0 : tb);
    const bool direct = out2 && blockIdx.x + block0 < n_direct;
    const uint32_t nbx = direct ? (uint32_t)((n_bytes - t0 < nb + LA) ? n_bytes - t0 : nb + LA) : nb;  // bytes extracted
    // this workgroup's row ids, one coalesced pass (ordinals past the end repeat the last item; masked below)
    for (uint32_t i = threadIdx.x; i < 8 * nbx; i += 256) {
        uint64_t it = 8 * t0 + i;
        if (it >= n_items) it = n_items ? n_items - 1 : 0;
        s_rows[i] = rows ? rows[it] : (uint32_t)it;
    }
    const uint32_t n_slots = ex_slots(omit, dst_off, 4 * NQ, s_slot, s_dst, s_cnt, stage_pitch);  // contains the barrier for s_rows
    if (!n_slots) return;
    // the quad words that hold an opened repetition (about 30 of 64 for a whole proof), compacted: thread = (output byte,
    // such a quad), so no lane idles on a quad nobody opened (k_extract_rows<0> 263 -> 240 us on the 10^7-gate circuit)
    if (threadIdx.x < 64) {
        const uint32_t q = threadIdx.x;
        bool act = false;
        if (q < NQ) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                act |= s_slot[4 * q + i] != 0xFF;
            }
        }
        const unsigned long long bal = __ballot(act);
        if (act) s_aq[__popcll(bal & ((1ull << q) - 1ull))] = (uint8_t)q;
        if (q == 0) s_naq = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    const uint32_t n_aq = s_naq;
    if (!n_aq) return;
    const uint32_t dtl = 256 / n_aq, da = 256 % n_aq;
    for (uint32_t tl = threadIdx.x / n_aq, a = threadIdx.x % n_aq; tl < nbx;) {
        const uint32_t q = s_aq[a];
        uint32_t sl[4], om[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            sl[i] = s_slot[4 * q + i];
            om[i] = omit[4 * q + i];
        }
        const uint64_t it0 = 8 * (t0 + tl);
        uint32_t w[8];
        // (read once: nontemporal, 0.43 -> 0.41 ms for the opening phase)
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = n_items ? __builtin_nontemporal_load(&stream[(size_t)s_rows[8 * tl + j] * NQ + q]) : 0u;
        // rows past the end contribute zero bits (their loads were clamped to the last item)
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (it0 + j >= n_items) w[j] = 0;
        // only the opened repetitions of the quad (usually one of the four) are worth the bit gathering
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (sl[i] == 0xFF) continue;
            const uint32_t sh = (KIND == 0) ? (31u - 8u * i - (om[i] & 7u)) : (24u - 8u * i);
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) acc |= ((w[j] >> sh) & 1u) << (7 - j);
            s_buf[sl[i] * SB + tl] = (uint8_t)acc;
        }
        a += da;
        tl += dtl;
        if (a >= n_aq) {
            a -= n_aq;
            tl++;
        }
    }
    __syncthreads();
    ex_flush<SB>(s_buf, s_dst, n_slots, t0, nb, out);
    if (direct) ex_flush_host<SB>(s_buf, s_dst, n_slots, t0, nb, nbx, out2);
}
};
template <int KIND>
__global__ __launch_bounds__(256) void k_extract_rows(const uint32_t* __restrict__ stream, const uint32_t* __restrict__ rows, uint64_t n_items, uint32_t NQ, uint32_t tb /* <= EX_TB */, const uint8_t* __restrict__ omit /*[R]*/, const uint64_t* __restrict__ dst_off /*[R]*/, uint8_t* __restrict__ out, uint64_t stage_pitch, uint32_t block0, uint8_t* __restrict__ out2, uint32_t n_direct) {
    B_k_extract_rows<KIND>{}(stream, rows, n_items, NQ, tb, omit, dst_off, out, stage_pitch, block0, out2, n_direct);
}

// Bit-per-rep source (the preprocessing stream, [n][NQ/2] bytes; nibble bit k of quad q <-> repetition 4q+3-k):
// the workgroup's 8*tb rows are contiguous in HBM and are copied to LDS in one coalesced pass; thread =
// (output byte, opened repetition) then picks its 8 bits out of LDS.
struct B_k_extract_from_bits {
    // (block0: added to blockIdx.x modulo 2^32 -- k_open_small runs this as one range of its grid)
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ bits, uint64_t n_items, uint32_t NQ, uint32_t tb /* <= EX_TB */, const OnlineList* __restrict__ olp, uint8_t* __restrict__ out, uint32_t rep_min, uint32_t block0 = 0) const {
    __shared__ uint8_t s_buf[RV_ONLINE_REPS * EX_TB];
    __shared__ uint64_t s_dst[RV_ONLINE_REPS];
    __shared__ uint32_t s_pos[RV_ONLINE_REPS];  // byte in the row << 3 | bit in the byte
    __shared__ __attribute__((aligned(16))) uint8_t s_pre[8 * EX_TB * 32];
    const uint32_t n_ol = olp->n < RV_ONLINE_REPS ? olp->n : RV_ONLINE_REPS;
    if (!n_ol) return;
    const uint64_t n_bytes = n_items / 8 + 1;
    const uint64_t t0 = (uint64_t)(uint32_t)(blockIdx.x + block0) * tb;
    const uint32_t nb = (uint32_t)((n_bytes - t0 < tb) ? n_bytes - t0 : tb);
    const uint32_t h = NQ >> 1;  // bytes per row
    const uint64_t r0 = 8 * t0;
    const uint32_t n_rows = (uint32_t)(r0 >= n_items ? 0 : (n_items - r0 < 8ull * nb ? n_items - r0 : 8ull * nb));
    const uint8_t* src = bits + r0 * h;  // 8-byte aligned (r0 is a multiple of 8); 16-byte when h is even
    const uint32_t total = n_rows * h;
    if ((h & 1) == 0) {
        for (uint32_t i = threadIdx.x * 16; i + 16 <= total; i += 256 * 16) *(uint4*)(s_pre + i) = *(const uint4*)(src + i);
        for (uint32_t i = (total & ~15u) + threadIdx.x; i < total; i += 256) s_pre[i] = src[i];
    } else {
        for (uint32_t i = threadIdx.x; i < total; i += 256) s_pre[i] = src[i];
    }
    if (threadIdx.x < n_ol) {
        const uint32_t r = olp->rep[threadIdx.x];
        s_dst[threadIdx.x] = r < rep_min ? ~0ull : olp->dst[threadIdx.x];  // (early corrections: the host has the vectors of the repetitions below rep_min)
        s_pos[threadIdx.x] = ((r >> 3) << 3) | (4 * ((r >> 2) & 1) + 3 - (r & 3));
    }
    __syncthreads();
    // thread = (opened repetition k, output-byte lane): k, and with it the byte / bit it picks out of a row, stay in
    // registers for the whole loop (an index split per output byte cost 4x the instructions: 123 -> 45 us per proof)
    {
        const uint32_t lanes = 256 / n_ol;  // output bytes in flight per repetition
        const uint32_t k = threadIdx.x % n_ol, tlane = threadIdx.x / n_ol;
        if (tlane < lanes) {
            const uint32_t pos = s_pos[k], bit = pos & 7;
            const uint8_t* col = s_pre + (pos >> 3);
            const uint32_t full = n_rows / 8;  // output bytes whose eight rows all exist
            for (uint32_t tl = tlane; tl < nb; tl += lanes) {
                uint32_t acc = 0;
                if (tl < full) {
#pragma unroll
                    for (int j = 0; j < 8; j++) acc |= (((uint32_t)col[(8 * tl + j) * h] >> bit) & 1u) << (7 - j);
                } else {
                    for (uint32_t j = 0; j < 8; j++)
                        if (8 * tl + j < n_rows) acc |= (((uint32_t)col[(8 * tl + j) * h] >> bit) & 1u) << (7 - j);
                }
                s_buf[k * EX_TB + tl] = (uint8_t)acc;
            }
        }
    }
    __syncthreads();
    ex_flush(s_buf, s_dst, n_ol, t0, nb, out);
}
};
__global__ __launch_bounds__(256) void k_extract_from_bits(const uint8_t* __restrict__ bits, uint64_t n_items, uint32_t NQ, uint32_t tb /* <= EX_TB */, const OnlineList* __restrict__ olp, uint8_t* __restrict__ out, uint32_t rep_min) {
    B_k_extract_from_bits{}(bits, n_items, NQ, tb, olp, out, rep_min);
}

static uint32_t ex_tb_for(uint64_t n_bytes, uint32_t cap = EX_TB) {
    // output bytes per workgroup: the full EX_TB when that still yields several workgroups per CU, fewer for
    // short vectors (a workgroup walks its bytes in a serial loop)
    uint32_t tb = cap;
    while (tb > 8 && (n_bytes + tb - 1) / tb < 2048) tb /= 2;
    return tb;
}

void launch_extract_from_bits(hipStream_t st, const uint8_t* d_bits, uint64_t n_items, uint32_t NQ, const OnlineList* d_ol,
                              uint8_t* d_out, uint32_t rep_min) {
    const uint64_t n_bytes = n_items / 8 + 1;
    const uint32_t tb = ex_tb_for(n_bytes);
    launch<B_k_extract_from_bits, 256>(k_extract_from_bits, st, dim3((unsigned)((n_bytes + tb - 1) / tb)), dim3(256), d_bits, n_items, NQ, tb, d_ol,
                       d_out, rep_min);
}

// ------------------------------------------------------------------------------------
// Fiat-Shamir on the device (one wavefront).  combine_hashes (proof/mod.rs:102-108): comm =
// BLAKE3 of the 256 digests = 8 chunks (lanes 0..7, 16 chained compressions each) + a 3-level
// tree.  RandomOracle (crypto/ro.rs:8-20) + challenge_to_opening (proof/mod.rs:68-83): XOF of
// "random-oracle challenge" || 0x00 || comm; 16-byte draws, u128 LE mod 256 then mod 8 = the
// first byte of each draw; every lane produces one 64-byte XOF block = two (rep, omit) pairs,
// lane 0 replays them in order (a re-drawn repetition overwrites its omit) until 40 distinct.
// ------------------------------------------------------------------------------------
// The shard form (rep_begin, R): h holds ALL 256 digests (after the all-gather they are on every GPU), the challenge is
// derived for all repetitions, and the offsets / OnlineList / omit[0..R) are produced for the shard's own repetitions.
// How many of them are opened is only known here, so the section starts (online records, then preprocessing records,
// per domain) are computed on the device from L.base[0] (= start of the output, 40 past it when framed) and returned
// in res = {n_online_local, n_preprocessing_local}; omit_all (nullable) receives the full map for the host.
struct B_k_fs_challenge {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ h, FsLayout L, uint32_t rep_begin, uint32_t R, uint8_t* __restrict__ comm, uint8_t* __restrict__ omit, uint8_t* __restrict__ omit_all, uint64_t* __restrict__ offs, OnlineList* __restrict__ ol, uint32_t* __restrict__ res, uint32_t* __restrict__ mbox = nullptr, uint32_t* __restrict__ mbox_flag = nullptr, uint32_t mbox_seq = 0) const {
    __shared__ uint32_t s_cv[8][8], s_t1[4][8], s_t2[2][8], s_comm[8];
    __shared__ uint32_t s_msg[16];
    __shared__ uint8_t s_draw[128][2];
    __shared__ uint8_t s_omit[RV_TOTAL_REPS];
    __shared__ uint32_t s_count;
    const uint32_t lane = threadIdx.x;
    const uint32_t* hw = (const uint32_t*)h;
    // The commitment's 8 chunks x 16 chained blocks and its three tree levels are 19 compressions one after the other on the path
    // between the hashes and the openings of EVERY proof: a quad of lanes per compression (b3.h: compress_q, the digests staged in
    // LDS for the quads to share) instead of a lane -- 41 -> about 25 us for the kernel.
    __shared__ uint32_t s_h[RV_TOTAL_REPS * 8];
#pragma unroll
    for (uint32_t i = 0; i < RV_TOTAL_REPS * 8 / 64; i++) s_h[i * 64 + lane] = hw[i * 64 + lane];
    for (uint32_t r = lane; r < RV_TOTAL_REPS; r += 64) s_omit[r] = RV_PLAYERS;
    if (lane == 0) s_count = 0;
    __syncthreads();
    const uint32_t qc = lane & 3, qi = lane >> 2;  // column, quad
    const b3::QuadSchedule qs = b3::quad_schedule(qc);
    const uint32_t iv_a = qc == 0 ? B3_IV0 : qc == 1 ? B3_IV1 : qc == 2 ? B3_IV2 : B3_IV3;
    const uint32_t iv_b = qc == 0 ? B3_IV4 : qc == 1 ? B3_IV5 : qc == 2 ? B3_IV6 : B3_IV7;
    {
        // (all 16 quads run -- quad_from moves data between the lanes of a quad, every lane must be active --, the first 8 count)
        const uint32_t ch = qi & 7;
        uint32_t cva = iv_a, cvb = iv_b;
        for (uint32_t b = 0; b < 16; b++)
            b3::compress_q<false>(cva, cvb, s_h + ch * 256 + b * 16, qs, qc, ch, 64, (b == 0 ? b3::CHUNK_START : 0u) | (b == 15 ? b3::CHUNK_END : 0u));
        if (qi < 8) s_cv[qi][qc] = cva, s_cv[qi][4 + qc] = cvb;
    }
    __syncthreads();
    {
        uint32_t cva = iv_a, cvb = iv_b;
        b3::compress_q<false>(cva, cvb, &s_cv[2 * (qi & 3)][0], qs, qc, 0, 64, b3::PARENT);  // (s_cv[2p], s_cv[2p + 1]: 16 consecutive words)
        if (qi < 4) s_t1[qi][qc] = cva, s_t1[qi][4 + qc] = cvb;
    }
    __syncthreads();
    {
        uint32_t cva = iv_a, cvb = iv_b;
        b3::compress_q<false>(cva, cvb, &s_t1[2 * (qi & 1)][0], qs, qc, 0, 64, b3::PARENT);
        if (qi < 2) s_t2[qi][qc] = cva, s_t2[qi][4 + qc] = cvb;
    }
    __syncthreads();
    {
        uint32_t cva = iv_a, cvb = iv_b;
        b3::compress_q<false>(cva, cvb, &s_t2[0][0], qs, qc, 0, 64, b3::PARENT | b3::ROOT);
        if (qi == 0) s_comm[qc] = cva, s_comm[4 + qc] = cvb;
    }
    __syncthreads();
    if (lane == 0) {
        // the random oracle's one input block: context string, a zero byte, comm; 56 bytes, zero padded
        const char ctx[] = "random-oracle challenge";  // proof/mod.rs:18
        uint8_t blk[64];
        for (int i = 0; i < 64; i++) blk[i] = 0;
        for (int i = 0; i < 23; i++) blk[i] = (uint8_t)ctx[i];
        for (int i = 0; i < 8; i++) {
            comm[4 * i + 0] = blk[24 + 4 * i + 0] = (uint8_t)(s_comm[i]);
            comm[4 * i + 1] = blk[24 + 4 * i + 1] = (uint8_t)(s_comm[i] >> 8);
            comm[4 * i + 2] = blk[24 + 4 * i + 2] = (uint8_t)(s_comm[i] >> 16);
            comm[4 * i + 3] = blk[24 + 4 * i + 3] = (uint8_t)(s_comm[i] >> 24);
            if (L.comm2) {
                L.comm2[4 * i + 0] = (uint8_t)(s_comm[i]);
                L.comm2[4 * i + 1] = (uint8_t)(s_comm[i] >> 8);
                L.comm2[4 * i + 2] = (uint8_t)(s_comm[i] >> 16);
                L.comm2[4 * i + 3] = (uint8_t)(s_comm[i] >> 24);
            }
        }
        for (int i = 0; i < 16; i++)
            s_msg[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) | ((uint32_t)blk[4 * i + 2] << 16) |
                       ((uint32_t)blk[4 * i + 3] << 24);
    }
    __syncthreads();
    uint32_t m[16], cv[8];
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = s_msg[k];
    b3::iv(cv);
    for (uint64_t base = 0;; base += 64) {
        uint32_t o[16];
        b3::compress<true>(cv, m, base + lane, 56, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT, o);
        s_draw[2 * lane][0] = (uint8_t)o[0];
        s_draw[2 * lane][1] = (uint8_t)(o[4] & 7u);
        s_draw[2 * lane + 1][0] = (uint8_t)o[8];
        s_draw[2 * lane + 1][1] = (uint8_t)(o[12] & 7u);
        __syncthreads();
        if (lane == 0) {
            uint32_t count = s_count;
            for (uint32_t i = 0; i < 128 && count < RV_ONLINE_REPS; i++) {
                const uint32_t rep = s_draw[i][0];
                if (s_omit[rep] == RV_PLAYERS) count++;
                s_omit[rep] = s_draw[i][1];
            }
            s_count = count;
        }
        __syncthreads();
        if (s_count >= RV_ONLINE_REPS) break;
    }
    // offsets of every repetition's record and of its vectors (the same arithmetic as the host path)
    if (omit_all)
        for (uint32_t r = lane; r < RV_TOTAL_REPS; r += 64) omit_all[r] = s_omit[r];
    uint32_t n_on = 0;  // opened repetitions of this shard
    for (uint32_t c = 0; c < (R + 63) / 64; c++) {
        const uint32_t r = 64 * c + lane;
        n_on += (uint32_t)__popcll(__ballot(r < R && s_omit[rep_begin + r] < RV_PLAYERS));
    }
    const uint32_t n_pre = R - n_on;
    // sections: [gf2 online | gf2 preprocessing | z64 online | z64 preprocessing]; when the caller frames the
    // output as bincode(Proof) (single shard) L.base[] already holds the four starts, otherwise only base[0] counts
    uint64_t base[4];
    if (L.framed) {
#pragma unroll
        for (int k = 0; k < 4; k++) base[k] = L.base[k];
    } else {
        base[0] = L.base[0];
        base[1] = base[0] + (uint64_t)n_on * L.sz2;
        base[2] = base[1] + (uint64_t)n_pre * 48;
        base[3] = base[2] + (uint64_t)n_on * L.sz64;
    }
    uint32_t k_on = 0, k_pre = 0;
    for (uint32_t c = 0; c < (R + 63) / 64; c++) {
        const uint32_t r = 64 * c + lane;
        const bool valid = r < R;
        const uint32_t om = valid ? s_omit[rep_begin + r] : RV_PLAYERS;
        const bool on = valid && om < RV_PLAYERS;
        const unsigned long long bal = __ballot(on), val = __ballot(valid);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t my_on = k_on + (uint32_t)__popcll(bal & lt), my_pre = k_pre + (uint32_t)__popcll(~bal & val & lt);
        if (valid) {
            omit[r] = (uint8_t)om;
            uint64_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (on) {
                v[0] = base[0] + (uint64_t)my_on * L.sz2;
                v[1] = base[2] + (uint64_t)my_on * L.sz64;
                v[2] = v[0] + 137;
                v[3] = v[0] + 145 + L.l2r;
                v[4] = v[0] + 153 + L.l2r + L.l2c;
                v[5] = v[1] + 137;
                v[6] = v[1] + 145 + L.l64r;
                v[7] = v[1] + 153 + L.l64r + L.l64c;
                if (my_on < RV_ONLINE_REPS) {
                    ol->rep[my_on] = r;
                    ol->dst[my_on] = v[3];
                }
            } else {
                v[0] = base[1] + (uint64_t)my_pre * 48;
                v[1] = base[3] + (uint64_t)my_pre * 48;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) offs[(size_t)j * R + r] = v[j];
        }
        k_on += (uint32_t)__popcll(bal);
        k_pre += (uint32_t)__popcll(~bal & val);
    }
    if (lane == 0) {
        ol->n = n_on < RV_ONLINE_REPS ? n_on : RV_ONLINE_REPS;
        if (res) {
            res[0] = n_on;
            res[1] = n_pre;
        }
    }
    if (mbox) {
        // rv_prove's early path: what k_publish used to copy for the host in a launch of its own -- comm, the opening map, the two
        // counts (the bytes behind `comm` in device memory, in that order) -- into the host-mapped mailbox, then the stamp
        if (lane < 8) mbox[lane] = s_comm[lane];
        mbox[8 + lane] = (uint32_t)s_omit[4 * lane] | ((uint32_t)s_omit[4 * lane + 1] << 8) | ((uint32_t)s_omit[4 * lane + 2] << 16) | ((uint32_t)s_omit[4 * lane + 3] << 24);
        if (lane == 0) mbox[8 + RV_TOTAL_REPS / 4] = n_on, mbox[9 + RV_TOTAL_REPS / 4] = n_pre;
        __threadfence_system();
        __syncthreads();
        if (lane == 0) __hip_atomic_store(mbox_flag, mbox_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
};
__global__ __launch_bounds__(64) void k_fs_challenge(const uint8_t* __restrict__ h, FsLayout L, uint32_t rep_begin, uint32_t R, uint8_t* __restrict__ comm, uint8_t* __restrict__ omit, uint8_t* __restrict__ omit_all, uint64_t* __restrict__ offs, OnlineList* __restrict__ ol, uint32_t* __restrict__ res, uint32_t* __restrict__ mbox, uint32_t* __restrict__ mbox_flag, uint32_t mbox_seq) {
    B_k_fs_challenge{}(h, L, rep_begin, R, comm, omit, omit_all, offs, ol, res, mbox, mbox_flag, mbox_seq);
}

void launch_fs_challenge(hipStream_t st, const uint8_t* d_h, const FsLayout& L, uint32_t rep_begin, uint32_t R, uint8_t* d_comm,
                         uint8_t* d_omit, uint8_t* d_omit_all, uint64_t* d_offs, OnlineList* d_ol, uint32_t* d_res, uint32_t* mbox, uint32_t* mbox_flag,
                         uint32_t mbox_seq) {
    launch<B_k_fs_challenge, 64>(k_fs_challenge, st, dim3(1), dim3(64), d_h, L, rep_begin, R, d_comm, d_omit, d_omit_all, d_offs, d_ol, d_res, mbox, mbox_flag,
                                 mbox_seq);
}

uint32_t extract_tile_bytes(uint64_t n_items) { return ex_tb_for(n_items / 8 + 1, EXR_TB); }
void launch_extract_bits(hipStream_t st, const void* d_stream, const uint32_t* d_rows, uint64_t n_items, uint32_t NQ,
                         int kind, const uint8_t* d_omit, const uint64_t* d_dst_off, uint8_t* d_out, uint8_t* d_out2, uint32_t n_direct) {
    const uint64_t n_bytes = n_items / 8 + 1;
    const uint32_t tb = ex_tb_for(n_bytes, EXR_TB);
    const dim3 grid((unsigned)((n_bytes + tb - 1) / tb));
    if (!n_direct) d_out2 = nullptr;
    if (kind == 0)
        launch<B_k_extract_rows<0>, 256>(k_extract_rows<0>, st, grid, dim3(256), (const uint32_t*)d_stream, d_rows, n_items, NQ, tb, d_omit,
                           d_dst_off, d_out, 0, 0, d_out2, n_direct);
    else
        launch<B_k_extract_rows<1>, 256>(k_extract_rows<1>, st, grid, dim3(256), (const uint32_t*)d_stream, d_rows, n_items, NQ, tb, d_omit,
                           d_dst_off, d_out, 0, 0, (uint8_t*)nullptr, 0u);
}

// Inverse for the verifier (Pack::unpack / PackSelected::unpack_selected): builds dense
// rows from the proof's bit vectors.  kind 0: bit placed at the omitted player's position;
// kind 1: smeared 0x00/0xFF byte.  Reps that are not online-verified, and items beyond a
// vector's end, read as zero (verifier/online.rs:124,162,170 `unwrap_or_default`).
// A workgroup rebuilds 8*UNP_TB consecutive rows: the UNP_TB source bytes of every opened repetition are staged in
// LDS first (coalesced reads, one slot per opened repetition), then thread = (row, quad) assembles its word from LDS
// and the rows leave as full-width coalesced stores.  (One thread per word with four scattered byte loads from the
// proof took 2.0 ms per vector on the headline circuit; this takes 0.3: the 1.28 GB of rows written are the cost.)
constexpr uint32_t UNP_TB = 64;
struct B_k_unpack_bits {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ src_off, const uint64_t* __restrict__ src_len, const uint8_t* __restrict__ omit, uint64_t n_items, uint32_t NQ, int kind, uint32_t* __restrict__ rows_out, uint32_t out_nq, uint64_t first_item) const {
    // first_item: the vectors' item the output starts at (the streaming verifier rebuilds a chunk's rows: any bit offset);
    // a slot's staged bytes are UNP_TB + 1 so that the shifted window of the last items has its second byte.  The bytes
    // arrive as ALIGNED 32-bit loads (a slot's window starts at any byte of the proof: up to 3 bytes of slack in front,
    // `s_mis`), bytes outside the vector zeroed -- bytewise loads from 40 streams made the staging the longest part
    constexpr uint32_t SB = UNP_TB + 1, SW = (SB + 3 + 3) / 4, SBP = 4 * SW;  // words / padded bytes per slot
    __shared__ __attribute__((aligned(4))) uint8_t s_bytes[RV_ONLINE_REPS * SBP];
    __shared__ uint8_t s_slot[256];
    __shared__ uint8_t s_mis[RV_ONLINE_REPS];
    __shared__ uint64_t s_off[RV_ONLINE_REPS], s_len[RV_ONLINE_REPS];
    __shared__ uint32_t s_cnt[4];
    const uint32_t R = 4 * NQ;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // slot of every opened repetition of the shard (rank among the opened ones; at most RV_ONLINE_REPS)
    const bool on = tid < R && omit[tid] < 8;
    const unsigned long long bal = __ballot(on);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += s_cnt[w];
    const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    const bool have = on && slot < RV_ONLINE_REPS;
    s_slot[tid] = have ? (uint8_t)slot : (uint8_t)0xFF;
    if (have) {
        s_off[slot] = src_off[tid];
        s_len[slot] = src_len[tid];
    }
    uint32_t n_slots = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (n_slots > RV_ONLINE_REPS) n_slots = RV_ONLINE_REPS;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * UNP_TB;  // first output byte column of this workgroup
    const uint64_t b0 = first_item / 8 + t0;              // ... and the source byte it starts in
    const uint32_t sh = (uint32_t)(first_item & 7);       // output item il of the workgroup <-> source bit sh + il from byte b0
    for (uint32_t i = tid; i < n_slots * SW; i += 256) {
        const uint32_t k = i / SW, j = i % SW;
        const uint64_t len = s_len[k];
        const uintptr_t start = (uintptr_t)blob + s_off[k] + b0;  // first wanted byte; the vector ends at vend (past it: zero)
        const uintptr_t vend = (uintptr_t)blob + s_off[k] + (len < b0 + SB ? len : b0 + SB);
        const uintptr_t a = (start & ~(uintptr_t)3) + 4 * j;
        uint32_t v = 0;
        if (b0 < len && a + 4 > start && a < vend) {
            v = *(const uint32_t*)a;  // (inside the proof's allocation: it contains a byte of the vector, and the arena rounds to 256)
            const uint32_t lo = start > a ? (uint32_t)(start - a) : 0u, hi = vend < a + 4 ? (uint32_t)(vend - a) : 4u;
            const uint32_t m = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
            v &= m;
        }
        ((uint32_t*)s_bytes)[k * SW + j] = v;
        if (j == 0) s_mis[k] = (uint8_t)(start & 3);
    }
    __syncthreads();
    const uint64_t it0 = 8 * t0;
    const uint64_t n_here = (n_items - it0 < 8ull * UNP_TB) ? n_items - it0 : 8ull * UNP_TB;
    if (256 % NQ == 0) {
        // a thread keeps its quad for the whole loop: which of its four repetitions are opened, where their bytes start in
        // LDS and the word each contributes stay in registers; only the quads that hold an opened repetition are written at
        // all -- the interpreter reads no others -- and the threads are dealt over exactly those quads (in the verifier's
        // slot order: the first ten).  A step takes one source byte column: two LDS bytes per repetition give eight items.
        __shared__ uint8_t s_quads[64];
        __shared__ uint32_t s_nq;
        if (tid < 64) {
            const bool has = tid < NQ && (s_slot[4 * tid] & s_slot[4 * tid + 1] & s_slot[4 * tid + 2] & s_slot[4 * tid + 3]) != 0xFF;
            // ... rounded to whole 32-byte sectors (eight quads; the others get zeros): a row's ten quads are a full sector and
            // a quarter of the next, and partial-sector writes cost the memory side a read-modify-write each
            const unsigned long long bh = __ballot(has);
            const bool wr = tid < NQ && ((bh >> (tid & ~7u)) & 0xFFull) != 0;
            const unsigned long long bq = __ballot(wr);
            if (wr) s_quads[__popcll(bq & ((1ull << tid) - 1ull))] = (uint8_t)tid;
            if (tid == 0) s_nq = (uint32_t)__popcll(bq);
        }
        __syncthreads();
        const uint32_t nq = s_nq;
        if (!nq || tid >= nq * (256 / nq)) return;
        const uint32_t q = s_quads[tid % nq], step = 256 / nq;
        uint32_t at[4], val[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t sl = s_slot[4 * q + i];
            at[i] = sl == 0xFF ? 0xFFFFFFFFu : sl * SBP + s_mis[sl];
            val[i] = (kind == 0) ? (1u << (31u - 8u * i - (omit[4 * q + i] & 7u))) : (0xFFu << (24 - 8 * i));
        }
        for (uint32_t t = tid / nq; 8 * t < n_here; t += step) {
            uint32_t bits[4];  // item j of the column <-> bit 7 - j
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bits[i] = 0;
                if (at[i] != 0xFFFFFFFFu)
                    bits[i] = ((((uint32_t)s_bytes[at[i] + t] << 8) | (uint32_t)s_bytes[at[i] + t + 1]) >> (8 - sh)) & 0xFFu;
            }
            const uint32_t nj = n_here - 8 * t < 8 ? (uint32_t)(n_here - 8 * t) : 8u;
            uint32_t* dst = rows_out + (it0 + 8 * t) * out_nq + q;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                uint32_t w = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) w |= ((bits[i] >> (7 - j)) & 1u) ? val[i] : 0u;
                if (j < nj) dst[(size_t)j * out_nq] = w;
            }
        }
        return;
    }
    // (odd row widths: the plain loop)
    auto src_bit = [&](uint32_t sl, uint32_t il) {
        return ((uint32_t)s_bytes[sl * SBP + s_mis[sl] + ((sh + il) >> 3)] >> (7 - ((sh + il) & 7))) & 1u;
    };
    for (uint32_t idx = tid; idx < n_here * NQ; idx += 256) {
        const uint32_t il = idx / NQ, q = idx % NQ;
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t sl = s_slot[4 * q + i];
            if (sl != 0xFF) {
                const uint32_t bit = src_bit(sl, il);
                if (bit) w |= (kind == 0) ? (1u << (31u - 8u * i - omit[4 * q + i])) : (0xFFu << (24 - 8 * i));
            }
        }
        if (q < out_nq) rows_out[(it0 + il) * out_nq + q] = w;
    }
}
};
__global__ __launch_bounds__(256) void k_unpack_bits(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ src_off, const uint64_t* __restrict__ src_len, const uint8_t* __restrict__ omit, uint64_t n_items, uint32_t NQ, int kind, uint32_t* __restrict__ rows_out, uint32_t out_nq, uint64_t first_item) {
    B_k_unpack_bits{}(blob, src_off, src_len, omit, n_items, NQ, kind, rows_out, out_nq, first_item);
}

void launch_unpack_bits(hipStream_t st, const uint8_t* d_blob, const uint64_t* d_src_off, const uint64_t* d_src_len,
                        const uint8_t* d_omit, uint64_t n_items, uint32_t NQ, int kind, uint32_t* d_rows_out, uint32_t out_nq, uint64_t first_item) {
    if (!n_items) return;
    const uint64_t n_bytes = (n_items + 7) / 8;
    launch<B_k_unpack_bits, 256>(k_unpack_bits, st, dim3((unsigned)((n_bytes + UNP_TB - 1) / UNP_TB)), dim3(256), d_blob, d_src_off, d_src_len,
                                 d_omit, n_items, NQ, kind, d_rows_out, out_nq, first_item);
}

// Fixed-size parts of the openings.
//   online rep : omit | keys[8][16] with the omitted key zeroed | u64 len | .. | u64 len | .. | u64 len | ..
//   other rep  : seed[16] | H_on[32]                           (proof/mod.rs:41-53, prover.rs:125-136,167-170)
// d_off2/d_off64 give each rep's record offset inside the shard's concatenated output.
struct B_k_open_headers {
    // workgroup = (repetition, domain), thread = one header byte (consecutive lanes write consecutive bytes: the stores
    // coalesce, also when the proof buffer is mapped host memory); a lane-per-repetition version that walked its ~300
    // bytes one by one took 14 us of a 0.5 ms AES-128 proof
    __device__ __forceinline__ void operator()(uint32_t R, const uint8_t* __restrict__ omit, const uint8_t* __restrict__ seeds, const uint8_t* __restrict__ keys, const uint32_t* __restrict__ on2, const uint32_t* __restrict__ on64, const uint64_t* __restrict__ off2, const uint64_t* __restrict__ off64, uint64_t l2r, uint64_t l2c, uint64_t l2i, uint64_t l64r, uint64_t l64c, uint64_t l64i, uint8_t* __restrict__ out) const {
    run(blockIdx.x, R, omit, seeds, keys, on2, on64, off2, off64, l2r, l2c, l2i, l64r, l64c, l64i, out);
    }
    static __device__ __forceinline__ void run(uint32_t bx, uint32_t R, const uint8_t* __restrict__ omit, const uint8_t* __restrict__ seeds, const uint8_t* __restrict__ keys, const uint32_t* __restrict__ on2, const uint32_t* __restrict__ on64, const uint64_t* __restrict__ off2, const uint64_t* __restrict__ off64, uint64_t l2r, uint64_t l2c, uint64_t l2i, uint64_t l64r, uint64_t l64c, uint64_t l64i, uint8_t* __restrict__ out) {
    const uint32_t r = bx >> 1, dom = bx & 1u, i = threadIdx.x;
    if (r >= R) return;
    const uint32_t om = omit[r];
    uint8_t* o = out + (dom == 0 ? off2[r] : off64[r]);
    if (om < 8) {
        const uint64_t lr = dom == 0 ? l2r : l64r, lc = dom == 0 ? l2c : l64c, li = dom == 0 ? l2i : l64i;
        if (i == 0) {
            o[0] = (uint8_t)om;
        } else if (i < 129) {
            const uint32_t p = (i - 1) >> 4;
            o[i] = (p == om) ? (uint8_t)0 : keys[(size_t)r * 128 + (i - 1)];
        } else if (i < 137) {
            o[i] = (uint8_t)(lr >> (8 * (i - 129)));
        } else if (i < 145) {
            o[137 + lr + (i - 137)] = (uint8_t)(lc >> (8 * (i - 137)));
        } else if (i < 153) {
            o[145 + lr + lc + (i - 145)] = (uint8_t)(li >> (8 * (i - 145)));
        }
    } else {
        if (i < 16) {
            o[i] = seeds[(size_t)r * 16 + i];
        } else if (i < 48) {
            const uint32_t* hon = (dom == 0 ? on2 : on64) + (size_t)r * 8;
            o[i] = (uint8_t)(hon[(i - 16) >> 2] >> (8 * ((i - 16) & 3)));
        }
    }
}
};
__global__ void k_open_headers(uint32_t R, const uint8_t* __restrict__ omit, const uint8_t* __restrict__ seeds, const uint8_t* __restrict__ keys, const uint32_t* __restrict__ on2, const uint32_t* __restrict__ on64, const uint64_t* __restrict__ off2, const uint64_t* __restrict__ off64, uint64_t l2r, uint64_t l2c, uint64_t l2i, uint64_t l64r, uint64_t l64c, uint64_t l64i, uint8_t* __restrict__ out) {
    B_k_open_headers{}(R, omit, seeds, keys, on2, on64, off2, off64, l2r, l2c, l2i, l64r, l64c, l64i, out);
}

void launch_open_headers(hipStream_t st, uint32_t R, const uint8_t* d_omit, const uint8_t* d_seeds, const uint8_t* d_keys,
                         const uint32_t* d_on2, const uint32_t* d_on64, const uint64_t* d_off2, const uint64_t* d_off64,
                         uint64_t l2r, uint64_t l2c, uint64_t l2i, uint64_t l64r, uint64_t l64c, uint64_t l64i, uint8_t* d_out) {
    launch<B_k_open_headers, 192>(k_open_headers, st, dim3(2 * R), dim3(192), R, d_omit, d_seeds, d_keys, d_on2, d_on64, d_off2,
                       d_off64, l2r, l2c, l2i, l64r, l64c, l64i, d_out);
}

// ONE small GF(2) proof's openings in one launch: the record heads, the broadcast vectors, the corrections vectors and the input vectors
// are four independent pieces of work behind the challenge, each a kernel of 5 - 11 us that occupies a corner of the chip -- as ranges
// of one grid they cost one launch (and the error word for the host rides along).  Large proofs keep the separate launches: the
// pieces' LDS adds up here (58 KB per workgroup).
struct OpenSmall {
    // heads
    uint32_t R;
    const uint8_t *omit, *seeds, *keys;
    const uint32_t *on2, *on64;
    const uint64_t *off2, *off64;
    uint64_t l2r, l2c, l2i, l64r, l64c, l64i;
    // vectors
    const uint32_t *on, *rec_rows, *in_rows;
    const uint8_t* pre;
    uint64_t n_rec, n_pre, n_in;
    uint32_t NQ, tb_rec, tb_pre, tb_in;
    const uint64_t *dst_rec, *dst_in;
    const OnlineList* ol;
    uint32_t corr_rep_min;
    uint32_t g_hdr, g_rec, g_pre;  // workgroups of the first three ranges
    uint8_t* out;
    const int* err_src;
    int* err_dst;  // (nullable) host-mapped
};
__global__ __launch_bounds__(256) void k_open_small(OpenSmall a) {
    const uint32_t bx = blockIdx.x;
    if (bx < a.g_hdr) {
        if (threadIdx.x < 192) B_k_open_headers::run(bx, a.R, a.omit, a.seeds, a.keys, a.on2, a.on64, a.off2, a.off64, a.l2r, a.l2c, a.l2i, a.l64r, a.l64c, a.l64i, a.out);
        if (bx == 0 && threadIdx.x == 255 && a.err_dst) *a.err_dst = *a.err_src;
    } else if (bx < a.g_hdr + a.g_rec) {
        B_k_extract_rows<0>{}(a.on, a.rec_rows, a.n_rec, a.NQ, a.tb_rec, a.omit, a.dst_rec, a.out, 0, 0u - a.g_hdr);
    } else if (bx < a.g_hdr + a.g_rec + a.g_pre) {
        B_k_extract_from_bits{}(a.pre, a.n_pre, a.NQ, a.tb_pre, a.ol, a.out, a.corr_rep_min, 0u - (a.g_hdr + a.g_rec));
    } else {
        B_k_extract_rows<1>{}(a.on, a.in_rows, a.n_in, a.NQ, a.tb_in, a.omit, a.dst_in, a.out, 0, 0u - (a.g_hdr + a.g_rec + a.g_pre));
    }
}
// true (and the launch made) when the proof is small enough and nothing records launches; otherwise the caller launches the pieces
// heads_inputs_only: the record heads and the input vectors only (a large proof: its broadcast and corrections vectors keep their launches)
bool launch_open_small(hipStream_t st, uint32_t R, const uint8_t* d_omit, const uint8_t* d_seeds, const uint8_t* d_keys, const uint32_t* d_on2,
                       const uint32_t* d_on64, const uint64_t* d_offs /* [8][R] as shard_open_impl lays them out */, uint64_t l2r, uint64_t l2c, uint64_t l2i,
                       uint64_t l64r, uint64_t l64c, uint64_t l64i, const uint32_t* d_on, const uint32_t* d_rec_rows, uint64_t n_rec, const uint8_t* d_pre,
                       uint64_t n_pre, const uint32_t* d_in_rows, uint64_t n_in, uint32_t NQ, const OnlineList* d_ol, uint32_t corr_rep_min, uint8_t* d_out,
                       const int* d_err, int* err_dst_mapped, bool heads_inputs_only) {
    if (g_recorder) return false;
    OpenSmall a{};
    a.tb_rec = ex_tb_for(n_rec / 8 + 1, EXR_TB), a.tb_pre = ex_tb_for(n_pre / 8 + 1), a.tb_in = ex_tb_for(n_in / 8 + 1, EXR_TB);
    a.g_hdr = 2 * R;
    a.g_rec = heads_inputs_only ? 0u : (uint32_t)((n_rec / 8 + 1 + a.tb_rec - 1) / a.tb_rec);
    a.g_pre = !heads_inputs_only && corr_rep_min < R ? (uint32_t)((n_pre / 8 + 1 + a.tb_pre - 1) / a.tb_pre) : 0u;
    const uint32_t g_in = (uint32_t)((n_in / 8 + 1 + a.tb_in - 1) / a.tb_in);
    // (the sizes at which a launch matters: vectors of a few KB.  Longer ones keep their own launches -- 58 KB of LDS per workgroup here
    // against 20 there: the 10^7-gate circuit's openings took 470 us this way instead of 360)
    if ((uint64_t)a.g_rec + a.g_pre + g_in > 2048) return false;
    a.R = R, a.omit = d_omit, a.seeds = d_seeds, a.keys = d_keys, a.on2 = d_on2, a.on64 = d_on64, a.off2 = d_offs, a.off64 = d_offs + R;
    a.l2r = l2r, a.l2c = l2c, a.l2i = l2i, a.l64r = l64r, a.l64c = l64c, a.l64i = l64i;
    a.on = d_on, a.rec_rows = d_rec_rows, a.in_rows = d_in_rows, a.pre = d_pre, a.n_rec = n_rec, a.n_pre = n_pre, a.n_in = n_in, a.NQ = NQ;
    a.dst_rec = d_offs + 2 * (size_t)R, a.dst_in = d_offs + 4 * (size_t)R, a.ol = d_ol, a.corr_rep_min = corr_rep_min;
    a.out = d_out, a.err_src = d_err, a.err_dst = err_dst_mapped;
    hipLaunchKernelGGL(k_open_small, dim3(a.g_hdr + a.g_rec + a.g_pre + g_in), dim3(256), 0, st, a);
    return true;
}

// the device error word into a host-mapped word (small proofs leave without a copy engine: api.hip, rv_prove_impl)
__global__ void k_store_word(const int* __restrict__ src, int* __restrict__ dst) { *dst = *src; }
void launch_store_word(hipStream_t st, const int* d_src, int* dst_mapped) { hipLaunchKernelGGL(k_store_word, dim3(1), dim3(1), 0, st, d_src, dst_mapped); }
// a few KB (the verifier's 256 digests) plus the error word into host-mapped memory, for the same reason
__global__ void k_store_words(const uint32_t* __restrict__ src, uint32_t n_words, uint32_t* __restrict__ dst, const int* __restrict__ err, int* __restrict__ dst_err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
    if (i == 0 && err) *dst_err = *err;
}
void launch_store_words(hipStream_t st, const uint32_t* d_src, uint32_t n_words, uint32_t* dst_mapped, const int* d_err, int* dst_err_mapped) {
    hipLaunchKernelGGL(k_store_words, dim3((n_words + 255) / 256), dim3(256), 0, st, d_src, n_words, dst_mapped, d_err, dst_err_mapped);
}

// ------------------------------------------------------------------------------------
// Early corrections (api.hip, rv_prove on large GF(2) circuits).  The corrections vector of an opened repetition
// (Pack of ReconGF2, gf2/recon.rs:189-239: one bit per Mul, 8 per byte MSB-first, n/8 + 1 bytes) depends on the
// challenge only through WHICH repetitions open, and it is half of the proof.  So the packed vector of EVERY repetition
// is produced while the interpreter still runs -- a range of the preprocessing rows at a time, as soon as the levels that
// write them are done -- and leaves for the host through the copy engine before the challenge exists; after the
// challenge the host copies the 40 it needs into the proof and only the other half crosses PCIe behind the last kernel.
//
// k_pack_corr_all: workgroup = PC_TB output bytes (8 * PC_TB rows of 32 bytes, contiguous) of all 256 repetitions.
// A thread takes eight consecutive rows x four byte columns at a time: eight 32-bit loads, byte transposes (v_perm) into
// four words pairs with a row per byte, and for each column the 8 x 8 bit transpose of Hacker's Delight (transpose8rS32)
// -- eight output bytes, one per repetition of the column's byte, for ~5 instructions each (a lane per repetition
// picking one bit out of each of its eight rows took 14 and two quarter-rate multiplications).  The bytes are collected
// per repetition in LDS and written out as runs of two whole sectors (PC_TB = 64 bytes per workgroup and repetition; 128: 27 instead
// of 24.5 us per 27 MB chunk -- twice the workgroups in flight).
// out: [256][pitch], pitch a multiple of 128; byte0 = first byte of the chunk within a repetition's vector.
// ------------------------------------------------------------------------------------
constexpr uint32_t PC_TB = 64;
constexpr uint32_t PC_OSTRIDE = PC_TB + 4;  // bytes per repetition in the LDS output tile: 33 dwords
__global__ __launch_bounds__(256) void k_pack_corr_all(const uint8_t* __restrict__ bits /*[n_items][32]*/, uint64_t n_items, uint64_t byte0,
                                                       uint64_t n_bytes, uint64_t pitch, uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_out[256 * PC_OSTRIDE];
    const uint64_t t0 = (uint64_t)blockIdx.x * PC_TB;
    const uint32_t nb = (uint32_t)((n_bytes - t0 < PC_TB) ? n_bytes - t0 : PC_TB);
    const uint64_t r0 = 8 * (byte0 + t0);
    const uint32_t n_rows = (uint32_t)(r0 >= n_items ? 0 : (n_items - r0 < 8ull * nb ? n_items - r0 : 8ull * nb));  // rows past the end are zero bits
    const uint8_t* src = bits + r0 * 32;
    // item = (output byte tl, column quad cq): rows 8 tl .. 8 tl + 7, byte columns 4 cq .. 4 cq + 3; a wavefront's 64 items
    // cover 64 consecutive rows.  All of a thread's 32 loads are issued before the first transpose.
    constexpr int ITEMS = PC_TB * 8 / 256;
    uint32_t w[ITEMS][8];
#pragma unroll
    for (int n = 0; n < ITEMS; n++) {
        const uint32_t it = threadIdx.x + 256 * n, cq = it & 7, tl = it >> 3;
#pragma unroll
        for (int j = 0; j < 8; j++) w[n][j] = (8 * tl + j < n_rows) ? *(const uint32_t*)(src + (size_t)(8 * tl + j) * 32 + 4 * cq) : 0u;
    }
#pragma unroll
    for (int n = 0; n < ITEMS; n++) {
        const uint32_t it = threadIdx.x + 256 * n, cq = it & 7, tl = it >> 3;
        if (tl >= nb) continue;
        // xs[k] = rows 0..3 of column 4 cq + k, row 0 in the top byte; ys[k] = rows 4..7
        uint32_t xs[4], ys[4];
        {
            const uint32_t a = __builtin_amdgcn_perm(w[n][2], w[n][3], 0x05010400u), b = __builtin_amdgcn_perm(w[n][2], w[n][3], 0x07030602u);
            const uint32_t c = __builtin_amdgcn_perm(w[n][0], w[n][1], 0x05010400u), d = __builtin_amdgcn_perm(w[n][0], w[n][1], 0x07030602u);
            xs[0] = __builtin_amdgcn_perm(c, a, 0x05040100u), xs[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
            xs[2] = __builtin_amdgcn_perm(d, b, 0x05040100u), xs[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
        }
        {
            const uint32_t a = __builtin_amdgcn_perm(w[n][6], w[n][7], 0x05010400u), b = __builtin_amdgcn_perm(w[n][6], w[n][7], 0x07030602u);
            const uint32_t c = __builtin_amdgcn_perm(w[n][4], w[n][5], 0x05010400u), d = __builtin_amdgcn_perm(w[n][4], w[n][5], 0x07030602u);
            ys[0] = __builtin_amdgcn_perm(c, a, 0x05040100u), ys[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
            ys[2] = __builtin_amdgcn_perm(d, b, 0x05040100u), ys[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = xs[k], y = ys[k], t;
            t = (x ^ (x >> 7)) & 0x00AA00AAu, x = x ^ t ^ (t << 7);
            t = (y ^ (y >> 7)) & 0x00AA00AAu, y = y ^ t ^ (t << 7);
            t = (x ^ (x >> 14)) & 0x0000CCCCu, x = x ^ t ^ (t << 14);
            t = (y ^ (y >> 14)) & 0x0000CCCCu, y = y ^ t ^ (t << 14);
            t = (x & 0xF0F0F0F0u) | ((y >> 4) & 0x0F0F0F0Fu);
            y = ((x << 4) & 0xF0F0F0F0u) | (y & 0x0F0F0F0Fu);
            x = t;
            // byte i of (x, y) from the top = input bit 7 - i of every row, row j at output bit 7 - j.  Nibble bit p of a byte
            // <-> repetition 8 c + 3 - p (p < 4), 8 c + 11 - p (p >= 4), as k_extract_from_bits: x = repetitions 8 c + 4 .. + 7, y = 8 c .. + 3
            uint8_t* o = s_out + (size_t)(8 * (4 * cq + k)) * PC_OSTRIDE + tl;
            o[4 * PC_OSTRIDE] = (uint8_t)(x >> 24), o[5 * PC_OSTRIDE] = (uint8_t)(x >> 16), o[6 * PC_OSTRIDE] = (uint8_t)(x >> 8), o[7 * PC_OSTRIDE] = (uint8_t)x;
            o[0 * PC_OSTRIDE] = (uint8_t)(y >> 24), o[1 * PC_OSTRIDE] = (uint8_t)(y >> 16), o[2 * PC_OSTRIDE] = (uint8_t)(y >> 8), o[3 * PC_OSTRIDE] = (uint8_t)y;
        }
    }
    __syncthreads();
    // 16 bytes per thread and step, PC_TB / 16 threads per repetition: whole sectors
    for (uint32_t i = threadIdx.x; i < 256 * (PC_TB / 16); i += 256) {
        const uint32_t r = i / (PC_TB / 16), k = i % (PC_TB / 16);
        if (16 * k < nb) {  // (the bytes past nb inside the last 16 are pitch padding)
            const uint32_t* sp = (const uint32_t*)(s_out + r * PC_OSTRIDE + 16 * k);
            *(uint4*)(out + (size_t)r * pitch + t0 + 16 * k) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        }
    }
}
void launch_pack_corr_all(hipStream_t st, const uint8_t* d_bits, uint64_t n_items, uint64_t byte0, uint64_t n_bytes, uint64_t pitch, uint8_t* d_out) {
    if (!n_bytes) return;
    hipLaunchKernelGGL(k_pack_corr_all, dim3((unsigned)((n_bytes + PC_TB - 1) / PC_TB)), dim3(256), 0, st, d_bits, n_items, byte0, n_bytes, pitch, d_out);
}

// The proof image in HBM to the page-locked proof buffer WITHOUT the corrections vectors of its first m online records, m =
// the opened repetitions below rep_limit (omit[r] < 8; all n_rec of them when rep_limit = 256).  Record j: image bytes
// [first + j * rec, ...), its corrections at [+ corr_at, + corr_at + corr_len).  Piece j (blockIdx.y) runs from the end of record
// j - 1's corrections to the start of record j's, piece m to the end of the image.  Source and destination offsets are equal,
// both bases 16-byte aligned.
__global__ __launch_bounds__(256) void k_copy_gaps(const uint8_t* __restrict__ img, uint8_t* __restrict__ dst_mapped, uint64_t total, uint64_t first,
                                                   uint64_t rec, uint64_t corr_at, uint64_t corr_len, uint32_t n_rec, const uint8_t* __restrict__ omit,
                                                   uint32_t rep_limit, OpenDirect od, const int* __restrict__ err_src, int* __restrict__ err_dst) {
    __shared__ uint32_t s_m;
    if (err_dst && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 255) *err_dst = *err_src;  // (the error word for the host rides along)
    if (threadIdx.x < 64) {
        uint32_t cnt = 0;
        for (uint32_t r = threadIdx.x; r < rep_limit && r < RV_TOTAL_REPS; r += 64) cnt += omit[r] < 8 ? 1u : 0u;
        for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (threadIdx.x == 0) s_m = cnt < n_rec ? cnt : n_rec;
    }
    __syncthreads();
    const uint32_t m = s_m, j = blockIdx.y;
    if (j > m) return;
    const uint64_t a = j == 0 ? 0 : first + (uint64_t)(j - 1) * rec + corr_at + corr_len;
    const uint64_t b = j == m ? total : first + (uint64_t)j * rec + corr_at;
    if (b <= a) return;
    uint64_t a16 = (a + 15) & ~15ull, b16 = b & ~15ull;
    if (a16 > b16) a16 = b16 = b;  // shorter than one aligned word: bytes only
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = a + tid; i < a16; i += nth) dst_mapped[i] = img[i];
    if (od.n_direct && j < m) {
        // record j's broadcast vector lies in this piece: the words inside the tiles the extraction kernel has sent already are left
        // out (the image holds every tile, so a word across the boundary is simply copied)
        const uint64_t v0 = first + (uint64_t)j * rec + od.rvec_at;
        const uint64_t v1 = std::min(v0 + (uint64_t)od.n_direct * od.tile, v0 + od.rvec_len);  // [v0, v1): the tiles sent already
        const uint64_t ve = v0 + od.rvec_len;
        for (uint64_t i = a16 + 16 * tid; i < b16; i += 16 * nth) {
            // a whole word of the vector that starts in one of those tiles is there (k_extract_rows: LA)
            if (i >= v0 && i < v1 && i + 16 <= ve) {
                const uint64_t last = std::min(v1, ve - 15);                        // first word start that is NOT there (or beyond)
                const uint64_t nsk = (last - i + 16 * nth - 1) / (16 * nth);  // this thread's words up to it
                i += (nsk - 1) * 16 * nth;
                continue;
            }
            *(uint4*)(dst_mapped + i) = *(const uint4*)(img + i);
        }
    } else {
        for (uint64_t i = a16 + 16 * tid; i < b16; i += 16 * nth) *(uint4*)(dst_mapped + i) = *(const uint4*)(img + i);
    }
    for (uint64_t i = b16 + tid; i < b; i += nth) dst_mapped[i] = img[i];
}
void launch_copy_gaps(hipStream_t st, const uint8_t* d_img, uint8_t* dst_mapped, uint64_t total, uint64_t first, uint64_t rec, uint64_t corr_at,
                      uint64_t corr_len, uint32_t n_rec, const uint8_t* d_omit, uint32_t rep_limit, OpenDirect od, const int* d_err, int* err_dst_mapped) {
    if (!od.n_direct || od.tile < 16 || (od.tile & (od.tile - 1))) od = OpenDirect();
    // (the last piece may be most of the image -- Z64 with few staged repetitions --: enough workgroups per piece to fill PCIe alone)
    hipLaunchKernelGGL(k_copy_gaps, dim3(rep_limit < RV_TOTAL_REPS ? 64 : 8, n_rec + 1), dim3(256), 0, st, d_img, dst_mapped, total, first, rec, corr_at, corr_len,
                       n_rec, d_omit, rep_limit, od, d_err, err_dst_mapped);
}

// n_words of device memory into host-mapped memory, then (ordered behind them at system scope) a sequence number the
// host polls: how the host learns the challenge in the middle of a proof without a stream synchronisation
__global__ __launch_bounds__(256) void k_publish(const uint32_t* __restrict__ src, uint32_t n_words, uint32_t* __restrict__ dst_mapped, uint32_t* __restrict__ flag_mapped, uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) dst_mapped[i] = src[i];  // (n_words = 0: a progress stamp only)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_mapped, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish(hipStream_t st, const uint32_t* d_src, uint32_t n_words, uint32_t* dst_mapped, uint32_t* flag_mapped, uint32_t seq) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(n_words ? 256 : 64), 0, st, d_src, n_words, dst_mapped, flag_mapped, seq);
}

}  // namespace rv
