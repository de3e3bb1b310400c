// Kernel launches that can be RECORDED and replayed for a whole batch of proofs (rv_prove_batch).
//
// The per-proof phases of a small proof (keys, masks, transcript digests, Fiat-Shamir, openings) are ~35 launches of
// kernels that each fill a few percent of the chip; proving 256 statements of one circuit used to issue 256 x 35 of
// them (the host API, not the GPU, was the limit: ~0.3 ms per proof).  Every proof of a batch runs the SAME sequence
// with the SAME grids -- only the buffers differ -- so the batch driver records each proof's sequence instead of
// launching it, then issues every step ONCE with gridDim.y = number of proofs; block (x, y) takes its arguments from
// element y of a device array of packed argument blocks.
//
// A kernel takes part by having its body in a functor (`struct B_foo { __device__ void operator()(args...) const; }`);
// the named __global__ kernel stays (profiles keep their kernel names) and forwards to the functor, and
// rv::launch<B_foo, LB>(k_foo, stream, grid, block, args...) either launches it or records the call.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace rv {

// plain aggregate of the kernel's parameters (host and device agree on the layout: same compiler, same type)
template <class... A>
struct Pack;
template <>
struct Pack<> {
    static constexpr int size = 0;
};
template <class H, class... T>
struct Pack<H, T...> {
    static constexpr int size = 1 + (int)sizeof...(T);
    H head;
    Pack<T...> tail;
};
template <class... A>
struct MakePack;
template <>
struct MakePack<> {
    static Pack<> make() { return {}; }
};
template <class H, class... T>
struct MakePack<H, T...> {
    static Pack<H, T...> make(H h, T... t) { return Pack<H, T...>{h, MakePack<T...>::make(t...)}; }
};
template <class Body, class P, class... B>
__device__ __forceinline__ void pack_call(const P& p, B... b) {
    if constexpr (P::size == 0)
        Body{}(b...);
    else
        pack_call<Body>(p.tail, b..., p.head);
}

template <class Body, int LB, class P>
__global__ __launch_bounds__(LB) void k_many(const P* __restrict__ arr) {
    const P p = arr[blockIdx.y];
    pack_call<Body>(p);
}

struct LaunchRecorder {
    struct Call {
        // kernel: issue the step for `batch` proofs whose argument blocks sit at d_args (stride arg_bytes)
        void (*replay)(hipStream_t, dim3, dim3, const void*, unsigned) = nullptr;
        dim3 grid, block;
        uint32_t arg_bytes = 0;
        // replay == nullptr: an asynchronous copy, replayed as it is
        void* dst = nullptr;
        const void* src = nullptr;
        size_t n = 0;
        hipMemcpyKind kind = hipMemcpyDefault;
        std::vector<uint8_t> args;
    };
    std::vector<Call> calls;
    unsigned batch = 1;  // proofs the recorded calls will be replayed for (launchers may size their grids by it)
};
// the recorder of the calling thread (null: launches go straight to the stream)
inline thread_local LaunchRecorder* g_recorder = nullptr;

template <class Body, int LB, class P>
void replay_many(hipStream_t st, dim3 grid, dim3 block, const void* d_args, unsigned batch) {
    hipLaunchKernelGGL((k_many<Body, LB, P>), dim3(grid.x, batch), block, 0, st, (const P*)d_args);
}

// grid.y / grid.z must be 1 (y carries the proof index when the call is replayed for a batch)
template <class Body, int LB, class... A, class... X>
void launch(void (*kern)(A...), hipStream_t st, dim3 grid, dim3 block, X... x) {
    if (LaunchRecorder* r = g_recorder) {
        using P = Pack<A...>;
        const P p = MakePack<A...>::make(static_cast<A>(x)...);
        LaunchRecorder::Call c;
        c.replay = &replay_many<Body, LB, P>;
        c.grid = grid;
        c.block = block;
        c.arg_bytes = (uint32_t)sizeof(P);
        c.args.resize(sizeof(P));
        memcpy(c.args.data(), &p, sizeof(P));
        r->calls.push_back(std::move(c));
    } else {
        hipLaunchKernelGGL(kern, grid, block, 0, st, static_cast<A>(x)...);
    }
}

inline hipError_t memcpy_async(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st) {
    if (LaunchRecorder* r = g_recorder) {
        LaunchRecorder::Call c;
        c.dst = dst;
        c.src = src;
        c.n = n;
        c.kind = kind;
        r->calls.push_back(std::move(c));
        return hipSuccess;
    }
    return hipMemcpyAsync(dst, src, n, kind, st);
}

}  // namespace rv
