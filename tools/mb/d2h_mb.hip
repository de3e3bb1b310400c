// Micro-benchmark: how fast can a KERNEL move finished pieces of the proof to page-locked host memory, next to the
// copy engine's hipMemcpyAsync, and does it run beside an HBM-bound kernel on another stream?
//   hipcc --offload-arch=gfx950 -O3 tools/mb/d2h_mb.hip -o tools/mb/d2h_mb.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
struct Piece { size_t off, len; };
constexpr int MAXP = 96;
struct Pieces { Piece p[MAXP]; int n; };

// each workgroup walks the pieces' 16-byte body in turns of gridDim.x * 4 KiB; heads / tails bytewise
template <int NT>
__global__ __launch_bounds__(256) void k_copy_pieces(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, Pieces P) {
    for (int i = 0; i < P.n; i++) {
        const size_t off = P.p[i].off, len = P.p[i].len;
        const size_t a = (off + 15) & ~(size_t)15, e = (off + len) & ~(size_t)15;
        if (blockIdx.x == 0) {
            if (off + threadIdx.x < a && threadIdx.x < len) dst[off + threadIdx.x] = src[off + threadIdx.x];
            if (e + threadIdx.x < off + len && e >= a) dst[e + threadIdx.x] = src[e + threadIdx.x];
        }
        if (e <= a) continue;
        const size_t nv = (e - a) >> 4;
        const v4u* s = (const v4u*)(src + a);
        v4u* d = (v4u*)(dst + a);
        for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (size_t)gridDim.x * 256) {
            v4u x = s[v];
            if (NT) __builtin_nontemporal_store(x, d + v); else d[v] = x;
        }
    }
}
__global__ void k_stream(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 x = a[i]; x.x ^= 1; b[i] = x;
    }
}
int main() {
    const size_t total = 50196120;
    uint8_t *d, *h, *h2;
    CK(hipMalloc(&d, total + (1 << 20)));
    CK(hipHostMalloc(&h, total + (1 << 20), hipHostMallocDefault));
    CK(hipHostMalloc(&h2, total + (1 << 20), hipHostMallocNonCoherent));
    CK(hipMemset(d, 1, total));
    hipStream_t st, st2;
    CK(hipStreamCreate(&st)); CK(hipStreamCreate(&st2));
    auto timeit = [&](auto&& f, const char* what, size_t bytes) {
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipStreamSynchronize(st); hipStreamSynchronize(st2);
            auto t0 = std::chrono::steady_clock::now();
            f();
            hipStreamSynchronize(st); hipStreamSynchronize(st2);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep && ms < best) best = ms;
        }
        printf("%-70s %.3f ms  %.1f GB/s\n", what, best, bytes / best / 1e6); fflush(stdout);
    };
    timeit([&] { hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st); }, "hipMemcpyAsync contiguous 50 MB (coherent host)", total);
    timeit([&] { hipMemcpyAsync(h2, d, total, hipMemcpyDeviceToHost, st); }, "hipMemcpyAsync contiguous 50 MB (non-coherent host)", total);
    char name[128];
    for (int host = 0; host < 2; host++) {
        uint8_t* H = host ? h2 : h;
        for (int grid : {8, 16, 32, 64, 128, 256, 512, 1024}) {
            Pieces P; P.n = 1; P.p[0] = {0, total};
            for (int nt = 0; nt < 2; nt++) {
                snprintf(name, sizeof name, "kernel copy contiguous, %s host, grid %d, %s", host ? "noncoh" : "coh", grid, nt ? "nontemporal" : "plain");
                if (nt) timeit([&] { hipLaunchKernelGGL(k_copy_pieces<1>, dim3(grid), dim3(256), 0, st, d, H, P); }, name, total);
                else timeit([&] { hipLaunchKernelGGL(k_copy_pieces<0>, dim3(grid), dim3(256), 0, st, d, H, P); }, name, total);
            }
        }
    }
    // the proof's geometry: 40 records of 1254903 bytes, each with two vectors of 626774 bytes at odd offsets; chunk = 1/8 of every vector
    for (int chunks : {4, 8, 16}) {
        for (int grid : {32, 64, 128, 256}) {
            const size_t rec = 1254903, vec = 626774, W = vec / chunks;
            snprintf(name, sizeof name, "kernel copy, %d chunk launches of 80 pieces x %zu B, grid %d (nt)", chunks, W, grid);
            timeit([&] {
                for (int c = 0; c < chunks; c++) {
                    Pieces P; P.n = 80;
                    for (int r = 0; r < 40; r++) for (int k = 0; k < 2; k++) P.p[2 * r + k] = {33000 + r * rec + 137 + k * (vec + 8) + c * W, W};
                    hipLaunchKernelGGL(k_copy_pieces<1>, dim3(grid), dim3(256), 0, st, d, h, P);
                }
            }, name, (size_t)chunks * 80 * W);
        }
    }
    // beside an HBM streamer on the other stream (2 x 1.5 GB)
    {
        const size_t n = (size_t)1536 << 20;
        uint4 *a, *b;
        CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n));
        CK(hipMemset(a, 0, n));
        timeit([&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, st2, a, b, n / 16); }, "HBM streamer alone (read 1.5 GB + write 1.5 GB)", 2 * n);
        Pieces P; P.n = 1; P.p[0] = {0, total};
        for (int grid : {32, 64, 128}) {
            snprintf(name, sizeof name, "copy kernel (grid %d, nt) + HBM streamer on the other stream", grid);
            timeit([&] {
                hipLaunchKernelGGL(k_copy_pieces<1>, dim3(grid), dim3(256), 0, st, d, h, P);
                hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, st2, a, b, n / 16);
            }, name, total);
        }
        timeit([&] {
            hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st);
            hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, st2, a, b, n / 16);
        }, "hipMemcpyAsync + HBM streamer on the other stream", total);
    }
    return 0;
}
