// Micro-benchmark: can the proof leave in column chunks while the openings are still being extracted?
// hipMemcpy2DAsync device -> page-locked host of 40 rows x W bytes out of records of `pitch` bytes, against one
// contiguous copy of the same number of bytes.   hipcc --offload-arch=gfx950 -O2 copy2d_mb.hip -o copy2d_mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t pitch = 1254800, rows = 40, total = pitch * rows;
    uint8_t *d, *h;
    CK(hipMalloc(&d, total));
    CK(hipHostMalloc(&h, total, hipHostMallocDefault));
    CK(hipMemset(d, 1, total));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int chunks : {1, 2, 4, 8}) {
        const size_t W = (pitch / chunks) & ~(size_t)255;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < chunks; c++) CK(hipMemcpy2DAsync(h + c * W, pitch, d + c * W, pitch, W, rows, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("2D  %d chunks of 40 x %zu B: %.3f ms  (%.1f GB/s)\n", chunks, W, ms, chunks * W * rows / ms / 1e6);
        }
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep) printf("1D  contiguous %zu B: %.3f ms  (%.1f GB/s)\n", total, ms, total / ms / 1e6);
    }
    // unaligned geometry (records of odd size, ranges that start at odd columns), as the proof has it
    {
        const size_t p2 = 1254903, W = 313007, off = 40 + 137;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < 3; c++) {
                auto a = std::chrono::steady_clock::now();
                CK(hipMemcpy2DAsync(h + off + c * W, p2, d + off + c * W, p2, W, 39, hipMemcpyDeviceToHost, st));
                double call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
                if (rep) printf("   call %d returned after %.3f ms\n", c, call_ms);
            }
            CK(hipStreamSynchronize(st));
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("2D  unaligned 3 x (39 x %zu B): %.3f ms  (%.1f GB/s)\n", W, ms, 3 * W * 39 / ms / 1e6);
        }
    }
    // device side aligned (pitch and column starts multiples of 256), host side as the proof has it
    for (int variant = 0; variant < 3; variant++) {
        const size_t pd = 1255168 /* 256 x 4903 */, ph = 1254903, W = variant == 2 ? 313007 : 313088 /* 256 x 1223 */;
        const size_t offh = variant == 0 ? 256 : 40 + 137;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < 3; c++) CK(hipMemcpy2DAsync(h + offh + c * W, ph, d + 256 + c * 313088, pd, W, 39, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("2D  src aligned, dst %s, width %zu: %.3f ms  (%.1f GB/s)\n", variant == 0 ? "offset 256 / odd pitch" : "odd offset / odd pitch", W, ms, 3 * W * 39 / ms / 1e6);
        }
    }
    // 40 separate contiguous copies of W bytes (one per record) as the alternative to a 2D copy
    {
        const size_t W = pitch / 4;
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (size_t r = 0; r < rows; r++) CK(hipMemcpyAsync(h + r * pitch, d + r * pitch, W, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("40 x 1D of %zu B: %.3f ms (%.1f GB/s)\n", W, ms, W * rows / ms / 1e6);
    }
    return 0;
}
