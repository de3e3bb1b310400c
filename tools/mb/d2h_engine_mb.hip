// Micro-benchmark: WHICH engine carries a device-to-host hipMemcpyAsync of a proof-sized buffer, and what it does to an
// HBM-bound kernel running beside it.  The runtime's copy of 50 MB into hipHostMalloc memory shows up in a kernel trace as
// __amd_rocclr_copyBuffer (a shader blit), and shader stores to host memory throttle HBM-bound kernels ~3x (DESIGN.md,
// "Openings in slices ..."); the SDMA engines do not.  Variants: destination kinds, hipMemcpyDtoHAsync, and the HSA
// runtime's hsa_amd_memory_async_copy called directly (CPU agent <- GPU agent: an SDMA engine).
//   hipcc --offload-arch=gfx950 -O3 tools/mb/d2h_engine_mb.hip -o tools/mb/d2h_engine_mb.bin -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_stream(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 x = a[i]; x.x ^= 1; b[i] = x;
    }
}
static hsa_agent_t g_gpu{}, g_cpu{};
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_gpu.handle) g_gpu = a;
    if (t == HSA_DEVICE_TYPE_CPU && !g_cpu.handle) g_cpu = a;
    return HSA_STATUS_SUCCESS;
}
int main() {
    const size_t total = 50196120;
    const size_t SN = (size_t)1 << 30;  // streamer: 1 GiB in, 1 GiB out
    uint8_t *d, *sa, *sb;
    CK(hipMalloc(&d, total + (1 << 20)));
    CK(hipMalloc(&sa, SN)); CK(hipMalloc(&sb, SN));
    CK(hipMemset(d, 1, total)); CK(hipMemset(sa, 2, SN));
    uint8_t *hA, *hB, *hC, *hD;
    CK(hipHostMalloc(&hA, total + 4096, hipHostMallocDefault));
    CK(hipHostMalloc(&hB, total + 4096, hipHostMallocNonCoherent));
    hC = (uint8_t*)aligned_alloc(4096, (total + 8191) & ~(size_t)4095);
    CK(hipHostRegister(hC, total + 4096, hipHostRegisterDefault));
    CK(hipHostMalloc(&hD, total + 4096, hipHostMallocPortable | hipHostMallocMapped));
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t e0, e1, c0, c1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
    auto streamer = [&]() { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, st, (const uint4*)sa, (uint4*)sb, SN / 16); };
    // alone
    for (int i = 0; i < 3; i++) { hipEventRecord(e0, st); streamer(); hipEventRecord(e1, st); hipStreamSynchronize(st); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("streamer alone                                   %.3f ms (%.2f TB/s)\n", ms, 2.0 * SN / ms / 1e9);
    struct V { const char* name; uint8_t* h; int api; } vs[] = {
        {"hipHostMalloc default, hipMemcpyAsync", hA, 0}, {"hipHostMalloc non-coherent, hipMemcpyAsync", hB, 0},
        {"malloc + hipHostRegister, hipMemcpyAsync", hC, 0}, {"hipHostMalloc portable|mapped, hipMemcpyAsync", hD, 0},
        {"hipHostMalloc default, hipMemcpyDtoHAsync", hA, 1},
    };
    for (auto& v : vs) {
        float best_c = 1e9, with_s = 0, with_c = 0;
        for (int i = 0; i < 3; i++) {
            hipEventRecord(c0, st2);
            if (v.api == 0) CK(hipMemcpyAsync(v.h, d, total, hipMemcpyDeviceToHost, st2));
            else CK(hipMemcpyDtoHAsync(v.h, (hipDeviceptr_t)d, total, st2));
            hipEventRecord(c1, st2); hipStreamSynchronize(st2);
            hipEventElapsedTime(&ms, c0, c1); if (ms < best_c) best_c = ms;
        }
        for (int i = 0; i < 3; i++) {
            hipEventRecord(e0, st); streamer(); hipEventRecord(e1, st);
            hipEventRecord(c0, st2);
            if (v.api == 0) CK(hipMemcpyAsync(v.h, d, total, hipMemcpyDeviceToHost, st2));
            else CK(hipMemcpyDtoHAsync(v.h, (hipDeviceptr_t)d, total, st2));
            hipEventRecord(c1, st2);
            hipStreamSynchronize(st); hipStreamSynchronize(st2);
            hipEventElapsedTime(&with_s, e0, e1); hipEventElapsedTime(&with_c, c0, c1);
        }
        printf("%-48s alone %.3f ms (%.1f GB/s) | together: copy %.3f ms, streamer %.3f ms\n", v.name, best_c, total / best_c / 1e6, with_c, with_s);
        fflush(stdout);
    }
    // what precedes the copy on its stream: nothing (above), a small kernel, an event wait on the other stream; and the copy
    // issued BEFORE the streamer
    __attribute__((unused)) auto tiny = [&](hipStream_t q) { hipLaunchKernelGGL(k_stream, dim3(1), dim3(64), 0, q, (const uint4*)sa, (uint4*)sb, (size_t)64); };
    for (int mode = 0; mode < 4; mode++) {
        float with_s = 0, with_c = 0;
        for (int i = 0; i < 3; i++) {
            if (mode == 3) {
                hipEventRecord(c0, st2);
                CK(hipMemcpyAsync(hA, d, total, hipMemcpyDeviceToHost, st2));
                hipEventRecord(c1, st2);
                hipEventRecord(e0, st); streamer(); hipEventRecord(e1, st);
            } else {
                if (mode == 2) tiny(st);
                hipEventRecord(c1, st);  // (reused as the dependency)
                hipEventRecord(e0, st); streamer(); hipEventRecord(e1, st);
                if (mode == 1) tiny(st2);
                if (mode == 2) hipStreamWaitEvent(st2, c1, 0);
                hipEventRecord(c0, st2);
                CK(hipMemcpyAsync(hA, d, total, hipMemcpyDeviceToHost, st2));
                hipEventRecord(c1, st2);
            }
            hipStreamSynchronize(st); hipStreamSynchronize(st2);
            hipEventElapsedTime(&with_s, e0, e1); hipEventElapsedTime(&with_c, c0, c1);
        }
        const char* names[4] = {"copy on an idle stream", "copy behind a tiny kernel on its stream", "copy behind an event wait", "copy first, then the streamer"};
        printf("%-48s together: copy %.3f ms, streamer %.3f ms\n", names[mode], with_c, with_s);
        fflush(stdout);
    }
    // the product's situation: the copy's dependency is still RUNNING when the copy is issued -- (a) a long kernel ahead of
    // it on its own stream, (b) an event recorded behind a long kernel on the other stream.  Does the runtime still pick
    // an SDMA engine?  (rocprofv3 --kernel-trace --stats: a blit shows up as __amd_rocclr_copyBuffer)
    for (int mode = 0; mode < 2; mode++) {
        float with_c = 0, with_s = 0;
        for (int i = 0; i < 3; i++) {
            hipStream_t q = mode == 0 ? st2 : st;
            hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, q, (const uint4*)sa, (uint4*)sb, SN / 16);
            if (mode == 1) { hipEventRecord(c1, st); hipStreamWaitEvent(st2, c1, 0); }
            if (!getenv("NO_EV")) hipEventRecord(c0, st2);   // NO_EV=1: nothing between the dependency and the copy, as in the product
            CK(hipMemcpyAsync(hA, d, total, hipMemcpyDeviceToHost, st2));
            if (!getenv("NO_EV")) hipEventRecord(c1, st2);
            // ... and a second kernel that runs beside the copy
            hipEventRecord(e0, st); streamer(); hipEventRecord(e1, st);
            hipStreamSynchronize(st); hipStreamSynchronize(st2);
            if (!getenv("NO_EV")) hipEventElapsedTime(&with_c, c0, c1);
            hipEventElapsedTime(&with_s, e0, e1);
        }
        printf("%-48s copy %.3f ms, a streamer beside it %.3f ms\n", mode == 0 ? "copy behind a RUNNING kernel on its stream" : "copy behind an event of a RUNNING kernel", with_c, with_s);
        fflush(stdout);
    }
    // the HSA runtime directly
    if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 0; }
    hsa_iterate_agents(agent_cb, nullptr);
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    for (int both = 0; both < 2; both++) {
        double best = 1e9; float with_s = 0;
        for (int i = 0; i < 3; i++) {
            hipStreamSynchronize(st);
            hsa_signal_store_relaxed(sig, 1);
            if (both) { hipEventRecord(e0, st); streamer(); hipEventRecord(e1, st); }
            auto t0 = std::chrono::steady_clock::now();
            hsa_status_t s = hsa_amd_memory_async_copy(hA, g_cpu, d, g_gpu, total, 0, nullptr, sig);
            if (s != HSA_STATUS_SUCCESS) { printf("hsa_amd_memory_async_copy: %d\n", (int)s); return 0; }
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (t < best) best = t;
            hipStreamSynchronize(st);
            if (both) hipEventElapsedTime(&with_s, e0, e1);
        }
        printf("hsa_amd_memory_async_copy (cpu <- gpu) %-9s host-timed %.3f ms (%.1f GB/s)%s", both ? "together:" : "alone:", best, total / best / 1e6, both ? "" : "\n");
        if (both) printf(", streamer %.3f ms\n", with_s);
    }
    // dst agent = GPU agent (how a runtime would ask for a blit / different engine choice)
    {
        hsa_signal_store_relaxed(sig, 1);
        auto t0 = std::chrono::steady_clock::now();
        hsa_status_t s = hsa_amd_memory_async_copy(hA, g_gpu, d, g_gpu, total, 0, nullptr, sig);
        if (s == HSA_STATUS_SUCCESS) {
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("hsa_amd_memory_async_copy (gpu <- gpu agents)    host-timed %.3f ms (%.1f GB/s)\n", t, total / t / 1e6);
        } else printf("hsa copy gpu<-gpu: status %d\n", (int)s);
    }
    return 0;
}
