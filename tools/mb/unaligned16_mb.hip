// Does global_store_dwordx4 / global_load_dwordx4 work at 8-byte (not 16-byte) aligned addresses on gfx950?  (and v_permlane{16,32}_swap semantics)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/un16 tools/mb/unaligned16_mb.hip && /tmp/un16
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
__global__ void k_st(uint64_t* o) {
    uint64_t* p = o + 1 + 2 * threadIdx.x;  // 8 mod 16
    u64x2 v = {1000 + 2 * threadIdx.x, 1001 + 2 * threadIdx.x};
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__global__ void k_ld(const uint64_t* o, uint64_t* r) {
    const uint64_t* p = o + 1 + 2 * threadIdx.x;
    u64x2 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    r[2 * threadIdx.x] = v.x;
    r[2 * threadIdx.x + 1] = v.y;
}
__global__ void k_swap(uint32_t* o) {
    uint32_t a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + threadIdx.x] = s[0];
    o[192 + threadIdx.x] = s[1];
}
int main() {
    uint64_t *d, *r;
    hipMalloc(&d, 4096);
    hipMalloc(&r, 4096);
    hipMemset(d, 0, 4096);
    k_st<<<1, 64>>>(d);
    k_ld<<<1, 64>>>(d, r);
    std::vector<uint64_t> h(512), g(512);
    hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(g.data(), r, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; i++) bad += (h[1 + i] != 1000 + (uint64_t)i) + (g[i] != 1000 + (uint64_t)i);
    printf("unaligned dwordx4 store/load: %s (h[0]=%llu h[1]=%llu h[128]=%llu h[129]=%llu)\n", bad ? "BAD" : "ok", (unsigned long long)h[0], (unsigned long long)h[1],
           (unsigned long long)h[128], (unsigned long long)h[129]);
    uint32_t* s;
    hipMalloc(&s, 1024);
    k_swap<<<1, 64>>>(s);
    std::vector<uint32_t> hs(256);
    hipMemcpy(hs.data(), s, 1024, hipMemcpyDeviceToHost);
    printf("permlane32_swap(a = lane, b = 100 + lane): a' lanes 0,31,32,63 = %u %u %u %u   b' = %u %u %u %u\n", hs[0], hs[31], hs[32], hs[63], hs[64], hs[95], hs[96], hs[127]);
    printf("permlane16_swap: a' lanes 0,15,16,31,32,48 = %u %u %u %u %u %u   b' = %u %u %u %u %u %u\n", hs[128], hs[143], hs[144], hs[159], hs[160], hs[176], hs[192], hs[207],
           hs[208], hs[223], hs[224], hs[240]);
    return bad;
}
