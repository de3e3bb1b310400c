// Thread scaling of the streaming prover's piece compiler (host only): T threads compile the same piece of ops K times each, as a stream's workers
// would (compile_ops_seq with a ChunkStart).  Build + run (tools/mb/chunk_compile_mb.sh):
//   hipcc -O3 -std=c++17 -x hip --offload-arch=gfx950 -Ireverie_amd/csrc -Iinclude reverie_amd/csrc/compile.cpp reverie_amd/csrc/compile_par.cpp tools/mb/chunk_compile_mb.cpp -o /tmp/ccmb
//   /tmp/ccmb piece.bin <gf2 wires> <K> <T> [<T> ...]
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <thread>
#include <vector>

#include "compile.h"
using namespace rv;
int main(int argc, char** argv) {
    if (argc < 5) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const size_t n = ftell(f) / sizeof(rv_op);
    fseek(f, 0, SEEK_SET);
    std::vector<rv_op> ops(n);
    if (fread(ops.data(), sizeof(rv_op), n, f) != n) return 1;
    fclose(f);
    const size_t wires = atol(argv[2]);
    const int K = atoi(argv[3]);
    for (int a = 4; a < argc; a++) {
        const int T = atoi(argv[a]);
        std::vector<double> ms(T, 0);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                ChunkStart cs;
                for (int k = 0; k < K; k++) {
                    Compiled cc;
                    const auto s = std::chrono::steady_clock::now();
                    if (compile_ops_seq(ops.data(), n, 0, wires, cc, &cs, 0)) abort();
                    ms[t] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s).count();
                }
            });
        for (auto& x : th) x.join();
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double sum = 0;
        for (double v : ms) sum += v;
        printf("%3d threads x %d compiles of %zu ops: %.1f ms per compile (mean), wall %.1f ms, %.1f compiles / s\n", T, K, n, sum / (T * K), wall, T * K / wall * 1e3);
    }
    return 0;
}
