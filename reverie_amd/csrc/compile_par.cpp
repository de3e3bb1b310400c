// Parallel gate-stream compiler: the same result as compile_ops_seq (compile.cpp), bit for bit, computed by several host
// threads.  The reference needs no such step -- it walks the raw op list inside every Proof::new
// (/root/reference/src/proof/mod.rs:150-152) -- so the drop-in pays the levelisation on the first proof of a circuit: one core
// took 0.9 s for the 10^7-gate benchmark circuit before a 6 ms proof.
//
// What is sequential in the op list and how each part is made parallel:
//   * counters (ShareGen::next() call numbers, transcript rows, ordinals, SSA ids) are prefix sums over the ops:
//     counted per contiguous range, scanned over the ranges, and kept per block of BLK ops (`blocks`);
//   * "which write does this read see" (the interpreter's wire vector, interpreter/single.rs:14-16): every range resolves
//     the reads that hit its own earlier writes in a range-local table; the rest are answered range by range from a global
//     table that the ranges' last writes are folded into in order (each fold and each answer round is itself parallel);
//   * fan-out counts: atomic increments;
//   * linear forms and dependency levels follow the data flow: blocks of ops are dealt round-robin to the threads, each
//     thread takes its blocks in order and waits (spinning on a per-block flag) only when an operand was produced by a
//     block that is not finished yet -- operands always point backwards, so the owner of the lowest unfinished block never
//     waits.  Wide circuits run all threads; a long dependency chain degenerates to the sequential order;
//   * the stable sort by (level, class) is a parallel counting sort; gates are built once, at their final position, with
//     their final row numbers (a materialised row is first named by the SSA id of the wire it carries and renumbered densely
//     in program order afterwards, which is the order the sequential compiler numbers them in).
// Programs with B2A gates, streaming chunks, small programs and anything that would be an error go to the sequential
// compiler (which also produces the canonical error code).
#include "compile.h"

#include <unistd.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <thread>

#if defined(__x86_64__)
#include <immintrin.h>
#define RV_PAUSE() _mm_pause()
#else
#define RV_PAUSE() ((void)0)
#endif

namespace rv {

namespace {

constexpr int K = RV_LIN_K;
constexpr uint32_t COMP = 0x80000000u;
constexpr uint32_t ZERO_ROW = COMP | 0;
constexpr uint32_t BLK = 1024;   // ops per block (counter snapshots, unit of the data-flow pass)
constexpr uint8_t CLS_NONE = 0xFF, CLS_Z64 = 0xFE;
constexpr size_t PF = 12;       // software prefetch distance (ops) for operand records

// A wire as XOR of n base rows (sorted, distinct) plus the constant c, and the dependency level of the wire (the largest
// level of its rows, -1 for a constant): 16 bytes, so that a gate finds everything it needs to know about an operand in ONE
// cache line -- on many cores nearly every operand was produced by another core, and each line read is a remote miss.
struct LinP {
    uint32_t b[K];
    uint32_t meta;  // (level + 1) << 4 | n << 1 | c
};
constexpr uint32_t MAX_LEVEL = (1u << 28) - 4;
inline uint32_t l_n(const LinP& L) { return (L.meta >> 1) & 3u; }
inline uint32_t l_c(const LinP& L) { return L.meta & 1u; }
inline int32_t l_lvl(const LinP& L) { return (int32_t)(L.meta >> 4) - 1; }
inline uint32_t l_meta(int32_t lvl, uint32_t n, uint32_t c) { return (uint32_t)(lvl + 1) << 4 | n << 1 | c; }
inline LinP lin_zero(uint32_t c = 0) {
    LinP L;
    for (int i = 0; i < K; i++) L.b[i] = 0;
    L.meta = l_meta(-1, 0, c);
    return L;
}
inline LinP lin_base(uint32_t row, int32_t lvl) {
    LinP L = lin_zero();
    L.b[0] = row;
    L.meta = l_meta(lvl, 1, 0);
    return L;
}

// counters in front of an op (the values the sequential compiler's members have when it reaches the op)
struct Ctr {
    uint32_t ssa = 1, masks = 0, on = 0, pre = 0, rec = 0, in = 0;                               // GF(2)
    uint32_t ssa64 = 1, masks64 = 0, in64 = 0, rec64 = 0, corr64 = 0, gates64 = 0;              // Z64
    uint64_t onw64 = 0, prew64 = 0;
    uint32_t gf2_linear_random = 0, gf2_inputs = 0, gf2_muls = 0, gf2_asserts = 0, z64_inputs = 0, z64_muls = 0, z64_asserts = 0, z64_linear = 0;
};
inline uint8_t make_kind(const rv_op& op) { return (uint8_t)(op.domain << 6 | op.opcode << 1 | (uint8_t)(op.imm & 1)); }
inline uint32_t k_dom(uint8_t k) { return k >> 6; }
inline uint32_t k_opc(uint8_t k) { return (k >> 1) & 0xF; }
inline uint32_t k_bit(uint8_t k) { return k & 1; }

inline void advance(Ctr& c, uint8_t k) {
    const uint32_t opc = k_opc(k);
    if (k_dom(k) == RV_DOM_GF2) {
        switch (opc) {
        case RV_OP_INPUT: c.ssa++, c.masks++, c.on++, c.in++, c.gf2_inputs++; break;
        case RV_OP_RANDOM: c.ssa++, c.masks++, c.gf2_linear_random++; break;
        case RV_OP_MUL: c.ssa++, c.masks += 2, c.on++, c.pre++, c.rec++, c.gf2_muls++; break;
        case RV_OP_ASSERTZERO: c.on++, c.rec++, c.gf2_asserts++; break;
        default: c.ssa++; break;  // Const, Add, Sub, AddConst, SubConst, MulConst
        }
    } else if (k_dom(k) == RV_DOM_Z64) {
        c.gates64++;
        switch (opc) {
        case RV_OP_INPUT: c.ssa64++, c.masks64++, c.onw64 += 1, c.in64++, c.z64_inputs++; break;
        case RV_OP_RANDOM: c.ssa64++, c.masks64++, c.z64_linear++; break;
        case RV_OP_MUL: c.ssa64++, c.masks64 += 2, c.prew64 += 1, c.corr64++, c.onw64 += 8, c.rec64++, c.z64_muls++; break;
        case RV_OP_ASSERTZERO: c.onw64 += 8, c.rec64++, c.z64_asserts++; break;
        default: c.ssa64++, c.z64_linear++; break;
        }
    }
}
inline void add_ctr(Ctr& a, const Ctr& d) {  // a += d where d was counted from a zero start (ssa / ssa64 start at 1: subtract it)
    a.ssa += d.ssa - 1, a.masks += d.masks, a.on += d.on, a.pre += d.pre, a.rec += d.rec, a.in += d.in;
    a.ssa64 += d.ssa64 - 1, a.masks64 += d.masks64, a.in64 += d.in64, a.rec64 += d.rec64, a.corr64 += d.corr64, a.gates64 += d.gates64;
    a.onw64 += d.onw64, a.prew64 += d.prew64;
    a.gf2_linear_random += d.gf2_linear_random, a.gf2_inputs += d.gf2_inputs, a.gf2_muls += d.gf2_muls, a.gf2_asserts += d.gf2_asserts;
    a.z64_inputs += d.z64_inputs, a.z64_muls += d.z64_muls, a.z64_asserts += d.z64_asserts, a.z64_linear += d.z64_linear;
}

// how many operands an op reads (GF(2) and Z64 alike)
inline int n_reads(uint32_t opc) {
    switch (opc) {
    case RV_OP_ADD: case RV_OP_SUB: case RV_OP_MUL: return 2;
    case RV_OP_ADDCONST: case RV_OP_SUBCONST: case RV_OP_MULCONST: case RV_OP_ASSERTZERO: return 1;
    default: return 0;
    }
}
inline bool has_dst(uint32_t opc) { return opc != RV_OP_ASSERTZERO; }

// ---- a small pool: run(f) executes f(thread) on every thread (the caller is thread 0) and returns when all are done ----
class Pool {
    int n_;
    std::vector<std::thread> th_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> pending_{0};
    std::atomic<bool> stop_{false};
    const std::function<void(int)>* fn_ = nullptr;
    // the first exception a pass throws on any thread (a bad_alloc while a pass grows its vectors is the realistic one): kept until
    // every thread has left the pass, then rethrown on the calling thread -- never out of a worker (std::terminate), and never
    // while workers still run over the caller's locals
    std::mutex err_mu_;
    std::exception_ptr err_;
    int pass_ = 0;
    const int throw_at_ = getenv("RV_TEST_POOL_THROW") ? atoi(getenv("RV_TEST_POOL_THROW")) : 0;
    void note_exception() {
        std::lock_guard<std::mutex> g(err_mu_);
        if (!err_) err_ = std::current_exception();
    }
    void worker(int t) {
        uint64_t seen = 0;
        for (;;) {
            uint32_t spins = 0;
            while (gen_.load(std::memory_order_acquire) == seen) {
                RV_PAUSE();
                if (++spins > 2000) {
                    std::this_thread::yield();
                    if (spins > 200000) std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load(std::memory_order_acquire)) return;
            try {
                // (test knob RV_TEST_POOL_THROW=k: the last worker fails with bad_alloc in the k-th pass of every compile)
                if (throw_at_ && t == n_ - 1 && pass_ == throw_at_) throw std::bad_alloc();
                (*fn_)(t);
            } catch (...) {
                note_exception();
            }
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

  public:
    explicit Pool(int n) : n_(n) {
        for (int t = 1; t < n; t++) th_.emplace_back([this, t] { worker(t); });
    }
    ~Pool() {
        stop_.store(true, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_acq_rel);
        for (auto& t : th_) t.join();
    }
    int size() const { return n_; }
    void run(const std::function<void(int)>& f) {
        fn_ = &f;
        pass_++;
        pending_.store(n_ - 1, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_acq_rel);
        try {
            f(0);
        } catch (...) {
            note_exception();
        }
        uint32_t spins = 0;
        while (pending_.load(std::memory_order_acquire)) {
            RV_PAUSE();
            if (++spins > 2000) std::this_thread::yield();
        }
        std::exception_ptr e;
        {
            std::lock_guard<std::mutex> g(err_mu_);
            e = err_;
            err_ = nullptr;
        }
        if (e) std::rethrow_exception(e);  // (every thread has left f: the caller may unwind)
    }
    // [lo, hi) of thread t's share of n items
    static void slice(size_t n, int t, int nt, size_t& lo, size_t& hi) {
        lo = n * (size_t)t / (size_t)nt;
        hi = n * (size_t)(t + 1) / (size_t)nt;
    }
};

template <class T>
struct Zeroed {  // zero pages, touched lazily (big_alloc: huge pages where the kernel grants them)
    T* p = nullptr;
    size_t bytes = 0;
    explicit Zeroed(size_t n) : bytes(std::max<size_t>(n, 1) * sizeof(T)) {
        p = (T*)big_alloc(bytes);
        if (!p) throw std::bad_alloc();
    }
    ~Zeroed() { release(); }
    void release() {
        big_free_later(p, bytes);
        p = nullptr;
    }
    Zeroed(const Zeroed&) = delete;
    Zeroed& operator=(const Zeroed&) = delete;
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};
template <class T>
using Raw = Zeroed<T>;  // (every element is written before it is read; the zero fill is the kernel's and costs nothing extra)

struct alignas(64) PaddedU32 {
    std::atomic<uint32_t> v{0};
};

inline int sym_diff(const LinP& A, const LinP& B, uint32_t* rows) {  // sorted symmetric difference (x ^ x = 0)
    int n = 0, i = 0, j = 0;
    const int an = (int)l_n(A), bn = (int)l_n(B);
    while (i < an || j < bn) {
        if (j >= bn || (i < an && A.b[i] < B.b[j]))
            rows[n++] = A.b[i++];
        else if (i >= an || B.b[j] < A.b[i])
            rows[n++] = B.b[j++];
        else {
            i++;
            j++;
        }
    }
    return n;
}

}  // namespace

void* big_alloc(size_t bytes) {
    if (bytes < ((size_t)4 << 20)) return calloc(bytes ? bytes : 1, 1);
    const size_t HP = (size_t)2 << 20;
    const size_t len = (bytes + HP - 1) & ~(HP - 1);
    // over-map by one huge page and trim, so that the block starts on a huge-page boundary
    uint8_t* raw = (uint8_t*)mmap(nullptr, len + HP, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw == MAP_FAILED) return nullptr;
    uint8_t* p = (uint8_t*)(((uintptr_t)raw + HP - 1) & ~(uintptr_t)(HP - 1));
    if (p > raw) munmap(raw, (size_t)(p - raw));
    const size_t tail = (size_t)((raw + len + HP) - (p + len));
    if (tail) munmap(p + len, tail);
    constexpr bool thp = true;
    if (thp) (void)madvise(p, len, MADV_HUGEPAGE);
    return p;
}
// Unmapping hundreds of MB takes tens of milliseconds and nobody waits for its result: blocks of 4 MiB and more released through
// big_free_later go to one process-wide background thread (started on first use, joined when the library is unloaded).
// munmap holds the address-space lock that every page fault and many driver calls need, so the thread stays out of the way:
// it frees nothing while a compile is running (compile_busy) and not before the releases have been quiet for a while -- the
// upload and the first proof follow a compile at once, and a loop of compiles would otherwise fault its fresh pages
// against the previous round's unmapping (20 - 30 ms per compile on the 10^7-gate circuit).  Above a cap it frees at once.
static void big_free_impl(void* p, size_t bytes, bool pause);
// library calls in flight (compiles, proofs, verifications: lib_busy in compile.h): the background thread gives nothing back while one runs
static std::atomic<int> g_lib_busy{0};
static std::atomic<int64_t> g_lib_idle_since_ns{0};  // when the last call in flight returned (steady clock)
static int64_t steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void lib_busy(int d) {
    if (d < 0) g_lib_idle_since_ns.store(steady_ns(), std::memory_order_relaxed);
    g_lib_busy.fetch_add(d, std::memory_order_relaxed);
}
// nothing in flight, and nothing for a few milliseconds: a loop of back-to-back calls leaves no gap the unmapper would take
static bool lib_quiet() {
    return g_lib_busy.load(std::memory_order_relaxed) <= 0 && steady_ns() - g_lib_idle_since_ns.load(std::memory_order_relaxed) > 3 * 1000 * 1000;
}
namespace {
struct Reaper {
    // more than this queued: free at once, in the one burst it then takes.  An eighth of the machine's memory, 4 .. 32 GiB: the
    // streaming prover queues ~6 GB per proof of the 10^7-gate circuit while its compile workers run, and with a 2 GiB cap the
    // unmapping ran against their page faults (0.125 -> 0.155 s per streamed proof)
    const size_t CAP_BYTES = [] {
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
        const size_t phys = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz : (size_t)32 << 30;
        return std::min<size_t>(std::max<size_t>(phys / 8, (size_t)4 << 30), (size_t)32 << 30);
    }();
    static constexpr int QUIET_MS = 250;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<void*, size_t>> q;
    size_t queued = 0;
    int busy = 0;
    std::chrono::steady_clock::time_point last_push;
    bool stop = false;
    std::thread th;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty() && stop) return;
            while (!stop && queued <= CAP_BYTES &&
                   (busy > 0 || std::chrono::steady_clock::now() - last_push < std::chrono::milliseconds(QUIET_MS)))
                cv.wait_for(lk, std::chrono::milliseconds(50));
            std::vector<std::pair<void*, size_t>> take;
            take.swap(q);
            const bool urgent = stop || queued > CAP_BYTES;  // (over the cap, or the library is unloading: at once, whatever runs beside it)
            queued = 0;
            lk.unlock();
            for (auto& e : take) big_free_impl(e.first, e.second, /*pause=*/!urgent);
            lk.lock();
        }
    }
    void push(void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        q.emplace_back(p, bytes);
        queued += bytes;
        last_push = std::chrono::steady_clock::now();
        cv.notify_one();
    }
    void set_busy(int d) {
        std::lock_guard<std::mutex> g(mu);
        busy += d;
    }
    ~Reaper() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
            cv.notify_one();
        }
        if (th.joinable()) th.join();
    }
};
Reaper& reaper() {
    static Reaper r;
    return r;
}
struct CompileBusy {  // while one lives, the background thread unmaps nothing
    CompileBusy() { reaper().set_busy(+1), lib_busy(+1); }
    ~CompileBusy() { reaper().set_busy(-1), lib_busy(-1); }
};
}  // namespace
void big_free_later(void* p, size_t bytes) {
    if (!p) return;
    static const bool off = getenv("RV_FREE_SYNC") && atoi(getenv("RV_FREE_SYNC")) != 0;
    if (bytes < ((size_t)4 << 20) || off)
        big_free(p, bytes);
    else
        reaper().push(p, bytes);
}
// Unmapping holds the address-space lock exclusively while the pages go back, ~25 ms per GB -- and every other thread of the process that
// faults a page, grows its heap or maps anything waits that long (round 6: the worker threads of rv_prove_batch stalled 15 - 100 ms in
// their kernel launches whenever the background thread returned the scratch of compiles done shortly before: 4.8 -> 8.7 ms per proof in
// the driver's bench).  So a block goes back in pieces of 16 MiB, the pages of a piece first through MADV_DONTNEED (shared lock), and
// from the background thread with a pause after every piece: nobody waits longer than one piece takes.
static void unmap_gently(void* p, size_t len, bool pause) {
    constexpr size_t PIECE = (size_t)16 << 20;
    uint8_t* q = (uint8_t*)p;
    while (len) {
        // (background thread: between the library's calls only -- but not for ever: a service that proves without a break gets its
        // memory back within ~2 s per block all the same)
        for (int spins = 0; pause && !lib_quiet() && spins < 10000; spins++) {
            timespec ts{0, 200 * 1000};
            nanosleep(&ts, nullptr);
        }
        const size_t n = std::min(len, PIECE);
        (void)madvise(q, n, MADV_DONTNEED);
        munmap(q, n);
        q += n;
        len -= n;
        if (pause && len) {
            timespec ts{0, 50 * 1000};
            nanosleep(&ts, nullptr);
        }
    }
}
static void big_free_impl(void* p, size_t bytes, bool pause) {
    if (!p) return;
    if (bytes < ((size_t)4 << 20)) {
        free(p);
        return;
    }
    const size_t HP = (size_t)2 << 20;
    unmap_gently(p, (bytes + HP - 1) & ~(HP - 1), pause);
}
void big_free(void* p, size_t bytes) { big_free_impl(p, bytes, false); }

// CPUs this process may use at a time: the logical CPUs, capped by the cgroup's CPU quota (cpu.max of cgroup v2, cfs_quota_us /
// cfs_period_us of v1).  The compiler's data-flow pass spin-waits: with more threads than granted CPUs the kernel throttles the whole
// group for the rest of each period (the benchmark boxes grant 16 of 256 logical CPUs -- 16 / 24 / 32 / 64 threads: 82 / 93 / 104 /
// 480 ms for a cold proof of the 10^7-gate circuit).
unsigned cpu_budget() {
    static const unsigned budget = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        auto cap = [&](double quota, double period) {
            if (quota > 0 && period > 0) n = std::min(n, std::max(1u, (unsigned)(quota / period + 0.5)));
        };
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            double p = 0;
            if (fscanf(f, "%31s %lf", q, &p) == 2 && strcmp(q, "max") != 0) cap(atof(q), p);
            fclose(f);
        } else {
            double q = 0, p = 0;
            FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
            FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (fq && fp && fscanf(fq, "%lf", &q) == 1 && fscanf(fp, "%lf", &p) == 1) cap(q, p);
            if (fq) fclose(fq);
            if (fp) fclose(fp);
        }
        return n;
    }();
    return budget;
}

int compile_threads() {
    if (const char* e = getenv("RV_COMPILE_THREADS")) return std::max(1, atoi(e));
    // (16: measured on the GPU boxes -- 8 / 16 threads 119 / 82 ms cold, and 16 is their whole CPU quota)
    return (int)std::min(16u, cpu_budget());
}

int compile_ops_par(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out, int force_lazy_k, int n_threads) {
    const auto t_begin = std::chrono::steady_clock::now();
    const bool stats = getenv("RV_COMPILE_STATS") != nullptr;
    auto lap = [&](const char* what) {
        if (stats)
            fprintf(stderr, "[rv compile/par] %-40s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    if (n_ops == 0 || n_ops > (1u << 30) || n_threads < 2) return RV_COMPILE_FALLBACK;
    const int T = n_threads;
    CompileBusy busy_guard;
    Pool pool(T);
    const size_t n_blk = (n_ops + BLK - 1) / BLK;
    out = Compiled();

    // ---------------- pass 1a: kinds, validation, counts per range ----------------
    // ranges are whole blocks; T1 ranges (bounded so that the range-local wire tables stay within ~4 GB of address space)
    Raw<uint8_t> kind(n_ops);
    std::atomic<bool> bail{false};
    std::vector<Ctr> rtot((size_t)T);
    std::vector<uint64_t> rhint2((size_t)T, 0), rhint64((size_t)T, 0);
    auto range_of = [&](int r, int R, size_t& lo, size_t& hi) {
        lo = std::min(n_ops, (n_blk * (size_t)r / (size_t)R) * BLK);
        hi = std::min(n_ops, (n_blk * (size_t)(r + 1) / (size_t)R) * BLK);
    };
    pool.run([&](int t) {
        size_t lo, hi;
        range_of(t, T, lo, hi);
        Ctr c;
        uint64_t h2 = 0, h64 = 0;
        for (size_t i = lo; i < hi; i++) {
            const rv_op& op = ops[i];
            if (op.reserved != 0 || op.domain > RV_DOM_SIZEHINT || op.domain == RV_DOM_B2A || (op.domain <= RV_DOM_Z64 && op.opcode > RV_OP_CONST)) {
                bail.store(true, std::memory_order_relaxed);
                return;
            }
            const uint8_t k = op.domain == RV_DOM_SIZEHINT ? (uint8_t)(RV_DOM_SIZEHINT << 6) : make_kind(op);
            kind[i] = k;
            if (op.domain == RV_DOM_SIZEHINT) {
                h2 = std::max<uint64_t>(h2, op.b);
                h64 = std::max<uint64_t>(h64, op.a);
            } else {
                advance(c, k);
            }
        }
        rtot[(size_t)t] = c;
        rhint2[(size_t)t] = h2;
        rhint64[(size_t)t] = h64;
    });
    if (bail.load()) return RV_COMPILE_FALLBACK;
    std::vector<Ctr> rstart((size_t)T + 1);
    std::vector<uint64_t> rnw2((size_t)T + 1), rnw64((size_t)T + 1);
    rnw2[0] = gf2_wires, rnw64[0] = z64_wires;
    for (int r = 0; r < T; r++) {
        rstart[(size_t)r + 1] = rstart[(size_t)r];
        add_ctr(rstart[(size_t)r + 1], rtot[(size_t)r]);
        rnw2[(size_t)r + 1] = std::max(rnw2[(size_t)r], rhint2[(size_t)r]);
        rnw64[(size_t)r + 1] = std::max(rnw64[(size_t)r], rhint64[(size_t)r]);
    }
    const Ctr tot = rstart[(size_t)T];
    const uint64_t W2 = rnw2[(size_t)T], W64 = rnw64[(size_t)T];
    {
        // anything near the 32-bit limits of the gate records is left to the sequential compiler (it reports RV_E_UNSUPPORTED
        // at the op where a counter overflows); SSA ids double as names of materialised rows and need bit 31 free
        const uint64_t lim = 1u << 30;
        const uint64_t masks_pad = ((uint64_t)tot.masks + 127) / 128 * 128;
        if (tot.ssa > lim || tot.masks > lim || tot.on > lim || tot.ssa64 > lim || tot.masks64 > lim || W2 > (1ull << 31) || W64 > (1ull << 31) ||
            masks_pad / 128 > RV_MAX_CTR_BLOCKS || ((uint64_t)tot.masks64 + 1) / 2 > RV_MAX_CTR_BLOCKS || masks_pad + tot.ssa + 1 > (1ull << 31))
            return RV_COMPILE_FALLBACK;
    }
    lap("pass 1a (kinds, counts)");

    // ---------------- pass 1b: range-local last-writer resolution ----------------
    const uint64_t table_bytes = 4 * (W2 + W64);
    int T1 = T;
    while (T1 > 1 && table_bytes * (uint64_t)T1 > (4ull << 30)) T1--;
    std::vector<Ctr> r1start((size_t)T1 + 1);
    std::vector<uint64_t> r1nw2((size_t)T1 + 1), r1nw64((size_t)T1 + 1);
    if (T1 == T) {
        r1start = rstart, r1nw2 = rnw2, r1nw64 = rnw64;
    } else {
        // fewer, longer ranges: recount their starts from the per-op kinds (cheap next to the tables they avoid)
        r1nw2[0] = gf2_wires, r1nw64[0] = z64_wires;
        for (int r = 0; r < T1; r++) {
            size_t lo, hi;
            range_of(r, T1, lo, hi);
            Ctr c = r1start[(size_t)r];
            uint64_t a2 = r1nw2[(size_t)r], a64 = r1nw64[(size_t)r];
            for (size_t i = lo; i < hi; i++) {
                if (k_dom(kind[i]) == RV_DOM_SIZEHINT) {
                    a2 = std::max<uint64_t>(a2, ops[i].b);
                    a64 = std::max<uint64_t>(a64, ops[i].a);
                } else {
                    advance(c, kind[i]);
                }
            }
            r1start[(size_t)r + 1] = c, r1nw2[(size_t)r + 1] = a2, r1nw64[(size_t)r + 1] = a64;
        }
    }
    Raw<uint32_t> ra(n_ops), rb(n_ops);
    Zeroed<uint32_t> uses(tot.ssa);
    std::vector<Ctr> blocks(n_blk + 1);
    blocks[n_blk] = tot;
    std::vector<std::unique_ptr<Zeroed<uint32_t>>> loc2((size_t)T1), loc64((size_t)T1);
    std::vector<std::vector<uint64_t>> unres2((size_t)T1), unres64((size_t)T1);  // op << 1 | operand slot
    auto use = [&](uint32_t ssa) {
        if (ssa) __atomic_fetch_add(&uses.p[ssa], 1u, __ATOMIC_RELAXED);
    };
    pool.run([&](int t) {
        for (int r = t; r < T1; r += T) {
            size_t lo, hi;
            range_of(r, T1, lo, hi);
            loc2[(size_t)r].reset(new Zeroed<uint32_t>(W2));
            loc64[(size_t)r].reset(new Zeroed<uint32_t>(W64));
            uint32_t* L2 = loc2[(size_t)r]->p;
            uint32_t* L64 = loc64[(size_t)r]->p;
            auto& U2 = unres2[(size_t)r];
            auto& U64 = unres64[(size_t)r];
            Ctr c = r1start[(size_t)r];
            uint64_t nw2 = r1nw2[(size_t)r], nw64 = r1nw64[(size_t)r];
            for (size_t i = lo; i < hi; i++) {
                if ((i & (BLK - 1)) == 0) blocks[i / BLK] = c;
                const uint8_t k = kind[i];
                const uint32_t dom = k_dom(k), opc = k_opc(k);
                const rv_op& op = ops[i];
                if (dom == RV_DOM_SIZEHINT) {
                    nw2 = std::max<uint64_t>(nw2, op.b);
                    nw64 = std::max<uint64_t>(nw64, op.a);
                    continue;
                }
                const int nr = n_reads(opc);
                const bool wr = has_dst(opc);
                const uint64_t nw = dom == RV_DOM_GF2 ? nw2 : nw64;
                if ((wr && op.dst >= nw) || (nr >= 1 && op.a >= nw) || (nr >= 2 && op.b >= nw)) {
                    bail.store(true, std::memory_order_relaxed);  // RV_E_WIRE_OOB: the sequential compiler reports it
                    return;
                }
                uint32_t va = 0, vb = 0;
                if (dom == RV_DOM_GF2) {
                    if (nr >= 1) {
                        va = L2[op.a];
                        if (va) uses.p[va]++; else U2.push_back((uint64_t)i << 1);  // (va is this range's own id: nobody else counts on it in this pass)
                    }
                    if (nr >= 2) {
                        vb = L2[op.b];
                        if (vb) uses.p[vb]++; else U2.push_back((uint64_t)i << 1 | 1);
                    }
                    if (wr) {
                        L2[op.dst] = c.ssa;
                    }
                } else {
                    if (nr >= 1) {
                        va = L64[op.a];
                        if (!va) U64.push_back((uint64_t)i << 1);
                    }
                    if (nr >= 2) {
                        vb = L64[op.b];
                        if (!vb) U64.push_back((uint64_t)i << 1 | 1);
                    }
                    if (wr) {
                        L64[op.dst] = c.ssa64;
                    }
                }
                ra[i] = va;
                rb[i] = vb;
                advance(c, k);
            }
        }
    });
    if (bail.load()) return RV_COMPILE_FALLBACK;
    lap("pass 1b (range-local writers)");

    // ---------------- pass 1c: reads that see a write of an earlier range ----------------
    // All range tables are complete now, so a read of range r looks the wire up in the tables of ranges r-1, r-2, ... -- the
    // first hit is the write it sees; reaching range 0 without one means the wire was never written (SSA 0).  Every read
    // on its own, all in parallel.  The walk stops after PROBE tables; reads that need to look further back ("far" reads:
    // a wire written long ago and read across many ranges) are answered the general way below.
    constexpr int PROBE = 4;
    std::vector<std::vector<uint64_t>> far2((size_t)T1), far64((size_t)T1);
    std::atomic<bool> any_far{false};
    pool.run([&](int t) {
        for (int r = t; r < T1; r += T) {
            if (r == 0) continue;  // (range 0's unresolved reads see never-written wires: SSA 0, already stored)
            for (int z = 0; z < 2; z++) {
                const auto& U = z ? unres64[(size_t)r] : unres2[(size_t)r];
                auto& F = z ? far64[(size_t)r] : far2[(size_t)r];
                for (const uint64_t e : U) {
                    const size_t i = (size_t)(e >> 1);
                    const bool second = e & 1;
                    const uint32_t w = second ? ops[i].b : ops[i].a;
                    uint32_t v = 0;
                    int q = r - 1;
                    for (; q >= 0 && q >= r - PROBE; q--) {
                        v = (z ? loc64[(size_t)q] : loc2[(size_t)q])->p[w];
                        if (v) break;
                    }
                    if (!v && q >= 0) {
                        F.push_back(e);
                        continue;
                    }
                    (second ? rb[i] : ra[i]) = v;
                    if (!z) use(v);
                }
                if (!F.empty()) any_far.store(true, std::memory_order_relaxed);
            }
        }
    });
    if (stats) {
        size_t nf = 0, nu = 0;
        for (int r = 0; r < T1; r++) nf += far2[(size_t)r].size() + far64[(size_t)r].size(), nu += unres2[(size_t)r].size() + unres64[(size_t)r].size();
        fprintf(stderr, "[rv compile/par] %zu reads cross a range boundary, %zu of them look further back than %d ranges\n", nu, nf, PROBE);
    }
    if (any_far.load()) {
        // general way: the ranges' last writes folded into one table in order, the far reads of range r answered when the
        // table holds ranges 0 .. r-1 (each fold and each answer round is parallel)
        Zeroed<uint32_t> cur2(W2), cur64(W64);
        for (int r = 1; r < T1; r++) {
            const uint32_t* P2 = loc2[(size_t)r - 1]->p;
            const uint32_t* P64 = loc64[(size_t)r - 1]->p;
            const auto& U2 = far2[(size_t)r];
            const auto& U64 = far64[(size_t)r];
            size_t flo, fhi;
            range_of(r - 1, T1, flo, fhi);
            pool.run([&](int t) {  // every write of range r - 1 stores the range's LAST value of its wire (the same value from every writer)
                size_t lo, hi;
                Pool::slice(fhi - flo, t, T, lo, hi);
                for (size_t i = flo + lo; i < flo + hi; i++) {
                    const uint32_t dom = k_dom(kind[i]);
                    if (dom > RV_DOM_Z64 || !has_dst(k_opc(kind[i]))) continue;
                    const uint32_t w = ops[i].dst;
                    if (dom == RV_DOM_GF2)
                        __atomic_store_n(&cur2.p[w], P2[w], __ATOMIC_RELAXED);
                    else
                        __atomic_store_n(&cur64.p[w], P64[w], __ATOMIC_RELAXED);
                }
            });
            if (U2.empty() && U64.empty()) continue;
            pool.run([&](int t) {
                size_t lo, hi;
                Pool::slice(U2.size(), t, T, lo, hi);
                for (size_t j = lo; j < hi; j++) {
                    const size_t i = (size_t)(U2[j] >> 1);
                    const bool second = U2[j] & 1;
                    const uint32_t v = cur2[second ? ops[i].b : ops[i].a];
                    (second ? rb[i] : ra[i]) = v;
                    use(v);
                }
                Pool::slice(U64.size(), t, T, lo, hi);
                for (size_t j = lo; j < hi; j++) {
                    const size_t i = (size_t)(U64[j] >> 1);
                    const bool second = U64[j] & 1;
                    (second ? rb[i] : ra[i]) = cur64[second ? ops[i].b : ops[i].a];
                }
            });
        }
    }
    loc2.clear();
    loc64.clear();
    unres2.clear(), unres64.clear();
    lap("pass 1c (cross-range readers, fan-out)");

    // ---------------- pass 2a: linear forms and levels, blocks dealt round-robin, data-flow waits ----------------
    Raw<LinP> lin(tot.ssa);
    Raw<int32_t> lvl_prg((size_t)tot.masks + 2);
    Raw<int32_t> lvl64(tot.ssa64);
    Raw<uint32_t> row64(tot.ssa64);
    Raw<uint32_t> oplvl(n_ops);
    Raw<uint8_t> opcls(n_ops);
    lin[0] = lin_zero();
    static_assert(K == 3, "LinP packs n into two bits");
    lvl64[0] = -1;
    row64[0] = 0;
    int lazy_k = 1;
    bool forced = false;
    if (force_lazy_k) {
        lazy_k = std::min(std::max(force_lazy_k, 1), K);
        forced = true;
    } else if (const char* e = getenv("RV_LAZY_K")) {
        lazy_k = std::min(std::max(atoi(e), 1), K);
        forced = true;
    }
    struct alignas(64) Acc {
        uint32_t max_level = 0;
        bool any = false;
        uint64_t gates2 = 0, n_mat = 0;
    };
    std::vector<Acc> acc((size_t)T);
    std::unique_ptr<std::atomic<uint8_t>[]> blockdone(new std::atomic<uint8_t>[n_blk]);
    std::vector<PaddedU32> cur_block((size_t)T);
    uint32_t n_levels = 0;
    uint64_t n_gates2 = 0, n_mat = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        for (size_t j = 0; j < n_blk; j++) blockdone[j].store(0, std::memory_order_relaxed);
        for (int t = 0; t < T; t++) {
            cur_block[(size_t)t].v.store((uint32_t)t, std::memory_order_relaxed);
            acc[(size_t)t] = Acc();
        }
        const int lk = lazy_k;
        const uint32_t slack = lazy_slack_for(lazy_k, forced);
        pool.run([&](int t) {
            Acc a;
            uint32_t safe2 = 0, safe64 = 0;  // SSA ids below these are final
            auto refresh = [&]() {
                uint32_t mb = 0xFFFFFFFFu;
                for (int u = 0; u < T; u++) mb = std::min(mb, cur_block[(size_t)u].v.load(std::memory_order_acquire));
                if (mb > n_blk) mb = (uint32_t)n_blk;
                safe2 = blocks[mb].ssa;
                safe64 = blocks[mb].ssa64;
            };
            // wait until the block that produces SSA id v (GF(2) if !z) is finished; own = first id of the caller's block
            auto wait_for = [&](uint32_t v, bool z, uint32_t own) {
                if (v >= own || v < (z ? safe64 : safe2)) return;
                size_t lo = 0, hi = n_blk;  // largest j with blocks[j].id <= v
                while (hi - lo > 1) {
                    const size_t mid = (lo + hi) / 2;
                    if ((z ? blocks[mid].ssa64 : blocks[mid].ssa) <= v) lo = mid; else hi = mid;
                }
                uint32_t spins = 0;
                while (!blockdone[lo].load(std::memory_order_acquire)) {
                    RV_PAUSE();
                    if (++spins > 1000) sched_yield();
                }
            };
            // (a materialised row is named by the SSA id of the wire it carries: that wire's record holds its level)
            auto row_level = [&](uint32_t r) -> int32_t { return (r & COMP) ? l_lvl(lin[r & ~COMP]) : lvl_prg[r]; };
            for (size_t blk = (size_t)t; blk < n_blk; blk += (size_t)T) {
                Ctr c = blocks[blk];
                const uint32_t own2 = c.ssa, own64 = c.ssa64;
                const size_t lo = blk * BLK, hi = std::min(n_ops, lo + BLK);
                refresh();  // (once per block; a miss below goes straight to the producing block's flag)
                for (size_t i = lo; i < hi; i++) {
                    const uint8_t k = kind[i];
                    const uint32_t dom = k_dom(k), opc = k_opc(k);
                    uint8_t cls = CLS_NONE;
                    uint32_t lvl = 0;
                    if (i + PF < hi) {  // operands of the op PF ahead: their lines come from other cores
                        if (k_dom(kind[i + PF]) == RV_DOM_GF2) {
                            __builtin_prefetch(&lin[ra[i + PF]]);
                            __builtin_prefetch(&lin[rb[i + PF]]);
                        } else {
                            __builtin_prefetch(&lvl64[ra[i + PF]]);
                            __builtin_prefetch(&lvl64[rb[i + PF]]);
                        }
                    }
                    if (dom == RV_DOM_GF2) {
                        const uint32_t s = c.ssa;
                        const int nr = n_reads(opc);
                        const uint32_t va = ra[i], vb = rb[i];
                        if (nr >= 1) wait_for(va, false, own2);
                        if (nr >= 2) wait_for(vb, false, own2);
                        switch (opc) {
                        case RV_OP_INPUT:
                        case RV_OP_RANDOM:
                            lin[s] = lin_base(c.masks, 0);
                            lvl_prg[c.masks] = 0;
                            cls = 4, lvl = 0;
                            break;
                        case RV_OP_CONST:
                            lin[s] = lin_zero(k_bit(k));
                            break;
                        case RV_OP_ADD:
                        case RV_OP_SUB: {
                            const LinP A = lin[va], B = lin[vb];
                            uint32_t rows[2 * K];
                            const int n = sym_diff(A, B, rows);
                            const uint32_t cc = l_c(A) ^ l_c(B);
                            const uint32_t f = uses[s];
                            const bool lazy = n <= 1 || (n <= lk && (uint64_t)f * (uint32_t)(n - 1) <= (uint64_t)n + slack);
                            if (f == 0) {
                                lin[s] = lin_zero();  // nobody reads it: no gate, no transcript entry, no mask
                                break;
                            }
                            // level of the surviving rows: the operands' levels unless rows cancelled (then row by row)
                            int32_t l = -1;
                            if (n == (int)(l_n(A) + l_n(B)))
                                l = std::max(l_lvl(A), l_lvl(B));
                            else
                                for (int q = 0; q < n; q++) l = std::max(l, row_level(rows[q]));
                            if (lazy) {
                                LinP L = lin_zero();
                                for (int q = 0; q < n; q++) L.b[q] = rows[q];
                                L.meta = l_meta(l, (uint32_t)n, cc);
                                lin[s] = L;
                            } else {
                                lvl = (uint32_t)(l + 1);
                                lin[s] = lin_base(COMP | s, (int32_t)lvl);
                                cls = n == 2 ? 2 : 3;
                                a.n_mat++;
                            }
                            break;
                        }
                        case RV_OP_ADDCONST:
                        case RV_OP_SUBCONST: {
                            LinP L = lin[va];
                            L.meta ^= k_bit(k);
                            lin[s] = L;
                            break;
                        }
                        case RV_OP_MULCONST:
                            lin[s] = k_bit(k) ? lin[va] : lin_zero();
                            break;
                        case RV_OP_MUL: {
                            const LinP A = lin[va], B = lin[vb];
                            lvl = (uint32_t)(std::max(l_lvl(A), l_lvl(B)) + 1);
                            cls = (l_n(A) == 1 && l_n(B) == 1) ? 0 : 1;
                            lvl_prg[c.masks] = (int32_t)lvl;
                            lvl_prg[c.masks + 1] = (int32_t)lvl;
                            lin[s] = lin_base(c.masks + 1, (int32_t)lvl);
                            break;
                        }
                        case RV_OP_ASSERTZERO:
                            lvl = (uint32_t)(l_lvl(lin[va]) + 1);
                            cls = 4;
                            break;
                        }
                        if (cls != CLS_NONE) a.gates2++;
                    } else if (dom == RV_DOM_Z64) {
                        const uint32_t s = c.ssa64;
                        const int nr = n_reads(opc);
                        const uint32_t va = ra[i], vb = rb[i];
                        if (nr >= 1) wait_for(va, true, own64);
                        if (nr >= 2) wait_for(vb, true, own64);
                        cls = CLS_Z64;
                        switch (opc) {
                        case RV_OP_INPUT:
                        case RV_OP_RANDOM:
                            lvl64[s] = 0;
                            row64[s] = G64_MASK_ROW | c.masks64;
                            break;
                        case RV_OP_CONST:
                            lvl64[s] = 0;
                            row64[s] = s;
                            break;
                        case RV_OP_ADD:
                        case RV_OP_SUB:
                            lvl = (uint32_t)(std::max(lvl64[va], lvl64[vb]) + 1);
                            lvl64[s] = (int32_t)lvl;
                            row64[s] = s;
                            break;
                        case RV_OP_MUL:
                            lvl = (uint32_t)(std::max(lvl64[va], lvl64[vb]) + 1);
                            lvl64[s] = (int32_t)lvl;
                            row64[s] = G64_MASK_ROW | (c.masks64 + 1);
                            break;
                        case RV_OP_ADDCONST:
                        case RV_OP_SUBCONST:
                        case RV_OP_MULCONST:
                            lvl = (uint32_t)(lvl64[va] + 1);
                            lvl64[s] = (int32_t)lvl;
                            row64[s] = s;
                            break;
                        case RV_OP_ASSERTZERO:
                            lvl = (uint32_t)(lvl64[va] + 1);
                            break;
                        }
                    }
                    if (cls != CLS_NONE) {
                        a.any = true;
                        if (lvl > a.max_level) a.max_level = lvl;
                    }
                    opcls[i] = cls;
                    oplvl[i] = lvl;
                    advance(c, k);
                }
                blockdone[blk].store(1, std::memory_order_release);
                cur_block[(size_t)t].v.store((uint32_t)std::min<size_t>(blk + (size_t)T, 0xFFFFFFF0u), std::memory_order_release);
            }
            cur_block[(size_t)t].v.store(0xFFFFFFF0u, std::memory_order_release);
            acc[(size_t)t] = a;
        });
        bool any = false;
        uint32_t max_level = 0;
        n_gates2 = 0, n_mat = 0;
        for (const Acc& a : acc) {
            any |= a.any;
            max_level = std::max(max_level, a.max_level);
            n_gates2 += a.gates2;
            n_mat += a.n_mat;
        }
        if (max_level >= MAX_LEVEL) return RV_COMPILE_FALLBACK;  // (levels share a word with the row count in LinP)
        n_levels = any ? max_level + 1 : 0;
        // deep, narrow circuits: keep XORs of up to K rows symbolic (compile.cpp, compile_ops_seq: the same rule)
        const bool deep_narrow = n_levels && lazy_forms_pay(n_levels, n_gates2);
        if (forced || lazy_k != 1 || !deep_narrow) break;
        // deep narrow circuits: the sequential compiler searches over the ways of splitting a sum (compile.cpp, `balance`)
        return RV_COMPILE_FALLBACK;
    }
    lap("pass 2a (linear forms, levels)");

    // ---------------- materialised rows: dense numbers in program order ----------------
    // SSA id s names a materialised row iff lin[s] is the single row COMP | s; its index = 1 + the number of such ids below
    // it: a bit per id and a running count per 64 ids
    const size_t n_words = ((size_t)tot.ssa + 63) / 64;
    Raw<uint64_t> comp_bits(n_words);
    Raw<uint32_t> comp_pre(n_words + 1);
    {
        std::vector<uint64_t> cnt((size_t)T + 1, 0);
        pool.run([&](int t) {
            size_t lo, hi;
            Pool::slice(n_words, t, T, lo, hi);
            uint64_t n = 0;
            for (size_t w = lo; w < hi; w++) {
                uint64_t bits = 0;
                const size_t s0 = w * 64, s1 = std::min<size_t>(s0 + 64, tot.ssa);
                for (size_t q = s0; q < s1; q++)
                    if (l_n(lin[q]) == 1 && lin[q].b[0] == (COMP | (uint32_t)q)) bits |= 1ull << (q - s0);
                comp_bits[w] = bits;
                comp_pre[w] = (uint32_t)n;  // (relative to the slice start; made absolute below)
                n += (uint64_t)__builtin_popcountll(bits);
            }
            cnt[(size_t)t + 1] = n;
        });
        for (int t = 0; t < T; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
        pool.run([&](int t) {
            size_t lo, hi;
            Pool::slice(n_words, t, T, lo, hi);
            const uint32_t base = (uint32_t)cnt[(size_t)t];
            for (size_t w = lo; w < hi; w++) comp_pre[w] += base;
        });
        if (cnt[(size_t)T] != n_mat) return RV_COMPILE_FALLBACK;  // (cannot happen)
    }
    const uint64_t n_comp = 1 + n_mat;
    out.n_masks = tot.masks, out.n_on = tot.on, out.n_pre = tot.pre, out.n_in = tot.in, out.n_rec = tot.rec;
    out.n_masks64 = tot.masks64, out.on_words64 = tot.onw64, out.pre_words64 = tot.prew64, out.n_in64 = tot.in64, out.n_rec64 = tot.rec64,
    out.n_corr64 = tot.corr64;
    out.n_ssa = tot.ssa;
    out.n_random_or_recon = tot.gf2_linear_random;
    out.n_ssa64 = tot.ssa64;
    out.n_masks_pad = (out.n_masks + 127) / 128 * 128;
    out.row_prg_base = 0;
    out.zero_row = out.n_masks_pad;
    out.n_rows = out.n_masks_pad + n_comp;
    const uint32_t comp_base = (uint32_t)out.n_masks_pad;
    auto fix = [&](uint32_t r) -> uint32_t {
        if (!(r & COMP)) return r;
        const uint32_t q = r & ~COMP;
        if (!q) return comp_base;  // the all-zero row
        return comp_base + 1 + comp_pre[q >> 6] + (uint32_t)__builtin_popcountll(comp_bits[q >> 6] & ((1ull << (q & 63)) - 1));
    };

    // ---------------- stable counting sort by (level, class): histograms per contiguous range ----------------
    const uint64_t bins2 = (uint64_t)n_levels * 5, bins64 = n_levels;
    int TH = T;
    while (TH > 1 && (bins2 + bins64 + n_levels) * (uint64_t)TH > (48ull << 20)) TH--;
    std::vector<std::vector<uint32_t>> h2((size_t)TH), h64((size_t)TH), need((size_t)TH);
    auto hrun = [&](const std::function<void(int)>& f) {  // f(range) for the TH ranges
        pool.run([&](int t) {
            for (int r = t; r < TH; r += T) f(r);
        });
    };
    hrun([&](int r) {
        h2[(size_t)r].assign(bins2 + 1, 0);
        h64[(size_t)r].assign(bins64 + 1, 0);
        need[(size_t)r].assign(n_levels, 0);
        size_t lo, hi;
        range_of(r, TH, lo, hi);
        uint32_t* H2 = h2[(size_t)r].data();
        uint32_t* H64 = h64[(size_t)r].data();
        for (size_t i = lo; i < hi; i++) {
            const uint8_t cls = opcls[i];
            if (cls == CLS_NONE) continue;
            if (cls == CLS_Z64) H64[oplvl[i]]++; else H2[(size_t)oplvl[i] * 5 + cls]++;
        }
    });
    std::vector<uint32_t> start2(bins2 + 1, 0), start64(bins64 + 1, 0);
    pool.run([&](int t) {
        size_t lo, hi;
        Pool::slice(bins2, t, T, lo, hi);
        for (size_t b = lo; b < hi; b++) {
            uint32_t s = 0;
            for (int r = 0; r < TH; r++) s += h2[(size_t)r][b];
            start2[b + 1] = s;
        }
        Pool::slice(bins64, t, T, lo, hi);
        for (size_t b = lo; b < hi; b++) {
            uint32_t s = 0;
            for (int r = 0; r < TH; r++) s += h64[(size_t)r][b];
            start64[b + 1] = s;
        }
    });
    for (size_t b = 0; b < bins2; b++) start2[b + 1] += start2[b];
    for (size_t b = 0; b < bins64; b++) start64[b + 1] += start64[b];
    pool.run([&](int t) {
        size_t lo, hi;
        Pool::slice(bins2, t, T, lo, hi);
        for (size_t b = lo; b < hi; b++) {
            uint32_t run = start2[b];
            for (int r = 0; r < TH; r++) {
                const uint32_t n = h2[(size_t)r][b];
                h2[(size_t)r][b] = run;
                run += n;
            }
        }
        Pool::slice(bins64, t, T, lo, hi);
        for (size_t b = lo; b < hi; b++) {
            uint32_t run = start64[b];
            for (int r = 0; r < TH; r++) {
                const uint32_t n = h64[(size_t)r][b];
                h64[(size_t)r][b] = run;
                run += n;
            }
        }
    });
    if (start2[bins2] != n_gates2 || start64[bins64] != tot.gates64) return RV_COMPILE_FALLBACK;  // (cannot happen)
    out.level_start.assign((size_t)n_levels + 1, 0);
    out.level_range.assign(n_levels, LevelRange{});
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t* e = &start2[(size_t)l * 5];
        out.level_start[l] = e[0];
        out.level_range[l] = LevelRange{e[0], e[1], e[2], e[3], e[4], e[5]};
    }
    out.level_start[n_levels] = (uint32_t)n_gates2;
    out.level_start64.assign(start64.begin(), start64.end());
    if (out.level_start64.size() != (size_t)n_levels + 1) out.level_start64.resize((size_t)n_levels + 1, 0);
    lap("histograms, level table");

    // ---------------- pass 2b: every gate built once, at its final place ----------------
    out.gates.resize(n_gates2);
    out.gates64.resize(tot.gates64);
    out.rec_rows.resize(tot.rec);
    out.in_rows.resize(tot.in);
    out.rec_offs64.resize(tot.rec64);
    out.in_offs64.resize(tot.in64);
    Raw<uint32_t> row_level(std::max<uint32_t>(tot.on, 1));
    struct alignas(64) Info {
        uint64_t operand_rows = 0, rows_written = 0;
    };
    std::vector<Info> infos((size_t)TH);
    hrun([&](int r) {
        size_t lo, hi;
        range_of(r, TH, lo, hi);
        if (lo >= hi) return;
        uint32_t* P2 = h2[(size_t)r].data();
        uint32_t* P64 = h64[(size_t)r].data();
        uint32_t* N = need[(size_t)r].data();
        Info inf;
        Ctr c = blocks[lo / BLK];  // (range starts are block starts)
        for (size_t i = lo; i < hi; i++) {
            const uint8_t k = kind[i];
            const uint8_t cls = opcls[i];
            if (i + PF < hi && opcls[i + PF] < CLS_Z64) {
                __builtin_prefetch(&lin[ra[i + PF]]);
                __builtin_prefetch(&lin[rb[i + PF]]);
            }
            if (cls != CLS_NONE && cls != CLS_Z64) {
                const uint32_t opc = k_opc(k), lvl = oplvl[i];
                Gate g{};
                auto fill = [&](const LinP& A, const LinP* B) {
                    const uint32_t an = l_n(A), bn = B ? l_n(*B) : 0;
                    for (uint32_t q = 0; q < (uint32_t)K; q++) {
                        g.a[q] = fix(q < an ? A.b[q] : ZERO_ROW);
                        g.b[q] = fix(q < bn ? B->b[q] : ZERO_ROW);
                    }
                    g.op |= an << 8 | bn << 12 | l_c(A) << 16 | (B ? l_c(*B) : 0) << 17;
                };
                const LinP Z = lin_zero();
                uint32_t last = 0;
                bool masks = false;
                switch (opc) {
                case RV_OP_INPUT:
                    g.op = G_INPUT;
                    fill(Z, nullptr);
                    g.m = c.masks, g.dst = c.masks, g.eo = c.on, g.x = c.in;
                    out.in_rows[c.in] = c.on;
                    row_level[c.on] = lvl;
                    last = c.masks, masks = true;
                    break;
                case RV_OP_RANDOM:
                    g.op = G_RANDOM;
                    fill(Z, nullptr);
                    g.m = c.masks, g.dst = c.masks;
                    last = c.masks, masks = true;
                    break;
                case RV_OP_MUL:
                    g.op = G_MUL;
                    fill(lin[ra[i]], &lin[rb[i]]);
                    g.m = c.masks, g.eo = c.on, g.ep = c.pre, g.x = c.rec, g.dst = c.masks + 1;
                    out.rec_rows[c.rec] = c.on;
                    row_level[c.on] = lvl;
                    last = c.masks + 1, masks = true;
                    inf.operand_rows += g_na(g) + g_nb(g);
                    break;
                case RV_OP_ASSERTZERO:
                    g.op = G_ASSERT;
                    fill(lin[ra[i]], nullptr);
                    g.eo = c.on, g.x = c.rec;
                    out.rec_rows[c.rec] = c.on;
                    row_level[c.on] = lvl;
                    inf.operand_rows += g_na(g);
                    break;
                default: {  // Add / Sub materialised as G_XORK
                    uint32_t rows[2 * K];
                    const LinP &A = lin[ra[i]], &B = lin[rb[i]];
                    const int n = sym_diff(A, B, rows);
                    const int na = std::min(n, K);
                    g.op = G_XORK;
                    for (int q = 0; q < K; q++) {
                        g.a[q] = fix(q < na ? rows[q] : ZERO_ROW);
                        g.b[q] = fix((K + q < n) ? rows[K + q] : ZERO_ROW);
                    }
                    g.op |= (uint32_t)na << 8 | (uint32_t)(n - na) << 12 | (l_c(A) ^ l_c(B)) << 16;
                    g.dst = fix(COMP | c.ssa);
                    inf.operand_rows += (uint32_t)n;
                    inf.rows_written++;
                    break;
                }
                }
                if (masks) N[lvl] = std::max(N[lvl], last / 128 + 1);
                out.gates[P2[(size_t)lvl * 5 + cls]++] = g;
            } else if (cls == CLS_Z64) {
                const uint32_t opc = k_opc(k), lvl = oplvl[i];
                const rv_op& op = ops[i];
                Gate64 g{};
                g.imm = op.imm;
                switch (opc) {
                case RV_OP_INPUT:
                    g.op = G64_INPUT, g.m = c.masks64, g.eo = c.onw64, g.x = c.in64, g.dst = c.ssa64;
                    out.in_offs64[c.in64] = c.onw64;
                    break;
                case RV_OP_RANDOM: g.op = G64_RANDOM, g.m = c.masks64, g.dst = c.ssa64; break;
                case RV_OP_CONST: g.op = G64_CONST, g.dst = c.ssa64; break;
                case RV_OP_ADD:
                case RV_OP_SUB:
                    g.op = opc == RV_OP_ADD ? G64_ADD : G64_SUB, g.a = ra[i], g.b = rb[i], g.dst = c.ssa64;
                    break;
                case RV_OP_MUL:
                    g.op = G64_MUL, g.a = ra[i], g.b = rb[i], g.m = c.masks64, g.ep = c.prew64, g.xc = c.corr64, g.eo = c.onw64, g.x = c.rec64,
                    g.dst = c.ssa64;
                    out.rec_offs64[c.rec64] = c.onw64;
                    break;
                case RV_OP_ADDCONST:
                case RV_OP_SUBCONST:
                case RV_OP_MULCONST:
                    g.op = opc == RV_OP_ADDCONST ? G64_ADDC : opc == RV_OP_SUBCONST ? G64_SUBC : G64_MULC, g.a = ra[i], g.dst = c.ssa64;
                    break;
                case RV_OP_ASSERTZERO:
                    g.op = G64_ASSERT, g.a = ra[i], g.eo = c.onw64, g.x = c.rec64;
                    out.rec_offs64[c.rec64] = c.onw64;
                    break;
                }
                g.am = row64[g.a];
                g.bm = row64[g.b];
                out.gates64[P64[lvl]++] = g;
            }
            advance(c, k);
        }
        infos[(size_t)r] = inf;
    });
    lap("pass 2b (gates at their sorted places)");

    // ---------------- pipelining tables (as compile_ops_seq) ----------------
    out.level_need_blocks.assign(n_levels, 0);
    out.level_done_on.assign(n_levels, 0);
    {
        uint32_t nb = 0;
        for (uint32_t l = 0; l < n_levels; l++) {
            for (int r = 0; r < TH; r++) nb = std::max(nb, need[(size_t)r][l]);
            out.level_need_blocks[l] = nb;
        }
        uint64_t e = 0;
        uint32_t run = 0;
        for (uint32_t l = 0; l < n_levels; l++) {
            while (e < out.n_on && std::max(run, row_level[e]) <= l) {
                run = std::max(run, row_level[e]);
                e++;
            }
            out.level_done_on[l] = (uint32_t)e;
        }
    }
    rv_circuit_info& info = out.info;
    info = rv_circuit_info{};
    info.n_ops = n_ops;
    info.gf2_inputs = tot.gf2_inputs, info.gf2_muls = tot.gf2_muls, info.gf2_asserts = tot.gf2_asserts;
    info.gf2_linear = (uint64_t)tot.gf2_linear_random + n_mat;
    info.gf2_masks = out.n_masks;
    info.z64_inputs = tot.z64_inputs, info.z64_muls = tot.z64_muls, info.z64_asserts = tot.z64_asserts, info.z64_linear = tot.z64_linear;
    info.z64_masks = out.n_masks64;
    info.levels = n_levels;
    for (const Info& f : infos) {
        info.gf2_operand_rows += f.operand_rows;
        info.gf2_rows_written += f.rows_written;
    }
    lap("tables done");
    return RV_OK;
}

// field-by-field comparison of two compiled circuits (test hook: the parallel compiler against the sequential one);
// 0 = identical, otherwise a small number naming the first field that differs
int compiled_diff(const Compiled& a, const Compiled& b) {
    auto veq = [](const auto& x, const auto& y) {
        return x.size() == y.size() && (x.empty() || memcmp(x.data(), y.data(), x.size() * sizeof(x[0])) == 0);
    };
    if (!veq(a.level_start, b.level_start)) return 2;
    if (!veq(a.level_range, b.level_range)) return 3;
    if (!veq(a.gates, b.gates)) return 1;
    if (!veq(a.level_need_blocks, b.level_need_blocks)) return 4;
    if (!veq(a.level_done_on, b.level_done_on)) return 5;
    if (!veq(a.rec_rows, b.rec_rows)) return 6;
    if (!veq(a.in_rows, b.in_rows)) return 7;
    if (a.n_ssa != b.n_ssa || a.n_masks_pad != b.n_masks_pad || a.n_rows != b.n_rows || a.n_masks != b.n_masks || a.n_on != b.n_on ||
        a.n_pre != b.n_pre || a.n_in != b.n_in || a.n_rec != b.n_rec || a.n_random_or_recon != b.n_random_or_recon)
        return 8;
    if (!veq(a.level_start64, b.level_start64)) return 10;
    if (!veq(a.gates64, b.gates64)) return 9;
    if (!veq(a.rec_offs64, b.rec_offs64)) return 11;
    if (!veq(a.in_offs64, b.in_offs64)) return 12;
    if (a.n_ssa64 != b.n_ssa64 || a.n_masks64 != b.n_masks64 || a.on_words64 != b.on_words64 || a.pre_words64 != b.pre_words64 ||
        a.n_in64 != b.n_in64 || a.n_rec64 != b.n_rec64 || a.n_corr64 != b.n_corr64 || a.row_prg_base != b.row_prg_base || a.zero_row != b.zero_row)
        return 13;
    if (memcmp(&a.info, &b.info, sizeof a.info) != 0) return 14;
    return 0;
}

}  // namespace rv
