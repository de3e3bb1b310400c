// C-ABI implementation (include/reverie_amd.h): contexts, HBM arenas, the prove / verify
// orchestration of /root/reference/src/proof/mod.rs:119-307 and the host-side Fiat-Shamir
// pieces (combine_hashes :102-108, challenge_to_opening :74-83, bincode layout :40-66).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#include <time.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "b3.h"
#include "compile.h"
#include "flat.h"
#include "internal.h"
#include "launch.h"
#include "repprog.h"
#include "ldsrun.h"

using namespace rv;

// ------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    return RV_E_DEVICE;
}
#define HIPCHK(x)                                                     \
    do {                                                              \
        hipError_t e_ = (x);                                          \
        if (e_ != hipSuccess) return hip_fail(e_, #x, __FILE__, __LINE__); \
    } while (0)

extern "C" const char* rv_last_error(void) { return g_last_error.c_str(); }
// 1: an experiment build (make EXTRA=-DRV_EXPERIMENTS: the rep-sliced path, the flat / split / chained schedules, the persistent level
// kernels, RV_EARLY_REC); 0: the library build() makes, where those knobs do nothing
extern "C" int rv_hook_experiments(void) {
#ifdef RV_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
extern "C" uint32_t rv_abi_version(void) { return 7; }  // 3: verification strict by default (RV_VERIFY_REFERENCE_COMPAT), rv_bristol_parse takes n_expected,
                                                        //    streaming prover, rv_prove_multi, reconstruct hooks
                                                        // 4: rv_circuit_compile_ex (a pure addition)
                                                        // 5: rv_prove_ops / rv_verify_ops, rv_hook_compile_compare (pure additions)
                                                        // 6: rv_circuit_info grew by early_staging_bytes (callers must pass the larger struct)
                                                        // 7: rv_circuit_info is its ABI-5 self again (a struct without a size field must not grow: a caller built
                                                        //    against the older header would have had 8 bytes written past its buffer); the value has a getter of
                                                        //    its own, rv_circuit_early_staging_bytes

extern "C" const char* rv_strerror(int code) {
    switch (code) {
    case RV_OK: return "ok";
    case RV_E_WITNESS_INVALID: return "witness is invalid (an AssertZero wire is not zero)";
    case RV_E_WITNESS_SHORT: return "witness is too short";
    case RV_E_WIRE_OOB: return "wire index out of range";
    case RV_E_PROOF_MALFORMED: return "proof bytes are malformed";
    case RV_E_BAD_OP: return "unknown operation";
    case RV_E_NOMEM: return "out of memory";
    case RV_E_DEVICE: return "GPU/HIP error (no usable gfx950 device?)";
    case RV_E_UNSUPPORTED: return "unsupported";
    case RV_E_ARG: return "bad argument";
    }
    return "unknown error";
}

// ------------------------------------------------------------------------------------
// Helper threads of the early-corrections path (rv_prove_impl): a handful of sleeping threads that copy the opened
// repetitions' corrections from the staging buffer into the proof while the GPU extracts the other half.
// ------------------------------------------------------------------------------------
struct HelperPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t gen = 0;
    bool stop = false;
    const std::function<void(int)>* job = nullptr;  // null again once the caller's own share is done: a helper that wakes up after that stays out
    std::atomic<int> running{0};
    explicit HelperPool(int n) {
        for (int i = 1; i < n; i++)
            th.emplace_back([this, i] {
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void(int)>* f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = job;
                        if (f) running.fetch_add(1, std::memory_order_relaxed);  // (under the lock: run() cannot miss it)
                    }
                    if (f) {
                        (*f)(i);
                        running.fetch_sub(1, std::memory_order_release);
                    }
                }
            });
    }
    int size() const { return (int)th.size() + 1; }
    // f(0) on the calling thread, f(1 ..) on the helpers that wake up in time; f must hand out its work dynamically (whoever
    // shows up takes the next piece) and return when nothing is left to take.  Returns when every thread that entered f has left it.
    void run(const std::function<void(int)>& f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &f;
            gen++;
        }
        cv.notify_all();
        f(0);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = nullptr;
        }
        for (uint32_t spins = 0; running.load(std::memory_order_acquire) > 0; spins++) {
            __builtin_ia32_pause();
            if (spins > 4000) std::this_thread::yield();  // (a helper descheduled in the middle of its last piece)
        }
    }
    ~HelperPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

// ------------------------------------------------------------------------------------
// context: one device, one stream, a caching arena (hipMalloc of GB-sized buffers costs
// milliseconds; proofs over the same circuit reuse the same sizes)
// ------------------------------------------------------------------------------------
struct rv_ctx {
    int device = 0;
    size_t lds_bytes = 0;  // hipDeviceAttributeMaxSharedMemoryPerBlock
    hipStream_t stream = nullptr;   // setup, AES masks, hashing, openings (VALU-heavy work)
    hipStream_t stream2 = nullptr;  // side stream: the early-corrections copies, the verifier's proof copy and unpack kernels
    hipStream_t stream3 = nullptr;  // the flat schedule's cleartext pass (k_clear), beside the mask generator
    hipStream_t stream_x = nullptr; // the flat schedule's XOR rows, running ahead of the Mul launches on `stream`
    hipStream_t stream_m = nullptr; // RV_OVERLAP: the lane-distributed mask generator, beside the interpreter's level launches (made on first use)
    hipEvent_t clear_a = nullptr, clear_b = nullptr;  // profiling: around k_clear on stream3 (rv_profile slot RV_PH_CLEAR)
    bool clear_timed = false;
    std::vector<rv_ctx*> workers;            // rv_prove_batch on large circuits: one worker context per host thread
    // Small proofs (AES-128: 99 KB) leave through this page-locked, device-mapped buffer: the opening kernels write into it
    // and a one-lane kernel adds the error word, instead of two copy-engine operations of ~25 us each behind them
    static constexpr size_t STAGE_BYTES = (size_t)1 << 20;
    uint8_t* h_stage = nullptr;
    // ... and the seeds and the GF(2) witness of a whole proof enter through this one (one copy instead of two pageable ones)
    static constexpr size_t IN_STAGE_BYTES = (size_t)1 << 20;
    uint8_t* h_in = nullptr;
    // page-locked staging of a compiled circuit's arrays on their way to HBM (circuit_upload; grown on demand up to
    // UP_STAGE_MAX): a pageable hipMemcpyAsync pins and unpins the source pages inside the call -- 3.6 ms for a 35 MB
    // chunk of the streaming prover at best, 9 - 22 ms when the kernel's address-space lock is busy
    static constexpr size_t UP_STAGE_MAX = (size_t)192 << 20;
    uint8_t* h_up = nullptr;
    size_t h_up_cap = 0;
    // ... and the streaming feeds' ring of page-locked slots: a worker thread copies a compiled piece's arrays into a slot
    // ahead of the main thread (circuit_stage), which then only issues the copies (stream.inc)
    std::vector<uint8_t*> h_ring;
    size_t h_ring_cap = 0;
    // early corrections (rv_prove_impl): page-locked staging for EVERY repetition's corrections vector, the mapped
    // mailbox the challenge arrives in ([0] = sequence number, from word 16 on the data), the helper threads
    uint8_t* h_ec = nullptr;
    size_t h_ec_cap = 0;
    uint8_t* d_ec = nullptr;  // the device side of the staging (outside the arena: the proof's other buffers keep the places they have without it)
    size_t d_ec_cap = 0;
    uint32_t* h_fs = nullptr;
    uint32_t fs_seq = 0;
    // ... and the staging of the opened repetitions' broadcast-bit vectors on their way out (RecStage below): [40][pitch] on the
    // device and page-locked on the host
    uint8_t* d_rs = nullptr;
    uint8_t* h_rs = nullptr;
    size_t rs_cap = 0;
    HelperPool* ec_pool = nullptr;
    double ec_wait_us[17] = {0};  // running averages of the early-corrections waits (per chunk stamp, [16] the challenge): mailbox_wait
    // rv_prove_ops / rv_verify_ops: the circuits compiled from raw op lists, kept by content (ops_cache_get): the reference's
    // Proof::new takes the op list at every call (proof/mod.rs:119-124), and a caller that proves one circuit again and again through
    // that signature should pay the 70 - 90 ms host compile once, not per proof
    struct OpsEntry {
        uint64_t h[2];
        size_t n_ops, z64_wires, gf2_wires;
        uint32_t flags;
        rv_circuit* c;
        uint64_t stamp;
    };
    std::vector<OpsEntry> ops_cache;
    uint64_t ops_clock = 0;
    std::vector<hipEvent_t> sync_pool;
    hipEvent_t get_sync_event() {
        if (!sync_pool.empty()) {
            hipEvent_t e = sync_pool.back();
            sync_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        return e;
    }
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live;
    size_t cached_bytes = 0;
    // event timing (rv_ctx_profile)
    bool profiling = false;
    rv_profile prof{};
    std::vector<hipEvent_t> ev_pool;
    struct Mark {
        int phase;
        hipEvent_t a, b;
        uint64_t launches;
    };
    std::vector<Mark> marks;
    std::map<hipEvent_t, int> ev_refs;  // events of the marks not yet collected
    int cur_phase = -1;
    hipEvent_t cur_start = nullptr;
    hipStream_t cur_stream = nullptr;
    uint64_t cur_launches = 0;

    hipEvent_t get_event() {
        if (!ev_pool.empty()) {
            hipEvent_t e = ev_pool.back();
            ev_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    // phase(p): closes the running phase and opens p (p < 0: just close); both events of a
    // phase are recorded on the stream its kernels run on
    void phase(int p, hipStream_t st = nullptr) {
        // (not while a batch is being recorded: the launches happen later, and a thousand event markers queued between
        // two phases of a batch kept the GPU idle for 7 ms)
        if (!profiling || g_recorder) return;
        // ONE marker per boundary: the event that ends a phase also starts the next one on the same stream (every marker
        // in the queue costs the proof ~5 us of idle GPU, and the bench's timed region runs with the phases on)
        hipStream_t next = p >= 0 ? (st ? st : stream) : nullptr;
        hipEvent_t e = nullptr;
        if (cur_phase >= 0) {
            e = get_event();
            (void)hipEventRecord(e, cur_stream);
            marks.push_back({cur_phase, cur_start, e, cur_launches});
            ++ev_refs[e];
        }
        cur_phase = p;
        cur_launches = 0;
        if (p >= 0) {
            if (e && next == cur_stream) {
                cur_start = e;
            } else {
                cur_start = get_event();
                (void)hipEventRecord(cur_start, next);
            }
            ++ev_refs[cur_start];
            cur_stream = next;
        }
    }
    void count(uint64_t n = 1) { cur_launches += n; }
    // call after a stream sync
    void collect() {
        if (!profiling) return;
        phase(-1);
        for (auto& m : marks) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, m.a, m.b) == hipSuccess) prof.ms[m.phase] += ms;
            prof.launches[m.phase] += m.launches;
        }
        if (clear_timed) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, clear_a, clear_b) == hipSuccess) prof.ms[RV_PH_CLEAR] += ms;
            prof.launches[RV_PH_CLEAR]++;
            clear_timed = false;
        }
        for (auto& kv : ev_refs) ev_pool.push_back(kv.first);  // (an event may be the end of one mark and the start of the next)
        ev_refs.clear();
        marks.clear();
    }

    int alloc(size_t bytes, void** out) {
        if (bytes == 0) bytes = 256;
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 4 + 4096) {
            *out = it->second;
            live[*out] = it->first;
            cached_bytes -= it->first;
            free_blocks.erase(it);
            return RV_OK;
        }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) {
            trim();
            e = hipMalloc(out, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            g_last_error = "hipMalloc failed";
            return RV_E_NOMEM;
        }
        live[*out] = bytes;
        return RV_OK;
    }
    void release(void* p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.emplace(it->second, p);
        cached_bytes += it->second;
        live.erase(it);
    }
    void trim() {
        for (auto& kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
        cached_bytes = 0;
    }
};

template <class T>
static int dalloc(rv_ctx* ctx, size_t count, T** out) {
    void* p = nullptr;
    int rc = ctx->alloc(count * sizeof(T), &p);
    *out = (T*)p;
    return rc;
}

// rv_prove_batch's worker contexts: worker k's main stream gets stream priority level k mod (levels of the device).  HIP deals the
// streams of ONE priority to four hardware queues in creation order, so which of a process's streams share a queue depends on how
// many it happened to create before -- and when the workers' main streams fell on one queue, their proofs in flight ran one after the
// other (a bench run with 10.2 ms per proof instead of 4.9 - 5.1).  Streams of different priorities never share a queue.
static int ctx_create_impl(int device_ordinal, rv_ctx** out, int main_prio_level /* -1: default priority */);
extern "C" int rv_ctx_create(int device_ordinal, rv_ctx** out) { return ctx_create_impl(device_ordinal, out, -1); }

static int ctx_create_impl(int device_ordinal, rv_ctx** out, int main_prio_level) {
    if (!out) return RV_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        g_last_error = "no HIP device visible: the reverie_amd product path needs an MI355X (gfx950); there is no CPU fallback";
        return RV_E_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) return RV_E_ARG;
    HIPCHK(hipSetDevice(device_ordinal));
    rv_ctx* c = new rv_ctx();
    c->device = device_ordinal;
    {
        // LDS a workgroup may have (160 KiB on gfx950): the LDS-run and rep-sliced paths size their wire stores by it and are
        // left out when it is too small for them (a build for another part must fall back to the row interpreter, not fail at launch)
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_ordinal) != hipSuccess) {
            (void)hipGetLastError();
            lds = 64 * 1024;
        }
        c->lds_bytes = (size_t)std::max(lds, 0);
        set_device_lds_limit(c->lds_bytes);
    }
    // The runtime multiplexes the streams of one priority over four hardware queues (a fifth stream would share the first one's),
    // and a queue that issues short kernels back to back keeps the dispatcher from a queue of the same or a lower priority.  The
    // main stream can therefore get the high priority (RV_MAIN_PRIO=1; default: all streams alike): its long kernels go out the moment their
    // dependencies are met, and the flat schedule's short XOR launches (stream_x) fill in beside them.  (Measured: no gain; off.)
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0, (void)hipGetLastError();
    static const bool main_prio = getenv("RV_MAIN_PRIO") && atoi(getenv("RV_MAIN_PRIO")) != 0;
    hipError_t se;
    if (main_prio_level >= 0 && prio_lo > prio_hi) {
        const int levels = prio_lo - prio_hi + 1;  // (numerically lower = higher priority)
        se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi + main_prio_level % levels);
    } else {
        se = main_prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    // (stream3 / stream_x -- the flat and split schedules' side streams -- are made by ctx_side_streams() when such a schedule first
    // runs: streams are dealt to the four hardware queues in creation order, so two idle ones per context put the main streams of
    // rv_prove_batch's worker contexts all on ONE queue and its proofs in flight ran one after the other: 4.9 -> 6.2 ms per proof)
    if (se != hipSuccess) {
        delete c;
        return hip_fail(se, "hipStreamCreate", __FILE__, __LINE__);
    }
    *out = c;
    return RV_OK;
}

#ifdef RV_EXPERIMENTS
// the side streams of the flat / split prover schedules (RV_FLAT != 0), on first use
static int ctx_side_streams(rv_ctx* c) {
    if (c->stream3 && c->stream_x) return RV_OK;
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0, (void)hipGetLastError();
    hipError_t se = hipSuccess;
    if (!c->stream3) se = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    // (RV_X_PRIO=1: the chain stream at the high priority -- its short dependent launches then win the dispatcher whenever
    // wavefront slots are free)
    static const bool x_prio = getenv("RV_X_PRIO") && atoi(getenv("RV_X_PRIO")) != 0;
    if (se == hipSuccess && !c->stream_x)
        se = x_prio ? hipStreamCreateWithPriority(&c->stream_x, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&c->stream_x, hipStreamNonBlocking);
    return se == hipSuccess ? RV_OK : hip_fail(se, "hipStreamCreate", __FILE__, __LINE__);
}
#endif

static void pinned_pool_trim();  // idle page-locked output buffers (defined with the pool below)

extern "C" void rv_ctx_destroy(rv_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    if (ctx->stream_x) (void)hipStreamSynchronize(ctx->stream_x);
    if (ctx->stream_m) (void)hipStreamSynchronize(ctx->stream_m);
    for (auto& e : ctx->ops_cache) rv_circuit_destroy(e.c);
    ctx->ops_cache.clear();
    ctx->trim();
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    for (uint8_t* p : ctx->h_ring) (void)hipHostFree(p);
    if (ctx->h_ec) (void)hipHostFree(ctx->h_ec);
    if (ctx->d_ec) (void)hipFree(ctx->d_ec);
    if (ctx->h_fs) (void)hipHostFree(ctx->h_fs);
    if (ctx->h_rs) (void)hipHostFree(ctx->h_rs);
    if (ctx->d_rs) (void)hipFree(ctx->d_rs);
    delete ctx->ec_pool;
    for (rv_ctx* w : ctx->workers) rv_ctx_destroy(w);
    (void)hipStreamDestroy(ctx->stream);
    (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream_x) (void)hipStreamDestroy(ctx->stream_x);
    if (ctx->stream_m) (void)hipStreamDestroy(ctx->stream_m);
    if (ctx->clear_a) (void)hipEventDestroy(ctx->clear_a);
    if (ctx->clear_b) (void)hipEventDestroy(ctx->clear_b);
    delete ctx;
    pinned_pool_trim();
}

extern "C" int rv_ctx_sync(rv_ctx* ctx) {
    if (!ctx) return RV_E_ARG;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RV_OK;
}

// Large outputs (proofs of big circuits: 50-640 MB) are returned in page-locked host memory from a small
// process-wide pool: the device-to-host copy then runs at PCIe rate instead of through the runtime's pageable
// staging path (~3x slower), and rv_free hands the buffer back for the next proof instead of unpinning it.
namespace {
struct PinnedPool {
    struct Buf {
        void* p;
        size_t cap;
        bool used;
        size_t shares = 0;  // > 0: the buffer was handed out as that many slices (rv_prove_batch); put() of a slice drops one
    };
    std::mutex mu;
    std::vector<Buf> bufs;
    static constexpr size_t MIN_BYTES = 1u << 20;  // below this plain malloc is as fast
    static constexpr size_t KEEP_FREE = 3;         // idle buffers kept for reuse
    void* get(size_t n) {
        if (n < MIN_BYTES) return nullptr;
        std::lock_guard<std::mutex> g(mu);
        int best = -1;
        for (size_t i = 0; i < bufs.size(); i++)
            if (!bufs[i].used && bufs[i].cap >= n && (best < 0 || bufs[i].cap < bufs[(size_t)best].cap)) best = (int)i;
        if (best >= 0) {
            bufs[(size_t)best].used = true;
            return bufs[(size_t)best].p;
        }
        void* p = nullptr;
        const size_t cap = (n + (n >> 3) + 0xFFFFF) & ~(size_t)0xFFFFF;  // 12 % headroom, whole MiB
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;  // caller falls back to malloc
        }
        bufs.push_back(Buf{p, cap, true});
        return p;
    }
    // the buffer `base` now belongs to `n` slices, each released on its own through put(any address inside)
    void share(void* base, size_t n) {
        std::lock_guard<std::mutex> g(mu);
        for (Buf& b : bufs)
            if (b.p == base) b.shares = n;
    }
    bool put(void* p) {
        std::lock_guard<std::mutex> g(mu);
        bool found = false;
        for (Buf& b : bufs) {
            if (b.shares) {
                if ((const uint8_t*)p >= (const uint8_t*)b.p && (const uint8_t*)p < (const uint8_t*)b.p + b.cap) {
                    found = true;
                    if (--b.shares == 0) b.used = false;
                    else return true;
                }
            } else if (b.p == p) {
                b.used = false;
                found = true;
            }
        }
        if (!found) return false;
        size_t idle = 0;
        for (const Buf& b : bufs) idle += !b.used;
        for (size_t i = 0; i < bufs.size() && idle > KEEP_FREE;) {  // drop the smallest idle buffers first
            size_t victim = bufs.size();
            for (size_t k = 0; k < bufs.size(); k++)
                if (!bufs[k].used && (victim == bufs.size() || bufs[k].cap < bufs[victim].cap)) victim = k;
            if (victim == bufs.size()) break;
            (void)hipHostFree(bufs[victim].p);
            bufs.erase(bufs.begin() + (long)victim);
            idle--;
        }
        return true;
    }
    void trim() {  // give idle page-locked buffers back (called when a context goes away)
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < bufs.size();) {
            if (!bufs[i].used) {
                (void)hipHostFree(bufs[i].p);
                bufs.erase(bufs.begin() + (long)i);
            } else {
                i++;
            }
        }
    }
};
PinnedPool g_pinned;
}  // namespace

static void pinned_pool_trim() { g_pinned.trim(); }

static void* out_alloc(size_t n) {
    void* p = g_pinned.get(n);
    return p ? p : malloc(n ? n : 1);
}

extern "C" void rv_free(void* p) {
    if (p && !g_pinned.put(p)) free(p);
}

extern "C" int rv_ctx_profile(rv_ctx* ctx, int enable, int reset, rv_profile* out) {
    if (!ctx) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->collect();
    if (out) *out = ctx->prof;
    if (reset) ctx->prof = rv_profile{};
    ctx->profiling = enable != 0;
    return RV_OK;
}

// ------------------------------------------------------------------------------------
// circuit
// ------------------------------------------------------------------------------------
// Early corrections (rv_prove_impl; kernels.hip "Early corrections"): which byte ranges of the repetitions' corrections
// vectors leave for the host after which level.  A pure function of the compiled circuit, computed on first use.
struct EarlyPlan {
    bool ok = false;
    struct Chunk {
        uint64_t byte0, nbytes, pitch;  // bytes [byte0, byte0 + nbytes) of every repetition's vector; row stride in the staging block
        size_t off;                     // the chunk's [256][pitch] block in the staging buffers (device and host alike)
        uint32_t ready_level;           // its preprocessing rows are final once levels 0 .. ready_level have run
    };
    std::vector<Chunk> chunks;
    size_t bytes = 0;
    // Z64 circuits (pure Z64, no B2A): a repetition's corrections vector IS its preprocessing transcript ([R][words] in HBM), so
    // there is nothing to pack -- a chunk is a word range of the first r_spec repetitions' rows, copied as one 2-D transfer into
    // staging rows of `pitch` bytes.  r_spec < 256 when all repetitions' vectors would not fit through PCIe beside the proof's
    // kernels: the opened repetitions beyond it are extracted and copied the plain way.
    bool z64 = false;
    uint32_t r_spec = RV_TOTAL_REPS;
};

struct rv_circuit {
    rv_ctx* ctx = nullptr;
    const uint8_t* staged = nullptr;  // circuit_stage: the arrays circuit_upload sends first, already in page-locked memory
    size_t staged_bytes = 0;
    Compiled cc;  // gates kept on the host too (level table, counts)
    Gate* d_gates = nullptr;
    uint32_t* d_rec_rows = nullptr;
    uint32_t* d_in_rows = nullptr;
    Gate64* d_gates64 = nullptr;
    // the fused Z64 prover (internal.h: Z64FParams): the gates of every level grouped Mul | linear | other, and the cipher
    // block runs [first, first + n) whose mask rows belong to Input gates (generated the plain way)
    bool z64f_ok = false;
    Gate64* d_gates64f = nullptr;
    std::vector<Z64FLevel> z64f_levels;
    std::vector<std::pair<uint64_t, uint64_t>> z64f_runs;
    uint64_t* d_rec_offs64 = nullptr;
    uint64_t* d_in_offs64 = nullptr;
    uint32_t* d_level_start = nullptr;
    LevelRange* d_level_range = nullptr;
    // maximal runs [first, last) of consecutive narrow GF(2)-only levels, executed by one workgroup each
    struct NarrowRun {
        uint32_t first, second;  // levels [first, second)
        int tiny;                // 1: the plain per-gate kernel, 0: the class-loop kernel (every level > 32 gates), 2: its lean variant
    };
    std::vector<NarrowRun> narrow_runs;
    std::vector<int32_t> run_of_level;  // index into narrow_runs or -1
    // MODE_PROVE_V (cleartext wire values instead of corr rows, internal.h) is possible: pure GF(2), no Random / B2A
    // gates, every level launched on its own (no single-workgroup narrow runs)
    bool vclr_ok = false;
    // rep-sliced prover path (repprog.h): present when the circuit is eligible
    bool rep_ok = false;
    RepProgram rp;  // host copy without the big vectors (only counts are read after the upload)
    RepLevel* d_rep_levels = nullptr;
    RepSeg* d_rep_segs = nullptr;
    RepRec* d_rep_recs = nullptr;
    // LDS runs (ldsrun.h): narrow stretches whose live wires fit the LDS; preferred over narrow_runs when the shard's
    // row width is a multiple of the run's slice width
    struct LdsPlan {
        LdsRun run;
        uint32_t qs;
    };
    std::vector<LdsPlan> lds_runs;
    std::vector<int32_t> lds_run_of_level;  // index into lds_runs or -1
    LdsRec* d_lds_recs = nullptr;
    mutable std::once_flag ec_once;
    mutable EarlyPlan ec_plan;
    // k_interp_persist: per row width the levels' step tables (built and uploaded at first use), and whether any level has enough
    // multi-base gates for the kernel variant with their loops
    struct PersistTab {
        std::vector<PLevel> h;
        PLevel* d = nullptr;
    };
    mutable std::mutex persist_mu;
    mutable std::map<uint32_t, PersistTab> persist_tab;
    bool persist_gen = false;
    // flat prover schedule (flat.h): present when the circuit is eligible (the big vectors live on the device only)
    FlatPlan flat;
    Gate* d_xgates = nullptr;
    MulRec* d_muls = nullptr;
    Gate* d_others = nullptr;
    uint32_t n_others = 0, n_other_inputs = 0;  // (the Input gates first)
    ClearRec* d_clear_s = nullptr;
    ClearRecK* d_clear_k = nullptr;
    ClearLevel* d_clear_levels = nullptr;
    ClearRec* d_lite_s = nullptr;
    ClearRecK* d_lite_k = nullptr;
    ClearLevel* d_lite_levels = nullptr;
    PLevel* d_chain_levels = nullptr;  // k_chain: step tables of the levels' XOR classes (rows of 64 quad words)
    bool chain_gen = false;
};

#ifdef RV_EXPERIMENTS
// RV_FLAT: 0 (default) = the level-synchronous interpreter everywhere; 1 = the flat schedule for the prover of eligible circuits
// of at least RV_FLAT_MIN gates (2^20: below, the cleartext pass does not hide behind the mask generator); 2 = for every eligible
// circuit.  Read at every call (tests switch it).  Off by default: byte-identical, but on the 10^7-gate circuit the ~140 dependent
// x-level launches of its XOR rows cost what the level boundaries saved ("Flat schedule").
static int flat_mode() {
    const char* e = getenv("RV_FLAT");
    return e ? atoi(e) : 0;
}
#endif
// workgroups of the cleartext pass = compute units the mask generator leaves free for them
static uint32_t clear_wgs() {
    static const uint32_t v = [] {
        const char* e = getenv("RV_CLEAR_WGS");
        return e ? (uint32_t)std::min(std::max(atoi(e), 1), 64) : 8u;
    }();
    return v;
}

// RV_Z64_FUSED: 1 (default) = the prover of eligible Z64 circuits runs its mask generator inside the interpreter's level launches
// (internal.h: Z64FParams); 0 = masks to HBM first, k_interp64 behind.  Read at every call (tests switch it).
static bool z64_fused_on() {
    const char* e = getenv("RV_Z64_FUSED");
    return !e || atoi(e) != 0;
}
// the circuit's Z64 gates grouped Mul | linear | other inside every level, the level table, and the cipher block runs whose rows
// are Input masks.  false: not eligible (a Random or B2A gate, a Mul whose two masks straddle cipher blocks, too many runs)
static bool build_z64_fused(const Compiled& cc, std::vector<Gate64>& sorted, std::vector<Z64FLevel>& levels, std::vector<std::pair<uint64_t, uint64_t>>& runs) {
    const size_t n_levels = cc.level_start64.empty() ? 0 : cc.level_start64.size() - 1;
    if (!n_levels || cc.gates64.size() >= (1ull << 32)) return false;
    auto cls = [](uint32_t op) -> int {
        switch (op) {
        case G64_MUL: return 0;
        case G64_ADD: case G64_SUB: case G64_ADDC: case G64_SUBC: case G64_MULC: return 1;
        case G64_INPUT: case G64_ASSERT: case G64_CONST: return 2;
        default: return -1;
        }
    };
    std::vector<uint64_t> in_blocks;
    for (const Gate64& g : cc.gates64) {
        const int k = cls(g.op);
        if (k < 0) return false;
        if (g.op == G64_MUL && (g.m & 1)) return false;
        if (g.op == G64_INPUT) in_blocks.push_back(g.m >> 1);
    }
    std::sort(in_blocks.begin(), in_blocks.end());
    runs.clear();
    for (uint64_t b : in_blocks) {
        if (!runs.empty() && b < runs.back().first + runs.back().second) continue;
        if (!runs.empty() && b == runs.back().first + runs.back().second)
            runs.back().second++;
        else
            runs.emplace_back(b, 1);
    }
    if (runs.size() > 64) return false;
    sorted.resize(cc.gates64.size());
    levels.assign(n_levels, Z64FLevel{});
    for (size_t l = 0; l < n_levels; l++) {
        const uint64_t lo = cc.level_start64[l], hi = cc.level_start64[l + 1];
        uint64_t n[3] = {0, 0, 0};
        for (uint64_t i = lo; i < hi; i++) n[cls(cc.gates64[i].op)]++;
        uint64_t at[3] = {lo, lo + n[0], lo + n[0] + n[1]};
        levels[l] = Z64FLevel{(uint32_t)lo, (uint32_t)at[1], (uint32_t)at[2], (uint32_t)hi};
        for (uint64_t i = lo; i < hi; i++) sorted[at[cls(cc.gates64[i].op)]++] = cc.gates64[i];
    }
    return true;
}

#ifdef RV_EXPERIMENTS
// LDS the rep-sliced interpreter may use for wire slots (a workgroup owns the CU: 160 KiB minus a little headroom)
static uint32_t rep_lds_budget(const rv_ctx* ctx) {
    static const uint32_t v = [] {
        const char* e = getenv("RV_REP_LDS");
        return e ? (uint32_t)atoi(e) : 156u * 1024u;
    }();
    const size_t dev = ctx->lds_bytes > 4096 ? ctx->lds_bytes - 4096 : 0;
    return (uint32_t)std::min<size_t>(v, dev);
}
#endif
// RV_REP: 0 (default) = the row path everywhere; 1 = the rep-sliced path for whole proofs (all 256 repetitions on this
// GPU) of circuits it accepts; 2 = for shards too.  Read at every call (tests switch it).  Off by default: measured on
// MI355X it is byte-identical but not yet faster than the row path ("Rep-sliced path").
static int rep_mode() {
#ifdef RV_EXPERIMENTS
    const char* e = getenv("RV_REP");
    return e ? atoi(e) : 0;
#else
    return 0;  // (the rep-sliced path exists in experiment builds only: csrc/Makefile)
#endif
}

static size_t scratch_bytes_for(const Compiled& cc, uint32_t R) {
    const size_t NQ = R / 4;
    size_t b = 0;
    b += cc.n_rows * NQ * 4;
    b += cc.n_rows * (NQ / 2);
    b += cc.n_on * NQ * 4 + cc.n_pre * (NQ / 2);
    b += 4 * b3_stream_scratch_words(std::max(cc.n_on, cc.n_pre), R) * 4;
    b += (size_t)R * (16 + 128 + 8 * RK_BYTES) + RK_AREAS * 128 * NQ * 4;
    // Z64: masks, wires, contiguous transcripts
    b += ((cc.n_masks64 + 1) / 2 * 2) * (size_t)R * 64;
    b += cc.n_ssa64 * (size_t)R * 72;
    b += (cc.on_words64 + cc.pre_words64) * (size_t)R * 8;
    return b;
}

static int rv_circuit_compile_impl(rv_ctx* ctx, const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags,
                                  rv_circuit** out);
static int circuit_upload(rv_ctx* ctx, rv_circuit* c);

extern "C" int rv_circuit_compile_ex(rv_ctx* ctx, const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags,
                                     rv_circuit** out) {
    if (flags & ~RV_COMPILE_WHOLE_PROVER) return RV_E_ARG;
    try {  // no C++ exception may cross the C boundary
        return rv_circuit_compile_impl(ctx, ops, n_ops, z64_wires, gf2_wires, flags, out);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

extern "C" int rv_circuit_compile(rv_ctx* ctx, const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires,
                                  rv_circuit** out) {
    return rv_circuit_compile_ex(ctx, ops, n_ops, z64_wires, gf2_wires, 0, out);
}

static int rv_circuit_compile_impl(rv_ctx* ctx, const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags,
                                  rv_circuit** out) {
    if (!ctx || !out || (n_ops && !ops)) return RV_E_ARG;
    *out = nullptr;
    rv_circuit* c = new rv_circuit();
    c->ctx = ctx;
    const auto t_begin = std::chrono::steady_clock::now();
    // RV_COMPILE_WHOLE_PROVER: lazy sums of up to RV_LIN_K rows for every circuit (the compiler chooses them on its own only
    // for deep, narrow ones); RV_LAZY_K still overrides
    int rc = compile_ops(ops, n_ops, z64_wires, gf2_wires, c->cc, nullptr, ((flags & RV_COMPILE_WHOLE_PROVER) && !getenv("RV_LAZY_K")) ? RV_LIN_K : 0);
    if (rc) {
        delete c;
        return rc;
    }
    const auto t_compiled = std::chrono::steady_clock::now();
    c->cc.info.compile_us = (uint64_t)std::chrono::duration<double, std::micro>(t_compiled - t_begin).count();
    if (getenv("RV_COMPILE_STATS"))
        fprintf(stderr, "[rv circuit] compile_ops: %.3f s for %zu ops\n", std::chrono::duration<double>(t_compiled - t_begin).count(),
                n_ops);
#ifdef RV_EXPERIMENTS
    // the rep-sliced program of the prover (pure GF(2) circuits whose live wires fit the LDS): from this compile when it
    // keeps one base row per wire, else from a second compile that does
    if (rep_mode() && c->cc.gates64.empty() && !c->cc.gates.empty()) {
        const char* why = "";
        c->rep_ok = build_rep_program(c->cc, rep_lds_budget(ctx), c->rp, &why);
        if (!c->rep_ok && strstr(why, "base")) {
            Compiled one;
            if (compile_ops(ops, n_ops, z64_wires, gf2_wires, one, nullptr, 1) == RV_OK) c->rep_ok = build_rep_program(one, rep_lds_budget(ctx), c->rp, &why);
        }
        if (getenv("RV_COMPILE_STATS"))
            fprintf(stderr, "[rv circuit] rep-sliced path: %s%s (%u levels, %zu segments, %u LDS slots) at %.3f s\n", c->rep_ok ? "yes" : "no: ", why,
                    c->rp.n_levels, c->rp.segs.size(), c->rp.lds_slots, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    }
#endif  // RV_EXPERIMENTS
    if ((rc = circuit_upload(ctx, c))) return rc;  // (destroys c on failure)
    if (getenv("RV_COMPILE_STATS"))
        fprintf(stderr, "[rv circuit] compiled + uploaded after %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    *out = c;
    return RV_OK;
}

// The arrays circuit_upload sends first, in its order, each rounded up to 256 bytes: their size, and a copy of them into `dst`
// (page-locked; any thread, no HIP call) that circuit_upload then sends from instead of going through the context's staging buffer
template <class F>
static void circuit_stage_each(const Compiled& cc, F f) {
    f(cc.gates.data(), cc.gates.size() * sizeof(Gate));
    f(cc.rec_rows.data(), cc.rec_rows.size() * 4);
    f(cc.in_rows.data(), cc.in_rows.size() * 4);
    f(cc.gates64.data(), cc.gates64.size() * sizeof(Gate64));
    f(cc.rec_offs64.data(), cc.rec_offs64.size() * 8);
    f(cc.in_offs64.data(), cc.in_offs64.size() * 8);
    f(cc.level_start.data(), cc.level_start.size() * 4);
    f(cc.level_range.data(), cc.level_range.size() * sizeof(LevelRange));
}
static size_t circuit_stage_bytes(const Compiled& cc) {
    size_t n = 0;
    circuit_stage_each(cc, [&](const void*, size_t b) { n += (b + 255) & ~(size_t)255; });
    return n;
}
static void circuit_stage(rv_circuit* c, uint8_t* dst) {
    size_t off = 0;
    circuit_stage_each(c->cc, [&](const void* p, size_t b) {
        if (b) memcpy(dst + off, p, b);
        off += (b + 255) & ~(size_t)255;
    });
    c->staged = dst;
    c->staged_bytes = off;
}

// the compiled gate stream (c->cc) to HBM + the narrow-run plan; c is destroyed on failure
static int circuit_upload(rv_ctx* ctx, rv_circuit* c) {
    const auto t_compiled = std::chrono::steady_clock::now();
    int rc;
// (c is destroyed on EVERY failure: the callers rely on it)
#define UPCHK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            const int code_ = hip_fail(e_, #x, __FILE__, __LINE__);     \
            rv_circuit_destroy(c);                                      \
            return code_;                                               \
        }                                                               \
    } while (0)
    UPCHK(hipSetDevice(ctx->device));
    const Compiled& cc = c->cc;
    // everything below goes through one page-locked buffer when it fits (the function waits for the stream before it
    // returns, so the buffer is free again for the next circuit)
    size_t stage_need = (size_t)1 << 20;  // (+ the LDS-run records, built further down: they fall back to a pageable copy when they do not fit)
    for (size_t b : {cc.gates.size() * sizeof(Gate), cc.rec_rows.size() * 4, cc.in_rows.size() * 4, cc.gates64.size() * sizeof(Gate64),
                     cc.rec_offs64.size() * 8, cc.in_offs64.size() * 8, cc.level_start.size() * 4, cc.level_range.size() * sizeof(LevelRange)})
        stage_need += (b + 255) & ~(size_t)255;
    static const bool stage_on = !(getenv("RV_UPLOAD_STAGE") && atoi(getenv("RV_UPLOAD_STAGE")) == 0);
    if (stage_on && stage_need <= rv_ctx::UP_STAGE_MAX && stage_need > ctx->h_up_cap) {
        if (ctx->h_up) {
            (void)hipStreamSynchronize(ctx->stream);  // (no copy out of the old buffer may still be pending)
            (void)hipHostFree(ctx->h_up);
        }
        ctx->h_up = nullptr;
        ctx->h_up_cap = 0;
        const size_t want = std::min(rv_ctx::UP_STAGE_MAX, std::max(stage_need + stage_need / 4, (size_t)8 << 20));
        if (hipHostMalloc((void**)&ctx->h_up, want, hipHostMallocDefault) == hipSuccess)
            ctx->h_up_cap = want;
        else
            (void)hipGetLastError();
    }
    size_t stage_off = 0;
    auto up = [&](const void* src, size_t bytes, void** dst) -> int {
        int r = ctx->alloc(bytes, dst);
        if (r) return r;
        if (!bytes) return RV_OK;
        if (c->staged && stage_off + bytes <= c->staged_bytes) {  // (a streaming piece: copied here by a worker thread)
            src = c->staged + stage_off;
            stage_off += (bytes + 255) & ~(size_t)255;
        } else if (stage_on && ctx->h_up && stage_off + bytes <= ctx->h_up_cap && stage_need <= rv_ctx::UP_STAGE_MAX) {
            memcpy(ctx->h_up + stage_off, src, bytes);
            src = ctx->h_up + stage_off;
            stage_off += (bytes + 255) & ~(size_t)255;
        }
        HIPCHK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return RV_OK;
    };
    if ((rc = up(cc.gates.data(), cc.gates.size() * sizeof(Gate), (void**)&c->d_gates)) ||
        (rc = up(cc.rec_rows.data(), cc.rec_rows.size() * 4, (void**)&c->d_rec_rows)) ||
        (rc = up(cc.in_rows.data(), cc.in_rows.size() * 4, (void**)&c->d_in_rows)) ||
        (rc = up(cc.gates64.data(), cc.gates64.size() * sizeof(Gate64), (void**)&c->d_gates64)) ||
        (rc = up(cc.rec_offs64.data(), cc.rec_offs64.size() * 8, (void**)&c->d_rec_offs64)) ||
        (rc = up(cc.in_offs64.data(), cc.in_offs64.size() * 8, (void**)&c->d_in_offs64)) ||
        (rc = up(cc.level_start.data(), cc.level_start.size() * 4, (void**)&c->d_level_start)) ||
        (rc = up(cc.level_range.data(), cc.level_range.size() * sizeof(LevelRange), (void**)&c->d_level_range))) {
        rv_circuit_destroy(c);
        return rc;
    }
    if (c->rep_ok) {
        if ((rc = up(c->rp.levels.data(), c->rp.levels.size() * sizeof(RepLevel), (void**)&c->d_rep_levels)) ||
            (rc = up(c->rp.segs.data(), c->rp.segs.size() * sizeof(RepSeg), (void**)&c->d_rep_segs)) ||
            (rc = up(c->rp.recs.data(), c->rp.recs.size() * sizeof(RepRec), (void**)&c->d_rep_recs))) {
            rv_circuit_destroy(c);
            return rc;
        }
    }
    UPCHK(hipStreamSynchronize(ctx->stream));
    c->staged = nullptr;  // (the slot belongs to the next piece from here on)
    c->staged_bytes = 0;
    if (c->rep_ok) {  // the device holds them now
        std::vector<RepRec>().swap(c->rp.recs);
        std::vector<RepSeg>().swap(c->rp.segs);
        std::vector<RepLevel>().swap(c->rp.levels);
    }
    c->cc.info.upload_us = (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_compiled).count();
    {
        const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
        c->run_of_level.assign(n_levels, -1);
        // gates; one 1024-thread workgroup covers 64 full-width gates per (4-way unrolled) step
        static const uint32_t NARROW = std::min<uint32_t>(getenv("RV_NARROW") ? (uint32_t)atoi(getenv("RV_NARROW")) : 256, 512);  // <= NARROW_WIN / 2
        size_t l = 0;
        while (l < n_levels) {
            auto narrow = [&](size_t i) {
                const bool no64 = cc.level_start64.empty() || cc.level_start64[i + 1] == cc.level_start64[i];
                return no64 && cc.level_start[i + 1] - cc.level_start[i] <= NARROW;
            };
            if (!narrow(l)) {
                l++;
                continue;
            }
            size_t e = l;
            while (e < n_levels && narrow(e)) e++;
            if (e - l >= 3) {
                // split the run: stretches of >= 8 levels that are all wider than 32 gates go to the class-loop kernel
                // (4 gates per wavefront step, ~2 us per 64 gates), everything else to the per-gate kernel (one gate
                // per wavefront, 1.2 us per 16 gates; measured on SHA-256 / AES-128)
                auto wide = [&](size_t i) { return cc.level_start[i + 1] - cc.level_start[i] > 32; };
                size_t a = l;
                while (a < e) {
                    size_t b = a;
                    const bool w = wide(a);
                    while (b < e && wide(b) == w) b++;
                    int tiny = (w && b - a >= 8) ? 0 : 1;
                    if (!tiny) {  // a class-loop stretch without multi-base gates takes the lean variant (8-gate Xor steps)
                        bool lean = true;
                        for (size_t i = a; i < b; i++) {
                            const LevelRange& lr = cc.level_range[i];
                            lean = lean && lr.mul == lr.mul11 && lr.xork == lr.xor2;
                        }
                        if (lean) tiny = 2;
                    }
                    if (tiny == 1 && a > l && !c->narrow_runs.empty() && c->narrow_runs.back().second == a && c->narrow_runs.back().tiny == 1) {
                        c->narrow_runs.back().second = (uint32_t)b;  // merge with the preceding per-gate piece
                    } else {
                        c->narrow_runs.push_back(rv_circuit::NarrowRun{(uint32_t)a, (uint32_t)b, tiny});
                    }
                    for (size_t i = a; i < b; i++) c->run_of_level[i] = (int32_t)c->narrow_runs.size() - 1;
                    a = b;
                }
            }
            l = e;
        }
    }
    {
        // LDS runs over the maximal narrow stretches (RV_LDS_RUN=0 switches them off; RV_LDS_QS=2/4 fixes the slice width)
        const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
        c->lds_run_of_level.assign(n_levels, -1);
        const int lds_on = getenv("RV_LDS_RUN") ? atoi(getenv("RV_LDS_RUN")) : 1;  // (read per circuit: the tests switch them)
        const uint32_t qs_env = getenv("RV_LDS_QS") ? (uint32_t)atoi(getenv("RV_LDS_QS")) : 0;
        static const uint32_t NARROW = std::min<uint32_t>(getenv("RV_NARROW") ? (uint32_t)atoi(getenv("RV_NARROW")) : 256, 512);
        bool any = false;
        for (size_t l = 0; l < n_levels && !any; l++) any = c->run_of_level[l] >= 0;
        if (lds_on && any && !cc.row_prg_base) {
            LdsRunScratch scratch;
            size_t first_narrow = 0;
            while (first_narrow < n_levels && c->run_of_level[first_narrow] < 0) first_narrow++;
            scratch.init(cc, (uint32_t)first_narrow);
            std::vector<LdsRec> recs;
            auto narrow = [&](size_t i) {
                const bool no64 = cc.level_start64.empty() || cc.level_start64[i + 1] == cc.level_start64[i];
                return no64 && cc.level_start[i + 1] - cc.level_start[i] <= NARROW;
            };
            size_t l = 0;
            while (l < n_levels) {
                if (!narrow(l)) {
                    l++;
                    continue;
                }
                size_t e = l;
                while (e < n_levels && narrow(e)) e++;
                if (e - l >= 3) {
                    // one quad word per slice: 64 gates per step, the fewest steps per level, and the most workgroups (the two
                    // quads sharing a byte of a bit-packed row then sit in different workgroups: lr_put_nibble, ldsrun.hip);
                    // 2 or 4 on request
                    const uint32_t qs = qs_env == 4 ? 4u : qs_env == 2 ? 2u : 1u;
                    const size_t budget = std::min<size_t>(160 * 1024, ctx->lds_bytes) - 1024;
                    const size_t fixed = lds_run_bytes(qs, 0);
                    if (budget < fixed + 64 * qs * 8) {  // not even a handful of wire slots next to the ring: the row interpreter's narrow runs
                        l = e;
                        continue;
                    }
                    const uint32_t max_slots = (uint32_t)std::min<size_t>((budget - fixed) / (qs * 8), LR_NONE - 1);
                    rv_circuit::LdsPlan plan{};
                    plan.qs = qs;
                    if (build_lds_run(cc, (uint32_t)l, (uint32_t)e, qs, max_slots, scratch, recs, plan.run)) {
                        for (size_t i = l; i < e; i++) c->lds_run_of_level[i] = (int32_t)c->lds_runs.size();
                        c->lds_runs.push_back(plan);
                    }
                }
                l = e;
            }
            if (!recs.empty()) {
                if ((rc = up(recs.data(), recs.size() * sizeof(LdsRec), (void**)&c->d_lds_recs))) {
                    rv_circuit_destroy(c);
                    return rc;
                }
                UPCHK(hipStreamSynchronize(ctx->stream));
            }
            if (getenv("RV_COMPILE_STATS"))
                for (const auto& pl : c->lds_runs)
                    fprintf(stderr, "[rv circuit] LDS run: levels [%u, %u), %u steps of %u gates, %u slots (%zu KiB of LDS)\n", pl.run.l0, pl.run.l1,
                            pl.run.n_steps, 64 / pl.qs, pl.run.n_slots, lds_run_bytes(pl.qs, pl.run.n_slots) >> 10);
        }
    }
    if (getenv("RV_COMPILE_STATS")) {
        size_t n_tiny = 0, n_med = 0, lv_tiny = 0, lv_med = 0;
        for (const auto& r : c->narrow_runs) (r.tiny == 1 ? n_tiny : n_med)++, (r.tiny == 1 ? lv_tiny : lv_med) += r.second - r.first;
        fprintf(stderr, "[rv circuit] narrow runs: %zu per-gate (%zu levels), %zu class-loop (%zu levels), %zu levels launched one by one\n",
                n_tiny, lv_tiny, n_med, lv_med, c->run_of_level.size() - lv_tiny - lv_med);
    }
    c->cc.info.device_bytes = cc.gates.size() * sizeof(Gate) + (cc.rec_rows.size() + cc.in_rows.size()) * 4 +
                              cc.gates64.size() * sizeof(Gate64) + (cc.rec_offs64.size() + cc.in_offs64.size()) * 8;
    c->cc.info.scratch_bytes = scratch_bytes_for(cc, RV_TOTAL_REPS);
    {
        // (MODE_PROVE_V launches every level on its own: fine when only a handful of levels sit in narrow runs)
        size_t narrow_levels = 0;
        for (const auto& r : c->narrow_runs) narrow_levels += r.second - r.first;
        c->vclr_ok = cc.gates64.empty() && narrow_levels <= 16 && !cc.row_prg_base;
    }
    if (cc.n_random_or_recon) c->vclr_ok = false;  // (values that differ between repetitions)
    if (!cc.gates64.empty() && z64_fused_on() && !cc.row_prg_base) {
        std::vector<Gate64> sorted;
        if (build_z64_fused(cc, sorted, c->z64f_levels, c->z64f_runs)) {
            if ((rc = up(sorted.data(), sorted.size() * sizeof(Gate64), (void**)&c->d_gates64f))) {
                rv_circuit_destroy(c);
                return rc;
            }
            UPCHK(hipStreamSynchronize(ctx->stream));
            c->z64f_ok = true;
            c->cc.info.device_bytes += sorted.size() * sizeof(Gate64);
        }
    }
    c->persist_gen = persist_general(cc.level_range.data(), cc.level_range.size());
#ifdef RV_EXPERIMENTS
    if (c->vclr_ok && flat_mode()) {
        // the flat schedule of the prover: Mul records in program order, XOR rows by x-level, the rest
        static const uint64_t flat_min = getenv("RV_FLAT_MIN") ? (uint64_t)atoll(getenv("RV_FLAT_MIN")) : (1ull << 20);
        const uint32_t bands = getenv("RV_FLAT_BANDS") ? (uint32_t)std::max(atoi(getenv("RV_FLAT_BANDS")), 1) : 8u;
        if ((flat_mode() >= 2 || cc.gates.size() >= flat_min) && build_flat_plan(cc, c->flat, bands)) {
            c->n_others = (uint32_t)c->flat.others.size();
            c->n_other_inputs = c->flat.n_other_inputs;
            if ((rc = up(c->flat.xgates.data(), c->flat.xgates.size() * sizeof(Gate), (void**)&c->d_xgates)) ||
                (rc = up(c->flat.muls.data(), c->flat.muls.size() * sizeof(MulRec), (void**)&c->d_muls)) ||
                (rc = up(c->flat.others.data(), c->flat.others.size() * sizeof(Gate), (void**)&c->d_others)) ||
                (rc = up(c->flat.clear_s.data(), c->flat.clear_s.size() * sizeof(ClearRec), (void**)&c->d_clear_s)) ||
                (rc = up(c->flat.clear_k.data(), c->flat.clear_k.size() * sizeof(ClearRecK), (void**)&c->d_clear_k)) ||
                (rc = up(c->flat.clear_levels.data(), c->flat.clear_levels.size() * sizeof(ClearLevel), (void**)&c->d_clear_levels)) ||
                (rc = up(c->flat.lite_s.data(), c->flat.lite_s.size() * sizeof(ClearRec), (void**)&c->d_lite_s)) ||
                (rc = up(c->flat.lite_k.data(), c->flat.lite_k.size() * sizeof(ClearRecK), (void**)&c->d_lite_k)) ||
                (rc = up(c->flat.lite_levels.data(), c->flat.lite_levels.size() * sizeof(ClearLevel), (void**)&c->d_lite_levels))) {
                rv_circuit_destroy(c);
                return rc;
            }
            c->chain_gen = chain_general(cc.level_range.data(), cc.level_range.size());
            std::vector<PLevel> xl(cc.level_range.size());
            build_chain_levels(cc.level_range.data(), cc.level_range.size(), 64, c->chain_gen, xl.data());
            if ((rc = up(xl.data(), xl.size() * sizeof(PLevel), (void**)&c->d_chain_levels))) {
                rv_circuit_destroy(c);
                return rc;
            }
            UPCHK(hipStreamSynchronize(ctx->stream));
            decltype(c->flat.xgates)().swap(c->flat.xgates);  // the device holds them now
            decltype(c->flat.muls)().swap(c->flat.muls);
            std::vector<Gate>().swap(c->flat.others);
            decltype(c->flat.clear_s)().swap(c->flat.clear_s);
            decltype(c->flat.clear_k)().swap(c->flat.clear_k);
            std::vector<ClearLevel>().swap(c->flat.clear_levels);
            decltype(c->flat.lite_s)().swap(c->flat.lite_s);
            decltype(c->flat.lite_k)().swap(c->flat.lite_k);
        }
    }
#endif  // RV_EXPERIMENTS
    return RV_OK;
#undef UPCHK
}

extern "C" void rv_circuit_destroy(rv_circuit* c) {
    if (!c) return;
    c->ctx->release(c->d_gates);
    c->ctx->release(c->d_gates64f);
    c->ctx->release(c->d_rec_rows);
    c->ctx->release(c->d_in_rows);
    c->ctx->release(c->d_gates64);
    c->ctx->release(c->d_rec_offs64);
    c->ctx->release(c->d_in_offs64);
    c->ctx->release(c->d_level_start);
    c->ctx->release(c->d_level_range);
    c->ctx->release(c->d_rep_levels);
    c->ctx->release(c->d_rep_segs);
    c->ctx->release(c->d_rep_recs);
    c->ctx->release(c->d_lds_recs);
    for (auto& kv : c->persist_tab) c->ctx->release(kv.second.d);
    c->ctx->release(c->d_xgates);
    c->ctx->release(c->d_muls);
    c->ctx->release(c->d_others);
    c->ctx->release(c->d_clear_s);
    c->ctx->release(c->d_clear_k);
    c->ctx->release(c->d_clear_levels);
    c->ctx->release(c->d_lite_s);
    c->ctx->release(c->d_lite_k);
    c->ctx->release(c->d_lite_levels);
    c->ctx->release(c->d_chain_levels);
    delete c;
}

extern "C" int rv_hook_compile_info(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, size_t chunk_ops,
                                    rv_circuit_info* info) {
    if (!info || (n_ops && !ops) || (flags & ~RV_COMPILE_WHOLE_PROVER)) return RV_E_ARG;
    try {
        Compiled cc;
        int rc = compile_ops(ops, n_ops, z64_wires, gf2_wires, cc, nullptr, ((flags & RV_COMPILE_WHOLE_PROVER) && !getenv("RV_LAZY_K")) ? RV_LIN_K : 0);
        if (rc) return rc;
        *info = cc.info;
        if (chunk_ops) {
            // the streaming prover's bookkeeping (stream.inc): pieces compiled independently, their ShareGen phases from
            // count_masks of the ops before them, must consume exactly the whole program's masks and transcript events
            uint64_t m2 = 0, m64 = 0, on = 0, pre = 0, muls = 0, onw = 0, prew = 0;
            for (size_t at = 0; at < n_ops; at += chunk_ops) {
                const size_t n = std::min(chunk_ops, n_ops - at);
                ChunkStart cs;
                cs.mask_phase = (uint32_t)(m2 % 128);
                cs.mask64_phase = (uint32_t)(m64 % 2);
                Compiled piece;
                if ((rc = compile_ops(ops + at, n, z64_wires, gf2_wires, piece, &cs))) return rc;
                uint64_t a = 0, b = 0;
                count_masks(ops + at, n, &a, &b);
                if (piece.n_masks - cs.mask_phase != a || piece.n_masks64 - cs.mask64_phase != b) return RV_E_DEVICE;
                // ... and the transcript events count_events predicts (the workers place a piece at the offsets they imply)
                StreamEvents ev;
                count_events(ops + at, n, &ev);
                if (piece.n_on != ev.in2 + ev.rec2 || piece.n_in != ev.in2 || piece.n_rec != ev.rec2 || piece.n_pre != ev.pre2 ||
                    piece.on_words64 != ev.on64 || piece.pre_words64 != ev.pre64)
                    return RV_E_DEVICE;
                const uint64_t on_before = piece.n_on, pre_before = piece.n_pre;
                relocate_chunk(piece, 7, 5, 3, 2);
                if (piece.n_on != on_before + 7 || piece.n_pre != pre_before + 5) return RV_E_DEVICE;
                m2 += a, m64 += b, on += on_before, pre += pre_before, muls += piece.info.gf2_muls;
                onw += piece.on_words64 - 3, prew += piece.pre_words64 - 2;
            }
            if (m2 != cc.n_masks || m64 != cc.n_masks64 || on != cc.n_on || pre != cc.n_pre || muls != cc.info.gf2_muls || onw != cc.on_words64 ||
                prew != cc.pre_words64)
                return RV_E_DEVICE;
        }
        return RV_OK;
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

extern "C" int rv_hook_compile_compare(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, int threads, int* diff) {
    if (!diff || (n_ops && !ops) || (flags & ~RV_COMPILE_WHOLE_PROVER) || threads < 2) return RV_E_ARG;
    try {
        const int k = ((flags & RV_COMPILE_WHOLE_PROVER) && !getenv("RV_LAZY_K")) ? RV_LIN_K : 0;
        Compiled a, b;
        const int rc = compile_ops_seq(ops, n_ops, z64_wires, gf2_wires, a, nullptr, k);
        const int rp = compile_ops_par(ops, n_ops, z64_wires, gf2_wires, b, k, threads);
        if (rp == RV_COMPILE_FALLBACK)
            *diff = -1;
        else if (rp != rc)
            *diff = 100;
        else
            *diff = rc == RV_OK ? compiled_diff(a, b) : 0;
        return rc;
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

static uint64_t early_staging_bytes_of(const rv_circuit* c);  // (with the early-corrections plan below)
extern "C" int rv_circuit_get_info(const rv_circuit* c, rv_circuit_info* info) {
    if (!c || !info) return RV_E_ARG;
    *info = c->cc.info;
    return RV_OK;
}
extern "C" int rv_circuit_early_staging_bytes(const rv_circuit* c, uint64_t* bytes) {
    if (!c || !bytes) return RV_E_ARG;
    try {
        *bytes = early_staging_bytes_of(c);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
    return RV_OK;
}

// ------------------------------------------------------------------------------------
// Fiat-Shamir (host): 8 KiB hash + a few XOF blocks
// ------------------------------------------------------------------------------------
extern "C" int rv_combine_digests(const uint8_t* h, uint8_t comm[RV_HASH_SIZE]) {
    if (!h || !comm) return RV_E_ARG;
    b3::Hasher hs;
    hs.update(h, RV_TOTAL_REPS * RV_HASH_SIZE);
    hs.finalize(comm);
    return RV_OK;
}

extern "C" int rv_challenge(const uint8_t comm[RV_HASH_SIZE], uint8_t omit[RV_TOTAL_REPS]) {
    if (!comm || !omit) return RV_E_ARG;
    static const char CTX[] = "random-oracle challenge";  // proof/mod.rs:18
    b3::Hasher hs;
    hs.update(CTX, sizeof CTX - 1);
    const uint8_t zero = 0;
    hs.update(&zero, 1);  // crypto/ro.rs:11
    hs.update(comm, RV_HASH_SIZE);
    memset(omit, RV_PLAYERS, RV_TOTAL_REPS);
    int count = 0;
    uint64_t pos = 0;
    while (count < RV_ONLINE_REPS) {
        uint8_t buf[32];
        hs.xof(pos, buf, 32);
        pos += 32;
        const unsigned rep = buf[0];      // u128 LE mod 256
        const unsigned om = buf[16] & 7;  // u128 LE mod 8
        if (omit[rep] == RV_PLAYERS) count++;
        omit[rep] = (uint8_t)om;  // a re-drawn repetition overwrites (HashMap::insert)
    }
    return RV_OK;
}

// ------------------------------------------------------------------------------------
// shard
// ------------------------------------------------------------------------------------
// one proof's early corrections: the plan, the device / host staging blocks, the chunks queued so far and the events that
// say a chunk has arrived on the host (owned by the shard's misc_events)
struct EarlyRun {
    const EarlyPlan* plan = nullptr;
    uint8_t* d_ec = nullptr;
    uint8_t* h_ec = nullptr;
    size_t next = 0;                 // chunks whose "ready" stamp is in the interpreter's stream
    size_t pumped = 0;               // chunks whose copy has been handed to the second stream
    std::vector<uint8_t> packed;     // per chunk: already packed when its stamp appears
    // progress stamps in the context's host-mapped mailbox (no HIP events: their status reached the host late, and the
    // host must see a chunk the moment it is ready): word 1 = (seq << 8 | chunks ready)
    uint32_t* box_dev = nullptr;
    volatile uint32_t* box = nullptr;
    uint32_t seq = 0;
    bool ready(size_t k) const {
        const uint32_t v = __atomic_load_n(&box[1], __ATOMIC_ACQUIRE);
        return (v >> 8) == (seq & 0xFFFFFFu) && (v & 0xFFu) > k;
    }
};

struct rv_shard {
    rv_ctx* ctx = nullptr;
    const rv_circuit* c = nullptr;
    uint32_t rep_begin = 0, R = 0, NQ = 0;
    const uint32_t* d_on_quads = nullptr;  // verifier: the quad words that hold an opened repetition (not owned)
    uint32_t n_on_quads = 0;
    uint8_t* d_seeds = nullptr;
    uint8_t* d_keys = nullptr;
    uint8_t* d_rkbytes = nullptr;
    uint32_t* d_rk = nullptr;
    uint32_t* d_masks = nullptr;  // share rows: PRG masks, then computed rows
    uint32_t* d_rk_c4 = nullptr;  // the lane-distributed generator's key image (aes_col4.hip), when it is the one that runs
    // RV_OVERLAP: the generator runs chunk by chunk on ctx->stream_m BESIDE the level launches; a chunk is submitted (and its event
    // waited for on the main stream) right before the first level that reads it -- see overlap_need()
    bool overlap = false;
    uint64_t ov_next = 0, ov_blocks = 0, ov_chunk = 0;  // next block to generate, all of them, blocks per chunk
    const uint32_t* ov_keep = nullptr;
    uint8_t* d_wires = nullptr;   // corr bits [n_ssa][NQ/2]
    uint32_t* d_on = nullptr;
    uint8_t* d_pre = nullptr;     // [n_pre][NQ/2]
    uint8_t* d_wit = nullptr;
    uint8_t* d_vclr = nullptr;  // MODE_PROVE_V: cleartext value per share row
    // flat schedule (flat.h): operand values per Mul, the cleartext pass's barrier words {arrivals, abort, error word}, its end
    bool flat = false, split = false;
    // split schedule with the transcript hashes band by band (RV_SPLIT_HASH, default on): chunks [0, *_chunks_done) of the two
    // transcripts have their chaining values in d_cv[0] (preprocessing) / d_cv[1] (online); d_cvx = the tree reductions' scratch
    bool split_hash = false;
    uint64_t pre_chunks_done = 0, on_chunks_done = 0;
    uint32_t* d_cvx = nullptr;
    uint8_t* d_vb = nullptr;
    uint32_t* d_sync = nullptr;
    hipEvent_t ev_clear = nullptr;
    // rep-sliced prover path (rep.hip): rep-major masks / transcripts instead of the row arrays above
    bool rep = false;
    uint8_t *d_masks_rep = nullptr, *d_on_rep = nullptr, *d_pre_rep = nullptr, *d_vbits = nullptr;
    uint32_t* d_rk_rep = nullptr;
    uint64_t mask_stride = 0, on_stride = 0, pre_stride = 0;
    // Z64 domain
    uint64_t* d_masks64 = nullptr;
    uint64_t* d_wmask64 = nullptr;
    uint64_t* d_wcorr64 = nullptr;
    bool z64f = false;            // the fused Z64 prover / verifier (internal.h: Z64FParams)
    hipEvent_t ev_sup64 = nullptr;        // ... set: the Z64 supplied values arrive on the side stream -- quad groups without an opened repetition run first
    std::function<int()> mid64;           // ... what brings them (the proof's copy, the unpack kernels), called once those groups' levels are queued
    const uint32_t* d_keep64z = nullptr;  // ... the verifier's kept streams per quad word (inside a block the caller tracks)
    uint64_t* d_v64 = nullptr;    // ... its cleartext values, one per Z64 SSA id
    uint64_t* d_on64 = nullptr;
    uint64_t* d_pre64 = nullptr;
    uint64_t* d_wit64 = nullptr;
    uint8_t* d_keys64 = nullptr;  // verifier only: the z64 openings carry their own key set
    uint32_t* d_rk64 = nullptr;
    uint8_t* d_omit64 = nullptr;
    uint32_t* d_cv[2] = {nullptr, nullptr};
    uint32_t* d_dig = nullptr;  // [4][R][8]: pre2, on2, pre64, on64
    uint8_t* d_h = nullptr;     // [R][32]
    int* d_err = nullptr;
    // open
    uint8_t* d_omit = nullptr;
    uint64_t* d_offs = nullptr;  // [5][R]
    uint8_t* d_out = nullptr;
    std::vector<void*> extra;
    // pipelining: the gf2 mask generator runs in chunks on ctx->stream; the interpreter (on
    // ctx->stream2) waits for the chunk a level needs
    std::vector<std::pair<uint64_t, hipEvent_t>> mask_chunks;  // (AES blocks complete, event)
    hipEvent_t ev_setup = nullptr;
    std::vector<hipEvent_t> misc_events;
    struct EarlyRun* ec = nullptr;  // early corrections of this proof (not owned)

    void destroy() {
        for (auto& c : mask_chunks) ctx->sync_pool.push_back(c.second);
        mask_chunks.clear();
        for (hipEvent_t e : misc_events) ctx->sync_pool.push_back(e);
        misc_events.clear();
        if (ev_setup) ctx->sync_pool.push_back(ev_setup);
        ev_setup = nullptr;
        if (ev_clear) ctx->sync_pool.push_back(ev_clear);
        ev_clear = nullptr;
        void* ps[] = {d_seeds, d_keys, d_rkbytes, d_rk,    d_masks,  d_wires,   d_on,     d_pre,    d_wit,  d_cv[0],
                      d_cv[1], d_dig,  d_h,       d_err,   d_omit,   d_offs,    d_out,    d_masks64, d_wmask64,
                      d_wcorr64, d_on64, d_pre64, d_wit64, d_keys64, d_rk64,    d_omit64, d_masks_rep, d_on_rep, d_pre_rep, d_vbits, d_rk_rep, d_vclr, d_vb, d_sync, d_cvx, d_v64, d_rk_c4};
        for (void* p : ps) ctx->release(p);
        for (void* p : extra) ctx->release(p);
    }
};

extern "C" void rv_shard_destroy(rv_shard* s) {
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    // work forked onto the second stream (the verifier's side copy of the proof, the two-stream pipeline): its buffers go back to
    // the arena below and the caller's host buffers leave scope -- nothing of it may still be in flight
    if (!s->misc_events.empty() || !s->mask_chunks.empty() || s->ev_setup || s->ec) (void)hipStreamSynchronize(s->ctx->stream2);
    if (s->overlap && s->ctx->stream_m) (void)hipStreamSynchronize(s->ctx->stream_m);
    if (s->ev_clear || s->split) {
        if (s->ctx->stream3) (void)hipStreamSynchronize(s->ctx->stream3);
        if (s->ctx->stream_x) (void)hipStreamSynchronize(s->ctx->stream_x);
    }
    s->destroy();
    delete s;
}

struct OvTrace {  // RV_OV_TRACE=1: timing events around every chunk and every group of levels, printed by overlap_trace_dump
    std::vector<std::pair<std::string, hipEvent_t>> ev;
    void mark(const char* what, uint64_t n, hipStream_t st) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, st);
        ev.emplace_back(std::string(what) + " " + std::to_string(n), e);
    }
};
static thread_local OvTrace* g_ov_trace = nullptr;
static void overlap_trace_dump() {
    if (!g_ov_trace) return;
    for (auto& kv : g_ov_trace->ev) {
        float ms = 0;
        (void)hipEventSynchronize(kv.second);
        (void)hipEventElapsedTime(&ms, g_ov_trace->ev[0].second, kv.second);
        fprintf(stderr, "[ov] %9.1f us  %s\n", ms * 1e3, kv.first.c_str());
    }
    for (auto& kv : g_ov_trace->ev) (void)hipEventDestroy(kv.second);
    delete g_ov_trace;
    g_ov_trace = nullptr;
}
// key material -> bitsliced round keys, masks
static int shard_setup_prg(rv_shard* s, const uint32_t* d_keep, const uint32_t* d_keep64 = nullptr) {
    rv_ctx* ctx = s->ctx;
    const Compiled& cc = s->c->cc;
    int rc;
    const uint64_t n_blocks64 = (cc.n_masks64 + 1) / 2;
    if (n_blocks64) {
        if ((rc = dalloc(ctx, (size_t)n_blocks64 * 2 * s->R * 8, &s->d_masks64))) return rc;
        if (s->d_keys64) {  // verifier: separate key set for the z64 transcript (reuses d_rkbytes as scratch below)
            if ((rc = dalloc(ctx, (size_t)RK_AREAS * 128 * s->NQ, &s->d_rk64))) return rc;
        }
    }
    if ((rc = dalloc(ctx, (size_t)s->R * 8 * RK_BYTES, &s->d_rkbytes))) return rc;
    if ((rc = dalloc(ctx, (size_t)RK_AREAS * 128 * s->NQ, &s->d_rk))) return rc;
    const uint64_t n_blocks = cc.n_masks_pad / 128;
    if ((rc = dalloc(ctx, (size_t)cc.n_rows * s->NQ, &s->d_masks))) return rc;
    launch_key_schedule(ctx->stream, s->d_keys, s->R * 8, s->d_rkbytes);
    launch_bitslice_rk(ctx->stream, s->d_rkbytes, s->NQ, s->d_rk);
    ctx->count(2);
    // which GF(2) generator: the lane-distributed one (aes_col4.hip) whenever it is to share the chip with the level launches
    // (RV_OVERLAP), or on request (RV_AES_COL4=1); recorded batches keep the 128-plane kernel (its launch is replayable)
    static const int col4_mode = [] {
        const char* e = getenv("RV_AES_COL4");
        return e ? atoi(e) : -1;
    }();
    const bool col4 = n_blocks && !g_recorder && aes_col4_supports(s->NQ) && (col4_mode > 0 || (col4_mode < 0 && s->overlap));
    if (s->overlap && !col4) s->overlap = false;
    if (col4) {
        if ((rc = dalloc(ctx, aes_col4_image_bytes(s->NQ) / 4, &s->d_rk_c4))) return rc;
        launch_rk_col4(ctx->stream, s->d_rk, s->NQ, s->d_rk_c4);
        ctx->count();
    }
    ctx->phase(RV_PH_MASKS);
    if (n_blocks64) {
        const uint32_t* rk64 = s->d_rk;  // prover: the same seeds feed both domains (proof/mod.rs:131-146)
        if (s->d_keys64) {
            launch_key_schedule(ctx->stream, s->d_keys64, s->R * 8, s->d_rkbytes);
            launch_bitslice_rk(ctx->stream, s->d_rkbytes, s->NQ, s->d_rk64);
            rk64 = s->d_rk64;
        }
        if (s->z64f) {
            // only the Input gates' rows: a Mul's two masks come out of the interpreter's own launches
            s->d_keep64z = s->d_keys64 ? d_keep64 : d_keep;
            for (const auto& run : s->c->z64f_runs) {
                launch_aes_z64_masks(ctx->stream, rk64, s->d_keep64z, s->NQ, run.second, s->d_masks64 + (size_t)run.first * 2 * s->R * 8, run.first);
                ctx->count(1);
            }
        } else {
            launch_aes_z64_masks(ctx->stream, rk64, s->d_keys64 ? d_keep64 : d_keep, s->NQ, n_blocks64, s->d_masks64);
            ctx->count(1);
        }
    }
    if (s->overlap) {
        // nothing is generated here: the level loop submits the chunks (overlap_need), each on ctx->stream_m behind this point
        if (!ctx->stream_m && hipStreamCreateWithFlags(&ctx->stream_m, hipStreamNonBlocking) != hipSuccess) return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
        s->ev_setup = ctx->get_sync_event();
        HIPCHK(hipEventRecord(s->ev_setup, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream_m, s->ev_setup, 0));
        static const uint64_t n_chunks = [] {
            const char* e = getenv("RV_OVERLAP_CHUNKS");
            return (uint64_t)std::min(std::max(e ? atoi(e) : 16, 1), 256);
        }();
        s->ov_blocks = n_blocks;
        s->ov_chunk = std::max<uint64_t>((n_blocks + n_chunks - 1) / n_chunks, 1024);
        s->ov_keep = d_keep;
        static const bool trace = getenv("RV_OV_TRACE") && atoi(getenv("RV_OV_TRACE")) != 0;
        if (trace) {
            overlap_trace_dump();  // (the previous proof's)
            g_ov_trace = new OvTrace();
            g_ov_trace->mark("main: keys done", 0, ctx->stream);
        }
        ctx->phase(-1);
        return RV_OK;
    }
    // every GF(2) mask before the first level, on the main stream
    if (n_blocks) {
        if (col4)
            launch_aes_gf2_masks_col4(ctx->stream, s->d_rk_c4, d_keep, s->NQ, 0, n_blocks, s->d_masks);
        else
            launch_aes_gf2_masks(ctx->stream, s->d_rk, d_keep, s->NQ, 0, n_blocks, s->d_masks, s->flat ? clear_wgs() : 0);
        ctx->count();
    }
    ctx->phase(-1);
    return RV_OK;
}

// RV_OVERLAP: make sure the masks of CTR blocks [0, need) are on their way and ordered before what the main stream queues next.
// hipStreamWaitEvent orders behind the OTHER stream's tail at the time it is queued (runtime lessons of round 4), so a
// chunk is submitted right before the first level that reads it and waited for at once: the host runs far ahead of the device,
// every chunk sits in stream_m's queue long before its turn, and the generator runs back to back beside the levels.
static int overlap_need(rv_shard* s, uint64_t need, hipStream_t sb) {
    rv_ctx* ctx = s->ctx;
    need = std::min(need, s->ov_blocks);
    while (s->ov_next < need) {
        if (g_ov_trace) g_ov_trace->mark("main: levels queued so far end; wait for chunk ending at block", s->ov_next, sb);
        if (g_ov_trace) g_ov_trace->mark("  masks: chunk start at block", s->ov_next, ctx->stream_m);
        // (the first chunk is what the proof waits for with nothing beside it: half size)
        static const uint64_t first = getenv("RV_OVERLAP_FIRST") ? strtoull(getenv("RV_OVERLAP_FIRST"), nullptr, 0) : 0;
        const uint64_t nb = std::min(s->ov_next == 0 ? (first ? first : std::max<uint64_t>(s->ov_chunk / 2, 512)) : s->ov_chunk, s->ov_blocks - s->ov_next);
        launch_aes_gf2_masks_col4(ctx->stream_m, s->d_rk_c4, s->ov_keep, s->NQ, s->ov_next, nb, s->d_masks + (size_t)s->ov_next * 128 * s->NQ);
        ctx->count();
        s->ov_next += nb;
        hipEvent_t e = ctx->get_sync_event();
        HIPCHK(hipEventRecord(e, ctx->stream_m));
        s->mask_chunks.emplace_back(s->ov_next, e);
        if (g_ov_trace) g_ov_trace->mark("  masks: chunk end at block", s->ov_next, ctx->stream_m);
        HIPCHK(hipStreamWaitEvent(sb, e, 0));
        if (g_ov_trace) g_ov_trace->mark("main: chunk arrived, levels go on; block", s->ov_next, sb);
    }
    return RV_OK;
}

// The interpreter phase of a shard in three steps (shard_run = all three; rv_prove_batch interleaves them over several
// shards): buffers + parameter blocks, the level loop, the transcript digests.
static int shard_run_alloc(rv_shard* s, InterpParams& p, Interp64Params& p64) {
    rv_ctx* ctx = s->ctx;
    const Compiled& cc = s->c->cc;
    int rc;
    if ((rc = dalloc(ctx, (size_t)cc.n_rows * (s->NQ / 2), &s->d_wires))) return rc;  // corr bits per base row
    if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_on, 1) * s->NQ, &s->d_on))) return rc;
    if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_pre, 1) * (s->NQ / 2), &s->d_pre))) return rc;
    if (!s->d_err && (rc = dalloc(ctx, 1, &s->d_err))) return rc;  // (rv_prove_batch hands every proof a slot of one array)
    const size_t cvw = b3_stream_scratch_words(std::max({cc.n_on, cc.n_pre, cc.on_words64 * 8, cc.pre_words64 * 8}), s->R);
    if ((rc = dalloc(ctx, cvw, &s->d_cv[0])) || (rc = dalloc(ctx, cvw, &s->d_cv[1]))) return rc;
    if ((rc = dalloc(ctx, (size_t)4 * s->R * 8, &s->d_dig))) return rc;
    if (!s->d_h && (rc = dalloc(ctx, (size_t)s->R * 32, &s->d_h))) return rc;  // (rv_verify_batch: a slot of one array)
    const bool has64 = !cc.gates64.empty();
    if (has64) {
        if ((rc = dalloc(ctx, (size_t)cc.n_ssa64 * s->R * 8, &s->d_wmask64))) return rc;
        if (s->z64f && !s->d_keys64) {  // (the fused PROVER: cleartext values; the verifier keeps per-repetition corrections)
            if ((rc = dalloc(ctx, (size_t)cc.n_ssa64, &s->d_v64))) return rc;
            HIPCHK(hipMemsetAsync(s->d_v64, 0, 8, ctx->stream));  // (SSA id 0 = the zero wire)
        } else {
            if ((rc = dalloc(ctx, (size_t)cc.n_ssa64 * s->R, &s->d_wcorr64))) return rc;
            HIPCHK(hipMemsetAsync(s->d_wcorr64, 0, (size_t)s->R * 8, ctx->stream));
        }
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.on_words64, 1) * s->R, &s->d_on64))) return rc;
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.pre_words64, 1) * s->R, &s->d_pre64))) return rc;
        HIPCHK(hipMemsetAsync(s->d_wmask64, 0, (size_t)s->R * 64, ctx->stream));
    }
    hipStream_t sb = ctx->stream;
    // error flag and the zero row (first computed row: mask 0, corr 0), one launch
    launch_shard_init(sb, s->d_err, s->d_masks + (size_t)cc.zero_row * s->NQ, s->NQ,
                      s->d_wires + (size_t)cc.zero_row * (s->NQ / 2), s->NQ / 2);
    p.NQ = s->NQ;
    p.rows = s->d_masks;
    p.corr = s->d_wires;
    p.on = s->d_on;
    p.pre = s->d_pre;
    p.err = s->d_err;
    p64.R = s->R;
    p64.wmask = s->d_wmask64;
    p64.wcorr = s->d_wcorr64;
    p64.masks = s->d_masks64;
    p64.on = s->d_on64;
    p64.pre = s->d_pre64;
    p64.on_words = cc.on_words64;
    p64.pre_words = cc.pre_words64;
    p64.corr2 = s->d_wires;
    p64.masks2 = s->d_masks;
    p64.NQ = s->NQ;
    p64.err = s->d_err;
    return RV_OK;
}

// Batched proofs: an LDS run takes NQ / qs workgroups per proof, each alone on a compute unit and mostly waiting on its
// own dependency chain -- better than one workgroup per proof (k_interp_narrow_b, ~3x slower per proof) only while the
// whole batch still finds room on the chip at once or nearly so (RV_LDS_BATCH_WGS overrides the limit)
// one-quad slices update nibbles of the bit-packed rows through aligned 32-bit words: rows of at least four bytes (32 repetitions)
static bool lds_run_fits_rows(uint32_t qs, uint32_t NQ) { return NQ % qs == 0 && (qs > 1 || NQ % 8 == 0); }

static bool lds_run_for_batch(const rv_circuit* c, size_t level, size_t batch) {
    if (c->lds_run_of_level[level] < 0) return false;
    const auto& pl = c->lds_runs[(size_t)c->lds_run_of_level[level]];
    const size_t limit = getenv("RV_LDS_BATCH_WGS") ? (size_t)atoll(getenv("RV_LDS_BATCH_WGS")) : 768;
    return batch * (RV_TOTAL_REPS / 4 / pl.qs) <= limit;
}

// The early-corrections plan of a circuit: per level the smallest preprocessing row any LATER level still writes (a
// Mul's row number g.ep; everything below it is final), the corrections vector cut into RV_EARLY_CHUNKS (default 4:
// 4 and 5 give the same proof time, 3 and 6 .. 10 a longer one; every chunk costs a ~25 us packing kernel)
// byte ranges, each with the level it is complete after.  Only for pure GF(2) circuits with at least RV_EARLY_MIN
// (default 2^21) Mul gates whose preprocessing rows complete roughly in step with the levels (a layered circuit; a
// circuit whose first rows are written by its last level gains nothing and keeps the plain path).
static std::atomic<uint64_t> g_overlap_commits{0};  // shard commitments whose mask generator ran beside the level launches (RV_OVERLAP)
extern "C" uint64_t rv_hook_overlap_commits(void) { return g_overlap_commits.load(std::memory_order_relaxed); }
static std::atomic<uint64_t> g_early_proofs{0};
extern "C" uint64_t rv_hook_early_proofs(void) { return g_early_proofs.load(std::memory_order_relaxed); }
static std::atomic<uint64_t> g_verify_vc{0};
extern "C" uint64_t rv_hook_verify_vc_count(void) { return g_verify_vc.load(std::memory_order_relaxed); }

// (the plan as the environment stands NOW: early_plan() below keeps the first one it builds for a circuit, rv_circuit_early_staging_bytes
// builds one of its own to answer with -- a query must not freeze the knobs the first proof would have read)
static void early_plan_build(const rv_circuit* c, EarlyPlan& P) {
    const Compiled& cc = c->cc;
    // (read per circuit, not once per process: the tests lower them)
    const uint64_t min_events = getenv("RV_EARLY_MIN") ? (uint64_t)atoll(getenv("RV_EARLY_MIN")) : (1ull << 21);
    // (the progress stamp carries the chunk count in eight bits; a bad knob gives no plan, i.e. the plain path, not a failed proof)
    const int n_chunks_env = getenv("RV_EARLY_CHUNKS") ? std::min(atoi(getenv("RV_EARLY_CHUNKS")), 255) : 4;
    auto reps_env = [](uint32_t dflt) -> uint32_t {
        const char* e = getenv("RV_EARLY_REPS");
        if (!e) return dflt;
        return (uint32_t)std::min(std::max(atoi(e), 0), (int)RV_TOTAL_REPS);
    };
    const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
    if (!cc.gates64.empty()) {
        // ---- Z64 ----
        const size_t n_lv64 = cc.level_start64.empty() ? 0 : cc.level_start64.size() - 1;
        const uint64_t min64 = getenv("RV_EARLY_MIN") ? min_events : (1ull << 17);
        if (cc.row_prg_base || cc.n_pre || cc.pre_words64 != cc.n_corr64 || cc.n_corr64 < min64 || (cc.n_corr64 & 1) || !n_lv64 || n_chunks_env < 1) return;
        std::vector<uint64_t> lo(n_lv64, UINT64_MAX);
        for (size_t l = 0; l < n_lv64; l++)
            for (uint32_t i = cc.level_start64[l]; i < cc.level_start64[l + 1]; i++) {
                const uint32_t op = cc.gates64[i].op;
                if (op == G64_B2A) return;
                if (op == G64_MUL) lo[l] = std::min<uint64_t>(lo[l], cc.gates64[i].ep);
            }
        std::vector<uint64_t> done(n_lv64);
        uint64_t m = cc.pre_words64;
        for (size_t l = n_lv64; l-- > 0;) {
            done[l] = m;
            m = std::min(m, lo[l]);
        }
        // how many repetitions' vectors fit through PCIe while the interpreter and the hashes run (rates of the 10^6-MUL
        // benchmark circuit: ~10 ns per gate, ~6 ns per Mul of hashing); RV_EARLY=2: RV_EARLY_REPS (default 128) whatever the estimate
        const uint64_t vec_bytes = 8 * cc.n_corr64;
        uint32_t r_spec;
        if (getenv("RV_EARLY") && atoi(getenv("RV_EARLY")) == 2) {
            r_spec = reps_env(128);
        } else {
            // (~19 ns per gate of interpreter + mask generator -- fused or not --, ~6 ns per Mul of hashing)
            const double t_window = (double)cc.gates64.size() * 19e-9 + (double)cc.n_corr64 * 6e-9;
            // 0.9 of what the window could carry.  Round 3 staged 128 repetitions of the benchmark circuit in FOUR chunks (more
            // made the proof slower: the last chunk, a quarter of everything, was still crossing PCIe at the challenge); in twelve
            // chunks the last one fits the hash phase and all 256 repetitions pay: 54.4 -> 52.7 (192) -> 51.8 ms (256), proofs of
            // both plans interleaved in one process (tools/z64_early_ab.py).  2 GB of page-locked staging instead of 1 GB.
            r_spec = (uint32_t)std::min<double>(RV_TOTAL_REPS, 0.9 * t_window * 55e9 / (double)vec_bytes);
        }
        r_spec = std::min<uint32_t>(r_spec, RV_TOTAL_REPS) & ~7u;
        if (r_spec < 64) return;
        const uint64_t pitch = (vec_bytes + 127) & ~127ull;
        const uint64_t K = getenv("RV_EARLY_CHUNKS") ? (uint64_t)n_chunks_env : 12;  // (Z64: twelve chunks unless told otherwise)
        const uint64_t per = ((vec_bytes + K - 1) / K + 127) & ~127ull;
        for (uint64_t b0 = 0; b0 < vec_bytes; b0 += per) {
            EarlyPlan::Chunk ch{};
            ch.byte0 = b0;
            ch.nbytes = std::min(per, vec_bytes - b0);
            ch.pitch = pitch;
            ch.off = 0;
            const uint64_t need = (ch.byte0 + ch.nbytes) / 8;
            ch.ready_level = (uint32_t)(std::lower_bound(done.begin(), done.end(), need) - done.begin());
            if (ch.ready_level >= n_lv64) return;
            const uint64_t k = P.chunks.size();
            if (ch.ready_level > n_lv64 * (k + 1) / K + n_lv64 / 4) return;
            P.chunks.push_back(ch);
        }
        P.bytes = (size_t)r_spec * pitch;
        P.z64 = true;
        P.r_spec = r_spec;
        P.ok = true;
        return;
    }
    if (cc.row_prg_base || cc.n_pre < min_events || !n_levels || n_chunks_env < 1) return;
    // The staged repetitions' vectors (1/8 byte per Mul each) must cross PCIe (~55 GB/s) while the interpreter and the hash kernels
    // run, or the copies pile up behind the challenge and the proof gets SLOWER (all 256 on the all-AND variant of the 10^7-gate
    // circuit, 320 MB against ~4.6 ms: 10.9 -> 11.0 - 11.6 ms).  So only as many repetitions as fit 0.9 of the estimated window
    // (the benchmark circuits' rates: a level launch >= 13 us and ~0.25 ns per gate, the hashes ~0.21 ns per Mul); the opened
    // repetitions beyond them are extracted and copied the plain way.  RV_EARLY=2: all of them (RV_EARLY_REPS overrides).
    uint32_t r_spec = RV_TOTAL_REPS;
    if (getenv("RV_EARLY") && atoi(getenv("RV_EARLY")) == 2) {
        r_spec = reps_env(RV_TOTAL_REPS);
    } else {
        const double t_window = std::max((double)n_levels * 13e-6, (double)cc.gates.size() * 0.25e-9) + (double)cc.n_pre * 0.21e-9 + 0.3e-3;
        r_spec = (uint32_t)std::min<double>(RV_TOTAL_REPS, 0.9 * t_window * 55e9 / ((double)cc.n_pre / 8.0));
    }
    r_spec = std::min<uint32_t>(r_spec, RV_TOTAL_REPS) & ~7u;
    if (r_spec < 64) return;
    // smallest row written per level, on a few threads (10^7 gate records are 0.4 GB)
    std::vector<uint64_t> lo(n_levels, UINT64_MAX);
    const int T = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
    auto scan = [&](size_t l0, size_t l1) {
        for (size_t l = l0; l < l1; l++) {
            uint64_t m = UINT64_MAX;
            for (uint32_t i = cc.level_start[l]; i < cc.level_start[l + 1]; i++)
                if (g_op(cc.gates[i]) == G_MUL) m = std::min<uint64_t>(m, cc.gates[i].ep);
            lo[l] = m;
        }
    };
    {
        std::vector<std::thread> th;
        size_t l0 = 0;
        for (int t = 0; t < T; t++) {
            // levels dealt by gate count
            const uint32_t want = (uint32_t)((uint64_t)cc.gates.size() * (t + 1) / T);
            size_t l1 = t + 1 == T ? n_levels : (size_t)(std::lower_bound(cc.level_start.begin(), cc.level_start.end(), want) - cc.level_start.begin());
            l1 = std::min(std::max(l1, l0), n_levels);
            if (l1 > l0) {
                if (t + 1 == T) scan(l0, l1); else th.emplace_back(scan, l0, l1);
            }
            l0 = l1;
        }
        for (auto& t : th) t.join();
    }
    // done[l] = rows final once levels 0 .. l have run
    std::vector<uint64_t> done(n_levels);
    uint64_t m = cc.n_pre;
    for (size_t l = n_levels; l-- > 0;) {
        done[l] = m;
        m = std::min(m, lo[l]);
    }
    const uint64_t l2c = cc.n_pre / 8 + 1;
    const uint64_t K = (uint64_t)n_chunks_env;
    const uint64_t per = ((l2c + K - 1) / K + 127) & ~127ull;
    size_t off = 0;
    for (uint64_t b0 = 0; b0 < l2c; b0 += per) {
        EarlyPlan::Chunk ch{};
        ch.byte0 = b0;
        ch.nbytes = std::min(per, l2c - b0);
        ch.pitch = (ch.nbytes + 127) & ~127ull;
        ch.off = off;
        off += (size_t)256 * ch.pitch;
        const uint64_t need = std::min<uint64_t>(8 * (ch.byte0 + ch.nbytes), cc.n_pre);
        ch.ready_level = (uint32_t)(std::lower_bound(done.begin(), done.end(), need) - done.begin());
        if (ch.ready_level >= n_levels) return;  // (cannot happen: done[last] = n_pre)
        const uint64_t k = P.chunks.size();
        if (ch.ready_level > n_levels * (k + 1) / K + n_levels / 4) return;  // completes too late to be worth sending ahead
        P.chunks.push_back(ch);
    }
    P.bytes = off;
    P.r_spec = r_spec;
    P.ok = true;
}
static const EarlyPlan* early_plan(const rv_circuit* c) {
    std::call_once(c->ec_once, [c] { early_plan_build(c, c->ec_plan); });
    return &c->ec_plan;
}
static uint64_t early_staging_bytes_of(const rv_circuit* c) {
    if (const char* e = getenv("RV_EARLY"))
        if (atoi(e) == 0) return 0;
    EarlyPlan P;
    early_plan_build(c, P);
    return P.ok ? (uint64_t)P.bytes : 0;
}

// Host-only view of the plan (tests): compiles the ops as rv_circuit_compile_ex would, builds the early-corrections plan and checks
// it against the gate records one by one -- no gate of a level after a chunk's ready_level may write a preprocessing row of the chunk,
// and some gate of the ready_level itself must (else the chunk could have left a level earlier).  out[0] = plan taken (0 / 1),
// [1] = Z64 form, [2] = staged repetitions, [3] = chunks, [4] = staging bytes, [5] = the check (1 = consistent), [6 + k] = chunk k's
// ready_level (k < 16).
extern "C" int rv_hook_early_plan(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, uint64_t out[22]) {
    if (!out || (n_ops && !ops) || (flags & ~RV_COMPILE_WHOLE_PROVER)) return RV_E_ARG;
    try {
        rv_circuit tmp;
        int rc = compile_ops(ops, n_ops, z64_wires, gf2_wires, tmp.cc, nullptr, ((flags & RV_COMPILE_WHOLE_PROVER) && !getenv("RV_LAZY_K")) ? RV_LIN_K : 0);
        if (rc) return rc;
        const EarlyPlan* P = early_plan(&tmp);
        for (int i = 0; i < 22; i++) out[i] = 0;
        out[0] = P->ok, out[1] = P->z64, out[2] = P->r_spec, out[3] = P->chunks.size(), out[4] = P->bytes;
        if (!P->ok) return RV_OK;
        const Compiled& cc = tmp.cc;
        bool good = true;
        for (size_t k = 0; k < P->chunks.size(); k++) {
            const auto& ch = P->chunks[k];
            if (k < 16) out[6 + k] = ch.ready_level;
            // rows (GF(2): 8 per byte; Z64: one word per 8 bytes) of the chunk
            const uint64_t row0 = P->z64 ? ch.byte0 / 8 : ch.byte0 * 8;
            const uint64_t row1 = P->z64 ? (ch.byte0 + ch.nbytes) / 8 : std::min<uint64_t>((ch.byte0 + ch.nbytes) * 8, cc.n_pre);
            bool at_ready = row1 <= row0;  // (the pad byte behind the last row belongs to no gate)
            const size_t n_levels = (P->z64 ? cc.level_start64.size() : cc.level_start.size()) - 1;
            for (size_t l = 0; l < n_levels; l++) {
                const uint32_t a = P->z64 ? cc.level_start64[l] : cc.level_start[l], b = P->z64 ? cc.level_start64[l + 1] : cc.level_start[l + 1];
                for (uint32_t i = a; i < b; i++) {
                    uint64_t ep;
                    if (P->z64) {
                        if (cc.gates64[i].op != G64_MUL) continue;
                        ep = cc.gates64[i].ep;
                    } else {
                        if (g_op(cc.gates[i]) != G_MUL) continue;
                        ep = cc.gates[i].ep;
                    }
                    if (ep < row1 && l > ch.ready_level) good = false;  // (rows below the chunk count too: chunks leave in order)
                    if (ep >= row0 && ep < row1 && l == ch.ready_level) at_ready = true;
                }
            }
            // (a chunk whose own last writer is earlier than a previous chunk's inherits that chunk's level: in-order delivery)
            if (!at_ready && !(k && ch.ready_level == P->chunks[k - 1].ready_level)) good = false;
            if (k && ch.ready_level < P->chunks[k - 1].ready_level) good = false;
            if (k && ch.byte0 != P->chunks[k - 1].byte0 + P->chunks[k - 1].nbytes) good = false;
        }
        out[5] = good;
        return RV_OK;
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

// Host-only view of the flat prover schedule (tests; csrc/flat.h): built as circuit_upload builds it, then replayed against the
// level-sorted gate stream it was made from.
extern "C" int rv_hook_flat_plan(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, uint32_t bands, uint64_t out[24]) {
#ifdef RV_EXPERIMENTS
    if (!out || (n_ops && !ops) || (flags & ~RV_COMPILE_WHOLE_PROVER)) return RV_E_ARG;
    try {
        Compiled cc;
        int rc = compile_ops(ops, n_ops, z64_wires, gf2_wires, cc, nullptr, ((flags & RV_COMPILE_WHOLE_PROVER) && !getenv("RV_LAZY_K")) ? RV_LIN_K : 0);
        if (rc) return rc;
        for (int i = 0; i < 24; i++) out[i] = 0;
        FlatPlan P;
        if (!build_flat_plan(cc, P, bands)) return RV_OK;
        out[0] = 1, out[1] = P.muls.size(), out[2] = P.xgates.size(), out[3] = P.xlevels.size(), out[4] = P.others.size();
        out[6] = P.n_clear_levels, out[7] = P.bands.size();
        for (size_t k = 0; k < P.bands.size() && k < 16; k++) out[8 + k] = P.bands[k].x1 - P.bands[k].x0;
        bool good = P.muls.size() + P.xgates.size() + P.others.size() == cc.gates.size() && !P.bands.empty();
        // replay: band after band, its x-levels in order, then its Mul range -- every computed row a gate reads must have been
        // written by an EARLIER launch, and is written exactly once
        std::vector<uint8_t> written(cc.n_rows - cc.zero_row, 0);  // 1: by an earlier launch, 2: by the launch in progress
        auto row_ready = [&](uint32_t row) { return row <= cc.zero_row || written[row - cc.zero_row] == 1; };
        uint32_t x_next = 0, mul_next = 0;
        for (const auto& B : P.bands) {
            if (!good) break;
            good = B.x0 == x_next && B.x1 >= B.x0 && B.x1 <= P.xlevels.size() && B.mul0 == mul_next && B.mul1 >= B.mul0 && B.mul1 <= P.muls.size() &&
                   B.mul0 % 1024 == 0;
            if (!good) break;
            for (uint32_t l = B.x0; l < B.x1 && good; l++) {
                const LevelRange& r = P.xlevels[l];
                good = r.lo == r.mul && r.xork == r.hi && (l == 0 ? r.lo == 0 : r.lo == P.xlevels[l - 1].hi) && r.hi > r.lo;
                for (uint32_t i = r.lo; i < r.hi && good; i++) {
                    const Gate& g = P.xgates[i];
                    good = g_op(g) == G_XORK && g.dst > cc.zero_row && !written[g.dst - cc.zero_row] && (g_na(g) == 2 && g_nb(g) == 0) == (i < r.xor2);
                    for (int k = 0; k < RV_LIN_K && good; k++) good = row_ready(g.a[k]) && row_ready(g.b[k]);
                    if (good) written[g.dst - cc.zero_row] = 2;
                }
                for (uint32_t i = r.lo; i < r.hi && good; i++) written[P.xgates[i].dst - cc.zero_row] = 1;
            }
            for (uint32_t i = B.mul0; i < B.mul1 && good; i++)
                for (int k = 0; k < RV_LIN_K && good; k++) good = row_ready(P.muls[i].a[k]) && row_ready(P.muls[i].b[k]);
            x_next = B.x1, mul_next = B.mul1;
        }
        good = good && x_next == P.xlevels.size() && mul_next == P.muls.size() &&
               (P.xlevels.empty() || P.xlevels.back().hi == P.xgates.size());
        for (const Gate& g : P.others)
            for (int k = 0; k < RV_LIN_K && good; k++) good = row_ready(g.a[k]);  // (AssertZero rows: behind every band)
        // Mul record i is the Mul gate with preprocessing row i, field by field
        size_t n_mul = 0, n_oth = 0;
        for (const Gate& g : cc.gates) {
            if (!good) break;
            const uint32_t op = g_op(g);
            if (op == G_MUL) {
                const MulRec& r = P.muls[g.ep];
                good = r.m == g.m && (r.eo_flags & MULREC_EO_MASK) == g.eo && ((r.eo_flags >> 30) & 1u) == g_ca(g) && (r.eo_flags >> 31) == g_cb(g) &&
                       ((r.eo_flags >> 26) & 3u) == std::max(g_na(g), 1u) - 1 && ((r.eo_flags >> 28) & 3u) == std::max(g_nb(g), 1u) - 1;
                for (int k = 0; k < RV_LIN_K; k++) good = good && r.a[k] == g.a[k] && r.b[k] == g.b[k];
                n_mul++;
            } else if (op != G_XORK) {
                n_oth++;
            }
        }
        good = good && n_mul == P.muls.size() && n_oth == P.others.size();
        out[5] = good;
        return RV_OK;
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
#else
    (void)ops, (void)n_ops, (void)z64_wires, (void)gf2_wires, (void)flags, (void)bands, (void)out;
    g_last_error = "rv_hook_flat_plan: the flat schedule exists in experiment builds only (make EXTRA=-DRV_EXPERIMENTS)";
    return RV_E_UNSUPPORTED;
#endif
}

// Early corrections, device side.  early_flush (called by the level loop) puts, behind the level that completes a chunk, the
// packing kernel and a stamp kernel into the interpreter's own stream; the host -- idle once a proof is queued -- sees the
// stamp in the mapped mailbox and hands the chunk's copy to the second stream (early_pump), which therefore only ever
// holds copy-engine work and the "arrived" stamps.  Measured on the 10^7-gate circuit (tools/early_ab.py, gpurun_out/e*.log):
//  * packing kernels on the second stream are not dispatched while the first stream issues its short level launches back
//    to back (kernel trace: the first one starts when the hash kernels do), so the copies piled up behind the challenge;
//    in the interpreter's stream they cost 6 x 26 us (RV_EARLY_PACK_STREAM=0 / 2: the old placements);
//  * HIP events instead of stamps (hipEventRecord + hipStreamWaitEvent, or hipEventQuery from the host) were no cheaper, and
//    neither was a second stream of the highest priority;
//  * the packing kernel takes ~27 us per 27 MB chunk whatever its instruction count (a lane per repetition with multiplications,
//    or the 8 x 8 bit transposes it has now: 2 300 vs 700 instructions per thread): one generation of workgroups that all
//    load, then all store; its tiles as extra workgroups at the end of the level launches' grids (k_interp_full with a packing
//    branch, built and measured: byte-identical) cost the interpreter the same ~0.15 ms -- it has no idle issue slots to give;
//  * NOTHING may wait on the second stream.  Its copies were first followed by an "arrived" stamp kernel each (before that by an
//    event): a packet that waits for the copy engine's signal at the head of another hardware queue is polled by the command
//    processor between the first queue's level launches, and the 163 launches paid 0.2 ms for it (interpreter phase 2.40 - 2.45 ms
//    against 2.17 - 2.2 now; tools/copy_beside.py: copy-engine transfers alone beside the levels cost 0.03 ms).  The host waits for
//    the stream itself after the challenge -- a signal wait on the host side, no packet.
static int early_flush_chunks(rv_shard* s, size_t last) {
    EarlyRun* e = s->ec;
    rv_ctx* ctx = s->ctx;
    const auto& chunks = e->plan->chunks;
    if (e->next >= chunks.size() || last <= e->next) return RV_OK;
    static const int pack_stream = getenv("RV_EARLY_PACK_STREAM") ? atoi(getenv("RV_EARLY_PACK_STREAM")) : 1;  // 1: every chunk in-stream, 0: the last one, 2: none
    const size_t first = e->next;
    for (size_t k = first; k < last; k++) {
        const bool in_stream = pack_stream == 1 || (pack_stream != 2 && k + 1 == chunks.size());
        const auto& ch = chunks[k];
        // (these launches sit inside the interpreter's phase but are not level launches: rv_profile counts them in slot 6)
        if (e->plan->z64) {
            e->packed.push_back(1);  // nothing to pack: the rows are the vectors
            continue;
        }
        if (in_stream) {
            launch_pack_corr_all(ctx->stream, s->d_pre, s->c->cc.n_pre, ch.byte0, ch.nbytes, ch.pitch, e->d_ec + ch.off);
            if (ctx->profiling) ctx->prof.launches[RV_PH_EARLY]++;
        }
        e->packed.push_back(in_stream ? 1 : 0);
    }
    launch_publish(ctx->stream, nullptr, 0, nullptr, e->box_dev + 1, (e->seq << 8) | (uint32_t)last);
    if (ctx->profiling) ctx->prof.launches[RV_PH_EARLY]++;
    e->next = last;
    return RV_OK;
}
// level-synchronous schedule: the chunks whose preprocessing rows are final once `levels_queued` levels are in the stream
static int early_flush(rv_shard* s, size_t levels_queued) {
    EarlyRun* e = s->ec;
    const auto& chunks = e->plan->chunks;
    if (e->next >= chunks.size() || chunks[e->next].ready_level >= levels_queued) return RV_OK;
    size_t last = e->next;
    while (last < chunks.size() && chunks[last].ready_level < levels_queued) last++;
    return early_flush_chunks(s, last);
}
#ifdef RV_EXPERIMENTS
// flat schedule: the chunks that lie inside the first `muls_queued` Mul gates of the program (preprocessing row = Mul ordinal)
static int early_flush_muls(rv_shard* s, uint64_t muls_queued) {
    EarlyRun* e = s->ec;
    const auto& chunks = e->plan->chunks;
    size_t last = e->next;
    while (last < chunks.size() && std::min<uint64_t>(8 * (chunks[last].byte0 + chunks[last].nbytes), s->c->cc.n_pre) <= muls_queued) last++;
    return early_flush_chunks(s, last);
}
#endif

// The host's waits of the early-corrections path (a chunk's stamp, the challenge): the mailbox is written by the GPU, so there is
// nothing to block on -- but the caller need not burn a core for the milliseconds a proof takes either.  The wait SLEEPS through
// most of what the same wait took last time on this context (`ema_us`, a running average per kind of wait: proofs of one circuit
// repeat their timing to a few percent), then spins for the rest, so the word is seen as promptly as before; RV_EARLY_SPIN=1 spins
// all the way.  Bounded: RV_EARLY_TIMEOUT_MS (default 20 000) without the awaited word is RV_E_DEVICE with a message, not a hang.
template <class Pred>
static int mailbox_wait(rv_ctx* ctx, Pred arrived, double* ema_us, const char* what) {
    static const bool spin_only = getenv("RV_EARLY_SPIN") && atoi(getenv("RV_EARLY_SPIN")) != 0;
    static const long timeout_ms = getenv("RV_EARLY_TIMEOUT_MS") ? std::max(atol(getenv("RV_EARLY_TIMEOUT_MS")), 1l) : 20000;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    if (!spin_only && ema_us && *ema_us > 200.0) {
        // sleep until ~80 % of the expected wait (minus the scheduler's slack) is over, in naps short enough to notice an early word
        const double until = 0.8 * *ema_us - 80.0;
        while (!arrived()) {
            const double left = until - elapsed_us();
            if (left < 30.0) break;
            timespec ts{0, (long)(std::min(left, 250.0) * 1000.0)};
            nanosleep(&ts, nullptr);
        }
    }
    int rc = RV_OK;
    for (uint64_t spins = 0; !arrived(); spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFF) == 0xFFFF) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q != hipErrorNotReady && !arrived()) {
                rc = q == hipSuccess ? RV_E_DEVICE : hip_fail(q, what, __FILE__, __LINE__);
                break;
            }
            if (elapsed_us() > 1e3 * (double)timeout_ms) {
                g_last_error = std::string(what) + ": no word from the device within RV_EARLY_TIMEOUT_MS";
                rc = RV_E_DEVICE;
                break;
            }
        }
    }
    if (rc == RV_OK && ema_us) *ema_us = *ema_us > 0 ? 0.7 * *ema_us + 0.3 * elapsed_us() : elapsed_us();
    return rc;
}

// host side: waits for every chunk's stamp in turn and queues its packing kernel (unless done), its copy to the host and the
// stamp that says it has arrived
static int early_pump(rv_shard* s) {
    EarlyRun* e = s->ec;
    rv_ctx* ctx = s->ctx;
    const auto& chunks = e->plan->chunks;
    for (; e->pumped < chunks.size(); e->pumped++) {
        const size_t k = e->pumped;
        if (k >= e->packed.size()) return RV_E_DEVICE;
        if (int rcw = mailbox_wait(ctx, [&] { return e->ready(k); }, k < 16 ? &ctx->ec_wait_us[k] : nullptr, "early corrections (pump)")) return rcw;
        const auto& ch = chunks[k];
        if (e->plan->z64) {
            // word range [byte0, byte0 + nbytes) of the first r_spec repetitions' preprocessing rows (16-byte multiples on both
            // sides: the copy engine's fast 2-D path)
            HIPCHK(hipMemcpy2DAsync(e->h_ec + ch.byte0, ch.pitch, (const uint8_t*)s->d_pre64 + ch.byte0, (size_t)s->c->cc.pre_words64 * 8, ch.nbytes,
                                    e->plan->r_spec, hipMemcpyDeviceToHost, ctx->stream2));
        } else {
            if (!e->packed[k]) launch_pack_corr_all(ctx->stream2, s->d_pre, s->c->cc.n_pre, ch.byte0, ch.nbytes, ch.pitch, e->d_ec + ch.off);
            HIPCHK(hipMemcpyAsync(e->h_ec + ch.off, e->d_ec + ch.off, (size_t)e->plan->r_spec * ch.pitch, hipMemcpyDeviceToHost, ctx->stream2));
        }
        // (NO kernel or event behind the copy: a packet that waits for the copy engine's signal at the head of the second queue is
        // polled by the command processor between the first queue's level launches and costs the interpreter 0.15 - 0.2 ms per proof --
        // tools/copy_beside.py: copies alone beside the levels cost 0.03.  The host waits for the stream instead, after the challenge.)
    }
    return RV_OK;
}

#ifdef RV_EXPERIMENTS  // the prover schedules that measured slower than the level path: experiment builds only
// RV_PERSIST: 1 = the levels of a MODE_PROVE_V run go through k_interp_persist (no launch per level), 0 (default) = one launch per level.
// Off: byte-identical, but the in-launch hand-off between levels (arrival counters + polling) costs more than the launch boundary
// it replaces ("Persistent level kernel").
static int persist_mode() {
    const char* e = getenv("RV_PERSIST");
    return e ? atoi(e) : 0;
}
// the circuit's step table for rows of NQ quad words, on the device (built at first use; the upload is queued on the context's stream
// ahead of the launch that reads it)
static const rv_circuit::PersistTab* persist_table(rv_ctx* ctx, const rv_circuit* c, uint32_t NQ) {
    std::lock_guard<std::mutex> lk(c->persist_mu);
    auto it = c->persist_tab.find(NQ);
    if (it != c->persist_tab.end()) return it->second.d ? &it->second : nullptr;
    rv_circuit::PersistTab& T = c->persist_tab[NQ];
    const size_t n = c->cc.level_range.size();
    T.h.resize(n);
    build_persist_levels(c->cc.level_range.data(), n, NQ, c->persist_gen, T.h.data());
    void* d = nullptr;
    if (ctx->alloc(std::max<size_t>(n, 1) * sizeof(PLevel), &d)) return nullptr;
    if (n && hipMemcpyAsync(d, T.h.data(), n * sizeof(PLevel), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        ctx->release(d);
        return nullptr;
    }
    T.d = (PLevel*)d;
    return &T;
}

// The flat schedule (flat.h): band after band the XOR rows x-level by x-level, then the band's Mul gates in program order;
// the Input / AssertZero transcript rows behind the last band.  The cleartext pass (queued on stream3 at commit time) must
// have ended before the first Mul launch.
static int shard_run_flat(rv_shard* s, const InterpParams& p) {
    rv_ctx* ctx = s->ctx;
    const rv_circuit* c = s->c;
    const FlatPlan& F = c->flat;
    hipStream_t st = ctx->stream;
    // RV_FLAT_XSTREAM (default 1): the XOR rows run on the second stream, band after band, AHEAD of the Mul launches of the main
    // stream (band b's Mul gates wait for band b's XOR rows only): the ~140 short, latency-bound x-level launches of the
    // 10^7-gate circuit then sit beside the eight long Mul launches instead of between them
    static const bool xstream = !(getenv("RV_FLAT_XSTREAM") && atoi(getenv("RV_FLAT_XSTREAM")) == 0);
    const bool two = xstream && F.bands.size() > 1;
    hipStream_t sx = two ? ctx->stream_x : st;
    ctx->phase(RV_PH_INTERP, st);
    int rc;
    if (two) {  // the second stream starts behind everything queued so far (masks, the zero row)
        hipEvent_t e = ctx->get_sync_event();
        s->misc_events.push_back(e);
        HIPCHK(hipEventRecord(e, st));
        HIPCHK(hipStreamWaitEvent(sx, e, 0));
    }
    bool joined = false;
    // (host order: band b's XOR launches and their event, THEN the main stream's wait for it -- the runtime resolves a wait for an
    // event of another stream to that stream's tail at the time the wait is queued: with every band's XOR launches queued first,
    // band 0's Mul gates waited for the last band's XOR rows)
    for (size_t b = 0; b < F.bands.size(); b++) {
        const auto& B = F.bands[b];
        for (uint32_t x = B.x0; x < B.x1; x++) {
            launch_interp(sx, MODE_PROVE_F, c->d_xgates, F.xlevels[x], p, nullptr);
            ctx->count();
        }
        if (two && B.x1 > B.x0) {
            hipEvent_t e = ctx->get_sync_event();
            s->misc_events.push_back(e);
            HIPCHK(hipEventRecord(e, sx));
            HIPCHK(hipStreamWaitEvent(st, e, 0));
        }
        if (!joined) {
            HIPCHK(hipStreamWaitEvent(st, s->ev_clear, 0));
            launch_or_word(st, s->d_err, (const int*)(s->d_sync + 2));
            joined = true;
        }
        if (B.mul1 > B.mul0) {
            launch_mul_flat(st, s->NQ, c->d_muls, B.mul0, B.mul1, p.rows, p.on, p.pre, s->d_vclr);
            ctx->count();
        }
        if (s->ec && (rc = early_flush_muls(s, B.mul1))) return rc;
    }
    if (!joined) {
        HIPCHK(hipStreamWaitEvent(st, s->ev_clear, 0));
        launch_or_word(st, s->d_err, (const int*)(s->d_sync + 2));
    }
    if (c->n_others) {
        launch_interp(st, MODE_PROVE_F, c->d_others, LevelRange{0, 0, 0, 0, 0, c->n_others}, p, nullptr);
        ctx->count();
    }
    if (s->ec && (rc = early_flush_muls(s, c->cc.n_pre))) return rc;
    HIPCHK(hipGetLastError());
    return RV_OK;
}

// MODE_PROVE_V without a launch per level: the levels in as few k_interp_persist launches as the early-corrections chunks allow
// (a chunk's packing kernel sits behind the level that completes it)
static int shard_run_persist(rv_shard* s, const InterpParams& p) {
    rv_ctx* ctx = s->ctx;
    const rv_circuit* c = s->c;
    const Compiled& cc = c->cc;
    const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
    const rv_circuit::PersistTab* T = persist_table(ctx, c, s->NQ);
    if (!T) return RV_E_NOMEM;
    // segment ends: after the level that completes a chunk (early corrections), and after the last level
    std::vector<uint32_t> cuts;
    if (s->ec)
        for (const auto& ch : s->ec->plan->chunks)
            if (ch.ready_level + 1 < n_levels && (cuts.empty() || cuts.back() != ch.ready_level + 1)) cuts.push_back(ch.ready_level + 1);
    cuts.push_back((uint32_t)n_levels);
    int rc;
    if ((rc = dalloc(ctx, cuts.size() * PERSIST_SYNC_WORDS, &s->d_sync))) return rc;
    HIPCHK(hipMemsetAsync(s->d_sync, 0, cuts.size() * PERSIST_SYNC_WORDS * 4, ctx->stream));
    const bool flow = persist_mode() >= 2;
    ctx->phase(RV_PH_INTERP, ctx->stream);
    if (flow) {
        // dataflow form: every value byte starts as "not ready", the zero row's as ready with value 0
        HIPCHK(hipMemsetAsync(s->d_vclr, 0, (size_t)cc.n_rows, ctx->stream));
        HIPCHK(hipMemsetAsync(s->d_vclr + cc.zero_row, 0x80, 1, ctx->stream));
    }
    uint32_t l0 = 0;
    for (size_t k = 0; k < cuts.size(); k++) {
        const uint32_t l1 = cuts[k];
        if (s->ec && (rc = early_flush(s, l0))) return rc;
        if (l1 > l0) {
            const uint64_t n_steps = (uint64_t)T->h[l1 - 1].step0 + T->h[l1 - 1].n_steps - T->h[l0].step0;
            launch_interp_persist(ctx->stream, s->NQ, c->persist_gen, c->d_gates, T->d, l0, l1, n_steps, p, s->d_sync + k * PERSIST_SYNC_WORDS, flow);
            ctx->count();
        }
        l0 = l1;
    }
    if (s->ec && (rc = early_flush(s, n_levels))) return rc;
    HIPCHK(hipGetLastError());
    return RV_OK;
}

// The split schedule (flat.h): the dependency levels as a CHAIN of light launches on a stream of its own -- per level the XOR
// gates and the other gates' cleartext values -- and, a band behind it on the main stream, the Mul gates in program order.
static int shard_run_split(rv_shard* s, const InterpParams& p) {
    rv_ctx* ctx = s->ctx;
    const rv_circuit* c = s->c;
    const Compiled& cc = c->cc;
    const FlatPlan& F = c->flat;
    hipStream_t st = ctx->stream;
    static const bool xstream = !(getenv("RV_FLAT_XSTREAM") && atoi(getenv("RV_FLAT_XSTREAM")) == 0);
    hipStream_t sx = xstream ? ctx->stream_x : st;
    const uint32_t n_levels = (uint32_t)F.n_clear_levels;
    ctx->phase(RV_PH_INTERP, st);
    int rc;
    auto fork = [&](hipStream_t from, hipStream_t to) -> int {
        if (from == to) return RV_OK;
        hipEvent_t e = ctx->get_sync_event();
        s->misc_events.push_back(e);
        HIPCHK(hipEventRecord(e, from));
        HIPCHK(hipStreamWaitEvent(to, e, 0));
        return RV_OK;
    };
    if ((rc = fork(st, sx))) return rc;  // the chain starts behind everything queued so far (masks, the zero row, the error word)
    uint32_t l = 0;
    // RV_FLAT=3: a band's levels in ONE launch on one XCD (k_chain) instead of a launch per level
    const bool one_xcd = flat_mode() == 3 && chain_supports(s->NQ);
    uint32_t* d_chain = nullptr;  // [0] the proof's chain XCD, [64] abort word, from [128] on two counters per level
    if (one_xcd) {
        if ((rc = dalloc(ctx, (size_t)128 + 2 * (size_t)n_levels, &s->d_sync))) return rc;
        d_chain = s->d_sync;
        HIPCHK(hipMemsetAsync(d_chain, 0, ((size_t)128 + 2 * (size_t)n_levels) * 4, st));
        HIPCHK(hipMemsetAsync(d_chain, 0xFF, 4, st));
        if ((rc = fork(st, sx))) return rc;
    }
    static const uint32_t chain_wgs = getenv("RV_CHAIN_WGS") ? (uint32_t)std::max(atoi(getenv("RV_CHAIN_WGS")), 1) : 512u;
    auto chain_to = [&](uint32_t l1) {
        if (one_xcd) {
            if (l1 > l) {
                launch_chain(sx, chain_wgs, c->chain_gen, c->d_gates, c->d_chain_levels, c->d_lite_levels, c->d_lite_s, c->d_lite_k, l, l1, d_chain, d_chain + 128 + 2 * l,
                             d_chain + 64, p);
                ctx->count();
            }
            l = std::max(l, l1);
            return;
        }
        for (; l < l1; l++) {
            if (cc.level_start[l + 1] == cc.level_start[l]) continue;
            launch_level_split(sx, c->d_gates, cc.level_range[l], F.lite_levels[l], c->d_lite_s, c->d_lite_k, p);
            ctx->count();
        }
    };
    // the Input gates' transcript rows depend on nothing: first, so that the online transcript completes from its head on
    if (c->n_other_inputs) {
        launch_interp(st, MODE_PROVE_F, c->d_others, LevelRange{0, 0, 0, 0, 0, c->n_other_inputs}, p, nullptr);
        ctx->count();
    }
    // RV_SPLIT_HASH (default 1): the BLAKE3 chunks of both transcripts that a band completes are hashed right behind the band's
    // Mul gates, on the main stream -- VALU-bound work in what is otherwise the shadow of the (latency-bound) level chain
    static const bool split_hash = !(getenv("RV_SPLIT_HASH") && atoi(getenv("RV_SPLIT_HASH")) == 0);
    const uint64_t pre_chunks = cc.n_pre == 0 ? 1 : (cc.n_pre + 1023) / 1024, on_chunks = cc.n_on == 0 ? 1 : (cc.n_on + 1023) / 1024;
    s->split_hash = split_hash && !s->d_on_quads && pre_chunks > 1 && on_chunks > 1;
    if (s->split_hash && (rc = dalloc(ctx, b3_stream_scratch_words(std::max(cc.n_on, cc.n_pre), s->R), &s->d_cvx))) return rc;
    auto hash_upto = [&](uint64_t pre_rows, uint64_t on_rows) {
        if (!s->split_hash) return;
        const uint64_t pc = std::min(pre_rows / 1024, pre_chunks - 1), oc = std::min(on_rows / 1024, on_chunks - 1);  // (the last chunk: shard_run_hash)
        if (pc > s->pre_chunks_done) {
            launch_b3_stream_bits_chunks(st, s->d_pre + s->pre_chunks_done * 1024 * (s->NQ / 2), (pc - s->pre_chunks_done) * 1024, s->NQ,
                                         s->d_cv[0] + s->pre_chunks_done * s->R * 8, s->pre_chunks_done, 0);
            s->pre_chunks_done = pc;
            ctx->count();
        }
        if (oc > s->on_chunks_done) {
            launch_b3_stream_chunks(st, s->d_on + s->on_chunks_done * 1024 * s->NQ, (oc - s->on_chunks_done) * 1024, s->NQ, s->d_cv[1] + s->on_chunks_done * s->R * 8,
                                    nullptr, 0, s->on_chunks_done, 0);
            s->on_chunks_done = oc;
            ctx->count();
        }
    };
    // (host order matters: a wait for another stream's event resolves to that stream's tail when the wait is queued)
    for (const auto& B : F.bands) {
        chain_to(std::min(B.level_end, n_levels));
        if ((rc = fork(sx, st))) return rc;
        if (B.mul1 > B.mul0) {
            launch_mul_flat(st, s->NQ, c->d_muls, B.mul0, B.mul1, p.rows, p.on, p.pre, s->d_vclr);
            ctx->count();
        }
        if (s->ec && (rc = early_flush_muls(s, B.mul1))) return rc;
        hash_upto(B.mul1, B.on_end);
    }
    chain_to(n_levels);
    if ((rc = fork(sx, st))) return rc;
    if (one_xcd && getenv("RV_CHAIN_DEBUG")) {
        uint32_t w[3] = {0, 0, 0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(w, d_chain, sizeof w, hipMemcpyDeviceToHost);
        fprintf(stderr, "[rv chain] XCD %u, workgroup launches that took part / left: %u / %u\n", w[0], w[1], w[2]);
    }
    if (c->n_others > c->n_other_inputs) {  // the AssertZero transcript rows
        launch_interp(st, MODE_PROVE_F, c->d_others + c->n_other_inputs, LevelRange{0, 0, 0, 0, 0, c->n_others - c->n_other_inputs}, p, nullptr);
        ctx->count();
    }
    if (s->ec && (rc = early_flush_muls(s, cc.n_pre))) return rc;
    HIPCHK(hipGetLastError());
    return RV_OK;
}
#endif  // RV_EXPERIMENTS

static int shard_run_levels(rv_shard* s, int mode, const InterpParams& p, const Interp64Params& p64) {
#ifdef RV_EXPERIMENTS
    if (s->split) return shard_run_split(s, p);
    if (s->flat) return shard_run_flat(s, p);
    if (mode == MODE_PROVE_V && persist_mode() && persist_supports(s->NQ) && !g_recorder) return shard_run_persist(s, p);
#endif
    rv_ctx* ctx = s->ctx;
    const Compiled& cc = s->c->cc;
    const bool has64 = !cc.gates64.empty();
    hipStream_t sb = ctx->stream;
    const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
    ctx->phase(RV_PH_INTERP, sb);
    size_t waited = 0;  // mask chunks already waited for
    int rc_ec;
    const uint32_t n_qg64 = s->NQ / 16;
    auto fused_params = [&](uint32_t qg0, uint32_t qgn) {
        Z64FParams zp{};
        zp.rk = s->d_keys64 ? s->d_rk64 : s->d_rk;
        zp.NQ = s->NQ;
        zp.wmask = p64.wmask;
        zp.masks = s->d_masks64;
        zp.on = p64.on;
        zp.pre = p64.pre;
        zp.on_words = p64.on_words;
        zp.pre_words = p64.pre_words;
        zp.wit = p64.wit;
        zp.v = s->d_v64;
        zp.err = p64.err;
        zp.first_block = 0;
        zp.qg0 = qg0;
        zp.qgn = qgn;
        if (mode == MODE_VERIFY) {
            zp.omit = p64.omit;
            zp.keep = s->d_keep64z;
            zp.wcorr = p64.wcorr;
            zp.sup_in = p64.sup_in;
            zp.sup_corr = p64.sup_corr;
            zp.sup_rec = p64.sup_rec;
            zp.sup_r = p64.sup_r;
        }
        return zp;
    };
    if (s->z64f && s->ev_sup64 && mode == MODE_VERIFY) {
        // (api: rv_verify_shard_impl, split64) the quad groups without an opened repetition first, every level; then, once the
        // supplied values are there, the first one
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                int rcm = s->mid64 ? s->mid64() : RV_OK;  // (the proof's copy and the unpack kernels, side stream)
                s->mid64 = nullptr;
                if (rcm) return rcm;
                HIPCHK(hipStreamWaitEvent(sb, s->ev_sup64, 0));
            }
            const Z64FParams zp = pass == 0 ? fused_params(1, n_qg64 - 1) : fused_params(0, 1);
            for (size_t l = 0; l < n_levels; l++)
                if (cc.level_start64[l + 1] > cc.level_start64[l]) {
                    launch_z64_fused(sb, s->c->d_gates64f, s->c->z64f_levels[l], zp);
                    ctx->count();
                }
        }
        return RV_OK;
    }
    for (size_t l = 0; l < n_levels; l++) {
        if (s->ec && (rc_ec = early_flush(s, l))) return rc_ec;
        if (s->overlap) {
            // (a run of levels in one launch needs its last level's masks)
            size_t l_need = l;
            if (s->c->lds_run_of_level[l] >= 0) l_need = std::max(l_need, (size_t)s->c->lds_runs[(size_t)s->c->lds_run_of_level[l]].run.l1 - 1);
            if (s->c->run_of_level[l] >= 0) l_need = std::max(l_need, (size_t)s->c->narrow_runs[(size_t)s->c->run_of_level[l]].second - 1);
            if ((rc_ec = overlap_need(s, cc.level_need_blocks[l_need], sb))) return rc_ec;
            waited = s->mask_chunks.size();
        }
        while (waited < s->mask_chunks.size() &&
               (waited == 0 ? 0 : s->mask_chunks[waited - 1].first) < cc.level_need_blocks[l]) {
            HIPCHK(hipStreamWaitEvent(sb, s->mask_chunks[waited].second, 0));
            waited++;
        }
        const bool own_launch = mode == MODE_PROVE_V || mode == MODE_VERIFY_C;  // (these modes exist in the one-launch-per-level kernel only)
        if (!own_launch && s->c->lds_run_of_level[l] >= 0 && lds_run_fits_rows(s->c->lds_runs[(size_t)s->c->lds_run_of_level[l]].qs, p.NQ)) {
            // a narrow stretch with its live wires in LDS: one launch, NQ / qs workgroups
            const auto& pl = s->c->lds_runs[(size_t)s->c->lds_run_of_level[l]];
            if (l == pl.run.l0) {
                while (waited < s->mask_chunks.size() &&
                       (waited == 0 ? 0 : s->mask_chunks[waited - 1].first) < cc.level_need_blocks[pl.run.l1 - 1]) {
                    HIPCHK(hipStreamWaitEvent(sb, s->mask_chunks[waited].second, 0));
                    waited++;
                }
                launch_interp_lds(sb, mode, pl.qs, p.NQ, s->c->d_lds_recs + pl.run.rec0, pl.run.n_steps, pl.run.n_slots, pl.run.eo0, pl.run.ep0, p, nullptr, 1);
                ctx->count();
            }
            continue;
        }
        if (s->c->run_of_level[l] >= 0 && !own_launch) {
            // a run of narrow levels: one launch for the whole run (its mask needs were waited for above
            // level by level as the loop advances, so wait for the run's last level first)
            const auto& run = s->c->narrow_runs[(size_t)s->c->run_of_level[l]];
            if (l == run.first) {
                while (waited < s->mask_chunks.size() &&
                       (waited == 0 ? 0 : s->mask_chunks[waited - 1].first) < cc.level_need_blocks[run.second - 1]) {
                    HIPCHK(hipStreamWaitEvent(sb, s->mask_chunks[waited].second, 0));
                    waited++;
                }
                launch_interp_narrow(sb, mode, s->c->d_gates, s->c->d_level_range, run.first, run.second, run.tiny, p);
                ctx->count();
            }
            continue;
        }
        if (cc.level_start[l + 1] > cc.level_start[l]) {
            // the level that follows as a launch of its own (not a narrow run) gets its first gate records prefetched
            const LevelRange* next = (l + 1 < n_levels && (s->c->run_of_level[l + 1] < 0 || own_launch) && cc.level_start[l + 2] > cc.level_start[l + 1])
                                         ? &cc.level_range[l + 1]
                                         : nullptr;
            launch_interp(sb, mode, s->c->d_gates, cc.level_range[l], p, next);
            ctx->count();
        }
        if (has64 && cc.level_start64[l + 1] > cc.level_start64[l]) {
            if (s->z64f) {
                const Z64FParams zp = fused_params(0, n_qg64);
                launch_z64_fused(sb, s->c->d_gates64f, s->c->z64f_levels[l], zp);
            } else {
                launch_interp64(sb, mode, s->c->d_gates64, cc.level_start64[l], cc.level_start64[l + 1], p64);
            }
            ctx->count();
        }
    }
    if (s->ec && (rc_ec = early_flush(s, n_levels))) return rc_ec;
    if (s->overlap && (rc_ec = overlap_need(s, s->ov_blocks, sb))) return rc_ec;  // (masks no level reads: padding)
    if (s->overlap) g_overlap_commits.fetch_add(1, std::memory_order_relaxed);
    if (g_ov_trace) g_ov_trace->mark("main: last level done", n_levels, sb);
    return RV_OK;
}

static int shard_run_hash(rv_shard* s) {
    rv_ctx* ctx = s->ctx;
    const Compiled& cc = s->c->cc;
    ctx->phase(RV_PH_HASH);
    uint32_t* dig = s->d_dig;
    const size_t DW = (size_t)s->R * 8;
    uint32_t n_launch;
    static const bool pair_on = !(getenv("RV_B3_PAIR") && atoi(getenv("RV_B3_PAIR")) == 0);
    if (s->split_hash) {
        // the bands hashed their chunks behind their Mul gates (shard_run_split): what is left of the two streams, then the trees
        const uint64_t pre_chunks = (cc.n_pre + 1023) / 1024, on_chunks = (cc.n_on + 1023) / 1024;
        launch_b3_stream_bits_chunks(ctx->stream, s->d_pre + s->pre_chunks_done * 1024 * (s->NQ / 2), cc.n_pre - s->pre_chunks_done * 1024, s->NQ,
                                     s->d_cv[0] + s->pre_chunks_done * s->R * 8, s->pre_chunks_done, 0);
        launch_b3_stream_chunks(ctx->stream, s->d_on + s->on_chunks_done * 1024 * s->NQ, cc.n_on - s->on_chunks_done * 1024, s->NQ,
                                s->d_cv[1] + s->on_chunks_done * s->R * 8, nullptr, 0, s->on_chunks_done, 0);
        n_launch = 2 + b3_reduce_tree(ctx->stream, s->d_cv[0], s->d_cvx, pre_chunks, s->R, dig + 0 * DW);
        n_launch += b3_reduce_tree(ctx->stream, s->d_cv[1], s->d_cvx, on_chunks, s->R, dig + 1 * DW);
    } else if (pair_on && launch_b3_pair_small(ctx->stream, s->d_pre, cc.n_pre, s->d_on, cc.n_on, s->NQ, s->d_cv[0], s->d_cv[1], dig + 0 * DW, dig + 1 * DW,
                                        s->d_on_quads, s->n_on_quads)) {
        n_launch = 2;  // short transcripts (small circuits): both streams in the same two launches
    } else {
        n_launch = launch_b3_stream_bits(ctx->stream, s->d_pre, cc.n_pre, s->NQ, s->d_cv[0], s->d_cv[1], dig + 0 * DW);
        n_launch += launch_b3_stream(ctx->stream, s->d_on, cc.n_on, s->NQ, s->d_cv[0], s->d_cv[1], dig + 1 * DW, s->d_on_quads, s->n_on_quads);
    }
    // Z64 transcripts; for a pure GF(2) circuit both are empty and every digest is BLAKE3("") (one fill, not four launches)
    if (cc.pre_words64 == 0 && cc.on_words64 == 0) {
        static const std::vector<uint32_t> empty = [] {
            b3::Hasher hs;
            uint8_t out[32];
            hs.finalize(out);
            std::vector<uint32_t> w(8);
            for (int k = 0; k < 8; k++) w[k] = (uint32_t)out[4 * k] | ((uint32_t)out[4 * k + 1] << 8) | ((uint32_t)out[4 * k + 2] << 16) | ((uint32_t)out[4 * k + 3] << 24);
            return w;
        }();
        launch_fill_digests(ctx->stream, dig + 2 * DW, 2 * s->R, empty.data());
        n_launch += 1;
    } else {
        n_launch += launch_b3_contig(ctx->stream, s->d_pre64, cc.pre_words64, s->R, s->d_cv[0], s->d_cv[1], dig + 2 * DW);
        n_launch += launch_b3_contig(ctx->stream, s->d_on64, cc.on_words64, s->R, s->d_cv[0], s->d_cv[1], dig + 3 * DW);
    }
    ctx->count(n_launch);
    ctx->phase(-1);
    return RV_OK;
}

static int shard_run(rv_shard* s, int mode, InterpParams& p, Interp64Params& p64) {
    int rc;
    if ((rc = shard_run_alloc(s, p, p64)) || (rc = shard_run_levels(s, mode, p, p64)) || (rc = shard_run_hash(s))) return rc;
    return RV_OK;
}

static int shard_join(rv_shard* s) {
    const size_t DW = (size_t)s->R * 8;
    s->ctx->phase(RV_PH_JOIN);
    s->ctx->count();
    launch_join(s->ctx->stream, s->d_dig, s->d_dig + DW, s->d_dig + 2 * DW, s->d_dig + 3 * DW, s->R, s->d_h);
    s->ctx->phase(-1);
    HIPCHK(hipGetLastError());
    return RV_OK;
}

#ifdef RV_EXPERIMENTS
// (a lane's mask window may start a few bytes before its segment's first mask: slack in front of every repetition's masks)
constexpr size_t REP_MASK_FRONT = 16;
// The rep-sliced prover (rep.hip): a workgroup per repetition, live wires in LDS, rep-major masks and transcripts.
// Same digests as shard_setup_prg + shard_run, for the circuits build_rep_program accepts.
static int shard_commit_rep(rv_shard* s) {
    rv_ctx* ctx = s->ctx;
    const rv_circuit* c = s->c;
    const Compiled& cc = c->cc;
    hipStream_t st = ctx->stream;
    const uint32_t R = s->R;
    int rc;
    s->rep = true;
    const uint64_t n_blocks = cc.n_masks_pad / 128, n4 = (n_blocks + 3) / 4;
    auto pad = [](uint64_t n) { return (n + 1024 + 1023) & ~(uint64_t)1023; };  // (lanes past a segment's end still load: slack behind)
    s->mask_stride = pad(512 * n4 + REP_MASK_FRONT);
    s->on_stride = pad(cc.n_on);
    s->pre_stride = pad(cc.n_pre);
    const size_t cvw = b3_stream_scratch_words(std::max(cc.n_on, cc.n_pre), R);
    if ((rc = dalloc(ctx, (size_t)R * 8 * RK_BYTES, &s->d_rkbytes)) || (rc = dalloc(ctx, (size_t)RK_AREAS * 128 * R, &s->d_rk_rep)) ||
        (rc = dalloc(ctx, (size_t)R * s->mask_stride + REP_MASK_FRONT, &s->d_masks_rep)) || (rc = dalloc(ctx, (size_t)R * s->on_stride, &s->d_on_rep)) ||
        (rc = dalloc(ctx, (size_t)R * s->pre_stride, &s->d_pre_rep)) || (rc = dalloc(ctx, std::max<size_t>(c->rp.n_vb_words, 1) * 4, &s->d_vbits)) ||
        (rc = dalloc(ctx, cvw, &s->d_cv[0])) || (rc = dalloc(ctx, cvw, &s->d_cv[1])) || (rc = dalloc(ctx, (size_t)4 * R * 8, &s->d_dig)))
        return rc;
    if (!s->d_err && (rc = dalloc(ctx, 1, &s->d_err))) return rc;
    if (!s->d_h && (rc = dalloc(ctx, (size_t)R * 32, &s->d_h))) return rc;
    launch_key_schedule(st, s->d_keys, R * 8, s->d_rkbytes);
    launch_bitslice_rk_rep(st, s->d_rkbytes, R, s->d_rk_rep);
    ctx->count(2);
    ctx->phase(RV_PH_MASKS);
    launch_aes_rep_masks(st, s->d_rk_rep, R, n_blocks, s->d_masks_rep + REP_MASK_FRONT, s->mask_stride);
    ctx->count();
    ctx->phase(RV_PH_INTERP);
    HIPCHK(hipMemsetAsync(s->d_err, 0, 8 * sizeof(int), st));
    launch_rep_clear(st, c->d_rep_levels, c->rp.n_levels, c->d_rep_segs, c->d_rep_recs, s->d_wit, (uint32_t*)s->d_vbits, s->d_err, c->rp.lds_slots);
    RepParams P{};
    P.levels = c->d_rep_levels;
    P.segs = c->d_rep_segs;
    P.recs = c->d_rep_recs;
    P.vbits = (const uint32_t*)s->d_vbits;
    P.wit = s->d_wit;
    P.masks = s->d_masks_rep + REP_MASK_FRONT;
    P.on = s->d_on_rep;
    P.pre = s->d_pre_rep;
    P.mask_stride = s->mask_stride;
    P.on_stride = s->on_stride;
    P.pre_stride = s->pre_stride;
    P.n_levels = c->rp.n_levels;
    if (getenv("RV_REP_DEBUG")) {
        int dbg[8];
        (void)hipMemcpyAsync(dbg, s->d_err, sizeof dbg, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "[rep debug] err=%d n=%d level=%d seg=%d k=%d a=%08x v=%d\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], (unsigned)dbg[5], dbg[6]);
    }
    launch_rep_interp(st, P, R, c->rp.lds_slots);
    ctx->count(2);
    ctx->phase(RV_PH_HASH);
    const size_t DW = (size_t)R * 8;
    uint32_t n_launch = launch_b3_bytes(st, s->d_pre_rep, s->pre_stride, cc.n_pre, R, s->d_cv[0], s->d_cv[1], s->d_dig + 0 * DW);
    n_launch += launch_b3_bytes(st, s->d_on_rep, s->on_stride, cc.n_on, R, s->d_cv[0], s->d_cv[1], s->d_dig + 1 * DW);
    {  // the Z64 transcripts of a pure GF(2) circuit are empty: BLAKE3("")
        b3::Hasher hs;
        uint8_t e[32];
        hs.finalize(e);
        uint32_t w[8];
        for (int k = 0; k < 8; k++) w[k] = (uint32_t)e[4 * k] | ((uint32_t)e[4 * k + 1] << 8) | ((uint32_t)e[4 * k + 2] << 16) | ((uint32_t)e[4 * k + 3] << 24);
        launch_fill_digests(st, s->d_dig + 2 * DW, 2 * R, w);
    }
    ctx->count(n_launch + 1);
    ctx->phase(-1);
    HIPCHK(hipGetLastError());
    return RV_OK;
}
#endif  // RV_EXPERIMENTS

// defer_sync: do not wait for the device (nor look at the invalid-witness flag): the caller queues more work behind
// the commitment and checks s->d_err itself after its own synchronisation
static int rv_shard_commit_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                               size_t n_z64, const uint8_t* seeds, uint32_t rep_begin, uint32_t rep_count, rv_shard** out,
                               bool defer_sync = false, EarlyRun* ec = nullptr);

extern "C" int rv_shard_commit(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                               size_t n_z64, const uint8_t* seeds, uint32_t rep_begin, uint32_t rep_count, rv_shard** out) {
    try {  // no C++ exception may cross the C boundary
        return rv_shard_commit_impl(ctx, c, wit_gf2, n_gf2, wit_z64, n_z64, seeds, rep_begin, rep_count, out);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

static int rv_shard_commit_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                               size_t n_z64, const uint8_t* seeds, uint32_t rep_begin, uint32_t rep_count, rv_shard** out,
                               bool defer_sync, EarlyRun* ec) {
    if (!ctx || !c || !out || !seeds) return RV_E_ARG;
    if (rep_count == 0 || rep_count % 8 || rep_begin % 8 || rep_begin + rep_count > RV_TOTAL_REPS) return RV_E_ARG;
    *out = nullptr;
    const Compiled& cc = c->cc;
    if (n_gf2 < cc.n_in || n_z64 < cc.n_in64) return RV_E_WITNESS_SHORT;
    if ((cc.n_in && !wit_gf2) || (cc.n_in64 && !wit_z64)) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    rv_shard* s = new rv_shard();
    s->ctx = ctx;
    s->c = c;
    s->rep_begin = rep_begin;
    s->R = rep_count;
    s->NQ = rep_count / 4;
    int rc = RV_OK;
    auto fail = [&](int code) {
        rv_shard_destroy(s);
        return code;
    };
    // defer_sync callers (rv_prove and its relatives) wait for the stream before they return, so the page-locked input
    // staging buffer is free again by the next call: seeds and witness go over in ONE asynchronous copy
    const size_t seed_bytes = (size_t)s->R * 16;
    const bool stage_in = defer_sync && seed_bytes + cc.n_in <= rv_ctx::IN_STAGE_BYTES &&
                          (ctx->h_in || hipHostMalloc((void**)&ctx->h_in, rv_ctx::IN_STAGE_BYTES, hipHostMallocDefault) == hipSuccess);
    if (!stage_in) (void)hipGetLastError();
    if (stage_in) {
        if ((rc = dalloc(ctx, seed_bytes + std::max<size_t>(cc.n_in, 1), &s->d_seeds)) || (rc = dalloc(ctx, (size_t)s->R * 128, &s->d_keys))) return fail(rc);
        s->d_wit = s->d_seeds + seed_bytes;  // (inside d_seeds' block: the arena ignores it on release)
        memcpy(ctx->h_in, seeds, seed_bytes);
        if (cc.n_in) memcpy(ctx->h_in + seed_bytes, wit_gf2, cc.n_in);
        if (hipMemcpyAsync(s->d_seeds, ctx->h_in, seed_bytes + cc.n_in, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(RV_E_DEVICE);
    } else {
        if ((rc = dalloc(ctx, seed_bytes, &s->d_seeds)) || (rc = dalloc(ctx, (size_t)s->R * 128, &s->d_keys)) ||
            (rc = dalloc(ctx, std::max<size_t>(cc.n_in, 1), &s->d_wit)))
            return fail(rc);
        if (hipMemcpyAsync(s->d_seeds, seeds, seed_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(RV_E_DEVICE);
        if (cc.n_in && hipMemcpyAsync(s->d_wit, wit_gf2, cc.n_in, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(RV_E_DEVICE);
    }
    if (cc.n_in64) {
        if ((rc = dalloc(ctx, cc.n_in64, &s->d_wit64))) return fail(rc);
        if (hipMemcpyAsync(s->d_wit64, wit_z64, cc.n_in64 * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return fail(RV_E_DEVICE);
    }
    static const bool vclr_on = [] {
        const char* e = getenv("RV_VCLR");
        return !e || atoi(e) != 0;
    }();
    const bool rep_path = c->rep_ok && (rep_mode() >= 2 || (rep_mode() == 1 && rep_count == RV_TOTAL_REPS));
    const bool use_vclr = !rep_path && vclr_on && c->vclr_ok && (s->NQ == 64 || s->NQ == 32 || s->NQ == 16 || s->NQ == 8);
#ifdef RV_EXPERIMENTS
    if (use_vclr && c->flat.ok && (flat_mode() == 1 || (flat_mode() == 3 && chain_supports(s->NQ))) && mul_flat_supports(s->NQ) && !g_recorder) {
        // split schedule: the level chain computes the values itself (shard_run_split)
        if ((rc = ctx_side_streams(ctx))) return fail(rc);
        s->split = true;
        if ((rc = dalloc(ctx, (size_t)cc.n_rows, &s->d_vclr))) return fail(rc);
        if (hipMemsetAsync(s->d_vclr + cc.zero_row, 0, 1, ctx->stream) != hipSuccess) return fail(RV_E_DEVICE);
    } else if (use_vclr && c->flat.ok && flat_mode() >= 2 && mul_flat_supports(s->NQ) && !g_recorder) {
        // flat schedule: the cleartext pass starts as soon as the witness is on the device, on a stream of its own, and runs
        // beside the key schedules and the mask generator (which leaves it clear_wgs() compute units)
        if ((rc = ctx_side_streams(ctx))) return fail(rc);
        s->flat = true;
        if ((rc = dalloc(ctx, (size_t)cc.n_rows, &s->d_vclr)) || (rc = dalloc(ctx, (size_t)4, &s->d_sync))) return fail(rc);
        hipEvent_t ev_in = ctx->get_sync_event();
        s->misc_events.push_back(ev_in);
        s->ev_clear = ctx->get_sync_event();
        if (hipEventRecord(ev_in, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream3, ev_in, 0) != hipSuccess ||
            hipMemsetAsync(s->d_sync, 0, 16, ctx->stream3) != hipSuccess || hipMemsetAsync(s->d_vclr + cc.zero_row, 0, 1, ctx->stream3) != hipSuccess)
            return fail(RV_E_DEVICE);
        const bool timed = ctx->profiling && !ctx->clear_timed;
        if (timed) {
            if (!ctx->clear_a) (void)hipEventCreate(&ctx->clear_a);
            if (!ctx->clear_b) (void)hipEventCreate(&ctx->clear_b);
            (void)hipEventRecord(ctx->clear_a, ctx->stream3);
        }
        launch_clear(ctx->stream3, clear_wgs(), c->d_clear_s, c->d_clear_k, c->d_clear_levels, (uint32_t)c->flat.n_clear_levels, s->d_wit, s->d_vclr,
                     (int*)(s->d_sync + 2), s->d_sync);
        if (timed) {
            (void)hipEventRecord(ctx->clear_b, ctx->stream3);
            ctx->clear_timed = true;
        }
        if (hipEventRecord(s->ev_clear, ctx->stream3) != hipSuccess) return fail(RV_E_DEVICE);
    }
#endif  // RV_EXPERIMENTS
    s->z64f = !rep_path && c->z64f_ok && z64_fused_on() && z64_fused_supports(s->NQ) && !g_recorder;
    {
        // the mask generator beside the level launches (RV_OVERLAP=0 turns it off; RV_OVERLAP_MIN = fewest CTR blocks): wide circuits
        // only -- a level must be long enough to hide a share of the cipher behind
        const int ov_mode = getenv("RV_OVERLAP") ? atoi(getenv("RV_OVERLAP")) : 1;  // (read at every call: bench.py and the tools switch it)
        const uint64_t ov_min = getenv("RV_OVERLAP_MIN") ? strtoull(getenv("RV_OVERLAP_MIN"), nullptr, 0) : 8192;  // (per call: the tests lower it)
        s->overlap = ov_mode != 0 && !rep_path && !s->flat && !s->split && !g_recorder && aes_col4_supports(s->NQ) &&
                     cc.n_masks_pad / 128 >= ov_min;
#ifdef RV_EXPERIMENTS
        if (persist_mode() && persist_supports(s->NQ)) s->overlap = false;
#endif
    }
    ctx->phase(RV_PH_SETUP);
    ctx->count();
    launch_expand_seeds(ctx->stream, s->d_seeds, s->R, s->d_keys);
#ifdef RV_EXPERIMENTS
    if (rep_path) {
        if ((rc = shard_commit_rep(s))) return fail(rc);
    } else
#endif
    {
        if ((rc = shard_setup_prg(s, nullptr))) return fail(rc);
        if (ec) s->ec = ec;  // (the caller made sure this path is taken: no rep-sliced prover, one stream)
        InterpParams p{};
        p.wit = s->d_wit;
        Interp64Params p64{};
        p64.wit = s->d_wit64;
        int mode = MODE_PROVE;
        if (s->split) {
            mode = MODE_PROVE_V;  // (the level chain keeps the value bytes: shard_run_split)
            p.vclr = s->d_vclr;
        } else if (s->flat) {
            mode = MODE_PROVE_F;  // (no corr rows, no value bytes: shard_run_flat)
        } else if (use_vclr) {
            // eligible circuits (whole proofs and the repetition shards with a specialised interpreter): cleartext values
            // instead of corr rows (internal.h: MODE_PROVE_V)
            if ((rc = dalloc(ctx, (size_t)cc.n_rows, &s->d_vclr))) return fail(rc);
            if (hipMemsetAsync(s->d_vclr + cc.zero_row, 0, 1, ctx->stream) != hipSuccess) return fail(RV_E_DEVICE);
            p.vclr = s->d_vclr;
            mode = MODE_PROVE_V;
        }
        if ((rc = shard_run(s, mode, p, p64))) return fail(rc);
    }
    if ((rc = shard_join(s))) return fail(rc);
    if (defer_sync) {
        ctx->prof.calls++;
        *out = s;
        return RV_OK;
    }
    int err = 0;
    if (hipMemcpyAsync(&err, s->d_err, sizeof err, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return fail(RV_E_DEVICE);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
        return fail(RV_E_DEVICE);
    }
    ctx->collect();
    ctx->prof.calls++;
    if (err) return fail((err & (RV_DEV_CLEAR_ABORT | RV_DEV_PERSIST_ABORT)) ? RV_E_DEVICE : RV_E_WITNESS_INVALID);
    *out = s;
    return RV_OK;
}

extern "C" int rv_shard_digests_device(rv_shard* s, void** dptr) {
    if (!s || !dptr) return RV_E_ARG;
    *dptr = s->d_h;
    return RV_OK;
}

extern "C" int rv_shard_digests(rv_shard* s, uint8_t* out) {
    if (!s || !out) return RV_E_ARG;
    HIPCHK(hipMemcpyAsync(out, s->d_h, (size_t)s->R * 32, hipMemcpyDeviceToHost, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    return RV_OK;
}

extern "C" int rv_shard_digests_to_device(rv_shard* s, void* dst_device) {
    if (!s || !dst_device) return RV_E_ARG;
    HIPCHK(hipMemcpyAsync(dst_device, s->d_h, (size_t)s->R * 32, hipMemcpyDeviceToDevice, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    return RV_OK;
}

extern "C" int rv_hook_shard_stream_digests(rv_shard* s, uint8_t* out) {
    if (!s || !out) return RV_E_ARG;
    std::vector<uint32_t> tmp((size_t)4 * s->R * 8);
    HIPCHK(hipMemcpyAsync(tmp.data(), s->d_dig, tmp.size() * 4, hipMemcpyDeviceToHost, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    for (uint32_t r = 0; r < s->R; r++)
        for (int k = 0; k < 4; k++) memcpy(out + ((size_t)r * 4 + k) * 32, &tmp[((size_t)k * s->R + r) * 8], 32);
    return RV_OK;
}

// layout of the opened shard in HBM: gf2_online | gf2_pre | z64_online | z64_pre
struct OpenLayout {
    uint64_t l2r, l2c, l2i, l64r, l64c, l64i, sz2, sz64;
    uint32_t n_on, n_pre;
    size_t len[4], base[4], total;
};

// `framed`: leave room for the bincode framing of a whole Proof around the four sections
// (comm[32] | u64 n | gf2_online | u64 n | gf2_pre | u64 n | z64_online | u64 n | z64_pre), so a
// single-shard proof can be produced in its final layout on the device and copied out once
static OpenLayout open_layout(const Compiled& cc, const uint8_t* omit_local, uint32_t R, bool framed = false) {
    OpenLayout L{};
    // GF(2) vectors: 8 items per byte, plus the always-present extra chunk
    L.l2r = cc.n_rec / 8 + 1;
    L.l2c = cc.n_pre / 8 + 1;
    L.l2i = cc.n_in / 8 + 1;
    // Z64 vectors: 8 bytes LE per item, no padding (z64/share.rs:42-48, z64/recon.rs:59-65)
    L.l64r = 8 * cc.n_rec64;
    L.l64c = 8 * cc.n_corr64;
    L.l64i = 8 * cc.n_in64;
    L.sz2 = 1 + 128 + 24 + L.l2r + L.l2c + L.l2i;
    L.sz64 = 1 + 128 + 24 + L.l64r + L.l64c + L.l64i;
    for (uint32_t r = 0; r < R; r++) (omit_local[r] < 8 ? L.n_on : L.n_pre)++;
    L.len[0] = (size_t)L.n_on * L.sz2;
    L.len[1] = (size_t)L.n_pre * 48;
    L.len[2] = (size_t)L.n_on * L.sz64;
    L.len[3] = (size_t)L.n_pre * 48;
    size_t off = framed ? 32 : 0;
    for (int k = 0; k < 4; k++) {
        if (framed) off += 8;
        L.base[k] = off;
        off += L.len[k];
    }
    L.total = off;
    return L;
}

extern "C" int rv_circuit_record_sizes(const rv_circuit* c, size_t* gf2_online_record, size_t* z64_online_record) {
    if (!c || !gf2_online_record || !z64_online_record) return RV_E_ARG;
    const uint8_t none[8] = {8, 8, 8, 8, 8, 8, 8, 8};
    const OpenLayout L = open_layout(c->cc, none, 8);
    *gf2_online_record = (size_t)L.sz2;
    *z64_online_record = (size_t)L.sz64;
    return RV_OK;
}

extern "C" int rv_shard_open_size(const rv_shard* s, const uint8_t omit[RV_TOTAL_REPS], size_t lens[4]) {
    if (!s || !omit || !lens) return RV_E_ARG;
    const OpenLayout L = open_layout(s->c->cc, omit + s->rep_begin, s->R);
    for (int k = 0; k < 4; k++) lens[k] = L.len[k];
    return RV_OK;
}

// fs_mailbox (host-mapped, nullable; device Fiat-Shamir only): the challenge is also published there -- sequence number fs_seq
// in word 0 once comm[32], the opening map [256] and {n_on, n_pre} stand from word 16 on --, and the corrections vectors are
// NOT extracted (early corrections: the host has them already, rv_prove_impl)
// rv_prove's early path, second half (round 4): the opened repetitions' broadcast-bit vectors -- the other 25 MB of a 50 MB proof --
// do not go into the proof image and out through one kernel copy behind the extraction (0.24 + 0.47 ms on the 10^7-gate circuit);
// they are extracted in slices into a dense staging block [40][pitch], every finished slice leaves through the copy engine (a 2-D
// transfer with 16-byte pitches: the fast path) while the next one is extracted, and the helper threads scatter the slices into
// the proof as they arrive.  cut[k] .. cut[k + 1] = byte range of slice k; ev[k] = its copy has arrived.
struct RecStage {
    uint8_t* d = nullptr;
    uint8_t* h = nullptr;
    uint64_t pitch = 0;
    std::vector<uint64_t> cut;
    std::vector<hipEvent_t> ev;
    // a slice's extraction is announced by a stamp in the mailbox (word 2: seq << 8 | slices extracted), and the HOST then hands
    // its copy to the second stream: a wait queued there ahead of time would sit at the head of that queue through the whole proof
    // (polled by the command processor between the main stream's launches) and hold the corrections' copies back behind it
    uint32_t* box_dev = nullptr;
    uint32_t seq = 0;
};

static int shard_open_impl(rv_shard* s, const uint8_t* omit /* NULL: device Fiat-Shamir */, void* dst, void** dptr, size_t lens[4],
                           bool framed = false, uint8_t* comm_out = nullptr, uint8_t* omit_out = nullptr, bool no_sync = false,
                           const uint8_t* d_all_h = nullptr /* device: all 256 digests (sharded proofs after the all-gather) */,
                           uint32_t* fs_mailbox = nullptr, uint32_t fs_seq = 0, uint32_t corr2_rep_min = 0, uint32_t corr64_rep_min = 0 /* early
                           corrections: the GF(2) / Z64 corrections vectors of the repetitions below these are not extracted */,
                           RecStage* rec_stage = nullptr /* the GF(2) broadcast-bit vectors leave in slices through this staging */);

extern "C" int rv_shard_open_device(rv_shard* s, const uint8_t omit[RV_TOTAL_REPS], void** dptr, size_t lens[4]) {
    if (!omit) return RV_E_ARG;
    return shard_open_impl(s, omit, nullptr, dptr, lens);
}

extern "C" int rv_shard_open_into(rv_shard* s, const uint8_t omit[RV_TOTAL_REPS], void* dst_device, size_t lens[4]) {
    if (!dst_device || !omit) return RV_E_ARG;
    void* d = nullptr;
    return shard_open_impl(s, omit, dst_device, &d, lens);
}

// `omit` == NULL: Fiat-Shamir on the device (k_fs_challenge) from the shard's own digests; only for a shard that
// holds all 256 repetitions.  comm_out / omit_out (nullable) then receive comm and the opening map.
static int shard_open_impl(rv_shard* s, const uint8_t* omit, void* dst, void** dptr, size_t lens[4], bool framed, uint8_t* comm_out,
                           uint8_t* omit_out, bool no_sync, const uint8_t* d_all_h, uint32_t* fs_mailbox, uint32_t fs_seq, uint32_t corr2_rep_min,
                           uint32_t corr64_rep_min, RecStage* rec_stage) {
    if (!s || !dptr || !lens) return RV_E_ARG;
    if (fs_mailbox && (omit || s->rep)) return RV_E_ARG;
    rv_ctx* ctx = s->ctx;
    const Compiled& cc = s->c->cc;
    HIPCHK(hipSetDevice(ctx->device));
    const bool self = omit == nullptr;
    const bool whole = s->rep_begin == 0 && s->R == RV_TOTAL_REPS;
    // device Fiat-Shamir needs all 256 digests: the shard's own when it holds every repetition, else the gathered ones;
    // a partial shard's output size depends on the challenge, so the caller provides the buffer (worst case, see header)
    if (self && !whole && (!d_all_h || !dst || framed || no_sync)) return RV_E_ARG;
    uint8_t canon[RV_TOTAL_REPS];  // any map with the 40 / 216 split gives the layout: sizes do not depend on WHICH reps open
    if (self) {
        for (uint32_t r = 0; r < RV_TOTAL_REPS; r++) canon[r] = r < RV_ONLINE_REPS ? 0 : RV_PLAYERS;
    }
    const uint8_t* om = self ? canon : omit + s->rep_begin;
    for (uint32_t r = 0; r < s->R; r++)
        if (om[r] > 8) return RV_E_ARG;
    const OpenLayout L = open_layout(cc, om, s->R, framed);
    constexpr size_t OL_WORDS = (sizeof(OnlineList) + 7) / 8;
    // off2, off64, gf2 rec/corr/in dst, z64 rec/corr/in dst; then the OnlineList
    std::vector<uint64_t> offs((size_t)8 * s->R + OL_WORDS);
    if (!self) {
        uint32_t k_on = 0, k_pre = 0;
        OnlineList ol{};
        for (uint32_t r = 0; r < s->R; r++) {
            if (om[r] < 8) {
                offs[r] = L.base[0] + (uint64_t)k_on * L.sz2;
                offs[s->R + r] = L.base[2] + (uint64_t)k_on * L.sz64;
                offs[2 * s->R + r] = offs[r] + 137;
                offs[3 * s->R + r] = offs[r] + 145 + L.l2r;
                offs[4 * s->R + r] = offs[r] + 153 + L.l2r + L.l2c;
                offs[5 * s->R + r] = offs[s->R + r] + 137;
                offs[6 * s->R + r] = offs[s->R + r] + 145 + L.l64r;
                offs[7 * s->R + r] = offs[s->R + r] + 153 + L.l64r + L.l64c;
                if (ol.n < RV_ONLINE_REPS) {
                    ol.rep[ol.n] = r;
                    ol.dst[ol.n] = offs[3 * s->R + r];
                    ol.n++;
                }
                k_on++;
            } else {
                offs[r] = L.base[1] + (uint64_t)k_pre * 48;
                offs[s->R + r] = L.base[3] + (uint64_t)k_pre * 48;
                k_pre++;
            }
        }
        memcpy(&offs[(size_t)8 * s->R], &ol, sizeof ol);
    }
    int rc;
    ctx->release(s->d_omit);
    ctx->release(s->d_offs);
    ctx->release(s->d_out);
    s->d_omit = nullptr;
    s->d_offs = nullptr;
    s->d_out = nullptr;
    // d_omit: [R] omit of the shard, then (device Fiat-Shamir) comm[32], the whole opening map [256], {n_on, n_pre}
    constexpr size_t FS_TAIL = 32 + RV_TOTAL_REPS + 8;
    if ((rc = dalloc(ctx, s->R + FS_TAIL, &s->d_omit)) || (rc = dalloc(ctx, offs.size(), &s->d_offs))) return rc;
    uint8_t* d_out = (uint8_t*)dst;
    if (!d_out) {
        if ((rc = dalloc(ctx, std::max<size_t>(L.total, 1), &s->d_out))) return rc;
        d_out = s->d_out;
    }
    const OnlineList* d_ol = (const OnlineList*)(s->d_offs + (size_t)8 * s->R);
    const size_t DW = (size_t)s->R * 8;
    ctx->phase(RV_PH_OPEN);
    if (self) {
        FsLayout F{};
        for (int k = 0; k < 4; k++) F.base[k] = L.base[k];
        F.sz2 = L.sz2, F.sz64 = L.sz64, F.l2r = L.l2r, F.l2c = L.l2c, F.l64r = L.l64r, F.l64c = L.l64c;
        F.framed = whole ? 1u : 0u;  // a whole shard opens 40 / 216: L is exact; a partial one gets its section starts on the device
        F.comm2 = framed ? d_out : nullptr;  // a framed proof starts with comm
        launch_fs_challenge(ctx->stream, d_all_h ? d_all_h : s->d_h, F, s->rep_begin, s->R, s->d_omit + s->R, s->d_omit,
                            s->d_omit + s->R + 32, s->d_offs, (OnlineList*)d_ol, (uint32_t*)(s->d_omit + s->R + 32 + RV_TOTAL_REPS));
        ctx->count();
        if (fs_mailbox) {
            launch_publish(ctx->stream, (const uint32_t*)(s->d_omit + s->R), (uint32_t)(FS_TAIL / 4), fs_mailbox + 16, fs_mailbox, fs_seq);
            ctx->count();
        }
    } else {
        HIPCHK(hipMemcpyAsync(s->d_omit, om, s->R, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(s->d_offs, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    const bool any_on = self || L.n_on;  // (device mode: how many of the shard's repetitions open is not known on the host yet)
    ctx->count(any_on ? 4 : 1);
    launch_open_headers(ctx->stream, s->R, s->d_omit, s->d_seeds, s->d_keys, s->d_dig + 1 * DW, s->d_dig + 3 * DW, s->d_offs,
                        s->d_offs + s->R, L.l2r, L.l2c, L.l2i, L.l64r, L.l64c, L.l64i, d_out);
#ifdef RV_EXPERIMENTS
    if (any_on && s->rep) {
        // rep-major transcripts: only the opened repetitions' bytes are read at all
        launch_rep_open(ctx->stream, s->d_on_rep, s->on_stride, s->c->d_rec_rows, cc.n_rec, 0, d_ol, s->d_omit, s->d_offs + 2 * s->R, d_out);
        launch_rep_open(ctx->stream, s->d_pre_rep, s->pre_stride, nullptr, cc.n_pre, 1, d_ol, s->d_omit, s->d_offs + 3 * s->R, d_out);
        launch_rep_open(ctx->stream, s->d_on_rep, s->on_stride, s->c->d_in_rows, cc.n_in, 1, d_ol, s->d_omit, s->d_offs + 4 * s->R, d_out);
    } else
#endif
    if (any_on) {
#ifdef RV_EXPERIMENTS
        if (rec_stage) {
            // slice by slice: extract into the staging block, then (second stream, behind an event) the slice's 2-D copy to the host
            // (host order: the event, then the other stream's wait for it -- a wait resolves to the stream's tail at queueing time)
            for (size_t k = 0; k + 1 < rec_stage->cut.size(); k++) {
                const uint64_t b0 = rec_stage->cut[k], b1 = rec_stage->cut[k + 1];
                launch_extract_bits_stage(ctx->stream, s->d_on, s->c->d_rec_rows, cc.n_rec, s->NQ, s->d_omit, rec_stage->d, rec_stage->pitch, b0, b1 - b0);
                launch_publish(ctx->stream, nullptr, 0, nullptr, rec_stage->box_dev + 2, (rec_stage->seq << 8) | (uint32_t)(k + 1));
                ctx->count(2);
            }
        } else
#endif
        {
            launch_extract_bits(ctx->stream, s->d_on, s->c->d_rec_rows, cc.n_rec, s->NQ, 0, s->d_omit, s->d_offs + 2 * s->R, d_out);
        }
        if (corr2_rep_min < s->R) launch_extract_from_bits(ctx->stream, s->d_pre, cc.n_pre, s->NQ, d_ol, d_out, corr2_rep_min);
        launch_extract_bits(ctx->stream, s->d_on, s->c->d_in_rows, cc.n_in, s->NQ, 1, s->d_omit, s->d_offs + 4 * s->R, d_out);
        launch_extract64(ctx->stream, s->d_on64, cc.on_words64, s->c->d_rec_offs64, cc.n_rec64, 1, s->R, s->d_omit,
                         s->d_offs + 5 * s->R, d_out, d_ol);
        launch_extract64(ctx->stream, s->d_pre64, cc.pre_words64, nullptr, cc.n_corr64, 0, s->R, s->d_omit, s->d_offs + 6 * s->R,
                         d_out, d_ol, corr64_rep_min);
        launch_extract64(ctx->stream, s->d_on64, cc.on_words64, s->c->d_in_offs64, cc.n_in64, 0, s->R, s->d_omit,
                         s->d_offs + 7 * s->R, d_out, d_ol);
    }
    HIPCHK(hipGetLastError());
    ctx->phase(-1);
    if (self && no_sync) {
        // rv_prove_batch: nothing on the host depends on this proof yet; the caller synchronises once per batch
        *dptr = d_out;
        for (int k = 0; k < 4; k++) lens[k] = L.len[k];
        return RV_OK;
    }
    if (self) {
        uint8_t back[FS_TAIL];
        HIPCHK(hipMemcpyAsync(back, s->d_omit + s->R, sizeof back, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (comm_out) memcpy(comm_out, back, 32);
        if (omit_out) memcpy(omit_out, back + 32, RV_TOTAL_REPS);
        uint32_t cnt[2];
        memcpy(cnt, back + 32 + RV_TOTAL_REPS, sizeof cnt);
        ctx->collect();
        *dptr = d_out;
        lens[0] = (size_t)cnt[0] * L.sz2;
        lens[1] = (size_t)cnt[1] * 48;
        lens[2] = (size_t)cnt[0] * L.sz64;
        lens[3] = (size_t)cnt[1] * 48;
        return RV_OK;
    } else {
        // the host vector `offs` must outlive the async copy
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->collect();
    *dptr = d_out;
    for (int k = 0; k < 4; k++) lens[k] = L.len[k];
    return RV_OK;
}

extern "C" int rv_shard_open_self(rv_shard* s, void* dst_device, uint8_t comm[RV_HASH_SIZE], uint8_t omit[RV_TOTAL_REPS],
                                  size_t lens[4]) {
    if (!dst_device || !comm || !omit) return RV_E_ARG;
    void* d = nullptr;
    return shard_open_impl(s, nullptr, dst_device, &d, lens, false, comm, omit);
}

extern "C" int rv_shard_open_gathered(rv_shard* s, const void* all_digests_device, void* dst_device, uint8_t comm[RV_HASH_SIZE],
                                      uint8_t omit[RV_TOTAL_REPS], size_t lens[4]) {
    if (!all_digests_device || !dst_device || !comm || !omit) return RV_E_ARG;
    void* d = nullptr;
    return shard_open_impl(s, nullptr, dst_device, &d, lens, false, comm, omit, false, (const uint8_t*)all_digests_device);
}

extern "C" int rv_shard_open(rv_shard* s, const uint8_t omit[RV_TOTAL_REPS], rv_shard_parts* parts) {
    if (!parts) return RV_E_ARG;
    memset(parts, 0, sizeof *parts);
    void* d = nullptr;
    size_t lens[4];
    int rc = rv_shard_open_device(s, omit, &d, lens);
    if (rc) return rc;
    uint8_t** dst[4] = {&parts->gf2_online, &parts->gf2_pre, &parts->z64_online, &parts->z64_pre};
    size_t* dl[4] = {&parts->gf2_online_len, &parts->gf2_pre_len, &parts->z64_online_len, &parts->z64_pre_len};
    size_t off = 0;
    for (int k = 0; k < 4; k++) {
        *dst[k] = (uint8_t*)malloc(lens[k] ? lens[k] : 1);
        if (!*dst[k]) return RV_E_NOMEM;
        if (lens[k]) HIPCHK(hipMemcpyAsync(*dst[k], (uint8_t*)d + off, lens[k], hipMemcpyDeviceToHost, s->ctx->stream));
        *dl[k] = lens[k];
        off += lens[k];
    }
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    const uint8_t* om = omit + s->rep_begin;
    for (uint32_t r = 0; r < s->R; r++) (om[r] < 8 ? parts->n_online : parts->n_pre)++;
    return RV_OK;
}

static void put_le64(uint8_t* p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

extern "C" int rv_assemble_proof(const uint8_t comm[RV_HASH_SIZE], const rv_shard_parts* parts, size_t n_parts, uint8_t** proof,
                                 size_t* proof_len) {
    if (!comm || !parts || !proof || !proof_len) return RV_E_ARG;
    size_t total = 32 + 4 * 8;
    uint64_t n_on = 0, n_pre = 0;
    for (size_t i = 0; i < n_parts; i++) {
        total += parts[i].gf2_online_len + parts[i].gf2_pre_len + parts[i].z64_online_len + parts[i].z64_pre_len;
        n_on += parts[i].n_online;
        n_pre += parts[i].n_pre;
    }
    uint8_t* out = (uint8_t*)malloc(total);
    if (!out) return RV_E_NOMEM;
    uint8_t* w = out;
    memcpy(w, comm, 32);
    w += 32;
    // Proof { comm, gf2: ProofSingle, z64: ProofSingle }, ProofSingle { online: Vec, preprocessing: Vec }
    for (int dom = 0; dom < 2; dom++) {
        put_le64(w, n_on);
        w += 8;
        for (size_t i = 0; i < n_parts; i++) {
            const uint8_t* p = dom == 0 ? parts[i].gf2_online : parts[i].z64_online;
            const size_t l = dom == 0 ? parts[i].gf2_online_len : parts[i].z64_online_len;
            if (l) memcpy(w, p, l);
            w += l;
        }
        put_le64(w, n_pre);
        w += 8;
        for (size_t i = 0; i < n_parts; i++) {
            const uint8_t* p = dom == 0 ? parts[i].gf2_pre : parts[i].z64_pre;
            const size_t l = dom == 0 ? parts[i].gf2_pre_len : parts[i].z64_pre_len;
            if (l) memcpy(w, p, l);
            w += l;
        }
    }
    *proof = out;
    *proof_len = total;
    return RV_OK;
}

// dst (nullable): page-locked destination of at least dst_cap bytes supplied by the caller (rv_prove_batch hands every
// proof a slice of one buffer); otherwise the proof gets a buffer of its own
// allow_early: the early-corrections path may be taken (a one-shot rv_prove_ops does without: its staging buffer is 3x the proof
// of page-locked memory, tens of milliseconds to map for a gain of half a millisecond)
static int rv_prove_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                        size_t n_z64, const uint8_t* seeds, uint8_t** proof, size_t* proof_len, uint8_t* dst = nullptr,
                        size_t dst_cap = 0, bool allow_early = true);

extern "C" int rv_prove(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                        size_t n_z64, const uint8_t* seeds, uint8_t** proof, size_t* proof_len) {
    try {  // no C++ exception may cross the C boundary
        return rv_prove_impl(ctx, c, wit_gf2, n_gf2, wit_z64, n_z64, seeds, proof, proof_len);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

static int rv_prove_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                        size_t n_z64, const uint8_t* seeds, uint8_t** proof, size_t* proof_len, uint8_t* dst, size_t dst_cap, bool allow_early) {
    if (!ctx || !c || !proof || !proof_len) return RV_E_ARG;
    *proof = nullptr;
    *proof_len = 0;
    uint8_t os_seeds[RV_TOTAL_REPS * RV_KEY_SIZE];
    if (!seeds) {  // proof/mod.rs:131-134 uses OsRng
        size_t got = 0;
        while (got < sizeof os_seeds) {
            ssize_t n = getrandom(os_seeds + got, sizeof os_seeds - got, 0);
            if (n <= 0) return RV_E_DEVICE;
            got += (size_t)n;
        }
        seeds = os_seeds;
    }
    rv_shard* s = nullptr;
    uint8_t* out = nullptr;
    // Early corrections (kernels.hip): half of a large GF(2) proof -- the corrections vectors -- does not depend on the
    // challenge beyond the choice of repetitions.  Every repetition's vector goes to a page-locked staging buffer through
    // the copy engine while the interpreter and the hash kernels run; once the challenge is known (published into a mapped
    // mailbox the host polls, no stream synchronisation) helper threads copy the 40 opened ones into the proof while the GPU
    // extracts the other half, which a kernel then writes around them into the same buffer.  RV_EARLY=0 turns it off.
    EarlyRun er;
    bool early = false;
    const bool early_stats = getenv("RV_EARLY_STATS") && atoi(getenv("RV_EARLY_STATS"));
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    double t_queued = 0, t_chal = 0, t_copied = 0, t_sync = 0;
    std::vector<double> t_chunk;
    uint8_t* out_dev = nullptr;
    uint32_t* fs_dev = nullptr;
    OpenLayout EL{};
    const bool early_on = !(getenv("RV_EARLY") && atoi(getenv("RV_EARLY")) == 0);
    if (early_on && allow_early && !dst && !g_recorder && rep_mode() == 0) {
        const EarlyPlan* pl = early_plan(c);
        if (pl->ok) {
            HIPCHK(hipSetDevice(ctx->device));
            uint8_t canon[RV_TOTAL_REPS];
            for (uint32_t r = 0; r < RV_TOTAL_REPS; r++) canon[r] = r < RV_ONLINE_REPS ? 0 : RV_PLAYERS;
            EL = open_layout(c->cc, canon, RV_TOTAL_REPS, true);
            bool ok = true;
            if (ctx->h_ec_cap < pl->bytes) {
                if (ctx->h_ec) (void)hipHostFree(ctx->h_ec);
                ctx->h_ec = nullptr;
                ctx->h_ec_cap = 0;
                if (hipHostMalloc((void**)&ctx->h_ec, pl->bytes, hipHostMallocDefault) == hipSuccess)
                    ctx->h_ec_cap = pl->bytes;
                else
                    ok = false;
            }
            if (ok && !pl->z64 && ctx->d_ec_cap < pl->bytes) {
                if (ctx->d_ec) (void)hipFree(ctx->d_ec);
                ctx->d_ec = nullptr;
                ctx->d_ec_cap = 0;
                if (hipMalloc((void**)&ctx->d_ec, pl->bytes) == hipSuccess)
                    ctx->d_ec_cap = pl->bytes;
                else
                    ok = false;
            }
            if (ok && !ctx->h_fs) {
                if (hipHostMalloc((void**)&ctx->h_fs, 4096, hipHostMallocMapped) == hipSuccess)
                    memset(ctx->h_fs, 0, 4096);
                else
                    ok = false;
            }
            if (ok && hipHostGetDevicePointer((void**)&fs_dev, ctx->h_fs, 0) != hipSuccess) ok = false;
            if (!ok) (void)hipGetLastError();
            if (ok && !ctx->ec_pool) {
                static const int n_helpers = getenv("RV_EARLY_THREADS") ? std::max(2, atoi(getenv("RV_EARLY_THREADS")) + 1) : 9;
                ctx->ec_pool = new HelperPool(n_helpers);
            }
            // (test knob: the staging buffer starts every proof as garbage, so that bytes copied out of it before they arrived show)
            if (ok && getenv("RV_EARLY_POISON") && atoi(getenv("RV_EARLY_POISON"))) memset(ctx->h_ec, 0x5A, pl->bytes);
            if (ok) {
                early = true;
                er.plan = pl;
                er.h_ec = ctx->h_ec;
                er.d_ec = ctx->d_ec;
                er.box_dev = fs_dev;
                er.box = ctx->h_fs;
                ctx->fs_seq = (ctx->fs_seq + 1) & 0xFFFFFFu;  // a fresh number even after a proof that failed half-way: its stamps must never match
                if (!ctx->fs_seq) ctx->fs_seq = 1;
                er.seq = ctx->fs_seq;
            }
        }
    }
    // ... and the broadcast-bit vectors in slices through the copy engine (RecStage; RV_EARLY_REC=1, off by default: through the
    // proof image and one kernel copy, as in round 3; RV_EARLY_REC_SLICES, default 4).  Measured on the 10^7-gate circuit: 5.74 -
    // 5.9 ms against 5.63 -- the link is still busy with the corrections' last chunk when the challenge arrives, the kernel copy
    // of the image already runs at link rate, and the host scatters 25 MB more
    RecStage rs;
    bool rec_staged = false;
    const uint64_t rec_min = getenv("RV_EARLY_REC_MIN") ? (uint64_t)atoll(getenv("RV_EARLY_REC_MIN")) : (1ull << 20);  // (read per call: the tests lower it)
#ifdef RV_EXPERIMENTS
    if (early && !er.plan->z64 && getenv("RV_EARLY_REC") && atoi(getenv("RV_EARLY_REC")) != 0 && c->cc.n_rec >= rec_min) {
        const uint64_t n_bytes = c->cc.n_rec / 8 + 1, g = extract_stage_granule(c->cc.n_rec);
        rs.pitch = (n_bytes + 255) & ~255ull;
        const size_t need = (size_t)RV_ONLINE_REPS * rs.pitch;
        bool ok = g % 16 == 0;
        if (ok && ctx->rs_cap < need) {
            if (ctx->h_rs) (void)hipHostFree(ctx->h_rs);
            if (ctx->d_rs) (void)hipFree(ctx->d_rs);
            ctx->h_rs = ctx->d_rs = nullptr;
            ctx->rs_cap = 0;
            if (hipHostMalloc((void**)&ctx->h_rs, need, hipHostMallocDefault) == hipSuccess && hipMalloc((void**)&ctx->d_rs, need) == hipSuccess)
                ctx->rs_cap = need;
            else
                ok = false, (void)hipGetLastError();
        }
        if (ok) {
            const int S = std::min(std::max(getenv("RV_EARLY_REC_SLICES") ? atoi(getenv("RV_EARLY_REC_SLICES")) : 4, 1), 32);
            // the first slice half the size of the others: the link starts sooner
            const uint64_t units = 2 * (uint64_t)S - 1;
            rs.cut.push_back(0);
            for (int k = 0; k < S; k++) {
                const uint64_t at = k + 1 == S ? rs.pitch : std::min<uint64_t>(((n_bytes * (2 * (uint64_t)k + 1) / units + g - 1) / g) * g, rs.pitch);
                if (at > rs.cut.back()) rs.cut.push_back(at);
            }
            for (size_t k = 0; k + 1 < rs.cut.size(); k++) rs.ev.push_back(ctx->get_sync_event());
            rs.d = ctx->d_rs, rs.h = ctx->h_rs;
            rs.box_dev = fs_dev, rs.seq = er.seq;
            rec_staged = true;
        }
    }
#else
    (void)rec_min;
#endif
    int rc = rv_shard_commit_impl(ctx, c, wit_gf2, n_gf2, wit_z64, n_z64, seeds, 0, RV_TOTAL_REPS, &s, /*defer_sync=*/true, early ? &er : nullptr);
    if (rc) {
        for (hipEvent_t e : rs.ev) ctx->sync_pool.push_back(e);
        return rc;
    }
    if (early) {
        // the proof's buffer, now that the GPU is busy (a fresh page-locked buffer of a 640 MB proof takes 45 ms to map).  Without it
        // the plain path below still works: the stamps in the stream are harmless, nothing was handed to the second stream yet
        if (!(out = (uint8_t*)out_alloc(EL.total)) || hipHostGetDevicePointer((void**)&out_dev, out, 0) != hipSuccess || ((uintptr_t)out_dev & 15)) {
            (void)hipGetLastError();
            rv_free(out);
            out = nullptr;
            early = false;
            s->ec = nullptr;
        }
    }
    if (early) do {
        void* d = nullptr;
        size_t lens[4];
        const uint32_t seq = er.seq;
        const bool z64 = er.plan->z64;
        if ((rc = shard_open_impl(s, nullptr, nullptr, &d, lens, true, nullptr, nullptr, /*no_sync=*/true, nullptr, fs_dev, seq, z64 ? 0 : er.plan->r_spec,
                                  z64 ? er.plan->r_spec : 0, rec_staged ? &rs : nullptr)))
            break;
        const size_t total = 32 + 4 * 8 + lens[0] + lens[1] + lens[2] + lens[3];
        if (total != EL.total || er.packed.size() != er.plan->chunks.size() || er.plan->chunks.size() > 255) {
            rc = RV_E_DEVICE;
            break;
        }
        // the image without the corrections vectors, then the error word (mailbox word 8)
        // (Z64: only the records of the opened repetitions below r_spec -- the kernel counts them -- go without their vectors)
        const uint64_t corr_at = 145 + (z64 ? EL.l64r : EL.l2r);
        const uint64_t rec_first = z64 ? EL.base[2] : EL.base[0], rec_size = z64 ? EL.sz64 : EL.sz2, corr_len = z64 ? EL.l64c : EL.l2c;
        if (rec_staged)
            launch_copy_gaps2(ctx->stream, (const uint8_t*)d, out_dev, total, rec_first, rec_size, 137, EL.l2r, corr_at, corr_len, RV_ONLINE_REPS, s->d_omit, er.plan->r_spec);
        else
            launch_copy_gaps(ctx->stream, (const uint8_t*)d, out_dev, total, rec_first, rec_size, corr_at, corr_len, RV_ONLINE_REPS, s->d_omit, er.plan->r_spec);
        launch_store_word(ctx->stream, s->d_err, (int*)(fs_dev + 8));
        if (hipGetLastError() != hipSuccess) {
            rc = RV_E_DEVICE;
            break;
        }
        t_queued = since();
        if ((rc = early_pump(s))) break;
        // the challenge: poll the mailbox (now and then make sure the stream is still alive)
        volatile uint32_t* box = ctx->h_fs;
        if ((rc = mailbox_wait(ctx, [&] { return __atomic_load_n(&box[0], __ATOMIC_ACQUIRE) == seq; }, &ctx->ec_wait_us[16], "early corrections (challenge)"))) break;
        t_chal = since();
        const uint8_t* omit_all = (const uint8_t*)(ctx->h_fs + 16) + 32;
        uint32_t opened[RV_ONLINE_REPS], n_open = 0;
        for (uint32_t r = 0; r < RV_TOTAL_REPS; r++)
            if (omit_all[r] < RV_PLAYERS && n_open < RV_ONLINE_REPS) opened[n_open++] = r;
        // the copies of the chunks: normally long done; the host waits for the second stream (its copies complete in order), then every
        // thread (this one included) claims (chunk, opened repetition) pieces from one counter: a helper that wakes up late takes fewer
        // pieces instead of holding its share back
        if (hipStreamSynchronize(ctx->stream2) != hipSuccess) {
            rc = hip_fail(hipGetLastError(), "early corrections (copies)", __FILE__, __LINE__);
            break;
        }
        if (early_stats) t_chunk.push_back(since());
        {
            const auto& chunks = er.plan->chunks;
            std::atomic<size_t> next_piece{0};
            std::atomic<int> bad{0};
            // (Z64: the staging buffer holds the first r_spec repetitions; the opened ones among them are the first n_staged ranks)
            uint32_t n_staged = 0;
            for (uint32_t j = 0; j < n_open; j++)
                if (opened[j] < er.plan->r_spec) n_staged = j + 1;
            const size_t n_pieces = chunks.size() * n_staged;
            const std::function<void(int)> job = [&](int t) {
                if (t == 0 && rec_staged) {
                    // the caller's thread first hands the slices of the broadcast-bit vectors to the copy engine, each as soon as its
                    // extraction is announced (the helpers scatter the corrections meanwhile)
                    for (size_t k = 0; k + 1 < rs.cut.size(); k++) {
                        auto extracted = [&] {  // (stamps overwrite each other: "at least k + 1 slices of THIS proof")
                            const uint32_t v = __atomic_load_n(&box[2], __ATOMIC_ACQUIRE);
                            return (v >> 8) == (rs.seq & 0xFFFFFFu) && (v & 0xFFu) > k;
                        };
                        if (mailbox_wait(ctx, extracted, nullptr, "early corrections (slice)")) {
                            bad.store(1);
                            return;
                        }
                        const uint64_t b0 = rs.cut[k], b1 = rs.cut[k + 1];
                        if (hipMemcpy2DAsync(rs.h + b0, rs.pitch, rs.d + b0, rs.pitch, b1 - b0, RV_ONLINE_REPS, hipMemcpyDeviceToHost, ctx->stream2) != hipSuccess ||
                            hipEventRecord(rs.ev[k], ctx->stream2) != hipSuccess) {
                            bad.store(1);
                            return;
                        }
                    }
                }
                for (;;) {
                    const size_t t = next_piece.fetch_add(1, std::memory_order_relaxed);
                    if (t >= n_pieces) return;
                    const size_t k = t / n_staged, j = t % n_staged;
                    const auto& ch = chunks[k];
                    memcpy(out + rec_first + j * rec_size + corr_at + ch.byte0, er.h_ec + ch.off + (size_t)opened[j] * ch.pitch + (z64 ? ch.byte0 : 0), ch.nbytes);
                }
            };
            ctx->ec_pool->run(job);
            t_copied = since();
            if (bad.load()) {
                rc = hip_fail(hipGetLastError(), "early corrections (copy)", __FILE__, __LINE__);
                break;
            }
        }
        if (rec_staged) {
            // the broadcast-bit vectors, slice by slice as their copies arrive: record j's bytes [cut[k], cut[k + 1]) from staging row j
            const uint64_t l2r = EL.l2r;
            for (size_t k = 0; k + 1 < rs.cut.size() && !rc; k++) {
                if (hipEventSynchronize(rs.ev[k]) != hipSuccess) {
                    rc = hip_fail(hipGetLastError(), "early corrections (broadcast bits)", __FILE__, __LINE__);
                    break;
                }
                const uint64_t b0 = rs.cut[k], b1 = std::min(rs.cut[k + 1], l2r);
                if (b1 <= b0) continue;
                std::atomic<uint32_t> next_rec{0};
                const std::function<void(int)> job = [&](int) {
                    for (;;) {
                        const uint32_t j = next_rec.fetch_add(1, std::memory_order_relaxed);
                        if (j >= n_open) return;
                        memcpy(out + rec_first + (size_t)j * rec_size + 137 + b0, rs.h + (size_t)j * rs.pitch + b0, b1 - b0);
                    }
                };
                ctx->ec_pool->run(job);
            }
            if (rc) break;
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = hip_fail(hipGetLastError(), "proof (early corrections)", __FILE__, __LINE__);
            break;
        }
        t_sync = since();
        if (early_stats) {
            fprintf(stderr, "early: queued %.3f  challenge %.3f  chunks", t_queued, t_chal);
            for (double t : t_chunk) fprintf(stderr, " %.3f", t);
            fprintf(stderr, "  copied %.3f  stream done %.3f ms\n", t_copied, t_sync);
        }
        ctx->collect();
        if (n_open != RV_ONLINE_REPS) {
            rc = RV_E_DEVICE;
            break;
        }
        if ((int)ctx->h_fs[8]) {
            rc = ((int)ctx->h_fs[8] & (RV_DEV_CLEAR_ABORT | RV_DEV_PERSIST_ABORT)) ? RV_E_DEVICE : RV_E_WITNESS_INVALID;
            break;
        }
        size_t off = 32;
        const uint64_t counts[4] = {RV_ONLINE_REPS, RV_PREPROCESSING_REPS, RV_ONLINE_REPS, RV_PREPROCESSING_REPS};
        for (int k = 0; k < 4; k++) {
            put_le64(out + off, counts[k]);
            off += 8 + lens[k];
        }
        *proof = out;
        *proof_len = total;
        out = nullptr;
        g_early_proofs.fetch_add(1, std::memory_order_relaxed);
    } while (0);
    else do {
        // commitment, challenge and openings all on the device (k_fs_challenge); the whole proof is laid out there
        // in its final bincode form (comm included) and leaves in ONE copy; the host waits for the device once
        void* d = nullptr;
        size_t lens[4];
        // small proofs: straight into the context's mapped staging buffer (the layout of a whole proof does not depend on
        // the challenge, so its size is known before the openings exist); RV_SMALL_STAGE=0: the copy-engine path
        static const bool small_stage = !(getenv("RV_SMALL_STAGE") && atoi(getenv("RV_SMALL_STAGE")) == 0);
        uint8_t* stage_dev = nullptr;
        size_t need = 0;
        if (small_stage && !g_recorder) {
            uint8_t canon[RV_TOTAL_REPS];
            for (uint32_t r = 0; r < RV_TOTAL_REPS; r++) canon[r] = r < RV_ONLINE_REPS ? 0 : RV_PLAYERS;
            need = open_layout(c->cc, canon, RV_TOTAL_REPS, true).total;
            if (need + 64 <= rv_ctx::STAGE_BYTES) {
                if (!ctx->h_stage && hipHostMalloc((void**)&ctx->h_stage, rv_ctx::STAGE_BYTES, hipHostMallocMapped) != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->h_stage = nullptr;
                }
                if (ctx->h_stage && hipHostGetDevicePointer((void**)&stage_dev, ctx->h_stage, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    stage_dev = nullptr;
                }
            }
        }
        if ((rc = shard_open_impl(s, nullptr, stage_dev, &d, lens, true, nullptr, nullptr, /*no_sync=*/true))) break;
        const size_t total = 32 + 4 * 8 + lens[0] + lens[1] + lens[2] + lens[3];
        if (dst && total > dst_cap) {
            rc = RV_E_ARG;
            break;
        }
        out = dst ? dst : (uint8_t*)out_alloc(total);
        if (!out) {
            rc = RV_E_NOMEM;
            break;
        }
        int err = 0;
        if (stage_dev) {
            if (total != need) {
                rc = RV_E_DEVICE;
                break;
            }
            const size_t err_at = (total + 15) & ~(size_t)15;
            launch_store_word(ctx->stream, s->d_err, (int*)(stage_dev + err_at));
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
                rc = hip_fail(hipGetLastError(), "proof (staged)", __FILE__, __LINE__);
                break;
            }
            memcpy(&err, ctx->h_stage + err_at, sizeof err);
            memcpy(out, ctx->h_stage, total);
        } else if (hipMemcpyAsync(out, d, total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                   hipMemcpyAsync(&err, s->d_err, sizeof err, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                   hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = hip_fail(hipGetLastError(), "proof D2H", __FILE__, __LINE__);
            break;
        }
        ctx->collect();
        if (err) {
            rc = (err & (RV_DEV_CLEAR_ABORT | RV_DEV_PERSIST_ABORT)) ? RV_E_DEVICE : RV_E_WITNESS_INVALID;
            break;
        }
        size_t off = 32;
        const uint64_t counts[4] = {RV_ONLINE_REPS, RV_PREPROCESSING_REPS, RV_ONLINE_REPS, RV_PREPROCESSING_REPS};
        for (int k = 0; k < 4; k++) {
            put_le64(out + off, counts[k]);
            off += 8 + lens[k];
        }
        *proof = out;
        *proof_len = total;
        out = nullptr;
    } while (0);
    rv_shard_destroy(s);  // (waits for both streams: nothing writes into `out` any more)
    for (hipEvent_t e : rs.ev) ctx->sync_pool.push_back(e);
    if (!dst) rv_free(out);
    return rc;
}

static int rv_prove_device_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                                size_t n_z64, const uint8_t* seeds, void* dst_device, uint8_t comm[RV_HASH_SIZE],
                                uint8_t omit[RV_TOTAL_REPS], size_t lens[4]) {
    if (!ctx || !c || !seeds || !dst_device || !comm || !omit || !lens) return RV_E_ARG;
    rv_shard* s = nullptr;
    int rc = rv_shard_commit_impl(ctx, c, wit_gf2, n_gf2, wit_z64, n_z64, seeds, 0, RV_TOTAL_REPS, &s, /*defer_sync=*/true);
    if (rc) return rc;
    do {
        void* d = nullptr;
        if ((rc = shard_open_impl(s, nullptr, dst_device, &d, lens, false, nullptr, nullptr, /*no_sync=*/true))) break;
        uint8_t back[RV_TOTAL_REPS + 32];  // omit of the (whole) shard, then comm
        int err = 0;
        if (hipMemcpyAsync(back, s->d_omit, sizeof back, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&err, s->d_err, sizeof err, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = hip_fail(hipGetLastError(), "rv_prove_device", __FILE__, __LINE__);
            break;
        }
        ctx->collect();
        if (err) {
            rc = (err & (RV_DEV_CLEAR_ABORT | RV_DEV_PERSIST_ABORT)) ? RV_E_DEVICE : RV_E_WITNESS_INVALID;
            break;
        }
        memcpy(omit, back, RV_TOTAL_REPS);
        memcpy(comm, back + RV_TOTAL_REPS, 32);
    } while (0);
    rv_shard_destroy(s);
    return rc;
}

extern "C" int rv_prove_device(rv_ctx* ctx, const rv_circuit* c, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64,
                               size_t n_z64, const uint8_t* seeds, void* dst_device, uint8_t comm[RV_HASH_SIZE],
                               uint8_t omit[RV_TOTAL_REPS], size_t lens[4]) {
    try {  // no C++ exception may cross the C boundary
        return rv_prove_device_impl(ctx, c, wit_gf2, n_gf2, wit_z64, n_z64, seeds, dst_device, comm, omit, lens);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

// ------------------------------------------------------------------------------------
// rv_prove_batch: `batch` proofs of ONE circuit, different witnesses and seeds, in one pass.
// Deep narrow circuits (AES, SHA: thousands of dependency levels of a few gates) are latency-bound: one proof keeps
// a single workgroup busy, and proofs in flight on separate contexts stop scaling at ~16 (host launch overhead).
// Here every level / narrow-run launch carries all proofs (grid.y resp. one workgroup per proof), so the latency
// chain is paid once per batch; the per-proof phases around it (keys, masks, digests, openings) are queued proof
// after proof on the same stream with a single synchronisation at the end.
// ------------------------------------------------------------------------------------
// Issues the calls recorded for the proofs of a batch (launch.h): every kernel step once, with gridDim.y = proof and
// the argument blocks in a device array; recorded copies one by one.  The staging buffers go to the caller's lists.
static int replay_recorded(rv_ctx* ctx, std::vector<LaunchRecorder>& recs, std::vector<void*>& pinned_tmp, std::vector<void*>& device_tmp) {
    const size_t batch = recs.size(), n = recs[0].calls.size();
    for (size_t b = 1; b < batch; b++)
        if (recs[b].calls.size() != n) return RV_E_DEVICE;
    size_t total = 0;
    std::vector<size_t> off(n, 0);
    for (size_t i = 0; i < n; i++) {
        const auto& c0 = recs[0].calls[i];
        for (size_t b = 1; b < batch; b++) {
            const auto& c = recs[b].calls[i];
            if (c.replay != c0.replay || c.grid.x != c0.grid.x || c.block.x != c0.block.x || c.arg_bytes != c0.arg_bytes) return RV_E_DEVICE;
        }
        if (!c0.replay) continue;
        if (c0.grid.y != 1 || c0.grid.z != 1) return RV_E_DEVICE;
        off[i] = total;
        total += ((size_t)c0.arg_bytes * batch + 15) & ~(size_t)15;
    }
    uint8_t* d_args = nullptr;
    if (total) {
        uint8_t* h = (uint8_t*)g_pinned.get(std::max<size_t>(total, PinnedPool::MIN_BYTES));
        if (!h) return RV_E_NOMEM;
        pinned_tmp.push_back(h);
        for (size_t i = 0; i < n; i++) {
            const uint32_t ab = recs[0].calls[i].arg_bytes;
            if (!recs[0].calls[i].replay) continue;
            for (size_t b = 0; b < batch; b++) memcpy(h + off[i] + b * ab, recs[b].calls[i].args.data(), ab);
        }
        int r = dalloc(ctx, total, &d_args);
        if (r) return r;
        device_tmp.push_back(d_args);
        if (hipMemcpyAsync(d_args, h, total, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return RV_E_DEVICE;
    }
    for (size_t i = 0; i < n; i++) {
        const auto& c0 = recs[0].calls[i];
        if (c0.replay) {
            c0.replay(ctx->stream, c0.grid, c0.block, d_args + off[i], (unsigned)batch);
        } else {
            for (size_t b = 0; b < batch; b++) {
                const auto& c = recs[b].calls[i];
                if (hipMemcpyAsync(c.dst, c.src, c.n, c.kind, ctx->stream) != hipSuccess) return RV_E_DEVICE;
            }
        }
    }
    if (hipGetLastError() != hipSuccess) return RV_E_DEVICE;
    for (auto& r : recs) r.calls.clear();
    return RV_OK;
}

static int rv_prove_batch_impl(rv_ctx* ctx, const rv_circuit* c, size_t batch, const uint8_t* wit_gf2, size_t n_gf2,
                               const uint64_t* wit_z64, size_t n_z64, const uint8_t* seeds, uint8_t** proofs, size_t* proof_lens) {
    if (!ctx || !c || !proofs || !proof_lens || !batch) return RV_E_ARG;
    const Compiled& cc = c->cc;
    for (size_t b = 0; b < batch; b++) proofs[b] = nullptr, proof_lens[b] = 0;
    if (n_gf2 < cc.n_in || n_z64 < cc.n_in64) return RV_E_WITNESS_SHORT;
    if ((cc.n_in && !wit_gf2) || (cc.n_in64 && !wit_z64)) return RV_E_ARG;
    std::vector<uint8_t> os_seeds;
    if (!seeds) {  // OsRng (proof/mod.rs:131-134)
        os_seeds.resize(batch * RV_TOTAL_REPS * 16);
        size_t got = 0;
        while (got < os_seeds.size()) {
            ssize_t n = getrandom(os_seeds.data() + got, os_seeds.size() - got, 0);
            if (n <= 0) return RV_E_DEVICE;
            got += (size_t)n;
        }
        seeds = os_seeds.data();
    }
    auto one_by_one = [&]() {  // Z64 / mixed circuits and batches of one: the plain entry point, proof after proof
        for (size_t b = 0; b < batch; b++) {
            int rc = rv_prove(ctx, c, wit_gf2 ? wit_gf2 + b * n_gf2 : nullptr, n_gf2, wit_z64 ? wit_z64 + b * n_z64 : nullptr, n_z64,
                              seeds + b * RV_TOTAL_REPS * 16, &proofs[b], &proof_lens[b]);
            if (rc) {
                for (size_t k = 0; k <= b; k++) rv_free(proofs[k]), proofs[k] = nullptr, proof_lens[k] = 0;
                return rc;
            }
        }
        return (int)RV_OK;
    };
    if (batch == 1 || !cc.gates64.empty()) return one_by_one();
    HIPCHK(hipSetDevice(ctx->device));
    static const size_t big_gates = [] {
        const char* e = getenv("RV_BATCH_BIG_GATES");  // circuits from this many gates on take the two-proofs-in-flight path
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 20;
    }();
    if (cc.gates.size() >= big_gates) {
        // Large circuits fill the GPU on their own; what is left to gain is overlapping one proof's VALU-bound phases
        // (masks, digests) with another's memory-bound interpreter, and a third one's 50 MB trip over PCIe.  A few host
        // threads, each with its own worker context (stream + arena; the circuit's device arrays are shared read-only),
        // prove alternate statements through the ordinary single-proof path.  Host bytes in, host proof bytes out on the
        // 10^7-gate circuit: 6.7 ms for a single rv_prove, 5.9 per proof with two threads, 5.4 with three, 5.5 with four
        // (device-resident proofs: 4.9 with two in flight, no gain from a third).
        constexpr size_t T_MAX = 8;
        static const size_t T = [] {
            const char* e = getenv("RV_BATCH_THREADS");
            return (size_t)std::min(std::max(e ? atoi(e) : 3, 1), (int)T_MAX);
        }();
        static const bool worker_prio = !(getenv("RV_BATCH_PRIO") && atoi(getenv("RV_BATCH_PRIO")) == 0);
        while (ctx->workers.size() < T) {
            rv_ctx* w = nullptr;
            int rcw = ctx_create_impl(ctx->device, &w, worker_prio ? (int)ctx->workers.size() : -1);
            if (rcw) return rcw;
            ctx->workers.push_back(w);
        }
        // every proof of the batch lands in a slice of ONE page-locked buffer (released when the last proof has been
        // rv_free'd): a buffer per proof meant a hipHostMalloc of tens of MB per proof as soon as the caller held more
        // proofs than the pool keeps idle -- 7.7 ms per proof at 16 proofs per call instead of 6.2 at 2
        uint8_t canon[RV_TOTAL_REPS];
        for (uint32_t r = 0; r < RV_TOTAL_REPS; r++) canon[r] = r < RV_ONLINE_REPS ? 0 : RV_PLAYERS;
        const size_t stride = (open_layout(cc, canon, RV_TOTAL_REPS, true).total + 4095) & ~(size_t)4095;
        uint8_t* slab = (uint8_t*)g_pinned.get(std::max<size_t>(stride * batch, PinnedPool::MIN_BYTES));
        if (!slab) return RV_E_NOMEM;
        int rcs[T_MAX] = {};
        std::thread th[T_MAX];
        for (size_t t = 0; t < T; t++)
            th[t] = std::thread([&, t] {
                try {
                    if (hipSetDevice(ctx->device) != hipSuccess) {
                        rcs[t] = RV_E_DEVICE;
                        return;
                    }
                    for (size_t b = t; b < batch && rcs[t] == RV_OK; b += T)
                        rcs[t] = rv_prove_impl(ctx->workers[t], c, wit_gf2 ? wit_gf2 + b * n_gf2 : nullptr, n_gf2, nullptr, 0,
                                               seeds + b * RV_TOTAL_REPS * 16, &proofs[b], &proof_lens[b], slab + b * stride, stride);
                } catch (...) {
                    rcs[t] = RV_E_NOMEM;
                }
            });
        for (size_t t = 0; t < T; t++) th[t].join();
        for (int r : rcs)
            if (r) {
                for (size_t b = 0; b < batch; b++) proofs[b] = nullptr, proof_lens[b] = 0;
                g_pinned.put(slab);
                return r;
            }
        g_pinned.share(slab, batch);  // from here on the proofs own it
        ctx->prof.calls += batch;
        return RV_OK;
    }
    std::vector<rv_shard*> sh(batch, nullptr);
    std::vector<InterpParams> pp(batch);
    InterpParams* d_pp = nullptr;
    int rc = RV_OK;
    uint8_t* staging = nullptr;
    std::vector<void*> pinned_tmp;  // argument blocks of the replayed launches (page-locked, returned at the end)
    std::vector<void*> device_tmp;
    auto cleanup = [&](int code) {
        g_recorder = nullptr;
        (void)hipStreamSynchronize(ctx->stream);
        for (rv_shard* s : sh)
            if (s) {
                s->destroy();
                delete s;
            }
        ctx->release(d_pp);
        for (void* q : device_tmp) ctx->release(q);
        for (void* q : pinned_tmp) g_pinned.put(q);
        if (staging) g_pinned.put(staging);
        if (code)
            for (size_t b = 0; b < batch; b++) rv_free(proofs[b]), proofs[b] = nullptr, proof_lens[b] = 0;
        return code;
    };
    const uint32_t R = RV_TOTAL_REPS;
    // The per-proof phases are strings of small kernels, the same string with the same grids for every proof of the
    // circuit.  Each proof's string is RECORDED (launch.h) instead of launched, then every step is issued once for the
    // whole batch (gridDim.y = proof, arguments from a device array): 35 launches per batch instead of 35 per proof.
    std::vector<LaunchRecorder> recs(batch);
    for (auto& r : recs) r.batch = (unsigned)batch;
    struct RecorderOff {  // whatever way this function is left (an exception included), launches go to the stream again
        ~RecorderOff() { g_recorder = nullptr; }
    } recorder_off;
    static const bool stats = getenv("RV_BATCH_STATS") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!stats) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[rv batch] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    auto replay = [&]() -> int { return replay_recorded(ctx, recs, pinned_tmp, device_tmp); };
    // ---- per proof: seeds, witness, keys, masks, buffers.  What the host sends or fetches per proof (seeds, witness,
    // error flag, the proof itself) lives in ONE allocation per kind, a slot per proof, so that it moves in one copy
    // per batch instead of one per proof (1 280 small copies were a fifth of a batch's GPU time).  The slots start
    // 256 bytes into their slab: no proof's pointer equals an arena block, the slabs are released exactly once below.
    constexpr size_t SLAB_HEAD = 256;
    const size_t wit_stride = (std::max<size_t>(cc.n_in, 1) + 15) & ~(size_t)15;
    uint8_t canon[RV_TOTAL_REPS];
    for (uint32_t r = 0; r < RV_TOTAL_REPS; r++) canon[r] = r < RV_ONLINE_REPS ? 0 : RV_PLAYERS;
    const size_t out_stride = (open_layout(cc, canon, R, true).total + 4 + 255) & ~(size_t)255;
    uint8_t *d_seeds_all = nullptr, *d_wit_all = nullptr, *d_out_all = nullptr;
    int* d_err_all = nullptr;
    if ((rc = dalloc(ctx, SLAB_HEAD + batch * (size_t)R * 16, &d_seeds_all))) return cleanup(rc);
    device_tmp.push_back(d_seeds_all);
    if ((rc = dalloc(ctx, SLAB_HEAD + batch * wit_stride, &d_wit_all))) return cleanup(rc);
    device_tmp.push_back(d_wit_all);
    if ((rc = dalloc(ctx, SLAB_HEAD + batch * out_stride, &d_out_all))) return cleanup(rc);
    device_tmp.push_back(d_out_all);
    if ((rc = dalloc(ctx, SLAB_HEAD / sizeof(int) + batch, &d_err_all))) return cleanup(rc);
    device_tmp.push_back(d_err_all);
    if (hipMemcpyAsync(d_seeds_all + SLAB_HEAD, seeds, batch * (size_t)R * 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        (cc.n_in && hipMemcpy2DAsync(d_wit_all + SLAB_HEAD, wit_stride, wit_gf2, n_gf2, cc.n_in, batch, hipMemcpyHostToDevice,
                                     ctx->stream) != hipSuccess))
        return cleanup(RV_E_DEVICE);
    for (size_t b = 0; b < batch && !rc; b++) {
        rv_shard* s = sh[b] = new rv_shard();
        s->ctx = ctx;
        s->c = c;
        s->rep_begin = 0;
        s->R = R;
        s->NQ = R / 4;
        s->d_seeds = d_seeds_all + SLAB_HEAD + b * (size_t)R * 16;
        s->d_wit = d_wit_all + SLAB_HEAD + b * wit_stride;
        s->d_err = d_err_all + SLAB_HEAD / sizeof(int) + b;
        if ((rc = dalloc(ctx, (size_t)R * 128, &s->d_keys))) break;
        g_recorder = &recs[b];
        launch_expand_seeds(ctx->stream, s->d_seeds, R, s->d_keys);
        Interp64Params p64{};
        pp[b] = InterpParams{};
        pp[b].wit = s->d_wit;
        if (!(rc = shard_setup_prg(s, nullptr))) rc = shard_run_alloc(s, pp[b], p64);
        g_recorder = nullptr;
    }
    if (rc) return cleanup(rc);
    mark("record setup/masks");
    ctx->phase(RV_PH_MASKS);  // (whole-batch phases: keys + masks, interpreter, digests + openings)
    if ((rc = replay())) return cleanup(rc);
    mark("replay setup/masks");
    // ---- all proofs level by level
    ctx->phase(RV_PH_INTERP);
    if ((rc = dalloc(ctx, batch, &d_pp))) return cleanup(rc);
    if (hipMemcpyAsync(d_pp, pp.data(), batch * sizeof(InterpParams), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return cleanup(RV_E_DEVICE);
    {
        const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
        for (size_t l = 0; l < n_levels; l++) {
            if (lds_run_for_batch(c, l, batch)) {
                const auto& pl = c->lds_runs[(size_t)c->lds_run_of_level[l]];
                if (l == pl.run.l0)
                    launch_interp_lds(ctx->stream, MODE_PROVE, pl.qs, RV_TOTAL_REPS / 4, c->d_lds_recs + pl.run.rec0, pl.run.n_steps, pl.run.n_slots,
                                      pl.run.eo0, pl.run.ep0, InterpParams{}, d_pp, (uint32_t)batch);
                continue;
            }
            if (c->run_of_level[l] >= 0) {
                const auto& run = c->narrow_runs[(size_t)c->run_of_level[l]];
                if (l == run.first)
                    launch_interp_narrow_batched(ctx->stream, c->d_gates, c->d_level_range, run.first, run.second, run.tiny, d_pp, (uint32_t)batch);
                continue;
            }
            launch_interp_batched(ctx->stream, c->d_gates, cc.level_range[l], d_pp, (uint32_t)batch);
        }
    }
    mark("interpreter launches");
    // ---- per proof: digests, commitment + challenge + openings on the device, proof bytes to the host
    size_t slot = 0;
    size_t lens[4] = {0, 0, 0, 0};  // the same for every proof of the circuit (40 / 216 split)
    for (size_t b = 0; b < batch && !rc; b++) {
        rv_shard* s = sh[b];
        void* d = nullptr;
        g_recorder = &recs[b];
        if (!(rc = shard_run_hash(s)) && !(rc = shard_join(s)))
            rc = shard_open_impl(s, nullptr, d_out_all + SLAB_HEAD + b * out_stride, &d, lens, true, nullptr, nullptr, /*no_sync=*/true);
        g_recorder = nullptr;
    }
    if (rc) return cleanup(rc);
    mark("record digests/openings");
    ctx->phase(RV_PH_HASH);
    if ((rc = replay())) return cleanup(rc);
    ctx->phase(-1);
    mark("replay digests/openings");
    const size_t total = 32 + 4 * 8 + lens[0] + lens[1] + lens[2] + lens[3];
    {
        // one page-locked staging area for the whole batch (a device-to-host copy into pageable memory would block the
        // host until the kernels have run): the proofs at their device stride, then the error flags
        if (total > out_stride) return cleanup(RV_E_DEVICE);
        slot = out_stride;
        staging = (uint8_t*)g_pinned.get(std::max<size_t>(slot * batch + batch * sizeof(int), PinnedPool::MIN_BYTES));
        if (!staging) return cleanup(RV_E_NOMEM);
        if (hipMemcpyAsync(staging, d_out_all + SLAB_HEAD, slot * batch, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(staging + slot * batch, d_err_all + SLAB_HEAD / sizeof(int), batch * sizeof(int), hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess)
            return cleanup(RV_E_DEVICE);
        for (size_t b = 0; b < batch; b++) proof_lens[b] = total;
    }
    mark("queue device-to-host copies");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(hip_fail(hipGetLastError(), "batch sync", __FILE__, __LINE__));
    ctx->collect();
    mark("wait for the GPU");
    for (size_t b = 0; b < batch; b++) {
        int err = 0;
        memcpy(&err, staging + slot * batch + b * sizeof(int), sizeof err);
        if (err) return cleanup(RV_E_WITNESS_INVALID);
    }
    {
        // The proofs are handed out where they landed: slices of the page-locked staging buffer, which returns to the
        // pool when the last of them has been rv_free'd (no second copy into 256 freshly mapped buffers, no 256
        // munmaps in the caller: together they cost more than the GPU work of an AES-128 batch).  RV_BATCH_COPY_OUT=1
        // gives every proof its own malloc'ed buffer instead (callers that keep single proofs of many batches alive).
        static const bool copy_out = getenv("RV_BATCH_COPY_OUT") != nullptr;
        const uint64_t counts[4] = {RV_ONLINE_REPS, RV_PREPROCESSING_REPS, RV_ONLINE_REPS, RV_PREPROCESSING_REPS};
        if (copy_out) {
            for (size_t b = 0; b < batch; b++) {
                proofs[b] = (uint8_t*)malloc(proof_lens[b]);
                if (!proofs[b]) return cleanup(RV_E_NOMEM);
                memcpy(proofs[b], staging + b * slot, proof_lens[b]);
            }
        } else {
            g_pinned.share(staging, batch);
            for (size_t b = 0; b < batch; b++) proofs[b] = staging + b * slot;
            staging = nullptr;  // owned by the proofs now
        }
        for (size_t b = 0; b < batch; b++) {
            size_t off = 32;
            for (int k = 0; k < 4; k++) {
                put_le64(proofs[b] + off, counts[k]);
                off += 8 + lens[k];
            }
        }
    }
    ctx->prof.calls += batch;
    mark("copy proofs out");
    rc = cleanup(RV_OK);
    mark("cleanup");
    return rc;
}

extern "C" int rv_prove_batch(rv_ctx* ctx, const rv_circuit* c, size_t batch, const uint8_t* wit_gf2, size_t n_gf2,
                              const uint64_t* wit_z64, size_t n_z64, const uint8_t* seeds, uint8_t** proofs, size_t* proof_lens) {
    try {  // no C++ exception may cross the C boundary
        return rv_prove_batch_impl(ctx, c, batch, wit_gf2, n_gf2, wit_z64, n_z64, seeds, proofs, proof_lens);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

// ------------------------------------------------------------------------------------
// proof parsing (bincode 1.3 fixint)
// ------------------------------------------------------------------------------------
namespace {
struct OnRec {
    uint8_t omit;
    size_t keys, rec, corr, in;  // offsets into the proof
    uint64_t n_rec, n_corr, n_in;
};
struct PreRec {
    size_t seed, comm_online;
};
struct Single {
    std::vector<OnRec> on;
    std::vector<PreRec> pre;
};
struct Parsed {
    Single gf2, z64;
};
struct Reader {
    const uint8_t* p;
    size_t len, pos = 0;
    bool bad = false;
    size_t take(uint64_t n) {
        if (bad || n > len - pos) {
            bad = true;
            return 0;
        }
        size_t at = pos;
        pos += (size_t)n;
        return at;
    }
    uint64_t u64() {
        size_t at = take(8);
        if (bad) return 0;
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)p[at + i] << (8 * i);
        return v;
    }
};
bool parse_single(Reader& r, Single& s) {
    uint64_t n = r.u64();
    if (r.bad || n > (r.len - r.pos) / 153 + 1) return false;
    s.on.resize((size_t)n);
    for (auto& o : s.on) {
        size_t at = r.take(1);
        if (r.bad) return false;
        o.omit = r.p[at];
        o.keys = r.take(128);
        o.n_rec = r.u64();
        o.rec = r.take(o.n_rec);
        o.n_corr = r.u64();
        o.corr = r.take(o.n_corr);
        o.n_in = r.u64();
        o.in = r.take(o.n_in);
        if (r.bad) return false;
    }
    n = r.u64();
    if (r.bad || n > (r.len - r.pos) / 48 + 1) return false;
    s.pre.resize((size_t)n);
    for (auto& q : s.pre) {
        q.seed = r.take(16);
        q.comm_online = r.take(32);
        if (r.bad) return false;
    }
    return true;
}
// returns RV_OK / RV_E_PROOF_MALFORMED; trailing bytes are ignored like bincode::deserialize_from (main.rs:101-103)
int parse_proof(const uint8_t* proof, size_t len, Parsed& out) {
    Reader r{proof, len};
    r.take(32);
    if (r.bad || !parse_single(r, out.gf2) || !parse_single(r, out.z64)) return RV_E_PROOF_MALFORMED;
    return RV_OK;
}
bool format_ok(const Parsed& p) {  // ProofSingle::check_format, proof/mod.rs:110-114
    return p.gf2.on.size() == RV_ONLINE_REPS && p.gf2.pre.size() == RV_PREPROCESSING_REPS &&
           p.z64.on.size() == RV_ONLINE_REPS && p.z64.pre.size() == RV_PREPROCESSING_REPS;
}
}  // namespace

static int rv_verify_shard_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t slot_begin,
                               uint32_t slot_count, uint8_t* digests, int* zero_checks_ok);

extern "C" int rv_verify_shard_ex(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t slot_begin,
                                  uint32_t slot_count, uint8_t* digests, int* zero_checks_ok) {
    try {  // no C++ exception may cross the C boundary
        return rv_verify_shard_impl(ctx, c, proof, proof_len, slot_begin, slot_count, digests, zero_checks_ok);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

extern "C" int rv_verify_shard(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t slot_begin,
                               uint32_t slot_count, uint8_t* digests) {
    return rv_verify_shard_ex(ctx, c, proof, proof_len, slot_begin, slot_count, digests, nullptr);
}

static int rv_verify_shard_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t slot_begin,
                               uint32_t slot_count, uint8_t* digests, int* zero_checks_ok) {
    if (!ctx || !c || !proof || !digests) return RV_E_ARG;
    if (slot_count == 0 || slot_count % 8 || slot_begin % 8 || slot_begin + slot_count > RV_TOTAL_REPS) return RV_E_ARG;
    Parsed P;
    int rc = parse_proof(proof, proof_len, P);
    if (rc) return rc;
    if (!format_ok(P)) return RV_E_PROOF_MALFORMED;  // callers check the format first (rv_verify returns ok=0)
    const Compiled& cc = c->cc;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t R = slot_count, NQ = R / 4;

    // ---- host-side preparation of the slots (VerifierTranscriptOnline::new, online.rs:25-119;
    //      VerifierTranscriptPreprocess::new, preprocess.rs:17-43)
    std::vector<uint8_t> seeds((size_t)R * 16, 0), omit(R, 8), seeds64((size_t)R * 16, 0), omit64(R, 8);
    std::vector<uint64_t> src((size_t)6 * R, 0);  // rec off,len ; corr off,len ; in off,len
    std::vector<uint64_t> src64((size_t)6 * R, 0);
    std::vector<uint32_t> keep(NQ, 0xFFFFFFFFu), onm(NQ, 0), keep64(NQ, 0xFFFFFFFFu);
    const bool has64 = !cc.gates64.empty();
    for (uint32_t g0 = 0; g0 < R; g0 += 8) {
        const uint32_t slot0 = slot_begin + g0;
        if (slot0 < RV_ONLINE_REPS) {
            const OnRec* o = &P.gf2.on[slot0];
            const OnRec* z = &P.z64.on[slot0];
            for (int i = 0; i < 8; i++) {
                if (o[i].omit >= 8 || z[i].omit >= 8) return RV_E_PROOF_MALFORMED;  // UB upstream (gf2/share.rs:167-199)
                // Recon::unpack indexes every vector up to the first one's length (gf2/recon.rs:241-259)
                if (o[i].n_corr < o[0].n_corr || o[i].n_in < o[0].n_in) return RV_E_PROOF_MALFORMED;
                // Share::unpack_selected asserts equal lengths (gf2/share.rs:157-164)
                if (o[i].n_rec != o[0].n_rec) return RV_E_PROOF_MALFORMED;
                const uint32_t r = g0 + i;
                omit[r] = o[i].omit;
                src[0 * R + r] = o[i].rec;
                src[1 * R + r] = o[0].n_rec;
                src[2 * R + r] = o[i].corr;
                src[3 * R + r] = o[0].n_corr;
                src[4 * R + r] = o[i].in;
                src[5 * R + r] = o[0].n_in;
                keep[r / 4] &= ~(1u << (31 - 8 * (r % 4) - o[i].omit));  // BatchGen skips the omitted player
                onm[r / 4] |= 0xFFu << (24 - 8 * (r % 4));
                // Z64 vectors: length of the group's first record, missing chunks read as zero
                // (z64/recon.rs:68-108, z64/share.rs:51-91)
                omit64[r] = z[i].omit;
                keep64[r / 4] &= ~(1u << (31 - 8 * (r % 4) - z[i].omit));
                src64[0 * R + r] = z[i].rec;
                src64[1 * R + r] = std::min(z[i].n_rec, z[0].n_rec / 8 * 8);
                src64[2 * R + r] = z[i].corr;
                src64[3 * R + r] = std::min(z[i].n_corr, z[0].n_corr / 8 * 8);
                src64[4 * R + r] = z[i].in;
                src64[5 * R + r] = std::min(z[i].n_in, z[0].n_in / 8 * 8);
            }
        } else {
            const PreRec* q = &P.gf2.pre[slot0 - RV_ONLINE_REPS];
            const PreRec* q64 = &P.z64.pre[slot0 - RV_ONLINE_REPS];
            for (int i = 0; i < 8; i++) {
                memcpy(&seeds[(size_t)(g0 + i) * 16], proof + q[i].seed, 16);
                memcpy(&seeds64[(size_t)(g0 + i) * 16], proof + q64[i].seed, 16);
            }
        }
    }

    rv_shard* s = new rv_shard();
    s->ctx = ctx;
    s->c = c;
    s->rep_begin = slot_begin;
    s->R = R;
    s->NQ = NQ;
    auto fail = [&](int code) {
        rv_shard_destroy(s);
        return code;
    };
    uint8_t* d_proof = nullptr;
    uint64_t* d_src = nullptr;
    hipEvent_t ev_arena = nullptr;
    uint32_t *d_keep = nullptr, *d_onm = nullptr, *d_sup_in = nullptr, *d_sup_corr = nullptr, *d_sup_rec = nullptr;
    auto track = [&](void* p) { s->extra.push_back(p); };
    std::vector<uint32_t> on_quads;
    for (uint32_t q = 0; q < NQ; q++)
        if (onm[q]) on_quads.push_back(q);
    // rows of the supplied values: sixteen quad words (two sectors, written whole) when the opened repetitions sit in the first
    // sixteen -- the verifier's slot order puts them into the first ten -- instead of full share rows
    const uint32_t sup_nq = (NQ > 16 && (on_quads.empty() || on_quads.back() < 16)) ? 16u : NQ;
    uint32_t sup_r = R;  // ... and the Z64 ones: the first 64 repetitions when no other is opened there
    if (R > 64) {
        sup_r = 64;
        for (uint32_t r = 64; r < R; r++)
            if (omit64[r] < 8) sup_r = R;
    }
    // ---- staging (one copy each instead of one per repetition): opened player keys (online.rs:101-113) and the
    //      online commitments the preprocessing slots carry over from the proof (preprocess.rs:55-57)
    std::vector<uint8_t> hkeys((size_t)R * 128, 0), hco((size_t)R * 32, 0), hkeys64, hco64((size_t)R * 32, 0);
    if (has64) hkeys64.assign((size_t)R * 128, 0);
    for (uint32_t r = 0; r < R; r++) {
        if (omit[r] < 8) {
            memcpy(&hkeys[(size_t)r * 128], proof + P.gf2.on[slot_begin + r].keys, 128);
            if (has64 && omit64[r] < 8) memcpy(&hkeys64[(size_t)r * 128], proof + P.z64.on[slot_begin + r].keys, 128);
        } else {
            const uint32_t k = slot_begin + r - RV_ONLINE_REPS;
            memcpy(&hco[(size_t)r * 32], proof + P.gf2.pre[k].comm_online, 32);
            memcpy(&hco64[(size_t)r * 32], proof + P.z64.pre[k].comm_online, 32);
        }
    }
    uint32_t* d_on_quads = nullptr;
    uint8_t *d_hkeys = nullptr, *d_hco = nullptr, *d_hkeys64 = nullptr, *d_hco64 = nullptr;
    // A small GF(2) proof goes over in ONE copy: every host array above and the proof itself are packed into the
    // page-locked input staging buffer and land in one device block (ten pageable copies of ~10 us each otherwise).
    // The function waits for the stream before it returns, so the buffer is free again by the next call.
    size_t blob_bytes = 0;
    auto seg = [&](size_t len) {
        const size_t o = blob_bytes;
        blob_bytes += (len + 15) & ~(size_t)15;
        return o;
    };
    const size_t o_seeds = seg(seeds.size()), o_omit = seg(omit.size()), o_keep = seg((size_t)NQ * 4), o_onm = seg((size_t)NQ * 4),
                 o_onq = seg(std::max<size_t>(on_quads.size(), 1) * 4), o_hkeys = seg(hkeys.size()), o_hco = seg(hco.size()),
                 o_hco64 = seg(hco64.size()), o_src = seg(src.size() * 8), o_proof = seg(proof_len);
    static const bool small_stage = !(getenv("RV_SMALL_STAGE") && atoi(getenv("RV_SMALL_STAGE")) == 0);
    bool blob = small_stage && !has64 && !g_recorder && blob_bytes <= rv_ctx::IN_STAGE_BYTES;
    if (blob && !ctx->h_in && hipHostMalloc((void**)&ctx->h_in, rv_ctx::IN_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->h_in = nullptr;
        blob = false;
    }
    if (blob) {
        if ((rc = dalloc(ctx, blob_bytes, &s->d_seeds)) || (rc = dalloc(ctx, (size_t)R * 128, &s->d_keys))) return fail(rc);
        uint8_t* b = s->d_seeds;  // (pointers inside d_seeds' block: the arena ignores them on release)
        s->d_omit = b + o_omit;
        d_keep = (uint32_t*)(b + o_keep);
        d_onm = (uint32_t*)(b + o_onm);
        d_on_quads = (uint32_t*)(b + o_onq);
        d_hkeys = b + o_hkeys;
        d_hco = b + o_hco;
        d_hco64 = b + o_hco64;
        d_src = (uint64_t*)(b + o_src);
        d_proof = b + o_proof;
        uint8_t* h = ctx->h_in;
        memcpy(h + o_seeds, seeds.data(), seeds.size());
        memcpy(h + o_omit, omit.data(), omit.size());
        memcpy(h + o_keep, keep.data(), (size_t)NQ * 4);
        memcpy(h + o_onm, onm.data(), (size_t)NQ * 4);
        if (!on_quads.empty()) memcpy(h + o_onq, on_quads.data(), on_quads.size() * 4);
        memcpy(h + o_hkeys, hkeys.data(), hkeys.size());
        memcpy(h + o_hco, hco.data(), hco.size());
        memcpy(h + o_hco64, hco64.data(), hco64.size());
        memcpy(h + o_src, src.data(), src.size() * 8);
        memcpy(h + o_proof, proof, proof_len);
    } else {
        if ((rc = dalloc(ctx, (size_t)R * 16, &s->d_seeds)) || (rc = dalloc(ctx, (size_t)R * 128, &s->d_keys)) ||
            (rc = dalloc(ctx, R, &s->d_omit)))
            return fail(rc);
        if ((rc = dalloc(ctx, proof_len, &d_proof))) return fail(rc);
        track(d_proof);
        if ((rc = dalloc(ctx, src.size(), &d_src))) return fail(rc);
        track(d_src);
        // d_proof / d_src may be filled from the SECOND stream further down (beside the mask kernels).  The arena hands blocks out
        // in the main stream's order, so the side stream first waits for everything the main stream holds NOW -- whatever used
        // these blocks last -- and nothing of this call's own kernels (they are queued after this point)
        if (!g_recorder && proof_len >= ((size_t)4 << 20)) {
            ev_arena = ctx->get_sync_event();
            s->misc_events.push_back(ev_arena);
            if (hipEventRecord(ev_arena, ctx->stream) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipEventRecord", __FILE__, __LINE__));
        }
        if ((rc = dalloc(ctx, NQ, &d_keep))) return fail(rc);
        track(d_keep);
        if ((rc = dalloc(ctx, NQ, &d_onm))) return fail(rc);
        track(d_onm);
        if ((rc = dalloc(ctx, std::max<size_t>(on_quads.size(), 1), &d_on_quads))) return fail(rc);
        track(d_on_quads);
        if ((rc = dalloc(ctx, hkeys.size(), &d_hkeys))) return fail(rc);
        track(d_hkeys);
        if ((rc = dalloc(ctx, hco.size(), &d_hco))) return fail(rc);
        track(d_hco);
        if ((rc = dalloc(ctx, hco64.size(), &d_hco64))) return fail(rc);
        track(d_hco64);
    }
    if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_in, 1) * sup_nq, &d_sup_in))) return fail(rc);
    track(d_sup_in);
    if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_pre, 1) * sup_nq, &d_sup_corr))) return fail(rc);
    track(d_sup_corr);
    if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_rec, 1) * sup_nq, &d_sup_rec))) return fail(rc);
    track(d_sup_rec);
    uint64_t *d_src64 = nullptr, *d_sup_in64 = nullptr, *d_sup_corr64 = nullptr, *d_sup_rec64 = nullptr;
    uint32_t* d_keep64 = nullptr;
    uint8_t* d_seeds64 = nullptr;
    if (has64) {
        if ((rc = dalloc(ctx, (size_t)R * 16, &d_seeds64))) return fail(rc);
        track(d_seeds64);
        if ((rc = dalloc(ctx, (size_t)R * 128, &s->d_keys64)) || (rc = dalloc(ctx, R, &s->d_omit64))) return fail(rc);
        if ((rc = dalloc(ctx, src64.size(), &d_src64))) return fail(rc);
        track(d_src64);
        if ((rc = dalloc(ctx, NQ, &d_keep64))) return fail(rc);
        track(d_keep64);
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_in64, 1) * sup_r, &d_sup_in64))) return fail(rc);
        track(d_sup_in64);
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_corr64, 1) * sup_r, &d_sup_corr64))) return fail(rc);
        track(d_sup_corr64);
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_rec64, 1) * sup_r, &d_sup_rec64))) return fail(rc);
        track(d_sup_rec64);
    }
#define HC(x)                                 \
    do {                                      \
        if ((x) != hipSuccess) {              \
            hip_fail(hipGetLastError(), #x, __FILE__, __LINE__); \
            return fail(RV_E_DEVICE);         \
        }                                     \
    } while (0)
    const size_t DW = (size_t)R * 8;
    if (has64) {
        if ((rc = dalloc(ctx, hkeys64.size(), &d_hkeys64))) return fail(rc);
        track(d_hkeys64);
    }
    // ---- stream 1: everything the mask generator needs, then the masks themselves
    if (blob) {
        HC(hipMemcpyAsync(s->d_seeds, ctx->h_in, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HC(hipMemcpyAsync(s->d_seeds, seeds.data(), seeds.size(), hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(s->d_omit, omit.data(), omit.size(), hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_keep, keep.data(), NQ * 4, hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_onm, onm.data(), NQ * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!on_quads.empty()) HC(hipMemcpyAsync(d_on_quads, on_quads.data(), on_quads.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_hkeys, hkeys.data(), hkeys.size(), hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_hco, hco.data(), hco.size(), hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_hco64, hco64.data(), hco64.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    s->d_on_quads = d_on_quads;
    s->n_on_quads = (uint32_t)on_quads.size();
    ctx->phase(RV_PH_SETUP);
    ctx->count(2);
    // (the side stream's unpack kernels read d_omit: uploaded by now)
    hipEvent_t ev_inputs = nullptr, ev_inputs64 = nullptr;
    if (ev_arena) {
        ev_inputs = ctx->get_sync_event();
        s->misc_events.push_back(ev_inputs);
        HC(hipEventRecord(ev_inputs, ctx->stream));
    }
    launch_expand_seeds(ctx->stream, s->d_seeds, R, s->d_keys);
    launch_overlay_rows(ctx->stream, (uint32_t*)s->d_keys, (const uint32_t*)d_hkeys, s->d_omit, R, 32, 1);
    if (has64) {
        HC(hipMemcpyAsync(d_seeds64, seeds64.data(), seeds64.size(), hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(s->d_omit64, omit64.data(), omit64.size(), hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_keep64, keep64.data(), NQ * 4, hipMemcpyHostToDevice, ctx->stream));
        HC(hipMemcpyAsync(d_hkeys64, hkeys64.data(), hkeys64.size(), hipMemcpyHostToDevice, ctx->stream));
        launch_expand_seeds(ctx->stream, d_seeds64, R, s->d_keys64);
        launch_overlay_rows(ctx->stream, (uint32_t*)s->d_keys64, (const uint32_t*)d_hkeys64, s->d_omit64, R, 32, 1);
        ctx->count(2);
        if (ev_arena) {
            ev_inputs64 = ctx->get_sync_event();
            s->misc_events.push_back(ev_inputs64);
            HC(hipEventRecord(ev_inputs64, ctx->stream));
        }
    }
    ctx->phase(-1);
    // The verifier's Z64 half through k_z64_fused<VERIFY> as well (RV_Z64_FUSED_VERIFY=0: k_aes_z64_masks, then k_interp64 per
    // level).  Its kernels take 40 ms instead of 53 on the 10^6-MUL circuit; the 640 MB proof's 12 ms of PCIe and the unpack
    // kernels, which used to hide beside the mask generator, hide beside the quad groups that hold no opened repetition (split64
    // below): rv_verify 59.5 -> 52.5 ms.
    s->z64f = has64 && c->z64f_ok && z64_fused_on() && z64_fused_supports(NQ) && !g_recorder &&
              !(getenv("RV_Z64_FUSED_VERIFY") && atoi(getenv("RV_Z64_FUSED_VERIFY")) == 0);
    if ((rc = shard_setup_prg(s, d_keep, d_keep64))) return fail(rc);
    // ---- the interpreter's stream: the proof itself (tens of MB from pageable memory: the host blocks in this
    //      copy while the mask kernels above already run) and the supplied-value rows unpacked from it
    hipStream_t sb = ctx->stream;
    hipStream_t su = sb;  // the stream of the GF(2) unpack kernels
    // on ONE stream the proof's copy would queue up behind the mask kernels; from the second stream it runs beside them
    // (copy engine next to compute) and the unpack kernels wait for its event
    static const bool side = !(getenv("RV_VERIFY_SIDE_COPY") && atoi(getenv("RV_VERIFY_SIDE_COPY")) == 0);
    // ... and so do the GF(2) supplied-value rows (round 4): three memory-bound transposes that find room beside the
    // VALU-bound mask generator instead of standing between it and the interpreter (RV_VERIFY_SIDE_UNPACK=0: behind it)
    static const bool side_unpack = !(getenv("RV_VERIFY_SIDE_UNPACK") && atoi(getenv("RV_VERIFY_SIDE_UNPACK")) == 0);
    // The fused Z64 verifier of a pure Z64 circuit whose opened repetitions all sit in the first quad group (sup_r == 64): only
    // that quad group's workgroups read supplied values, so the other groups' levels are queued FIRST (shard_run_levels), then the
    // proof's copy and the unpack kernels go to the side stream and hide beside them (a copy from pageable memory blocks the host
    // until the bytes are staged: issued up front it kept the level launches from being queued), and the first group's levels
    // wait for ev_sup64.
    const bool split64 = has64 && s->z64f && cc.gates.empty() && !blob && side && ev_arena && side_unpack && ev_inputs && ev_inputs64 && sup_r == 64 && NQ >= 32;
    if (!blob && !split64) {
        hipStream_t sc = (side && ev_arena) ? ctx->stream2 : sb;
        if (sc != sb) HC(hipStreamWaitEvent(sc, ev_arena, 0));
        HC(hipMemcpyAsync(d_proof, proof, proof_len, hipMemcpyHostToDevice, sc));
        HC(hipMemcpyAsync(d_src, src.data(), src.size() * 8, hipMemcpyHostToDevice, sc));
        if (sc != sb) {
            if (side_unpack && ev_inputs) {
                HC(hipStreamWaitEvent(sc, ev_inputs, 0));
                su = sc;
            } else {
                hipEvent_t e = ctx->get_sync_event();
                s->misc_events.push_back(e);
                HC(hipEventRecord(e, sc));
                HC(hipStreamWaitEvent(sb, e, 0));
            }
        }
    }
    if (!split64) {  // (split64: a circuit without GF(2) gates has none of these)
        launch_unpack_bits(su, d_proof, d_src + 4 * R, d_src + 5 * R, s->d_omit, cc.n_in, NQ, 1, d_sup_in, sup_nq);
        launch_unpack_bits(su, d_proof, d_src + 2 * R, d_src + 3 * R, s->d_omit, cc.n_pre, NQ, 1, d_sup_corr, sup_nq);
        launch_unpack_bits(su, d_proof, d_src + 0 * R, d_src + 1 * R, s->d_omit, cc.n_rec, NQ, 0, d_sup_rec, sup_nq);
        if (su != sb) {
            hipEvent_t e = ctx->get_sync_event();
            s->misc_events.push_back(e);
            HC(hipEventRecord(e, su));
            HC(hipStreamWaitEvent(sb, e, 0));
        }
    }
    Interp64Params p64{};
    if (has64) {
        if (split64) {
            s->ev_sup64 = ctx->get_sync_event();
            s->misc_events.push_back(s->ev_sup64);
            s->mid64 = [&, s, ctx]() -> int {
                hipStream_t sc = ctx->stream2;
                if (hipStreamWaitEvent(sc, ev_arena, 0) != hipSuccess || hipMemcpyAsync(d_proof, proof, proof_len, hipMemcpyHostToDevice, sc) != hipSuccess ||
                    hipStreamWaitEvent(sc, ev_inputs64, 0) != hipSuccess ||
                    hipMemcpyAsync(d_src64, src64.data(), src64.size() * 8, hipMemcpyHostToDevice, sc) != hipSuccess)
                    return hip_fail(hipGetLastError(), "rv_verify: the proof's copy", __FILE__, __LINE__);
                launch_unpack64(sc, d_proof, d_src64 + 4 * R, d_src64 + 5 * R, s->d_omit64, cc.n_in64, R, d_sup_in64, sup_r);
                launch_unpack64(sc, d_proof, d_src64 + 2 * R, d_src64 + 3 * R, s->d_omit64, cc.n_corr64, R, d_sup_corr64, sup_r);
                launch_unpack64(sc, d_proof, d_src64 + 0 * R, d_src64 + 1 * R, s->d_omit64, cc.n_rec64, R, d_sup_rec64, sup_r);
                if (hipEventRecord(s->ev_sup64, sc) != hipSuccess) return hip_fail(hipGetLastError(), "hipEventRecord", __FILE__, __LINE__);
                return RV_OK;
            };
        } else {
            HC(hipMemcpyAsync(d_src64, src64.data(), src64.size() * 8, hipMemcpyHostToDevice, sb));
            launch_unpack64(sb, d_proof, d_src64 + 4 * R, d_src64 + 5 * R, s->d_omit64, cc.n_in64, R, d_sup_in64, sup_r);
            launch_unpack64(sb, d_proof, d_src64 + 2 * R, d_src64 + 3 * R, s->d_omit64, cc.n_corr64, R, d_sup_corr64, sup_r);
            launch_unpack64(sb, d_proof, d_src64 + 0 * R, d_src64 + 1 * R, s->d_omit64, cc.n_rec64, R, d_sup_rec64, sup_r);
        }
        p64.omit = s->d_omit64;
        p64.sup_in = d_sup_in64;
        p64.sup_corr = d_sup_corr64;
        p64.sup_rec = d_sup_rec64;
        p64.sup_r = sup_r;
    }
    InterpParams p{};
    p.on_mask = d_onm;
    p.sup_in = d_sup_in;
    p.sup_corr = d_sup_corr;
    p.sup_rec = d_sup_rec;
    p.sup_nq = sup_nq;
    // whole proofs of eligible circuits (the conditions of the prover's MODE_PROVE_V, and every opened repetition in the first
    // sixteen quad words -- the verifier's slot order puts them into the first ten): one u64 of corrections per row instead of
    // corr rows (internal.h: MODE_VERIFY_C; RV_VERIFY_VC=0: corr rows)
    const bool vc_on = !(getenv("RV_VERIFY_VC") && atoi(getenv("RV_VERIFY_VC")) == 0);  // (read at every call: tests switch it)
    int vmode = MODE_VERIFY;
    // (not for gate streams with multi-base levels -- the prover's lazy linear forms: their kernel variant runs at 4 - 5 wavefronts
    // per SIMD either way and measured 0.07 ms SLOWER with the compact corrections; one-base streams: -0.03 ... -0.08 ms)
    if (vc_on && c->vclr_ok && !c->persist_gen && NQ == 64 && sup_nq == 16 && !on_quads.empty() && !g_recorder) {
        uint64_t* d_vc = nullptr;
        if ((rc = dalloc(ctx, (size_t)cc.n_rows, &d_vc))) return fail(rc);
        track(d_vc);
        HC(hipMemsetAsync(d_vc + cc.zero_row, 0, 8, ctx->stream));
        p.vc = d_vc;
        vmode = MODE_VERIFY_C;
        g_verify_vc.fetch_add(1, std::memory_order_relaxed);
    }
    if ((rc = shard_run(s, vmode, p, p64))) return fail(rc);
    // preprocessing slots: the online commitment is the one carried by the proof (preprocess.rs:55-57)
    launch_overlay_rows(ctx->stream, s->d_dig + 1 * DW, (const uint32_t*)d_hco, s->d_omit, R, 8, 0);
    launch_overlay_rows(ctx->stream, s->d_dig + 3 * DW, (const uint32_t*)d_hco64, s->d_omit, R, 8, 0);
    if ((rc = shard_join(s))) return fail(rc);
    int dev_flags = 0;  // RV_DEV_ZERO_CHECK: an AssertZero of an opened repetition did not reconstruct to zero
    // the digests and the flag word leave through the mapped staging buffer (one small kernel instead of two copy-engine
    // operations of ~25 us each; see rv_prove_impl), unless it could not be had
    uint8_t* stage_dev = nullptr;
    if (small_stage && !g_recorder) {
        if (!ctx->h_stage && hipHostMalloc((void**)&ctx->h_stage, rv_ctx::STAGE_BYTES, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            ctx->h_stage = nullptr;
        }
        if (ctx->h_stage && hipHostGetDevicePointer((void**)&stage_dev, ctx->h_stage, 0) != hipSuccess) {
            (void)hipGetLastError();
            stage_dev = nullptr;
        }
    }
    if (stage_dev) {
        launch_store_words(ctx->stream, (const uint32_t*)s->d_h, R * 8, (uint32_t*)stage_dev, zero_checks_ok ? s->d_err : nullptr, (int*)(stage_dev + (size_t)R * 32));
        HC(hipStreamSynchronize(ctx->stream));
        memcpy(digests, ctx->h_stage, (size_t)R * 32);
        if (zero_checks_ok) memcpy(&dev_flags, ctx->h_stage + (size_t)R * 32, sizeof dev_flags);
    } else {
        HC(hipMemcpyAsync(digests, s->d_h, (size_t)R * 32, hipMemcpyDeviceToHost, ctx->stream));
        if (zero_checks_ok) HC(hipMemcpyAsync(&dev_flags, s->d_err, sizeof dev_flags, hipMemcpyDeviceToHost, ctx->stream));
        HC(hipStreamSynchronize(ctx->stream));
    }
    if (zero_checks_ok) *zero_checks_ok = !(dev_flags & RV_DEV_ZERO_CHECK);
    ctx->collect();
    ctx->prof.calls++;
#undef HC
    rv_shard_destroy(s);
    return RV_OK;
}

// Verification is STRICT unless the caller asks for the reference's behaviour (RV_VERIFY_REFERENCE_COMPAT): flags 0 and
// RV_VERIFY_STRICT mean the same thing; both bits together are a contradiction
static bool verify_flags_ok(uint32_t flags) {
    return !(flags & ~(uint32_t)(RV_VERIFY_STRICT | RV_VERIFY_REFERENCE_COMPAT)) &&
           (flags & (RV_VERIFY_STRICT | RV_VERIFY_REFERENCE_COMPAT)) != (RV_VERIFY_STRICT | RV_VERIFY_REFERENCE_COMPAT);
}
static bool verify_is_strict(uint32_t flags) { return !(flags & RV_VERIFY_REFERENCE_COMPAT); }

static int rv_verify_finish_impl(const uint8_t* proof, size_t proof_len, const uint8_t* slot_digests, uint32_t flags,
                                 int zero_checks_ok, int* ok) {
    if (!proof || !slot_digests || !ok || proof_len < 32 || !verify_flags_ok(flags)) return RV_E_ARG;
    uint8_t omit[RV_TOTAL_REPS];
    rv_challenge(proof, omit);  // proof/mod.rs:290
    b3::Hasher hs;
    size_t on = 0, pre = RV_ONLINE_REPS;
    for (int i = 0; i < RV_TOTAL_REPS; i++) hs.update(slot_digests + 32 * (omit[i] < 8 ? on++ : pre++), 32);
    uint8_t comm[32];
    hs.finalize(comm);
    *ok = memcmp(comm, proof, 32) == 0;
    if (verify_is_strict(flags)) {
        // the reference computes `okay` without reading it (online.rs:21,175-177) and only checks WHICH
        // repetitions are opened, never the records' omitted player (proof/mod.rs:292-302)
        if (!zero_checks_ok) *ok = 0;
        Parsed P;
        int rc = parse_proof(proof, proof_len, P);
        if (rc) return rc;
        if (!format_ok(P)) {
            *ok = 0;
            return RV_OK;
        }
        size_t k = 0;
        for (int i = 0; i < RV_TOTAL_REPS; i++)
            if (omit[i] < 8) {
                if (P.gf2.on[k].omit != omit[i] || P.z64.on[k].omit != omit[i]) *ok = 0;
                k++;
            }
    }
    return RV_OK;
}

extern "C" int rv_verify_finish_ex(const uint8_t* proof, size_t proof_len, const uint8_t* slot_digests, uint32_t flags,
                                   int zero_checks_ok, int* ok) {
    try {
        return rv_verify_finish_impl(proof, proof_len, slot_digests, flags, zero_checks_ok, ok);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

extern "C" int rv_verify_finish(const uint8_t* proof, size_t proof_len, const uint8_t* slot_digests, int* ok) {
    // no zero-check input here: this entry point is the reference's final check and nothing more (see the header)
    return rv_verify_finish_ex(proof, proof_len, slot_digests, RV_VERIFY_REFERENCE_COMPAT, 1, ok);
}

static int rv_verify_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t flags, int* ok);

extern "C" int rv_verify_ex(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t flags, int* ok) {
    try {  // no C++ exception may cross the C boundary
        return rv_verify_impl(ctx, c, proof, proof_len, flags, ok);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

// Proof::new / Proof::verify on the raw op list (proof/mod.rs:119-125,224-232): compile + prove / verify + release
// 128 bits of the op array's content, in parallel over host threads.  Per 16 bytes one 64 x 64 -> 128-bit multiply folded onto
// itself (the mixing step of wyhash), four independent lanes per piece, pieces combined in order.  Not a cryptographic hash and it
// need not be: a collision makes the PROVER use another circuit's gate stream, and that proof does not verify against the caller's.
static std::atomic<uint64_t> g_ops_cache_hits{0};
extern "C" uint64_t rv_hook_ops_cache_hits(void) { return g_ops_cache_hits.load(std::memory_order_relaxed); }
static inline uint64_t ops_mum(uint64_t a, uint64_t b) {
    const __uint128_t m = (__uint128_t)a * b;
    return (uint64_t)m ^ (uint64_t)(m >> 64);
}
static void ops_hash_piece(const uint8_t* p, size_t n, uint64_t seed, uint64_t out[2]) {
    constexpr uint64_t K0 = 0xa0761d6478bd642full, K1 = 0xe7037ed1a0b428dbull, K2 = 0x8ebc6af09c88c6e3ull, K3 = 0x589965cc75374cc3ull;
    uint64_t a = seed ^ K0, b = seed ^ K1, c = seed ^ K2, d = seed ^ K3;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        uint64_t w[8];
        memcpy(w, p + i, 64);
        a = ops_mum(w[0] ^ K1, w[1] ^ a);
        b = ops_mum(w[2] ^ K2, w[3] ^ b);
        c = ops_mum(w[4] ^ K3, w[5] ^ c);
        d = ops_mum(w[6] ^ K0, w[7] ^ d);
    }
    for (; i < n; i += 8) {
        uint64_t w = 0;
        memcpy(&w, p + i, std::min<size_t>(8, n - i));
        a = ops_mum(w ^ K1, a ^ K2 ^ (uint64_t)(n - i));
    }
    out[0] = ops_mum(a ^ K2, b ^ (uint64_t)n) ^ ops_mum(c ^ K0, d ^ K3);
    out[1] = ops_mum(a ^ c ^ K1, b ^ d ^ K0) ^ (uint64_t)n * K3;
}
static void ops_hash(const void* ptr, size_t bytes, uint64_t out[2]) {
    const uint8_t* p = (const uint8_t*)ptr;
    constexpr size_t PIECE = (size_t)4 << 20;
    const size_t n_pieces = std::max<size_t>((bytes + PIECE - 1) / PIECE, 1);
    std::vector<uint64_t> d(2 * n_pieces);
    static const unsigned n_thr = [] {
        if (const char* e = getenv("RV_OPS_HASH_THREADS")) return (unsigned)std::max(atoi(e), 1);
        return std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
    }();
    const unsigned T = (unsigned)std::min<size_t>(n_thr, n_pieces);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < n_pieces;) {
            const size_t lo = k * PIECE, hi = std::min(bytes, lo + PIECE);
            ops_hash_piece(p + lo, hi > lo ? hi - lo : 0, (uint64_t)k * 0x9e3779b97f4a7c15ull, &d[2 * k]);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    ops_hash_piece((const uint8_t*)d.data(), d.size() * 8, (uint64_t)bytes, out);
}

// the compiled circuit of an op list, from the context's cache or compiled now (and kept: at most RV_OPS_CACHE entries, default 2,
// the least recently used one leaves; RV_OPS_CACHE=0: nothing is kept, *owned = the caller destroys it)
static int ops_cache_get(rv_ctx* ctx, const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, rv_circuit** out, bool* hit,
                         bool* owned) {
    *hit = false;
    *owned = true;
    const size_t cap = getenv("RV_OPS_CACHE") ? (size_t)std::max(atoi(getenv("RV_OPS_CACHE")), 0) : 2;  // (read at every call: tests and bench switch it)
    if (!ctx || cap == 0 || !ops || n_ops < 1024) return rv_circuit_compile_ex(ctx, ops, n_ops, z64_wires, gf2_wires, flags, out);
    uint64_t h[2];
    ops_hash(ops, n_ops * sizeof(rv_op), h);
    {
        // (the knobs that change what the compilers and circuit_upload make of an op list are part of the key: tests switch them between calls)
        std::string knobs;
        for (const char* k : {"RV_LAZY_K", "RV_LAZY_SLACK", "RV_LAZY_BALANCE", "RV_NARROW", "RV_COMPILE_SEQ", "RV_LDS_RUN", "RV_LDS_QS", "RV_FLAT", "RV_REP", "RV_PERSIST"}) {
            const char* v = getenv(k);
            knobs += v ? v : "";
            knobs += ';';
        }
        uint64_t hk[2];
        ops_hash_piece((const uint8_t*)knobs.data(), knobs.size(), 0x6b6e6f6273ull, hk);
        h[0] ^= hk[0];
        h[1] += hk[1];
    }
    for (auto& e : ctx->ops_cache)
        if (e.h[0] == h[0] && e.h[1] == h[1] && e.n_ops == n_ops && e.z64_wires == z64_wires && e.gf2_wires == gf2_wires && e.flags == flags) {
            e.stamp = ++ctx->ops_clock;
            g_ops_cache_hits.fetch_add(1, std::memory_order_relaxed);
            *out = e.c;
            *hit = true;
            *owned = false;
            return RV_OK;
        }
    int rc = rv_circuit_compile_ex(ctx, ops, n_ops, z64_wires, gf2_wires, flags, out);
    if (rc) return rc;
    while (ctx->ops_cache.size() >= cap) {
        size_t v = 0;
        for (size_t i = 1; i < ctx->ops_cache.size(); i++)
            if (ctx->ops_cache[i].stamp < ctx->ops_cache[v].stamp) v = i;
        rv_circuit_destroy(ctx->ops_cache[v].c);
        ctx->ops_cache.erase(ctx->ops_cache.begin() + (long)v);
    }
    ctx->ops_cache.push_back({{h[0], h[1]}, n_ops, z64_wires, gf2_wires, flags, *out, ++ctx->ops_clock});
    *owned = false;
    return RV_OK;
}
extern "C" int rv_ctx_ops_cache_clear(rv_ctx* ctx) {
    if (!ctx) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto& e : ctx->ops_cache) rv_circuit_destroy(e.c);
    ctx->ops_cache.clear();
    return RV_OK;
}

extern "C" int rv_prove_ops(rv_ctx* ctx, const rv_op* ops, size_t n_ops, const uint8_t* wit_gf2, size_t n_gf2, const uint64_t* wit_z64, size_t n_z64,
                            size_t z64_wires, size_t gf2_wires, const uint8_t* seeds, uint8_t** proof, size_t* proof_len) {
    if (!proof || !proof_len) return RV_E_ARG;
    rv_circuit* c = nullptr;
    bool hit = false, owned = true;
    int rc;
    try {
        rc = ops_cache_get(ctx, ops, n_ops, z64_wires, gf2_wires, RV_COMPILE_WHOLE_PROVER, &c, &hit, &owned);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
    if (rc) return rc;
    try {  // (the first proof of a circuit, or of one nobody keeps: without the early-corrections staging; a circuit seen before takes it)
        rc = rv_prove_impl(ctx, c, wit_gf2, n_gf2, wit_z64, n_z64, seeds, proof, proof_len, nullptr, 0, /*allow_early=*/hit);
    } catch (...) {
        g_last_error = "out of host memory";
        rc = RV_E_NOMEM;
    }
    if (owned) rv_circuit_destroy(c);
    return rc;
}

extern "C" int rv_verify_ops(rv_ctx* ctx, const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, const uint8_t* proof, size_t proof_len,
                             uint32_t flags, int* ok) {
    if (!ok) return RV_E_ARG;
    rv_circuit* c = nullptr;
    bool hit = false, owned = true;
    int rc;
    try {
        rc = ops_cache_get(ctx, ops, n_ops, z64_wires, gf2_wires, 0, &c, &hit, &owned);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
    if (rc) return rc;
    rc = rv_verify_ex(ctx, c, proof, proof_len, flags, ok);
    if (owned) rv_circuit_destroy(c);
    return rc;
}

extern "C" int rv_verify(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, int* ok) {
    return rv_verify_ex(ctx, c, proof, proof_len, 0, ok);  // flags 0 = strict
}

static int rv_verify_impl(rv_ctx* ctx, const rv_circuit* c, const uint8_t* proof, size_t proof_len, uint32_t flags, int* ok) {
    if (!ctx || !c || !proof || !ok || !verify_flags_ok(flags)) return RV_E_ARG;
    *ok = 0;
    Parsed P;
    int rc = parse_proof(proof, proof_len, P);
    if (rc) return rc;
    if (!format_ok(P)) return RV_OK;  // wrong repetition counts: `false`, not an error (proof/mod.rs:225-230)
    std::vector<uint8_t> dig(RV_TOTAL_REPS * 32);
    int zc = 1;
    if ((rc = rv_verify_shard_ex(ctx, c, proof, proof_len, 0, RV_TOTAL_REPS, dig.data(), &zc))) return rc;
    return rv_verify_finish_ex(proof, proof_len, dig.data(), flags, zc, ok);
}

// ------------------------------------------------------------------------------------
// rv_verify_batch: many proofs of one circuit in one pass -- the verifier's counterpart of rv_prove_batch's fused path
// (pure GF(2) circuits below the large-circuit threshold; everything else verifies proof after proof).  Per proof the
// host only parses the bincode framing and fills its slot of ONE page-locked staging slab (seeds, omitted players,
// masks, opened keys, carried-over commitments, source offsets and the proof bytes themselves), which goes to the
// device in one copy; the per-proof kernel strings are recorded and replayed once per batch (launch.h), the levels run
// through the batched interpreter kernels in verify mode, and the slot digests plus the zero-check flags come back in
// one copy each.  The final check (rv_verify_finish_ex) is host work per proof.
// ------------------------------------------------------------------------------------
static int rv_verify_batch_impl(rv_ctx* ctx, const rv_circuit* c, size_t batch, const uint8_t* const* proofs, const size_t* proof_lens,
                                uint32_t flags, int* ok) {
    if (!ctx || !c || !batch || !proofs || !proof_lens || !ok || !verify_flags_ok(flags)) return RV_E_ARG;
    for (size_t b = 0; b < batch; b++) {
        ok[b] = 0;
        if (!proofs[b]) return RV_E_ARG;
    }
    const Compiled& cc = c->cc;
    // A proof that cannot be parsed is a rejected proof (ok[b] = 0), not a failed call: one bad proof from an untrusted
    // peer must not keep the others from being verified.  Non-zero return codes are left to argument / device errors.
    auto one_by_one = [&]() -> int {
        for (size_t b = 0; b < batch; b++) {
            const int rc = rv_verify_ex(ctx, c, proofs[b], proof_lens[b], flags, &ok[b]);
            if (rc == RV_E_PROOF_MALFORMED) {
                ok[b] = 0;
                continue;
            }
            if (rc) return rc;
        }
        return RV_OK;
    };
    static const size_t big_gates = [] {
        const char* e = getenv("RV_BATCH_BIG_GATES");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 20;
    }();
    if (batch == 1 || !cc.gates64.empty() || cc.gates.size() >= big_gates) return one_by_one();
    // ---- parse; proofs with the wrong repetition counts are `false` (proof/mod.rs:225-230) and take no further part
    std::vector<Parsed> P(batch);
    std::vector<size_t> live;  // indices of the proofs that go to the GPU
    size_t max_len = 0;
    // what the verifier groups require of the online records (the checks of fill() below, made before anything is
    // staged so that a malformed proof simply drops out of the batch)
    auto records_ok = [](const Parsed& Q) {
        for (uint32_t g0 = 0; g0 < RV_ONLINE_REPS; g0 += 8) {
            const OnRec* o = &Q.gf2.on[g0];
            const OnRec* z = &Q.z64.on[g0];
            for (int i = 0; i < 8; i++) {
                if (o[i].omit >= 8 || z[i].omit >= 8) return false;
                if (o[i].n_corr < o[0].n_corr || o[i].n_in < o[0].n_in || o[i].n_rec != o[0].n_rec) return false;
            }
        }
        return true;
    };
    for (size_t b = 0; b < batch; b++) {
        if (parse_proof(proofs[b], proof_lens[b], P[b]) != RV_OK) continue;  // ok[b] stays 0
        if (!format_ok(P[b]) || !records_ok(P[b])) continue;
        live.push_back(b);
        max_len = std::max(max_len, proof_lens[b]);
    }
    if (live.size() < 2) {
        for (size_t b : live) {
            const int rc = rv_verify_ex(ctx, c, proofs[b], proof_lens[b], flags, &ok[b]);
            if (rc == RV_E_PROOF_MALFORMED) {
                ok[b] = 0;
                continue;
            }
            if (rc) return rc;
        }
        return RV_OK;
    }
    HIPCHK(hipSetDevice(ctx->device));
    {  // a pass keeps one proof's working set resident per proof: larger batches run as consecutive chunks
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return RV_E_DEVICE;
        const size_t per_proof = std::max<size_t>(cc.info.scratch_bytes + 3 * (size_t)std::max<uint64_t>({cc.n_in, cc.n_pre, cc.n_rec, 1}) * 256, 1);
        size_t chunk = std::min<size_t>(std::max<size_t>((free_b + ctx->cached_bytes) / 2 / per_proof, 2), 4096);
        if (const char* e = getenv("RV_BATCH_MAX")) chunk = std::min<size_t>(chunk, (size_t)std::max(atoi(e), 2));
        if (batch > chunk) {
            for (size_t b0 = 0; b0 < batch; b0 += chunk) {
                const int rc = rv_verify_batch_impl(ctx, c, std::min(chunk, batch - b0), proofs + b0, proof_lens + b0, flags, ok + b0);
                if (rc) return rc;
            }
            return RV_OK;
        }
    }
    const size_t B = live.size();
    const uint32_t R = RV_TOTAL_REPS, NQ = R / 4;
    // ---- the staging slab: one slot per proof
    struct Slot {
        size_t seeds, omit, keep, onm, quads, hkeys, hco, hco64, src, proof, stride;
    } L{};
    {
        size_t o = 0;
        auto take = [&](size_t n) {
            const size_t at = o;
            o += (n + 255) & ~(size_t)255;
            return at;
        };
        L.seeds = take((size_t)R * 16);
        L.omit = take(R);
        L.keep = take((size_t)NQ * 4);
        L.onm = take((size_t)NQ * 4);
        L.quads = take((size_t)NQ * 4);
        L.hkeys = take((size_t)R * 128);
        L.hco = take((size_t)R * 32);
        L.hco64 = take((size_t)R * 32);
        L.src = take((size_t)6 * R * 8);
        L.proof = take(max_len);
        L.stride = o;
    }
    std::vector<rv_shard*> sh(B, nullptr);
    std::vector<void*> pinned_tmp, device_tmp;
    std::vector<LaunchRecorder> recs(B);
    for (auto& r : recs) r.batch = (unsigned)B;
    struct RecorderOff {
        ~RecorderOff() { g_recorder = nullptr; }
    } recorder_off;
    InterpParams* d_pp = nullptr;
    auto cleanup = [&](int code) {
        g_recorder = nullptr;
        (void)hipStreamSynchronize(ctx->stream);
        for (rv_shard* s : sh)
            if (s) {
                s->destroy();
                delete s;
            }
        ctx->release(d_pp);
        for (void* q : device_tmp) ctx->release(q);
        for (void* q : pinned_tmp) g_pinned.put(q);
        return code;
    };
    constexpr size_t HEAD = 256;  // (no slot pointer equals an arena block: the slabs are released exactly once)
    uint8_t* h_slab = (uint8_t*)g_pinned.get(std::max<size_t>(L.stride * B, PinnedPool::MIN_BYTES));
    if (!h_slab) return cleanup(RV_E_NOMEM);
    pinned_tmp.push_back(h_slab);
    uint8_t* d_slab = nullptr;
    int rc;
    if ((rc = dalloc(ctx, HEAD + L.stride * B, &d_slab))) return cleanup(rc);
    device_tmp.push_back(d_slab);
    uint8_t* d_out = nullptr;  // per proof: 256 x 32 digest bytes, then the device flag word
    const size_t out_stride = (size_t)R * 32 + 256;
    if ((rc = dalloc(ctx, HEAD + out_stride * B, &d_out))) return cleanup(rc);
    device_tmp.push_back(d_out);
    std::vector<uint32_t> n_quads(B, 0);
    auto fill = [&](size_t k) -> int {
        const size_t b = live[k];
        const Parsed& Q = P[b];
        uint8_t* h = h_slab + k * L.stride;
        memset(h, 0, L.proof);  // everything in front of the proof bytes
        uint8_t* omit = h + L.omit;
        memset(omit, 8, R);
        uint32_t* keep = (uint32_t*)(h + L.keep);
        uint32_t* onm = (uint32_t*)(h + L.onm);
        for (uint32_t q = 0; q < NQ; q++) keep[q] = 0xFFFFFFFFu;
        uint64_t* src = (uint64_t*)(h + L.src);
        // VerifierTranscriptOnline::new (online.rs:25-119) / VerifierTranscriptPreprocess::new (preprocess.rs:17-43),
        // as in rv_verify_shard: slots 0..39 are the online records in proof order, 40..255 the preprocessing ones
        for (uint32_t g0 = 0; g0 < R; g0 += 8) {
            if (g0 < RV_ONLINE_REPS) {
                const OnRec* o = &Q.gf2.on[g0];
                const OnRec* z = &Q.z64.on[g0];
                for (int i = 0; i < 8; i++) {
                    if (o[i].omit >= 8 || z[i].omit >= 8) return RV_E_PROOF_MALFORMED;
                    if (o[i].n_corr < o[0].n_corr || o[i].n_in < o[0].n_in || o[i].n_rec != o[0].n_rec) return RV_E_PROOF_MALFORMED;
                    const uint32_t r = g0 + i;
                    omit[r] = o[i].omit;
                    src[0 * R + r] = L.proof + o[i].rec;  // offsets into this proof's slot
                    src[1 * R + r] = o[0].n_rec;
                    src[2 * R + r] = L.proof + o[i].corr;
                    src[3 * R + r] = o[0].n_corr;
                    src[4 * R + r] = L.proof + o[i].in;
                    src[5 * R + r] = o[0].n_in;
                    keep[r / 4] &= ~(1u << (31 - 8 * (r % 4) - o[i].omit));
                    onm[r / 4] |= 0xFFu << (24 - 8 * (r % 4));
                    memcpy(h + L.hkeys + (size_t)r * 128, proofs[b] + o[i].keys, 128);
                }
            } else {
                const PreRec* q = &Q.gf2.pre[g0 - RV_ONLINE_REPS];
                const PreRec* q64 = &Q.z64.pre[g0 - RV_ONLINE_REPS];
                for (int i = 0; i < 8; i++) {
                    memcpy(h + L.seeds + (size_t)(g0 + i) * 16, proofs[b] + q[i].seed, 16);
                    memcpy(h + L.hco + (size_t)(g0 + i) * 32, proofs[b] + q[i].comm_online, 32);
                    memcpy(h + L.hco64 + (size_t)(g0 + i) * 32, proofs[b] + q64[i].comm_online, 32);
                }
            }
        }
        uint32_t* quads = (uint32_t*)(h + L.quads);
        for (uint32_t q = 0; q < NQ; q++)
            if (onm[q]) quads[n_quads[k]++] = q;
        memcpy(h + L.proof, proofs[b], proof_lens[b]);
        return RV_OK;
    };
    {
        // host work per proof (a few hundred KB of copies each): shared by a few threads for large batches
        const size_t n_thr = B >= 32 ? std::min<size_t>({(size_t)8, B / 8, (size_t)std::max(1u, std::thread::hardware_concurrency())}) : 1;
        std::vector<int> rcs(std::max<size_t>(n_thr, 1), RV_OK);
        auto range = [&](size_t t, size_t k0, size_t k1) {
            try {  // (runs on a worker thread: nothing may escape it)
                for (size_t k = k0; k < k1 && rcs[t] == RV_OK; k++) rcs[t] = fill(k);
            } catch (...) {
                rcs[t] = RV_E_NOMEM;
            }
        };
        if (n_thr <= 1) {
            range(0, 0, B);
        } else {
            std::vector<std::thread> th;
            th.reserve(n_thr);
            try {
                for (size_t t = 0; t < n_thr; t++) th.emplace_back(range, t, B * t / n_thr, B * (t + 1) / n_thr);
            } catch (...) {  // a thread could not be started: the ranges without one are done here
                for (size_t t = th.size(); t < n_thr; t++) range(t, B * t / n_thr, B * (t + 1) / n_thr);
            }
            for (auto& x : th) x.join();
        }
        for (int r : rcs)
            if (r) return cleanup(r);
    }
    if (hipMemcpyAsync(d_slab + HEAD, h_slab, L.stride * B, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return cleanup(RV_E_DEVICE);
    // ---- per proof (recorded): keys, masks, supplied-value rows, buffers
    std::vector<InterpParams> pp(B);
    for (size_t k = 0; k < B && !rc; k++) {
        uint8_t* d = d_slab + HEAD + k * L.stride;
        rv_shard* s = sh[k] = new rv_shard();
        s->ctx = ctx;
        s->c = c;
        s->rep_begin = 0;
        s->R = R;
        s->NQ = NQ;
        s->d_seeds = d + L.seeds;  // slab slots: not arena blocks, destroy() ignores them
        s->d_omit = d + L.omit;
        s->d_h = d_out + HEAD + k * out_stride;
        s->d_err = (int*)(d_out + HEAD + k * out_stride + (size_t)R * 32);
        s->d_on_quads = (const uint32_t*)(d + L.quads);
        s->n_on_quads = n_quads[k];
        if ((rc = dalloc(ctx, (size_t)R * 128, &s->d_keys))) break;
        uint32_t *d_sup_in = nullptr, *d_sup_corr = nullptr, *d_sup_rec = nullptr;
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_in, 1) * NQ, &d_sup_in))) break;
        s->extra.push_back(d_sup_in);
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_pre, 1) * NQ, &d_sup_corr))) break;
        s->extra.push_back(d_sup_corr);
        if ((rc = dalloc(ctx, (size_t)std::max<uint64_t>(cc.n_rec, 1) * NQ, &d_sup_rec))) break;
        s->extra.push_back(d_sup_rec);
        g_recorder = &recs[k];
        launch_expand_seeds(ctx->stream, s->d_seeds, R, s->d_keys);
        launch_overlay_rows(ctx->stream, (uint32_t*)s->d_keys, (const uint32_t*)(d + L.hkeys), s->d_omit, R, 32, 1);
        if (!(rc = shard_setup_prg(s, (const uint32_t*)(d + L.keep)))) {
            const uint64_t* d_src = (const uint64_t*)(d + L.src);
            launch_unpack_bits(ctx->stream, d, d_src + 4 * R, d_src + 5 * R, s->d_omit, cc.n_in, NQ, 1, d_sup_in, NQ);
            launch_unpack_bits(ctx->stream, d, d_src + 2 * R, d_src + 3 * R, s->d_omit, cc.n_pre, NQ, 1, d_sup_corr, NQ);
            launch_unpack_bits(ctx->stream, d, d_src + 0 * R, d_src + 1 * R, s->d_omit, cc.n_rec, NQ, 0, d_sup_rec, NQ);
            Interp64Params p64{};
            pp[k] = InterpParams{};
            pp[k].on_mask = (const uint32_t*)(d + L.onm);
            pp[k].sup_in = d_sup_in;
            pp[k].sup_corr = d_sup_corr;
            pp[k].sup_rec = d_sup_rec;
            pp[k].sup_nq = NQ;
            rc = shard_run_alloc(s, pp[k], p64);
        }
        g_recorder = nullptr;
    }
    if (rc) return cleanup(rc);
    if ((rc = replay_recorded(ctx, recs, pinned_tmp, device_tmp))) return cleanup(rc);
    // ---- all proofs level by level, verify mode
    if ((rc = dalloc(ctx, B, &d_pp))) return cleanup(rc);
    if (hipMemcpyAsync(d_pp, pp.data(), B * sizeof(InterpParams), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return cleanup(RV_E_DEVICE);
    {
        const size_t n_levels = cc.level_start.empty() ? 0 : cc.level_start.size() - 1;
        for (size_t l = 0; l < n_levels; l++) {
            if (lds_run_for_batch(c, l, B)) {
                const auto& pl = c->lds_runs[(size_t)c->lds_run_of_level[l]];
                if (l == pl.run.l0)
                    launch_interp_lds(ctx->stream, MODE_VERIFY, pl.qs, RV_TOTAL_REPS / 4, c->d_lds_recs + pl.run.rec0, pl.run.n_steps, pl.run.n_slots,
                                      pl.run.eo0, pl.run.ep0, InterpParams{}, d_pp, (uint32_t)B);
                continue;
            }
            if (c->run_of_level[l] >= 0) {
                const auto& run = c->narrow_runs[(size_t)c->run_of_level[l]];
                if (l == run.first)
                    launch_interp_narrow_batched(ctx->stream, c->d_gates, c->d_level_range, run.first, run.second, run.tiny, d_pp, (uint32_t)B, MODE_VERIFY);
                continue;
            }
            launch_interp_batched(ctx->stream, c->d_gates, cc.level_range[l], d_pp, (uint32_t)B, MODE_VERIFY);
        }
    }
    // ---- per proof (recorded): digests, the commitments the preprocessing slots carry over, join
    for (size_t k = 0; k < B && !rc; k++) {
        uint8_t* d = d_slab + HEAD + k * L.stride;
        rv_shard* s = sh[k];
        const size_t DW = (size_t)R * 8;
        g_recorder = &recs[k];
        if (!(rc = shard_run_hash(s))) {
            launch_overlay_rows(ctx->stream, s->d_dig + 1 * DW, (const uint32_t*)(d + L.hco), s->d_omit, R, 8, 0);
            launch_overlay_rows(ctx->stream, s->d_dig + 3 * DW, (const uint32_t*)(d + L.hco64), s->d_omit, R, 8, 0);
            rc = shard_join(s);
        }
        g_recorder = nullptr;
    }
    if (rc) return cleanup(rc);
    if ((rc = replay_recorded(ctx, recs, pinned_tmp, device_tmp))) return cleanup(rc);
    uint8_t* h_out = (uint8_t*)g_pinned.get(std::max<size_t>(out_stride * B, PinnedPool::MIN_BYTES));
    if (!h_out) return cleanup(RV_E_NOMEM);
    pinned_tmp.push_back(h_out);
    if (hipMemcpyAsync(h_out, d_out + HEAD, out_stride * B, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return cleanup(hip_fail(hipGetLastError(), "verify batch sync", __FILE__, __LINE__));
    // ---- the final check per proof (proof/mod.rs:283-306)
    for (size_t k = 0; k < B; k++) {
        const size_t b = live[k];
        int dev_flags = 0;
        memcpy(&dev_flags, h_out + k * out_stride + (size_t)R * 32, sizeof dev_flags);
        if ((rc = rv_verify_finish_ex(proofs[b], proof_lens[b], h_out + k * out_stride, flags, !(dev_flags & RV_DEV_ZERO_CHECK), &ok[b])))
            return cleanup(rc);
    }
    ctx->prof.calls += B;
    return cleanup(RV_OK);
}

extern "C" int rv_verify_batch(rv_ctx* ctx, const rv_circuit* c, size_t batch, const uint8_t* const* proofs, const size_t* proof_lens,
                               uint32_t flags, int* ok) {
    try {  // no C++ exception may cross the C boundary
        return rv_verify_batch_impl(ctx, c, batch, proofs, proof_lens, flags, ok);
    } catch (...) {
        g_last_error = "out of host memory";
        return RV_E_NOMEM;
    }
}

// ------------------------------------------------------------------------------------
// parity-test hooks
// ------------------------------------------------------------------------------------
extern "C" int rv_hook_expand_seed(rv_ctx* ctx, const uint8_t* seeds, size_t n, uint8_t* keys) {
    if (!ctx || !seeds || !keys || !n) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    uint8_t *ds = nullptr, *dk = nullptr;
    int rc;
    if ((rc = dalloc(ctx, n * 16, &ds)) || (rc = dalloc(ctx, n * 128, &dk))) return rc;
    HIPCHK(hipMemcpyAsync(ds, seeds, n * 16, hipMemcpyHostToDevice, ctx->stream));
    launch_expand_seeds(ctx->stream, ds, (uint32_t)n, dk);
    HIPCHK(hipMemcpyAsync(keys, dk, n * 128, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->release(ds);
    ctx->release(dk);
    return RV_OK;
}

extern "C" int rv_hook_prg_blocks(rv_ctx* ctx, const uint8_t* keys, size_t n_keys, uint64_t first_block, size_t n_blocks,
                                  uint8_t* out) {
    if (!ctx || !keys || !out || !n_keys || !n_blocks) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    uint8_t *dk = nullptr, *drk = nullptr, *dout = nullptr;
    int rc;
    if ((rc = dalloc(ctx, n_keys * 16, &dk)) || (rc = dalloc(ctx, n_keys * RK_BYTES, &drk)) || (rc = dalloc(ctx, n_keys * n_blocks * 16, &dout)))
        return rc;
    HIPCHK(hipMemcpyAsync(dk, keys, n_keys * 16, hipMemcpyHostToDevice, ctx->stream));
    launch_key_schedule(ctx->stream, dk, (uint32_t)n_keys, drk);
    launch_aes_blocks(ctx->stream, drk, (uint32_t)n_keys, first_block, n_blocks, dout);
    HIPCHK(hipMemcpyAsync(out, dout, n_keys * n_blocks * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->release(dk);
    ctx->release(drk);
    ctx->release(dout);
    return RV_OK;
}

extern "C" int rv_hook_sharegen_gf2(rv_ctx* ctx, const uint8_t* keys, const uint32_t omit[8], size_t n, uint64_t* out) {
    if (!ctx || !keys || !omit || !out || !n) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t R = 8, NQ = 2;
    uint8_t *dk = nullptr, *drk = nullptr;
    uint32_t *d_rk = nullptr, *d_keep = nullptr, *d_masks = nullptr;
    uint32_t keep[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t r = 0; r < 8; r++) {
        if (omit[r] > 8) return RV_E_ARG;
        if (omit[r] < 8) keep[r / 4] &= ~(1u << (31 - 8 * (r % 4) - omit[r]));
    }
    const uint64_t n_blocks = (n + 127) / 128;
    int rc;
    if ((rc = dalloc(ctx, (size_t)R * 128, &dk)) || (rc = dalloc(ctx, (size_t)R * 8 * RK_BYTES, &drk)) ||
        (rc = dalloc(ctx, (size_t)RK_AREAS * 128 * NQ, &d_rk)) || (rc = dalloc(ctx, NQ, &d_keep)) ||
        (rc = dalloc(ctx, (size_t)n_blocks * 128 * NQ, &d_masks)))
        return rc;
    HIPCHK(hipMemcpyAsync(dk, keys, (size_t)R * 128, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_keep, keep, sizeof keep, hipMemcpyHostToDevice, ctx->stream));
    launch_key_schedule(ctx->stream, dk, R * 8, drk);
    launch_bitslice_rk(ctx->stream, drk, NQ, d_rk);
    launch_aes_gf2_masks(ctx->stream, d_rk, d_keep, NQ, 0, n_blocks, d_masks);
    std::vector<uint32_t> tmp((size_t)n_blocks * 128 * NQ);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_masks, tmp.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (size_t m = 0; m < n; m++) out[m] = ((uint64_t)tmp[2 * m] << 32) | tmp[2 * m + 1];
    ctx->release(dk);
    ctx->release(drk);
    ctx->release(d_rk);
    ctx->release(d_keep);
    ctx->release(d_masks);
    return RV_OK;
}

extern "C" int rv_hook_sharegen_z64(rv_ctx* ctx, const uint8_t* keys, const uint32_t omit[8], size_t n, uint64_t* out) {
    if (!ctx || !keys || !omit || !out || !n) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t R = 8, NQ = 2;
    uint8_t *dk = nullptr, *drk = nullptr;
    uint32_t *d_rk = nullptr, *d_keep = nullptr;
    uint64_t* d_masks = nullptr;
    uint32_t keep[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t r = 0; r < 8; r++) {
        if (omit[r] > 8) return RV_E_ARG;
        if (omit[r] < 8) keep[r / 4] &= ~(1u << (31 - 8 * (r % 4) - omit[r]));
    }
    const uint64_t n_blocks = (n + 1) / 2;
    int rc;
    if ((rc = dalloc(ctx, (size_t)R * 128, &dk)) || (rc = dalloc(ctx, (size_t)R * 8 * RK_BYTES, &drk)) ||
        (rc = dalloc(ctx, (size_t)RK_AREAS * 128 * NQ, &d_rk)) || (rc = dalloc(ctx, NQ, &d_keep)) ||
        (rc = dalloc(ctx, (size_t)n_blocks * 2 * R * 8, &d_masks)))
        return rc;
    HIPCHK(hipMemcpyAsync(dk, keys, (size_t)R * 128, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_keep, keep, sizeof keep, hipMemcpyHostToDevice, ctx->stream));
    launch_key_schedule(ctx->stream, dk, R * 8, drk);
    launch_bitslice_rk(ctx->stream, drk, NQ, d_rk);
    launch_aes_z64_masks(ctx->stream, d_rk, d_keep, NQ, n_blocks, d_masks);
    HIPCHK(hipMemcpyAsync(out, d_masks, n * 64 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->release(dk);
    ctx->release(drk);
    ctx->release(d_rk);
    ctx->release(d_keep);
    ctx->release(d_masks);
    return RV_OK;
}

extern "C" int rv_hook_blake3(rv_ctx* ctx, const uint8_t* data, size_t n_streams, size_t len, uint8_t* out) {
    if (!ctx || !out || !n_streams || (len && !data)) return RV_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    // lay the streams out in transcript row format: row e, repetition r = stream r
    const uint32_t R = (uint32_t)((n_streams + 7) / 8 * 8), NQ = R / 4;
    std::vector<uint32_t> rows((size_t)std::max<size_t>(len, 1) * NQ, 0);
    for (size_t r = 0; r < n_streams; r++)
        for (size_t e = 0; e < len; e++) rows[e * NQ + r / 4] |= (uint32_t)data[r * len + e] << (24 - 8 * (r % 4));
    uint32_t *d_rows = nullptr, *cva = nullptr, *cvb = nullptr, *dig = nullptr;
    const size_t cvw = b3_stream_scratch_words(len, R);
    int rc;
    if ((rc = dalloc(ctx, rows.size(), &d_rows)) || (rc = dalloc(ctx, cvw, &cva)) || (rc = dalloc(ctx, cvw, &cvb)) ||
        (rc = dalloc(ctx, (size_t)R * 8, &dig)))
        return rc;
    HIPCHK(hipMemcpyAsync(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_b3_stream(ctx->stream, d_rows, len, NQ, cva, cvb, dig);
    std::vector<uint32_t> h((size_t)R * 8);
    HIPCHK(hipMemcpyAsync(h.data(), dig, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(out, h.data(), n_streams * 32);
    ctx->release(d_rows);
    ctx->release(cva);
    ctx->release(cvb);
    ctx->release(dig);
    return RV_OK;
}

// DomainGF2::reconstruct / DomainZ64::reconstruct (gf2/domain.rs:47-63, z64/domain.rs:53-61) through the device
// functions the interpreters use (recon32 on the two quad words of a packed share; the 4-lane shuffle sum)
extern "C" int rv_hook_gf2_reconstruct(rv_ctx* ctx, const uint64_t* shares, size_t n, uint64_t* out) {
    if (!ctx || (n && (!shares || !out))) return RV_E_ARG;
    if (!n) return RV_OK;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = dalloc(ctx, n, &d_in)) || (rc = dalloc(ctx, n, &d_out))) return rc;
    HIPCHK(hipMemcpyAsync(d_in, shares, n * 8, hipMemcpyHostToDevice, ctx->stream));
    launch_hook_recon_gf2(ctx->stream, d_in, n, d_out);
    HIPCHK(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->release(d_in);
    ctx->release(d_out);
    return RV_OK;
}

extern "C" int rv_hook_z64_reconstruct(rv_ctx* ctx, const uint64_t* shares, size_t n, uint64_t* out) {
    if (!ctx || (n && (!shares || !out))) return RV_E_ARG;
    if (!n) return RV_OK;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = dalloc(ctx, n * 64, &d_in)) || (rc = dalloc(ctx, n * 8, &d_out))) return rc;
    HIPCHK(hipMemcpyAsync(d_in, shares, n * 64 * 8, hipMemcpyHostToDevice, ctx->stream));
    launch_hook_recon_z64(ctx->stream, d_in, n, d_out);
    HIPCHK(hipMemcpyAsync(out, d_out, n * 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->release(d_in);
    ctx->release(d_out);
    return RV_OK;
}

#include "stream.inc"
#include "comm.inc"
