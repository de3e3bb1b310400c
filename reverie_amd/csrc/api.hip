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
#include <memory>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "b3.h"
#include "compile.h"
#include "internal.h"
#include "launch.h"
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
extern "C" uint32_t rv_abi_version(void) { return 8; }  // 3: verification strict by default (RV_VERIFY_REFERENCE_COMPAT), rv_bristol_parse takes n_expected,
                                                        //    streaming prover, rv_prove_multi, reconstruct hooks
                                                        // 4: rv_circuit_compile_ex (a pure addition)
                                                        // 5: rv_prove_ops / rv_verify_ops, rv_hook_compile_compare (pure additions)
                                                        // 6: rv_circuit_info grew by early_staging_bytes (callers must pass the larger struct)
                                                        // 7: rv_circuit_info is its ABI-5 self again (a struct without a size field must not grow: a caller built
                                                        //    against the older header would have had 8 bytes written past its buffer); the value has a getter of
                                                        //    its own, rv_circuit_early_staging_bytes
                                                        // 8: rv_stream_same_cuts (a pure addition); rv_stream_info.reserved (always 0) is now kept_mib

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
// a kept copy of an op array (opscache.inc): plain uninitialised memory -- a std::vector would zero 240 MB before the copy overwrites them
struct OpsCopy {
    uint8_t* p;
    size_t bytes;
    explicit OpsCopy(size_t n);
    ~OpsCopy();
    OpsCopy(const OpsCopy&) = delete;
    OpsCopy& operator=(const OpsCopy&) = delete;
};
struct rv_ctx {
    int device = 0;
    size_t lds_bytes = 0;  // hipDeviceAttributeMaxSharedMemoryPerBlock
    hipStream_t stream = nullptr;   // setup, AES masks, hashing, openings (VALU-heavy work)
    hipStream_t stream2 = nullptr;  // side stream: the early-corrections copies, the verifier's proof copy and unpack kernels
    bool has_prio = false;          // the context's streams carry a stream priority of their own (rv_prove_batch's worker contexts)
    int prio = 0, mask_prio = 0;    // ... the main stream's (and stream2's), the mask generator stream's
    hipStream_t stream_m = nullptr; // RV_OVERLAP: the lane-distributed mask generator, beside the interpreter's level launches (made on first use)
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
    std::vector<hipEvent_t> ring_ev;  // per slot: recorded behind the copies out of it (a worker waits for it before it fills the slot again)
    // ... and pass 2's small ring: a chunk's host-built tables (proof offsets, item -> row lists) go to the device as ONE copy out
    // of a page-locked slot; a slot is written again only after the copy out of it has finished (its event).  Kept by the context:
    // unmapping four slots at the end of every stream was 20 ms of a 110 ms streamed proof.
    static constexpr int OPEN_SLOTS = 4;
    uint8_t* h_open[OPEN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t h_open_cap[OPEN_SLOTS] = {0, 0, 0, 0};
    hipEvent_t ev_open[OPEN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    unsigned open_next = 0;
    uint8_t* open_slot(size_t bytes, int* slot) {
        const int k = (int)(open_next++ % OPEN_SLOTS);
        if (ev_open[k]) (void)hipEventSynchronize(ev_open[k]);
        if (h_open_cap[k] < bytes) {
            if (h_open[k]) (void)hipHostFree(h_open[k]);
            h_open[k] = nullptr;
            h_open_cap[k] = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            if (hipHostMalloc((void**)&h_open[k], want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            h_open_cap[k] = want;
        }
        if (!ev_open[k] && hipEventCreateWithFlags(&ev_open[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ev_open[k] = nullptr;
            return nullptr;
        }
        *slot = k;
        return h_open[k];
    }
    // early corrections (rv_prove_impl): page-locked staging for EVERY repetition's corrections vector, the mapped
    // mailbox the challenge arrives in ([0] = sequence number, from word 16 on the data), the helper threads
    uint8_t* h_ec = nullptr;
    size_t h_ec_cap = 0;
    uint8_t* d_ec = nullptr;  // the device side of the staging (outside the arena: the proof's other buffers keep the places they have without it)
    size_t d_ec_cap = 0;
    uint32_t* h_fs = nullptr;
    uint32_t fs_seq = 0;
    HelperPool* ec_pool = nullptr;
    double ec_wait_us[17] = {0};  // running averages of the early-corrections waits (per chunk stamp, [16] the challenge): mailbox_wait
    // rv_prove_ops / rv_verify_ops: the circuits compiled from raw op lists, kept by content (ops_cache_get): the reference's
    // Proof::new takes the op list at every call (proof/mod.rs:119-124), and a caller that proves one circuit again and again through
    // that signature should pay the 70 - 90 ms host compile once, not per proof
    struct OpsEntry {
        std::shared_ptr<OpsCopy> ops;  // the op array the circuit was compiled from: a hit is a comparison against it
        std::string knobs;
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
    int profiling = 0;  // rv_ctx_profile: 0 off, 1 every phase, 2 the interpreter's phase only
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
        if (profiling == 2 && p != RV_PH_INTERP) p = -1;  // (only the dominant phase is marked: two markers per proof instead of seven)
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
        static const bool trace = getenv("RV_PINNED_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) {
            trim();
            e = hipMalloc(out, bytes);
        }
        if (trace) fprintf(stderr, "[rv arena] hipMalloc %zu KiB: %.2f ms (ctx %p)\n", bytes >> 10,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (void*)this);
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
// main_prio_level / mask_prio_level: the priority class (0 = the device's highest) of the main stream and of the mask generator's stream;
// -1 = the default priority.  A context with classes of its own (rv_prove_batch's workers) makes its second stream on first use.
static int ctx_create_impl(int device_ordinal, rv_ctx** out, int main_prio_level /* -1: default priority */, int mask_prio_level = -1);
// (main stream and mask-generator stream of a caller's context in the DEFAULT priority class: the highest / lowest classes measured the same, round 6)
extern "C" int rv_ctx_create(int device_ordinal, rv_ctx** out) { return ctx_create_impl(device_ordinal, out, -1); }

static int ctx_create_impl(int device_ordinal, rv_ctx** out, int main_prio_level, int mask_prio_level) {
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
    // The runtime multiplexes the streams of one priority class over four hardware queues (a fifth stream shares the first one's);
    // streams of different classes never share one (batch.inc: the workers of rv_prove_batch get classes of their own).
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0, (void)hipGetLastError();
    hipError_t se;
    if (main_prio_level >= 0 && prio_lo > prio_hi) {
        const int levels = prio_lo - prio_hi + 1;  // (numerically lower = higher priority)
        c->has_prio = true;
        c->prio = prio_hi + main_prio_level % levels;
        c->mask_prio = mask_prio_level >= 0 ? prio_hi + mask_prio_level % levels : c->prio;
        se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, c->prio);
    } else {
        se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    // (a worker context of rv_prove_batch makes no second stream before something asks for one (ctx_stream2): an idle stream still
    // holds a share of a hardware queue of its class, and which queue the NEXT stream of that class gets depends on it)
    if (se == hipSuccess && !c->has_prio) se = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete c;
        return hip_fail(se, "hipStreamCreate", __FILE__, __LINE__);
    }
    *out = c;
    return RV_OK;
}

// the context's second stream (worker contexts: made on first use, in the main stream's priority class)
static int ctx_stream2(rv_ctx* c) {
    if (c->stream2) return RV_OK;
    const hipError_t se = c->has_prio ? hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, c->prio) : hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    return se == hipSuccess ? RV_OK : hip_fail(se, "hipStreamCreate", __FILE__, __LINE__);
}


static void pinned_pool_trim();  // idle page-locked output buffers (defined with the pool below)

extern "C" void rv_ctx_destroy(rv_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream_m) (void)hipStreamSynchronize(ctx->stream_m);
    for (auto& e : ctx->ops_cache) rv_circuit_destroy(e.c);
    ctx->ops_cache.clear();
    ctx->trim();
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    for (uint8_t* p : ctx->h_ring) (void)hipHostFree(p);
    for (hipEvent_t e : ctx->ring_ev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < rv_ctx::OPEN_SLOTS; k++) {
        if (ctx->ev_open[k]) (void)hipEventDestroy(ctx->ev_open[k]);
        if (ctx->h_open[k]) (void)hipHostFree(ctx->h_open[k]);
    }
    if (ctx->h_ec) (void)hipHostFree(ctx->h_ec);
    if (ctx->d_ec) (void)hipFree(ctx->d_ec);
    if (ctx->h_fs) (void)hipHostFree(ctx->h_fs);
    delete ctx->ec_pool;
    for (rv_ctx* w : ctx->workers) rv_ctx_destroy(w);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream_m) (void)hipStreamDestroy(ctx->stream_m);
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
    static bool trace() {
        static const bool on = getenv("RV_PINNED_TRACE") != nullptr;
        return on;
    }
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
        const auto t0 = std::chrono::steady_clock::now();
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;  // caller falls back to malloc
        }
        if (trace()) fprintf(stderr, "[rv pinned] hipHostMalloc %zu MiB: %.2f ms (%zu buffers)\n", cap >> 20,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), bufs.size() + 1);
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
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipHostFree(bufs[victim].p);
            if (trace()) fprintf(stderr, "[rv pinned] hipHostFree %zu MiB: %.2f ms\n", bufs[victim].cap >> 20,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
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
    ctx->profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
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
    int staged_slot = -1;             // ... in this slot of rv_ctx::h_ring
    bool upload_pending = false;      // circuit_upload(async_staged) left the copies in flight on the context's stream
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
    bool persist_gen = false;  // some level has enough multi-base Mul / Xor gates for the kernel variant with their loops (kernels.hip: level_is_general)
};

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
static int circuit_upload(rv_ctx* ctx, rv_circuit* c, bool async_staged = false);

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
static void circuit_stage(rv_circuit* c, uint8_t* dst, int slot = -1) {
    size_t off = 0;
    circuit_stage_each(c->cc, [&](const void* p, size_t b) {
        if (b) memcpy(dst + off, p, b);
        off += (b + 255) & ~(size_t)255;
    });
    c->staged = dst;
    c->staged_bytes = off;
    c->staged_slot = slot;
}

// the compiled gate stream (c->cc) to HBM + the narrow-run plan; c is destroyed on failure
// async_staged (the streaming feeds): a piece whose arrays ALL came out of its page-locked ring slot is not waited for -- the
// slot's event is recorded behind the copies instead (the worker that fills the slot next waits for it), and the caller's kernels
// follow on the same stream
static int circuit_upload(rv_ctx* ctx, rv_circuit* c, bool async_staged) {
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
    constexpr bool stage_on = true;
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
    bool all_staged = c->staged != nullptr;
    auto up = [&](const void* src, size_t bytes, void** dst) -> int {
        int r = ctx->alloc(bytes, dst);
        if (r) return r;
        if (!bytes) return RV_OK;
        const bool from_slot = c->staged && stage_off + bytes <= c->staged_bytes;
        all_staged = all_staged && from_slot;
        if (from_slot) {  // (a streaming piece: copied here by a worker thread)
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
    if (async_staged && all_staged && c->staged_slot >= 0 && (size_t)c->staged_slot < ctx->ring_ev.size() && ctx->ring_ev[c->staged_slot]) {
        UPCHK(hipEventRecord(ctx->ring_ev[c->staged_slot], ctx->stream));
        c->upload_pending = true;
    } else {
        UPCHK(hipStreamSynchronize(ctx->stream));
    }
    c->staged = nullptr;  // (the slot belongs to the next piece from here on)
    c->staged_bytes = 0;
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
    c->ctx->release(c->d_lds_recs);
    // the early-corrections staging of a LARGE plan (2 GB of page-locked memory for the 10^6-MUL Z64 circuit) does not outlive the
    // circuit that needed it: a context that moves on to small circuits would otherwise hold it until rv_ctx_destroy.
    // Small stagings (the 160 MB of the 10^7-gate GF(2) circuit) stay: re-mapping them costs more than they weigh.
    if (c->ec_plan.ok && c->ec_plan.bytes >= ((size_t)1 << 30)) {
        auto drop = [](rv_ctx* x) {
            if (x->h_ec_cap < ((size_t)1 << 30)) return;
            (void)hipSetDevice(x->device);
            (void)hipStreamSynchronize(x->stream);
            if (x->stream2) (void)hipStreamSynchronize(x->stream2);
            if (x->h_ec) (void)hipHostFree(x->h_ec);
            if (x->d_ec) (void)hipFree(x->d_ec);
            x->h_ec = x->d_ec = nullptr;
            x->h_ec_cap = x->d_ec_cap = 0;
        };
        drop(c->ctx);
        for (rv_ctx* w : c->ctx->workers) drop(w);
    }
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

#include "shard.inc"
#include "open.inc"
#include "prove.inc"
#include "batch.inc"
#include "verify.inc"
#include "opscache.inc"
#include "verify_batch.inc"
#include "hooks.inc"
#include "stream.inc"
#include "comm.inc"
