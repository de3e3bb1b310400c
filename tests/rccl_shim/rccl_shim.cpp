// TEST INFRASTRUCTURE (never shipped, never loaded by the product on its own): a stand-in for librccl with ranks that are host
// threads of ONE process sharing ONE GPU, so that the library's multi-rank code path -- rv_comm_create_all, rv_prove_sharded's
// ncclAllGather of digests and the grouped ncclSend / ncclRecv of the openings (reverie_amd/csrc/comm.inc; the reference's
// gather point is /root/reference/src/proof/mod.rs:160-172) -- runs with 2, 4 and 8 ranks on a 1-GPU box (real RCCL refuses
// two ranks on one device).  Selected with RV_RCCL_PATH=<this library> in a fresh process (tests/test_gpu_multirank.py).
//
// Semantics kept from NCCL: calls are collective over the communicator's ranks; data is read / written in stream order on the
// stream handed in; point-to-point operations between a pair of ranks match in issue order; operations between ncclGroupStart
// and ncclGroupEnd are issued together at ncclGroupEnd.  Simplification: the host thread blocks until its part is complete.
//   hipcc -shared -fPIC -O2 -std=c++17 tests/rccl_shim/rccl_shim.cpp -o tests/rccl_shim/_build/librccl_shim.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg {
    const void* ptr;
    size_t bytes;
    bool done = false;  // set by the receiver once its copy has completed
};
struct World {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    // barrier
    int arrived = 0;
    uint64_t phase = 0;
    // all-gather staging: every rank's send pointer
    std::vector<const void*> send;
    // mailboxes [src][dst]
    std::vector<std::vector<std::deque<std::shared_ptr<Msg>>>> box;
    int attached = 0;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t ph = phase;
        if (++arrived == n) {
            arrived = 0;
            phase++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return phase != ph; });
        }
    }
};
struct Comm {
    std::shared_ptr<World> w;
    int rank = 0;
};
struct Op {
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    Comm* c;
    hipStream_t st;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::mutex g_reg_mu;
std::map<std::string, std::weak_ptr<World>> g_worlds;  // unique id -> world (ncclCommInitRank)
uint64_t g_next_id = 1;

size_t type_size(int t) {
    switch (t) {
    case 0: case 1: return 1;            // ncclInt8 / ncclUint8
    case 2: case 3: case 7: return 4;    // int32 / uint32 / float32
    case 4: case 5: case 8: return 8;    // int64 / uint64 / float64
    case 6: case 9: return 2;            // float16 / bfloat16
    default: return 1;
    }
}

int flush(std::vector<Op>& ops) {
    // sends first: the data must be complete on the sender's stream, then it is published and the sender waits for the copy
    std::vector<std::shared_ptr<Msg>> mine;
    for (const Op& o : ops)
        if (o.send) {
            if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
            auto m = std::make_shared<Msg>();
            m->ptr = o.ptr, m->bytes = o.bytes;
            World& w = *o.c->w;
            {
                std::lock_guard<std::mutex> g(w.mu);
                w.box[(size_t)o.c->rank][(size_t)o.peer].push_back(m);
            }
            w.cv.notify_all();
            mine.push_back(m);
        }
    std::vector<std::pair<std::shared_ptr<Msg>, World*>> got;
    for (const Op& o : ops)
        if (!o.send) {
            World& w = *o.c->w;
            std::shared_ptr<Msg> m;
            {
                std::unique_lock<std::mutex> lk(w.mu);
                auto& q = w.box[(size_t)o.peer][(size_t)o.c->rank];
                w.cv.wait(lk, [&] { return !q.empty(); });
                m = q.front();
                q.pop_front();
            }
            if (m->bytes != o.bytes) return 2;  // (NCCL would hang or corrupt: a mismatch is the caller's bug)
            if (hipMemcpyAsync(o.ptr, m->ptr, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return 1;
            got.emplace_back(m, &w);
        }
    for (const Op& o : ops)
        if (!o.send && hipStreamSynchronize(o.st) != hipSuccess) return 1;
    for (auto& g : got) {
        {
            std::lock_guard<std::mutex> lk(g.second->mu);
            g.first->done = true;
        }
        g.second->cv.notify_all();
    }
    if (!mine.empty()) {
        World& w = *ops[0].c->w;
        for (auto& m : mine) {
            std::unique_lock<std::mutex> lk(w.mu);
            w.cv.wait(lk, [&] { return m->done; });
        }
    }
    return 0;
}

std::shared_ptr<World> make_world(int n) {
    auto w = std::make_shared<World>();
    w->n = n;
    w->send.assign((size_t)n, nullptr);
    w->box.assign((size_t)n, std::vector<std::deque<std::shared_ptr<Msg>>>((size_t)n));
    return w;
}

}  // namespace

extern "C" {

struct ncclUniqueId {
    char internal[128];
};

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return 4;
    memset(id->internal, 0, sizeof id->internal);
    std::lock_guard<std::mutex> g(g_reg_mu);
    const uint64_t v = g_next_id++;
    memcpy(id->internal, "rv-shim", 7);
    memcpy(id->internal + 8, &v, 8);
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 4;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        const std::string key(id.internal, sizeof id.internal);
        w = g_worlds[key].lock();
        if (!w) {
            w = make_world(nranks);
            g_worlds[key] = w;
        }
    }
    if (w->n != nranks) return 4;
    Comm* c = new Comm();
    c->w = w;
    c->rank = rank;
    *comm = c;
    w->barrier();  // (collective, as the real call)
    return 0;
}

int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
    (void)devlist;  // the point of the shim: several ranks on one device are fine
    if (!comms || ndev < 1) return 4;
    auto w = make_world(ndev);
    for (int i = 0; i < ndev; i++) {
        Comm* c = new Comm();
        c->w = w;
        c->rank = i;
        comms[i] = c;
    }
    return 0;
}

int ncclCommDestroy(void* comm) {
    delete (Comm*)comm;
    return 0;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c) return 4;
    World& w = *c->w;
    const size_t bytes = sendcount * type_size(datatype);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;  // this rank's contribution is complete
    {
        std::lock_guard<std::mutex> g(w.mu);
        w.send[(size_t)c->rank] = sendbuff;
    }
    w.barrier();
    for (int r = 0; r < w.n; r++)
        if (hipMemcpyAsync((uint8_t*)recvbuff + (size_t)r * bytes, w.send[(size_t)r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    w.barrier();  // nobody reuses its send buffer before every rank has read it
    return 0;
}

int ncclGroupStart() {
    g_depth++;
    return 0;
}

int ncclGroupEnd() {
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return ops.empty() ? 0 : flush(ops);
}

int ncclSend(const void* sendbuff, size_t count, int datatype, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->w->n || peer == c->rank) return 4;
    g_ops.push_back(Op{true, (void*)sendbuff, count * type_size(datatype), peer, c, stream});
    if (g_depth == 0) {
        std::vector<Op> ops;
        ops.swap(g_ops);
        return flush(ops);
    }
    return 0;
}

int ncclRecv(void* recvbuff, size_t count, int datatype, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->w->n || peer == c->rank) return 4;
    g_ops.push_back(Op{false, recvbuff, count * type_size(datatype), peer, c, stream});
    if (g_depth == 0) {
        std::vector<Op> ops;
        ops.swap(g_ops);
        return flush(ops);
    }
    return 0;
}

const char* ncclGetErrorString(int code) {
    switch (code) {
    case 0: return "no error";
    case 1: return "unhandled HIP error (rccl test shim)";
    case 2: return "send / recv sizes differ (rccl test shim)";
    case 4: return "invalid argument (rccl test shim)";
    case 5: return "invalid usage (rccl test shim)";
    default: return "error (rccl test shim)";
    }
}

}  // extern "C"
