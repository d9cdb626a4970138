/* rccl_stub.cpp — an IN-PROCESS stand-in for librccl.so. Test infrastructure: never shipped, and loaded by the product only when
 * GPUART_HIP_RCCL_LIBRARY names it. Two purposes:
 *
 *  1. HOLDING calls (tests/test_stall_path.py): the call RCCL_STUB_HOLD names (init_all | init_rank | destroy | group_end |
 *     all_gather) sleeps for RCCL_STUB_HOLD_MS milliseconds (default: one hour, "never returns" on the scale of a test) before it
 *     goes on, so that the bounded waits of libgpuart_hip.so's multi-GPU read-out can be exercised.
 *
 *  2. SERVING communicators of N ranks whose ranks all live in THIS process (tests/test_gather_inprocess.py): the N > 1 branches of
 *     gather.h / gpuart_hip.hip — receive offsets of the root, the peers' sends, the share exchange — run on a box with ONE GPU.
 *       * ncclCommInitAll(n) hands out n handles of one world; ncclCommInitRank(n, id, rank) called by n threads with one id joins them
 *         (it returns when all n have arrived, like RCCL's bootstrap).
 *       * ncclSend / ncclRecv are matched per (source rank, destination rank) in posting order and served as ONE device-to-device
 *         hipMemcpyAsync on the RECEIVER's stream, ordered after the sender's stream at the time of the send (event) — and the sender's
 *         stream is made to wait for the copy (its buffer is in use until then), which is what a stream-ordered ncclSend promises.
 *       * ncclAllGather: when all n ranks have posted their call, every rank's stream copies the n contributions.
 *       * Outside a group a call is its own group. ncclGroupEnd (or the lone call) returns when everything this thread posted has
 *         been matched and queued on the streams — a peer that never posts leaves the caller waiting, as RCCL would (that wait is the
 *         product's to bound).
 *     Nothing is checked against a real RCCL: the stand-in implements the documented stream semantics of the five calls, no more.
 *
 * Build: g++ -shared -fPIC -O1 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o librccl_stub.so rccl_stub.cpp \
 *            -L/opt/rocm/lib -lamdhip64 -lpthread          (tests/conftest.py: fixture `rccl_stub`) */
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
}

namespace {

enum { OK = 0, UNHANDLED_HIP = 1, INTERNAL = 3, INVALID_ARG = 4, INVALID_USAGE = 5 };

void hold(const char *what) {
    const char *h = getenv("RCCL_STUB_HOLD");
    if (!h || strcmp(h, what) != 0) return;
    const char *m = getenv("RCCL_STUB_HOLD_MS");
    long ms = m ? atol(m) : 3600000L;
    struct timespec ts = {ms / 1000, (ms % 1000) * 1000000L};
    while (nanosleep(&ts, &ts) != 0) {}
}

size_t type_size(int t) {  // ncclDataType_t of rccl.h
    switch (t) {
    case 0: case 1: case 10: case 11: return 1;  // int8 / uint8 / fp8
    case 6: case 9: return 2;                    // half / bfloat16
    case 2: case 3: case 7: return 4;            // int32 / uint32 / float
    case 4: case 5: case 8: return 8;            // int64 / uint64 / double
    default: return 0;
    }
}

struct World;
struct Comm {
    World *w;
    int rank;
    uint64_t collectives = 0;  ///< how many all-gathers this rank has posted: the k-th of every rank belong together
};

struct Op {
    enum Kind { SEND, RECV, ALLGATHER } kind;
    Comm *c;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    uint64_t seq;
    hipStream_t stream;
    int device;
    hipEvent_t ready = nullptr;  ///< the posting stream's position when the call was made
    bool done = false;
    int error = OK;
};

struct World {
    int n = 0, joined = 0, alive = 0;
    std::vector<std::unique_ptr<Comm>> comms;
    std::map<std::pair<int, int>, std::deque<Op *>> sends, recvs;  ///< (source, destination) -> posted, unmatched
    std::map<uint64_t, std::vector<Op *>> gathers;                 ///< collective number -> the ranks' calls so far
};

// one lock for everything: a test stand-in, a handful of calls per test
std::mutex g_m;
std::condition_variable g_cv;
std::map<std::string, World *> g_by_id;
uint64_t g_next_id = 1;
std::string g_error = "no error";
uint64_t g_served[3] = {0, 0, 0};  ///< point-to-point transfers, their bytes, all-gathers

thread_local int t_depth = 0;
thread_local std::vector<std::shared_ptr<Op>> t_posted;

struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return OK;
    g_error = std::string("rccl_stub: ") + what + ": " + hipGetErrorString(e);
    return UNHANDLED_HIP;
}

/// `dst_stream` (on dst_device) copies `bytes` from src to dst once `src_ready` has passed; `src_stream` then waits for that copy.
int copy_between(const void *src, hipEvent_t src_ready, hipStream_t src_stream, int src_device, void *dst, hipStream_t dst_stream,
                 int dst_device, size_t bytes) {
    int r;
    if ((r = hip(hipSetDevice(dst_device), "hipSetDevice"))) return r;
    if ((r = hip(hipStreamWaitEvent(dst_stream, src_ready, 0), "hipStreamWaitEvent"))) return r;
    if (bytes && (r = hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dst_stream), "hipMemcpyAsync"))) return r;
    if (src_stream == dst_stream) return OK;
    hipEvent_t copied;
    if ((r = hip(hipEventCreateWithFlags(&copied, hipEventDisableTiming), "hipEventCreate"))) return r;
    r = hip(hipEventRecord(copied, dst_stream), "hipEventRecord");
    if (!r && !(r = hip(hipSetDevice(src_device), "hipSetDevice"))) r = hip(hipStreamWaitEvent(src_stream, copied, 0), "hipStreamWaitEvent");
    (void)hipEventDestroy(copied);  // (released by the runtime when the recorded work has passed)
    return r;
}

/// Everything that can be served now is queued on the streams. Called with g_m held.
void serve(World *w) {
    DeviceGuard guard;
    for (auto &kv : w->sends) {
        std::deque<Op *> &ss = kv.second, &rr = w->recvs[kv.first];
        while (!ss.empty() && !rr.empty()) {
            Op *s = ss.front(), *r = rr.front();
            ss.pop_front(); rr.pop_front();
            int e = OK;
            if (s->bytes != r->bytes) { e = INVALID_ARG; g_error = "rccl_stub: a send and its receive disagree about the size"; }
            else e = copy_between(s->src, s->ready, s->stream, s->device, r->dst, r->stream, r->device, s->bytes);
            s->error = r->error = e;
            s->done = r->done = true;
            g_served[0]++; g_served[1] += s->bytes;
        }
    }
    for (auto it = w->gathers.begin(); it != w->gathers.end();) {
        std::vector<Op *> &ops = it->second;
        if ((int)ops.size() < w->n) { ++it; continue; }
        int e = OK;
        for (Op *a : ops) if (a->bytes != ops[0]->bytes) { e = INVALID_ARG; g_error = "rccl_stub: the ranks of an all-gather disagree about the size"; }
        for (Op *to : ops)
            for (Op *from : ops)
                if (!e) e = copy_between(from->src, from->ready, from->stream, from->device, (char *)to->dst + (size_t)from->c->rank * from->bytes,
                                         to->stream, to->device, from->bytes);
        for (Op *a : ops) { a->error = e; a->done = true; }
        g_served[2]++;
        it = w->gathers.erase(it);
    }
}

/// The thread's posted calls enter their worlds; returns when all of them have been served.
int flush_thread() {
    std::vector<std::shared_ptr<Op>> mine;
    mine.swap(t_posted);
    std::unique_lock<std::mutex> lk(g_m);
    std::vector<World *> worlds;
    for (auto &sp : mine) {
        Op *o = sp.get();
        World *w = o->c->w;
        if (o->kind == Op::SEND) w->sends[{o->c->rank, o->peer}].push_back(o);
        else if (o->kind == Op::RECV) w->recvs[{o->peer, o->c->rank}].push_back(o);
        else w->gathers[o->seq].push_back(o);
        bool seen = false;
        for (World *x : worlds) seen |= x == w;
        if (!seen) worlds.push_back(w);
    }
    for (World *w : worlds) serve(w);
    g_cv.notify_all();
    g_cv.wait(lk, [&] { for (auto &sp : mine) if (!sp->done) return false; return true; });
    int e = OK;
    for (auto &sp : mine) {
        if (sp->ready) (void)hipEventDestroy(sp->ready);
        if (sp->error && !e) e = sp->error;
    }
    return e;
}

int post(Op::Kind kind, Comm *c, const void *src, void *dst, size_t count, int type, int peer, hipStream_t stream) {
    const size_t ts = type_size(type);
    if (!c || !ts || (kind != Op::ALLGATHER && (peer < 0 || peer >= c->w->n || peer == c->rank))) {
        g_error = "rccl_stub: bad argument (null communicator, unknown type, or a peer that is not another rank)";
        return INVALID_ARG;
    }
    auto o = std::make_shared<Op>();
    o->kind = kind; o->c = c; o->src = src; o->dst = dst; o->bytes = count * ts; o->peer = peer; o->stream = stream;
    if (hip(hipGetDevice(&o->device), "hipGetDevice")) return UNHANDLED_HIP;
    if (kind == Op::ALLGATHER) o->seq = c->collectives++;
    if (kind != Op::RECV) {
        if (hip(hipEventCreateWithFlags(&o->ready, hipEventDisableTiming), "hipEventCreate")) return UNHANDLED_HIP;
        if (hip(hipEventRecord(o->ready, stream), "hipEventRecord")) return UNHANDLED_HIP;
    }
    t_posted.push_back(o);
    return t_depth ? OK : flush_thread();
}

World *new_world(int n) {
    World *w = new World;
    w->n = n; w->alive = n;
    for (int k = 0; k < n; k++) { w->comms.emplace_back(new Comm); w->comms.back()->w = w; w->comms.back()->rank = k; }
    return w;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::lock_guard<std::mutex> lk(g_m);
    memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id++;
    memcpy(id->internal, &v, sizeof v);
    memcpy(id->internal + 8, "rccl_stub", 9);
    return OK;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return INVALID_ARG;
    hold("init_rank");
    const std::string key(id.internal, sizeof id.internal);
    std::unique_lock<std::mutex> lk(g_m);
    World *&slot = g_by_id[key];
    if (!slot) slot = new_world(nranks);
    World *w = slot;
    if (w->n != nranks) { g_error = "rccl_stub: the ranks of one id disagree about nranks"; return INVALID_ARG; }
    w->joined++;
    g_cv.notify_all();
    g_cv.wait(lk, [&] { return w->joined >= w->n; });  // RCCL's bootstrap: everybody, or nobody returns
    *comm = w->comms[(size_t)rank].get();
    return OK;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs) {
    (void)devs;  // the stand-in takes the device of a call from the calling thread's current device, as the streams do
    if (!comms || n < 1) return INVALID_ARG;
    hold("init_all");
    std::lock_guard<std::mutex> lk(g_m);
    World *w = new_world(n);
    w->joined = n;
    for (int k = 0; k < n; k++) comms[k] = w->comms[(size_t)k].get();
    return OK;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    hold("destroy");
    if (!comm) return INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_m);
    World *w = ((Comm *)comm)->w;
    if (--w->alive == 0) {
        for (auto it = g_by_id.begin(); it != g_by_id.end();) it = it->second == w ? g_by_id.erase(it) : std::next(it);
        delete w;
    }
    return OK;
}

ncclResult_t ncclGroupStart(void) { t_depth++; return OK; }
ncclResult_t ncclGroupEnd(void) {
    if (t_depth <= 0) return INVALID_USAGE;
    if (--t_depth) return OK;
    hold("group_end");
    return flush_thread();
}

ncclResult_t ncclSend(const void *buf, size_t count, int type, int peer, ncclComm_t comm, void *stream) {
    return post(Op::SEND, (Comm *)comm, buf, nullptr, count, type, peer, (hipStream_t)stream);
}
ncclResult_t ncclRecv(void *buf, size_t count, int type, int peer, ncclComm_t comm, void *stream) {
    return post(Op::RECV, (Comm *)comm, nullptr, buf, count, type, peer, (hipStream_t)stream);
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int type, ncclComm_t comm, void *stream) {
    hold("all_gather");
    return post(Op::ALLGATHER, (Comm *)comm, send, recv, count, type, -1, (hipStream_t)stream);
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { if (!c || !n) return INVALID_ARG; *n = ((Comm *)c)->w->n; return OK; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { if (!c || !r) return INVALID_ARG; *r = ((Comm *)c)->rank; return OK; }
const char *ncclGetErrorString(ncclResult_t e) {
    static thread_local std::string s;
    if (e == OK) return "no error";
    std::lock_guard<std::mutex> lk(g_m);
    s = g_error + " (rccl_stub code " + std::to_string(e) + ")";
    return s.c_str();
}

/// For the tests: point-to-point transfers served, their bytes, all-gathers served — a test that believes it exercised the N > 1
/// branch of the product checks that here.
void rccl_stub_served(uint64_t out[3]) {
    std::lock_guard<std::mutex> lk(g_m);
    memcpy(out, g_served, sizeof g_served);
}
}  // extern "C"
