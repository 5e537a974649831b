// rccl_stub.cpp — TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl that lets W processes sharing ONE GPU run
// librsx's multi-rank framebuffer exchange (source_amd/csrc/rsx_comm.hpp) for real — the (me +- k) % W send / receive schedule, the
// [W][mine] scratch arithmetic of rsx_allreduce_frame, rsx_frame_segment offsets on the device, k_pack_bins — without a second GPU.
// librsx opens it through RSX_RCCL_LIB (rsx_comm.hpp: rccl::api); it exports the eleven nccl* symbols librsx binds.
//
// Transport: files in a directory every rank sees (RSX_STUB_DIR, default /dev/shm). A message is written under a temporary name and
// published by rename(); the receiver polls for the name, reads and unlinks it. Semantics kept from NCCL: calls between
// ncclGroupStart / ncclGroupEnd are deferred to the group's end; sends and receives between a pair of ranks match in issue order;
// everything is ordered after the work already enqueued on the stream (the group end synchronises the stream, moves the bytes with
// blocking copies and returns — later work on the stream sees them). Sends never block (so a group of sends and receives cannot
// deadlock whatever the order of the peers); a receive that waits longer than RSX_STUB_TIMEOUT_S (default 120) fails.
//
// RSX_STUB_ASYNC=1 (with RSX_STUB_DELAY_MS, default 20): the point-to-point groups complete ASYNCHRONOUSLY, like the real library's —
// ncclGroupEnd returns at once; a helper thread waits for the work enqueued on the stream before the group (an event), sleeps the
// delay, moves the bytes over its own non-blocking stream and only then releases the caller's stream, which has been parked behind
// a one-lane kernel spinning on a pinned flag. So the receive buffers change well after the call has returned, and only work that
// is ordered behind the group ON THE GROUP'S STREAM may read them: a consumer on another stream, a host read without a
// synchronisation, or a send buffer recycled too early shows up as a wrong frame (the synchronous form hides all three).
//
// build: hipcc -shared -fPIC -O2 tests/stub_rccl/rccl_stub.cpp -o librccl_stub.so -lpthread
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4 };

struct Comm {
    std::string dir, id;
    int n = 0, rank = 0;
    std::vector<uint64_t> sent, received;   // per peer: messages issued so far (their sequence numbers match across the pair)
    uint64_t collectives = 0;
};

struct Op {
    int kind;                               // 0 send, 1 recv
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

size_t type_bytes(int t) {
    switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}

double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
double timeout_s() { const char *e = getenv("RSX_STUB_TIMEOUT_S"); return e ? atof(e) : 120.0; }

std::string p2p_name(const Comm *c, int src, int dst, uint64_t seq) {
    char b[96];
    snprintf(b, sizeof b, "/rsxstub_%s_p_%d_%d_%llu", c->id.c_str(), src, dst, (unsigned long long)seq);
    return c->dir + b;
}
std::string coll_name(const Comm *c, uint64_t seq, int rank) {
    char b[96];
    snprintf(b, sizeof b, "/rsxstub_%s_c_%llu_%d", c->id.c_str(), (unsigned long long)seq, rank);
    return c->dir + b;
}

int publish(const std::string &name, const void *data, size_t bytes) {
    const std::string tmp = name + ".tmp";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return kSystemError;
    const char *p = static_cast<const char *>(data);
    size_t left = bytes;
    while (left) { const ssize_t w = write(fd, p, left); if (w <= 0) { close(fd); return kSystemError; } p += w; left -= (size_t)w; }
    close(fd);
    return rename(tmp.c_str(), name.c_str()) == 0 ? kSuccess : kSystemError;
}

int collect(const std::string &name, void *data, size_t bytes, bool remove) {
    const double t0 = now_s(), limit = timeout_s();
    int fd;
    while ((fd = open(name.c_str(), O_RDONLY)) < 0) {
        if (now_s() - t0 > limit) { fprintf(stderr, "rccl_stub: timed out waiting for %s\n", name.c_str()); return kSystemError; }
        usleep(200);
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size != bytes) {
        fprintf(stderr, "rccl_stub: %s holds %lld bytes, the receive asks for %zu (send / receive sizes of a pair must match)\n", name.c_str(), (long long)st.st_size, bytes);
        close(fd);
        return kInvalidArgument;
    }
    char *p = static_cast<char *>(data);
    size_t left = bytes;
    while (left) { const ssize_t r = read(fd, p, left); if (r <= 0) { close(fd); return kSystemError; } p += r; left -= (size_t)r; }
    close(fd);
    if (remove) unlink(name.c_str());
    return kSuccess;
}

// copies on `copier` (a non-blocking stream of the helper thread) when given, else blocking copies
int copy_down(void *host, const void *dev, size_t bytes, hipStream_t copier) {
    if (!bytes) return kSuccess;
    if (!copier) return hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess ? kSuccess : kUnhandledCudaError;
    if (hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, copier) != hipSuccess) return kUnhandledCudaError;
    return hipStreamSynchronize(copier) == hipSuccess ? kSuccess : kUnhandledCudaError;
}
int copy_up(void *dev, const void *host, size_t bytes, hipStream_t copier) {
    if (!bytes) return kSuccess;
    if (!copier) return hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess ? kSuccess : kUnhandledCudaError;
    if (hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, copier) != hipSuccess) return kUnhandledCudaError;
    return hipStreamSynchronize(copier) == hipSuccess ? kSuccess : kUnhandledCudaError;
}

int transfer(std::vector<Op> &ops, hipStream_t copier) {
    std::vector<char> host;
    for (const Op &op : ops) {                                   // all sends first: they never wait
        if (op.kind != 0) continue;
        host.resize(op.bytes);
        int rc = copy_down(host.data(), op.buf, op.bytes, copier);
        if (rc) return rc;
        rc = publish(p2p_name(op.comm, op.comm->rank, op.peer, op.comm->sent[(size_t)op.peer]++), host.data(), op.bytes);
        if (rc) return rc;
    }
    for (const Op &op : ops) {
        if (op.kind != 1) continue;
        host.resize(op.bytes);
        int rc = collect(p2p_name(op.comm, op.peer, op.comm->rank, op.comm->received[(size_t)op.peer]++), host.data(), op.bytes, true);
        if (rc) return rc;
        rc = copy_up(op.buf, host.data(), op.bytes, copier);
        if (rc) return rc;
    }
    return kSuccess;
}

// ---- asynchronous completion (RSX_STUB_ASYNC=1) ---------------------------------------------------------------------------------------
__global__ void k_park(volatile int *flag) {
    // holds the stream until the helper thread has moved the group's bytes (or ~60 s have passed: a dead helper must not hang the GPU)
    const long long t0 = wall_clock64();
    while (__atomic_load_n(const_cast<int *>(flag), __ATOMIC_RELAXED) == 0) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 6000000000LL) break;
    }
}

struct Job { std::vector<Op> ops; hipEvent_t before; int *flag; };

struct Helper {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    std::thread thread;
    std::atomic<int> failed{0};
    bool started = false, stop = false, busy = false;
    int device = 0;
    int *flags = nullptr;                                       // pinned, one int per group in flight (ring)
    size_t next_flag = 0;
    static constexpr size_t kFlags = 4096;

    void loop() {
        (void)hipSetDevice(device);
        hipStream_t copier = nullptr;
        if (hipStreamCreateWithFlags(&copier, hipStreamNonBlocking) != hipSuccess) { failed = kUnhandledCudaError; copier = nullptr; }
        const char *e = getenv("RSX_STUB_DELAY_MS");
        const int delay_ms = e ? atoi(e) : 20;
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) break;
                job = std::move(jobs.front());
                jobs.pop_front();
                busy = true;
            }
            int rc = hipEventSynchronize(job.before) == hipSuccess ? kSuccess : kUnhandledCudaError;      // the stream's earlier work
            usleep((useconds_t)delay_ms * 1000);
            if (!rc && copier) rc = transfer(job.ops, copier);
            if (rc) { failed = rc; fprintf(stderr, "rccl_stub (async): a group failed with %d\n", rc); }
            __atomic_store_n(job.flag, 1, __ATOMIC_RELEASE);    // releases the caller's stream
            (void)hipEventDestroy(job.before);
            { std::lock_guard<std::mutex> lock(m); busy = false; }
            cv.notify_all();
        }
        if (copier) (void)hipStreamDestroy(copier);
    }

    int submit(std::vector<Op> &ops) {
        if (failed) return failed;
        std::lock_guard<std::mutex> lock(m);
        if (!started) {
            if (hipGetDevice(&device) != hipSuccess) return kUnhandledCudaError;
            if (hipHostMalloc(reinterpret_cast<void **>(&flags), kFlags * sizeof(int), hipHostMallocDefault) != hipSuccess) return kUnhandledCudaError;
            memset(flags, 0, kFlags * sizeof(int));
            thread = std::thread([this] { loop(); });
            thread.detach();                                    // (lives as long as the process; the object itself is never destroyed)
            started = true;
        }
        Job job;
        job.flag = flags + (next_flag++ % kFlags);
        __atomic_store_n(job.flag, 0, __ATOMIC_RELEASE);
        const hipStream_t stream = ops.front().stream;
        for (const Op &op : ops) if (op.stream != stream) return kInvalidArgument;      // (librsx issues a group on one stream)
        if (hipEventCreateWithFlags(&job.before, hipEventDisableTiming) != hipSuccess) return kUnhandledCudaError;
        if (hipEventRecord(job.before, stream) != hipSuccess) return kUnhandledCudaError;
        hipLaunchKernelGGL(k_park, dim3(1), dim3(1), 0, stream, job.flag);
        if (hipGetLastError() != hipSuccess) return kUnhandledCudaError;
        job.ops = std::move(ops);
        jobs.push_back(std::move(job));
        cv.notify_all();
        return kSuccess;
    }

    void drain() {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return jobs.empty() && !busy; });
    }
};

Helper &g_helper = *new Helper();
bool async_mode() { static const bool on = [] { const char *e = getenv("RSX_STUB_ASYNC"); return e && *e && *e != '0'; }(); return on; }

int run(std::vector<Op> &ops) {
    if (ops.empty()) return kSuccess;
    if (async_mode()) return g_helper.submit(ops);
    // everything enqueued on the streams before the group comes first
    for (const Op &op : ops) if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandledCudaError;
    return transfer(ops, nullptr);
}

int submit(const Op &op) {
    if (!op.comm || op.peer < 0 || op.peer >= op.comm->n || op.peer == op.comm->rank) return kInvalidArgument;
    g_queue.push_back(op);
    if (g_depth > 0) return kSuccess;
    std::vector<Op> ops;
    ops.swap(g_queue);
    return run(ops);
}

template <typename T> void reduce(T *acc, const T *other, size_t count, int op) {
    for (size_t i = 0; i < count; ++i) {
        if (op == 0) acc[i] = acc[i] + other[i];                 // ncclSum
        else if (op == 2) acc[i] = other[i] > acc[i] ? other[i] : acc[i];     // ncclMax
        else if (op == 3) acc[i] = other[i] < acc[i] ? other[i] : acc[i];     // ncclMin
        else if (op == 1) acc[i] = acc[i] * other[i];            // ncclProd
    }
}

}  // namespace

extern "C" {

typedef Comm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *out) {
    if (!out) return kInvalidArgument;
    memset(out->internal, 0, sizeof out->internal);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(out->internal, sizeof out->internal, "%x%lx%lx", (unsigned)getpid(), (unsigned long)ts.tv_sec, (unsigned long)ts.tv_nsec);
    return kSuccess;
}

int ncclCommInitRank(ncclComm_t *comm, int n_ranks, ncclUniqueId id, int rank) {
    if (!comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return kInvalidArgument;
    Comm *c = new Comm();
    const char *dir = getenv("RSX_STUB_DIR");
    c->dir = dir && *dir ? dir : "/dev/shm";
    id.internal[sizeof id.internal - 1] = 0;
    c->id = id.internal;
    c->n = n_ranks; c->rank = rank;
    c->sent.assign((size_t)n_ranks, 0); c->received.assign((size_t)n_ranks, 0);
    *comm = c;
    return kSuccess;
}

int ncclCommDestroy(ncclComm_t c) {
    if (!c) return kSuccess;
    if (async_mode() && g_helper.started) g_helper.drain();
    // (this rank's files of the last two collective rounds stay: a slower rank may not have read them yet — a few bytes each, in the
    // test's own directory)
    delete c;
    return kSuccess;
}

const char *ncclGetErrorString(int r) {
    switch (r) {
        case kSuccess: return "no error";
        case kUnhandledCudaError: return "stub: HIP call failed";
        case kSystemError: return "stub: file transport failed or timed out";
        case kInvalidArgument: return "stub: invalid argument";
        default: return "stub: internal error";
    }
}

int ncclCommCount(const ncclComm_t c, int *n) { if (!c || !n) return kInvalidArgument; *n = c->n; return kSuccess; }

int ncclGroupStart() { ++g_depth; return kSuccess; }

int ncclGroupEnd() {
    if (g_depth <= 0) return kInvalidArgument;
    if (--g_depth > 0) return kSuccess;
    std::vector<Op> ops;
    ops.swap(g_queue);
    return run(ops);
}

int ncclSend(const void *buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!w) return kInvalidArgument;
    return submit(Op{0, const_cast<void *>(buf), count * w, peer, c, stream});
}

int ncclRecv(void *buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!w) return kInvalidArgument;
    return submit(Op{1, buf, count * w, peer, c, stream});
}

// (collectives run at once, also inside a group: librsx only groups point-to-point calls and RSX_GATHER=broadcast's in-place
// broadcasts, which do not depend on one another)
int ncclBroadcast(const void *send, void *recv, size_t count, int type, int root, ncclComm_t c, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!c || !w || root < 0 || root >= c->n) return kInvalidArgument;
    if (async_mode() && g_helper.started) g_helper.drain();     // (collectives stay synchronous: after the groups issued before them)
    if (hipStreamSynchronize(stream) != hipSuccess) return kUnhandledCudaError;
    std::vector<char> host(count * w);
    if (c->rank == root) {
        if (host.size() && hipMemcpy(host.data(), send, host.size(), hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledCudaError;
        for (int r = 0; r < c->n; ++r) {
            if (r == root) continue;
            const int rc = publish(p2p_name(c, root, r, c->sent[(size_t)r]++), host.data(), host.size());
            if (rc) return rc;
        }
        if (recv != send && host.size() && hipMemcpy(recv, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
        return kSuccess;
    }
    const int rc = collect(p2p_name(c, root, c->rank, c->received[(size_t)root]++), host.data(), host.size(), true);
    if (rc) return rc;
    if (host.size() && hipMemcpy(recv, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
    return kSuccess;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int type, int op, ncclComm_t c, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!c || !w || (type != 8 && type != 7 && type != 2 && type != 4)) return kInvalidArgument;
    if (async_mode() && g_helper.started) g_helper.drain();
    if (hipStreamSynchronize(stream) != hipSuccess) return kUnhandledCudaError;
    const uint64_t seq = c->collectives++;
    // a rank publishes round s only after it has read every rank's round s - 1, so once this rank begins round s everybody is done
    // with round s - 2: its own file of that round can go
    if (seq >= 2) unlink(coll_name(c, seq - 2, c->rank).c_str());
    std::vector<char> mine(count * w), other(count * w);
    if (mine.size() && hipMemcpy(mine.data(), send, mine.size(), hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledCudaError;
    int rc = publish(coll_name(c, seq, c->rank), mine.data(), mine.size());
    if (rc) return rc;
    std::vector<char> acc;
    for (int r = 0; r < c->n; ++r) {                              // rank order: the same result, bit for bit, on every rank
        const char *src = mine.data();
        if (r != c->rank) { rc = collect(coll_name(c, seq, r), other.data(), other.size(), false); if (rc) return rc; src = other.data(); }
        if (r == 0) { acc.assign(src, src + mine.size()); continue; }
        if (type == 8) reduce(reinterpret_cast<double *>(acc.data()), reinterpret_cast<const double *>(src), count, op);
        else if (type == 7) reduce(reinterpret_cast<float *>(acc.data()), reinterpret_cast<const float *>(src), count, op);
        else if (type == 2) reduce(reinterpret_cast<int32_t *>(acc.data()), reinterpret_cast<const int32_t *>(src), count, op);
        else reduce(reinterpret_cast<int64_t *>(acc.data()), reinterpret_cast<const int64_t *>(src), count, op);
    }
    if (acc.size() && hipMemcpy(recv, acc.data(), acc.size(), hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
    return kSuccess;
}

}  // extern "C"
