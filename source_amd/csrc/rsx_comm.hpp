// rsx_comm.hpp — part of librsx's device translation unit (included by rsx_device.hip after the host API).
// Multi-GPU exchange of the spectral framebuffer over RCCL (xGMI), one process per GPU.
//
// The reference's counterpart is the result queue of MulticoreEngine (raysect/core/workflow.py:201-251): workers push
// (task, [(mean, variance) per pipeline], ray_count) messages and the parent folds them into the frame with
// StatsArray3D.combine_samples (pipeline/spectral/power.pyx:424-437). Here every rank renders into its own device-resident frame with
// no traffic during traversal, and the frames meet once:
//   * tile sharding   — ranks own disjoint runs of the x-major frame: rsx_allgather_frame moves the runs, no arithmetic;
//   * sample sharding — every rank holds a full frame of its own samples: rsx_allreduce_frame folds them with the combine_samples
//     law in rank order (deterministic, same result on every rank), routed as reduce-scatter + all-gather because xGMI links are
//     point to point: each rank receives 2 (W-1)/W frames instead of W-1.
// librccl is opened with dlopen at the first rsx_comm_* call: single-GPU users never load it, and a process that already holds
// an RCCL (PyTorch bundles one) shares that copy instead of getting a second set of global symbols.
#pragma once

#include <dlfcn.h>

namespace rccl {

// the handful of RCCL types the calls below need (rccl/rccl.h, ABI-stable across RCCL 2.x)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt32 = 2, ncclUint8 = 1, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2 };

struct Api {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    std::string error;
};

static Api &api() {
    static Api a;
    if (a.handle || !a.error.empty()) return a;
    const char *chosen = std::getenv("RSX_RCCL_LIB");     // set: that library or none (no silent fallback to another copy)
    const bool only = chosen && *chosen;
    const char *names[] = {chosen, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        if (only && n != chosen) break;
        a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.handle) break;
    }
    if (!a.handle) {
        const char *why = dlerror();                        // (read once: the call clears the message)
        a.error = std::string("librccl could not be loaded: ") + (why ? why : "not found");
        return a;
    }
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(a.handle, name);
        if (!p && a.error.empty()) a.error = std::string("librccl lacks ") + name;
        return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    if (!a.error.empty()) { dlclose(a.handle); a.handle = nullptr; }
    return a;
}

}  // namespace rccl

struct rsx_comm {
    rsx_ctx *ctx;
    rccl::ncclComm_t comm;
    int32_t n_ranks, rank;
    void *scratch;             // sample sharding: the ranks' copies of this rank's frame segment
    size_t scratch_bytes;
    double *word;              // one device double for barrier / max
};

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        const int r_ = (expr);                                                                                 \
        if (r_ != rccl::ncclSuccess) return rsx_fail(RSX_EHIP, "%s: %s", #expr, rccl::api().GetErrorString ? rccl::api().GetErrorString(r_) : "RCCL error"); \
    } while (0)

// Inside ncclGroupStart .. ncclGroupEnd an early return would leave the communicator inside an open group (every later call on this
// thread would be queued behind it or hang): the calls of a group record their first failure and the group is always closed.
struct GroupGuard {
    rccl::Api &a;
    int first = rccl::ncclSuccess;
    const char *what = nullptr;
    bool open = false;
    explicit GroupGuard(rccl::Api &api_) : a(api_) {}
    void start() { const int r = a.GroupStart(); if (r == rccl::ncclSuccess) open = true; else note(r, "ncclGroupStart"); }
    void note(int r, const char *w) { if (r != rccl::ncclSuccess && first == rccl::ncclSuccess) { first = r; what = w; } }
    int finish() {
        if (open) { open = false; note(a.GroupEnd(), "ncclGroupEnd"); }
        if (first != rccl::ncclSuccess) return rsx_fail(RSX_EHIP, "%s: %s", what, a.GetErrorString ? a.GetErrorString(first) : "RCCL error");
        return RSX_OK;
    }
    ~GroupGuard() { if (open) (void)a.GroupEnd(); }
};
#define RCCL_IN_GROUP(g, expr) (g).note((expr), #expr)

// Do all ranks agree that a step succeeded? (max over the ranks of a flag; an allocation that failed on one rank must stop every rank
// before the exchange, or the others would wait in their sends and receives for ever)
static int ranks_agree(rsx_comm *c, bool ok_here, bool *ok_everywhere);

extern "C" int rsx_comm_unique_id(void *id128) {
    if (!id128) return rsx_fail(RSX_EINVAL, "rsx_comm_unique_id: null buffer");
    rccl::Api &a = rccl::api();
    if (!a.handle) return rsx_fail(RSX_EUNSUPPORTED, "%s", a.error.c_str());
    rccl::ncclUniqueId id;
    RCCL_TRY(a.GetUniqueId(&id));
    std::memcpy(id128, id.internal, sizeof(id.internal));
    return RSX_OK;
}

extern "C" int rsx_comm_create(rsx_ctx *ctx, int32_t n_ranks, int32_t rank, const void *id128, rsx_comm **out) {
    if (!ctx || !id128 || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return rsx_fail(RSX_EINVAL, "rsx_comm_create: bad arguments");
    rccl::Api &a = rccl::api();
    if (!a.handle) return rsx_fail(RSX_EUNSUPPORTED, "%s", a.error.c_str());
    HIP_TRY(hipSetDevice(ctx->device));
    rccl::ncclUniqueId id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    rsx_comm *c = new (std::nothrow) rsx_comm();
    if (!c) return rsx_fail(RSX_ENOMEM, "out of host memory");
    c->ctx = ctx; c->comm = nullptr; c->n_ranks = n_ranks; c->rank = rank; c->scratch = nullptr; c->scratch_bytes = 0; c->word = nullptr;
    const int r = a.CommInitRank(&c->comm, n_ranks, id, rank);
    if (r != rccl::ncclSuccess) { delete c; return rsx_fail(RSX_EHIP, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, a.GetErrorString(r)); }
    if (hipMalloc(&c->word, 64) != hipSuccess) { a.CommDestroy(c->comm); delete c; return rsx_fail(RSX_ENOMEM, "rsx_comm_create: out of device memory"); }
    *out = c;
    return RSX_OK;
}

extern "C" void rsx_comm_free(rsx_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) rccl::api().CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->word) (void)hipFree(c->word);
    delete c;
}

// max over the ranks of one host double (bench timing: the slowest rank defines the step); doubles as a barrier
extern "C" int rsx_comm_max_f64(rsx_comm *c, double *value) {
    if (!c || !value) return rsx_fail(RSX_EINVAL, "rsx_comm_max_f64: null argument");
    rsx_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(c->word, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    RCCL_TRY(rccl::api().AllReduce(c->word, c->word, 1, rccl::ncclFloat64, rccl::ncclMax, c->comm, ctx->stream));
    HIP_TRY(hipMemcpyAsync(value, c->word, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSX_OK;
}

extern "C" int rsx_comm_barrier(rsx_comm *c) {
    double zero = 0.0;
    return rsx_comm_max_f64(c, &zero);
}

// Tile sharding: rank r owns frame elements [shard_begin[r], shard_begin[r+1]) of the x-major (mean, variance, samples) arrays
// (column tiles of a frame are contiguous runs). After the call every rank holds the whole frame. In place, no arithmetic: the
// result is bit-identical to a single-GPU render (Philox counters are per pixel).
// Every rank holds run [begin[r], begin[r + 1]) of rank r afterwards, in place. xGMI is point to point — every GPU has its own link
// to every other — so the exchange is one group of direct sends and receives: a rank sends its run to each peer over that peer's
// link and receives the peers' runs straight into their place, all links busy at once and every byte crossing one link once
// (a ring-routed ncclBroadcast per run would walk the runs around the ring one after the other). RSX_GATHER=broadcast keeps the
// grouped in-place broadcasts of round 2's first version.
static int gather_runs(rsx_comm *c, double *mean, double *variance, int32_t *samples, const int64_t *begin) {
    rsx_ctx *ctx = c->ctx;
    rccl::Api &a = rccl::api();
    static const bool by_broadcast = [] { const char *e = std::getenv("RSX_GATHER"); return e && std::strcmp(e, "broadcast") == 0; }();
    const int W = c->n_ranks, me = c->rank;
    GroupGuard g(a);
    g.start();
    if (by_broadcast) {
        for (int r = 0; r < W; ++r) {
            const size_t off = (size_t)begin[r], len = (size_t)(begin[r + 1] - begin[r]);
            if (!len) continue;
            RCCL_IN_GROUP(g, a.Broadcast(mean + off, mean + off, len, rccl::ncclFloat64, r, c->comm, ctx->stream));
            RCCL_IN_GROUP(g, a.Broadcast(variance + off, variance + off, len, rccl::ncclFloat64, r, c->comm, ctx->stream));
            RCCL_IN_GROUP(g, a.Broadcast(samples + off, samples + off, len, rccl::ncclInt32, r, c->comm, ctx->stream));
        }
    } else {
        const size_t my_off = (size_t)begin[me], my_len = (size_t)(begin[me + 1] - begin[me]);
        for (int k = 1; k < W; ++k) {
            const int to = (me + k) % W, from = (me - k + W) % W;         // (every pair meets once per direction; staggered so that no peer is everyone's first)
            if (my_len) {
                RCCL_IN_GROUP(g, a.Send(mean + my_off, my_len, rccl::ncclFloat64, to, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Send(variance + my_off, my_len, rccl::ncclFloat64, to, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Send(samples + my_off, my_len, rccl::ncclInt32, to, c->comm, ctx->stream));
            }
            const size_t off = (size_t)begin[from], len = (size_t)(begin[from + 1] - begin[from]);
            if (len) {
                RCCL_IN_GROUP(g, a.Recv(mean + off, len, rccl::ncclFloat64, from, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Recv(variance + off, len, rccl::ncclFloat64, from, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Recv(samples + off, len, rccl::ncclInt32, from, c->comm, ctx->stream));
            }
        }
    }
    return g.finish();
}

extern "C" int rsx_allgather_frame(rsx_comm *c, double *mean, double *variance, int32_t *samples, const int64_t *shard_begin) {
    if (!c || !mean || !variance || !samples || !shard_begin) return rsx_fail(RSX_EINVAL, "rsx_allgather_frame: null argument");
    for (int r = 0; r < c->n_ranks; ++r) if (shard_begin[r] < 0 || shard_begin[r + 1] < shard_begin[r]) return rsx_fail(RSX_EINVAL, "rsx_allgather_frame: shard offsets must be non-decreasing");
    rsx_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    return gather_runs(c, mean, variance, samples, shard_begin);
}

static int ranks_agree(rsx_comm *c, bool ok_here, bool *ok_everywhere) {
    double bad = ok_here ? 0.0 : 1.0;
    const int rc = rsx_comm_max_f64(c, &bad);
    if (rc) return rc;
    *ok_everywhere = bad == 0.0;
    return RSX_OK;
}

extern "C" int rsx_comm_size(rsx_comm *c, int32_t *n_ranks) {
    if (!c || !n_ranks) return rsx_fail(RSX_EINVAL, "rsx_comm_size: null argument");
    int n = c->n_ranks;
    if (rccl::api().CommCount) RCCL_TRY(rccl::api().CommCount(c->comm, &n));     // what RCCL itself says the communicator spans
    *n_ranks = n;
    return RSX_OK;
}

// The 1/W segment of an n-element frame that rank r owns in the reduce-scatter of rsx_allreduce_frame: ceil(n / W) elements each, the
// last ranks' segments cut at n (possibly empty). Host arithmetic, exported so that it can be tested without a GPU.
extern "C" int rsx_frame_segment(int64_t n, int32_t n_ranks, int32_t rank, int64_t *offset, int64_t *length) {
    if (n < 0 || n_ranks < 1 || rank < 0 || rank > n_ranks || !offset || !length) return rsx_fail(RSX_EINVAL, "rsx_frame_segment: bad arguments");
    const uint64_t N = (uint64_t)n, W = (uint64_t)n_ranks, seg = (N + W - 1) / W;
    const uint64_t off = std::min(N, seg * (uint64_t)rank), end = std::min(N, seg * ((uint64_t)rank + 1));
    *offset = (int64_t)off;
    *length = rank == n_ranks ? 0 : (int64_t)(end - off);                  // (rank == n_ranks: the end of the last segment)
    return RSX_OK;
}

// Sample sharding: every rank holds a full n-element frame of its own samples; afterwards every rank holds
// combine(...combine(combine(rank 0, rank 1), rank 2)..., rank W-1) — StatsArray3D.combine_samples (statsarray.pyx:780-859) folded in
// rank order, element by element, whatever the routing. Reduce-scatter (point-to-point sends of 1/W segments, fold on the owner)
// followed by an all-gather of the merged segments.
extern "C" int rsx_allreduce_frame(rsx_comm *c, double *mean, double *variance, int32_t *samples, int64_t n) {
    if (!c || !mean || !variance || !samples || n < 0) return rsx_fail(RSX_EINVAL, "rsx_allreduce_frame: bad arguments");
    if (n == 0 || c->n_ranks == 1) return RSX_OK;
    rsx_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    const int W = c->n_ranks, me = c->rank;
    auto seg_off = [&](int r) { int64_t o = 0, l = 0; (void)rsx_frame_segment(n, W, r, &o, &l); return (size_t)o; };
    auto seg_len = [&](int r) { int64_t o = 0, l = 0; (void)rsx_frame_segment(n, W, r, &o, &l); return (size_t)l; };
    const size_t mine = seg_len(me);
    // scratch: W copies of my segment, as [W][mine] doubles (mean), [W][mine] doubles (variance), [W][mine] int32 (samples)
    const size_t need = (size_t)W * mine * 20 + 256;
    bool have = true;
    if (need > c->scratch_bytes) {
        if (c->scratch) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(c->scratch)); c->scratch = nullptr; c->scratch_bytes = 0; }
        if (hipMalloc(&c->scratch, need) == hipSuccess) c->scratch_bytes = need;
        else { (void)hipGetLastError(); c->scratch = nullptr; have = false; }
    }
    bool all_have = false;
    const int arc = ranks_agree(c, have, &all_have);
    if (arc) return arc;
    if (!all_have) return rsx_fail(RSX_ENOMEM, "rsx_allreduce_frame: %s could not allocate its %zu-byte segment workspace", have ? "another rank" : "this rank", need);
    double *sm = static_cast<double *>(c->scratch), *sv = sm + (size_t)W * mine;
    int32_t *sn = reinterpret_cast<int32_t *>(sv + (size_t)W * mine);
    rccl::Api &a = rccl::api();
    {
        GroupGuard g(a);
        g.start();
        for (int r = 0; r < W; ++r) {
            if (r == me) continue;
            const size_t off = seg_off(r), len = seg_len(r);
            if (len) {
                RCCL_IN_GROUP(g, a.Send(mean + off, len, rccl::ncclFloat64, r, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Send(variance + off, len, rccl::ncclFloat64, r, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Send(samples + off, len, rccl::ncclInt32, r, c->comm, ctx->stream));
            }
            if (mine) {
                RCCL_IN_GROUP(g, a.Recv(sm + (size_t)r * mine, mine, rccl::ncclFloat64, r, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Recv(sv + (size_t)r * mine, mine, rccl::ncclFloat64, r, c->comm, ctx->stream));
                RCCL_IN_GROUP(g, a.Recv(sn + (size_t)r * mine, mine, rccl::ncclInt32, r, c->comm, ctx->stream));
            }
        }
        const int grc = g.finish();
        if (grc) return grc;
    }
    if (mine) {
        const size_t off = seg_off(me);
        HIP_TRY(hipMemcpyAsync(sm + (size_t)me * mine, mean + off, mine * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(sv + (size_t)me * mine, variance + off, mine * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(sn + (size_t)me * mine, samples + off, mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        for (int r = 1; r < W; ++r) {                       // fold in rank order into rank 0's copy
            hipLaunchKernelGGL(k_frame_combine, dim3((unsigned)((mine + 255) / 256)), dim3(256), 0, ctx->stream, (long long)mine, sm, sv, sn,
                               sm + (size_t)r * mine, sv + (size_t)r * mine, sn + (size_t)r * mine);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(mean + off, sm, mine * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(variance + off, sv, mine * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(samples + off, sn, mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    std::vector<int64_t> begin((size_t)W + 1);
    for (int r = 0; r <= W; ++r) begin[(size_t)r] = (int64_t)seg_off(r);
    return gather_runs(c, mean, variance, samples, begin.data());
}

// Slice sharding (the reference's slice loop, observer.pyx:299-340): rank r rendered the spectral slices that fill bins
// [bin_begin[r], bin_begin[r + 1]) of every pixel of the [n_pixels, bins] frame arrays (bin fastest, as StatsArray3D). A rank's share is
// a strided set of bin planes, so it travels packed: pack own bins -> one group of direct sends / receives of the packed blocks (every
// byte crosses one xGMI link once) -> unpack the peers' blocks into place. No arithmetic: bit-identical to a one-GPU render.
template <typename T>
__global__ void k_pack_bins(const T *frame, T *block, long long n_pixels, int bins, int b0, int nb, int unpack) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels * nb) return;
    const long long p = i / nb;
    const int j = (int)(i - p * nb);
    if (unpack) const_cast<T *>(frame)[p * bins + b0 + j] = block[i];
    else block[i] = frame[p * bins + b0 + j];
}

extern "C" int rsx_allgather_bins(rsx_comm *c, double *mean, double *variance, int32_t *samples, int64_t n_pixels, int32_t bins, const int32_t *bin_begin) {
    if (!c || !mean || !variance || !samples || !bin_begin || n_pixels < 0 || bins < 1) return rsx_fail(RSX_EINVAL, "rsx_allgather_bins: bad arguments");
    const int W = c->n_ranks, me = c->rank;
    if (bin_begin[0] != 0 || bin_begin[W] != bins) return rsx_fail(RSX_EINVAL, "rsx_allgather_bins: the bin ranges must cover [0, bins)");
    for (int r = 0; r < W; ++r) if (bin_begin[r + 1] < bin_begin[r]) return rsx_fail(RSX_EINVAL, "rsx_allgather_bins: bin offsets must be non-decreasing");
    if (W == 1 || n_pixels == 0) return RSX_OK;
    rsx_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    // The packed blocks of all ranks together are one frame's worth (10.7 GB for configs[4]). They never need to exist at once: the frame
    // travels in chunks of whole pixels — pack own bins of the chunk, one group of direct sends / receives, unpack the peers' — through a
    // workspace of at most RSX_BINS_SCRATCH_BYTES (default 512 MB; stream order keeps a chunk's unpack ahead of the next chunk's receives;
    // read at every call, every rank must see the same value: the chunks of a pair of ranks have to match).
    const char *cap_env = std::getenv("RSX_BINS_SCRATCH_BYTES");
    const size_t scratch_cap = (size_t)std::max(1ll, cap_env ? std::atoll(cap_env) : (512ll << 20));
    const size_t per_pixel = (size_t)bins * 20;
    const size_t chunk_pixels = std::max<size_t>(1, std::min<size_t>((size_t)n_pixels, scratch_cap / per_pixel));
    const size_t need = chunk_pixels * per_pixel + 256;
    bool have = true;
    if (need > c->scratch_bytes) {
        if (c->scratch) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(c->scratch)); c->scratch = nullptr; c->scratch_bytes = 0; }
        if (hipMalloc(&c->scratch, need) == hipSuccess) c->scratch_bytes = need;
        else { (void)hipGetLastError(); c->scratch = nullptr; have = false; }
    }
    bool all_have = false;
    const int arc = ranks_agree(c, have, &all_have);
    if (arc) return arc;
    if (!all_have) return rsx_fail(RSX_ENOMEM, "rsx_allgather_bins: %s could not allocate its %zu-byte packing workspace", have ? "another rank" : "this rank", need);
    rccl::Api &a = rccl::api();
    for (size_t p0 = 0; p0 < (size_t)n_pixels; p0 += chunk_pixels) {
        const size_t P = std::min(chunk_pixels, (size_t)n_pixels - p0), E = P * (size_t)bins;      // this chunk: pixels [p0, p0 + P)
        double *pm = static_cast<double *>(c->scratch), *pv = pm + E;
        int32_t *pn = reinterpret_cast<int32_t *>(pv + E);
        double *cm = mean + p0 * (size_t)bins, *cv = variance + p0 * (size_t)bins;
        int32_t *cn = samples + p0 * (size_t)bins;
        auto block_off = [&](int r) { return P * (size_t)bin_begin[r]; };      // blocks in rank order, [P][nb_r] each
        auto block_len = [&](int r) { return P * (size_t)(bin_begin[r + 1] - bin_begin[r]); };
        auto launch = [&](int r, int unpack) -> int {
            const size_t len = block_len(r);
            if (!len) return RSX_OK;
            const dim3 grid((unsigned)((len + 255) / 256));
            const int b0 = bin_begin[r], nb = bin_begin[r + 1] - bin_begin[r];
            hipLaunchKernelGGL(k_pack_bins<double>, grid, dim3(256), 0, ctx->stream, cm, pm + block_off(r), (long long)P, (int)bins, b0, nb, unpack);
            hipLaunchKernelGGL(k_pack_bins<double>, grid, dim3(256), 0, ctx->stream, cv, pv + block_off(r), (long long)P, (int)bins, b0, nb, unpack);
            hipLaunchKernelGGL(k_pack_bins<int32_t>, grid, dim3(256), 0, ctx->stream, cn, pn + block_off(r), (long long)P, (int)bins, b0, nb, unpack);
            HIP_TRY(hipGetLastError());
            return RSX_OK;
        };
        int rc = launch(me, 0);
        if (rc) return rc;
        {
            GroupGuard g(a);
            g.start();
            const size_t my_off = block_off(me), my_len = block_len(me);
            for (int k = 1; k < W; ++k) {
                const int to = (me + k) % W, from = (me - k + W) % W;
                if (my_len) {
                    RCCL_IN_GROUP(g, a.Send(pm + my_off, my_len, rccl::ncclFloat64, to, c->comm, ctx->stream));
                    RCCL_IN_GROUP(g, a.Send(pv + my_off, my_len, rccl::ncclFloat64, to, c->comm, ctx->stream));
                    RCCL_IN_GROUP(g, a.Send(pn + my_off, my_len, rccl::ncclInt32, to, c->comm, ctx->stream));
                }
                const size_t off = block_off(from), len = block_len(from);
                if (len) {
                    RCCL_IN_GROUP(g, a.Recv(pm + off, len, rccl::ncclFloat64, from, c->comm, ctx->stream));
                    RCCL_IN_GROUP(g, a.Recv(pv + off, len, rccl::ncclFloat64, from, c->comm, ctx->stream));
                    RCCL_IN_GROUP(g, a.Recv(pn + off, len, rccl::ncclInt32, from, c->comm, ctx->stream));
                }
            }
            rc = g.finish();
            if (rc) return rc;
        }
        for (int r = 0; r < W; ++r) if (r != me && (rc = launch(r, 1))) return rc;
    }
    return RSX_OK;
}
