// rsx_device.hip — gfx950 (MI355X / CDNA4) device side of librsx: KD-tree traversal, ray/primitive
// intersection, pinhole ray generation, closed-form shading and spectral accumulation.
//
// Design:
//  * one ray per lane, 64-wide wavefronts, persistent workgroups pulling 64-ray units from per-XCD work lists; a unit is 64
//    consecutive rays of the numbering pixel * spp + sample (one pixel x 64 samples at 64 spp), so the lanes of a wave walk nearly
//    the same path;
//  * explicit per-lane traversal stacks in LDS, laid out [level][lane] so that every ds_read/ds_write of a wave is bank-conflict
//    free whatever level each lane is at; an entry is (far node id, far tmax): the far range's tmin is the tmax of the leaf that
//    was just exhausted, so it is never stored; deeper levels spill to a per-wave global array;
//  * KD nodes are 16 B (one dwordx4 load, fetched as (node, node+1) pairs); mesh leaves read leaf-ordered 64 B triangle records
//    (9 vertex floats + face normal + triangle id: no index indirection); big leaves are tested by the whole wave;
//  * arithmetic follows the reference operation for operation (f64 traversal and analytic primitives, f32 watertight triangle
//    test with its f64 casts): compiled with -ffp-contract=off, IEEE div/sqrt — the per-step division is the one exception in
//    form, not in value: a hoisted, refined reciprocal reproduces the correctly rounded quotient (exact_div);
//  * 168 registers per wave = three waves per SIMD; the register diet that got there is described in DESIGN.md section 5;
//  * no MFMA: the path is branchy traversal, bound by instruction issue and divergence, not by a contraction.
//
// The device code is one translation unit split by subject: dev_common.hpp (records, stacks, KD step), dev_mesh.hpp, dev_analytic.hpp,
// dev_csg.hpp, dev_world.hpp, dev_query_kernels.hpp (hit / roots / contains), dev_render.hpp (observe: trace, scheduling, Welford).
//
// Reference lines each device function restates are cited at the function.
#include <hip/hip_runtime.h>

#include <time.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <tuple>
#include <vector>
#include <functional>
#include <map>
#include <unordered_map>

#include "../../include/rsx.h"
#include "rsx_internal.h"

#define WAVE 64
#ifndef WG_WAVES
#define WG_WAVES 4
#endif
#define WG_THREADS (WAVE * WG_WAVES)
#ifndef RSX_COHERENT_MIN_SPP
#define RSX_COHERENT_MIN_SPP 1      // passes with more samples per pixel than this run the always-stage instantiation of k_render_trace
#endif
#ifndef RSX_WORLD_LDS_LEVELS
#define RSX_WORLD_LDS_LEVELS 3      // LDS-resident traversal stack entries per lane, world tree. 3 + 10 levels + the staging area are
                                    // 13.0 KB per wave = 52 KB per workgroup: three workgroups (three waves per SIMD) fit the 160 KB LDS
#endif
#ifndef RSX_MESH_LDS_LEVELS
#define RSX_MESH_LDS_LEVELS 10      // ... mesh tree (deeper entries spill to global memory)
#endif
#ifndef RSX_MAX_WG_PER_CU
#define RSX_MAX_WG_PER_CU 8
#endif
#define STAGE_BYTES (WAVE * 52)     // per-wave leaf staging area: 64 x (48-byte triangle record + 4-byte id)
#ifndef RSX_MAX_LANES
#define RSX_MAX_LANES 8
#endif
#ifndef RSX_RENDER_WG_PER_CU
#define RSX_RENDER_WG_PER_CU 1
#endif
#ifndef RSX_LPT_SCHEDULE
#define RSX_LPT_SCHEDULE 1          // reorder the 64-ray units of a repeated pass longest-first using the previous pass's timings
#endif
#ifndef RSX_MIN_WAVES_PER_SIMD
#define RSX_MIN_WAVES_PER_SIMD 3    // __launch_bounds__ second argument for the mesh/analytic traversal kernels: 168 registers per wave.
                                    // History on configs[2] (268 M rays): unconstrained the compiler took 260 registers and the hardware
                                    // ran ONE wave per SIMD, 163 ms; capped at 256 (two waves) 87 ms; after the register diet (uniform
                                    // stack bases, scalar per-primitive loads, kernarg re-reads, scalar owner ray; tools/vgpr_live.py)
                                    // three waves with ~30 cold spills, 72 ms. Four waves still spill hot values and lose.
#endif

#ifndef RSX_CSG_KEEP_DIRECTION
#define RSX_CSG_KEEP_DIRECTION 1    // csg_fast_hit_uniform: steps into spaces whose to_local keeps directions skip the direction's arithmetic and reciprocals
#endif
#ifndef RSX_PREFILL_CULL
#define RSX_PREFILL_CULL 0          // the CSG prefill round skips solids whose box lies beyond the nearest answer of the wide analytic primitives:
                                    // exact (boxes are padded by 1e-9, csg.pyx:39) and without effect — configs[4] 8.90 s per step without, 8.95 with: a wave
                                    // evaluates a solid when ANY lane asks, and among 38 incoherent rays one nearly always does. With the questions packed 64 to a
                                    // turn (round 6, the gate of the packed round): 7.05 s without, 7.10 with — 92 questions per round are two turns either way. Off.
#endif
#ifndef RSX_PREFILL_UNIFORM_MIN
#define RSX_PREFILL_UNIFORM_MIN 65  // packed prefill: a solid at least this many lanes ask about is answered by the wave-wide evaluator instead (65: never, the
                                    // block is not compiled). Measured at 32 (round 5): configs[4] 7.95 -> 12.0 s per step, frames equal — two inlined
                                    // evaluators in one kernel of 256 registers; the packed turns alone are the better form.
#endif
#ifndef RSX_PREFILL_PACK
#define RSX_PREFILL_PACK 1          // the CSG prefill round of the path kernels deals its (ray, solid) questions out 64 to a turn (dev_world.hpp)
#endif
#ifndef RSX_PREFILL_UNIFORM
#define RSX_PREFILL_UNIFORM 1       // the CSG prefill round of the path kernels through the wave-wide evaluator (dev_world.hpp)
#endif
#ifndef RSX_PACKET_MIN_WAVES
#define RSX_PACKET_MIN_WAVES 4      // launch-bounds waves per SIMD of the packet instantiation of k_render_trace (128 registers)
#endif
#ifndef RSX_PACKET_CSG_MIN_WAVES
#define RSX_PACKET_CSG_MIN_WAVES 2  // ... and of its CSG form (the state-free evaluator's registers)
#endif
#ifndef RSX_PACKET_MIN_SPP
#define RSX_PACKET_MIN_SPP 8        // passes with at least this many samples per pixel walk the trees as packets (dev_packet.hpp): a 64-ray
                                    // unit then holds at most eight pixels (scenes of a few primitives: from 4, CSG scenes: always — render()). Measured on the configs[2] scene at 1024 x 1024 (trace kernel,
                                    // per-lane walk -> packet walk): 64 spp 8.2 -> 5.6 ms per 2^24 rays, 32 spp 7.2 -> 6.5, 16 spp 9.9 -> 8.8,
                                    // 8 spp 6.0 -> 5.9, 4 spp 3.2 -> 3.6, 2 spp 1.8 -> 2.4, 1 spp (8 x 8 pixel tiles) 0.85 -> 1.20: the union of
                                    // the nodes 16 or more different pixels visit outgrows what the shared walk saves. $RSX_PACKET_MIN_SPP.
#endif

#ifndef RSX_UTIL_PROF
#define RSX_UTIL_PROF 0            // 1: lane-utilisation counters per loop level into the rsx_debug_unit_times buffer (tuning builds only)
#endif
#if RSX_UTIL_PROF
// one elected lane adds (active lanes, 64) to a pair of per-unit counters; `acc` points at the unit's slots in global memory
#define UTIL_COUNT(acc, slot) { const unsigned long long ex_ = __ballot(true); \
    if (acc && (int)(threadIdx.x % WAVE) == __ffsll((long long)ex_) - 1) { (acc)[slot] += __popcll(ex_); (acc)[(slot) + 1] += WAVE; } }
#else
#define UTIL_COUNT(acc, slot)
#endif

#ifndef RSX_CSG_ARENA_WG_PER_CU
#define RSX_CSG_ARENA_WG_PER_CU 2   // grid limit (workgroups per CU) of scenes whose CSG node states live in the scene's arena
#endif
#ifndef RSX_CSG_MIN_WAVES
#define RSX_CSG_MIN_WAVES 1         // launch-bounds waves per SIMD of the CSG instantiations of the traversal kernels
#endif

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
static thread_local std::string g_error;

// crude host-side profiler of the render call path (RSX_HOST_PROF=1): seconds spent per section, printed by rsx_synchronize
static double g_hp[8];
static long g_hp_calls;
static bool g_hp_on = std::getenv("RSX_HOST_PROF") != nullptr;
static inline double hp_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define HP_BEGIN double hp_t0_ = g_hp_on ? hp_now() : 0.0;
#define HP_MARK(slot) if (g_hp_on) { const double n_ = hp_now(); g_hp[slot] += n_ - hp_t0_; hp_t0_ = n_; }

int rsx_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

extern "C" const char *rsx_last_error(void) { return g_error.c_str(); }
extern "C" const char *rsx_version(void) { return "librsx 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return rsx_fail(RSX_EHIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// device code (one translation unit; the pieces are split by subject)
// ---------------------------------------------------------------------------------------------------
#include "dev_common.hpp"
#include "dev_mesh.hpp"
#include "dev_analytic.hpp"
#include "dev_csg.hpp"
#include "dev_world.hpp"
#include "dev_packet.hpp"
#include "dev_query_kernels.hpp"
#include "dev_render.hpp"
#include "dev_wavefront.hpp"
#include "dev_selftest.hpp"

// ---------------------------------------------------------------------------------------------------
// host API
// ---------------------------------------------------------------------------------------------------
#define RING_SLOTS 512
enum { POOL_MATERIALS, POOL_TABLES, POOL_TASKS, POOL_QUERY, POOL_MEAN, POOL_VAR, POOL_SLOTS };

// Per-stream state of the traversal kernels. `main` runs on the ctx stream (hit / roots / contains batches, unpipelined renders);
// up to RSX_MAX_LANES more lanes with private streams let consecutive small render passes overlap: the long tail of pass p (a few waves walking grazing
// rays through hundreds of cells) runs while pass p+1's bulk fills the rest of the chip. Accumulation into the frame stays on the
// ctx stream, in call order.
struct TraceLane {
    hipStream_t stream = nullptr;
    unsigned long long *ticket = nullptr;
    bool ticket_armed = false;         // ticket is known to be zero on the stream (left so by k_accumulate)
    long long cost_zeroed = 0;         // units of unit_cost known to be zero on the stream (left so by k_order_units after a path pass), 0 = unknown
    long long redo_zeroed = 0;         // units of the redo mask known to be zero on the stream (left so by k_accumulate), 0 = unknown
    void *spill = nullptr;             // global spill regions for the traversal stacks
    size_t spill_bytes = 0;
    uint32_t *unit_cost = nullptr, *unit_order = nullptr, *n_work = nullptr;   // longest-first scheduling state
    size_t unit_capacity = 0;
    long long cost_units = 0;          // number of units unit_cost currently describes (0 = none)
    long long order_units = 0;         // number of units unit_order was sorted for (0 = no valid work list)
    uint64_t cost_signature = 0;       // (scene, camera, tasks) the costs were measured on
    int sorts_done = 0;                // longest-first sorts since the signature last changed
    void *redo = nullptr;              // CSG scenes: per-unit lane masks handed from the fast pass to the redo pass
    size_t redo_bytes = 0;
    void *samples = nullptr, *uniforms = nullptr, *terms = nullptr, *tail = nullptr;   // terms / tail: PathTerm blocks of the path kernel
    size_t samples_bytes = 0, uniforms_bytes = 0, terms_bytes = 0, tail_bytes = 0;
    void *order_counts = nullptr;      // k_order_natural_*: per-block list counts / offsets
    size_t order_counts_bytes = 0;
    void *xs = nullptr;                // k_path_values -> k_merge_passes: one value per (pixel, bin, record) of a call of several one-sample path passes
    size_t xs_bytes = 0;
    void *ring = nullptr;              // fused passes: FUSE_UNITS x 64 sample records per wave of the grid
    size_t ring_bytes = 0;
    void *path_queue = nullptr;        // path passes: PathState records handed from the first launch's retiring waves to the drain launch
    size_t path_queue_bytes = 0;
    void *wf_paths = nullptr, *wf_hits = nullptr, *wf_lists = nullptr, *wf_counts = nullptr;   // path passes in stages (dev_wavefront.hpp)
    size_t wf_paths_bytes = 0, wf_hits_bytes = 0, wf_lists_bytes = 0, wf_counts_bytes = 0;
    unsigned int *overflow = nullptr;
    // pinned host words the lane's small read-backs land in (flags, ray counts, list counts): a device-to-host copy into pageable memory
    // — a stack variable — goes through the runtime's pin / staging path, and that path is where a process that had made a second device
    // scene lost 60 - 90 ms, two or three times in the following 300 ms (tools/r6_world_stalls.py, profiles/r06_world_stalls.txt: the time
    // sits inside hipMemcpyAsync, the device idle behind the copy kernel)
    unsigned long long *host_words = nullptr;
    hipEvent_t traced = nullptr, merged = nullptr;
    bool in_flight = false;
    hipEvent_t sort_from = nullptr, sorted = nullptr;   // a lone path pass's work-list sort on the context's side stream (render())
    bool sort_pending = false;
    // deferred path passes (rsx_defer_path_checks): the lane's own copies of the per-call material / table blobs, and the check that
    // is still owed for the pass it ran last
    void *mat_dev = nullptr, *tab_dev = nullptr;
    size_t mat_dev_bytes = 0, tab_dev_bytes = 0;
    std::vector<unsigned char> mat_host, tab_host;
    bool check_pending = false;
    int32_t pending_call = -1;
};

struct rsx_ctx {
    int device;
    hipStream_t stream;        // launch stream (own or external)
    hipStream_t own_stream;
    hipEvent_t ev0, ev1, ev2;  // ev0..ev1 = last traversal kernel, ev1..ev2 = last accumulate kernel
    TraceLane main, lanes[RSX_MAX_LANES];
    int render_wg_override;    // env RSX_RENDER_WG: workgroups per CU of a pipelined pass (tuning aid; 0 = heuristic)
    int pipeline_depth;        // 1 = renders run on the ctx stream only; n = rotate over n private lanes
    int path_lanes;            // lanes the deferred path passes rotate over (>= pipeline_depth lanes exist then)
    long long max_in_flight;   // render passes the host may run ahead of the device
    bool timing;               // record per-call timing events (rsx_render_history); off removes four timed events per pass
    std::vector<hipEvent_t> gate;   // untimed completion event per recent pass (host run-ahead throttle)
    int n_cus;
    float last_ms;
    bool have_accum;
    // ring of per-render-call event triples so a caller can time K back-to-back async renders without syncing
    std::vector<hipEvent_t> ring;      // 4 events per slot: trace begin/end (lane stream), merge begin/end (ctx stream)
    long long render_calls;
    unsigned long long *unit_times;    // debug: per-unit timestamps of the next render calls (caller-owned device buffer)
    std::vector<unsigned char> shadow[3];   // host copies of what POOL_MATERIALS / POOL_TABLES / POOL_TASKS hold
    // grow-only device workspace so steady-state render calls never hipMalloc
    void *pool[POOL_SLOTS];
    size_t pool_bytes[POOL_SLOTS];
    void *staging;             // pinned host mirror of small query workspaces: one copy in, one copy out per call
    size_t staging_bytes;
    // Device blocks released by scenes and frames, kept for the next scene / frame (cached_alloc / cached_release below): on this
    // runtime a hipFree in a process of a few GB costs ~80 ms, and a host program that builds a second world meets one for every buffer
    // of the first whenever Python's collector gets round to it (tools/r5_path_batches_diag.py: two or three 85 ms calls among 2 ms ones).
    std::multimap<size_t, void *> cache_free;
    std::unordered_map<void *, size_t> cache_size;
    size_t cache_bytes = 0;
    // rsx_defer_path_checks: path passes run on the private lanes and their end-of-pass checks are collected later
    int32_t wf_mode = -1;                          // rsx_set_path_stages
    long long wf_min_paths = -1;
    bool path_general = false;                     // rsx_set_path_stages mode 2 / 3: scenes without a mesh keep the kernel forms that carry the mesh walk
    bool defer_path = false;
    int32_t deferred_calls = 0;                    // path passes issued since deferral was switched on
    std::vector<int32_t> deferred_failed;          // ... of which these must be rendered again (term arena ran out, too many volumes at a point)
    unsigned int deferred_error_flags = 0;
    unsigned long long deferred_rays = 0;
    double *acc_consts = nullptr;                  // {(double)i, refine_rcp(i)} for the accumulate kernel's steps (k_fill_acc_consts)
    hipStream_t sort_stream = nullptr;             // k_order_units of lone path passes: beside the replay instead of in front of it
};

// the check a deferred path pass still owes: wait for its merge, read its flags and ray count
static int settle_lane(rsx_ctx *ctx, TraceLane &ln) {
    if (!ln.check_pending) return RSX_OK;
    HIP_TRY(hipEventSynchronize(ln.merged));
    // (pinned words, see TraceLane::host_words; a blocking copy that waits for nothing but itself: the lane's merge is over — the event —
    // and the context stream holds the merges of every OTHER lane in flight, which this read-back must not wait for)
    unsigned long long *words = ln.host_words;
    HIP_TRY(hipMemcpy(words, ln.overflow, 16, hipMemcpyDeviceToHost));
    const unsigned int flags = (unsigned int)(words[0] & 0xffffffffu);
    if (!(flags & 7u)) ctx->deferred_rays += words[1];    // (a pass that is rendered again reports its rays then)
    if (flags & 2u) ctx->deferred_error_flags |= 2u;
    else if (flags & 5u) ctx->deferred_failed.push_back(ln.pending_call);
    ln.check_pending = false;
    return RSX_OK;
}

// hipMalloc for the library's own working buffers (pools, lane buffers, spill regions, work lists): when the device is full, the blocks
// the cache holds back for the next scene / frame (cached_release: up to 8 GB) are returned to the runtime and the request is made
// once more — they are free memory as far as any caller can tell.
static hipError_t malloc_or_flush(rsx_ctx *ctx, void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && !ctx->cache_free.empty()) {
        (void)hipGetLastError();
        for (auto &b : ctx->cache_free) { ctx->cache_size.erase(b.second); (void)hipFree(b.second); }
        ctx->cache_free.clear(); ctx->cache_bytes = 0;
        e = hipMalloc(out, bytes);
    }
    return e;
}
template <typename T> static hipError_t malloc_or_flush(rsx_ctx *ctx, T **out, size_t bytes) { return malloc_or_flush(ctx, reinterpret_cast<void **>(out), bytes); }

static int pool_get(rsx_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes > ctx->pool_bytes[slot]) {
        if (ctx->pool[slot]) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->pool[slot])); ctx->pool[slot] = nullptr; ctx->pool_bytes[slot] = 0; }
        const size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(malloc_or_flush(ctx, &ctx->pool[slot], want));
        ctx->pool_bytes[slot] = want;
        if (slot <= POOL_TASKS) ctx->shadow[slot].clear();   // new storage holds nothing yet
    }
    *out = ctx->pool[slot];
    return RSX_OK;
}

struct RelJobHost { int32_t prim, mesh; long long offset, n_items; };
struct rsx_scene {
    rsx_ctx *ctx;
    DScene d;
    std::vector<void *> allocs;
    int32_t n_world;
    bool has_csg;
    // camera-relative leaf records of the world's mesh instances (dev_packet.hpp), made on demand for the camera of a packet pass
    std::vector<RelJobHost> rel_jobs;
    long long rel_records = 0;
    void *rel = nullptr, *rel_info = nullptr, *rel_jobs_dev = nullptr;
    double rel_camera[16];
    bool rel_valid = false, rel_refused = false;
};

extern "C" int rsx_init(int device_ordinal, rsx_ctx **out) {
    if (!out) return rsx_fail(RSX_EINVAL, "rsx_init: null out");
    // The render lanes are HIP streams that must run side by side; the runtime maps a process's streams onto four hardware queues unless
    // told otherwise, and two lanes on one queue run one after the other (prism, 512 slices: 3.6 s with four queues, 2.0 s with
    // twelve). Read when the runtime initialises: a process that has used HIP before this call keeps what it had.
    setenv("GPU_MAX_HW_QUEUES", "12", 0);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return rsx_fail(RSX_ENODEV, "no HIP device visible");
    if (device_ordinal < 0 || device_ordinal >= count) return rsx_fail(RSX_ENODEV, "device ordinal %d out of range (0..%d)", device_ordinal, count - 1);
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rsx_fail(RSX_ENODEV, "device %d is %s; librsx is built for gfx950 only", device_ordinal, prop.gcnArchName);
    rsx_ctx *ctx = new (std::nothrow) rsx_ctx();
    if (!ctx) return rsx_fail(RSX_ENOMEM, "out of host memory");
    struct Guard { rsx_ctx *c; ~Guard() { if (c) rsx_free(c); } } guard{ctx};      // a failing HIP call below must not leak the half-built ctx
    ctx->stream = ctx->own_stream = nullptr;
    ctx->ev0 = ctx->ev1 = ctx->ev2 = nullptr;
    ctx->device = device_ordinal;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->last_ms = 0.f;
    ctx->have_accum = false;
    ctx->render_calls = 0;
    ctx->unit_times = nullptr;
    {
        const char *env = std::getenv("RSX_PIPELINE");
        ctx->pipeline_depth = env ? std::atoi(env) : 3;    // three lanes x one workgroup per CU = the three waves per SIMD the kernel fits
        if (ctx->pipeline_depth < 1) ctx->pipeline_depth = 1;
        if (ctx->pipeline_depth > RSX_MAX_LANES) ctx->pipeline_depth = RSX_MAX_LANES;
        env = std::getenv("RSX_RENDER_WG");
        ctx->render_wg_override = env ? std::atoi(env) : 0;
        // path passes whose checks are deferred (the spectral slices of one observe()): each is a bulk of a few milliseconds and a
        // drain launch of ten and more (a few paths trapped in glass), so many of them are kept in flight — one lane, one stream each
        // (six by default: a configs[4] step takes the same time on four, six or eight lanes — round 5: 9.81 / 9.72 / 9.75 s — and every stream is an
        // HSA queue whose scratch the runtime reserves for the largest kernel it has run (the CSG redo form: 7 KB per lane): with eight
        // lanes a process that had rendered three other path-traced scenes before the prism aborted with HSA_STATUS_ERROR_OUT_OF_RESOURCES
        // at the eighth queue, with seven it did not (tools/r5_handed_case.py). Two queues of margin.)
        env = std::getenv("RSX_PATH_LANES");
        ctx->path_lanes = env ? std::atoi(env) : 6;
        if (ctx->path_lanes < 1) ctx->path_lanes = 1;
        if (ctx->path_lanes > RSX_MAX_LANES) ctx->path_lanes = RSX_MAX_LANES;
        const char *env2 = std::getenv("RSX_MAX_IN_FLIGHT");
        ctx->max_in_flight = env2 ? std::atoll(env2) : 16;
        if (ctx->max_in_flight < 1) ctx->max_in_flight = 1;
        if (ctx->max_in_flight > 48) ctx->max_in_flight = 48;
        const char *env4 = std::getenv("RSX_TIMING");
        ctx->timing = env4 ? std::atoi(env4) != 0 : true;
    }
    for (int i = 0; i < POOL_SLOTS; ++i) { ctx->pool[i] = nullptr; ctx->pool_bytes[i] = 0; }
    HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));
    HIP_TRY(hipEventCreate(&ctx->ev2));
    ctx->main.stream = ctx->stream;
    int lane_no = -1;
    std::vector<TraceLane *> all_lanes{&ctx->main};
    for (TraceLane &l8 : ctx->lanes) all_lanes.push_back(&l8);
    for (TraceLane *ln : all_lanes) {
        if (ln != &ctx->main && ++lane_no >= std::max(ctx->pipeline_depth, ctx->path_lanes)) continue;      // only the lanes in use get a stream (= an HSA queue)
        if (ln != &ctx->main && ctx->pipeline_depth < 2) continue;
        if (ln != &ctx->main) HIP_TRY(hipStreamCreateWithFlags(&ln->stream, hipStreamNonBlocking));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ln->host_words), 4096, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&ln->ticket, 2 * 9 * 16 * sizeof(unsigned long long)));   // one ticket per XCD list, a cache line apart; two sets (the second: the redo pass of a CSG path pass)
        HIP_TRY(hipEventCreateWithFlags(&ln->traced, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ln->merged, hipEventDisableTiming));
    }
    {
        const int n = ACC_CONSTS_ENTRIES;
        HIP_TRY(hipMalloc(&ctx->acc_consts, (size_t)n * 16));
        hipLaunchKernelGGL(k_fill_acc_consts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->acc_consts, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    guard.c = nullptr;
    *out = ctx;
    return RSX_OK;
}

extern "C" void rsx_free(rsx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->sort_stream) { (void)hipStreamSynchronize(ctx->sort_stream); (void)hipStreamDestroy(ctx->sort_stream); }
    std::vector<TraceLane *> all_lanes{&ctx->main};
    for (TraceLane &l8 : ctx->lanes) all_lanes.push_back(&l8);
    for (TraceLane *ln : all_lanes) {
        if (ln != &ctx->main && ln->stream) { (void)hipStreamSynchronize(ln->stream); (void)hipStreamDestroy(ln->stream); }
        for (void *q : {(void *)ln->ticket, ln->spill, (void *)ln->unit_cost, (void *)ln->unit_order, (void *)ln->n_work, ln->samples, ln->uniforms, ln->terms, ln->tail, ln->redo, ln->ring, ln->path_queue, ln->wf_paths, ln->wf_hits, ln->wf_lists, ln->wf_counts, (void *)ln->overflow, ln->mat_dev, ln->tab_dev, ln->order_counts, ln->xs})
            if (q) (void)hipFree(q);
        if (ln->host_words) (void)hipHostFree(ln->host_words);
        if (ln->traced) (void)hipEventDestroy(ln->traced);
        if (ln->merged) (void)hipEventDestroy(ln->merged);
        if (ln->sort_from) (void)hipEventDestroy(ln->sort_from);
        if (ln->sorted) (void)hipEventDestroy(ln->sorted);
    }
    for (int i = 0; i < POOL_SLOTS; ++i) if (ctx->pool[i]) (void)hipFree(ctx->pool[i]);
    for (auto &e : ctx->cache_free) (void)hipFree(e.second);
    if (ctx->acc_consts) (void)hipFree(ctx->acc_consts);
    if (ctx->staging) (void)hipHostFree(ctx->staging);
    for (hipEvent_t e : {ctx->ev0, ctx->ev1, ctx->ev2}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ring) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->gate) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int rsx_set_stream(rsx_ctx *ctx, void *hip_stream) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->main.stream = ctx->stream;
    return RSX_OK;
}

extern "C" int rsx_synchronize(rsx_ctx *ctx) {
#ifdef CSGF_COUNT
    { unsigned long long c[3] = {0, 0, 0}; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_csgf), sizeof(c));
      unsigned long long w[4] = {0, 0, 0, 0}; (void)hipMemcpyFromSymbol(w, HIP_SYMBOL(g_csgf_why), sizeof(w));
      double ex[8]; (void)hipMemcpyFromSymbol(ex, HIP_SYMBOL(g_csgf_ex), sizeof(ex));
      std::fprintf(stderr, "tie example: prim %g leaves %g %g t %.17g origin %.6f %.6f %.6f dx %.6f\n", ex[0], ex[1], ex[2], ex[3], ex[4], ex[5], ex[6], ex[7]);
      std::fprintf(stderr, "csg_fast_hit: fallback %llu miss %llu hit %llu | why: nan %llu pattern %llu tie %llu\n", c[0], c[1], c[2], w[0], w[1], w[2]); }
#endif
#ifdef RSX_PKT_PROF
    { unsigned long long c[16] = {0}; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_pkt), sizeof(c));
      if (c[0]) { const double u = (double)c[0];
        std::fprintf(stderr, "[packet walk, per unit over %llu units] world: steps %.2f divisions %.2f leaves %.2f items %.2f | mesh: visits %.3f steps %.2f leaves %.2f triangles %.2f slow steps %.2f | pushes %.2f pops %.2f deferrals %.4f\n",
                     c[0], c[1] / u, c[2] / u, c[3] / u, c[4] / u, c[5] / u, c[6] / u, c[7] / u, c[8] / u, c[12] / u, c[9] / u, c[10] / u, c[11] / u);
        unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pkt), z, sizeof(z)); } }
#endif
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    if (g_hp_on && g_hp_calls) {
        fprintf(stderr, "[rsx host prof] %ld render calls: throttle %.3f ms/call, setup+trace launch %.3f, events/order/wait %.3f, merge launch %.3f\n",
                g_hp_calls, 1e3 * g_hp[0] / g_hp_calls, 1e3 * g_hp[1] / g_hp_calls, 1e3 * g_hp[2] / g_hp_calls, 1e3 * g_hp[3] / g_hp_calls);
        g_hp_calls = 0; for (double &v : g_hp) v = 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->sort_stream) HIP_TRY(hipStreamSynchronize(ctx->sort_stream));
    return RSX_OK;
}

extern "C" int rsx_idle(rsx_ctx *ctx, int32_t *idle) {
    if (!ctx || !idle) return rsx_fail(RSX_EINVAL, "rsx_idle: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    *idle = 1;
    for (TraceLane &ln : ctx->lanes) {
        if (!ln.in_flight) continue;
        const hipError_t e = hipStreamQuery(ln.stream);
        if (e == hipErrorNotReady) { *idle = 0; return RSX_OK; }
        if (e != hipSuccess) return rsx_fail(RSX_EHIP, "hipStreamQuery: %s", hipGetErrorString(e));
    }
    const hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipErrorNotReady) *idle = 0;
    else if (e != hipSuccess) return rsx_fail(RSX_EHIP, "hipStreamQuery: %s", hipGetErrorString(e));
    if (*idle && ctx->sort_stream) {                        // (the work-list sort of a lone path pass runs beside its replay)
        const hipError_t es = hipStreamQuery(ctx->sort_stream);
        if (es == hipErrorNotReady) *idle = 0;
        else if (es != hipSuccess) return rsx_fail(RSX_EHIP, "hipStreamQuery: %s", hipGetErrorString(es));
    }
    return RSX_OK;
}

extern "C" int rsx_last_kernel_ms(rsx_ctx *ctx, float *ms) {
    if (!ctx || !ms) return rsx_fail(RSX_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = *ms;
    return RSX_OK;
}

extern "C" int rsx_last_render_ms(rsx_ctx *ctx, float *trace_ms, float *accumulate_ms) {
    if (!ctx || !trace_ms || !accumulate_ms) return rsx_fail(RSX_EINVAL, "null argument");
    if (!ctx->have_accum) return rsx_fail(RSX_EINVAL, "no render call has been issued on this context");
    return rsx_render_history(ctx, 1, trace_ms, accumulate_ms);
}

extern "C" int rsx_render_history(rsx_ctx *ctx, int32_t n, float *trace_ms, float *accumulate_ms) {
    if (!ctx || n < 1 || !trace_ms || !accumulate_ms) return rsx_fail(RSX_EINVAL, "rsx_render_history: bad arguments");
    if (!ctx->timing) return rsx_fail(RSX_EINVAL, "rsx_render_history: timing events are disabled (RSX_TIMING=0)");
    if (n > ctx->render_calls || n > RING_SLOTS) return rsx_fail(RSX_EINVAL, "rsx_render_history: only %lld calls recorded (ring of %d)", ctx->render_calls, RING_SLOTS);
    HIP_TRY(hipSetDevice(ctx->device));
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int32_t i = 0; i < n; ++i) {
        const long long call = ctx->render_calls - n + i;
        hipEvent_t *re = &ctx->ring[(size_t)(call % RING_SLOTS) * 4];
        HIP_TRY(hipEventElapsedTime(&trace_ms[i], re[0], re[1]));
        HIP_TRY(hipEventElapsedTime(&accumulate_ms[i], re[3], re[2]));
    }
    return RSX_OK;
}

extern "C" int rsx_selftest_exact_division(rsx_ctx *ctx, uint64_t n, uint64_t seed, uint64_t *mismatches) {
    if (!ctx || !mismatches) return rsx_fail(RSX_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 9 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d, 0, 9 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_selftest_division, dim3(ctx->n_cus * 8), dim3(256), 0, ctx->stream, (unsigned long long)n, (unsigned long long)seed, d);
    HIP_TRY(hipGetLastError());
    unsigned long long h[9] = {0};
    HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d));
    *mismatches = h[0];
    if (h[0]) rsx_fail(RSX_OK, "exact_div mismatches by class: %llu %llu %llu %llu %llu %llu %llu %llu", h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    return RSX_OK;
}

extern "C" int rsx_debug_unit_times(rsx_ctx *ctx, void *dev_buffer) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    // (production builds compile the stamps out — they cost the per-lane kernels registers in every pass: a buffer would stay unwritten and
    // only change which kernels the passes take. Switching it off is always allowed.)
    if (dev_buffer && !RSX_UNIT_STAMPS)
        return rsx_fail(RSX_EUNSUPPORTED, "rsx_debug_unit_times: this build has no unit stamps (build librsx with -DRSX_UNIT_STAMPS=1, -DRSX_PHASE_PROF=.. or -DRSX_UTIL_PROF=1)");
    ctx->unit_times = static_cast<unsigned long long *>(dev_buffer);
    return RSX_OK;
}

extern "C" int rsx_render_timeline(rsx_ctx *ctx, int32_t n, float *t) {
    if (!ctx || n < 1 || !t) return rsx_fail(RSX_EINVAL, "rsx_render_timeline: bad arguments");
    if (!ctx->timing || n > ctx->render_calls || n > RING_SLOTS) return rsx_fail(RSX_EINVAL, "rsx_render_timeline: not enough timed calls recorded");
    HIP_TRY(hipSetDevice(ctx->device));
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipEvent_t origin = ctx->ring[(size_t)((ctx->render_calls - n) % RING_SLOTS) * 4];
    for (int32_t i = 0; i < n; ++i) {
        hipEvent_t *re = &ctx->ring[(size_t)((ctx->render_calls - n + i) % RING_SLOTS) * 4];
        const int order[4] = {0, 1, 3, 2};                 // trace begin, trace end, merge begin, merge end
        for (int k = 0; k < 4; ++k) HIP_TRY(hipEventElapsedTime(&t[4 * i + k], origin, re[order[k]]));
    }
    return RSX_OK;
}

namespace {
// A block of at least `bytes`: a released one of up to twice the size when the cache holds one, else hipMalloc. The caller owns the
// stream order: blocks are released only after the streams that used them have been synchronised (rsx_scene_free, rsx_dev_free).
int cached_alloc(rsx_ctx *ctx, void **out, size_t bytes) {
    bytes = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    auto it = ctx->cache_free.lower_bound(bytes);
    if (it != ctx->cache_free.end() && it->first <= 2 * bytes + 4096) {
        *out = it->second;
        ctx->cache_bytes -= it->first;
        ctx->cache_free.erase(it);
        return RSX_OK;
    }
    HIP_TRY(malloc_or_flush(ctx, out, bytes));                 // (out of memory: the cache is given back and the request made once more)
    ctx->cache_size[*out] = bytes;
    return RSX_OK;
}

void cached_release(rsx_ctx *ctx, void *p) {
    if (!p) return;
    auto it = ctx->cache_size.find(p);
    if (it == ctx->cache_size.end()) { (void)hipFree(p); return; }
    const size_t bytes = it->second;
    if (bytes <= ((size_t)1 << 30) && ctx->cache_bytes + bytes <= ((size_t)8 << 30)) {
        ctx->cache_free.emplace(bytes, p);
        ctx->cache_bytes += bytes;
        return;
    }
    ctx->cache_size.erase(it);
    (void)hipFree(p);
}
}  // namespace

extern "C" int rsx_dev_alloc(rsx_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return rsx_fail(RSX_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return cached_alloc(ctx, dptr, bytes);
}

extern "C" int rsx_dev_free(rsx_ctx *ctx, void *dptr) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    // the block may go straight to the next rsx_dev_alloc / scene upload: everything that could still touch it must be over — the
    // context stream, passes in flight on the private lanes and the side stream of the work-list sort (what hipFree waited for by itself)
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) HIP_TRY(hipStreamSynchronize(ln.stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->sort_stream) HIP_TRY(hipStreamSynchronize(ctx->sort_stream));
    cached_release(ctx, dptr);
    return RSX_OK;
}

extern "C" int rsx_dev_upload(rsx_ctx *ctx, void *dptr, const void *host, size_t bytes) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSX_OK;
}

extern "C" int rsx_dev_download(rsx_ctx *ctx, void *host, const void *dptr, size_t bytes) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSX_OK;
}

extern "C" int rsx_dev_memset(rsx_ctx *ctx, void *dptr, int value, size_t bytes) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return RSX_OK;
}

// -- scene upload -----------------------------------------------------------------------------------
namespace {

template <typename T>
int upload(rsx_scene *sc, const T *host, size_t count, const T **dev) {
    void *d = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16) + 64;   // slack: traversal reads node pairs (id, id+1)
    const int rc_alloc = cached_alloc(sc->ctx, &d, bytes);
    if (rc_alloc) return rc_alloc;
    // (on the context's own stream: the legacy null stream is a queue of its own, see rsx_scene_create)
    static const bool null_stream = [] { const char *e = std::getenv("RSX_UPLOAD_NULL_STREAM"); return e && std::atoi(e) != 0; }();
    sc->allocs.push_back(d);
    if (null_stream) {
        HIP_TRY(hipMemset(d, 0, bytes));
        if (count) HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMemsetAsync(d, 0, bytes, sc->ctx->stream));
        if (count) HIP_TRY(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, sc->ctx->stream));
        HIP_TRY(hipStreamSynchronize(sc->ctx->stream));             // (the host array may go away after the call)
    }
    *dev = static_cast<const T *>(d);
    return RSX_OK;
}

// bit 0: every split of the tree lies inside its bounds; bit 1: no split is closer to zero than 2^-240 without being zero
// (what packet_space, dev_packet.hpp, needs to know about the numerators split - origin)
int splits_summary(const rsx_kdtree &kd) {
    int out = 3;
    for (int32_t k = 0; k < kd.n_nodes; ++k) {
        const rsx_kdnode &nd = kd.nodes[k];
        if (nd.type < 0 || nd.type > 2) continue;
        if (!(nd.u.split >= kd.lower[nd.type] && nd.u.split <= kd.upper[nd.type])) out &= ~1;
        if (nd.u.split != 0.0 && std::fabs(nd.u.split) < 0x1p-240) out &= ~2;
    }
    return out;
}

int tree_depth(const rsx_kdtree &kd) {
    // deepest chain of branch nodes = stack levels a traversal can need; iterative over the pre-order layout
    if (kd.n_nodes <= 0) return 0;
    std::vector<int32_t> depth((size_t)kd.n_nodes, 0);
    int best = 0;
    for (int32_t i = 0; i < kd.n_nodes; ++i) {
        const rsx_kdnode &nd = kd.nodes[i];
        if (nd.type >= 0) {
            const int d = depth[i] + 1;
            if (i + 1 < kd.n_nodes) depth[i + 1] = d;
            if (nd.count > 0 && nd.count < kd.n_nodes) depth[nd.count] = d;
            if (d > best) best = d;
        }
    }
    return best;
}

int validate_tree(const rsx_kdtree &kd, int32_t n_ids, const char *what) {
    if (kd.n_nodes < 1 || !kd.nodes) return rsx_fail(RSX_EINVAL, "%s: empty KD-tree", what);
    for (int32_t i = 0; i < kd.n_nodes; ++i) {
        const rsx_kdnode &nd = kd.nodes[i];
        if (nd.type >= 0) {
            if (nd.type > 2 || nd.count <= i || nd.count >= kd.n_nodes || i + 1 >= kd.n_nodes)
                return rsx_fail(RSX_EINVAL, "%s: malformed branch node %d", what, i);
        } else {
            if (nd.count < 0 || nd.u.leaf.first_item < 0 || (int64_t)nd.u.leaf.first_item + nd.count > kd.n_items)
                return rsx_fail(RSX_EINVAL, "%s: malformed leaf node %d", what, i);
            for (int32_t k = 0; k < nd.count; ++k) {
                const int32_t id = kd.items[nd.u.leaf.first_item + k];
                if (id < 0 || id >= n_ids) return rsx_fail(RSX_EINVAL, "%s: item id %d out of range", what, id);
            }
        }
    }
    return RSX_OK;
}

}  // namespace

extern "C" void rsx_scene_free(rsx_scene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->ctx->device);
    // nothing of the scene is freed while a pass may still read it: the private lanes first, then the context stream
    for (TraceLane &ln : scene->ctx->lanes) if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    (void)hipStreamSynchronize(scene->ctx->stream);
    cached_release(scene->ctx, scene->rel);
    cached_release(scene->ctx, scene->rel_info);
    cached_release(scene->ctx, scene->rel_jobs_dev);
    for (void *p : scene->allocs) cached_release(scene->ctx, p);
    delete scene;
}

extern "C" int rsx_scene_create(rsx_ctx *ctx, const rsx_scene_desc *desc, rsx_scene **out) {
    if (!ctx || !desc || !out) return rsx_fail(RSX_EINVAL, "rsx_scene_create: null argument");
    if (desc->n_world < 0 || desc->n_world > desc->n_primitives) return rsx_fail(RSX_EINVAL, "n_world out of range");
    for (int32_t i = 0; i < desc->n_primitives; ++i) {
        const rsx_primitive &p = desc->primitives[i];
        if (p.type == RSX_PRIM_UNION || p.type == RSX_PRIM_INTERSECT || p.type == RSX_PRIM_SUBTRACT) {
            if (p.child_a <= i || p.child_a >= desc->n_primitives || p.child_b <= i || p.child_b >= desc->n_primitives)
                return rsx_fail(RSX_EINVAL, "primitive %d: CSG operands must follow their node in the primitive table", i);
        }
        if (p.type == RSX_PRIM_MESH && (p.mesh < 0 || p.mesh >= desc->n_meshes)) return rsx_fail(RSX_EINVAL, "primitive %d: bad mesh index", i);
        if (p.type < 0 || p.type > RSX_PRIM_NULL) return rsx_fail(RSX_EINVAL, "primitive %d: unknown type %d", i, p.type);
    }
    int rc = validate_tree(desc->world_kd, desc->n_world, "world tree");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    rsx_scene *sc = new (std::nothrow) rsx_scene();
    if (!sc) return rsx_fail(RSX_ENOMEM, "out of host memory");
    sc->ctx = ctx;
    sc->n_world = desc->n_world;
    sc->has_csg = false;
    DScene &d = sc->d;
    std::memset(&d, 0, sizeof(d));
    // CSG bookkeeping: per operand tree assign per-lane state slots, parent links and nesting depth
    std::vector<CsgInfo> info((size_t)desc->n_primitives, CsgInfo{-1, 0, 0, -1});
    int32_t max_slots = 0;                                  // nodes of the biggest operand tree
    for (int32_t top = 0; top < desc->n_primitives; ++top) {
        const int tt = desc->primitives[top].type;
        const bool csg_node = tt == RSX_PRIM_UNION || tt == RSX_PRIM_INTERSECT || tt == RSX_PRIM_SUBTRACT;
        if (!csg_node || info[(size_t)top].top >= 0) continue;       // operands were claimed by their top node already
        sc->has_csg = true;
        int32_t slots = 0;
        std::vector<std::pair<int32_t, int>> todo{{top, 0}};
        info[(size_t)top].top = top;
        while (!todo.empty()) {
            const auto [i, depth] = todo.back();
            todo.pop_back();
            info[(size_t)i].slot = slots++;
            const rsx_primitive &q = desc->primitives[i];
            const bool inner = q.type == RSX_PRIM_UNION || q.type == RSX_PRIM_INTERSECT || q.type == RSX_PRIM_SUBTRACT;
            if (!inner) continue;
            if (depth >= CSG_STACK_MAX) { delete sc; return rsx_fail(RSX_EUNSUPPORTED, "primitive %d: CSG nesting deeper than %d levels", top, CSG_STACK_MAX); }
            for (int side = 0; side < 2; ++side) {
                const int32_t c = side ? q.child_b : q.child_a;
                if (info[(size_t)c].top >= 0) { delete sc; return rsx_fail(RSX_EINVAL, "primitive %d is an operand of two CSG nodes", c); }
                info[(size_t)c] = CsgInfo{i, 0, side, top};
                todo.push_back({c, depth + 1});
            }
        }
        max_slots = std::max(max_slots, slots);
    }
    // flattened operand trees for the state-free first-hit evaluator (csg_fast_hit): analytic leaves only
    std::vector<CsgFast> fast;
    bool any_fast = false;
    int fast_levels = 0;                                    // LDS levels csg_fast_hit needs: two root slots per leaf of the biggest tree
    if (sc->has_csg) {
        fast.assign((size_t)desc->n_primitives, CsgFast{});
        for (int32_t top = 0; top < desc->n_primitives; ++top) {
            if (info[(size_t)top].top != top) continue;
            CsgFast f{};
            bool ok = true;
            std::vector<int32_t> chain;
            std::function<void(int32_t, int)> visit = [&](int32_t i, int parity) {
                if (!ok) return;
                const rsx_primitive &q = desc->primitives[i];
                const bool inner = q.type == RSX_PRIM_UNION || q.type == RSX_PRIM_INTERSECT || q.type == RSX_PRIM_SUBTRACT;
                if (i != top) chain.push_back(i);
                if (!inner) {
                    const bool analytic = q.type == RSX_PRIM_SPHERE || q.type == RSX_PRIM_BOX || q.type == RSX_PRIM_CYLINDER;
                    if (!analytic || f.n_leaves >= CSGF_MAX_LEAVES || (int)chain.size() > CSGF_MAX_CHAIN) ok = false;
                    else {
                        const int k = f.n_leaves++;
                        f.leaf[k] = i; f.parity[k] = parity & 1; f.chain_len[k] = (int32_t)chain.size();
                        for (size_t j = 0; j < chain.size(); ++j) f.chain[k][j] = chain[j];
                        f.ops[f.n_ops++] = (int8_t)k;
                    }
                } else {
                    const int a_lo = f.n_leaves;
                    visit(q.child_a, parity);
                    const int a_hi = f.n_leaves;
                    visit(q.child_b, parity + (q.type == RSX_PRIM_SUBTRACT ? 1 : 0));
                    if (ok) f.ops[f.n_ops++] = q.type == RSX_PRIM_UNION ? (int8_t)-1 : q.type == RSX_PRIM_INTERSECT ? (int8_t)-2 : (int8_t)-3;
                    // CSGPrimitive.hit at EVERY node (csg.pyx:146-150): operand a of an Intersect / Subtract without a hit -> no hit, operand b
                    // is not looked at. The leaves of b remember a's leaf range (the innermost such node's)
                    if (ok && q.type != RSX_PRIM_UNION)
                        for (int k = a_hi; k < f.n_leaves; ++k) if (f.guard_hi[k] == 0) { f.guard_lo[k] = (int8_t)a_lo; f.guard_hi[k] = (int8_t)a_hi; }
                }
                if (i != top) chain.pop_back();
            };
            {   // operand a of the top node first: an Intersect / Subtract whose operand a has no root at all has none either (csg.pyx:148-150)
                const rsx_primitive &tq = desc->primitives[top];
                f.top_type = tq.type;
                visit(tq.child_a, 0);
                f.top_a_leaves = f.n_leaves;
                visit(tq.child_b, tq.type == RSX_PRIM_SUBTRACT ? 1 : 0);
                if (ok) f.ops[f.n_ops++] = tq.type == RSX_PRIM_UNION ? (int8_t)-1 : tq.type == RSX_PRIM_INTERSECT ? (int8_t)-2 : (int8_t)-3;
            }
            if (ok && f.n_leaves > 0) {                     // the postfix program run once per combination of leaf bits
                for (uint32_t mask = 0; mask < (1u << f.n_leaves); ++mask) {
                    uint32_t stack = 0;
                    int sp = 0;
                    for (int o = 0; o < f.n_ops; ++o) {
                        const int op = f.ops[o];
                        if (op >= 0) { stack |= ((mask >> op) & 1u) << sp; ++sp; }
                        else {
                            const uint32_t b = (stack >> (sp - 1)) & 1u, a = (stack >> (sp - 2)) & 1u;
                            const uint32_t res = op == -1 ? (a | b) : op == -2 ? (a & b) : (a & (b ^ 1u));
                            sp -= 2;
                            stack = (stack & ~(3u << sp)) | (res << sp);
                            ++sp;
                        }
                    }
                    if (stack & 1u) f.truth[mask >> 6] |= 1ULL << (mask & 63u);
                }
            }
            if (ok && f.n_leaves > 0) { fast[(size_t)top] = f; any_fast = true; fast_levels = std::max(fast_levels, 2 * f.n_leaves); }
        }
    }
#define UP(expr) do { rc = (expr); if (rc) { rsx_scene_free(sc); return rc; } } while (0)
    {
        // the device copy's `pad` carries what the host can say about a record once and for all. Bit 0 (PRIM_KEEPS_DIRECTION): to_local is
        // affine with the rotation part exactly the identity (+1.0 on the diagonal, +0.0 elsewhere: translate() and the default transform) —
        // a direction without a zero or non-finite component then comes out of Vector3D.transform bit for bit as it went in
        // ((1 * dx + 0 * dy) + 0 * dz = dx), and so do its reciprocals: csg_fast_hit_uniform keeps both (dev_csg.hpp)
        std::vector<rsx_primitive> dev_prims(desc->primitives, desc->primitives + desc->n_primitives);
        for (rsx_primitive &pr : dev_prims) {
            const double *m = pr.to_local;
            auto is = [](double v, double want) { return std::memcmp(&v, &want, sizeof(double)) == 0; };     // (bitwise: -0.0 is not +0.0)
            const bool keeps = is(m[0], 1.0) && is(m[1], 0.0) && is(m[2], 0.0) && is(m[4], 0.0) && is(m[5], 1.0) && is(m[6], 0.0) &&
                               is(m[8], 0.0) && is(m[9], 0.0) && is(m[10], 1.0) && is(m[12], 0.0) && is(m[13], 0.0) && is(m[14], 0.0) && is(m[15], 1.0);
            const bool affine = is(m[12], 0.0) && is(m[13], 0.0) && is(m[14], 0.0) && is(m[15], 1.0);   // bit 1: Point3D.transform's w is exactly 1 for a finite point
            pr.pad = (keeps ? 1 : 0) | (affine ? 2 : 0);
        }
        UP(upload(sc, dev_prims.data(), dev_prims.size(), &d.prims));
    }
    d.prims_uniform = d.prims;
    if (sc->has_csg) UP(upload(sc, info.data(), info.size(), &d.csg));
    d.csg_arena = nullptr; d.csg_arena_slots = 0; d.csg_arena_lanes = 0;
    if (max_slots > CSG_MAX_SLOTS) {
        // an operand tree with more nodes than the kernels' private state arrays hold: the stream merge keeps its node states in a
        // per-lane region of this arena instead (dev_csg.hpp: csg_slots), and plan() launches such a scene with at most
        // RSX_CSG_ARENA_WG_PER_CU workgroups per CU so that the arena covers every lane of the grid
        const size_t lanes = (size_t)ctx->n_cus * RSX_CSG_ARENA_WG_PER_CU * WG_THREADS;
        const size_t bytes = lanes * (size_t)max_slots * sizeof(NodeSt);
        void *arena = nullptr;
        if (malloc_or_flush(ctx, &arena, bytes) != hipSuccess) {
            (void)hipGetLastError();
            rsx_scene_free(sc);
            return rsx_fail(RSX_ENOMEM, "CSG tree with %d nodes: %zu bytes of stream-merge state could not be allocated", max_slots, bytes);
        }
        sc->allocs.push_back(arena);
        d.csg_arena = static_cast<NodeSt *>(arena); d.csg_arena_slots = max_slots; d.csg_arena_lanes = (int32_t)lanes;
    }
    if (any_fast) UP(upload(sc, fast.data(), fast.size(), &d.csgfast));
    d.csgfast_uniform = d.csgfast;
    // (the world nodes are uploaded below, after the wide primitives are known: wide-only leaves are tagged in the device copy)
    UP(upload(sc, desc->world_kd.items, (size_t)desc->world_kd.n_items, &d.witems));
    std::memcpy(d.wlower, desc->world_kd.lower, 24);
    std::memcpy(d.wupper, desc->world_kd.upper, 24);
    d.n_prims = desc->n_primitives;
    d.n_world = desc->n_world;
    d.n_meshes = desc->n_meshes;
    d.n_wnodes = desc->world_kd.n_nodes; d.n_witems = desc->world_kd.n_items;
    // (+ 3: the packet walk may push a node a second time for lanes that take its children in the other order — only where the
    // rays' common origin lies exactly on a split plane, at most once per axis along a root-to-leaf path; dev_packet.hpp)
    d.wdepth = tree_depth(desc->world_kd) + 1 + 3;
    d.mdepth = 1;
    {
        // the (up to eight) analytic world primitives that occur in the most world leaves (at least two): see DScene::wide
        std::vector<int32_t> leaves((size_t)std::max(1, desc->n_world), 0);
        for (int32_t n = 0; n < desc->world_kd.n_nodes; ++n) {
            const rsx_kdnode &nd = desc->world_kd.nodes[n];
            if (nd.type >= 0) continue;
            for (int32_t k = 0; k < nd.count; ++k) {
                const int32_t idx = desc->world_kd.items[nd.u.leaf.first_item + k];
                if (idx >= 0 && idx < desc->n_world) leaves[(size_t)idx]++;
            }
        }
        for (int32_t &w : d.wide) w = -1;
        if (!std::getenv("RSX_NO_WIDE")) {
            std::vector<int32_t> cand;
            for (int32_t i = 0; i < desc->n_world; ++i) {
                const int32_t t = desc->primitives[i].type;
                if ((t == RSX_PRIM_SPHERE || t == RSX_PRIM_BOX || t == RSX_PRIM_CYLINDER) && leaves[(size_t)i] >= 2) cand.push_back(i);
            }
            std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return leaves[(size_t)a] > leaves[(size_t)b]; });
            // A room of at most eight analytic primitives (a Cornell box): the ones that sit in a single leaf take the slots that are
            // left (never the first two: those are what the two-slot kernels answer for every ray), so that EVERY leaf is tagged and
            // every subtree can be culled against the nearest answer — a walk of such a world touches no primitive record at all.
            // (Cornell box: the ceiling light sits in one leaf; a quarter of the Lambert samples aim at it.)
            if (!std::getenv("RSX_NO_WIDE_ALL") && cand.size() >= 2 && desc->n_world <= 8) {
                bool all_analytic = true;
                for (int32_t i = 0; i < desc->n_world; ++i) {
                    const int32_t t = desc->primitives[i].type;
                    all_analytic = all_analytic && (t == RSX_PRIM_SPHERE || t == RSX_PRIM_BOX || t == RSX_PRIM_CYLINDER);
                }
                if (all_analytic) for (int32_t i = 0; i < desc->n_world; ++i) if (leaves[(size_t)i] == 1) cand.push_back(i);
            }
            // A CSG scene's path and query kernels answer RSX_CSG_WIDE analytic slots (registers): when all its analytic primitives
            // fit, the single-leaf ones come in from slot two on (a hole in slot one is fine: -1 matches nothing) — the prism scene's
            // floor, slit light and top light; with its four solids in the wide_csg slots every leaf of it is answered.
            if (!std::getenv("RSX_NO_WIDE_ALL") && sc->has_csg && !cand.empty()) {
                size_t n_analytic = 0;
                for (int32_t i = 0; i < desc->n_world; ++i) {
                    const int32_t t = desc->primitives[i].type;
                    if ((t == RSX_PRIM_SPHERE || t == RSX_PRIM_BOX || t == RSX_PRIM_CYLINDER) && leaves[(size_t)i] >= 1) ++n_analytic;
                }
                if (n_analytic <= (size_t)RSX_CSG_WIDE) {
                    while (cand.size() < 2) cand.push_back(-1);
                    for (int32_t i = 0; i < desc->n_world; ++i) {
                        const int32_t t = desc->primitives[i].type;
                        if ((t == RSX_PRIM_SPHERE || t == RSX_PRIM_BOX || t == RSX_PRIM_CYLINDER) && leaves[(size_t)i] == 1) cand.push_back(i);
                    }
                }
            }
            for (size_t k = 0; k < cand.size() && k < 8; ++k) d.wide[k] = cand[k];
        }
        d.wide_plain = 0;
        for (int k = 0; k < 8; ++k) {
            if (d.wide[k] < 0) continue;
            const rsx_primitive &wp = desc->primitives[d.wide[k]];
            static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            bool plain = wp.type == RSX_PRIM_BOX;
            for (int c = 0; plain && c < 12; ++c) plain = wp.to_local[c] == ident[c];      // (numerically: -0.0 entries leave x * 1 + y * 0 + z * 0 + 0 = x as well)
            if (plain) d.wide_plain |= 1 << k;
        }
        // cluster boxes of everything the packet kernel does not answer up front (DScene::pkt_clusters): the primitives' own (padded)
        // bounding boxes, grouped by two median cuts of their centres along the widest spread
        {
            d.pkt_clusters = -1; d.all_wide8 = 0; d.all_answered_csg = 0; d.scene_pad = 0;
            std::memset(d.cluster_lo, 0, sizeof(d.cluster_lo)); std::memset(d.cluster_hi, 0, sizeof(d.cluster_hi));
            std::memset(d.cluster_members, 0, sizeof(d.cluster_members)); std::memset(d.member_lo, 0, sizeof(d.member_lo)); std::memset(d.member_hi, 0, sizeof(d.member_hi));
            // (a world tree of a handful of nodes — one mesh, a floor and a sky — has no walk worth skipping: the boxes' tests cost the `flat`
            // workload 3 % for nothing)
            bool tame = !std::getenv("RSX_NO_PKT_CLUSTERS") && desc->world_kd.n_nodes >= 8;
            // the short cut's argument needs rounding errors of positions far below BOX_PADDING = 1e-9 (box.pyx:37 ...): coordinates below 1e4
            for (int k = 0; k < 3; ++k) tame = tame && std::isfinite(desc->world_kd.lower[k]) && std::isfinite(desc->world_kd.upper[k]) &&
                                              std::fabs(desc->world_kd.lower[k]) <= 1e4 && std::fabs(desc->world_kd.upper[k]) <= 1e4;
            {   // (the eight-slot kernels' form of the same argument: a world all of whose primitives are answered before the walk)
                bool coords = !std::getenv("RSX_NO_PKT_CLUSTERS");
                for (int k = 0; k < 3; ++k) coords = coords && std::isfinite(desc->world_kd.lower[k]) && std::isfinite(desc->world_kd.upper[k]) &&
                                                     std::fabs(desc->world_kd.lower[k]) <= 1e4 && std::fabs(desc->world_kd.upper[k]) <= 1e4;
                bool all = coords && !sc->has_csg && desc->n_world > 0;
                for (int32_t i = 0; all && i < desc->n_world; ++i) {
                    bool found = desc->primitives[i].type == RSX_PRIM_NULL;
                    for (int q = 0; q < 8; ++q) found = found || d.wide[q] == i;
                    all = found;
                }
                d.all_wide8 = all ? 1 : 0;
            }
            if (tame) {
                std::vector<int32_t> rest;
                for (int32_t i = 0; i < desc->n_world; ++i) if (i != d.wide[0] && i != d.wide[1] && desc->primitives[i].type != RSX_PRIM_NULL) rest.push_back(i);
                std::vector<std::vector<int32_t>> groups;
                if (!rest.empty()) groups.push_back(rest);
                auto centre = [&](int32_t i, int k) { return 0.5 * (desc->primitives[i].box_lower[k] + desc->primitives[i].box_upper[k]); };
                while (groups.size() < 4) {
                    int best = -1, axis = 0;
                    double spread = 0.0;
                    for (size_t g = 0; g < groups.size(); ++g) {
                        if (groups[g].size() < 2) continue;
                        for (int k = 0; k < 3; ++k) {
                            double lo = INFINITY, hi = -INFINITY;
                            for (int32_t i : groups[g]) { lo = std::min(lo, centre(i, k)); hi = std::max(hi, centre(i, k)); }
                            if (hi - lo > spread) { spread = hi - lo; best = (int)g; axis = k; }
                        }
                    }
                    if (best < 0) break;
                    std::vector<int32_t> all = groups[(size_t)best];
                    std::sort(all.begin(), all.end(), [&](int32_t a, int32_t b) { return centre(a, axis) < centre(b, axis); });
                    const size_t half = all.size() / 2;
                    groups[(size_t)best].assign(all.begin(), all.begin() + (long)half);
                    groups.emplace_back(all.begin() + (long)half, all.end());
                }
                d.pkt_clusters = (int32_t)groups.size();
                for (size_t g = 0; g < groups.size(); ++g)
                    for (int k = 0; k < 3; ++k) {
                        double lo = INFINITY, hi = -INFINITY;
                        for (int32_t i : groups[g]) { lo = std::min(lo, desc->primitives[i].box_lower[k]); hi = std::max(hi, desc->primitives[i].box_upper[k]); }
                        d.cluster_lo[g][k] = lo; d.cluster_hi[g][k] = hi;
                    }
                for (size_t g = 0; g < groups.size(); ++g) {
                    d.cluster_members[g] = groups[g].size() <= 4 ? (int32_t)groups[g].size() : 0;
                    for (int32_t m = 0; m < d.cluster_members[g]; ++m)
                        for (int k = 0; k < 3; ++k) {
                            d.member_lo[g][m][k] = desc->primitives[groups[g][(size_t)m]].box_lower[k];
                            d.member_hi[g][m][k] = desc->primitives[groups[g][(size_t)m]].box_upper[k];
                        }
                }
            }
        }
        for (int32_t &w : d.wide_csg) w = -1;
        if (any_fast && !std::getenv("RSX_NO_WIDE_CSG")) {
            std::vector<int32_t> cand;
            for (int32_t i = 0; i < desc->n_world; ++i) if (fast[(size_t)i].n_leaves > 0 && leaves[(size_t)i] >= 2) cand.push_back(i);
            std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return leaves[(size_t)a] > leaves[(size_t)b]; });
            // (as above: when every CSG solid with a flattened program fits the four slots, the single-leaf ones are answered in the
            // same round before the traversal — the subtree of a solid most scattered rays aim at, a prism of importance 9, can then
            // be culled like the others)
            if (!std::getenv("RSX_NO_WIDE_ALL")) {
                size_t n_programs = 0;
                for (int32_t i = 0; i < desc->n_world; ++i) if (fast[(size_t)i].n_leaves > 0 && leaves[(size_t)i] >= 1) ++n_programs;
                if (!cand.empty() && n_programs <= 4) for (int32_t i = 0; i < desc->n_world; ++i) if (fast[(size_t)i].n_leaves > 0 && leaves[(size_t)i] == 1) cand.push_back(i);
            }
            for (size_t k = 0; k < cand.size() && k < 4; ++k) d.wide_csg[k] = cand[k];
        }
        {   // every world primitive answered before the walk of the CSG fast forms? (world_trace_wave's short cut; the argument is DScene::all_wide8's)
            bool all = sc->has_csg && desc->n_world > 0 && d.wide_csg[0] >= 0 && !std::getenv("RSX_NO_PKT_CLUSTERS");
            for (int k = 0; k < 3; ++k) all = all && std::isfinite(desc->world_kd.lower[k]) && std::isfinite(desc->world_kd.upper[k]) &&
                                              std::fabs(desc->world_kd.lower[k]) <= 1e4 && std::fabs(desc->world_kd.upper[k]) <= 1e4;
            for (int32_t i = 0; all && i < desc->n_world; ++i) {
                bool found = false;
                for (int q = 0; q < RSX_CSG_WIDE; ++q) found = found || d.wide[q] == i;
                for (int q = 0; q < 4; ++q) found = found || d.wide_csg[q] == i;
                all = found;
            }
            d.all_answered_csg = all ? 1 : 0;
        }
        // Device copies of the world nodes. A leaf whose items are ALL wide primitives (most leaves of a scene with a floor and an
        // enclosing emitter; every leaf of a room of boxes) carries its whole item list in the node's spare word — bit 31, the item count
        // in bits 28..30, the wide slot of item j in bits 3j..3j+2, list order kept — so that visiting it costs no item loads and no
        // per-item rounds (world_trace_wave). Two copies: tagged for the first two slots (primary rays) and for all eight (scattered rays).
        for (int n_slots : {2, 8}) {
            // (a CSG scene has no eight-slot kernels: its second copy serves the kernels that answer the wide CSG primitives before
            // the traversal — two slots for the leaf tags, and cull bits that count those CSG primitives as answered too)
            const int tag_slots = (n_slots == 8 && sc->has_csg) ? RSX_CSG_WIDE : n_slots;
            const bool csg_answered = n_slots == 8 && sc->has_csg;
            std::vector<rsx_kdnode> wnodes(desc->world_kd.nodes, desc->world_kd.nodes + desc->world_kd.n_nodes);
            for (rsx_kdnode &nd : wnodes) {
                if (nd.type >= 0) continue;
                nd.u.leaf.pad = 0;
                if (std::getenv("RSX_NO_WIDE_LEAVES") || nd.count < 1 || nd.count > (tag_slots == 2 ? 2 : 6)) continue;
                uint32_t tag = 0x80000000u | ((uint32_t)nd.count << 28);
                bool all_wide = true;
                for (int32_t k = 0; k < nd.count; ++k) {
                    const int32_t idx = desc->world_kd.items[nd.u.leaf.first_item + k];
                    int slot = -1;
                    for (int q = 0; q < tag_slots; ++q) if (d.wide[q] >= 0 && idx == d.wide[q]) slot = q;
                    if (slot < 0) all_wide = false; else tag |= (uint32_t)slot << (3 * k);
                }
                if (all_wide) nd.u.leaf.pad = (int32_t)tag;
            }
            // inner nodes: bits 2 / 3 of the axis word say that every item below the lower / upper child is one of these wide
            // primitives (world_step's cull; children come after their parent in the pre-order array, so one backward sweep does it)
            {
                std::vector<char> only_wide(wnodes.size(), 0);
                for (size_t n = wnodes.size(); n-- > 0;) {
                    rsx_kdnode &nd = wnodes[n];
                    if (nd.type < 0) {
                        bool all = true;
                        for (int32_t k = 0; k < nd.count && all; ++k) {
                            const int32_t idx = desc->world_kd.items[nd.u.leaf.first_item + k];
                            bool is_wide = false;
                            for (int q = 0; q < tag_slots; ++q) if (d.wide[q] >= 0 && idx == d.wide[q]) is_wide = true;
                            if (csg_answered) for (int32_t w : d.wide_csg) if (w >= 0 && idx == w) is_wide = true;
                            all = is_wide;
                        }
                        only_wide[n] = all;
                    } else {
                        const size_t lower = n + 1, upper = (size_t)nd.count;
                        const bool lo = lower < wnodes.size() && only_wide[lower], up = upper < wnodes.size() && only_wide[upper];
                        only_wide[n] = lo && up;
                        if (!std::getenv("RSX_NO_WORLD_CULL")) nd.type |= (lo ? 4 : 0) | (up ? 8 : 0);
                    }
                }
            }
            if (n_slots == 2) UP(upload(sc, wnodes.data(), wnodes.size(), &d.wnodes));
            else UP(upload(sc, wnodes.data(), wnodes.size(), &d.wnodes_scatter));
        }
    }
    std::vector<DMesh> meshes((size_t)desc->n_meshes);
    for (int32_t i = 0; i < desc->n_meshes; ++i) {
        const rsx_meshdata &m = desc->meshes[i];
        rc = validate_tree(m.kd, m.n_triangles, "mesh tree");
        if (rc) { rsx_scene_free(sc); return rc; }
        DMesh &dm = meshes[(size_t)i];
        std::memset(&dm, 0, sizeof(dm));
        // pre-gather: 48-byte triangle records (vertices + face normal) -> no index indirection on the device
        std::vector<float4> tris((size_t)m.n_triangles * 3);
        std::vector<int32_t> nidx;
        if (m.vertex_normals && m.tri_stride >= 6) nidx.resize((size_t)m.n_triangles * 3);
        for (int32_t t = 0; t < m.n_triangles; ++t) {
            const int32_t *tr = m.triangles + (size_t)t * m.tri_stride;
            for (int k = 0; k < 3; ++k)
                if (tr[k] < 0 || tr[k] >= m.n_vertices) { rsx_scene_free(sc); return rsx_fail(RSX_EINVAL, "mesh %d triangle %d: vertex index out of range", i, t); }
            const float *a = m.vertices + 3 * (size_t)tr[0], *b = m.vertices + 3 * (size_t)tr[1], *c = m.vertices + 3 * (size_t)tr[2];
            const float *fn = m.face_normals + 3 * (size_t)t;
            tris[3 * (size_t)t] = make_float4(a[0], a[1], a[2], b[0]);
            tris[3 * (size_t)t + 1] = make_float4(b[1], b[2], c[0], c[1]);
            tris[3 * (size_t)t + 2] = make_float4(c[2], fn[0], fn[1], fn[2]);
            if (!nidx.empty()) for (int k = 0; k < 3; ++k) {
                if (tr[3 + k] < 0 || tr[3 + k] >= m.n_normals) { rsx_scene_free(sc); return rsx_fail(RSX_EINVAL, "mesh %d triangle %d: normal index out of range", i, t); }
                nidx[3 * (size_t)t + k] = tr[3 + k];
            }
        }
        UP(upload(sc, tris.data(), tris.size(), &dm.tris));
        {
            std::vector<float4> leaf((size_t)m.kd.n_items * 4);
            for (int32_t k = 0; k < m.kd.n_items; ++k) {
                const int32_t t = m.kd.items[k];
                leaf[4 * (size_t)k] = tris[3 * (size_t)t]; leaf[4 * (size_t)k + 1] = tris[3 * (size_t)t + 1]; leaf[4 * (size_t)k + 2] = tris[3 * (size_t)t + 2];
                float4 idrec = make_float4(0.f, 0.f, 0.f, 0.f);
                std::memcpy(&idrec.x, &t, 4);
                leaf[4 * (size_t)k + 3] = idrec;
            }
            UP(upload(sc, leaf.data(), leaf.size(), &dm.leaf));
        }
        UP(upload(sc, m.kd.nodes, (size_t)m.kd.n_nodes, &dm.nodes));
        UP(upload(sc, m.kd.items, (size_t)m.kd.n_items, &dm.items));
        if (!nidx.empty()) {
            UP(upload(sc, m.vertex_normals, (size_t)m.n_normals * 3, &dm.vnormals));
            UP(upload(sc, nidx.data(), nidx.size(), &dm.nidx));
        }
        std::memcpy(dm.lower, m.kd.lower, 24);
        std::memcpy(dm.upper, m.kd.upper, 24);
        dm.smoothing = m.smoothing; dm.closed = m.closed; dm.n_tris = m.n_triangles;
        // the packet walk tests the range of a quotient's numerator once per walk from the tree's bounds (packet_space, dev_packet.hpp):
        // sound while every split lies inside them — true of any tree the builder makes; a tree read from a file is checked, not trusted
        dm.splits_bounded = splits_summary(m.kd);
        d.mdepth = std::max(d.mdepth, tree_depth(m.kd) + 1 + 3);
    }
    UP(upload(sc, meshes.data(), meshes.size(), &d.meshes));
    d.rel = nullptr; d.rel_info = nullptr;
    d.wsplits_bounded = splits_summary(desc->world_kd); d.csg_fast_rows = any_fast ? fast_levels : 0;
    for (int32_t i = 0; i < desc->n_world; ++i) {
        const rsx_primitive &p = desc->primitives[i];
        if (p.type != RSX_PRIM_MESH) continue;
        RelJobHost job;
        job.prim = i; job.mesh = p.mesh; job.offset = sc->rel_records; job.n_items = desc->meshes[p.mesh].kd.n_items;
        if (job.n_items <= 0) continue;
        sc->rel_jobs.push_back(job);
        sc->rel_records += job.n_items;
    }
    if (any_fast) d.mdepth = std::max(d.mdepth, fast_levels);           // csg_fast_hit keeps the leaf roots in the mesh-stack LDS levels
    d.wlds = std::min(d.wdepth, RSX_WORLD_LDS_LEVELS);
    d.mlds = std::min(d.mdepth, std::max(RSX_MESH_LDS_LEVELS, fast_levels));
    if (const char *env = std::getenv("RSX_WORLD_LDS")) d.wlds = std::max(0, std::min(d.wdepth, std::atoi(env)));   // tuning aids
    if (const char *env = std::getenv("RSX_MESH_LDS")) d.mlds = std::max(0, std::min(d.mdepth, std::atoi(env)));
#undef UP
    *out = sc;
    return RSX_OK;
}

// -- launches ---------------------------------------------------------------------------------------
namespace {

struct Launch {
    dim3 grid;
    size_t lds;
};

// persistent grid: enough workgroups to fill every CU at the occupancy the LDS stacks allow
int plan(rsx_scene *sc, long long work_items, TraceLane &lane, Launch &l, int wg_per_cu_cap = RSX_MAX_WG_PER_CU) {
    if (sc->d.csg_arena) wg_per_cu_cap = std::min(wg_per_cu_cap, RSX_CSG_ARENA_WG_PER_CU);      // the arena holds the states of that many lanes
    const int lds_levels = sc->d.wlds + sc->d.mlds;
    l.lds = (size_t)WG_WAVES * ((size_t)lds_levels * WAVE * 12 + STAGE_BYTES);
    if (l.lds > 160 * 1024) return rsx_fail(RSX_EUNSUPPORTED, "traversal stack does not fit LDS (%d levels)", lds_levels);
    int per_cu = (int)std::min<size_t>((size_t)wg_per_cu_cap, (160 * 1024) / std::max<size_t>(l.lds, 1));
    if (per_cu < 1) per_cu = 1;
    long long wgs = (long long)sc->ctx->n_cus * per_cu;
    const long long needed = (work_items + WG_THREADS - 1) / WG_THREADS;
    if (wgs > needed) wgs = needed;
    if (wgs < 1) wgs = 1;
    l.grid = dim3((unsigned)wgs);
    // global spill regions: one per wave of the largest grid a lane launches (lanes run concurrently, so each has its own)
    const size_t need = (size_t)sc->ctx->n_cus * RSX_MAX_WG_PER_CU * WG_WAVES * spill_wave_bytes(sc->d.wdepth, sc->d.wlds, sc->d.mdepth, sc->d.mlds);
    if (need > lane.spill_bytes) {
        if (lane.spill) { HIP_TRY(hipStreamSynchronize(lane.stream)); HIP_TRY(hipFree(lane.spill)); lane.spill = nullptr; lane.spill_bytes = 0; }
        HIP_TRY(malloc_or_flush(sc->ctx, &lane.spill, need));
        lane.spill_bytes = need;
    }
    sc->d.spill = static_cast<char *>(lane.spill);
    return RSX_OK;
}

int reset_ticket(TraceLane &lane) {
    HIP_TRY(hipMemsetAsync(lane.ticket, 0, 2 * 9 * 16 * sizeof(unsigned long long), lane.stream));
    lane.ticket_armed = false;          // whoever launches next dirties it again
    return RSX_OK;
}

// Device workspace of one synchronous batch query (hit / roots / contains): a single grow-only allocation in the ctx, carved into
// 256-byte aligned pieces — a per-call hipMalloc / hipFree per array cost more than the kernel for small batches (World.hit(ray)
// is a batch of one). Safe to reuse: every query call synchronises the stream before it returns.
struct Carver {
    char *at = nullptr;
    size_t left = 0;
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    template <typename T> T *take(size_t bytes) {
        if (bytes == 0) return nullptr;
        T *out = reinterpret_cast<T *>(at);
        at += padded(bytes); left -= padded(bytes);
        return out;
    }
};

int query_workspace(rsx_ctx *ctx, std::initializer_list<size_t> sizes, Carver &c) {
    size_t total = 256;
    for (size_t b : sizes) total += Carver::padded(b);
    void *base = nullptr;
    int rc = pool_get(ctx, POOL_QUERY, total, &base);
    if (rc) return rc;
    c.at = static_cast<char *>(base);
    c.left = total;
    return RSX_OK;
}

// One synchronous query = inputs up, kernel, outputs back. Workspaces up to 256 MiB go both ways through the ctx's pinned mirror of
// the workspace: one transfer each way instead of one per array (World.hit(ray) is a batch of one), and the caller's pageable arrays
// are never a DMA target. The second half matters to the host-callback render path (optical/hybrid.py), whose parent process answers
// the ray waves of forked material workers: pageable transfers into heap pages still shared copy-on-write with 16 busy children
// cost 3 ms per call instead of 0.2 (tools/r5_hybrid_workers.py); through the mirror the call no longer depends on who shares the pages.
struct QueryIO {
    rsx_ctx *ctx;
    char *ws_begin, *ws_end;
    bool staged;
    std::vector<std::tuple<void *, const void *, size_t>> in;       // (device, host, bytes)
    std::vector<std::tuple<void *, const void *, size_t>> out;      // (host, device, bytes)
    char *lowest_out = nullptr;

    int begin(rsx_ctx *c, void *first, char *end) {
        ctx = c; ws_begin = static_cast<char *>(first); ws_end = end;
        const size_t need = (size_t)(ws_end - ws_begin);
        staged = need <= ((size_t)256 << 20);
        if (staged && ctx->staging_bytes < need) {
            size_t bytes = std::max(ctx->staging_bytes, (size_t)1 << 20);
            while (bytes < need) bytes *= 2;
            if (ctx->staging) HIP_TRY(hipHostFree(ctx->staging));
            ctx->staging = nullptr; ctx->staging_bytes = 0;
            if (hipHostMalloc(&ctx->staging, bytes, hipHostMallocDefault) != hipSuccess) {      // (no pinned memory left: pageable transfers)
                (void)hipGetLastError();
                ctx->staging = nullptr; staged = false;
                return RSX_OK;
            }
            ctx->staging_bytes = bytes;
        }
        return RSX_OK;
    }
    void input(void *dev, const void *host, size_t bytes) { if (dev && bytes) in.emplace_back(dev, host, bytes); }
    void output(void *host, const void *dev, size_t bytes) {
        if (!host || !dev || !bytes) return;
        out.emplace_back(host, dev, bytes);
        char *d = const_cast<char *>(static_cast<const char *>(dev));
        if (!lowest_out || d < lowest_out) lowest_out = d;
    }
    int upload() {
        if (!staged) {
            for (auto &e : in) HIP_TRY(hipMemcpyAsync(std::get<0>(e), std::get<1>(e), std::get<2>(e), hipMemcpyHostToDevice, ctx->stream));
            return RSX_OK;
        }
        char *h = static_cast<char *>(ctx->staging);
        size_t hi = 0;
        for (auto &e : in) {
            const size_t off = (size_t)(static_cast<char *>(std::get<0>(e)) - ws_begin);
            std::memcpy(h + off, std::get<1>(e), std::get<2>(e));
            hi = std::max(hi, off + std::get<2>(e));
        }
        if (hi) HIP_TRY(hipMemcpyAsync(ws_begin, h, hi, hipMemcpyHostToDevice, ctx->stream));
        return RSX_OK;
    }
    int download() {                                                  // synchronises the stream
        if (!staged) {
            for (auto &e : out) HIP_TRY(hipMemcpyAsync(std::get<0>(e), std::get<1>(e), std::get<2>(e), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            return RSX_OK;
        }
        char *h = static_cast<char *>(ctx->staging);
        if (lowest_out) HIP_TRY(hipMemcpyAsync(h + (lowest_out - ws_begin), lowest_out, (size_t)(ws_end - lowest_out), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (auto &e : out) std::memcpy(std::get<0>(e), h + (static_cast<const char *>(std::get<1>(e)) - ws_begin), std::get<2>(e));
        return RSX_OK;
    }
};

}  // namespace

extern "C" int rsx_hit_batch_dev(rsx_scene *scene, int64_t n, const double *origin, const double *direction, const double *max_distance,
                                 int32_t *prim, double *t, uint8_t *exiting, int32_t *tri, float *uvw, double *geom) {
    if (!scene || n < 0 || !origin || !direction || !max_distance || !prim) return rsx_fail(RSX_EINVAL, "rsx_hit_batch_dev: bad arguments");
    if (n == 0) return RSX_OK;
    rsx_ctx *ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    Launch l;
    int rc = plan(scene, n, ctx->main, l);
    if (rc) return rc;
    HIP_TRY(hipFuncSetAttribute(scene->has_csg ? reinterpret_cast<const void *>(k_hit_batch<true>) : reinterpret_cast<const void *>(k_hit_batch<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
    if ((rc = reset_ticket(ctx->main))) return rc;
    HitOut out = {prim, t, exiting, tri, uvw, geom};
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (scene->has_csg && scene->d.csgfast) {               // fast pass, then the stream merge for the rays marked HIT_REDO
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hit_batch<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hit_batch<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
        hipLaunchKernelGGL((k_hit_batch<true, 1>), l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, (long long)n, origin, direction, max_distance, out, ctx->main.ticket);
        if ((rc = reset_ticket(ctx->main))) return rc;
        hipLaunchKernelGGL((k_hit_batch<true, 2>), l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, (long long)n, origin, direction, max_distance, out, ctx->main.ticket);
    } else if (scene->has_csg) hipLaunchKernelGGL(k_hit_batch<true>, l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, (long long)n, origin, direction, max_distance, out, ctx->main.ticket);
    else hipLaunchKernelGGL(k_hit_batch<false>, l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, (long long)n, origin, direction, max_distance, out, ctx->main.ticket);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    return RSX_OK;
}

extern "C" int rsx_hit_batch(rsx_scene *scene, int64_t n, const double *origin, const double *direction, const double *max_distance,
                             int32_t *prim, double *t, uint8_t *exiting, int32_t *tri, float *uvw, double *geom) {
    if (!scene || n < 0 || !origin || !direction || !max_distance || !prim) return rsx_fail(RSX_EINVAL, "rsx_hit_batch: bad arguments");
    if (n == 0) return RSX_OK;
    rsx_ctx *ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    Carver c;
    int rc = query_workspace(ctx, {N * 24, N * 24, N * 8, N * 4, t ? N * 8 : 0, exiting ? N : 0, tri ? N * 4 : 0, uvw ? N * 12 : 0, geom ? N * 96 : 0}, c);
    if (rc) return rc;
    double *d_o = c.take<double>(N * 24), *d_d = c.take<double>(N * 24), *d_m = c.take<double>(N * 8);
    int32_t *d_prim = c.take<int32_t>(N * 4);
    double *d_t = c.take<double>(t ? N * 8 : 0);
    uint8_t *d_ex = c.take<uint8_t>(exiting ? N : 0);
    int32_t *d_tri = c.take<int32_t>(tri ? N * 4 : 0);
    float *d_uvw = c.take<float>(uvw ? N * 12 : 0);
    double *d_geom = c.take<double>(geom ? N * 96 : 0);
    QueryIO io;
    if ((rc = io.begin(ctx, d_o, c.at))) return rc;
    io.input(d_o, origin, N * 24); io.input(d_d, direction, N * 24); io.input(d_m, max_distance, N * 8);
    if ((rc = io.upload())) return rc;
    rc = rsx_hit_batch_dev(scene, n, d_o, d_d, d_m, d_prim, d_t, d_ex, d_tri, d_uvw, d_geom);
    if (rc) return rc;
    io.output(prim, d_prim, N * 4); io.output(t, d_t, N * 8); io.output(exiting, d_ex, N); io.output(tri, d_tri, N * 4);
    io.output(uvw, d_uvw, N * 12); io.output(geom, d_geom, N * 96);
    return io.download();
}

extern "C" int rsx_roots_batch(rsx_scene *scene, int32_t primitive, int64_t n, const double *origin, const double *direction,
                               const double *max_distance, int32_t max_roots, int32_t *counts, double *t, uint8_t *exiting,
                               double *geometry, int32_t *triangle, float *uvw) {
    if (!scene || n < 0 || !origin || !direction || !max_distance || !counts || !t || !exiting || max_roots < 1)
        return rsx_fail(RSX_EINVAL, "rsx_roots_batch: bad arguments");
    if (primitive < 0 || primitive >= scene->d.n_prims) return rsx_fail(RSX_EINVAL, "rsx_roots_batch: primitive index out of range");
    if (n == 0) return RSX_OK;
    rsx_ctx *ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n, R = (size_t)max_roots;
    Carver c;
    int rc = query_workspace(ctx, {N * 24, N * 24, N * 8, N * 4, N * R * 8, N * R, geometry ? N * R * 96 : 0, triangle ? N * R * 4 : 0, uvw ? N * R * 12 : 0}, c);
    if (rc) return rc;
    double *d_o = c.take<double>(N * 24), *d_d = c.take<double>(N * 24), *d_m = c.take<double>(N * 8);
    int32_t *d_c = c.take<int32_t>(N * 4);
    double *d_t = c.take<double>(N * R * 8);
    uint8_t *d_ex = c.take<uint8_t>(N * R);
    double *d_g = c.take<double>(geometry ? N * R * 96 : 0);
    int32_t *d_tri = c.take<int32_t>(triangle ? N * R * 4 : 0);
    float *d_uvw = c.take<float>(uvw ? N * R * 12 : 0);
    if (d_g) HIP_TRY(hipMemsetAsync(d_g, 0, N * R * 96, ctx->stream));
    if (d_tri) HIP_TRY(hipMemsetAsync(d_tri, 0xff, N * R * 4, ctx->stream));
    if (d_uvw) HIP_TRY(hipMemsetAsync(d_uvw, 0, N * R * 12, ctx->stream));
    QueryIO io;
    if ((rc = io.begin(ctx, d_o, c.at))) return rc;
    io.input(d_o, origin, N * 24); io.input(d_d, direction, N * 24); io.input(d_m, max_distance, N * 8);
    if ((rc = io.upload())) return rc;
    HIP_TRY(hipMemsetAsync(d_t, 0, N * R * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_ex, 0, N * R, ctx->stream));
    Launch l;
    if ((rc = plan(scene, n, ctx->main, l))) return rc;
    HIP_TRY(hipFuncSetAttribute(scene->has_csg ? reinterpret_cast<const void *>(k_roots<true>) : reinterpret_cast<const void *>(k_roots<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
    if ((rc = reset_ticket(ctx->main))) return rc;
    if (scene->has_csg) hipLaunchKernelGGL(k_roots<true>, l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, primitive, (long long)n, d_o, d_d, d_m,
                                           max_roots, d_c, d_t, d_ex, d_g, d_tri, d_uvw, ctx->main.ticket);
    else hipLaunchKernelGGL(k_roots<false>, l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, primitive, (long long)n, d_o, d_d, d_m,
                            max_roots, d_c, d_t, d_ex, d_g, d_tri, d_uvw, ctx->main.ticket);
    HIP_TRY(hipGetLastError());
    io.output(counts, d_c, N * 4); io.output(t, d_t, N * R * 8); io.output(exiting, d_ex, N * R);
    io.output(geometry, d_g, N * R * 96); io.output(triangle, d_tri, N * R * 4); io.output(uvw, d_uvw, N * R * 12);
    return io.download();
}

extern "C" int rsx_contains_batch(rsx_scene *scene, int64_t n, const double *points, uint8_t *inside) {
    if (!scene || n < 0 || !points || !inside) return rsx_fail(RSX_EINVAL, "rsx_contains_batch: bad arguments");
    if (n == 0) return RSX_OK;
    rsx_ctx *ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n, W = (size_t)std::max(1, scene->d.n_world);
    Carver c;
    int rc = query_workspace(ctx, {N * 24, N * W}, c);
    if (rc) return rc;
    double *d_p = c.take<double>(N * 24);
    uint8_t *d_in = c.take<uint8_t>(N * W);
    QueryIO io;
    if ((rc = io.begin(ctx, d_p, c.at))) return rc;
    io.input(d_p, points, N * 24);
    if ((rc = io.upload())) return rc;
    Launch l;
    if ((rc = plan(scene, n, ctx->main, l))) return rc;
    HIP_TRY(hipFuncSetAttribute(scene->has_csg ? reinterpret_cast<const void *>(k_contains<true>) : reinterpret_cast<const void *>(k_contains<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
    if ((rc = reset_ticket(ctx->main))) return rc;
    if (scene->has_csg) hipLaunchKernelGGL(k_contains<true>, l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, (long long)n, d_p, d_in, ctx->main.ticket);
    else hipLaunchKernelGGL(k_contains<false>, l.grid, dim3(WG_THREADS), l.lds, ctx->stream, scene->d, (long long)n, d_p, d_in, ctx->main.ticket);
    HIP_TRY(hipGetLastError());
    io.output(inside, d_in, N * (size_t)scene->d.n_world);
    return io.download();
}

namespace {

// Camera-relative leaf records for a packet pass from `cam` (dev_packet.hpp): (re)made when the camera changed, on `stream`, which the
// pass itself is queued on next. 48 bytes per (mesh instance, leaf item); scenes that would need more than RSX_REL_MAX_GB (default 64)
// go without — the walk then takes the vertices and translates per ray.
int ensure_camera_relative(rsx_scene *scene, const rsx_camera &cam, hipStream_t stream) {
    static const bool enabled = [] { const char *e = std::getenv("RSX_CAMERA_RELATIVE"); return !e || std::atoi(e) != 0; }();
    static const double max_gb = [] { const char *e = std::getenv("RSX_REL_MAX_GB"); return e ? std::atof(e) : 64.0; }();
    if (!enabled || scene->rel_jobs.empty() || scene->rel_refused) { scene->d.rel = nullptr; scene->d.rel_info = nullptr; return RSX_OK; }
    rsx_ctx *ctx = scene->ctx;
    if (!scene->rel) {
        const size_t bytes = (size_t)scene->rel_records * 48;
        if ((double)bytes > max_gb * 1073741824.0 || scene->rel_jobs.size() > 65535) { scene->rel_refused = true; return RSX_OK; }
        if (cached_alloc(scene->ctx, &scene->rel, bytes) != RSX_OK) { (void)hipGetLastError(); scene->rel = nullptr; scene->rel_refused = true; return RSX_OK; }
        int rc_a;
        if ((rc_a = cached_alloc(scene->ctx, &scene->rel_info, (size_t)scene->d.n_prims * sizeof(RelInfo)))) return rc_a;
        if ((rc_a = cached_alloc(scene->ctx, &scene->rel_jobs_dev, scene->rel_jobs.size() * sizeof(RelJob)))) return rc_a;
        static_assert(sizeof(RelJobHost) == sizeof(RelJob), "same layout");
        HIP_TRY(hipMemcpy(scene->rel_jobs_dev, scene->rel_jobs.data(), scene->rel_jobs.size() * sizeof(RelJob), hipMemcpyHostToDevice));
        std::vector<RelInfo> none((size_t)scene->d.n_prims);
        for (RelInfo &ri : none) { ri.offset = -1; ri.o[0] = ri.o[1] = ri.o[2] = 0.0; }
        HIP_TRY(hipMemcpy(scene->rel_info, none.data(), none.size() * sizeof(RelInfo), hipMemcpyHostToDevice));
        scene->rel_valid = false;
    }
    scene->d.rel = static_cast<const float4 *>(scene->rel);
    scene->d.rel_info = static_cast<const RelInfo *>(scene->rel_info);
    if (scene->rel_valid && std::memcmp(scene->rel_camera, cam.to_root, sizeof(scene->rel_camera)) == 0) return RSX_OK;
    // passes in flight on other streams may still read the records of the previous camera
    for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    long long most = 0;
    for (const RelJobHost &j : scene->rel_jobs) most = std::max(most, j.n_items);
    hipLaunchKernelGGL(k_camera_relative, dim3((unsigned)((most + 255) / 256), (unsigned)scene->rel_jobs.size()), dim3(256), 0, stream, scene->d, cam,
                       static_cast<const RelJob *>(scene->rel_jobs_dev), static_cast<float4 *>(scene->rel), static_cast<RelInfo *>(scene->rel_info));
    HIP_TRY(hipGetLastError());
    // The records are read by whichever lane renders with this camera next — private, non-blocking streams without an ordering against
    // `stream` — and that lane only sees rel_valid on the host: the fill is complete before anybody is told. (Once per camera.)
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(scene->rel_camera, cam.to_root, sizeof(scene->rel_camera));
    scene->rel_valid = true;
    return RSX_OK;
}

// shared body of rsx_render_pinhole / rsx_render_pinhole_frame
int render(rsx_scene *scene, const rsx_render_desc *desc, double *h_mean, double *h_var, double *fmean, double *fvar, int32_t *fn,
           int32_t frame_bins, int32_t slice_offset, uint64_t *ray_count, const double *h_xyz = nullptr, double delta_wavelength = 0.0) {
    if (!scene || !desc) return rsx_fail(RSX_EINVAL, "render: null argument");
    if (desc->n_tasks < 0 || desc->spp < 1 || desc->bins < 1) return rsx_fail(RSX_EINVAL, "render: n_tasks/spp/bins out of range");
    if (desc->rng_mode == RSX_RNG_STREAM && !desc->uniforms) return rsx_fail(RSX_EINVAL, "render: RSX_RNG_STREAM needs uniforms");
    if (!desc->tasks) {
        const long long w = desc->rect[2] - desc->rect[0], h = desc->rect[3] - desc->rect[1];
        if (w <= 0 || h <= 0 || w * h != desc->n_tasks) return rsx_fail(RSX_EINVAL, "render: rect does not match n_tasks");
    }
    for (int32_t i = 0; i < desc->n_materials; ++i)
        if (desc->materials[i].type != RSX_MAT_ABSORBER && desc->materials[i].type != RSX_MAT_NULL &&
            (desc->materials[i].table < 0 || desc->materials[i].table >= desc->n_tables))
            return rsx_fail(RSX_EINVAL, "render: material %d references table %d of %d", i, desc->materials[i].table, desc->n_tables);
    // passes = K > 1: K consecutive passes in one launch. The trace side sees ONE pass of K * spp samples per pixel — the Philox counter of
    // sample s of pass p is sample_offset + p * spp + s either way, and a pixel's records follow one another pass by pass; the accumulate
    // kernel runs the recurrence and the frame merge once per pass. Small passes (1024^2 x 1 spp is 16 k waves, a fraction of a millisecond)
    // are bound by launch tails and by incoherent waves; K of them in one launch are neither.
    const int32_t passes = desc->passes > 1 ? desc->passes : 1;
    rsx_render_desc widened;
    if (passes > 1) {
        if (!fmean || h_mean || h_xyz) return rsx_fail(RSX_EUNSUPPORTED, "render: passes > 1 accumulates into a device frame (rsx_render_pinhole_frame)");
        if (desc->rng_mode != RSX_RNG_PHILOX) return rsx_fail(RSX_EUNSUPPORTED, "render: passes > 1 needs RSX_RNG_PHILOX (a serial stream is consumed pass by pass)");
        if ((long long)desc->spp * passes > (1 << 20)) return rsx_fail(RSX_EINVAL, "render: spp * passes out of range");
        // (path-traced scenes too: the path kernel sees one pass of K * spp samples per pixel, whose term arena grows and is traced again
        // like any pass's before anything is merged; k_accumulate replays the lists pass by pass)
        bool path_terms = false;
        for (int32_t i = 0; i < desc->n_materials; ++i) {
            const int32_t mt = desc->materials[i].type;
            path_terms = path_terms || mt == RSX_MAT_NULL || mt == RSX_MAT_UNIFORM_VOLUME_EMITTER || mt == RSX_MAT_LAMBERT || mt == RSX_MAT_DIELECTRIC;
        }
        // K x spp that does not divide the 64 rays of a unit (or exceeds them) would cut pixels across units and keep the recurrence out of the
        // packet kernel: such a call is served as several — the largest power-of-two groups that fit, then the rest (24 passes of 1 spp
        // = 16 + 8) — passes are merged in their order either way
        if (!path_terms && desc->spp < WAVE && WAVE % desc->spp == 0 && (WAVE % (desc->spp * passes) != 0 || desc->spp * passes > WAVE)) {
            uint64_t rays = 0;
            int32_t done = 0;
            while (done < passes) {
                int32_t k = 1;
                while (2 * k <= passes - done && desc->spp * 2 * k <= WAVE) k *= 2;
                rsx_render_desc part = *desc;
                part.passes = k;
                part.sample_offset = desc->sample_offset + (uint64_t)done * (uint64_t)desc->spp;
                uint64_t count = 0;
                const int rc_part = render(scene, &part, h_mean, h_var, fmean, fvar, fn, frame_bins, slice_offset, &count, h_xyz, delta_wavelength);
                if (rc_part) return rc_part;
                rays += count;
                done += k;
            }
            if (ray_count) *ray_count = rays;
            return RSX_OK;
        }
        widened = *desc;
        widened.spp = desc->spp * passes;
        widened.passes = 1;
        desc = &widened;
    }
    if (ray_count) *ray_count = (uint64_t)desc->n_tasks * (uint64_t)desc->spp;
    if (desc->n_tasks == 0) return RSX_OK;
    rsx_ctx *ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t T = (size_t)desc->n_tasks, S = T * (size_t)desc->spp, B = (size_t)desc->bins;
    void *d_mat = nullptr, *d_tab = nullptr, *d_tasks = nullptr, *d_mean = nullptr, *d_var = nullptr;
    int rc;
    // which lane traces this pass: frame renders alternate the two private lanes (passes overlap), everything else stays on the ctx stream
    // Small passes are tail-bound and overlap well; a large pass fills the chip by itself (its merge kernel could not even get
    // registers next to it), so it runs alone on the ctx stream.
    const long long rect_w = desc->rect[2] - desc->rect[0], rect_h = desc->rect[3] - desc->rect[1];
    // 64-ray units over the rays g = pixel * spp + sample, pixels tile by tile (see unit_pixel in dev_render.hpp)
    const long long tiles_x_all = (rect_w + 7) >> 3, tiles_y_all = (rect_h + 7) >> 3;
    const long long n_units_all = desc->tasks ? (desc->n_tasks * (long long)desc->spp + WAVE - 1) / WAVE : tiles_x_all * tiles_y_all * (long long)desc->spp;
    // transparent boundaries, volume emitters and scattering surfaces take the path kernel (k_render_trace_path), un-pipelined
    bool has_vol = false, has_scatter = false;
    for (int32_t i = 0; i < desc->n_materials; ++i) {
        const int32_t mt = desc->materials[i].type;
        has_vol = has_vol || mt == RSX_MAT_NULL || mt == RSX_MAT_UNIFORM_VOLUME_EMITTER || mt == RSX_MAT_LAMBERT || mt == RSX_MAT_DIELECTRIC;
        has_scatter = has_scatter || mt == RSX_MAT_LAMBERT || mt == RSX_MAT_DIELECTRIC;
    }
    if (has_scatter) {
        if (desc->rng_mode != RSX_RNG_PHILOX)
            return rsx_fail(RSX_EUNSUPPORTED, "render: scattering materials draw per-path random numbers; use RSX_RNG_PHILOX (a serial MT19937-64 stream cannot be consumed in parallel)");
        if (desc->ray_extinction_min_depth < 1) return rsx_fail(RSX_EINVAL, "render: the minimum extinction depth cannot be less than 1");   // ray.pyx:276
        if (desc->ray_max_depth < desc->ray_extinction_min_depth) return rsx_fail(RSX_EINVAL, "render: the maximum depth cannot be less than the minimum extinction depth");
        if (!(desc->ray_extinction_prob >= 0.0 && desc->ray_extinction_prob <= 1.0)) return rsx_fail(RSX_EINVAL, "render: the extinction probability must lie in [0, 1]");
        if (desc->n_important > 0 && !(desc->important_path_weight >= 0.0 && desc->important_path_weight <= 1.0))
            return rsx_fail(RSX_EINVAL, "render: the important path weight must lie in [0, 1]");               // ray.pyx:107
        if (desc->ray_max_depth >= (1 << 15)) return rsx_fail(RSX_EUNSUPPORTED, "render: ray_max_depth %d exceeds the Philox draw counter's range (32767)", desc->ray_max_depth);
    }
    // CSG scenes render in two passes (state-free evaluator, then the stream merge for the rays that need it); their passes are not pipelined
    const bool two_pass_csg = scene->has_csg && scene->d.csgfast != nullptr;
    // Path passes whose end-of-pass checks the caller collects later (rsx_defer_path_checks: the spectral slices of one observe())
    // run on the private lanes like small primary passes: the tail of a slice — a few paths bouncing inside a prism for hundreds of
    // segments — then drains while the next slice's bulk fills the chip. Their per-call inputs live in lane-owned buffers.
    // (a scene whose stream-merge states live in the scene's arena — operand trees above CSG_MAX_SLOTS nodes — has ONE arena, indexed by
    // the launch's own blockIdx: two of its passes in flight on different lanes would write each other's node states in the middle of a
    // merge, so its passes run one after the other on the context stream)
    const bool one_at_a_time = scene->d.csg_arena != nullptr;
    const bool deferred = !one_at_a_time && has_vol && ctx->defer_path && fmean && !h_mean && !h_xyz && ctx->pipeline_depth > 1 && n_units_all <= (long long)RSX_LPT_MAX_UNITS;
    // (a call of several path passes may be deferred like any other: a failed call is left out of the frame as a whole and issued again)
    // Fused form (dev_render.hpp, "Welford in the trace kernel"): a pass that runs alone on the context stream, merges into a frame,
    // has closed-form materials only and whole pixels per 64-ray unit keeps its sample records in per-wave rings
    // — round 2, per-lane walk: measured on configs[2], 2048^2 x 64 spp: two kernels 34.9 + 5.8 = 40.8 ms per pass, fused 41.9 ms. That
    // trace kernel is bound by instruction issue (VALU busy 0.75), so the recurrence finds no idle slots to hide in, and inside the
    // wave it runs at 60 of 64 lanes plus the staging; the 6.4 GB it saves were never the bound (HBM at 2 % of peak).
    // From how many samples per pixel on the packet walk wins depends on the scene (tools/spp_sweep.py, 1024^2 frames): the instanced
    // configs[2] scene from 16 (equal at 8: a unit of eight pixels straddles world leaves and instances), a single mesh (configs[1])
    // from 4 (+21 % there, +22 % at 8, 2.2x at 32), the CSG demo at every count. RSX_PACKET_MIN_SPP pins it (0: never).
    static const int packet_min_env = [] { const char *e = std::getenv("RSX_PACKET_MIN_SPP"); return e ? std::atoi(e) : -1; }();
    const int packet_min_spp = packet_min_env >= 0 ? packet_min_env : scene->has_csg ? 1 : scene->d.n_world <= 4 ? 4 : RSX_PACKET_MIN_SPP;
    static const int fuse_env = [] { const char *e = std::getenv("RSX_FUSE"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();
    // few pixels per 64-ray unit: the wave walks the trees as one packet (dev_packet.hpp), with its own, smaller LDS layout
    // (a task list — FullFrameSampler2D shuffles its pixels, an adaptive sampler picks them — puts unrelated pixels side by side: there a
    // unit must be ONE pixel's samples for its rays to share their way through the trees)
    // (CSG scenes: the fast pass of the two — state-free evaluator — may be the packet kernel; RSX_PACKET_CSG=0 keeps the per-lane one)
    static const bool packet_csg = [] { const char *e = std::getenv("RSX_PACKET_CSG"); return !e || std::atoi(e) != 0; }();
    // (the packet walk keeps the node ids of its pending entries in the lanes of one vector register: world levels + mesh levels <= 64)
    const bool use_packet = !has_vol && (!scene->has_csg || (two_pass_csg && packet_csg && !h_xyz)) && packet_min_spp > 0 && desc->spp >= (desc->tasks ? std::max(packet_min_spp, WAVE) : packet_min_spp) &&
                            !ctx->unit_times && scene->d.wdepth + scene->d.mdepth <= WAVE;
    const size_t wave_lds = use_packet ? packet_lds_bytes(scene->d.wdepth, scene->d.mdepth, scene->d.csg_fast_rows) : (size_t)(scene->d.wlds + scene->d.mlds) * WAVE * 12 + STAGE_BYTES;
    const size_t fuse_fixed = (size_t)FUSE_UNITS * WAVE * 20 + ((size_t)desc->spp + 2) * 8;
    // (default: on for packet passes — round 3: their trace kernel waits on latency, not on instruction issue, and hides the recurrence:
    // configs[2] 24.5 + 5.3 ms as two kernels, 27.9 ms fused — off otherwise; RSX_FUSE=0 / 1 forces either)
    const bool fuse_enabled = fuse_env < 0 ? use_packet : fuse_env != 0;
    // (what decides the fused form apart from "runs alone": batch_fusable below takes multi-pass calls off the pipelined lanes exactly when this holds)
    const bool fusable = fuse_enabled && (passes == 1 || use_packet) && fmean && !h_mean && !has_vol && !scene->has_csg && desc->spp <= WAVE && WAVE % desc->spp == 0 &&
                         !ctx->unit_times && wave_lds >= fuse_fixed;
    // (a call of several passes — rsx_render_desc.passes, what batched small passes arrive as — runs alone as well: on the context stream it
    // may take the fused form, the recurrence and its K frame merges inside the packet kernel; on a private lane it would need
    // k_accumulate's multi-pass form, which costs four times the trace of a 16-pass batch of configs[1]. RSX_BATCH_ALONE=0: as before)
    static const bool batch_alone = [] { const char *e = std::getenv("RSX_BATCH_ALONE"); return !e || std::atoi(e) != 0; }();
    // (desc->spp is the widened count here: passes x samples per pass; the packet walk's threshold as further down, without its env override)
    const bool batch_fusable = batch_alone && passes > 1 && fusable && !h_xyz;
    // A path pass of its own (one slice per observe(): the Cornell box) also renders on a private lane — two of them, used in turn: its replay
    // (k_accumulate over the term lists, on the context stream) then runs beside the NEXT pass's path kernel instead of in front of it, and
    // the end-of-pass check waits for the lane's stream only. The check itself stays where it was: the pass is known to be complete before
    // its merge is enqueued, passes merge in call order, frames are those of the un-pipelined form bit for bit.
    // Measured (round 5, Cornell box 1024^2 x 16 spp, bench.py --workload c1): 30.61 ms per pass without, 31.10 with — the next pass's
    // persistent workgroups take every register the replay's waves would need, so the two still run one after the other. Opt-in
    // (RSX_PATH_PIPELINE=1) until the path kernel leaves room.
    static const bool path_pipeline = [] { const char *e = std::getenv("RSX_PATH_PIPELINE"); return e && std::atoi(e) != 0; }();
    const size_t solo_lane_bytes = S * (sizeof(Sample) + (size_t)PATH_BLOCK * sizeof(PathTerm) * 5 / 4 + 4) + 1;
    const bool solo_path = path_pipeline && !deferred && !one_at_a_time && has_vol && fmean && !h_mean && !h_xyz && ctx->pipeline_depth > 1 && !ctx->timing &&
                           n_units_all <= (long long)RSX_LPT_MAX_UNITS && 2 * solo_lane_bytes <= ((size_t)64 << 30);
    const bool pipelined = !one_at_a_time && !batch_fusable && (deferred || solo_path || (!h_mean && !has_vol && !two_pass_csg)) && ctx->pipeline_depth > 1 &&
                           n_units_all <= (long long)RSX_LPT_MAX_UNITS;
    // Scattering passes can run level by level (dev_wavefront.hpp: one launch per path segment over lists of live paths filed by material
    // arm) instead of in the one persistent kernel. Measured on the Cornell box (round 5, profiles/r05_wf_*): lane utilisation 0.42 -> 0.64,
    // a third fewer vector instructions, and 31.2 ms against 28.1 — the arm's 3.5 us per 64-path iteration come back as exposed memory
    // latency (list entry -> path record -> atomic, behind the iteration's own stores on the in-order vector-memory counter) at two waves per
    // SIMD. So the form is opt-in (rsx_set_path_stages / RSX_WAVEFRONT=1) until its loads are moved off the critical path; never for the
    // overlapping slices of one observe() (their host never waits for a launch; the levels read counts back between batches) or a CSG scene
    // without the state-free evaluator.
    static const int wf_env = [] { const char *e = std::getenv("RSX_WAVEFRONT"); return e ? std::atoi(e) : 0; }();
    static const long long wf_min_paths = [] { const char *e = std::getenv("RSX_WF_MIN_PATHS"); return e ? std::atoll(e) : (1LL << 18); }();
    const bool use_wf = (ctx->wf_mode < 0 ? wf_env != 0 : ctx->wf_mode != 0) && has_scatter && !deferred && (!scene->has_csg || two_pass_csg) &&
                        (long long)S >= (ctx->wf_min_paths >= 0 ? ctx->wf_min_paths : wf_min_paths) && (!ctx->unit_times || RSX_PHASE_PROF == 3);
    // (two lanes for path passes: the path kernel fits two workgroups per CU, and each pass brings a grid of that size — the
    // next slice's workgroups move in as this slice's retire; prism, 32 slices: 3 lanes x 1 workgroup per CU 741 ms, 2 x 2 582 ms)
    // (round 3: up to eight lanes — with the trapped paths handed to a small drain launch a pass gives its workgroup places back after
    // its bulk, and what limits the slices per second is how many passes are in flight; each lane keeps its own sample records and
    // term blocks, 0.4 KB per path: the lanes in use stay below 48 GB)
    const size_t lane_bytes = S * (sizeof(Sample) + (size_t)PATH_BLOCK * sizeof(PathTerm) * 5 / 4 + 4) + 1;
    const int path_lanes = (int)std::max<size_t>(2, std::min<size_t>((size_t)ctx->path_lanes, ((size_t)48 << 30) / lane_bytes));
    TraceLane &lane = deferred ? ctx->lanes[ctx->deferred_calls % path_lanes] : solo_path ? ctx->lanes[ctx->render_calls % 2] :
                      pipelined ? ctx->lanes[ctx->render_calls % ctx->pipeline_depth] : ctx->main;
    // (a pass that runs alone waits for the private lanes' traces. Their `in_flight` stays set: the lane's merge kernel sits on the ctx
    // stream and may still be reading the lane's sample records — the lane's next pass must wait for `merged` before it overwrites them.
    // Clearing the flag here let a pipelined pass that followed a lone one race its lane's previous merge: round 5, found by submitting
    // partial batches of small passes eagerly, tools/r5_eager_repro.py.)
    if (!pipelined) for (TraceLane &ln : ctx->lanes) if (ln.in_flight) HIP_TRY(hipStreamSynchronize(ln.stream));
    if ((rc = settle_lane(ctx, lane))) return rc;          // (a deferred pass this lane ran before: its buffers are about to be reused)
    // (the work lists of this lane's previous lone path pass were sorted on the side stream: everything this pass puts on the lane's stream comes after)
    if (lane.sort_pending) { HIP_TRY(hipStreamWaitEvent(lane.stream, lane.sorted, 0)); lane.sort_pending = false; }

    // materials and the importance manager's spheres share one small buffer: [materials][spheres]
    const int n_important = has_scatter && desc->important ? std::max(0, desc->n_important) : 0;
    const size_t mat_bytes = sizeof(rsx_material) * (size_t)std::max(1, desc->n_materials), imp_bytes = sizeof(rsx_important_sphere) * (size_t)n_important;
    std::vector<unsigned char> mat_blob(mat_bytes + imp_bytes, 0);
    if (desc->n_materials) std::memcpy(mat_blob.data(), desc->materials, sizeof(rsx_material) * (size_t)desc->n_materials);
    if (n_important) std::memcpy(mat_blob.data() + mat_bytes, desc->important, imp_bytes);
    // A dielectric whose transmission is exactly 1 in every bin of this slice leaves every spectrum unchanged in evaluate_volume
    // (dielectric.pyx:300-328: samples *= pow(1, length) = 1, and x * 1 = x): the library marks it in ITS copy of the materials
    // (light_dir[2] = 1) and the path kernel does not list it among the volumes of a segment — no world.contains() pass at all when
    // nothing else in the scene has a volume contribution (a Cornell box with clear glass), no attenuation terms to replay.
    int32_t n_vol_contributors = 0;
    {
        rsx_material *bm = reinterpret_cast<rsx_material *>(mat_blob.data());
        for (int32_t i = 0; i < desc->n_materials; ++i) {
            if (bm[i].type == RSX_MAT_UNIFORM_VOLUME_EMITTER) ++n_vol_contributors;
            else if (bm[i].type == RSX_MAT_DIELECTRIC) {
                bool unit = desc->tables != nullptr;
                const double *row = desc->tables + (size_t)bm[i].table * B;
                for (size_t b = 0; unit && b < B; ++b) unit = row[b] == 1.0;
                bm[i].light_dir[2] = unit ? 1.0 : 0.0;
                if (!unit) ++n_vol_contributors;
            }
        }
    }
    if ((rc = pool_get(ctx, POOL_MATERIALS, mat_blob.size(), &d_mat)) ||
        (rc = pool_get(ctx, POOL_TABLES, 8 * B * (size_t)(std::max(1, desc->n_tables) + 3), &d_tab))) return rc;
    // small per-call inputs are uploaded only when they differ from what the device already holds (steady-state
    // passes of one observe() loop re-send identical materials / tables / task lists); a change drains the pipeline first
    auto upload_if_changed = [&](int slot, void *dst, const void *src, size_t bytes) -> int {
        std::vector<unsigned char> &sh = ctx->shadow[slot];
        if (sh.size() == bytes && std::memcmp(sh.data(), src, bytes) == 0) return RSX_OK;
        for (TraceLane &ln : ctx->lanes) if (ln.in_flight) { HIP_TRY(hipStreamSynchronize(ln.stream)); ln.in_flight = false; }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        sh.assign(static_cast<const unsigned char *>(src), static_cast<const unsigned char *>(src) + bytes);
        HIP_TRY(hipMemcpyAsync(dst, sh.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return RSX_OK;
    };
    // the lane's previous pass must have been merged before its buffers (inputs, samples, scheduling state) are reused
    if (pipelined && lane.in_flight) HIP_TRY(hipStreamWaitEvent(lane.stream, lane.merged, 0));
    auto lane_upload = [&](void *&dev, size_t &dev_bytes, std::vector<unsigned char> &host, const void *src, size_t bytes) -> int {
        if (bytes > dev_bytes) {
            if (dev) { HIP_TRY(hipStreamSynchronize(lane.stream)); HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(dev)); dev = nullptr; dev_bytes = 0; }
            HIP_TRY(malloc_or_flush(ctx, &dev, bytes + 256));
            dev_bytes = bytes + 256;
        }
        host.assign(static_cast<const unsigned char *>(src), static_cast<const unsigned char *>(src) + bytes);   // (lives until the lane's next pass)
        if (bytes) HIP_TRY(hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, lane.stream));
        return RSX_OK;
    };
    if (deferred) {                                        // every slice brings its own materials and tables: no shared copy, no drain
        if ((rc = lane_upload(lane.mat_dev, lane.mat_dev_bytes, lane.mat_host, mat_blob.data(), mat_blob.size()))) return rc;
        d_mat = lane.mat_dev;
    } else
    if (desc->n_materials && (rc = upload_if_changed(0, d_mat, mat_blob.data(), mat_blob.size()))) return rc;
    // spectral tables; the XYZ form appends the three resampled CIE curves as rows n_tables .. n_tables + 2 ([channel][bin])
    std::vector<double> tab_blob((size_t)(desc->n_tables + (h_xyz ? 3 : 0)) * B);
    if (desc->n_tables) std::memcpy(tab_blob.data(), desc->tables, 8 * B * (size_t)desc->n_tables);
    if (h_xyz) for (size_t b = 0; b < B; ++b) for (int c = 0; c < 3; ++c) tab_blob[((size_t)desc->n_tables + c) * B + b] = h_xyz[3 * b + c];
    if (deferred) {
        if ((rc = lane_upload(lane.tab_dev, lane.tab_dev_bytes, lane.tab_host, tab_blob.data(), 8 * tab_blob.size()))) return rc;
        d_tab = lane.tab_dev;
    } else
    if (!tab_blob.empty() && (rc = upload_if_changed(1, d_tab, tab_blob.data(), 8 * tab_blob.size()))) return rc;
    if (desc->tasks) {
        if ((rc = pool_get(ctx, POOL_TASKS, T * 8, &d_tasks))) return rc;
        if ((rc = upload_if_changed(2, d_tasks, desc->tasks, T * 8))) return rc;
    }
    if (h_mean) {
        const size_t out_channels = h_xyz ? 3 : B;
        if ((rc = pool_get(ctx, POOL_MEAN, T * out_channels * 8, &d_mean)) || (rc = pool_get(ctx, POOL_VAR, T * out_channels * 8, &d_var))) return rc;
    }
    auto lane_buffer = [&](void *&buf, size_t &have, size_t bytes) -> int {
        if (bytes > have) {
            if (buf) { HIP_TRY(hipStreamSynchronize(lane.stream)); HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(buf)); buf = nullptr; have = 0; }
            const size_t want = bytes + bytes / 8 + 256;
            HIP_TRY(malloc_or_flush(ctx, &buf, want));
            have = want;
        }
        return RSX_OK;
    };
    const bool fused = fusable && !pipelined;
    if (!fused && (rc = lane_buffer(lane.samples, lane.samples_bytes, S * sizeof(Sample)))) return rc;
    {
        void *const redo_before = lane.redo;
        if (two_pass_csg && (rc = lane_buffer(lane.redo, lane.redo_bytes, (size_t)n_units_all * 8))) return rc;
        if (lane.redo != redo_before) lane.redo_zeroed = 0;
    }
    // path terms: every ray owns one PATH_BLOCK-slot block; longer paths chain blocks out of a shared arena
    size_t arena_blocks = 0;
    if (has_vol) {
        // (the path kernel keeps a path's pixel index — the Philox counter's first word — in 32 bits)
        if ((unsigned long long)desc->camera.nx * (unsigned long long)desc->camera.ny > 0xffffffffULL)
            return rsx_fail(RSX_EUNSUPPORTED, "render: path-traced passes address at most 2^32 camera pixels (%d x %d asked for)", desc->camera.nx, desc->camera.ny);
        // (two blocks per path for the lists themselves; a third because waves take blocks ARENA_BATCH at a time — arena_block, dev_render.hpp —
        // and a wave that needs one block holds sixty-four)
        arena_blocks = std::max<size_t>((size_t)1 << 17, (has_scatter ? 3 : 1) * S);
        if (const char *e = std::getenv("RSX_PATH_ARENA")) arena_blocks = (size_t)std::max(0ll, std::atoll(e));
        const size_t pool_bytes = (S + arena_blocks) * PATH_BLOCK * sizeof(PathTerm);
        if (S + arena_blocks >= ((size_t)1 << 31) || pool_bytes > ((size_t)48 << 30))
            return rsx_fail(RSX_EUNSUPPORTED, "render: %zu rays with path terms in one call; split the call (at most %zu rays)", S, ((size_t)48 << 30) / (3 * PATH_BLOCK * sizeof(PathTerm)));
        if ((rc = lane_buffer(lane.terms, lane.terms_bytes, pool_bytes)) || (rc = lane_buffer(lane.tail, lane.tail_bytes, S * sizeof(int32_t)))) return rc;
        if (!lane.overflow) HIP_TRY(hipMalloc(&lane.overflow, 64));
        HIP_TRY(hipMemsetAsync(lane.overflow, 0, 64, lane.stream));
    }
    if (desc->rng_mode == RSX_RNG_STREAM) {
        if ((rc = lane_buffer(lane.uniforms, lane.uniforms_bytes, S * 16))) return rc;
        HIP_TRY(hipMemcpyAsync(lane.uniforms, desc->uniforms, S * 16, hipMemcpyHostToDevice, lane.stream));
    }

    bool want_order = false;
    long long order_n = 0;
    int order_tiles_x = 0;
    RenderParams rp;
    rp.cam = desc->camera;
    camera_origin(rp.cam, rp.origin);
    rp.materials = static_cast<const rsx_material *>(d_mat);
    rp.tasks = desc->tasks ? static_cast<const int32_t *>(d_tasks) : nullptr;
    rp.uniforms = desc->rng_mode == RSX_RNG_STREAM ? static_cast<const double *>(lane.uniforms) : nullptr;
    rp.n_tasks = desc->n_tasks;
    std::memcpy(rp.rect, desc->rect, sizeof(rp.rect));
    rp.spp = desc->spp;
    rp.rng_mode = desc->rng_mode;
    rp.redo_mask = two_pass_csg ? static_cast<unsigned long long *>(lane.redo) : nullptr;
    rp.seed = desc->seed;
    rp.sample_offset = desc->sample_offset;
    rp.important = reinterpret_cast<const rsx_important_sphere *>(static_cast<const unsigned char *>(d_mat) + mat_bytes);
    rp.n_important = n_important; rp.passes = 1; rp.important_path_weight = desc->important_path_weight;
    rp.n_vol_emitters = 0; rp.world_lds = 0; rp.prims_lds = 0; rp.bank_lds = 0;
    rp.n_vol_emitters = n_vol_contributors;
    rp.ray_max_depth = desc->ray_max_depth; rp.ray_min_depth = desc->ray_extinction_min_depth; rp.ray_extinction_prob = desc->ray_extinction_prob;
    rp.unit_times = ctx->unit_times;
    // longest-first unit schedule from the costs this lane's previous pass over the same units measured
    {
        const long long w = desc->rect[2] - desc->rect[0], h = desc->rect[3] - desc->rect[1];
        const long long n_units = n_units_all;
        (void)w; (void)h;
        uint64_t sig = 1469598103934665603ULL;                       // FNV-1a over what defines the units' content
        auto mix = [&sig](const void *q, size_t bytes) { const unsigned char *c = static_cast<const unsigned char *>(q); for (size_t i = 0; i < bytes; ++i) { sig ^= c[i]; sig *= 1099511628211ULL; } };
        const void *scene_id = scene;
        mix(&scene_id, sizeof(scene_id)); mix(&desc->camera, sizeof(desc->camera)); mix(desc->rect, sizeof(desc->rect));
        mix(&desc->n_tasks, sizeof(desc->n_tasks)); mix(&desc->spp, sizeof(desc->spp));
        if (desc->tasks) mix(desc->tasks, (size_t)std::min<long long>(desc->n_tasks, 4096) * 8);
        if ((size_t)n_units > lane.unit_capacity) {
            HIP_TRY(hipStreamSynchronize(lane.stream));
            if (lane.unit_cost) HIP_TRY(hipFree(lane.unit_cost));
            if (lane.unit_order) HIP_TRY(hipFree(lane.unit_order));
            lane.unit_capacity = (size_t)n_units + (size_t)n_units / 8 + 64;
            lane.cost_zeroed = 0;
            HIP_TRY(malloc_or_flush(ctx, &lane.unit_cost, lane.unit_capacity * 4));
            HIP_TRY(malloc_or_flush(ctx, &lane.unit_order, lane.unit_capacity * 4));
            if (!lane.n_work) HIP_TRY(hipMalloc(&lane.n_work, 64));
            lane.cost_units = 0;
            lane.order_units = 0;
        }
        rp.unit_cost = lane.unit_cost;
        rp.unit_order = nullptr;
        rp.seg = nullptr;
        // the work lists for this pass were sorted right after the lane's previous pass over the same units (see below);
        // a first pass (or a changed camera / task list) sorts zero costs, i.e. natural order split over the XCD lists
        if (n_units >= (1LL << 26)) return rsx_fail(RSX_EUNSUPPORTED, "render: more than 2^26 work units in one launch; split the call");
        order_tiles_x = desc->tasks ? 0 : (int)tiles_x_all;
        if (!(lane.order_units == n_units && lane.cost_signature == sig)) {
            lane.sorts_done = 0;
            HIP_TRY(hipMemsetAsync(lane.unit_cost, 0, (size_t)n_units * 4, lane.stream));
            lane.cost_zeroed = n_units;
            {   // no costs yet: natural order per XCD list, laid out by all CUs (k_order_natural_*)
                const int n_blocks = (int)((n_units + ORDER_BLOCK_UNITS - 1) / ORDER_BLOCK_UNITS);
                if ((rc = lane_buffer(lane.order_counts, lane.order_counts_bytes, (size_t)n_blocks * 8 * 4 + 64))) return rc;
                uint32_t *bc = static_cast<uint32_t *>(lane.order_counts);
                hipLaunchKernelGGL(k_order_natural_count, dim3((unsigned)n_blocks), dim3(1024), 0, lane.stream, bc, n_units, order_tiles_x, (int)desc->spp);
                hipLaunchKernelGGL(k_order_natural_scan, dim3(1), dim3(64), 0, lane.stream, bc, lane.n_work, n_blocks);
                hipLaunchKernelGGL(k_order_natural_scatter, dim3((unsigned)n_blocks), dim3(1024), 0, lane.stream, bc, lane.n_work, lane.unit_order, n_units, order_tiles_x, (int)desc->spp);
            }
            HIP_TRY(hipGetLastError());
        }
        rp.unit_order = lane.unit_order;
        rp.seg = lane.n_work;
        // Longest-first ordering pays when a pass is tail-bound (few units per wave); a pass with thousands of units per wave
        // balances by itself, so it keeps the natural (XCD-blocked) order and the kernel skips the cost bookkeeping.
        // Path passes (has_vol) are tail-bound at any size — a few paths trapped by total internal reflection run to the depth limit —
        // and record the longest path of each unit (atomicMax in k_render_trace_path); spectral slices of one observe() share camera and
        // units, so slice k + 1 starts the units that held slice k's longest paths first.
        static const int path_lpt = [] { const char *e = std::getenv("RSX_PATH_LPT"); return e ? std::atoi(e) : 2; }();
        // (a call of several passes — batched small passes — is as many times the units of one and balances by itself; the one-workgroup sort
        // behind it cost a 16-pass batch of configs[1] 0.25 - 0.8 ms next to a 1.3 ms kernel: profiles/r05a_c2_kernel_stats.csv. Path passes stay
        // tail-bound however many a call carries: they keep the ordering — 96^2 x 16 spp Cornell passes, two per call: 3.7 ms per pass without)
        want_order = RSX_LPT_SCHEDULE != 0 && n_units <= (long long)RSX_LPT_MAX_UNITS && (passes == 1 || has_vol) && (has_vol ? path_lpt > 0 && !use_wf : !two_pass_csg);
        rp.measure_cost = want_order ? 1 : 0;
        if (has_vol && path_lpt == 1) want_order = false;           // (measure, do not re-order: tuning aid)
        // Primary-ray passes over the same units cost the same from pass to pass (the rays differ only by their jitter): the list
        // sorted from the first measured passes stays good, so later passes neither measure nor sort (k_order_units is one
        // workgroup: 124 us per 16 384 units, a sixth of a 1024 x 1024 pass on the lane's stream).
        static const int order_passes = [] { const char *e = std::getenv("RSX_ORDER_PASSES"); return e ? std::atoi(e) : 2; }();
        if (want_order && !has_vol && lane.sorts_done >= order_passes) {
            want_order = false;
            rp.measure_cost = 0;
            lane.order_units = n_units;                      // the list stays valid
        }
        if (want_order) lane.sorts_done++;
        // (path passes measure in every pass: k_order_units zeroes the costs it consumed, so only a list of unknown state is cleared here)
        if (want_order && has_vol && lane.cost_zeroed != n_units) HIP_TRY(hipMemsetAsync(lane.unit_cost, 0, (size_t)n_units * 4, lane.stream));
        if (rp.measure_cost) lane.cost_zeroed = 0;
        lane.order_units = want_order ? 0 : n_units;
        order_n = n_units;
        lane.cost_units = n_units;
        lane.cost_signature = sig;
    }

    Launch l;
    // pipelined passes share the chip: each takes RSX_RENDER_WG_PER_CU workgroups per CU so that the other lane's pass, the merge
    // kernel and the sort always find free slots (a persistent grid that filled every slot would serialise them behind its tail)
    // A pipelined (small, tail-bound) pass takes one workgroup per CU and the neighbouring lane's pass fills the idle CUs.
    int wg_cap = pipelined && !solo_path ? RSX_RENDER_WG_PER_CU : RSX_MAX_WG_PER_CU;
    // (the overlapping slices of an observe(): ONE workgroup per CU and launch — eight lanes keep the two places of every CU taken, and a
    // bulk launch that brings two per CU holds the second place through its tail; configs[4]: 3.37e8 -> 3.49e8 paths/s. RSX_PATH_WG pins it)
    static const int path_wg_env = [] { const char *e = std::getenv("RSX_PATH_WG"); return e ? std::atoi(e) : 0; }();
    if (deferred) wg_cap = path_wg_env > 0 ? std::min(path_wg_env, RSX_PATH_MIN_WAVES) : 1;
    if (pipelined && ctx->render_wg_override > 0) wg_cap = ctx->render_wg_override;
    if ((rc = plan(scene, (long long)S, lane, l, wg_cap))) return rc;
    // (overlapping slices: RSX_PATH_GRID caps the workgroups one slice's path launch brings — fewer, longer-lived waves have shorter tails per
    // unit; measured on configs[4]: 256 (one per CU, the default) 7.04 s per step, 128: 7.00, 64: 9.24 — the six streams no longer fill the chip)
    static const int path_grid_env = [] { const char *e = std::getenv("RSX_PATH_GRID"); return e ? std::atoi(e) : 0; }();
    if (deferred && has_vol && path_grid_env > 0) l.grid.x = std::min<unsigned>(l.grid.x, (unsigned)path_grid_env);
    rp.world_lds = 0; rp.prims_lds = 0; rp.bank_lds = 0;
    if (has_vol) {
        rp.bank_lds = (int32_t)l.lds; l.lds += (size_t)WG_WAVES * RAY_BANK_BYTES;      // the path waves' ray banks (k_render_trace_path)
        // The path kernel runs two workgroups per CU (256 registers per lane), so 80 KB of LDS per workgroup are there for the taking:
        // a world tree that fits behind the traversal stacks is staged there. Scattered rays walk it with a dependent load per step
        // at two waves per SIMD — latency, not issue, is their bound, and LDS answers in a fraction of an L2 round trip.
        static const bool stage_world = [] { const char *e = std::getenv("RSX_WORLD_LDS_STAGE"); return !e || std::atoi(e) != 0; }();
        const size_t need = ((size_t)scene->d.n_wnodes * sizeof(rsx_kdnode) + (size_t)scene->d.n_witems * 4 + 15) & ~(size_t)15;
        if (stage_world && l.lds + need <= (160 / RSX_PATH_MIN_WAVES) * 1024 && need <= 24 * 1024) { rp.world_lds = (int32_t)l.lds; l.lds += need; }
        // ... and, for a scene of a few dozen primitives, the primitive records (376 bytes each) and the flattened CSG programs that
        // the lanes read one by one, a different one in every lane
        static const bool stage_prims = [] { const char *e = std::getenv("RSX_PRIMS_LDS_STAGE"); return !e || std::atoi(e) != 0; }();
        static_assert(sizeof(rsx_primitive) % 8 == 0 && sizeof(CsgFast) % 8 == 0, "staged with 8-byte copies");
        const size_t pneed = (size_t)scene->d.n_prims * (sizeof(rsx_primitive) + (scene->d.csgfast ? sizeof(CsgFast) : 0));
        if (rp.world_lds > 0 && stage_prims && l.lds + pneed <= (160 / RSX_PATH_MIN_WAVES) * 1024 && pneed <= 24 * 1024) { rp.prims_lds = (int32_t)l.lds; l.lds += pneed; }
    }
    HIP_TRY(hipFuncSetAttribute(scene->has_csg ? reinterpret_cast<const void *>(k_render_trace<true>) : reinterpret_cast<const void *>(k_render_trace<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
    if (!lane.ticket_armed && (rc = reset_ticket(lane))) return rc;
    FuseParams fz;
    std::memset(&fz, 0, sizeof(fz));
    // the packet kernel's grid: its stacks are smaller (one range per lane and level, no leaf staging), four waves per SIMD fit the
    // LDS — and the registers
    const size_t plds = (size_t)WG_WAVES * packet_lds_bytes(scene->d.wdepth, scene->d.mdepth, scene->d.csg_fast_rows);
    dim3 pgrid = l.grid;
    if (use_packet && !pipelined) {
        const long long per_cu = std::min<long long>(RSX_MAX_WG_PER_CU, (long long)((160 * 1024) / std::max<size_t>(plds, 1)));
        const long long needed = ((long long)S + WG_THREADS - 1) / WG_THREADS;
        pgrid = dim3((unsigned)std::max<long long>(1, std::min<long long>((long long)ctx->n_cus * per_cu, needed)));
    }
    if (fused) {
        const size_t n_waves = (size_t)(use_packet ? pgrid.x : l.grid.x) * WG_WAVES;
        if ((rc = lane_buffer(lane.ring, lane.ring_bytes, n_waves * FUSE_UNITS * WAVE * sizeof(Sample)))) return rc;
        fz.ring = static_cast<Sample *>(lane.ring);
        fz.tables = static_cast<const double *>(d_tab);
        fz.fmean = fmean; fz.fvar = fvar; fz.fn = fn;
        fz.sensitivity = desc->camera.sensitivity;
        fz.n_tables = desc->n_tables; fz.bins = desc->bins; fz.power = desc->power; fz.ny = desc->camera.ny;
        fz.frame_bins = frame_bins; fz.slice_offset = slice_offset;
        fz.lds_bytes = (int32_t)wave_lds;
        fz.consts = ctx->acc_consts;
        fz.passes = passes; fz.pass_spp = desc->spp / passes;
        fz.tables_in_lds = fuse_fixed + ((size_t)std::max(1, desc->n_tables) + 1) * B * 8 <= wave_lds ? 1 : 0;   // (+ the zero row of fused_chains)
    }
    const int slot = (int)(ctx->render_calls % RING_SLOTS);
    while (ctx->ring.size() < (size_t)(slot + 1) * 4) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ctx->ring.push_back(e); }
    hipEvent_t *re = &ctx->ring[(size_t)slot * 4];
    // keep the host at most `max_in_flight` passes ahead of the GPU: a host that queues hundreds of launches ahead fills the HSA
    // queues and the runtime's back-pressure wait then opens millisecond gaps between kernels (measured: 1.9 vs 0.9 ms per pass)
    HP_BEGIN
    // Keep the host at most `max_in_flight` passes ahead of the GPU so that long render loops cannot overflow the HSA queues.
    while (ctx->gate.size() < (size_t)(2 * RSX_MAX_LANES + 64)) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->gate.push_back(e); }
    if (ctx->render_calls >= ctx->max_in_flight) {
        const long long old = ctx->render_calls - ctx->max_in_flight;
        HIP_TRY(hipEventSynchronize(ctx->gate[(size_t)(old % (long long)ctx->gate.size())]));
    }
    HP_MARK(0)
    const bool timed = ctx->timing;
    if (timed) HIP_TRY(hipEventRecord(re[0], lane.stream));
    if (has_vol) {
        HIP_TRY(hipFuncSetAttribute(scene->has_csg ? reinterpret_cast<const void *>(k_render_trace_path<true>) : reinterpret_cast<const void *>(k_render_trace_path<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
        // Paths are deterministic (counter-based random numbers), so a pass whose term arena ran out is simply traced again with
        // a four times larger one — before its records are merged into the frame (one host round trip per path pass, small next
        // to the pass itself). RSX_PATH_ARENA pins the size instead.
        bool rewalk = false;                                 // second form of the kernels: any number of volumes at a point (slower)
        bool wf_off = false;                                 // the level-by-level form gave up (a sub-list overflowed: cannot happen, checked anyway): the one-kernel form renders the pass
        for (int attempt = 0;; ++attempt) {
            PathStore ps;
            ps.pool = static_cast<PathTerm *>(lane.terms); ps.tail = static_cast<int32_t *>(lane.tail); ps.n_records = (long long)S;
            ps.arena_blocks = (unsigned int)arena_blocks; ps.flags = lane.overflow; ps.arena_next = lane.overflow + 1;
            // waves that run out of new rays with a few long paths left hand them to a second, small launch (dev_render.hpp, PathState)
            static const int path_donate = [] { const char *e = std::getenv("RSX_PATH_DONATE"); return e ? std::atoi(e) : 1; }();
            ps.queue = nullptr; ps.queue_count = lane.overflow + 4; ps.queue_cap = 0; ps.drain = 0;
            // (passes that overlap others — the slices of one observe(): a pass that runs alone frees its places for nobody, and its last
            // paths would only move to a smaller grid; RSX_PATH_DONATE=2 hands on in every path pass, 0 in none)
            if (has_scatter && (path_donate > 1 || (path_donate == 1 && deferred))) {
                const size_t cap = (size_t)l.grid.x * WG_WAVES * PATH_DONATE_MAX;
                if ((rc = lane_buffer(lane.path_queue, lane.path_queue_bytes, cap * sizeof(PathState)))) return rc;
                ps.queue = static_cast<PathState *>(lane.path_queue); ps.queue_cap = (unsigned int)cap;
            }
            // (the drain launch: 32 workgroups — the handed-on paths of a 1 M-path slice fill a few hundred waves at first and a handful for
            // most of its 13 ms; a bigger grid only holds more places that the other slices' bulk launches could use. configs[4], paths/s:
            // 8 workgroups 3.19e8, 16 3.31e8, 24 3.34e8, 32 3.37e8, 48 3.28e8, 64 3.29e8, 128 3.23e8, 256 3.21e8, 512 3.17e8. RSX_DRAIN_GRID pins it)
            static const int drain_grid_env = [] { const char *e = std::getenv("RSX_DRAIN_GRID"); return e ? std::atoi(e) : 0; }();
            const dim3 drain_grid(std::min<unsigned>(l.grid.x, drain_grid_env > 0 ? (unsigned)drain_grid_env : 32u));
            auto launch = [&](const void *kernel, dim3 grid, int ticket_set = 0) -> int {
                HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
                unsigned long long *tickets = lane.ticket + (size_t)ticket_set * 9 * 16;
                void *args[] = {(void *)&scene->d, (void *)&rp, (void *)&lane.samples, (void *)&tickets, (void *)&ps};
                HIP_TRY(hipLaunchKernel(kernel, grid, dim3(WG_THREADS), args, l.lds, lane.stream));
                return RSX_OK;
            };
            // a launch and, behind it, the drain launch of the same kernel for the paths it handed on
            // (`with_queue`: the kernel's QUEUE instantiation, or null — only the forms of small scenes, staged in LDS, have one; any other
            // form runs without the hand-over)
            auto launch_drained = [&](const void *kernel, const void *with_queue, dim3 grid) -> int {
                PathState *const queue = ps.queue;
                if (!queue || !with_queue) {
                    ps.queue = nullptr;
                    const int rc1 = launch(kernel, grid);
                    ps.queue = queue;
                    return rc1;
                }
                int rc2 = launch(with_queue, grid);
                if (rc2) return rc2;
                ps.drain = 1;
                rc2 = launch(with_queue, drain_grid);
                ps.drain = 0;
                return rc2;
            };
            // The pass level by level (dev_wavefront.hpp), chunk by chunk: level 0 makes a chunk's paths and walks their first segment, every
            // further launch runs the material arms of the segments the previous one walked and walks the daughters' segments; after a
            // batch of levels the counts of live paths are read back, and once they are few the one-kernel form's drain launch walks them to
            // their ends (`drain_kernel`; without one the levels run until no path is left).
            auto run_staged = [&](const void *level_kernel, const void *drain_kernel) -> int {
                static const long long chunk_env = [] { const char *e = std::getenv("RSX_WF_CHUNK"); return e ? std::atoll(e) : (1LL << 25); }();
                static const int first_levels = [] { const char *e = std::getenv("RSX_WF_LEVELS"); return e ? std::min(32, std::max(1, std::atoi(e))) : 12; }();
                static const long long drain_below = [] { const char *e = std::getenv("RSX_WF_DRAIN_BELOW"); return e ? std::atoll(e) : (1LL << 15); }();
                const int more_levels = 8, rows = 64;
                const long long chunk_units = std::max<long long>(1, std::min<long long>(n_units_all, chunk_env / WAVE));
                const size_t chunk_n = (size_t)chunk_units * WAVE;
                const long long lds_per_cu = (long long)((160 * 1024) / std::max<size_t>(l.lds + 2048, 1));
                const long long level_wgs = (long long)ctx->n_cus * std::max<long long>(1, std::min<long long>(RSX_WF_MIN_WAVES, lds_per_cu));
                // a sub-list takes what the waves of one number modulo WF_SUB file: their share of the chunk's paths plus one partial chunk per
                // wave and segment (see wf_append: checked on the device as well)
                const size_t sub_stride = ((chunk_n / WF_SUB + 63) & ~(size_t)63) + 64 * ((size_t)(level_wgs * WG_WAVES) / WF_SUB + 2) + 64 * (WF_SEGS / WF_SUB + 1) + 4096;
                const size_t list_entries = (size_t)WF_SEGS * sub_stride;
                if ((rc = lane_buffer(lane.wf_paths, lane.wf_paths_bytes, chunk_n * sizeof(WfPath))) ||
                    (scene->has_csg && (rc = lane_buffer(lane.wf_hits, lane.wf_hits_bytes, chunk_n * sizeof(Hit)))) ||
                    (rc = lane_buffer(lane.wf_lists, lane.wf_lists_bytes, 2 * list_entries * 4)) || (rc = lane_buffer(lane.wf_counts, lane.wf_counts_bytes, (size_t)rows * WF_SEGS * 4))) return rc;
                const size_t queue_cap = std::max<size_t>((size_t)1 << 16, chunk_n / 8);
                PathState *const queue_before = ps.queue; const unsigned int cap_before = ps.queue_cap;
                if (drain_kernel) {
                    if ((rc = lane_buffer(lane.path_queue, lane.path_queue_bytes, queue_cap * sizeof(PathState)))) return rc;
                    ps.queue = static_cast<PathState *>(lane.path_queue); ps.queue_cap = (unsigned int)queue_cap;
                }
                HIP_TRY(hipFuncSetAttribute(level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
                WfStore wf;
                wf.paths = static_cast<WfPath *>(lane.wf_paths); wf.csg_hits = static_cast<Hit *>(lane.wf_hits);
                uint32_t *const lists = static_cast<uint32_t *>(lane.wf_lists), *const counts = static_cast<uint32_t *>(lane.wf_counts);
                wf.sub_stride = (uint32_t)sub_stride; wf.pad = 0;
                std::vector<uint32_t> row(WF_SEGS);
                for (long long first_unit = 0; first_unit < n_units_all; first_unit += chunk_units) {
                    const long long units = std::min<long long>(chunk_units, n_units_all - first_unit);
                    wf.n = units * WAVE; wf.first_unit = first_unit;
                    HIP_TRY(hipMemsetAsync(counts, 0, (size_t)rows * WF_SEGS * 4, lane.stream));
                    long long live = units * WAVE;                          // (upper bound until the first counts come back)
                    for (int level = 0, batch = first_levels;; batch = more_levels) {
                        for (int b = 0; b < batch; ++b, ++level) {
                            wf.cnt_in = level == 0 ? nullptr : counts + (size_t)((level - 1) % rows) * WF_SEGS;
                            wf.cnt_out = counts + (size_t)(level % rows) * WF_SEGS;
                            wf.list_in = lists + (size_t)((level + 1) & 1) * list_entries; wf.list_out = lists + (size_t)(level & 1) * list_entries;
                            const long long wave_chunks = (live + WAVE - 1) / WAVE + (level == 0 ? 0 : WF_SEGS), need_wgs = (wave_chunks + WG_WAVES - 1) / WG_WAVES;
                            void *args[] = {(void *)&scene->d, (void *)&rp, (void *)&lane.samples, (void *)&wf, (void *)&ps};
                            HIP_TRY(hipLaunchKernel(level_kernel, dim3((unsigned)std::max<long long>(1, std::min(level_wgs, need_wgs))), dim3(WG_THREADS), args, l.lds, lane.stream));
                        }
                        static_assert(64 + WF_SEGS * 4 <= 4096, "the lane's pinned read-back words hold a row of list counts");
                        HIP_TRY(hipMemcpyAsync(lane.host_words + 8, counts + (size_t)((level - 1) % rows) * WF_SEGS, WF_SEGS * 4, hipMemcpyDeviceToHost, lane.stream));
                        HIP_TRY(hipStreamSynchronize(lane.stream));
                        std::memcpy(row.data(), lane.host_words + 8, WF_SEGS * 4);
                        live = 0;
                        for (uint32_t v : row) live += v;
                        if (live == 0) break;
                        if (drain_kernel && (size_t)live <= queue_cap && live <= drain_below) {
                            wf.cnt_in = counts + (size_t)((level - 1) % rows) * WF_SEGS; wf.list_in = lists + (size_t)((level + 1) & 1) * list_entries;
                            hipLaunchKernelGGL(k_wf_to_queue, dim3((unsigned)std::min<long long>(1024, (live + 255) / 256)), dim3(256), 0, lane.stream, wf, ps);
                            ps.drain = 1;
                            const long long drain_wgs = std::min<long long>((long long)l.grid.x, (live + WG_THREADS - 1) / WG_THREADS);
                            const int rc_d = launch(drain_kernel, dim3((unsigned)std::max<long long>(1, drain_wgs)));
                            ps.drain = 0;
                            if (rc_d) return rc_d;
                            break;
                        }
                        // the rows the next batch counts in: zero again (all but the one that holds the live counts)
                        for (int b = 0; b < more_levels; ++b) HIP_TRY(hipMemsetAsync(counts + (size_t)((level + b) % rows) * WF_SEGS, 0, WF_SEGS * 4, lane.stream));
                    }
                }
                ps.queue = queue_before; ps.queue_cap = cap_before;
                HIP_TRY(hipGetLastError());
                return RSX_OK;
            };
#define WF_LEVEL(...) reinterpret_cast<const void *>(k_wf_level<__VA_ARGS__>)
#define PATH_KERNEL(...) reinterpret_cast<const void *>(k_render_trace_path<__VA_ARGS__>)
            // the staged forms (scenes of a few dozen primitives) come with and without the mesh walk: see k_render_trace_path, MESHES
#define PATH_STAGED(c, m, v, qd) (nomesh ? PATH_KERNEL(c, m, v, false, true, qd, false) : PATH_KERNEL(c, m, v, false, true, qd))
#define WF_STAGED(c, m, v) (nomesh ? WF_LEVEL(c, m, v, true, false) : WF_LEVEL(c, m, v, true))
            static const bool path_meshes_always = [] { const char *e = std::getenv("RSX_PATH_MESHES"); return e && std::atoi(e) != 0; }();   // (A/B and the parity test of the two forms)
            const bool nomesh = scene->d.n_meshes == 0 && !path_meshes_always && !ctx->path_general;       // (rsx_scene_create: a mesh primitive needs a mesh record)
            if (two_pass_csg) {
                const bool vols = rp.n_vol_emitters > 0;
                // (the mask is left zeroed by the k_accumulate of the lane's previous pass over as many units, the second ticket set too)
                if (lane.redo_zeroed != n_units_all) HIP_TRY(hipMemsetAsync(lane.redo, 0, (size_t)n_units_all * 8, lane.stream));
                lane.redo_zeroed = 0;
                const bool staged = rp.prims_lds > 0;
                if (use_wf && !rewalk && !wf_off) {
                    if (!vols) rc = staged ? run_staged(WF_STAGED(true, 1, false), PATH_STAGED(true, 1, false, true)) : run_staged(WF_LEVEL(true, 1, false, false), nullptr);
                    else rc = staged ? run_staged(WF_STAGED(true, 1, true), PATH_STAGED(true, 1, true, true)) : run_staged(WF_LEVEL(true, 1, true, false), nullptr);
                    if (rc) return rc;
                } else
                if ((rc = launch_drained(!vols ? (staged ? PATH_STAGED(true, 1, false, false) : PATH_KERNEL(true, 1, false)) : rewalk ? PATH_KERNEL(true, 1, true, true) :
                                         staged ? PATH_STAGED(true, 1, true, false) : PATH_KERNEL(true, 1, true),
                                         !staged || rewalk ? nullptr : !vols ? PATH_STAGED(true, 1, false, true) : PATH_STAGED(true, 1, true, true), l.grid))) return rc;
                PathState *const queue = ps.queue;                             // (the redo pass walks the same work lists: with the second ticket set)
                ps.queue = nullptr;                                            // (the redo pass — usually a handful of paths — keeps them)
                // The redo pass carries the stream merge: one wave per SIMD, a whole register file per wave — each of its workgroups has to
                // wait for a CU whose SIMDs have drained completely, behind the persistent workgroups of the other slices in flight, and
                // the slice's k_accumulate waits behind it. It usually finds a handful of paths (units are strided over whatever grid
                // there is), so among overlapping slices it brings a small grid: few places to wait for. RSX_REDO_GRID pins the size.
                static const int redo_grid_env = [] { const char *e = std::getenv("RSX_REDO_GRID"); return e ? std::atoi(e) : 0; }();
                const unsigned redo_wgs = redo_grid_env > 0 ? (unsigned)redo_grid_env : deferred ? 32u : l.grid.x;
                if ((rc = launch(rewalk ? PATH_KERNEL(true, 2, true, true) : PATH_KERNEL(true, 2), dim3(std::min(l.grid.x, redo_wgs)), 1))) return rc;
                ps.queue = queue;
            } else if (scene->has_csg) { if ((rc = launch_drained(rewalk ? PATH_KERNEL(true, 0, true, true) : PATH_KERNEL(true), nullptr, l.grid))) return rc; }
            else if (use_wf && !rewalk && !wf_off) {
                const bool staged = rp.prims_lds > 0;
                if (rp.n_vol_emitters == 0) rc = staged ? run_staged(WF_STAGED(false, 0, false), PATH_STAGED(false, 0, false, true)) : run_staged(WF_LEVEL(false, 0, false, false), nullptr);
                else rc = staged ? run_staged(WF_STAGED(false, 0, true), PATH_STAGED(false, 0, true, true)) : run_staged(WF_LEVEL(false, 0, true, false), nullptr);
                if (rc) return rc;
            }
            else if (rp.n_vol_emitters == 0) {            // nothing with a volume contribution (clear glass counts as nothing): the form without the world.contains() pass
                if ((rc = launch_drained(rp.prims_lds > 0 ? PATH_STAGED(false, 0, false, false) : PATH_KERNEL(false, 0, false),
                                         rp.prims_lds > 0 ? PATH_STAGED(false, 0, false, true) : nullptr, l.grid))) return rc;
            }
            else if ((rc = launch_drained(rewalk ? PATH_KERNEL(false, 0, true, true) : rp.prims_lds > 0 ? PATH_STAGED(false, 0, true, false) : PATH_KERNEL(false),
                                          !rewalk && rp.prims_lds > 0 ? PATH_STAGED(false, 0, true, true) : nullptr, l.grid))) return rc;
#undef PATH_STAGED
#undef WF_STAGED
#undef PATH_KERNEL
#undef WF_LEVEL
            HIP_TRY(hipGetLastError());
            // has_scatter: the arena can run out; volumes: a point can lie in more of them than the fast form keeps
            if (deferred) break;                             // (the flags are read when the caller collects; k_accumulate leaves a failed pass out of the frame)
            if (!(has_scatter && !std::getenv("RSX_PATH_ARENA")) && !(rp.n_vol_emitters > PATH_VOL_OVERLAP && !rewalk)) break;
            HIP_TRY(hipMemcpyAsync(lane.host_words, lane.overflow, sizeof(unsigned int), hipMemcpyDeviceToHost, lane.stream));   // (pinned words)
            HIP_TRY(hipStreamSynchronize(lane.stream));
            const unsigned int flags = *reinterpret_cast<const unsigned int *>(lane.host_words);
            const bool grow = (flags & 1u) && has_scatter && !std::getenv("RSX_PATH_ARENA"), again = (flags & 4u) && !rewalk, refile = (flags & 8u) && !wf_off;
            static const bool path_debug = std::getenv("RSX_PATH_DEBUG") != nullptr;
            if (path_debug) std::fprintf(stderr, "rsx path pass: attempt %d flags %u S %zu arena %zu passes %d\n", attempt, flags, S, arena_blocks, (int)passes);
            if (!grow && !again && !refile) break;
            if (refile) wf_off = true;
            if (again) rewalk = true;
            if (grow) {
                const size_t bigger = arena_blocks * 4, pool_bytes = (S + bigger) * PATH_BLOCK * sizeof(PathTerm);
                if (attempt >= 5 || S + bigger >= ((size_t)1 << 31) || pool_bytes > ((size_t)96 << 30)) break;   // reported below
                arena_blocks = bigger;
                if ((rc = lane_buffer(lane.terms, lane.terms_bytes, pool_bytes))) return rc;
            }
            HIP_TRY(hipMemsetAsync(lane.overflow, 0, 64, lane.stream));
            if ((rc = reset_ticket(lane))) return rc;
        }
    } else if (two_pass_csg) {
        // fast pass (state-free CSG evaluator, several waves per SIMD), then the redo pass for the rays it could not finish
        lane.redo_zeroed = 0;                              // (the fast pass writes every unit's mask: whatever a path pass left zeroed is not zero any more)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_render_trace<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_render_trace<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
        if (use_packet) {
            if ((rc = ensure_camera_relative(scene, desc->camera, lane.stream))) return rc;
            const void *kernel = reinterpret_cast<const void *>(k_render_trace<true, 1, 1, 0, true>);
            HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
            Sample *records = static_cast<Sample *>(lane.samples);
            void *args[] = {(void *)&scene->d, (void *)&rp, (void *)&records, (void *)&lane.ticket, (void *)&fz};
            HIP_TRY(hipLaunchKernel(kernel, pgrid, dim3(WG_THREADS), args, plds, lane.stream));
        } else
        hipLaunchKernelGGL((k_render_trace<true, 1>), l.grid, dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(lane.samples), lane.ticket, fz);
        hipLaunchKernelGGL((k_render_trace<true, 2>), dim3((unsigned)ctx->n_cus), dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(lane.samples), lane.ticket, fz);
    } else if (scene->has_csg) hipLaunchKernelGGL(k_render_trace<true>, l.grid, dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(lane.samples), lane.ticket, fz);
    else if (use_packet) {
        if ((rc = ensure_camera_relative(scene, desc->camera, lane.stream))) return rc;
        const void *kernel = fused ? (passes > 1 ? reinterpret_cast<const void *>(k_render_trace<false, 0, 1, 2, true>) : reinterpret_cast<const void *>(k_render_trace<false, 0, 1, 1, true>))
                                   : reinterpret_cast<const void *>(k_render_trace<false, 0, 1, 0, true>);
        Sample *records = fused ? nullptr : static_cast<Sample *>(lane.samples);
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
        void *args[] = {(void *)&scene->d, (void *)&rp, (void *)&records, (void *)&lane.ticket, (void *)&fz};
        HIP_TRY(hipLaunchKernel(kernel, pgrid, dim3(WG_THREADS), args, plds, lane.stream));
    }
    else if (fused) {
        if (desc->spp > RSX_COHERENT_MIN_SPP) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_render_trace<false, 0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
            hipLaunchKernelGGL((k_render_trace<false, 0, 1, 1>), l.grid, dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(nullptr), lane.ticket, fz);
        } else {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_render_trace<false, 0, RSX_STAGE_MIN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
            hipLaunchKernelGGL((k_render_trace<false, 0, RSX_STAGE_MIN, 1>), l.grid, dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(nullptr), lane.ticket, fz);
        }
    }
    else if (desc->spp > RSX_COHERENT_MIN_SPP) {            // coherent waves (several samples of a pixel side by side): always stage big leaves
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_render_trace<false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
        hipLaunchKernelGGL((k_render_trace<false, 0, 1>), l.grid, dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(lane.samples), lane.ticket, fz);
    } else hipLaunchKernelGGL(k_render_trace<false>, l.grid, dim3(WG_THREADS), l.lds, lane.stream, scene->d, rp, static_cast<Sample *>(lane.samples), lane.ticket, fz);
    HIP_TRY(hipGetLastError());
    HP_MARK(1)
    if (timed) HIP_TRY(hipEventRecord(re[1], lane.stream));
    if (pipelined) HIP_TRY(hipEventRecord(lane.traced, lane.stream));
    if (want_order) {
        // longest-first work list for this lane's NEXT pass over the same units, sorted while this pass's waves drain
        // (a path pass that runs alone: the one-workgroup sort — 0.54 ms for the 262 144 units of a Cornell-box pass — goes to a side stream
        // and runs beside the replay of the term lists instead of in front of it; the lane's next pass waits for it. RSX_SIDE_SORT=0: in line)
        static const bool side_sort = [] { const char *e = std::getenv("RSX_SIDE_SORT"); return !e || std::atoi(e) != 0; }();
        hipStream_t sort_on = lane.stream;
        if (side_sort && has_vol && !pipelined && !deferred && !use_wf) {
            if (!ctx->sort_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->sort_stream, hipStreamNonBlocking));
            if (!lane.sort_from) { HIP_TRY(hipEventCreateWithFlags(&lane.sort_from, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&lane.sorted, hipEventDisableTiming)); }
            HIP_TRY(hipEventRecord(lane.sort_from, lane.stream));
            HIP_TRY(hipStreamWaitEvent(ctx->sort_stream, lane.sort_from, 0));
            sort_on = ctx->sort_stream;
        }
        hipLaunchKernelGGL(k_order_units, dim3(1), dim3(1024), 0, sort_on, lane.unit_cost, lane.unit_order, lane.n_work, order_n, order_tiles_x, (int)desc->spp, has_vol ? 1 : 0);
        HIP_TRY(hipGetLastError());
        if (sort_on != lane.stream) { HIP_TRY(hipEventRecord(lane.sorted, sort_on)); lane.sort_pending = true; }
        lane.order_units = order_n;
        if (has_vol) lane.cost_zeroed = order_n;
    }
    if (pipelined) HIP_TRY(hipStreamWaitEvent(ctx->stream, lane.traced, 0));   // the merge runs on the ctx stream, in call order

    AccumParams ap;
    ap.samples = static_cast<const Sample *>(lane.samples);
    ap.tables = static_cast<const double *>(d_tab);
    ap.tasks = rp.tasks;
    ap.n_tasks = desc->n_tasks;
    std::memcpy(ap.rect, desc->rect, sizeof(ap.rect));
    ap.ny = desc->camera.ny; ap.bins = desc->bins; ap.spp = desc->spp / passes; ap.power = desc->power;
    ap.passes = passes; ap.pad_passes = 0;
    ap.sensitivity = desc->camera.sensitivity;
    ap.mean = h_mean ? static_cast<double *>(d_mean) : nullptr;
    ap.variance = h_mean ? static_cast<double *>(d_var) : nullptr;
    ap.fmean = fmean; ap.fvar = fvar; ap.fn = fn;
    ap.frame_bins = frame_bins; ap.slice_offset = slice_offset;
    ap.ticket = lane.ticket;
    ap.pool = has_vol ? static_cast<const PathTerm *>(lane.terms) : nullptr;
    ap.tail = has_vol ? static_cast<const int32_t *>(lane.tail) : nullptr;
    ap.n_records = (long long)S;
    ap.roulette_norm = 1 / (1 - desc->ray_extinction_prob);      // ray.pyx:388
    ap.abort_flags = deferred ? lane.overflow : nullptr;
    ap.consts = ctx->acc_consts;
    ap.zero = nullptr; ap.zero_n = 0;
    if (two_pass_csg && has_vol && !fused) {                // the redo mask of this pass: consumed; zero for the lane's next one
        ap.zero = static_cast<unsigned int *>(lane.redo); ap.zero_n = 2 * n_units_all;
        lane.redo_zeroed = n_units_all;
    }
    lane.ticket_armed = true;
    const long long total = (long long)T * (long long)B;
    HP_MARK(2)
    if (timed) HIP_TRY(hipEventRecord(re[3], ctx->stream));
    ap.n_tables = desc->n_tables;
    // LDS of the staged accumulate kernel: the Welford reciprocals and, when they fit next to them, the spectral tables (a 512-bin
    // slice with twenty materials does not: the kernel then reads the tables from global memory, L1 / L2 resident)
    const size_t rcp_lds = (ap.spp <= ACC_RCP_TABLE_MAX ? (size_t)ap.spp + 2 : 2) * 8, tab_lds = (size_t)std::max(1, desc->n_tables) * B * 8;
    // (path passes also keep ACC_PATH_CHUNK sample values per thread there: k_accumulate's flattened list walk)
    const size_t xs_lds = has_vol && !h_xyz && ap.spp >= 4 ? (size_t)ACC_PATH_CHUNK * 256 * 8 : 0;
    ap.tables_in_lds = rcp_lds + tab_lds + xs_lds <= 60 * 1024 ? 1 : 0;
    const size_t acc_lds = rcp_lds + (ap.tables_in_lds ? tab_lds : 8) + xs_lds;
    const dim3 acc_grid((unsigned)((total + 255) / 256));
    if (fused) {                                            // the trace kernel merged its own samples; its tickets are re-armed by the next launch
        lane.ticket_armed = false;
        if (timed) HIP_TRY(hipEventRecord(re[2], ctx->stream));
        HIP_TRY(hipEventRecord(ctx->gate[(size_t)(ctx->render_calls % (long long)ctx->gate.size())], ctx->stream));
        ctx->render_calls++;
        ctx->have_accum = true;
        if (g_hp_on) ++g_hp_calls;
        return RSX_OK;
    }
    bool has_dielectric = false;                            // an absorbing dielectric: only then do the terms need pow() (60 more registers)
    for (int32_t i = 0; has_vol && i < desc->n_materials; ++i) {
        if (desc->materials[i].type != RSX_MAT_DIELECTRIC) continue;
        const double *row = desc->tables + (size_t)desc->materials[i].table * B;
        for (size_t b = 0; b < B; ++b) has_dielectric = has_dielectric || row[b] != 1.0;
    }
    if (h_xyz) {
        const dim3 xyz_grid((unsigned)(((long long)T * 3 + 255) / 256));
        if (has_vol && has_dielectric) hipLaunchKernelGGL((k_accumulate_xyz<2>), xyz_grid, dim3(256), 0, ctx->stream, ap, desc->n_tables, delta_wavelength);
        else if (has_vol) hipLaunchKernelGGL((k_accumulate_xyz<1>), xyz_grid, dim3(256), 0, ctx->stream, ap, desc->n_tables, delta_wavelength);
        else hipLaunchKernelGGL((k_accumulate_xyz<0>), xyz_grid, dim3(256), 0, ctx->stream, ap, desc->n_tables, delta_wavelength);
    } else {
        const int vol = !has_vol ? 0 : has_dielectric ? 2 : 1;
        const bool staged = ap.spp >= 4, in_lds = ap.tables_in_lds != 0;
        // several one-sample path passes per call: the replay with a thread per record, then the merge (k_path_values / k_merge_passes;
        // RSX_RECORD_REPLAY=0: k_accumulate's multi-pass form)
        static const bool record_replay = [] { const char *e = std::getenv("RSX_RECORD_REPLAY"); return !e || std::atoi(e) != 0; }();
        const size_t xs_need = (size_t)total * (size_t)passes * 8;
        if (record_replay && passes > 1 && has_vol && ap.spp == 1 && fmean && xs_need <= ((size_t)8 << 30)) {
            if ((rc = lane_buffer(lane.xs, lane.xs_bytes, xs_need))) return rc;
            const dim3 val_grid((unsigned)(((long long)total * passes + 255) / 256));
            double *xs = static_cast<double *>(lane.xs);
            if (vol == 2) hipLaunchKernelGGL((k_path_values<2>), val_grid, dim3(256), 0, ctx->stream, ap, xs);
            else hipLaunchKernelGGL((k_path_values<1>), val_grid, dim3(256), 0, ctx->stream, ap, xs);
            hipLaunchKernelGGL(k_merge_passes, acc_grid, dim3(256), 0, ctx->stream, ap, static_cast<const double *>(xs));
        } else {
#define ACC(V) (!staged ? reinterpret_cast<const void *>(k_accumulate<false, V>) : in_lds ? reinterpret_cast<const void *>(k_accumulate<true, V>) \
                                                                                           : reinterpret_cast<const void *>(k_accumulate<true, V, false>))
#define ACC_MULTI(V) (!staged ? reinterpret_cast<const void *>(k_accumulate<false, V, false, true>) : in_lds ? reinterpret_cast<const void *>(k_accumulate<true, V, true, true>) \
                                                                                                     : reinterpret_cast<const void *>(k_accumulate<true, V, false, true>))
        const void *kernel = passes > 1 ? (vol == 0 ? ACC_MULTI(0) : vol == 1 ? ACC_MULTI(1) : ACC_MULTI(2)) : vol == 0 ? ACC(0) : vol == 1 ? ACC(1) : ACC(2);
#undef ACC_MULTI
#undef ACC
        void *args[] = {(void *)&ap};
        HIP_TRY(hipLaunchKernel(kernel, acc_grid, dim3(256), args, staged ? acc_lds : 0, ctx->stream));
        }
    }
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(re[2], ctx->stream));
    HIP_TRY(hipEventRecord(ctx->gate[(size_t)(ctx->render_calls % (long long)ctx->gate.size())], ctx->stream));
    if (pipelined) { HIP_TRY(hipEventRecord(lane.merged, ctx->stream)); lane.in_flight = true; }
    ctx->render_calls++;
    ctx->have_accum = true;
    HP_MARK(3)
    if (g_hp_on) ++g_hp_calls;
    if (h_mean) {
        HIP_TRY(hipMemcpyAsync(h_mean, d_mean, T * (h_xyz ? 3 : B) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h_var, d_var, T * (h_xyz ? 3 : B) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (deferred) {                                         // checked by settle_lane: at the lane's next pass or when the caller collects
        lane.check_pending = true;
        lane.pending_call = ctx->deferred_calls++;
        if (ray_count) *ray_count = ~0ULL;                 // "deferred": the count comes with rsx_collect_path_checks
    } else
    if (has_vol) {                                          // a ray that ran out of term slots or segments must not go unnoticed
        const hipStream_t check = solo_path ? lane.stream : ctx->stream;      // (solo_path: the merge on the ctx stream is not waited for)
        unsigned long long *words = lane.host_words;                          // [0] low half: flags, [1]: rays spawned (pinned words)
        HIP_TRY(hipMemcpyAsync(words, lane.overflow, 16, hipMemcpyDeviceToHost, check));
        HIP_TRY(hipStreamSynchronize(check));
        const unsigned int flags = (unsigned int)(words[0] & 0xFFFFFFFFull);
        if (ray_count) *ray_count = words[1];
        if (flags & 1u) return rsx_fail(RSX_EUNSUPPORTED, "render: the path-term arena (%zu blocks of %d terms) ran out; render fewer rays per call or raise RSX_PATH_ARENA", arena_blocks, PATH_BLOCK - 1);
        if (flags & 2u) return rsx_fail(RSX_EUNSUPPORTED, "render: a path crossed more than %d surfaces (limit of this build)", PATH_MAX_SEGMENTS);
        if (flags & 4u) return rsx_fail(RSX_EUNSUPPORTED, "render: more than %d volume emitters overlap at one point (limit of this build)", PATH_VOL_OVERLAP);
        if (flags & 8u) return rsx_fail(RSX_EHIP, "render: a path list of the level-by-level form overflowed (internal error; RSX_WAVEFRONT=0 renders the pass with the one-kernel form)");
    }
    // frame form: asynchronous — the pooled workspace stays alive in the ctx, stream order protects reuse
    return RSX_OK;
}

}  // namespace

extern "C" int rsx_set_path_stages(rsx_ctx *ctx, int32_t mode, int64_t min_paths) {
    if (!ctx || mode < -1 || mode > 3) return rsx_fail(RSX_EINVAL, "rsx_set_path_stages: bad arguments");
    ctx->wf_mode = mode < 0 ? -1 : (mode & 1);
    ctx->path_general = mode >= 2;
    ctx->wf_min_paths = min_paths < 0 ? -1 : (long long)min_paths;
    return RSX_OK;
}

extern "C" int rsx_defer_path_checks(rsx_ctx *ctx, int32_t on) {
    if (!ctx) return rsx_fail(RSX_EINVAL, "rsx_defer_path_checks: null context");
    if (on && !ctx->defer_path) { ctx->deferred_calls = 0; ctx->deferred_failed.clear(); ctx->deferred_error_flags = 0; ctx->deferred_rays = 0; }
    ctx->defer_path = on != 0;
    return RSX_OK;
}

extern "C" int rsx_collect_path_checks(rsx_ctx *ctx, int32_t *failed_calls, int32_t capacity, int32_t *n_failed, uint64_t *ray_count) {
    if (!ctx || !n_failed || capacity < 0 || (capacity > 0 && !failed_calls)) return rsx_fail(RSX_EINVAL, "rsx_collect_path_checks: bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    for (TraceLane &ln : ctx->lanes) if ((rc = settle_lane(ctx, ln))) return rc;
    std::sort(ctx->deferred_failed.begin(), ctx->deferred_failed.end());
    *n_failed = (int32_t)ctx->deferred_failed.size();
    for (int32_t i = 0; i < *n_failed && i < capacity; ++i) failed_calls[i] = ctx->deferred_failed[(size_t)i];
    if (ray_count) *ray_count = ctx->deferred_rays;
    const unsigned int err = ctx->deferred_error_flags;
    const bool truncated = *n_failed > capacity;
    ctx->deferred_calls = 0; ctx->deferred_failed.clear(); ctx->deferred_error_flags = 0; ctx->deferred_rays = 0;
    if (err & 2u) return rsx_fail(RSX_EUNSUPPORTED, "render: a path crossed more than %d surfaces (limit of this build)", PATH_MAX_SEGMENTS);
    if (truncated) return rsx_fail(RSX_EINVAL, "rsx_collect_path_checks: %d passes failed, room for %d", *n_failed, capacity);
    return RSX_OK;
}

extern "C" int rsx_render_pinhole(rsx_scene *scene, const rsx_render_desc *desc, double *mean, double *variance, uint64_t *ray_count) {
    if (!mean || !variance) return rsx_fail(RSX_EINVAL, "rsx_render_pinhole: null output");
    return render(scene, desc, mean, variance, nullptr, nullptr, nullptr, 0, 0, ray_count);
}

extern "C" int rsx_render_pinhole_xyz(rsx_scene *scene, const rsx_render_desc *desc, const double *resampled_xyz, double delta_wavelength,
                                      double *mean, double *variance, uint64_t *ray_count) {
    if (!mean || !variance || !resampled_xyz) return rsx_fail(RSX_EINVAL, "rsx_render_pinhole_xyz: null argument");
    return render(scene, desc, mean, variance, nullptr, nullptr, nullptr, 0, 0, ray_count, resampled_xyz, delta_wavelength);
}

extern "C" int rsx_render_pinhole_frame(rsx_scene *scene, const rsx_render_desc *desc, double *frame_mean, double *frame_variance,
                                        int32_t *frame_samples, int32_t frame_bins, int32_t slice_offset, uint64_t *ray_count) {
    if (!frame_mean || !frame_variance || !frame_samples) return rsx_fail(RSX_EINVAL, "rsx_render_pinhole_frame: null frame");
    if (!desc || slice_offset < 0 || slice_offset + desc->bins > frame_bins) return rsx_fail(RSX_EINVAL, "rsx_render_pinhole_frame: slice outside frame");
    return render(scene, desc, nullptr, nullptr, frame_mean, frame_variance, frame_samples, frame_bins, slice_offset, ray_count);
}

extern "C" int rsx_frame_combine_dev(rsx_ctx *ctx, int64_t n, double *mean_a, double *var_a, int32_t *n_a, const double *mean_b,
                                     const double *var_b, const int32_t *n_b) {
    if (!ctx || n < 0 || !mean_a || !var_a || !n_a || !mean_b || !var_b || !n_b) return rsx_fail(RSX_EINVAL, "rsx_frame_combine_dev: bad arguments");
    if (n == 0) return RSX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_frame_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, mean_a, var_a, n_a, mean_b, var_b, n_b);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    return RSX_OK;
}

// ---------------------------------------------------------------------------------------------------
// device-side known-answer entry points (dev_selftest.hpp)
// ---------------------------------------------------------------------------------------------------
namespace {
struct DevBuf {                        // small RAII device buffer for the selftests (not on any hot path)
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? RSX_OK : rsx_fail(RSX_ENOMEM, "selftest: out of device memory"); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};
}  // namespace

extern "C" int rsx_selftest_aabb(rsx_ctx *ctx, int64_t n, const double *lower, const double *upper, const double *origin, const double *direction,
                                 double *result, uint64_t *mismatches) {
    if (!ctx || n < 0 || !lower || !upper || !origin || !direction || !result || !mismatches) return rsx_fail(RSX_EINVAL, "rsx_selftest_aabb: bad arguments");
    *mismatches = 0;
    if (n == 0) return RSX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    DevBuf in, out, mm;
    int rc;
    if ((rc = in.alloc(N * 96)) || (rc = out.alloc(N * 24)) || (rc = mm.alloc(8))) return rc;
    double *d = in.as<double>();
    const double *src[4] = {lower, upper, origin, direction};
    for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpyAsync(d + 3 * N * k, src[k], N * 24, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(mm.p, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_selftest_aabb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, d, d + 3 * N, d + 6 * N, d + 9 * N,
                       out.as<double>(), mm.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(result, out.p, N * 24, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long m = 0;
    HIP_TRY(hipMemcpyAsync(&m, mm.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *mismatches = m;
    return RSX_OK;
}

extern "C" int rsx_selftest_camera_rays(rsx_ctx *ctx, const rsx_render_desc *desc, double *rays) {
    if (!ctx || !desc || !rays || desc->n_tasks < 0 || desc->spp < 1) return rsx_fail(RSX_EINVAL, "rsx_selftest_camera_rays: bad arguments");
    if (desc->rng_mode == RSX_RNG_STREAM && !desc->uniforms) return rsx_fail(RSX_EINVAL, "rsx_selftest_camera_rays: RSX_RNG_STREAM needs uniforms");
    if (!desc->tasks) return rsx_fail(RSX_EINVAL, "rsx_selftest_camera_rays: needs an explicit task list");
    if (desc->n_tasks == 0) return RSX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t T = (size_t)desc->n_tasks, S = T * (size_t)desc->spp;
    DevBuf tasks, uni, out;
    int rc;
    if ((rc = tasks.alloc(T * 8)) || (rc = uni.alloc(S * 16)) || (rc = out.alloc(S * 56))) return rc;
    HIP_TRY(hipMemcpyAsync(tasks.p, desc->tasks, T * 8, hipMemcpyHostToDevice, ctx->stream));
    if (desc->rng_mode == RSX_RNG_STREAM) HIP_TRY(hipMemcpyAsync(uni.p, desc->uniforms, S * 16, hipMemcpyHostToDevice, ctx->stream));
    RenderParams rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.cam = desc->camera;
    camera_origin(rp.cam, rp.origin);
    rp.tasks = tasks.as<int32_t>();
    rp.uniforms = desc->rng_mode == RSX_RNG_STREAM ? uni.as<double>() : nullptr;
    rp.n_tasks = desc->n_tasks;
    rp.spp = desc->spp; rp.rng_mode = desc->rng_mode; rp.seed = desc->seed; rp.sample_offset = desc->sample_offset;
    hipLaunchKernelGGL(k_selftest_camera, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, rp, out.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rays, out.p, S * 56, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSX_OK;
}

extern "C" int rsx_selftest_welford(rsx_ctx *ctx, int64_t n_chains, int32_t spp, const double *x, double *mean, double *variance) {
    if (!ctx || n_chains < 0 || spp < 1 || !x || !mean || !variance) return rsx_fail(RSX_EINVAL, "rsx_selftest_welford: bad arguments");
    if (n_chains == 0) return RSX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n_chains, S = N * (size_t)spp;
    DevBuf dx, ds, dm, dv, one;
    int rc;
    if ((rc = dx.alloc(S * 8)) || (rc = ds.alloc(S * sizeof(Sample))) || (rc = dm.alloc(N * 16)) || (rc = dv.alloc(N * 16)) || (rc = one.alloc(8))) return rc;
    HIP_TRY(hipMemcpyAsync(dx.p, x, S * 8, hipMemcpyHostToDevice, ctx->stream));
    const double unit = 1.0;
    HIP_TRY(hipMemcpyAsync(one.p, &unit, 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_fill_samples, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, (long long)S, dx.as<double>(), ds.as<Sample>());
    AccumParams ap;
    std::memset(&ap, 0, sizeof(ap));
    ap.samples = ds.as<Sample>(); ap.tables = one.as<double>(); ap.n_tasks = n_chains;
    ap.rect[0] = 0; ap.rect[1] = 0; ap.rect[2] = 1; ap.rect[3] = (int32_t)n_chains;      // one column of n_chains pixels: slot order = task order
    ap.consts = ctx->acc_consts;
    ap.ny = (int32_t)n_chains; ap.bins = 1; ap.spp = spp; ap.passes = 1; ap.n_tables = 1; ap.tables_in_lds = 1; ap.sensitivity = 1.0; ap.roulette_norm = 1.0;
    // both instantiations the render path uses: the lean one (few samples per pixel) and the staged one (LDS tables, batched loads)
    for (int staged = 0; staged < 2; ++staged) {
        ap.mean = dm.as<double>() + (staged ? N : 0); ap.variance = dv.as<double>() + (staged ? N : 0);
        const dim3 grid((unsigned)((N + 255) / 256));
        if (staged) hipLaunchKernelGGL((k_accumulate<true, 0>), grid, dim3(256), ((size_t)std::min(spp, ACC_RCP_TABLE_MAX) + 2 + 1) * 8, ctx->stream, ap);
        else hipLaunchKernelGGL((k_accumulate<false, 0>), grid, dim3(256), 0, ctx->stream, ap);
        HIP_TRY(hipGetLastError());
    }
    std::vector<double> hm(2 * N), hv(2 * N);
    HIP_TRY(hipMemcpyAsync(hm.data(), dm.p, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(hv.data(), dv.p, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < N; ++i)
        if (std::memcmp(&hm[i], &hm[N + i], 8) || std::memcmp(&hv[i], &hv[N + i], 8))
            return rsx_fail(RSX_EHIP, "rsx_selftest_welford: the two k_accumulate instantiations disagree on chain %zu", i);
    std::memcpy(mean, hm.data(), N * 8);
    std::memcpy(variance, hv.data(), N * 8);
    return RSX_OK;
}

extern "C" int rsx_selftest_math(rsx_ctx *ctx, int32_t op, int64_t n, const double *a, const double *b, double *out0, double *out1) {
    if (!ctx || n < 0 || op < 0 || op > 2 || !a || !out0 || (op == 0 && !b) || (op == 1 && !out1)) return rsx_fail(RSX_EINVAL, "rsx_selftest_math: bad arguments");
    if (n == 0) return RSX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    DevBuf buf;
    int rc;
    if ((rc = buf.alloc(N * 32))) return rc;
    double *d = buf.as<double>();
    HIP_TRY(hipMemcpyAsync(d, a, N * 8, hipMemcpyHostToDevice, ctx->stream));
    if (b) HIP_TRY(hipMemcpyAsync(d + N, b, N * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)op, (long long)n, d, d + N, d + 2 * N, d + 3 * N);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out0, d + 2 * N, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out1) HIP_TRY(hipMemcpyAsync(out1, d + 3 * N, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSX_OK;
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU frame exchange (RCCL over xGMI)
// ---------------------------------------------------------------------------------------------------
#include "rsx_comm.hpp"
