// rsx_host.cpp — host-side builders of librsx (no device code): SAH KD-tree construction, mesh
// preprocessing, MT19937-64 stream. These are CPU work in the reference too; they are restated here in
// C++ with their own design (sorted lower/upper edge arrays merged on the fly, task-parallel subtree
// construction spliced into pre-order) and must reproduce the reference's results bit for bit:
//   KD-tree        raysect/core/math/spatial/kdtree3d.pyx:126-486
//   mesh prep      raysect/primitive/mesh/mesh.pyx:363-504, 835-859
//   MT19937-64     raysect/core/math/random.pyx:99-265
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include <cstdio>
#include <sched.h>

#include "../../include/rsx.h"
#include "rsx_internal.h"

// The host builders' OpenMP team. Two facts of the machines this runs on shape it (tools/r6_world_stalls.py, profiles/r06_world_stalls.txt):
// a container sees every hardware thread of the node (256) but may only USE its cgroup's CPU quota (cpu.max: 16 cores per 100 ms period),
// and LLVM's OpenMP runtime keeps the workers of a finished parallel region spinning for KMP_BLOCKTIME = 200 ms. A KD build on 256 threads
// therefore left 255 spinning threads that burnt the quota of the next two or three 100-ms periods in a few milliseconds each — and the
// kernel's bandwidth control froze the WHOLE process for the rest of every such period: 60 - 90 ms stalls of a render loop, the device
// idle, two or three times after every new scene (DESIGN 8.7 of round 5). So: no more threads than the process can run, and workers
// that go to sleep when their region ends.
extern "C" void kmp_set_blocktime(int) __attribute__((weak));     // (LLVM libomp; absent from other runtimes)
namespace {
int host_team_size() {
    static const int team = [] {
        int n = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
        if (n <= 0) n = 1;
        long long quota = -1, period = 100000;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (std::fscanf(f, "%31s %lld", q, &period) >= 1 && q[0] != 'm') quota = std::atoll(q);
            std::fclose(f);
        } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
            std::fclose(g);
            if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(h, "%lld", &period) != 1) period = 100000; std::fclose(h); }
        }
        if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
        if (const char *e = std::getenv("RSX_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) n = v; }
        return n;
    }();
    static const bool spin = [] { const char *e = std::getenv("RSX_HOST_SPIN"); return e && std::atoi(e) != 0; }();   // (1: the runtime's own waiting policy — reproduces the fault)
    if (kmp_set_blocktime && !spin) kmp_set_blocktime(0);                                // (per calling thread in libomp)
    return team;
}
}  // namespace

extern "C" int rsx_host_team_size(void) { return host_team_size(); }

namespace {

struct Item {
    int32_t id;
    double lo[3], hi[3];
};

struct Subtree {
    std::vector<rsx_kdnode> nodes;
    std::vector<int32_t> items;
};

struct BuildParams {
    int32_t max_depth, min_items;
    double hit_cost, empty_bonus;
};

inline double box_area(const double *lo, const double *hi) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}

inline int longest_axis(const double *lo, const double *hi) {
    double e[3];
    for (int k = 0; k < 3; ++k) { e[k] = hi[k] - lo[k]; if (!(e[k] > 0.0)) e[k] = 0.0; }
    int axis = 0;
    if (e[1] > e[axis]) axis = 1;
    if (e[2] > e[axis]) axis = 2;
    return axis;
}

void emit_leaf(Subtree &out, const std::vector<Item> &items) {
    rsx_kdnode nd;
    nd.type = -1;
    nd.count = (int32_t)items.size();
    nd.u.leaf.first_item = (int32_t)out.items.size();
    nd.u.leaf.pad = 0;
    out.nodes.push_back(nd);
    for (const Item &it : items) out.items.push_back(it.id);
}

// Sweep over candidate planes on one axis. The reference sorts 2n (value, is_upper) edges with lower edges
// ahead of upper edges at equal value; two independently sorted arrays merged with "<=" give the same
// sequence. Counts are updated around the cost evaluation exactly as kdtree3d.pyx:240-278 does.
bool sweep_axis(const std::vector<Item> &items, int axis, const double *lo, const double *hi, const BuildParams &bp,
                double recip_area, double &best_cost, double &best_split, std::vector<double> &lows, std::vector<double> &highs) {
    const size_t n = items.size();
    lows.resize(n);
    highs.resize(n);
    for (size_t i = 0; i < n; ++i) { lows[i] = items[i].lo[axis]; highs[i] = items[i].hi[axis]; }
    std::sort(lows.begin(), lows.end());
    std::sort(highs.begin(), highs.end());

    bool found = false;
    int32_t below = 0, above = (int32_t)n;
    size_t il = 0, ih = 0;
    double side_lo[3] = {lo[0], lo[1], lo[2]}, side_hi[3] = {hi[0], hi[1], hi[2]};
    while (il < n || ih < n) {
        const bool take_low = il < n && (ih >= n || lows[il] <= highs[ih]);
        const double split = take_low ? lows[il] : highs[ih];
        if (!take_low) --above;
        if (lo[axis] < split && split < hi[axis]) {
            side_hi[axis] = split;
            const double area_below = box_area(lo, side_hi);
            side_hi[axis] = hi[axis];
            side_lo[axis] = split;
            const double area_above = box_area(side_lo, hi);
            side_lo[axis] = lo[axis];
            double bonus = 1.0;
            if (below == 0 || above == 0) bonus -= bp.empty_bonus;
            const double cost = 1 + bonus * (area_below * below + area_above * above) * recip_area * bp.hit_cost;
            if (cost < best_cost) { best_cost = cost; best_split = split; found = true; }
        }
        if (take_low) { ++below; ++il; } else { ++ih; }
    }
    return found;
}

void splice(Subtree &dst, const Subtree &src) {
    const int32_t node_base = (int32_t)dst.nodes.size(), item_base = (int32_t)dst.items.size();
    for (rsx_kdnode nd : src.nodes) {
        if (nd.type < 0) nd.u.leaf.first_item += item_base; else nd.count += node_base;
        dst.nodes.push_back(nd);
    }
    dst.items.insert(dst.items.end(), src.items.begin(), src.items.end());
}

void build_node(Subtree &out, std::vector<Item> &items, const double *lo, const double *hi, int depth, const BuildParams &bp) {
    const size_t n = items.size();
    if (depth == bp.max_depth || (int64_t)n <= bp.min_items) { emit_leaf(out, items); return; }

    double best_cost = (double)(int32_t)n * bp.hit_cost, best_split = 0;
    int best_axis = -1;
    const double recip_area = 1.0 / box_area(lo, hi);
    const int first = longest_axis(lo, hi);
    {
        std::vector<double> lows, highs;
        for (int a = 0; a < 3 && best_axis < 0; ++a) {      // other axes only when no split at all was found
            const int axis = (first + a) % 3;
            if (sweep_axis(items, axis, lo, hi, bp, recip_area, best_cost, best_split, lows, highs)) best_axis = axis;
        }
    }
    if (best_axis < 0) { emit_leaf(out, items); return; }

    std::vector<Item> below, above;
    below.reserve(n);
    above.reserve(n);
    for (const Item &it : items) {                          // split plane belongs to the upper node
        if (it.lo[best_axis] < best_split) below.push_back(it);
        if (it.hi[best_axis] > best_split) above.push_back(it);
    }
    std::vector<Item>().swap(items);                        // parent's list is dead from here on

    double below_hi[3] = {hi[0], hi[1], hi[2]}, above_lo[3] = {lo[0], lo[1], lo[2]};
    below_hi[best_axis] = best_split;
    above_lo[best_axis] = best_split;

    const size_t self = out.nodes.size();
    rsx_kdnode nd;
    nd.type = best_axis;
    nd.count = 0;
    nd.u.split = best_split;
    out.nodes.push_back(nd);

    const bool parallel = n > 32768 && depth < 8;
    if (parallel) {
        Subtree sub_below, sub_above;
#pragma omp task shared(sub_below, below, below_hi) firstprivate(depth)
        build_node(sub_below, below, lo, below_hi, depth + 1, bp);
#pragma omp task shared(sub_above, above, above_lo) firstprivate(depth)
        build_node(sub_above, above, above_lo, hi, depth + 1, bp);
#pragma omp taskwait
        splice(out, sub_below);
        out.nodes[self].count = (int32_t)out.nodes.size();  // pre-order: upper child follows the whole lower subtree
        splice(out, sub_above);
    } else {
        build_node(out, below, lo, below_hi, depth + 1, bp);
        out.nodes[self].count = (int32_t)out.nodes.size();
        build_node(out, above, above_lo, hi, depth + 1, bp);
    }
}

}  // namespace

struct rsx_kd {
    Subtree tree;
    int32_t max_depth;
    double lower[3], upper[3];
};

extern "C" int rsx_kd_build(const double *aabbs, int32_t n, int32_t max_depth, int32_t min_items, double hit_cost,
                            double empty_bonus, rsx_kd **out) {
    if (!out || n < 0 || (n > 0 && !aabbs)) return rsx_fail(RSX_EINVAL, "rsx_kd_build: bad arguments");
    if (empty_bonus < 0.0 || empty_bonus > 1.0)
        return rsx_fail(RSX_EINVAL, "The empty_bonus cost modifier must lie in the range [0.0, 1.0].");
    BuildParams bp;
    bp.empty_bonus = empty_bonus;
    bp.max_depth = std::max(0, max_depth);
    bp.min_items = std::max(1, min_items);
    bp.hit_cost = std::max(1.0, hit_cost);
    if (bp.max_depth == 0) bp.max_depth = (int32_t)std::ceil(8 + 1.3 * std::log((double)n));

    rsx_kd *kd = new rsx_kd();
    kd->max_depth = bp.max_depth;
    const double inf = std::numeric_limits<double>::infinity();
    for (int k = 0; k < 3; ++k) { kd->lower[k] = inf; kd->upper[k] = -inf; }
    std::vector<Item> items((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        items[i].id = i;
        for (int k = 0; k < 3; ++k) {
            items[i].lo[k] = aabbs[6 * (size_t)i + k];
            items[i].hi[k] = aabbs[6 * (size_t)i + 3 + k];
            kd->lower[k] = std::min(kd->lower[k], items[i].lo[k]);
            kd->upper[k] = std::max(kd->upper[k], items[i].hi[k]);
        }
    }
#pragma omp parallel num_threads(host_team_size())
#pragma omp single
    build_node(kd->tree, items, kd->lower, kd->upper, 0, bp);
    *out = kd;
    return RSX_OK;
}

extern "C" int rsx_kd_info(const rsx_kd *kd, rsx_kdtree *view) {
    if (!kd || !view) return rsx_fail(RSX_EINVAL, "rsx_kd_info: null argument");
    view->nodes = kd->tree.nodes.data();
    view->items = kd->tree.items.data();
    view->n_nodes = (int32_t)kd->tree.nodes.size();
    view->n_items = (int32_t)kd->tree.items.size();
    view->max_depth = kd->max_depth;
    view->pad = 0;
    std::memcpy(view->lower, kd->lower, sizeof(kd->lower));
    std::memcpy(view->upper, kd->upper, sizeof(kd->upper));
    return RSX_OK;
}

extern "C" void rsx_kd_free(rsx_kd *kd) { delete kd; }

extern "C" int64_t rsx_kd_serialise(const rsx_kd *kd, int32_t min_items, double hit_cost, double empty_bonus, uint8_t *out,
                                    int64_t capacity) {
    if (!kd) return rsx_fail(RSX_EINVAL, "rsx_kd_serialise: null tree");
    int64_t need = 4 + 4 + 8 + 8 + 48 + 4;
    for (const rsx_kdnode &nd : kd->tree.nodes) need += nd.type < 0 ? 8 + 4 * (int64_t)nd.count : 16;
    if (!out || capacity < need) return need;
    uint8_t *p = out;
    auto put = [&p](const void *src, size_t bytes) { std::memcpy(p, src, bytes); p += bytes; };
    const int32_t mi = std::max(1, min_items), n_nodes = (int32_t)kd->tree.nodes.size();
    const double hc = std::max(1.0, hit_cost);
    put(&kd->max_depth, 4); put(&mi, 4); put(&hc, 8); put(&empty_bonus, 8);
    put(kd->lower, 24); put(kd->upper, 24); put(&n_nodes, 4);
    for (const rsx_kdnode &nd : kd->tree.nodes) {
        put(&nd.type, 4);
        if (nd.type < 0) { put(&nd.count, 4); put(kd->tree.items.data() + nd.u.leaf.first_item, 4 * (size_t)nd.count); }
        else { put(&nd.u.split, 8); put(&nd.count, 4); }
    }
    return need;
}

// ---------------------------------------------------------------------------------------------------
// mesh preprocessing
// ---------------------------------------------------------------------------------------------------
namespace {

struct V3 { double x, y, z; };

inline V3 vertex(const float *v, int32_t i) { return {(double)v[3 * (size_t)i], (double)v[3 * (size_t)i + 1], (double)v[3 * (size_t)i + 2]}; }

// (p2 - p1) x (p3 - p1) with the reference's component expressions (vector.pyx:306-310)
inline V3 edge_cross(const float *verts, const int32_t *tri) {
    const V3 p1 = vertex(verts, tri[0]), p2 = vertex(verts, tri[1]), p3 = vertex(verts, tri[2]);
    const V3 a = {p2.x - p1.x, p2.y - p1.y, p2.z - p1.z}, b = {p3.x - p1.x, p3.y - p1.y, p3.z - p1.z};
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}

}  // namespace

extern "C" int32_t rsx_mesh_filter_triangles(const float *vertices, int32_t *triangles, int32_t n, int32_t stride) {
    if (!vertices || !triangles || n < 0 || stride < 3) return rsx_fail(RSX_EINVAL, "rsx_mesh_filter_triangles: bad arguments");
    int32_t kept = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t *tri = triangles + (size_t)i * stride;
        const V3 c = edge_cross(vertices, tri);
        if (std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z) == 0.0) continue;   // degenerate (mesh.pyx:363-399)
        if (kept != i) std::memmove(triangles + (size_t)kept * stride, tri, sizeof(int32_t) * stride);
        ++kept;
    }
    return kept;
}

extern "C" int rsx_mesh_face_normals(const float *vertices, const int32_t *triangles, int32_t n, int32_t stride, float *out) {
    if (!vertices || !triangles || !out || n < 0 || stride < 3) return rsx_fail(RSX_EINVAL, "rsx_mesh_face_normals: bad arguments");
#pragma omp parallel for schedule(static) num_threads(host_team_size())
    for (int32_t i = 0; i < n; ++i) {
        V3 c = edge_cross(vertices, triangles + (size_t)i * stride);
        const double s = 1.0 / std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z);
        out[3 * (size_t)i] = (float)(c.x * s);
        out[3 * (size_t)i + 1] = (float)(c.y * s);
        out[3 * (size_t)i + 2] = (float)(c.z * s);
    }
    return RSX_OK;
}

extern "C" int rsx_mesh_triangle_aabbs(const float *vertices, const int32_t *triangles, int32_t n, int32_t stride, double *out) {
    if (!vertices || !triangles || !out || n < 0 || stride < 3) return rsx_fail(RSX_EINVAL, "rsx_mesh_triangle_aabbs: bad arguments");
    const double padding = 1e-6;                                            // BOX_PADDING, mesh.pyx:42
#pragma omp parallel for schedule(static) num_threads(host_team_size())
    for (int32_t i = 0; i < n; ++i) {
        const int32_t *tri = triangles + (size_t)i * stride;
        double lo[3], hi[3], widest = 0.0;
        for (int k = 0; k < 3; ++k) {
            const float a = vertices[3 * (size_t)tri[0] + k], b = vertices[3 * (size_t)tri[1] + k], c = vertices[3 * (size_t)tri[2] + k];
            lo[k] = std::min(std::min(a, b), c);
            hi[k] = std::max(std::max(a, b), c);
            widest = std::max(widest, std::max(0.0, hi[k] - lo[k]));
        }
        const double pad = std::max(padding, widest * padding);
        for (int k = 0; k < 3; ++k) { out[6 * (size_t)i + k] = lo[k] - pad; out[6 * (size_t)i + 3 + k] = hi[k] + pad; }
    }
    return RSX_OK;
}

extern "C" int rsx_mesh_world_bbox(const float *vertices, int32_t nv, const double *m, double *out) {
    if (!vertices || !m || !out || nv < 0) return rsx_fail(RSX_EINVAL, "rsx_mesh_world_bbox: bad arguments");
    const double inf = std::numeric_limits<double>::infinity(), padding = 1e-6;
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int32_t i = 0; i < nv; ++i) {
        const V3 p = vertex(vertices, i);
        double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];        // Point3D.transform, point.pyx:253-284
        w = 1.0 / w;
        const double q[3] = {(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3]) * w, (m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7]) * w,
                             (m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]) * w};
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], q[k] - padding); hi[k] = std::max(hi[k], q[k] + padding); }
    }
    for (int k = 0; k < 3; ++k) { out[k] = lo[k]; out[3 + k] = hi[k]; }
    return RSX_OK;
}

// ---------------------------------------------------------------------------------------------------
// MT19937-64 (Matsumoto & Nishimura 2004), stream-compatible with raysect.core.math.random
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int kN = 312, kM = 156;
constexpr uint64_t kUpper = 0xFFFFFFFF80000000ULL, kLower = 0x7FFFFFFFULL, kMatrix = 0xB5026F5AA96619E9ULL;

void mt_refill(rsx_mt *st) {
    uint64_t *mt = st->mt;
    for (int i = 0; i < kN; ++i) {
        const uint64_t x = (mt[i] & kUpper) | (mt[(i + 1) % kN] & kLower);
        mt[i] = mt[(i + kM) % kN] ^ (x >> 1) ^ ((x & 1) ? kMatrix : 0ULL);
    }
    st->mti = 0;
}
}  // namespace

extern "C" void rsx_mt_seed_words(rsx_mt *st, const uint64_t *key, uint64_t key_length) {
    uint64_t *mt = st->mt;
    mt[0] = 19650218ULL;
    for (int i = 1; i < kN; ++i) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
    unsigned i = 1, j = 0;
    for (uint64_t k = std::max<uint64_t>(kN, key_length); k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 62)) * 3935559000370003845ULL)) + key[j] + j;
        if (++i >= (unsigned)kN) { mt[0] = mt[kN - 1]; i = 1; }
        if (++j >= key_length) j = 0;
    }
    for (int k = kN - 1; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 62)) * 2862933555777941757ULL)) - i;
        if (++i >= (unsigned)kN) { mt[0] = mt[kN - 1]; i = 1; }
    }
    mt[0] = 1ULL << 63;
    st->mti = kN;
}

extern "C" void rsx_mt_uniform(rsx_mt *st, int64_t n, double *out) {
    for (int64_t k = 0; k < n; ++k) {
        if (st->mti >= kN) mt_refill(st);
        uint64_t x = st->mt[st->mti++];
        x ^= (x >> 29) & 0x5555555555555555ULL;
        x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
        x ^= (x << 37) & 0xFFF7EEE000000000ULL;
        x ^= (x >> 43);
        out[k] = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
}
