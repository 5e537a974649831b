// dev_mesh.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// Mesh: watertight triangle test, per-lane and wave-cooperative KD traversal (mesh.pyx).
#pragma once

// ---------------------------------------------------------------------------------------------------
// Mesh — raysect/primitive/mesh/mesh.pyx:506-713 (MeshData.trace / _trace_leaf / _hit_triangle)
// ---------------------------------------------------------------------------------------------------
struct MeshHit {
    float u, v, w, t;
    int32_t tri;
};

// ray-space constants of the watertight test — _calc_rayspace_transform, mesh.pyx:566-610
struct TriRay {
    double ox, oy, oz, maxd;
    float sx, sy, sz;
    int ix, iy, iz;
};

__device__ __forceinline__ TriRay tri_ray(const Ray &r) {
    TriRay q;
    int ix, iy, iz;
    const double ax = fabs(r.dx), ay = fabs(r.dy), az = fabs(r.dz);
    if (ax > ay && ax > az) { ix = 1; iy = 2; iz = 0; }
    else if (ay > ax && ay > az) { ix = 2; iy = 0; iz = 1; }
    else { ix = 0; iy = 1; iz = 2; }
    const float rdz = (float)sel3(iz, r.dx, r.dy, r.dz);
    if (rdz < 0.0f) { const int tmp = ix; ix = iy; iy = tmp; }
    q.sz = (float)(1.0 / (double)rdz);
    q.sx = (float)(sel3(ix, r.dx, r.dy, r.dz) * (double)q.sz);
    q.sy = (float)(sel3(iy, r.dx, r.dy, r.dz) * (double)q.sz);
    q.ix = ix; q.iy = iy; q.iz = iz;
    q.ox = r.ox; q.oy = r.oy; q.oz = r.oz; q.maxd = r.maxd;
    return q;
}

// _hit_triangle, mesh.pyx:616-713, in three stages so that the packet walk can do the first two once per triangle instead of once per
// (ray, triangle) — rays from one origin share the translated vertices, rays with one dominant axis share the permutation.
// (1) vertices relative to the ray origin: f32 vertex minus f64 origin, rounded to f32 (mesh.pyx:633-643)
struct TriVerts { float x1, y1, z1, x2, y2, z2, x3, y3, z3; };
__device__ __forceinline__ TriVerts tri_translate(double ox, double oy, double oz, const float4 q0, const float4 q1, const float4 q2) {
    TriVerts t;
    t.x1 = (float)((double)q0.x - ox); t.y1 = (float)((double)q0.y - oy); t.z1 = (float)((double)q0.z - oz);
    t.x2 = (float)((double)q0.w - ox); t.y2 = (float)((double)q1.x - oy); t.z2 = (float)((double)q1.y - oz);
    t.x3 = (float)((double)q1.z - ox); t.y3 = (float)((double)q1.w - oy); t.z3 = (float)((double)q2.x - oz);
    return t;
}
// (2) the ray space's axis permutation (mesh.pyx:645-655): x* = component ix, y* = component iy, z* = component iz
__device__ __forceinline__ TriVerts tri_permute(int ix, int iy, int iz, const TriVerts &t) {
    TriVerts p;
    p.x1 = sel3f(ix, t.x1, t.y1, t.z1); p.y1 = sel3f(iy, t.x1, t.y1, t.z1); p.z1 = sel3f(iz, t.x1, t.y1, t.z1);
    p.x2 = sel3f(ix, t.x2, t.y2, t.z2); p.y2 = sel3f(iy, t.x2, t.y2, t.z2); p.z2 = sel3f(iz, t.x2, t.y2, t.z2);
    p.x3 = sel3f(ix, t.x3, t.y3, t.z3); p.y3 = sel3f(iy, t.x3, t.y3, t.z3); p.z3 = sel3f(iz, t.x3, t.y3, t.z3);
    return p;
}
// (3) shear, scaled barycentrics, range tests, normalisation (mesh.pyx:657-713). Returns true with normalised (t,u,v,w) on a hit.
// (The early returns are the reference's and they are the cheapest form: most tested triangles are missed at the edge test. Folding
// them into one predicate was measured — every lane then computes det and t: +3 % vector instructions, 25.5 against 24.7 ms.)
__device__ __forceinline__ bool tri_finish(float sx, float sy, float sz, double maxd, const TriVerts &p, float &ht, float &hu, float &hv, float &hw) {
    const float a1 = p.x1, b1 = p.y1, c1 = p.z1, a2 = p.x2, b2 = p.y2, c2 = p.z2, a3 = p.x3, b3 = p.y3, c3 = p.z3;
    const float x1 = a1 - sx * c1, x2 = a2 - sx * c2, x3 = a3 - sx * c3;
    const float y1 = b1 - sy * c1, y2 = b2 - sy * c2, y3 = b3 - sy * c3;
    float u = x3 * y2 - y3 * x2, v = x1 * y3 - y1 * x3, w = x2 * y1 - y2 * x1;
    const bool on_edge = (u == 0.0f) | (v == 0.0f) | (w == 0.0f);           // rare: recompute in f64 (mesh.pyx:668-680), behind a wave-level test
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(on_edge) != 0ULL, 0)) {
        if (on_edge) {
            u = (float)((double)x3 * (double)y2 - (double)y3 * (double)x2);
            v = (float)((double)x1 * (double)y3 - (double)y1 * (double)x3);
            w = (float)((double)x2 * (double)y1 - (double)y2 * (double)x1);
        }
    }
    if ((u < 0.0f || v < 0.0f || w < 0.0f) && (u > 0.0f || v > 0.0f || w > 0.0f)) return false;
    const float det = u + v + w;
    if (det == 0.0f) return false;
    const float z1 = sz * c1, z2 = sz * c2, z3 = sz * c3;
    const float t = u * z1 + v * z2 + w * z3;
    if (det > 0.0f) { if (t < 0.0f || (double)t > maxd * (double)det) return false; }
    else            { if (t > 0.0f || (double)t < maxd * (double)det) return false; }
    const float rdet = (float)(1.0 / (double)det);
    ht = t * rdet; hu = u * rdet; hv = v * rdet; hw = w * rdet;
    return true;
}
__device__ __forceinline__ bool tri_test(const TriRay &q, const float4 q0, const float4 q1, const float4 q2, float &ht, float &hu, float &hv, float &hw) {
    return tri_finish(q.sx, q.sy, q.sz, q.maxd, tri_permute(q.ix, q.iy, q.iz, tri_translate(q.ox, q.oy, q.oz, q0, q1, q2)), ht, hu, hv, hw);
}

#ifndef RSX_LEAF_INLINE
#define RSX_LEAF_INLINE 1          // 1: leaves read 64-byte leaf-ordered triangle records; 0: items[] -> 48-byte records by id
#endif

// fetch leaf item `pos` (absolute position in items[]): its triangle id and 48-byte record
__device__ __forceinline__ void leaf_fetch(const int32_t *items, const float4 *tris, const float4 *leaf, int32_t pos, int32_t &tri, float4 &a,
                                           float4 &b, float4 &c) {
#if RSX_LEAF_INLINE
    const float4 *rec = leaf + 4 * (size_t)pos;
    a = rec[0]; b = rec[1]; c = rec[2];
    tri = __float_as_int(rec[3].x);
#else
    tri = items[pos];
    const float4 *rec = tris + 3 * (size_t)tri;
    a = rec[0]; b = rec[1]; c = rec[2];
#endif
}

#ifndef RSX_LEAF_BATCH
#define RSX_LEAF_BATCH 3           // triangles whose loads are issued together before the tests (latency hiding inside a leaf)
#endif
#define RSX_LEAF_BATCH_DEFAULT RSX_LEAF_BATCH

__device__ bool mesh_trace(const DMesh &m, const Ray &r, Stack st, MeshHit &out) {
    double tmin, tmax;
    if (!aabb(m.lower, m.upper, r, tmin, tmax)) return false;                 // kdtree3d.pyx:589-607
    const TriRay q = tri_ray(r);
    const AxisDiv ad = axis_div(r);

    int32_t node = 0, sp = 0;
    for (;;) {
        const rsx_kdnode nd = descend(m.nodes, node, r, ad, tmin, tmax, st, sp);
        // _trace_leaf, mesh.pyx:520-563 — items are tested in leaf order, strict `<` keeps the first of equal distances
        double distance = r.maxd < tmax ? r.maxd : tmax;
        int32_t closest = -1;
        float bu = 0, bv = 0, bw = 0;
        const int32_t first = nd.u.leaf.first_item;
        const int32_t count = nd.count;
        for (int32_t k = 0; k < count; k += RSX_LEAF_BATCH) {
            int32_t tri[RSX_LEAF_BATCH];
            float4 t0[RSX_LEAF_BATCH], t1[RSX_LEAF_BATCH], t2[RSX_LEAF_BATCH];
#pragma unroll
            for (int j = 0; j < RSX_LEAF_BATCH; ++j)
                leaf_fetch(m.items, m.tris, m.leaf, first + (k + j < count ? k + j : count - 1), tri[j], t0[j], t1[j], t2[j]);
#pragma unroll
            for (int j = 0; j < RSX_LEAF_BATCH; ++j) {
                float ht, hu, hv, hw;
                if (k + j < count && tri_test(q, t0[j], t1[j], t2[j], ht, hu, hv, hw) && (double)ht < distance) {
                    distance = (double)ht; closest = tri[j]; bu = hu; bv = hv; bw = hw;
                }
            }
        }
        if (closest >= 0) { out.u = bu; out.v = bv; out.w = bw; out.t = (float)distance; out.tri = closest; return true; }
        if (sp == 0) return false;
        --sp;
        tmin = tmax;                    // far range starts where the exhausted near range ended
        stack_pop(st, sp, node, tmax);
    }
}

#ifndef RSX_STAGE_MIN
#define RSX_STAGE_MIN 4            // >= this many rays of the wave in the same big leaf: stage the leaf through LDS instead
#endif
#ifndef RSX_PHASE_PROF
#define RSX_PHASE_PROF 0           // 1: accumulate per-phase s_memtime cycles of the wave-cooperative mesh traversal (tuning builds only)
#endif
#if RSX_PHASE_PROF == 1
#define PHASE_DECL unsigned long long ph_t = clock64();
#define PHASE_ADD(slot) { const unsigned long long now_ = clock64(); phase_acc[slot] += now_ - ph_t; ph_t = now_; }
#else
#define PHASE_DECL
#define PHASE_ADD(slot)
#endif
#ifndef RSX_SKIP_EMPTY
#define RSX_SKIP_EMPTY 1           // coherent passes: a lane that reaches an empty leaf pops and descends again (this many times) before the wave's leaf phase
#endif
#ifndef RSX_COOP_LEAF
#define RSX_COOP_LEAF 24           // leaves with at least this many triangles are tested by the whole wave for one ray at a time
#endif

__device__ __forceinline__ double shfl_f64(double x, int lane) { return __shfl(x, lane, WAVE); }
__device__ __forceinline__ double readlane_f64(double x, int lane) {          // lane must be wave-uniform; the result is scalar
    const long long b = __double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Wave-cooperative MeshData.trace: every lane of the wave calls this together on ONE mesh (`m` is wave-uniform; `want` = lane has
// a ray for it). Lanes walk their own rays through the tree; small leaves are tested per lane, but a leaf with >= RSX_COOP_LEAF
// triangles (high-valence vertices produce leaves of hundreds, mesh.pyx builds them because the depth cap stops the SAH split) is
// tested by all 64 lanes for one ray at a time: 64 triangles per step instead of 1, then a (t, leaf position) lexicographic
// wave-min, which is exactly what the reference's sequential scan with strict `<` returns (the first item among those with the
// smallest distance). Idle lanes — rays that already finished, or never needed this mesh — serve as helpers.
// STAGE_MIN: rays of the wave that must share a big leaf before it is staged through LDS (below that the wave serves one ray at a
// time). 1 = always stage: the serve-one-ray code is then not compiled in, which is worth 11 % on passes whose waves are coherent
// (several samples per pixel: -11 % on configs[2], -9 % at 100 spp, -4 % at 4 spp) and loses on 1-spp passes (+45 % on configs[1]).
template <int STAGE_MIN = RSX_STAGE_MIN>
__device__ bool mesh_trace_wave(bool want, UMesh m, const Ray &r, const Stack &st, MeshHit &out, uint32_t &work, unsigned long long *phase_acc = nullptr) {
    const int lane = threadIdx.x % WAVE;
#if RSX_PHASE_PROF == 2
    const unsigned long long ph2_t0 = clock64();
#endif
    constexpr int LEAF_BATCH = STAGE_MIN <= 1 ? 2 : RSX_LEAF_BATCH_DEFAULT;
    constexpr int SKIP_EMPTY = STAGE_MIN <= 1 ? RSX_SKIP_EMPTY : 0;   // coherent passes: two records in flight are enough (-3 % on configs[2])
    PHASE_DECL
    const rsx_kdnode *nodes = m->nodes;                     // scalar loads: the bases sit in SGPRs, lanes supply 32-bit offsets
    const float4 *leaf = m->leaf, *tris = m->tris;
    const int32_t *items = m->items;
    double tmin = 0, tmax = 0;
    bool active;
    const AxisDiv ad = axis_div(r);
    {
        // BoundingBox3D.intersect (kdtree3d.pyx:589-607) needs 1.0 / d per axis: formed from the refined reciprocals that the
        // branch steps use anyway (exact_div(1, d) is the correctly rounded quotient), not by three more full divisions
        const double lo[3] = {m->lower[0], m->lower[1], m->lower[2]}, hi[3] = {m->upper[0], m->upper[1], m->upper[2]};
        const double rx = exact_div(1.0, r.dx, ad.yx, ad.safe & 1), ry = exact_div(1.0, r.dy, ad.yy, (ad.safe >> 1) & 1),
                     rz = exact_div(1.0, r.dz, ad.yz, (ad.safe >> 2) & 1);
        active = want && aabb_rcp(lo, hi, r, rx, ry, rz, tmin, tmax);
    }
    const TriRay q = tri_ray(r);
    bool hit = false;
    int32_t node = 0, sp = 0;
#if RSX_PHASE_PROF == 2
    phase_acc[3] += clock64() - ph2_t0;
#endif
    while (__any(active)) {
        double distance = 0;
        int32_t closest = -1, count = 0, first = 0;
        float bu = 0, bv = 0, bw = 0;
        work += 8;                                   // one descend + small-leaf round of the wave (scheduling weight, see k_order_units)
        PHASE_ADD(0)
        rsx_kdnode nd;
        nd.count = 0; nd.u.leaf.first_item = 0;
#if RSX_UTIL_PROF == 2
        if (active) nd = descend(nodes, node, r, ad, tmin, tmax, st, sp, nullptr);
#else
        if (active) { UTIL_COUNT(phase_acc, 2) }
        if (active) nd = descend(nodes, node, r, ad, tmin, tmax, st, sp, phase_acc);
#endif
        const bool leafy = active;
        // An empty leaf (70 % of the leaves a ray crosses: the SAH's empty-space cuts) ends nothing: the lane pops and walks on inside
        // the same round — once. Coherent passes: 41.1 -> 40.0 ms on configs[2]; walking on further (4 times, without bound) loses it
        // again (40.8, 41.2 ms: fewer but longer and less balanced rounds), and 1-spp passes do not gain, so they keep the plain round.
        if (SKIP_EMPTY > 0 && leafy) {
            int budget = SKIP_EMPTY;
            while (nd.count == 0 && sp > 0 && budget-- > 0) {
                --sp;
                tmin = tmax;
                stack_pop(st, sp, node, tmax);
                nd = descend(nodes, node, r, ad, tmin, tmax, st, sp, phase_acc);
            }
        }
        PHASE_ADD(1)
        if (leafy) {
            distance = r.maxd < tmax ? r.maxd : tmax;                         // _trace_leaf, mesh.pyx:520-563
            count = nd.count;
            first = nd.u.leaf.first_item;
            if (count < RSX_COOP_LEAF) {
                for (int32_t k = 0; k < count; k += LEAF_BATCH) {
#if RSX_UTIL_PROF == 1
                    UTIL_COUNT(phase_acc, 6)
#endif
                    int32_t tri[LEAF_BATCH];
                    float4 t0[LEAF_BATCH], t1[LEAF_BATCH], t2[LEAF_BATCH];
#pragma unroll
                    for (int j = 0; j < LEAF_BATCH; ++j)
                        leaf_fetch(items, tris, leaf, first + (k + j < count ? k + j : count - 1), tri[j], t0[j], t1[j], t2[j]);
#pragma unroll
                    for (int j = 0; j < LEAF_BATCH; ++j) {
                        float ht, hu, hv, hw;
                        if (k + j < count && tri_test(q, t0[j], t1[j], t2[j], ht, hu, hv, hw) && (double)ht < distance) {
                            distance = (double)ht; closest = tri[j]; bu = hu; bv = hv; bw = hw;
                        }
                    }
                }
            }
        }
        PHASE_ADD(2)
        // ---- cooperative stage for big leaves
        unsigned long long big = __ballot(active && count >= RSX_COOP_LEAF);
#if RSX_PHASE_PROF == 1
        phase_acc[5] += 1; phase_acc[6] += __popcll(big); phase_acc[7] += __popcll(__ballot(active));
#endif
        while (big) {
            const int leader = __ffsll((long long)big) - 1;
            const int32_t lcount = __builtin_amdgcn_readlane(count, leader), lfirst = __builtin_amdgcn_readlane(first, leader);
            // lanes whose ray sits in the same leaf as the leader's
            const bool same = active && count >= RSX_COOP_LEAF && first == lfirst;
            const unsigned long long group = __ballot(same);
            big &= ~group;
            if (STAGE_MIN <= 1 || __popcll(group) >= STAGE_MIN) {
                // (a) coherent rays: stage the leaf through LDS 64 triangles at a time; every lane of the group tests them all,
                //     in leaf order (the reference's own loop), reading each record as an LDS broadcast
                float4 *rec = st.stage;
                int32_t *ids = reinterpret_cast<int32_t *>(st.stage + 3 * WAVE);
                work += 2 + (uint32_t)lcount / 8;
                for (int32_t c = 0; c < lcount; c += WAVE) {
                    __builtin_amdgcn_wave_barrier();
                    const int32_t k = c + lane;
                    if (k < lcount) {
                        int32_t tri;
                        float4 a, b, cc;
                        leaf_fetch(items, tris, leaf, lfirst + k, tri, a, b, cc);
                        rec[3 * lane] = a; rec[3 * lane + 1] = b; rec[3 * lane + 2] = cc;
                        ids[lane] = tri;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (same) {
                        const int32_t nj = lcount - c < WAVE ? lcount - c : WAVE;
                        for (int32_t j = 0; j < nj; ++j) {
                            float ht, hu, hv, hw;
                            if (tri_test(q, rec[3 * j], rec[3 * j + 1], rec[3 * j + 2], ht, hu, hv, hw) && (double)ht < distance) {
                                distance = (double)ht; closest = ids[j]; bu = hu; bv = hv; bw = hw;
                            }
                        }
                    }
                }
                continue;
            }
            // (b) isolated rays: one ray at a time, all 64 lanes testing 64 triangles per step
            unsigned long long todo = group;
            while (todo) {
                const int owner = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                work += 2 + (uint32_t)lcount / 32;
                // the owner's ray constants are read into scalar registers (owner is wave-uniform): no per-lane copy of the ray
                TriRay lq;
                lq.ox = readlane_f64(q.ox, owner); lq.oy = readlane_f64(q.oy, owner); lq.oz = readlane_f64(q.oz, owner);
                lq.maxd = readlane_f64(q.maxd, owner);
                lq.sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.sx), owner));
                lq.sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.sy), owner));
                lq.sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.sz), owner));
                const int axes = __builtin_amdgcn_readlane(q.ix | (q.iy << 2) | (q.iz << 4), owner);
                lq.ix = axes & 3; lq.iy = (axes >> 2) & 3; lq.iz = (axes >> 4) & 3;
                const double limit = readlane_f64(distance, owner);
                // each lane scans positions lane, lane+64, ... in ascending order (strict `<` keeps its earliest minimum)
                float mt = INFINITY, mu = 0, mv = 0, mw = 0;
                int32_t mk = 0x7fffffff, mtri = -1;
                for (int32_t k = lane; k < lcount; k += WAVE) {
                    int32_t tri;
                    float4 a, b, c;
                    leaf_fetch(items, tris, leaf, lfirst + k, tri, a, b, c);
                    float ht, hu, hv, hw;
                    if (tri_test(lq, a, b, c, ht, hu, hv, hw) && (double)ht < limit && ht < mt) { mt = ht; mk = k; mtri = tri; mu = hu; mv = hv; mw = hw; }
                }
                // wave-wide lexicographic min of (t, position) == the reference's sequential scan with strict `<`
                float wt = mt;
                int32_t wk = mk;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float ot = __shfl_xor(wt, off, WAVE);
                    const int32_t ok = __shfl_xor(wk, off, WAVE);
                    if (ot < wt || (ot == wt && ok < wk)) { wt = ot; wk = ok; }
                }
                const int winner = wk & (WAVE - 1);              // position k was scanned by lane k % 64
                const bool found = wk != 0x7fffffff;
                const float ru = __shfl(mu, winner, WAVE), rv = __shfl(mv, winner, WAVE), rw = __shfl(mw, winner, WAVE);
                const int32_t rtri = __shfl(mtri, winner, WAVE);
                if (lane == owner && found) { distance = (double)wt; closest = rtri; bu = ru; bv = rv; bw = rw; }
            }
        }
        PHASE_ADD(3)
        if (leafy) {
            if (closest >= 0) { out.u = bu; out.v = bv; out.w = bw; out.t = (float)distance; out.tri = closest; hit = true; active = false; }
            else if (sp == 0) active = false;
            else {
                --sp;
                tmin = tmax;
                stack_pop(st, sp, node, tmax);
            }
        }
        PHASE_ADD(4)
    }
    return hit;
}

