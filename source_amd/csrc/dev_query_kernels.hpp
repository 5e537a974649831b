// dev_query_kernels.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// Kernels of the batch queries: hit, roots (hit + next_intersection sequences), contains.
#pragma once

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// carve the wave's LDS region and global spill region into the world stack and the mesh stack
__device__ __forceinline__ void wave_stacks(const DScene &sc, Stack &ws, Stack &ms) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));       // uniform by construction: tell the compiler
    const int lds_levels = sc.wlds + sc.mlds;
    const uint32_t base = (uint32_t)wave * (uint32_t)(lds_levels * WAVE * 12 + STAGE_BYTES);
    const uint32_t lds_t = base, lds_id = base + (uint32_t)lds_levels * WAVE * 8;
    const int spill_levels = (sc.wdepth - sc.wlds) + (sc.mdepth - sc.mlds);
    const size_t gwave = (size_t)blockIdx.x * (blockDim.x / WAVE) + wave;
    char *gbase = sc.spill + gwave * spill_wave_bytes(sc.wdepth, sc.wlds, sc.mdepth, sc.mlds);
    char *gt = gbase, *gid = gbase + (size_t)spill_levels * WAVE * 8;
    float4 *stage = reinterpret_cast<float4 *>(smem + base + (size_t)lds_levels * WAVE * 12);
    ws.stage = stage; ms.stage = stage;
    ws.lds_t = lds_t; ws.lds_id = lds_id; ws.gt = gt; ws.gid = gid; ws.lds_levels = sc.wlds;
    ms.lds_t = lds_t + (uint32_t)sc.wlds * WAVE * 8; ms.lds_id = lds_id + (uint32_t)sc.wlds * WAVE * 4;
    ms.gt = gt + (size_t)(sc.wdepth - sc.wlds) * WAVE * 8; ms.gid = gid + (size_t)(sc.wdepth - sc.wlds) * WAVE * 4; ms.lds_levels = sc.mlds;
}

// XCD (accelerator complex die) this wave runs on: HW_REG_XCC_ID, bits [3:0]
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u); }

// wave-level ticket: lane 0 takes the next batch of 64 work items
__device__ __forceinline__ long long next_batch(unsigned long long *ticket) {
    long long base = 0;
    if (threadIdx.x % WAVE == 0) base = (long long)atomicAdd(ticket, 64ULL);
    return __shfl(base, 0, WAVE);
}

struct HitOut {
    int32_t *prim; double *t; uint8_t *exiting; int32_t *tri; float *uvw; double *geom;
};

// CSG scenes run it twice, like the render kernels: MODE 1 has only the state-free CSG evaluator and marks the rays it cannot finish
// (prim = HIT_REDO), MODE 2 carries the stream merge and traces exactly those again.
#define HIT_REDO (-2)
template <bool CSG, int MODE = 0>
#ifndef RSX_QUERY_MIN_WAVES
// The batch query's plain form at TWO waves per SIMD: built for three (168 registers, like the render kernels' per-lane walk) it spills 44
// registers around every world step, and scattered rays — what a query batch is — pay for them in every step: 2^22 rays in a Cornell box
// 1.65 -> 1.38 ms, on the configs[2] scene 1.14 -> 1.01 ms (tools/hit_batch_rate.py; the render kernels keep three: configs[1] 0.191 vs 0.197 ms per step).
#define RSX_QUERY_MIN_WAVES 2
#endif
__global__ __launch_bounds__(WG_THREADS, !CSG ? RSX_QUERY_MIN_WAVES : MODE == 1 ? RSX_CSGFAST_MIN_WAVES : RSX_CSG_MIN_WAVES)
void k_hit_batch(DScene sc, long long n, const double *origin, const double *direction, const double *maxd, HitOut out, unsigned long long *ticket) {
    Stack st, ms;
    wave_stacks(sc, st, ms);
    const int lane = threadIdx.x % WAVE;
    NodeSt csg_state[CSG && MODE != 1 ? CSG_MAX_SLOTS : 1];
    for (;;) {
        const long long base = next_batch(ticket);
        if (base >= n) break;
        const long long i = base + lane;
        bool valid = i < n;                     // lanes without a ray still walk the loops: they help on big mesh leaves
        if constexpr (MODE == 2) {
            valid = valid && out.prim[i] == HIT_REDO;
            if (!__any(valid)) continue;
        }
        Ray r;
        r.ox = r.oy = r.oz = 0.0; r.dx = r.dy = 0.0; r.dz = 1.0; r.maxd = 0.0;
        if (valid) {
            r.ox = origin[3 * i]; r.oy = origin[3 * i + 1]; r.oz = origin[3 * i + 2];
            r.dx = direction[3 * i]; r.dy = direction[3 * i + 1]; r.dz = direction[3 * i + 2];
            r.maxd = maxd[i];
        }
        Hit h;
        uint32_t work = 0;
        const bool hit = world_trace_wave<CSG, MODE == 1, RSX_STAGE_MIN, false, CSG && MODE == 1 && RSX_CSG_MAILBOX >= 4 ? RSX_CSG_WIDE : 2>(valid, sc, r, st, ms, csg_state, h, work);
        if (!valid) continue;
        if constexpr (MODE == 1) { if (work >> 31) { out.prim[i] = HIT_REDO; continue; } }
        out.prim[i] = hit ? h.prim : -1;
        bool mesh = hit && sc.prims[h.prim].type == RSX_PRIM_MESH;
        // (a mesh reports `t + Mesh._ray_distance`, mesh.pyx:1240-1275 — + 0.0 for a first hit: a root at -0.0, an origin ON the surface, leaves as
        // +0.0 while the geometry below is formed with the root as found)
        if (out.t) out.t[i] = hit ? (mesh ? h.t + 0.0 : h.t) : NAN;
        if constexpr (CSG) {   // a CSG node hands back its operand's MeshIntersection (triangle, u, v, w survive the lift)
            if (hit && is_csg(sc.prims[h.prim].type)) mesh = sc.prims[h.leaf].type == RSX_PRIM_MESH;
        }
        if (out.tri) out.tri[i] = mesh ? h.a0 : -1;
        if (out.uvw) { out.uvw[3 * i] = mesh ? h.u : 0.0f; out.uvw[3 * i + 1] = mesh ? h.v : 0.0f; out.uvw[3 * i + 2] = mesh ? h.w : 0.0f; }
        if (out.exiting || out.geom) {
            Geom g;
            if (hit) finalise<CSG>(sc, r, h, g);
            if (out.exiting) out.exiting[i] = hit ? (g.exiting ? 1 : 0) : 0;
            if (out.geom) {
                double *o = out.geom + 12 * i;
                for (int k = 0; k < 3; ++k) {
                    o[k] = hit ? g.hit[k] : NAN; o[3 + k] = hit ? g.inside[k] : NAN;
                    o[6 + k] = hit ? g.outside[k] : NAN; o[9 + k] = hit ? g.normal[k] : NAN;
                }
            }
        }
    }
}

// Primitive.hit + next_intersection() sequence on one primitive (tests / Primitive API parity).
// Mesh.next_intersection re-traces from hit + d*1e-6 with max - t - 1e-6 (mesh.pyx:1240-1275).
template <bool CSG>
__global__ __launch_bounds__(WG_THREADS) void k_roots(DScene sc, int32_t pidx, long long n, const double *origin, const double *direction,
                                                      const double *maxd, int32_t max_roots, int32_t *counts, double *t, uint8_t *exiting,
                                                      double *geom, int32_t *tri, float *uvw, unsigned long long *ticket) {
    Stack st, ms;
    wave_stacks(sc, st, ms);
    const int lane = threadIdx.x % WAVE;
    const rsx_primitive &p = sc.prims[pidx];
    NodeSt csg_state[CSG ? CSG_MAX_SLOTS : 1];
    for (;;) {
        const long long base = next_batch(ticket);
        if (base >= n) break;
        const long long i = base + lane;
        if (i >= n) continue;
        Ray r;
        r.ox = origin[3 * i]; r.oy = origin[3 * i + 1]; r.oz = origin[3 * i + 2];
        r.dx = direction[3 * i]; r.dy = direction[3 * i + 1]; r.dz = direction[3 * i + 2];
        r.maxd = maxd[i];
        Ray l = to_local(p, r);
        int32_t c = 0;
        // optional per-root outputs: Intersection geometry in primitive space (hit, inside, outside, normal) and, for mesh
        // surfaces, the MeshIntersection extras (triangle, u, v, w) — intersection.pyx:36-106, mesh.pyx:85-135
        auto emit = [&](int32_t k, const Geom &g, int32_t triangle, float bu, float bv, float bw) {
            const size_t at = (size_t)i * max_roots + k;
            if (geom) {
                double *o = geom + 12 * at;
                for (int q = 0; q < 3; ++q) { o[q] = g.hit[q]; o[3 + q] = g.inside[q]; o[6 + q] = g.outside[q]; o[9 + q] = g.normal[q]; }
            }
            if (tri) tri[at] = triangle;
            if (uvw) { uvw[3 * at] = bu; uvw[3 * at + 1] = bv; uvw[3 * at + 2] = bw; }
        };
        if (CSG && is_csg(p.type)) {
            if constexpr (CSG) {
                CsgEval e;
                e.sc = &sc; e.st = csg_state;
                e.mesh_stack = ms;
                Rec rec;
                csg_first(e, pidx, r, rec);
                while ((rec.flags & F_VALID) && c < max_roots) {
                    t[i * max_roots + c] = rec.t;
                    exiting[i * max_roots + c] = (rec.flags & F_EXIT) ? 1 : 0;
                    if (geom || tri || uvw) {
                        Hit h;
                        h.prim = pidx; h.t = rec.t; h.a0 = rec.a0; h.a1 = rec.a1; h.u = rec.u; h.v = rec.v; h.w = rec.w;
                        h.leaf = rec.leaf; h.flags = rec.flags; h.hx = rec.hx; h.hy = rec.hy; h.hz = rec.hz;
                        Geom g;
                        csg_geom(sc, r, h, g);
                        const bool on_mesh = sc.prims[rec.leaf].type == RSX_PRIM_MESH;
                        emit(c, g, on_mesh ? rec.a0 : -1, on_mesh ? rec.u : 0.0f, on_mesh ? rec.v : 0.0f, on_mesh ? rec.w : 0.0f);
                    }
                    ++c;
                    csg_next(e, pidx, rec);
                }
            }
        } else if (p.type == RSX_PRIM_MESH) {
            const DMesh &m = sc.meshes[p.mesh];
            double accumulated = 0;
            MeshHit mh;
            while (c < max_roots && mesh_trace(m, l, ms, mh)) {
                Geom g;
                mesh_geom(m, l, (double)mh.t, mh.tri, mh.u, mh.v, mh.w, g);
                const double dist = (double)mh.t + accumulated;
                t[i * max_roots + c] = dist;
                exiting[i * max_roots + c] = g.exiting ? 1 : 0;
                emit(c, g, mh.tri, mh.u, mh.v, mh.w);
                ++c;
                l.ox = g.hit[0] + l.dx * MESH_EPS; l.oy = g.hit[1] + l.dy * MESH_EPS; l.oz = g.hit[2] + l.dz * MESH_EPS;
                l.maxd = l.maxd - (double)mh.t - MESH_EPS;
                accumulated = dist + MESH_EPS;
            }
        } else if (p.type <= RSX_PRIM_CYLINDER) {
            Roots roots;
            roots.n = 0;
            if (p.type == RSX_PRIM_SPHERE) sphere_roots(p, l, roots);
            else if (p.type == RSX_PRIM_BOX) box_roots(p, l, roots);
            else cylinder_roots(p, l, roots);
            for (int k = 0; k < roots.n && c < max_roots; ++k) {
                Geom g;
                analytic_geom(p, l, roots.t[k], roots.a0[k], roots.a1[k], g);
                t[i * max_roots + c] = roots.t[k];
                exiting[i * max_roots + c] = g.exiting ? 1 : 0;
                emit(c, g, -1, 0.0f, 0.0f, 0.0f);
                ++c;
            }
        }
        counts[i] = c;
    }
}

// World.contains — kdtree3d.pyx:736-792, kdtree.pyx:126-162, primitive contains():
//   sphere.pyx:202-214, box.pyx:344-361, cylinder.pyx:356-372, mesh.pyx:1277-1297 (+802-830)
template <bool CSG>
__global__ __launch_bounds__(WG_THREADS) void k_contains(DScene sc, long long n, const double *points, uint8_t *inside,
                                                         unsigned long long *ticket) {
    Stack st, ms;
    wave_stacks(sc, st, ms);
    const int lane = threadIdx.x % WAVE;
    for (;;) {
        const long long base = next_batch(ticket);
        if (base >= n) break;
        const long long i = base + lane;
        if (i >= n) continue;
        const double px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
        for (int j = 0; j < sc.n_world; ++j) inside[i * sc.n_world + j] = 0;
        if (!aabb_contains(sc.wlower, sc.wupper, px, py, pz)) continue;
        int32_t node = 0;
        rsx_kdnode nd = load_node(sc.wnodes, node);
        while (nd.type >= 0) {
            node = sel3(nd.type & 3, px, py, pz) < nd.u.split ? node + 1 : nd.count;
            nd = load_node(sc.wnodes, node);
        }
        for (int32_t k = 0; k < nd.count; ++k) {
            const int32_t idx = sc.witems[nd.u.leaf.first_item + k];
            const rsx_primitive &p = sc.prims[idx];
            bool in;
            if constexpr (CSG) {                                                                // BoundPrimitive.contains: box gate first
                if (is_csg(p.type) && sc.csgfast && sc.csgfast[idx].n_leaves > 0) in = csg_fast_contains(sc, idx, px, py, pz, ms);   // flattened analytic tree
                else in = node_contains(sc, idx, px, py, pz, ms);
            }
            else in = aabb_contains(p.box_lower, p.box_upper, px, py, pz) && leaf_contains(sc, p, px, py, pz, ms);
            inside[i * sc.n_world + idx] = in ? 1 : 0;
        }
    }
}

