// dev_world.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// World.hit for a wave of rays: world KD traversal, BoundPrimitive gate, closest-hit rule.
#pragma once

// ---------------------------------------------------------------------------------------------------
// World.hit — core/scenegraph/world.pyx:125-146, core/acceleration/kdtree.pyx:73-122,170-175,
//             boundprimitive.pyx:42-51
// ---------------------------------------------------------------------------------------------------
#ifdef CSGF_COUNT
__device__ unsigned long long g_csgf[3];      // [fallback, miss, hit] counts of csg_fast_hit (diagnostic builds only)
#endif
// FASTONLY (k_render_trace<true, 1>): the stream merge is not compiled in; a ray that would need it raises `needs_stream`, the
// primitive counts as missed, and the ray is traced again by the redo pass (k_render_trace<true, 2>), which has the merge.
template <bool CSG, bool FASTONLY = false>
__device__ __forceinline__ void primitive_first_hit(const DScene &sc, int32_t idx, const rsx_primitive &p, const Ray &r, Stack mesh_stack,
                                                    NodeSt *csg_state, Hit &cand, bool &needs_stream) {
    cand.prim = -1;
    if constexpr (CSG) {
        if (is_csg(p.type)) {
            if constexpr (FASTONLY) {
                int fast = -1;
                if (sc.csgfast && sc.csgfast[idx].n_leaves > 0 && mesh_stack.lds_levels >= 2 * sc.csgfast[idx].n_leaves) fast = csg_fast_hit(sc, idx, r, mesh_stack, cand);
                if (fast <= 0) cand.prim = -1;
                if (fast < 0) needs_stream = true;
                return;
            }
            if (sc.csgfast && sc.csgfast[idx].n_leaves > 0 && mesh_stack.lds_levels >= 2 * sc.csgfast[idx].n_leaves) {
                const int fast = csg_fast_hit(sc, idx, r, mesh_stack, cand);
#ifdef CSGF_COUNT
                atomicAdd(&g_csgf[fast + 1], 1ULL);
#endif
                if (fast == 0) cand.prim = -1;
                if (fast >= 0) return;
                cand.prim = -1;
            }
            CsgEval e;
            e.sc = &sc; e.st = csg_state; e.mesh_stack = mesh_stack;
            Rec rec;
            csg_first<CSG_MAX_DEPTH>(e, idx, r, rec);
            if (rec.flags & F_VALID) {
                cand.prim = idx; cand.t = rec.t; cand.a0 = rec.a0; cand.a1 = rec.a1; cand.u = rec.u; cand.v = rec.v; cand.w = rec.w;
                cand.leaf = rec.leaf; cand.flags = rec.flags; cand.hx = rec.hx; cand.hy = rec.hy; cand.hz = rec.hz;
            }
            return;
        }
    }
    // (mesh primitives never come here: world_trace_wave traces them wave-cooperatively, one instance at a time)
    const Ray l = to_local(p, r);
    Roots roots;
    roots.n = 0;
    if (p.type == RSX_PRIM_SPHERE) sphere_roots(p, l, roots);
    else if (p.type == RSX_PRIM_BOX) box_roots(p, l, roots);
    else if (p.type == RSX_PRIM_CYLINDER) cylinder_roots(p, l, roots);
    if (roots.n > 0) { cand.prim = idx; cand.t = roots.t[0]; cand.a0 = roots.a0[0]; cand.a1 = roots.a1[0]; cand.u = cand.v = cand.w = 0.0f; }
}

// World.hit for the 64 rays of a wave: every lane calls it together (`valid` = lane has a ray) and all loops are wave-uniform, so
// that mesh primitives can be traced with mesh_trace_wave (idle lanes help on big leaves). Leaf items are tested in leaf order and
// the closest kept with `<=` (later item wins ties, kdtree.pyx:113); a hit inside the leaf's range ends the traversal.
template <bool CSG, bool FASTONLY = false, int STAGE_MIN = RSX_STAGE_MIN>
__device__ bool world_trace_wave(bool valid, const DScene &sc, const Ray &r, const Stack &st, const Stack &mesh_stack, NodeSt *csg_state, Hit &best,
                                 uint32_t &work, unsigned long long *phase_acc = nullptr) {
    best.prim = -1;
    double tmin = 0, tmax = 0;
    const double rx = 1.0 / r.dx, ry = 1.0 / r.dy, rz = 1.0 / r.dz;
    bool active = valid && aabb_rcp(sc.wlower, sc.wupper, r, rx, ry, rz, tmin, tmax);
    // the world tree is a handful of nodes per ray: its branch steps use the plain division, which keeps three refined
    // reciprocals out of the registers that stay live across the mesh traversal
    AxisDiv ad;
    ad.yx = ad.yy = ad.yz = 0.0; ad.safe = 0;
    int32_t node = 0, sp = 0;
    while (__any(active)) {
        double distance = 0;
        int32_t count = 0;
        const int32_t *items = sc.witems;
        if (active) { UTIL_COUNT(phase_acc, 0) }
        if (active) {
            const rsx_kdnode nd = descend(sc.wnodes, node, r, ad, tmin, tmax, st, sp);
            distance = r.maxd < tmax ? r.maxd : tmax;
            items += nd.u.leaf.first_item;
            count = nd.count;
        }
        for (int32_t k = 0; __any(k < count); ++k) {
            const bool have = k < count;
            const int32_t idx = have ? items[k] : 0;
            const rsx_primitive &p = sc.prims[idx];
            double f, b;
            const bool gate = have && aabb_rcp(p.box_lower, p.box_upper, r, rx, ry, rz, f, b);   // BoundPrimitive.hit gate
            const bool is_mesh = gate && p.type == RSX_PRIM_MESH;
            Hit cand;
            cand.prim = -1;
            work += CSG ? 16 : 4;
            // Mesh primitives are traced one primitive at a time with everything about the primitive wave-uniform (matrix, mesh
            // descriptor, array bases: scalar loads, SGPRs). Coherent waves meet one instance per leaf item; a wave that straddles
            // several instances takes one turn per instance.
            unsigned long long todo = __ballot(is_mesh);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int32_t uidx = __builtin_amdgcn_readlane(idx, leader);
                const bool mine = is_mesh && idx == uidx;
                todo &= ~__ballot(mine);
                const UPrim up = uniform_prim(sc.prims, uidx);
                Ray l = r;
                if (mine) l = to_local_uniform(up, r);
                const UMesh um = (UMesh)(unsigned long long)(sc.meshes + up->mesh);
                MeshHit mh;
                if (mesh_trace_wave<STAGE_MIN>(mine, um, l, mesh_stack, mh, work, phase_acc)) {
                    cand.prim = idx; cand.t = (double)mh.t; cand.a0 = mh.tri; cand.a1 = 0; cand.u = mh.u; cand.v = mh.v; cand.w = mh.w;
                }
            }
            if (gate && !is_mesh) {
                bool needs_stream = false;
                primitive_first_hit<CSG, FASTONLY>(sc, idx, p, r, mesh_stack, csg_state, cand, needs_stream);
                if (FASTONLY && needs_stream) work |= 0x80000000u;            // top bit of the cost counter: trace this ray again with the stream merge
            }
            if (cand.prim >= 0 && cand.t <= distance) { distance = cand.t; best = cand; }   // `<=`: later item wins ties
        }
        if (active) {
            if (best.prim >= 0 || sp == 0) active = false;
            else {
                --sp;
                tmin = tmax;
                stack_pop(st, sp, node, tmax);
            }
        }
    }
    return best.prim >= 0;
}

template <bool CSG>
__device__ void finalise(const DScene &sc, const Ray &r, const Hit &h, Geom &g) {
    const rsx_primitive &p = sc.prims[h.prim];
    if constexpr (CSG) {
        if (is_csg(p.type)) { csg_geom(sc, r, h, g); return; }
    }
    const Ray l = to_local(p, r);
    if (p.type == RSX_PRIM_MESH) mesh_geom(sc.meshes[p.mesh], l, h.t, h.a0, h.u, h.v, h.w, g);
    else analytic_geom(p, l, h.t, h.a0, h.a1, g);
}

