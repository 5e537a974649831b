// dev_packet.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// Packet traversal: the 64 camera rays of a unit walk the world tree and the mesh trees TOGETHER.
#pragma once

// ---------------------------------------------------------------------------------------------------
// Why. The rays of a unit leave one point (the pinhole) through one pixel or a few neighbouring ones (unit_pixel, dev_render.hpp), so
// they visit nearly the same nodes and leaves. world_trace_wave / mesh_trace_wave let every lane walk its own way and wait for the
// slowest at each leaf: on configs[2] a wave ran 18.7 node steps per unit for rays that need 10.6, with 0.54 of the lanes busy, and
// its leaf batches ran at 0.46 — and each step cost ~70 vector instructions, half of them addressing (per-lane node ids, 64-bit
// address arithmetic, per-lane stack pointers, exec-mask bookkeeping around every branch).
//
// How. The WAVE walks the tree: node id, stack pointer and the whole control flow are wave-uniform (scalar registers, scalar branches),
// nodes and triangle records come in over the scalar data path (s_load: no vector addressing at all), and a lane keeps only what is
// its own — the range [tmin, tmax] of its ray inside the current node. A lane whose ray does not enter a child carries an EMPTY range
// there (tmax = -inf) and every test it makes fails by itself (`t < min(max_distance, -inf)` is false), so no lane is ever masked
// off explicitly.
//
// It is the reference's traversal ray by ray (kdtree3d.pyx:609-700): a ray meets exactly the nodes, in exactly the order, with exactly
// the ranges of its own recursion —
//  * `_trace_branch` visits the near child with [tmin, plane], then the far child with [plane, tmax], or only one of them with
//    [tmin, tmax]. Per lane that is: range in the first-visited child F and in the second S. The far range's tmin is not stored: it
//    equals the tmax of the last leaf the lane's ray was in (ranges of consecutively visited leaves abut), so a lane sets
//    tmin = tmax whenever it leaves a leaf in which it had a range, and the stack keeps only S's tmax per lane (8 bytes) next to the
//    wave's node id.
//  * which child is "near" depends on the ray (`below_split`: origin < split, or on the plane moving down). Rays from one origin agree
//    unless the origin lies exactly ON the plane and the directions differ in sign; lanes that cross BOTH children fix the order of the
//    visit (lanes that enter one child do not care), and when both-crossing lanes disagree the minority is set aside — the node is
//    pushed again for them alone — so each lane still sees its own near child first.
//  * a lane that found its hit in a leaf (`_trace_leaf` returned True) takes no part any more: its popped ranges are discarded.
// The division `(split - origin) / direction` is the reference's (exact quotient: refine_rcp sequence in the mesh trees, the plain
// division in the world tree, as in branch_step / world_step) — but when no lane of the wave approaches the plane (numerator zero
// or of the opposite sign to the direction: quotient <= 0, "near only" by kdtree3d.pyx:686) nobody needs the quotient and the step
// skips it; with a common origin and directions inside one pixel that is a wave-uniform outcome.
// ---------------------------------------------------------------------------------------------------

struct UNode {                 // a KD node in scalar registers
    int32_t type, count;       // type: -1 leaf | axis (+ cull bits, world copies);  count: upper child | number of items
    uint32_t lo, hi;           // branch: split (f64 bits);  leaf: first item, tag
};

struct alignas(16) PodI4 { int32_t x, y, z, w; };          // plain aggregates: loadable through the constant address space
struct alignas(16) PodF4 { float x, y, z, w; };
__device__ __forceinline__ UNode load_node_u(const rsx_kdnode *nodes, int32_t id) {
    const RSX_CONST_AS PodI4 *p = (const RSX_CONST_AS PodI4 *)(unsigned long long)(nodes + id);   // uniform address: s_load_dwordx4
    UNode nd;                                            // (field by field: the four loads merge into one)
    nd.type = p->x; nd.count = p->y; nd.lo = (uint32_t)p->z; nd.hi = (uint32_t)p->w;
    return nd;
}
__device__ __forceinline__ float4 load_f4_u(const RSX_CONST_AS PodF4 *p) { return make_float4(p->x, p->y, p->z, p->w); }
__device__ __forceinline__ double unode_split(const UNode &nd) { return __longlong_as_double((long long)(((unsigned long long)nd.hi << 32) | nd.lo)); }

// Stack of the packet: entry `sp` (wave-uniform) = the node to visit (same value in every lane's slot) + each lane's tmax there.
// Same LDS / spill layout as the per-lane stacks (Stack), so the launch plan is shared.
__device__ __forceinline__ void pstack_push(const Stack &st, int32_t sp, int32_t id, double t) {
    const int lane = (int)(threadIdx.x % WAVE);
    if (sp < st.lds_levels) {                            // (scalar branch)
        const int slot = sp * WAVE + lane;
        *reinterpret_cast<double *>(smem + st.lds_t + slot * 8) = t;
        *reinterpret_cast<int32_t *>(smem + st.lds_id + slot * 4) = id;
    } else {
        const int g = (sp - st.lds_levels) * WAVE + lane;
        reinterpret_cast<double *>(st.gt)[g] = t;
        reinterpret_cast<int32_t *>(st.gid)[g] = id;
    }
}
__device__ __forceinline__ void pstack_pop(const Stack &st, int32_t sp, int32_t &id, double &t) {
    const int lane = (int)(threadIdx.x % WAVE);
    int32_t v;
    if (sp < st.lds_levels) {
        const int slot = sp * WAVE + lane;
        t = *reinterpret_cast<const double *>(smem + st.lds_t + slot * 8);
        v = *reinterpret_cast<const int32_t *>(smem + st.lds_id + slot * 4);
    } else {
        const int g = (sp - st.lds_levels) * WAVE + lane;
        t = reinterpret_cast<const double *>(st.gt)[g];
        v = reinterpret_cast<const int32_t *>(st.gid)[g];
    }
    id = __builtin_amdgcn_readfirstlane(v);
}

#define PKT_EMPTY (-INFINITY)

// One branch node for the packet. In: the lane's range [tmin, tmax] (tmax == PKT_EMPTY: none). Out: the node to go to and the lane's
// range there; the other child, when some lane enters it too, is pushed.
// WORLD: plain division and the cull of world_step (cull bits in nd.type >> 2, t_cull); else the hoisted-reciprocal quotient.
template <bool WORLD>
__device__ __forceinline__ int32_t packet_step(const UNode &nd, int32_t node, const Ray &r, const AxisDiv &ad, double &tmin, double &tmax, const Stack &st,
                                               int32_t &sp, double t_cull) {
    const int axis = nd.type & 3;
    const double split = unode_split(nd);
    const int32_t lower = node + 1, upper = nd.count;
    const double o = sel3(axis, r.ox, r.oy, r.oz), d = sel3(axis, r.dx, r.dy, r.dz);
    const double num = split - o;
    const bool par = d == 0.0;                                             // kdtree3d.pyx:661-667
    // quotient <= 0 whatever its magnitude: zero numerator, or numerator and direction of opposite sign (kdtree3d.pyx:686 "near only")
    const bool away = num == 0.0 || ((num < 0.0) != (d < 0.0));
    bool in = tmax != PKT_EMPTY;
    double plane = 0.0;
    if (__any(in && !par && !away)) {
        if constexpr (WORLD) plane = num / d;
        else {
            const double y = sel3(axis, ad.yx, ad.yy, ad.yz);
            const double q0 = num * y;
            const double rem = __builtin_fma(-d, q0, num);
            plane = __builtin_fma(rem, y, q0);
            const bool exact = ((ad.safe >> axis) & 1) && div_operand_safe(num);
            if (__builtin_expect(__any(!exact && !away && !par), 0)) {
                if (!exact) plane = num / d;
            }
        }
    }
    const bool lower_near = par ? (o < split) : (o < split || (o == split && d < 0.0));      // kdtree3d.pyx:664, 675
    const bool near_only = par || away || plane > tmax || plane <= 0.0;
    bool far_only = !near_only && plane < tmin;
    bool both = in && !near_only && !far_only;
    if constexpr (WORLD) {
        // world_step's cull: the near subtree holds wide primitives only and ends before the nearest wide answer — straight to the far child
        const int32_t cull_bits = nd.type >> 2;
        const bool cull = both && (((lower_near ? cull_bits : cull_bits >> 1) & 1) != 0) && plane < t_cull;
        if (cull) { tmin = plane; far_only = true; both = false; }
    }
    unsigned long long b_up = __ballot(both && !lower_near);
    if (__builtin_expect(b_up != 0ULL, 0)) {
        if (__ballot(both && lower_near) != 0ULL) {
            // both-crossing lanes disagree on the near child (the common origin lies exactly on the plane): the upper-first lanes come back
            // to this node alone, after the others are through with it
            const bool defer = both && !lower_near;
            pstack_push(st, sp, node, defer ? tmax : PKT_EMPTY);
            ++sp;
            if (defer) { tmax = PKT_EMPTY; in = false; both = false; }
            b_up = 0ULL;
        }
    }
    const bool single_lower = near_only == lower_near;                     // the one child of a lane that enters one: near_only ? near : far
    const bool want_lower = in && (both || single_lower), want_upper = in && (both || !single_lower);
    const bool upper_first = b_up != 0ULL;
    const int32_t first = upper_first ? upper : lower, second = upper_first ? lower : upper;
    const bool want_f = upper_first ? want_upper : want_lower, want_s = upper_first ? want_lower : want_upper;
    const double t_f = want_f ? (both ? plane : tmax) : PKT_EMPTY;         // (a both-crossing lane's near child is `first` by construction)
    const double t_s = want_s ? tmax : PKT_EMPTY;
    if (__any(want_f)) {
        if (__any(want_s)) { pstack_push(st, sp, second, t_s); ++sp; }
        tmax = t_f;
        return first;
    }
    tmax = t_s;
    return second;
}

// Pops until some lane has a range again. `done` lanes (their ray found its hit) discard theirs. False: the stack is empty.
__device__ __forceinline__ bool packet_pop(const Stack &st, int32_t &sp, int32_t &node, double &tmax, bool done) {
    while (sp > 0) {
        --sp;
        double t;
        pstack_pop(st, sp, node, t);
        tmax = done ? PKT_EMPTY : t;
        if (__any(tmax != PKT_EMPTY)) return true;
    }
    return false;
}

// MeshData.trace (mesh.pyx:506-563) for the packet: `m` and the ray space are wave-uniform, `want` = the lane's ray passed the
// BoundPrimitive gate. Leaves of any size are walked the same way: every record comes in once over the scalar data path (the next one
// while this one is tested) and every lane with a range tests it — in leaf order, strict `<`: the reference's own loop.
__device__ bool mesh_trace_packet(bool want, UMesh m, const Ray &r, const Stack &st, MeshHit &out, uint32_t &work) {
    const rsx_kdnode *nodes = m->nodes;
    const float4 *leaf = m->leaf;
    const AxisDiv ad = axis_div(r);
    double tmin = 0, tmax = 0;
    {
        const double lo[3] = {m->lower[0], m->lower[1], m->lower[2]}, hi[3] = {m->upper[0], m->upper[1], m->upper[2]};
        const double rx = exact_div(1.0, r.dx, ad.yx, ad.safe & 1), ry = exact_div(1.0, r.dy, ad.yy, (ad.safe >> 1) & 1),
                     rz = exact_div(1.0, r.dz, ad.yz, (ad.safe >> 2) & 1);
        if (!(want && aabb_rcp(lo, hi, r, rx, ry, rz, tmin, tmax))) tmax = PKT_EMPTY;       // kdtree3d.pyx:589-607
    }
    if (!__any(tmax != PKT_EMPTY)) return false;
    const TriRay q = tri_ray(r);
    bool hit = false;
    int32_t node = 0, sp = 0;
    for (;;) {
        UNode nd = load_node_u(nodes, node);
        while (nd.type >= 0) {
            node = packet_step<false>(nd, node, r, ad, tmin, tmax, st, sp, 0.0);
            nd = load_node_u(nodes, node);
            work += 1;
        }
        const int32_t count = nd.count;
        if (count > 0) {                                                       // _trace_leaf, mesh.pyx:520-563
            double distance = r.maxd < tmax ? r.maxd : tmax;                   // (no range: -inf, nothing is accepted)
            int32_t closest = -1;
            float bu = 0, bv = 0, bw = 0;
            const RSX_CONST_AS PodF4 *rec = (const RSX_CONST_AS PodF4 *)(unsigned long long)(leaf + 4 * (size_t)nd.lo);
            float4 a = load_f4_u(rec), b = load_f4_u(rec + 1), c = load_f4_u(rec + 2);
            int32_t tri = __float_as_int(rec[3].x);
            work += (uint32_t)count;
            for (int32_t k = 0; k < count; ++k) {
                const int32_t kn = k + 1 < count ? k + 1 : k;
                const float4 na = load_f4_u(rec + 4 * kn), nb = load_f4_u(rec + 4 * kn + 1), nc = load_f4_u(rec + 4 * kn + 2);
                const int32_t ntri = __float_as_int(rec[4 * kn + 3].x);
                float ht, hu, hv, hw;
                if (tri_test(q, a, b, c, ht, hu, hv, hw) && (double)ht < distance) { distance = (double)ht; closest = tri; bu = hu; bv = hv; bw = hw; }
                a = na; b = nb; c = nc; tri = ntri;
            }
            if (closest >= 0) { out.u = bu; out.v = bv; out.w = bw; out.t = (float)distance; out.tri = closest; hit = true; }
        }
        if (tmax != PKT_EMPTY) tmin = tmax;                                    // the next range of this ray begins where this one ended
        if (!packet_pop(st, sp, node, tmax, hit)) break;
    }
    return hit;
}

// World.hit for the packet (kdtree.pyx:73-122, boundprimitive.pyx:42-51): world_trace_wave<false, false, 1, true> with the walk above.
// Leaf items are wave-uniform by construction (the wave is in ONE leaf); wide primitives, leaf tags and the cull as there.
__device__ bool world_trace_packet(bool valid, const DScene &sc, const Ray &r, const Stack &st, const Stack &mesh_stack, Hit &best, uint32_t &work) {
    best.prim = -1;
    double tmin = 0, tmax = 0;
    const double rx = 1.0 / r.dx, ry = 1.0 / r.dy, rz = 1.0 / r.dz;
    const bool enters = valid && aabb_rcp(sc.wlower, sc.wupper, r, rx, ry, rz, tmin, tmax);
    if (!enters) tmax = PKT_EMPTY;
    if (!__any(enters)) return false;
    AxisDiv ad;
    ad.yx = ad.yy = ad.yz = 0.0; ad.safe = 0;
    const rsx_kdnode *wnodes = sc.wnodes;
    WideSet8 wide;
#pragma unroll
    for (int j = 0; j < 8; ++j) wide.t[j] = -1.0;
    wide.faces[0] = wide.faces[1] = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (sc.wide[j] >= 0) {
            int32_t f = 0;
            analytic_first_root(sc, uniform_prim(sc.prims_uniform, sc.wide[j]), sc.wide[j], enters, r, rx, ry, rz, wide.t[j], f);
            wide.faces[0] |= (uint32_t)f << (8 * j);
        }
    }
    double t_cull = INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j) if (wide.t[j] >= 0.0 && wide.t[j] < t_cull) t_cull = wide.t[j];
#if RSX_WORLD_CULL == 0
    t_cull = -INFINITY;
#endif
    int32_t node = 0, sp = 0;
    for (;;) {
        UNode nd = load_node_u(wnodes, node);
        while (nd.type >= 0) {
            node = packet_step<true>(nd, node, r, ad, tmin, tmax, st, sp, t_cull);
            nd = load_node_u(wnodes, node);
            work += 1;
        }
        double distance = r.maxd < tmax ? r.maxd : tmax;                       // (no range: -inf, `t <= distance` fails)
        const int32_t tag = (int32_t)nd.hi;
        if (tag < 0) {
            // wide-only leaf: its item list rides in the node (rsx_scene_create), the answers are in registers
            const int n_tagged = (tag >> 28) & 7;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j < n_tagged) {
                    const int slot = (tag >> (3 * j)) & 7;
                    double t;
                    int32_t faces;
                    wide_lookup<2>(wide, slot, t, faces);
                    if (t >= 0.0 && t <= distance) {
                        distance = t;
                        best.prim = slot == 1 ? sc.wide[1] : sc.wide[0]; best.t = t; best.a0 = (faces & 15) - 1; best.a1 = (faces >> 4) - 1;
                        best.u = best.v = best.w = 0.0f;
                    }
                }
            }
        } else {
            const int32_t count = nd.count;
            const RSX_CONST_AS int32_t *items = (const RSX_CONST_AS int32_t *)(unsigned long long)(sc.witems + nd.lo);
            for (int32_t k = 0; k < count; ++k) {
                const int32_t idx = items[k];                                  // (scalar load)
                work += 4;
                Hit cand;
                cand.prim = -1;
                const bool in = tmax != PKT_EMPTY;
                if (idx == sc.wide[0] || idx == sc.wide[1]) {                  // answered before the traversal began
                    double t;
                    int32_t faces;
                    wide_lookup<2>(wide, idx == sc.wide[0] ? 0 : 1, t, faces);
                    if (t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
                } else {
                    const UPrim up = uniform_prim(sc.prims_uniform, idx);
                    const int32_t type = up->type;
                    if (type == RSX_PRIM_MESH) {
                        const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
                        double f, b;
                        const bool gate = in && aabb_rcp(lo, hi, r, rx, ry, rz, f, b);      // BoundPrimitive.hit gate (boundprimitive.pyx:42-51)
                        if (__any(gate)) {
                            Ray l = r;
                            if (gate) l = to_local_uniform(up, r);
                            const UMesh um = (UMesh)(unsigned long long)(sc.meshes + up->mesh);
                            MeshHit mh;
                            if (mesh_trace_packet(gate, um, l, mesh_stack, mh, work)) {
                                cand.prim = idx; cand.t = (double)mh.t; cand.a0 = mh.tri; cand.a1 = 0; cand.u = mh.u; cand.v = mh.v; cand.w = mh.w;
                            }
                        }
                    } else {                                                   // sphere / box / cylinder: Primitive.hit, first root
                        double t;
                        int32_t faces;
                        analytic_first_root(sc, up, idx, in, r, rx, ry, rz, t, faces);
                        if (t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
                    }
                }
                if (cand.prim >= 0 && cand.t <= distance) { distance = cand.t; best = cand; }   // `<=`: later item wins ties (kdtree.pyx:113)
            }
        }
        if (tmax != PKT_EMPTY) tmin = tmax;
        if (!packet_pop(st, sp, node, tmax, best.prim >= 0)) break;
    }
    return best.prim >= 0;
}
