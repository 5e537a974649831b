// dev_csg.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// CSG: lazy two-stream merge of operand roots (csg.pyx).
#pragma once

// ---------------------------------------------------------------------------------------------------
// CSG — raysect/primitive/csg.pyx:132-234 (hit / next_intersection / _identify_intersection / _closest_intersection),
//       :326-348 Union, :421-446 Intersect, :523-568 Subtract (+_modify_intersection)
//
// The reference merges two lazily evaluated, ordered root streams per CSG node and keeps the stream heads cached on the
// node object. Here the same state machine runs per lane with the per-node state in a slot array — private memory for trees of up
// to CSG_MAX_SLOTS nodes, a per-lane region of DScene::csg_arena for bigger ones — and the recursion over nested CSG nodes is a loop
// over an explicit frame stack (csg_run). Only the kernels instantiated with CSG=true contain this code.
// ---------------------------------------------------------------------------------------------------
#ifndef RSX_CSGFAST_MIN_WAVES
#define RSX_CSGFAST_MIN_WAVES 2             // waves per SIMD of the kernels that carry only the state-free CSG evaluator
#endif
#define CSG_MAX_SLOTS 16
#define F_VALID 1u
#define F_EXIT 2u
#define F_FLIP 4u

struct Rec {                       // one root of a stream
    double t, hx, hy, hz;          // distance along the (shared) ray parameter; mesh leaves: hit point in leaf space
    int32_t leaf, a0, a1;
    uint32_t flags;
    float u, v, w, pad;
};

struct NodeSt {
    Rec a, b;                      // CSG node: cached stream heads (_cache_intersection_a/_b)
    double maxd;                   // CSG node: max_distance of the ray hit() was called with
    int32_t last_is_a, invalid;    // _cache_last_intersection is a / _cache_invalid
    int32_t tested;                // BoundPrimitive._primitive_tested
    int32_t further;               // analytic leaf: cached second root
    double next_t;
    int32_t next_a0, next_a1;
    uint32_t next_flags;
    int32_t seek;                  // mesh leaf: _seek_next_intersection
    double nox, noy, noz, ndx, ndy, ndz, nmaxd, acc;   // mesh leaf: _next_local_ray, _ray_distance
};

struct CsgEval {
    const DScene *sc;
    NodeSt *st;
    Stack mesh_stack;
};

// where this lane keeps its node states: the kernel's private array, or — scenes with an operand tree of more than CSG_MAX_SLOTS nodes —
// the lane's region of the scene's arena (rsx_scene_create sizes it for the grids such scenes are launched with)
__device__ __forceinline__ NodeSt *csg_arena_slots(const DScene &sc) {
    return sc.csg_arena + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)sc.csg_arena_slots;
}

__device__ __forceinline__ bool is_csg(int type) { return type == RSX_PRIM_UNION || type == RSX_PRIM_INTERSECT || type == RSX_PRIM_SUBTRACT; }

// Mesh.hit / next_intersection as a stream (mesh.pyx:1178-1275)
__device__ __noinline__ void mesh_stream_step(CsgEval &e, int32_t idx, NodeSt &st, const Ray &l, Rec &out) {
    const DMesh &m = e.sc->meshes[e.sc->prims[idx].mesh];
    MeshHit mh;
    out.flags = 0;
    if (!mesh_trace(m, l, e.mesh_stack, mh)) { st.seek = 0; return; }
    Geom g;
    mesh_geom(m, l, (double)mh.t, mh.tri, mh.u, mh.v, mh.w, g);
    out.t = (double)mh.t + st.acc;
    out.hx = g.hit[0]; out.hy = g.hit[1]; out.hz = g.hit[2];
    out.leaf = idx; out.a0 = mh.tri; out.a1 = 0; out.u = mh.u; out.v = mh.v; out.w = mh.w;
    out.flags = F_VALID | (g.exiting ? F_EXIT : 0u);
    st.seek = 1;
    st.nox = g.hit[0] + l.dx * MESH_EPS; st.noy = g.hit[1] + l.dy * MESH_EPS; st.noz = g.hit[2] + l.dz * MESH_EPS;
    st.ndx = l.dx; st.ndy = l.dy; st.ndz = l.dz;
    st.nmaxd = l.maxd - (double)mh.t - MESH_EPS;
    st.acc = out.t + MESH_EPS;
}

__device__ __noinline__ void leaf_first(CsgEval &e, int32_t idx, NodeSt &st, const Ray &pr, Rec &out) {
    const rsx_primitive &p = e.sc->prims[idx];
    const Ray l = to_local(p, pr);
    out.flags = 0;
    st.further = 0;
    st.seek = 0;
    if (p.type == RSX_PRIM_MESH) { st.acc = 0; mesh_stream_step(e, idx, st, l, out); return; }
    Roots roots;
    roots.n = 0;
    if (p.type == RSX_PRIM_SPHERE) sphere_roots(p, l, roots);
    else if (p.type == RSX_PRIM_BOX) box_roots(p, l, roots);
    else if (p.type == RSX_PRIM_CYLINDER) cylinder_roots(p, l, roots);
    if (roots.n == 0) return;
    Geom g;
    analytic_geom(p, l, roots.t[0], roots.a0[0], roots.a1[0], g);
    out.t = roots.t[0]; out.leaf = idx; out.a0 = roots.a0[0]; out.a1 = roots.a1[0]; out.u = out.v = out.w = 0.0f;
    out.hx = out.hy = out.hz = 0.0;
    out.flags = F_VALID | (g.exiting ? F_EXIT : 0u);
    if (roots.n == 2) {
        analytic_geom(p, l, roots.t[1], roots.a0[1], roots.a1[1], g);
        st.further = 1; st.next_t = roots.t[1]; st.next_a0 = roots.a0[1]; st.next_a1 = roots.a1[1];
        st.next_flags = F_VALID | (g.exiting ? F_EXIT : 0u);
    }
}

__device__ __noinline__ void leaf_next(CsgEval &e, int32_t idx, NodeSt &st, Rec &out) {
    const rsx_primitive &p = e.sc->prims[idx];
    out.flags = 0;
    if (p.type == RSX_PRIM_MESH) {
        if (!st.seek) return;
        Ray l;
        l.ox = st.nox; l.oy = st.noy; l.oz = st.noz; l.dx = st.ndx; l.dy = st.ndy; l.dz = st.ndz; l.maxd = st.nmaxd;
        mesh_stream_step(e, idx, st, l, out);
        return;
    }
    if (!st.further) return;
    st.further = 0;
    out.t = st.next_t; out.leaf = idx; out.a0 = st.next_a0; out.a1 = st.next_a1; out.u = out.v = out.w = 0.0f;
    out.hx = out.hy = out.hz = 0.0;
    out.flags = st.next_flags;
}

// operator truth tables on (inside_a, inside_b, which stream supplied the closest root)
__device__ __forceinline__ bool csg_valid(int type, const Rec &a, const Rec &b, bool closest_is_a) {
    const bool ia = (a.flags & F_VALID) && (a.flags & F_EXIT), ib = (b.flags & F_VALID) && (b.flags & F_EXIT);
    if (type == RSX_PRIM_UNION) return (!ia && !ib) || (ia && !ib && closest_is_a) || (!ia && ib && !closest_is_a);
    if (type == RSX_PRIM_INTERSECT) return (ia && ib) || (ia && !ib && !closest_is_a) || (!ia && ib && closest_is_a);
    return (!ia && !ib && closest_is_a) || (ia && !ib) || (ia && ib && !closest_is_a);
}

// _closest_intersection: 1 = a, 0 = b, -1 = none (a wins only when strictly closer)
__device__ __forceinline__ int csg_closest(const Rec &a, const Rec &b) {
    if (!(a.flags & F_VALID)) return (b.flags & F_VALID) ? 0 : -1;
    if (!(b.flags & F_VALID) || a.t < b.t) return 1;
    return 0;
}

// The merge as one loop over an explicit stack (csg_run). The reference recurses: hit() / next_intersection() of a CSG node call those
// of its operands (csg.pyx:132-234, through BoundPrimitive, boundprimitive.pyx:42-60), to any depth. Here a frame is (node, where to
// resume) in a per-lane array; everything else a frame of the recursion would hold lives in the node's state slot — the two stream
// heads are advanced IN PLACE (the reference advances local copies and stores them only when it returns a root; the copies it drops
// belong to a stream that has just ended, and an ended stream is never asked again: _closest_intersection cannot select a head that
// is None), and a CSG node keeps the ray in its own space in the slot's mesh-leaf fields, which it does not use otherwise.
// Depth is bounded by the frame array alone (CSG_STACK_MAX nested levels), the number of nodes by the state slots the launch provides.
#ifndef CSG_STACK_MAX
#define CSG_STACK_MAX 64
#endif
enum { CSGP_FIRST = 0, CSGP_GOT_FA, CSGP_GOT_FB, CSGP_NEXT, CSGP_GOT_NA, CSGP_GOT_NB, CSGP_IDENT };

__device__ __noinline__ void csg_run(CsgEval &e, int32_t top, bool first, const Ray &pr, Rec &out) {
    int32_t frames[CSG_STACK_MAX];              // (node << 3) | phase to resume at
    int sp = 0;
    int32_t cur = top;
    int phase = first ? CSGP_FIRST : CSGP_NEXT;
    Ray ray_in = pr;                            // CSGP_FIRST: the ray in the PARENT's space (hit() argument)
    Rec ret;
    ret.flags = 0;
    bool finished = false;                      // (one loop, one exit test, no `continue` inside: see node_contains)
    while (!finished) {
        const rsx_primitive &p = e.sc->prims[cur];
        NodeSt &st = e.st[e.sc->csg[cur].slot];
        int32_t child = -1;
        bool want_first = false;
        int resume = 0;
        bool done = false;                      // this node's call has its result in `ret`
        if (phase == CSGP_FIRST) {                                                                 // CSGPrimitive.hit, csg.pyx:132-155
            st.invalid = 1;
            st.maxd = ray_in.maxd;
            const Ray l = to_local(p, ray_in);
            st.nox = l.ox; st.noy = l.oy; st.noz = l.oz; st.ndx = l.dx; st.ndy = l.dy; st.ndz = l.dz;   // (its max_distance is INFINITY)
            child = p.child_a; want_first = true; resume = CSGP_GOT_FA;
        } else if (phase == CSGP_GOT_FA) {
            st.a = ret;
            if (p.type != RSX_PRIM_UNION && !(ret.flags & F_VALID)) { ret.flags = 0; done = true; }    // terminate_early
            else { child = p.child_b; want_first = true; resume = CSGP_GOT_FB; }
        } else if (phase == CSGP_NEXT) {                                                           // next_intersection, csg.pyx:160-179
            if (st.invalid) { ret.flags = 0; done = true; }
            else if (st.last_is_a) { child = p.child_a; resume = CSGP_GOT_NA; }
            else { child = p.child_b; resume = CSGP_GOT_NB; }
        } else {
            if (phase == CSGP_GOT_FB || phase == CSGP_GOT_NB) st.b = ret;
            else if (phase == CSGP_GOT_NA) st.a = ret;
            // _identify_intersection, csg.pyx:181-222 — one turn of its loop per visit of this block
            const int closest = csg_closest(st.a, st.b);
            if (closest < 0) { ret.flags = 0; done = true; }
            else if (csg_valid(p.type, st.a, st.b, closest != 0)) {
                const Rec &c = closest ? st.a : st.b;
                if (c.t <= st.maxd) {
                    st.last_is_a = closest; st.invalid = 0;
                    ret = c;
                    if (p.type == RSX_PRIM_SUBTRACT && !closest) ret.flags ^= (F_EXIT | F_FLIP);       // _modify_intersection
                } else ret.flags = 0;
                done = true;
            } else if (closest) { child = p.child_a; resume = CSGP_GOT_NA; }
            else { child = p.child_b; resume = CSGP_GOT_NB; }
        }
        if (!done) {
            // BoundPrimitive.hit / next_intersection over the operand (boundprimitive.pyx:42-60): a leaf answers at once, a CSG operand
            // becomes the current node and this one waits in a frame
            const rsx_primitive &cp = e.sc->prims[child];
            NodeSt &cst = e.st[e.sc->csg[child].slot];
            bool answered = true;
            ret.flags = 0;
            if (want_first) {
                Ray l;
                l.ox = st.nox; l.oy = st.noy; l.oz = st.noz; l.dx = st.ndx; l.dy = st.ndy; l.dz = st.ndz; l.maxd = INFINITY;
                double f, b;
                if (!aabb(cp.box_lower, cp.box_upper, l, f, b)) cst.tested = 0;
                else {
                    cst.tested = 1;
                    if (is_csg(cp.type)) { answered = false; ray_in = l; }
                    else if (cp.type != RSX_PRIM_NULL) leaf_first(e, child, cst, l, ret);
                }
            } else if (cst.tested) {
                if (is_csg(cp.type)) answered = false;
                else if (cp.type != RSX_PRIM_NULL) leaf_next(e, child, cst, ret);
            }
            if (answered) phase = resume;
            else {
                frames[sp] = (cur << 3) | resume;               // (rsx_scene_create refuses trees nested deeper than the array)
                sp += 1;
                cur = child;
                phase = want_first ? CSGP_FIRST : CSGP_NEXT;
            }
        } else if (sp == 0) {
            finished = true;
        } else {
            sp -= 1;
            const int32_t fr = frames[sp];
            cur = fr >> 3;
            phase = fr & 7;
        }
    }
    out = ret;
}

// (the two homes of the node states are two calls rather than one call through a selected pointer: the callee then sees a private or a
// global address, not a generic one)
__device__ __forceinline__ void csg_call(CsgEval &e, int32_t idx, bool first, const Ray &pr, Rec &out) {
    if (e.sc->csg_arena) {
        CsgEval big;
        big.sc = e.sc; big.st = csg_arena_slots(*e.sc); big.mesh_stack = e.mesh_stack;
        csg_run(big, idx, first, pr, out);
    } else csg_run(e, idx, first, pr, out);
}
__device__ __forceinline__ void csg_first(CsgEval &e, int32_t idx, const Ray &pr, Rec &out) { csg_call(e, idx, true, pr, out); }      // CSGPrimitive.hit
__device__ __forceinline__ void csg_next(CsgEval &e, int32_t idx, Rec &out) {                                                          // next_intersection
    Ray none;
    none.ox = none.oy = none.oz = none.dx = none.dy = none.dz = 0.0; none.maxd = 0.0;
    csg_call(e, idx, false, none, out);
}

// ---------------------------------------------------------------------------------------------------
// Fast first hit of a CSG tree of analytic leaves.
//
// The stream merge above keeps ~260 B of cached state per operand node per lane in scratch: on demos/csg.py's tree that is 150 KB
// of scattered state per wave, and the kernel waits on it 77 % of the time. For hit() — the only entry the world traversal needs
// — of a tree whose leaves are convex analytic solids, the streams can be merged without any state: every leaf has at most two
// roots (enter, exit); walking all leaf roots in increasing t, the first root across which "inside the solid" flips is the first
// root the reference's merge accepts: csg_valid() is the truth table of exactly that flip, with "inside operand" read off the
// exit flag of the operand's next root, and a nested node's stream is by induction the list of its own flips. The two orders can
// differ only when a leaf reports an inconsistent enter/exit pattern (grazing), which is detected and sent to the stream merge.
// Exact ties between roots of different leaves (coplanar faces, e.g. the two cutting boxes of demos/prism.py's prism) are taken
// in the merge's own order: it prefers operand b at every node (csg.pyx:231-234), i.e. the later leaf in depth-first numbering.
// Leaf roots live in the (idle) mesh-stack LDS levels: t in the f64 array, (face, axis, exit) packed in the i32 array.
// Returns 1 = hit (cand filled), 0 = no hit, -1 = use the stream merge.
#ifdef CSGF_COUNT
__device__ double g_csgf_ex[8];                 // one example of a tie: idx, leaf a, leaf b, t, ray origin
__device__ unsigned long long g_csgf_why[4];    // diagnostic builds: why csg_fast_hit gave up [nan, enter/exit pattern, tie, steps]
#define CSGF_WHY(k) atomicAdd(&g_csgf_why[k], 1ULL)
#else
#define CSGF_WHY(k)
#endif
// to_local for the per-lane evaluator, with what the host has noted about the record (rsx_scene_create, bits of the device copy's `pad`):
// bit 1 — the matrix is affine: for a finite point Point3D.transform's w is (0 * x + 0 * y) + 0 * z + 1 = 1 exactly and x * (1.0 / 1.0) = x,
// so the w row, its division and the three multiplications go; bit 0 — the rotation part is the identity: a direction without a zero or
// non-finite component comes out as it went in (`kept`: the caller then keeps its reciprocals too). Same bits as to_local() either way.
#ifndef RSX_CSG_LANE_FLAGS
#define RSX_CSG_LANE_FLAGS 0             // measured (round 5, configs[4]): 7.94 -> 11.0 s per step with the flags read per lane — the per-lane tests and the second
#endif                                  // copy of every transform cost the 256-register kernel more than the arithmetic they skip; 0: to_local() as before

__device__ __forceinline__ Ray to_local_flagged(const rsx_primitive &p, const Ray &q, bool &kept) {
    const double *m = p.to_local;
    const int flags = RSX_CSG_LANE_FLAGS ? p.pad : 0;
    Ray l;
    if ((flags & 2) && __builtin_amdgcn_class(q.ox, 0x1f8) && __builtin_amdgcn_class(q.oy, 0x1f8) && __builtin_amdgcn_class(q.oz, 0x1f8)) {   // finite
        l.ox = m[0] * q.ox + m[1] * q.oy + m[2] * q.oz + m[3];
        l.oy = m[4] * q.ox + m[5] * q.oy + m[6] * q.oz + m[7];
        l.oz = m[8] * q.ox + m[9] * q.oy + m[10] * q.oz + m[11];
    } else xform_point(m, q.ox, q.oy, q.oz, l.ox, l.oy, l.oz);
    kept = (flags & 1) && __builtin_amdgcn_class(q.dx, 0x198) && __builtin_amdgcn_class(q.dy, 0x198) && __builtin_amdgcn_class(q.dz, 0x198);   // finite, not zero
    if (kept) { l.dx = q.dx; l.dy = q.dy; l.dz = q.dz; }
    else xform_vector(m, q.dx, q.dy, q.dz, l.dx, l.dy, l.dz);
    l.maxd = q.maxd;
    return l;
}
// box_roots with the reciprocals of the ray's direction components given (box_slab forms 1.0 / d itself and multiplies by it: the same quotients)
__device__ __forceinline__ void box_roots_rcp(const rsx_primitive &p, const Ray &l, double rx, double ry, double rz, Roots &out) {
    double near_t = -INFINITY, far_t = INFINITY;
    int nf = NO_FACE, ff = NO_FACE, na = -1, fa = -1;
    box_slab_rcp(0, l.ox, l.dx, rx, p.params[0], p.params[3], near_t, far_t, nf, ff, na, fa);
    box_slab_rcp(1, l.oy, l.dy, ry, p.params[1], p.params[4], near_t, far_t, nf, ff, na, fa);
    box_slab_rcp(2, l.oz, l.dz, rz, p.params[2], p.params[5], near_t, far_t, nf, ff, na, fa);
    pick_roots(near_t, far_t, nf, na, ff, fa, l.maxd, out);
}

__device__ int csg_fast_hit(const DScene &sc, int32_t idx, const Ray &r, const Stack &ms, Hit &cand) {
    const CsgFast &P = sc.csgfast[idx];
    const int lane = threadIdx.x % WAVE;
    double *lds_t = reinterpret_cast<double *>(smem + ms.lds_t);
    int32_t *lds_m = reinterpret_cast<int32_t *>(smem + ms.lds_id);
    bool kept;
    Ray l0 = to_local_flagged(sc.prims[idx], r, kept);
    l0.maxd = INFINITY;
    uint32_t nroots = 0;                                     // 2 bits per leaf
    uint32_t lone_exit = 0;                                  // bit k: leaf k has one root, an exit (the ray starts inside it)
    // The chain of a leaf is walked from the top unless it only differs from the previous leaf's in its last entry (siblings):
    // then the parent-space ray, its reciprocals and the verdict of the gates above are kept. The reciprocals 1 / d are shared by
    // every box tested in one space (aabb_rcp: bit-identical to aabb).
    const double l0rx = 1.0 / l0.dx, l0ry = 1.0 / l0.dy, l0rz = 1.0 / l0.dz;
    Ray cur = l0;
    double crx = l0rx, cry = l0ry, crz = l0rz;
    bool prefix_alive = true;
    for (int k = 0; k < P.n_leaves; ++k) {
        // CSGPrimitive.hit's early exit (csg.pyx:148-150): operand a of an Intersect / Subtract has no root -> no hit, operand b is
        // not even looked at
        if (k == P.top_a_leaves && P.top_type != RSX_PRIM_UNION && nroots == 0) return 0;
        const int len = P.chain_len[k];
        bool sibling = k > 0 && P.chain_len[k - 1] == len;
        for (int j = 0; sibling && j + 1 < len; ++j) sibling = P.chain[k][j] == P.chain[k - 1][j];
        if (!sibling) {
            cur = l0; crx = l0rx; cry = l0ry; crz = l0rz;
            prefix_alive = true;
            for (int j = 0; j + 1 < len; ++j) {              // BoundPrimitive gates on the way down (boundprimitive.pyx:42-51)
                const rsx_primitive &node = sc.prims[P.chain[k][j]];
                double f, b;
                if (!aabb_rcp(node.box_lower, node.box_upper, cur, crx, cry, crz, f, b)) { prefix_alive = false; break; }
                cur = to_local_flagged(node, cur, kept);     // csg_first: the operands see the ray in the node's space
                if (!kept) { crx = 1.0 / cur.dx; cry = 1.0 / cur.dy; crz = 1.0 / cur.dz; }
            }
        }
        if (!prefix_alive) continue;
        if (P.guard_hi[k] > P.guard_lo[k]) {                 // operand a of the node this leaf is (part of) operand b of has no root: not looked at
            const uint32_t range = ((1u << (2 * P.guard_hi[k])) - 1u) & ~((1u << (2 * P.guard_lo[k])) - 1u);
            if ((nroots & range) == 0u) continue;
        }
        const rsx_primitive &leaf = sc.prims[P.leaf[k]];
        {
            double f, b;
            if (!aabb_rcp(leaf.box_lower, leaf.box_upper, cur, crx, cry, crz, f, b)) continue;
        }
        const Ray ll = to_local_flagged(leaf, cur, kept);
        Roots roots;
        roots.n = 0;
        if (leaf.type == RSX_PRIM_SPHERE) sphere_roots(leaf, ll, roots);
        else if (leaf.type == RSX_PRIM_BOX) { if (kept) box_roots_rcp(leaf, ll, crx, cry, crz, roots); else box_roots(leaf, ll, roots); }
        else cylinder_roots(leaf, ll, roots);
        for (int j = 0; j < roots.n; ++j) {
            const bool exiting = analytic_exiting(leaf, ll, roots.t[j], roots.a0[j], roots.a1[j]);
            if (!(roots.t[j] == roots.t[j])) { CSGF_WHY(0); return -1; }
            if (exiting != (j == roots.n - 1)) { CSGF_WHY(1); return -1; }    // convex solid: (enter, exit) or a lone exit; anything else: stream merge
            lds_t[(2 * k + j) * WAVE + lane] = roots.t[j];
            lds_m[(2 * k + j) * WAVE + lane] = (roots.a0[j] & 0xff) | ((roots.a1[j] & 0xff) << 8) | ((exiting ? 1 : 0) << 16);
        }
        if (roots.n == 1) lone_exit |= 1u << k;
        nroots |= (uint32_t)roots.n << (2 * k);
    }
    // "inside the solid" is the tree's truth table (CsgFast::truth) looked up with the "inside leaf k" bits: a leaf is inside when
    // its next root is an exit, i.e. at the start when it has a lone exit root
    uint32_t consumed = 0, inside_bits = lone_exit, solid = csg_truth(P, lone_exit);
    for (int step = 0; step <= 2 * P.n_leaves; ++step) {
        int best = -1;
        double best_t = INFINITY;
        for (int k = 0; k < P.n_leaves; ++k) {
            const uint32_t c = (consumed >> (2 * k)) & 3u, n = (nroots >> (2 * k)) & 3u;
            if (c >= n) continue;
            const double t = lds_t[(2 * k + (int)c) * WAVE + lane];
            // exact ties go to the LATER leaf: leaves are numbered operand a before operand b at every node, and the reference's
            // merge takes b's root when the two heads tie (csg.pyx:231-234), level by level
            if (best < 0 || t <= best_t) { best = k; best_t = t; }
        }
        if (best < 0) return 0;
        const uint32_t c = (consumed >> (2 * best)) & 3u;
        consumed += 1u << (2 * best);
        inside_bits ^= 1u << best;                           // an enter root puts the ray inside the leaf, an exit root outside
        const uint32_t before = solid;
        solid = csg_truth(P, inside_bits);
        if (solid == before) continue;                       // not a surface of the solid: csg_valid() rejects it, next root
        if (!(best_t <= r.maxd)) return 0;                   // csg_identify: accepted only within the ray's reach
        const int32_t m = lds_m[(2 * best + (int)c) * WAVE + lane];
        const uint32_t exiting = (uint32_t)(m >> 16) & 1u, parity = (uint32_t)P.parity[best] & 1u;
        cand.prim = idx; cand.t = best_t;
        cand.a0 = (int32_t)(int8_t)(m & 0xff); cand.a1 = (int32_t)(int8_t)((m >> 8) & 0xff);
        cand.u = cand.v = cand.w = 0.0f;
        cand.leaf = P.leaf[best];
        cand.flags = F_VALID | ((exiting ^ parity) ? F_EXIT : 0u) | (parity ? F_FLIP : 0u);
        cand.hx = cand.hy = cand.hz = 0.0;
        return 1;
    }
    return -1;
}

// csg_fast_hit for a wave whose lanes all ask about the SAME top-level primitive (primary rays: the leaf items of a coherent wave are
// one primitive at a time, world_trace_wave) — every lane calls it, `want` marks the lanes that ask. The flattened tree, every node's
// and leaf's box, matrix and parameters come in over the scalar data path (no 376-byte record per lane and operand), the walk over
// leaves and chains is scalar control flow, and a lane that is done is switched off by a predicate: the function has no exit in the
// middle of a loop (see the toolchain note at node_contains). Same operations on the same values per lane as csg_fast_hit.
// Returns per lane 1 = hit (cand filled), 0 = no hit, -1 = use the stream merge; lanes without `want` get 0.
__device__ __forceinline__ uint32_t csg_truth_uniform(const RSX_CONST_AS CsgFast *P, uint32_t inside_bits) {
    const uint64_t w0 = P->truth[0], w1 = P->truth[1], w2 = P->truth[2], w3 = P->truth[3];
    const uint32_t word = inside_bits >> 6;
    const uint64_t w = word == 0 ? w0 : word == 1 ? w1 : word == 2 ? w2 : w3;
    return (uint32_t)(w >> (inside_bits & 63u)) & 1u;
}

// (the scene's tables are passed one by one: the packet walk holds the scene behind a kernel-argument pointer, the per-lane walks by value)
__device__ int csg_fast_hit_uniform(const CsgFast *csgfast, const rsx_primitive *prims_uniform, const rsx_primitive *prims, int32_t uidx, bool want, const Ray &r,
                                    const Stack &ms, Hit &cand) {
    const RSX_CONST_AS CsgFast *P = (const RSX_CONST_AS CsgFast *)(unsigned long long)(csgfast + uidx);
    const int lane = threadIdx.x % WAVE;
    double *lds_t = reinterpret_cast<double *>(smem + ms.lds_t);
    int32_t *lds_m = reinterpret_cast<int32_t *>(smem + ms.lds_id);
    const int n_leaves = P->n_leaves, top_a_leaves = P->top_a_leaves, top_type = P->top_type;
    // A record whose to_local keeps directions (rsx_scene_create: bit 0 of the device copy's `pad` — translate() and default transforms,
    // most nodes of a CSG tree) hands a direction without a zero or non-finite component on bit for bit ((1 * d + 0 * e) + 0 * f = d; a zero
    // could change its sign, a non-finite component turns the zeros into NaN), and with it its reciprocals: the nine multiply-adds and
    // the three divisions of such a step are skipped when every asking lane's CURRENT direction is of that kind (`dir_ok`, re-established
    // by three class tests whenever the direction has changed).
    auto dirs_plain = [&](const Ray &q) {                    // wave-uniform: every component of every asking lane's direction is finite and not zero
        const bool ok = __builtin_amdgcn_class(q.dx, 0x198) && __builtin_amdgcn_class(q.dy, 0x198) && __builtin_amdgcn_class(q.dz, 0x198);   // -normal | -denormal | +denormal | +normal
        return RSX_CSG_KEEP_DIRECTION != 0 && __builtin_amdgcn_ballot_w64(want && !ok) == 0ULL;
    };
    auto origin_only = [](UPrim p, const Ray &q, double &ox, double &oy, double &oz) {   // to_local_uniform's origin for an affine matrix (w = 1.0, x * 1.0 = x)
        const RSX_CONST_AS double *m = p->to_local;
        ox = (m[0] * q.ox + m[1] * q.oy + m[2] * q.oz + m[3]) * 1.0;
        oy = (m[4] * q.ox + m[5] * q.oy + m[6] * q.oz + m[7]) * 1.0;
        oz = (m[8] * q.ox + m[9] * q.oy + m[10] * q.oz + m[11]) * 1.0;
    };
    const UPrim top = uniform_prim(prims_uniform, uidx);
    Ray l0;
    bool l0_ok;
    if (dirs_plain(r) && (top->pad & 1) != 0) { l0 = r; origin_only(top, r, l0.ox, l0.oy, l0.oz); l0_ok = true; }
    else { l0 = to_local_uniform(top, r); l0_ok = dirs_plain(l0); }
    l0.maxd = INFINITY;
    int result = 0;
    bool on = want;                                          // this lane is still being answered
    uint32_t nroots = 0, lone_exit = 0;
    const double l0rx = 1.0 / l0.dx, l0ry = 1.0 / l0.dy, l0rz = 1.0 / l0.dz;
    bool dir_ok = l0_ok;
    Ray cur = l0;
    double crx = l0rx, cry = l0ry, crz = l0rz;
    bool prefix_alive = true;
    for (int k = 0; k < n_leaves; ++k) {
        // CSGPrimitive.hit's early exit (csg.pyx:148-150): operand a of an Intersect / Subtract has no root -> no hit
        if (k == top_a_leaves && top_type != RSX_PRIM_UNION) { if (on && nroots == 0) { result = 0; on = false; } }
        const int len = P->chain_len[k];
        bool sibling = k > 0 && P->chain_len[k > 0 ? k - 1 : 0] == len;
        for (int j = 0; sibling && j + 1 < len; ++j) sibling = P->chain[k][j] == P->chain[k > 0 ? k - 1 : 0][j];
        if (!sibling) {
            cur = l0; crx = l0rx; cry = l0ry; crz = l0rz;
            prefix_alive = true;
            dir_ok = l0_ok;
            for (int j = 0; j + 1 < len; ++j) {              // BoundPrimitive gates on the way down (boundprimitive.pyx:42-51)
                const UPrim node = uniform_prim(prims_uniform, P->chain[k][j]);
                const double lo[3] = {node->box_lower[0], node->box_lower[1], node->box_lower[2]}, hi[3] = {node->box_upper[0], node->box_upper[1], node->box_upper[2]};
                double f, b;
                const bool through = aabb_rcp(lo, hi, cur, crx, cry, crz, f, b);
                // (a lane whose gate failed keeps its ray where it stopped; it is not looked at again before the next reset)
                const bool go = prefix_alive && through;
                if (dir_ok && (node->pad & 1) != 0) {                        // (wave-uniform) the node keeps directions: only the origin moves
                    double dox, doy, doz;
                    origin_only(node, cur, dox, doy, doz);
                    cur.ox = go ? dox : cur.ox; cur.oy = go ? doy : cur.oy; cur.oz = go ? doz : cur.oz;
                } else {
                    const Ray down = to_local_uniform(node, cur);
                    cur.ox = go ? down.ox : cur.ox; cur.oy = go ? down.oy : cur.oy; cur.oz = go ? down.oz : cur.oz;
                    cur.dx = go ? down.dx : cur.dx; cur.dy = go ? down.dy : cur.dy; cur.dz = go ? down.dz : cur.dz;
                    crx = 1.0 / cur.dx; cry = 1.0 / cur.dy; crz = 1.0 / cur.dz;
                    dir_ok = dirs_plain(cur);
                }
                prefix_alive = go;
            }
        }
        const int32_t leaf_id = P->leaf[k];
        const UPrim leaf = uniform_prim(prims_uniform, leaf_id);
        bool meets = on && prefix_alive;
        {
            const int glo = P->guard_lo[k], ghi = P->guard_hi[k];    // (see csg_fast_hit)
            if (ghi > glo) meets = meets && (nroots & (((1u << (2 * ghi)) - 1u) & ~((1u << (2 * glo)) - 1u))) != 0u;
        }
        {
            const double lo[3] = {leaf->box_lower[0], leaf->box_lower[1], leaf->box_lower[2]}, hi[3] = {leaf->box_upper[0], leaf->box_upper[1], leaf->box_upper[2]};
            double f, b;
            meets = aabb_rcp(lo, hi, cur, crx, cry, crz, f, b) && meets;
        }
        if (__builtin_amdgcn_ballot_w64(meets) != 0ULL) {
            // (a leaf that keeps directions: the ray's direction and its reciprocals are the parent space's)
            const bool leaf_keeps = dir_ok && (leaf->pad & 1) != 0;
            Ray ll;
            if (leaf_keeps) { ll = cur; origin_only(leaf, cur, ll.ox, ll.oy, ll.oz); }
            else ll = to_local_uniform(leaf, cur);
            const int32_t type = leaf->type;
            Roots roots;
            roots.n = 0;
            if (type == RSX_PRIM_SPHERE) sphere_roots_uniform(leaf->params[0], ll, roots);
            else if (type == RSX_PRIM_BOX) {
                const double prm[6] = {leaf->params[0], leaf->params[1], leaf->params[2], leaf->params[3], leaf->params[4], leaf->params[5]};
                box_roots_uniform(prm, ll, leaf_keeps, crx, cry, crz, roots);
            } else cylinder_roots(prims[leaf_id], ll, roots);
            const int n = meets ? roots.n : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j < n) {
                    const bool exiting = analytic_exiting(type, ll, roots.t[j], roots.a0[j], roots.a1[j]);
                    // NaN roots, or anything but (enter, exit) / a lone exit of a convex solid: the stream merge answers this lane
                    if (!(roots.t[j] == roots.t[j]) || exiting != (j == n - 1)) { if (on) result = -1; on = false; }
                    lds_t[(2 * k + j) * WAVE + lane] = roots.t[j];
                    lds_m[(2 * k + j) * WAVE + lane] = (roots.a0[j] & 0xff) | ((roots.a1[j] & 0xff) << 8) | ((exiting ? 1 : 0) << 16);
                }
            }
            if (n == 1) lone_exit |= 1u << k;
            nroots |= (uint32_t)n << (2 * k);
        }
    }
    // "inside the solid" is the tree's truth table looked up with the "inside leaf k" bits (see csg_fast_hit)
    uint32_t consumed = 0, inside_bits = lone_exit, solid = csg_truth_uniform(P, lone_exit);
    for (int step = 0; step <= 2 * n_leaves && __builtin_amdgcn_ballot_w64(on) != 0ULL; ++step) {
        int best = -1;
        double best_t = INFINITY;
        for (int k = 0; k < n_leaves; ++k) {
            const uint32_t c = (consumed >> (2 * k)) & 3u, n = (nroots >> (2 * k)) & 3u;
            const bool has = on && c < n;
            const double t = has ? lds_t[(2 * k + (int)c) * WAVE + lane] : 0.0;
            // exact ties go to the LATER leaf (csg.pyx:231-234), level by level
            if (has && (best < 0 || t <= best_t)) { best = k; best_t = t; }
        }
        if (on && best < 0) { result = 0; on = false; }
        if (on) {
            const uint32_t c = (consumed >> (2 * best)) & 3u;
            consumed += 1u << (2 * best);
            inside_bits ^= 1u << best;                       // an enter root puts the ray inside the leaf, an exit root outside
            const uint32_t before = solid;
            solid = csg_truth_uniform(P, inside_bits);
            if (solid != before) {                           // a surface of the solid (otherwise csg_valid() rejects it: next root)
                if (!(best_t <= r.maxd)) result = 0;         // csg_identify: accepted only within the ray's reach
                else {
                    const int32_t m = lds_m[(2 * best + (int)c) * WAVE + lane];
                    int32_t parity_bit = 0, leaf_of_best = 0;
                    for (int k = 0; k < n_leaves; ++k) if (k == best) { parity_bit = P->parity[k] & 1; leaf_of_best = P->leaf[k]; }
                    const uint32_t exiting = (uint32_t)(m >> 16) & 1u, parity = (uint32_t)parity_bit;
                    cand.prim = uidx; cand.t = best_t;
                    cand.a0 = (int32_t)(int8_t)(m & 0xff); cand.a1 = (int32_t)(int8_t)((m >> 8) & 0xff);
                    cand.u = cand.v = cand.w = 0.0f;
                    cand.leaf = leaf_of_best;
                    cand.flags = F_VALID | ((exiting ^ parity) ? F_EXIT : 0u) | (parity ? F_FLIP : 0u);
                    cand.hx = cand.hy = cand.hz = 0.0;
                    result = 1;
                }
                on = false;
            }
        }
    }
    if (on) result = -1;                                     // (more roots than the steps allow: cannot happen; as csg_fast_hit)
    return result;
}

// contains(): csg.pyx:350-353, :448-451, :570-573 over BoundPrimitive.contains (box gate + primitive.contains)
// MESHES false: the caller knows the scene holds no mesh (the ray cast and its registers are left out)
template <bool MESHES = true>
__device__ bool leaf_contains(const DScene &sc, const rsx_primitive &p, double px, double py, double pz, Stack mesh_stack) {
    double qx, qy, qz;
    xform_point(p.to_local, px, py, pz, qx, qy, qz);
    if (p.type == RSX_PRIM_SPHERE) return (qx * qx + qy * qy + qz * qz) <= p.params[0] * p.params[0];
    if (p.type == RSX_PRIM_BOX) return aabb_contains(p.params, p.params + 3, qx, qy, qz);
    if (p.type == RSX_PRIM_CYLINDER) return (0.0 <= qz && qz <= p.params[1]) && ((qx * qx + qy * qy) <= (p.params[0] * p.params[0]));
    if constexpr (MESHES) if (p.type == RSX_PRIM_MESH) {
        const DMesh &m = sc.meshes[p.mesh];
        if (!m.closed) return false;
        Ray zr;
        zr.ox = qx; zr.oy = qy; zr.oz = qz; zr.dx = 0; zr.dy = 0; zr.dz = 1; zr.maxd = INFINITY;
        MeshHit mh;
        if (mesh_trace(m, zr, mesh_stack, mh)) return m.tris[3 * (size_t)mh.tri + 2].w > 0.0f;
    }
    return false;
}

// contains() of a flattened analytic operand tree (csg.pyx:350-353, :448-451, :570-573 over BoundPrimitive.contains): the point is
// taken down each leaf's chain (box gate in the parent's space, then into the node's space), the leaf tests give one bit each and
// the postfix program combines them. A gate that fails anywhere leaves the bits below it clear, and any union / intersection /
// difference of all-clear operands is clear — which is what the reference's early `return False` yields for that node.
__device__ bool csg_fast_contains(const DScene &sc, int32_t idx, double px, double py, double pz, const Stack &ms) {
    const CsgFast &P = sc.csgfast[idx];
    const rsx_primitive &top = sc.prims[idx];
    if (!aabb_contains(top.box_lower, top.box_upper, px, py, pz)) return false;
    double tx, ty, tz;
    xform_point(top.to_local, px, py, pz, tx, ty, tz);
    uint32_t inside = 0;
    for (int k = 0; k < P.n_leaves; ++k) {
        double x = tx, y = ty, z = tz;
        bool alive = true;
        for (int j = 0; j < P.chain_len[k]; ++j) {
            const rsx_primitive &node = sc.prims[P.chain[k][j]];
            if (!aabb_contains(node.box_lower, node.box_upper, x, y, z)) { alive = false; break; }
            if (j + 1 < P.chain_len[k]) { double nx, ny, nz; xform_point(node.to_local, x, y, z, nx, ny, nz); x = nx; y = ny; z = nz; }
        }
        if (alive && leaf_contains(sc, sc.prims[P.leaf[k]], x, y, z, ms)) inside |= 1u << k;
    }
    return csg_truth(P, inside) != 0;
}

// contains() of any operand tree: the recursion of csg.pyx:350-353 / :448-451 / :570-573 over BoundPrimitive.contains as a loop over
// an explicit stack of (node, what is known of it); a CSG node's point in its own space is kept per level.
// (One loop with one exit test and no `continue` / `return` inside: written with early continues and a return in the middle, hipcc 7.2
// compiled a loop that is right for a single lane — traced with printf — and wrong when the lanes of a wave leave it at different
// turns; tests/test_gpu_parity.py::test_csg_trees_of_any_depth_and_size_vs_oracle pins this form.)
__device__ __noinline__ bool node_contains(const DScene &sc, int32_t top, double px, double py, double pz, Stack mesh_stack) {
    int32_t frames[CSG_STACK_MAX];              // (node << 1) | 1 when operand a is known and operand b is being evaluated
    double pts[3 * (CSG_STACK_MAX + 1)];        // level k: the point in the space the node at level k is tested in
    int sp = 0;
    int32_t cur = top;
    pts[0] = px; pts[1] = py; pts[2] = pz;
    bool result = false;
    bool have = false;                          // `result` answers the node whose frame was popped last (or `top`)
    bool finished = false;
    while (!finished) {
        if (!have) {
            const rsx_primitive &p = sc.prims[cur];
            const double x = pts[3 * sp], y = pts[3 * sp + 1], z = pts[3 * sp + 2];
            if (!aabb_contains(p.box_lower, p.box_upper, x, y, z)) { result = false; have = true; }
            else if (!is_csg(p.type)) { result = leaf_contains(sc, p, x, y, z, mesh_stack); have = true; }
            else if (sp >= CSG_STACK_MAX) { result = false; have = true; }
            else {
                double qx, qy, qz;
                xform_point(p.to_local, x, y, z, qx, qy, qz);
                pts[3 * sp + 3] = qx; pts[3 * sp + 4] = qy; pts[3 * sp + 5] = qz;
                frames[sp] = cur << 1;
                sp += 1;
                cur = p.child_a;
            }
        } else if (sp == 0) {
            finished = true;
        } else {
            const int32_t fr = frames[sp - 1];
            const rsx_primitive &p = sc.prims[fr >> 1];
            if (!(fr & 1)) {                    // operand a is known: a || b, a && b, a && !b
                const bool need_b = p.type == RSX_PRIM_UNION ? !result : result;
                if (need_b) { frames[sp - 1] = fr | 1; cur = p.child_b; have = false; }
                else sp -= 1;                   // (a is the node's answer: true for a Union, false for the others)
            } else {
                if (p.type == RSX_PRIM_SUBTRACT) result = !result;
                sp -= 1;
            }
        }
    }
    return result;
}

// Rebuild the Intersection a CSG node returns for a root: leaf geometry in the leaf's space, lifted operand by operand into
// the top node's space (csg.pyx:198-208), Subtract's swap/negate applied by parity (it commutes with the affine lifts).
__device__ void csg_geom(const DScene &sc, const Ray &r, const Hit &h, Geom &g) {
    int32_t chain[CSG_STACK_MAX + 2];
    int n = 0;
    for (int32_t i = h.leaf; i != h.prim && n < CSG_STACK_MAX + 1; i = sc.csg[i].parent) chain[n++] = i;
    Ray l = to_local(sc.prims[h.prim], r);
    for (int k = n - 1; k >= 0; --k) l = to_local(sc.prims[chain[k]], l);
    const rsx_primitive &leaf = sc.prims[h.leaf];
    if (leaf.type == RSX_PRIM_MESH) {
        const DMesh &m = sc.meshes[leaf.mesh];
        Ray at = l;                                     // mesh_geom recomputes hit = o + d*t; feed the stored hit point instead
        at.ox = h.hx; at.oy = h.hy; at.oz = h.hz;
        mesh_geom(m, at, 0.0, h.a0, h.u, h.v, h.w, g);
        g.hit[0] = h.hx; g.hit[1] = h.hy; g.hit[2] = h.hz;
        const float4 q2 = m.tris[3 * (size_t)h.a0 + 2];
        const double fx = (double)q2.y, fy = (double)q2.z, fz = (double)q2.w;
        g.inside[0] = h.hx - fx * MESH_EPS; g.inside[1] = h.hy - fy * MESH_EPS; g.inside[2] = h.hz - fz * MESH_EPS;
        g.outside[0] = h.hx + fx * MESH_EPS; g.outside[1] = h.hy + fy * MESH_EPS; g.outside[2] = h.hz + fz * MESH_EPS;
    } else {
        analytic_geom(leaf, l, h.t, h.a0, h.a1, g);
    }
    for (int k = 0; k < n; ++k) {
        const rsx_primitive &c = sc.prims[chain[k]];
        double x, y, z;
        xform_point(c.to_root, g.hit[0], g.hit[1], g.hit[2], x, y, z); g.hit[0] = x; g.hit[1] = y; g.hit[2] = z;
        xform_point(c.to_root, g.inside[0], g.inside[1], g.inside[2], x, y, z); g.inside[0] = x; g.inside[1] = y; g.inside[2] = z;
        xform_point(c.to_root, g.outside[0], g.outside[1], g.outside[2], x, y, z); g.outside[0] = x; g.outside[1] = y; g.outside[2] = z;
        const double *mi = c.to_local;                  // Normal3D.transform(to_root) = multiply by inverse transpose
        x = mi[0] * g.normal[0] + mi[4] * g.normal[1] + mi[8] * g.normal[2];
        y = mi[1] * g.normal[0] + mi[5] * g.normal[1] + mi[9] * g.normal[2];
        z = mi[2] * g.normal[0] + mi[6] * g.normal[1] + mi[10] * g.normal[2];
        g.normal[0] = x; g.normal[1] = y; g.normal[2] = z;
    }
    if (h.flags & F_FLIP) {
        for (int k = 0; k < 3; ++k) { const double tmp = g.inside[k]; g.inside[k] = g.outside[k]; g.outside[k] = tmp; g.normal[k] = -g.normal[k]; }
    }
    g.exiting = (h.flags & F_EXIT) != 0;
}

