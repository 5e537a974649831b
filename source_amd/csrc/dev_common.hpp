// dev_common.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// Scene structs, ray / hit records, traversal stack, box and KD-step primitives shared by every device routine.
#pragma once

// ---------------------------------------------------------------------------------------------------
// device-resident scene
// ---------------------------------------------------------------------------------------------------
struct DMesh {
    const rsx_kdnode *nodes;
    const int32_t *items;
    const float4 *tris;        // 3 x float4 per triangle: v1.xyz v2.x | v2.yz v3.xy | v3.z fn.xyz
    const float4 *leaf;        // 4 x float4 per LEAF ITEM, in items[] order: the 3 above + (triangle id, -, -, -): one 64-byte line per
                               // test, no id->record indirection, big leaves stream as contiguous memory
    const float *vnormals;     // [nn,3] or null
    const int32_t *nidx;       // [nt,3] vertex-normal indices or null
    double lower[3], upper[3];
    int32_t smoothing, closed, n_tris, splits_bounded;     // splits_bounded: every split of `nodes` lies inside [lower, upper]
};

struct DScene {
    const rsx_primitive *prims;
    const DMesh *meshes;
    const rsx_kdnode *wnodes;
    const int32_t *witems;
    double wlower[3], wupper[3];
    int32_t n_prims, n_world, n_meshes;
    int32_t wdepth, mdepth;    // stack levels a traversal of the world tree / the deepest mesh tree can need
    int32_t wlds, mlds;        // how many of those levels are held in LDS (the rest spill)
    char *spill;               // per-wave global spill regions
    const struct CsgInfo *csg; // per primitive: parent CSG node, per-lane state slot, operand side (null without CSG)
    const struct CsgFast *csgfast;   // per primitive: flattened operand tree of a top-level CSG node whose leaves are all analytic (n_leaves = 0: none)
    const struct CsgFast *csgfast_uniform;   // the same table, never redirected to a kernel's LDS copy: what the scalar data path reads (csg_fast_hit_uniform)
    int32_t n_wnodes, n_witems;   // sizes of the world tree (the path kernel stages a small one in LDS)
    int32_t wide[8];           // analytic world primitives that sit in several world leaves (floors, walls, enclosing emitters), most leaves
                               // first, -1 = none: world_trace_wave computes their first root once per ray instead of once per leaf
                               // visited — the first two for primary rays, all eight for scattered rays (see there)
    int32_t wide_csg[4];       // top-level CSG primitives (state-free evaluator form) that sit in several world leaves, -1 = none: the path
                               // kernel of a CSG scene answers them for the whole wave before the traversal (world_trace_wave's mailbox)
    const rsx_primitive *prims_uniform;   // = prims in global memory: what the scalar data path reads (the path kernel may point `prims` at an LDS copy)
    struct NodeSt *csg_arena;             // node states of the CSG stream merge for trees too big for the kernels' private arrays, or null:
    int32_t csg_arena_slots, csg_arena_lanes;   // [csg_arena_lanes][csg_arena_slots] (dev_csg.hpp: csg_slots)
    int32_t wide_plain;        // bit j: wide[j] is a box without any transform (a floor, an enclosing emitter): the packet walk answers it in a form
                               // specialised for rays that agree in the signs of their direction (dev_packet.hpp: plain_box_first_root)
    int32_t wsplits_bounded, csg_fast_rows;   // every split of the world tree lies inside [wlower, wupper] (packet_space, dev_packet.hpp);
                                          // LDS rows the state-free CSG evaluator needs: two per leaf of the scene's biggest flattened tree (0: none)
    const float4 *rel;                    // camera-relative leaf records of the mesh instances (dev_packet.hpp: RelInfo), or null
    const struct RelInfo *rel_info;       // per primitive
    // The packet walk's short cut (dev_packet.hpp: world_trace_packet): up to four boxes that together hold the bounding box of every world
    // primitive the packet kernel does not answer before its walk (everything but wide[0] and wide[1]). A unit none of whose rays enters any of
    // them meets those two answers only. pkt_clusters: number of boxes, -1 = no short cut (coordinates too large for its margin argument).
    int32_t pkt_clusters;
    int32_t all_wide8;         // 1: every world primitive is one of wide[0 .. 7] (and the coordinates are tame): world_trace_wave's eight-slot form has answered them all before its walk
    int32_t all_answered_csg;  // 1: a CSG scene all of whose world primitives are answered before the per-lane walk of its fast forms — the analytic ones in
    int32_t scene_pad;         //    wide[0 .. RSX_CSG_WIDE - 1], the solids in wide_csg[0 .. 3] (and the coordinates are tame): world_trace_wave
    double cluster_lo[4][3], cluster_hi[4][3];
    // ... and, for a cluster of at most four primitives, their own bounding boxes: a unit that enters the cluster's box is asked the
    // BoundPrimitive gates themselves (0 members: the cluster's box decides)
    int32_t cluster_members[4];
    double member_lo[4][4][3], member_hi[4][4][3];
    const rsx_kdnode *wnodes_scatter;   // the world nodes annotated for the kernels of scattered rays: leaf tags and cull bits for the eight-slot set, or — CSG
                                        // scenes — two slots plus the CSG primitives answered before the traversal (wnodes: for the first two slots)
};

struct CsgInfo {
    int32_t parent, slot, is_b, top;
};

// A top-level CSG primitive whose operand tree has only sphere / box / cylinder leaves, flattened on the host (rsx_scene_create):
// leaves in depth-first order with the chain of nodes that leads to each, the parity of Subtract-b ancestors (how often
// _modify_intersection flips the root, csg.pyx:551-568), and the tree as a postfix program over "inside leaf k" bits.
#define CSGF_MAX_LEAVES 8
#define CSGF_MAX_CHAIN 6
struct CsgFast {
    int32_t n_leaves, n_ops;
    int32_t top_type, top_a_leaves;                   // the top node's operator and how many leaves its operand a holds (they come first)
    int32_t leaf[CSGF_MAX_LEAVES];
    int32_t parity[CSGF_MAX_LEAVES];
    int32_t chain_len[CSGF_MAX_LEAVES];
    int32_t chain[CSGF_MAX_LEAVES][CSGF_MAX_CHAIN];   // primitive ids from the top node's operand down to the leaf (inclusive)
    int8_t ops[2 * CSGF_MAX_LEAVES];                  // >= 0: push inside(leaf slot); -1 union, -2 intersect, -3 subtract
    uint64_t truth[4];                                // the program's value for every combination of "inside leaf k" bits (bit = the mask)
    int8_t guard_lo[CSGF_MAX_LEAVES], guard_hi[CSGF_MAX_LEAVES];   // leaf k lies in operand b of an Intersect / Subtract whose operand a is the leaves
                                                      // [guard_lo, guard_hi): no root among those and k is not looked at (hi = 0: no such node)
};
__device__ __forceinline__ uint32_t csg_truth(const CsgFast &P, uint32_t inside_bits) { return (uint32_t)(P.truth[inside_bits >> 6] >> (inside_bits & 63u)) & 1u; }

struct Ray {
    double ox, oy, oz, dx, dy, dz, maxd;
};

// candidate kept while searching; full geometry is regenerated once at the end (finalise)
struct Hit {
    double t;
    int32_t prim;              // -1 = none
    int32_t a0, a1;            // mesh: triangle, - ; box: face, axis ; cylinder: face, type
    float u, v, w;
    // CSG hits only: the operand leaf that produced the root, Subtract flip parity / exiting, mesh-leaf hit point
    int32_t leaf;
    uint32_t flags;
    double hx, hy, hz;
};

// Per-lane traversal stack: the first `lds_levels` entries live in LDS ([level][lane], conflict free), deeper ones spill to a
// per-wave global buffer with the same layout. 99 % of camera rays on the 69k-triangle mesh never have more than 9 far nodes
// pending (oracle histogram), so the spill path is cold but keeps the traversal exact for any depth.
struct Stack {
    // Everything here is wave-uniform (lives in SGPRs); the lane's own slot is addressed as base + (level * WAVE + lane) * size
    // at each access, so the stack costs the traversal loop no per-lane pointer registers.
    uint32_t lds_t, lds_id;    // byte offsets in the workgroup's dynamic LDS: t[level][lane] (f64), id[level][lane] (i32)
    char *gt, *gid;            // spill arrays with the same layout, levels >= lds_levels
    int32_t lds_levels;
    float4 *stage;             // per-wave LDS staging area: WAVE triangle records (3 x float4) + WAVE triangle ids
};

extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ void stack_push(const Stack &st, int32_t sp, int32_t id, double t) {
    const int slot = sp * WAVE + (int)(threadIdx.x % WAVE);
    // the global spill levels are cold: one wave-level test keeps them out of the common path
    if (__builtin_expect(__any(sp >= st.lds_levels), 0)) {
        if (sp >= st.lds_levels) {
            const int g = slot - st.lds_levels * WAVE;
            reinterpret_cast<double *>(st.gt)[g] = t;
            reinterpret_cast<int32_t *>(st.gid)[g] = id;
            return;
        }
    }
    *reinterpret_cast<double *>(smem + st.lds_t + slot * 8) = t;
    *reinterpret_cast<int32_t *>(smem + st.lds_id + slot * 4) = id;
}

__device__ __forceinline__ void stack_pop(const Stack &st, int32_t sp, int32_t &id, double &t) {
    const int slot = sp * WAVE + (int)(threadIdx.x % WAVE);
    if (__builtin_expect(__any(sp >= st.lds_levels), 0)) {
        if (sp >= st.lds_levels) {
            const int g = slot - st.lds_levels * WAVE;
            t = reinterpret_cast<const double *>(st.gt)[g];
            id = reinterpret_cast<const int32_t *>(st.gid)[g];
            return;
        }
    }
    t = *reinterpret_cast<const double *>(smem + st.lds_t + slot * 8);
    id = *reinterpret_cast<const int32_t *>(smem + st.lds_id + slot * 4);
}

__device__ __forceinline__ double sel3(int i, double x, double y, double z) { return i == 0 ? x : (i == 1 ? y : z); }
__device__ __forceinline__ float sel3f(int i, float x, float y, float z) { return i == 0 ? x : (i == 1 ? y : z); }

// Point3D.transform / Vector3D.transform — core/math/point.pyx:253-284, vector.pyx:339-369
__device__ __forceinline__ void xform_point(const double *m, double x, double y, double z, double &ox, double &oy, double &oz) {
    double w = m[12] * x + m[13] * y + m[14] * z + m[15];
    w = 1.0 / w;
    ox = (m[0] * x + m[1] * y + m[2] * z + m[3]) * w;
    oy = (m[4] * x + m[5] * y + m[6] * z + m[7]) * w;
    oz = (m[8] * x + m[9] * y + m[10] * z + m[11]) * w;
}

__device__ __forceinline__ void xform_vector(const double *m, double x, double y, double z, double &ox, double &oy, double &oz) {
    ox = m[0] * x + m[1] * y + m[2] * z;
    oy = m[4] * x + m[5] * y + m[6] * z;
    oz = m[8] * x + m[9] * y + m[10] * z;
}

// Scene tables (primitives, mesh descriptors) never change while a kernel runs. Read through the constant address space with a
// wave-uniform index they come in over the scalar data path into SGPRs: no vector registers for a 4x4 matrix or a mesh descriptor.
#define RSX_CONST_AS __attribute__((address_space(4)))
typedef const RSX_CONST_AS rsx_primitive *UPrim;
typedef const RSX_CONST_AS struct DMesh *UMesh;
__device__ __forceinline__ UPrim uniform_prim(const rsx_primitive *base, int32_t idx) { return (UPrim)(unsigned long long)(base + idx); }

__device__ __forceinline__ Ray to_local_uniform(UPrim p, const Ray &r) {
    const RSX_CONST_AS double *m = p->to_local;
    Ray l;
    // Point3D.transform divides by the homogeneous w (point.pyx:253-284). For an affine matrix (last row 0 0 0 1 — every matrix
    // translate/rotate produce) w is exactly 1 and x * (1.0 / 1.0) == x bit for bit, so the wave-uniform test skips a division.
    double w = 1.0;
    const bool affine = m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0;
    if (!affine) { w = m[12] * r.ox + m[13] * r.oy + m[14] * r.oz + m[15]; w = 1.0 / w; }
    l.ox = (m[0] * r.ox + m[1] * r.oy + m[2] * r.oz + m[3]) * w;
    l.oy = (m[4] * r.ox + m[5] * r.oy + m[6] * r.oz + m[7]) * w;
    l.oz = (m[8] * r.ox + m[9] * r.oy + m[10] * r.oz + m[11]) * w;
    l.dx = m[0] * r.dx + m[1] * r.dy + m[2] * r.dz;
    l.dy = m[4] * r.dx + m[5] * r.dy + m[6] * r.dz;
    l.dz = m[8] * r.dx + m[9] * r.dy + m[10] * r.dz;
    l.maxd = r.maxd;
    return l;
}

__device__ __forceinline__ Ray to_local(const rsx_primitive &p, const Ray &r) {
    Ray l;
    xform_point(p.to_local, r.ox, r.oy, r.oz, l.ox, l.oy, l.oz);
    xform_vector(p.to_local, r.dx, r.dy, r.dz, l.dx, l.dy, l.dz);
    l.maxd = r.maxd;
    return l;
}

// BoundingBox3D._slab / intersect — core/boundingbox.pyx:180-245
__device__ __forceinline__ void slab(double o, double d, double lo, double hi, double &front, double &back) {
    double tmin, tmax;
    const double inf = INFINITY;
    if (d != 0.0) {
        const double rcp = 1.0 / d;
        if (d > 0) { tmin = (lo - o) * rcp; tmax = (hi - o) * rcp; }
        else       { tmin = (hi - o) * rcp; tmax = (lo - o) * rcp; }
    } else {
        if (o < lo)      { tmin = -inf; tmax = -inf; }
        else if (o > hi) { tmin = inf;  tmax = inf; }
        else             { tmin = -inf; tmax = inf; }
    }
    if (tmin > front) front = tmin;
    if (tmax < back) back = tmax;
}

__device__ __forceinline__ bool aabb(const double *lo, const double *hi, const Ray &r, double &front, double &back) {
    front = -INFINITY;
    back = INFINITY;
    slab(r.ox, r.dx, lo[0], hi[0], front, back);
    slab(r.oy, r.dy, lo[1], hi[1], front, back);
    slab(r.oz, r.dz, lo[2], hi[2], front, back);
    if (front > back) return false;
    if (front < 0.0 && back < 0.0) return false;
    return true;
}

// Same test with the three reciprocals 1.0/d hoisted by the caller (bit-identical: the reference recomputes the same
// quotient for every box it tests a ray against).
__device__ __forceinline__ void slab_rcp(double o, double d, double rcp, double lo, double hi, double &front, double &back) {
    double tmin, tmax;
    const double inf = INFINITY;
    if (d != 0.0) {
        if (d > 0) { tmin = (lo - o) * rcp; tmax = (hi - o) * rcp; }
        else       { tmin = (hi - o) * rcp; tmax = (lo - o) * rcp; }
    } else {
        if (o < lo)      { tmin = -inf; tmax = -inf; }
        else if (o > hi) { tmin = inf;  tmax = inf; }
        else             { tmin = -inf; tmax = inf; }
    }
    if (tmin > front) front = tmin;
    if (tmax < back) back = tmax;
}

__device__ __forceinline__ bool aabb_rcp(const double *lo, const double *hi, const Ray &r, double rx, double ry, double rz, double &front, double &back) {
    front = -INFINITY;
    back = INFINITY;
    slab_rcp(r.ox, r.dx, rx, lo[0], hi[0], front, back);
    slab_rcp(r.oy, r.dy, ry, lo[1], hi[1], front, back);
    slab_rcp(r.oz, r.dz, rz, lo[2], hi[2], front, back);
    if (front > back) return false;
    if (front < 0.0 && back < 0.0) return false;
    return true;
}

__device__ __forceinline__ bool aabb_contains(const double *lo, const double *hi, double x, double y, double z) {
    if (x < lo[0] || x > hi[0]) return false;
    if (y < lo[1] || y > hi[1]) return false;
    if (z < lo[2] || z > hi[2]) return false;
    return true;
}

__device__ __forceinline__ rsx_kdnode load_node(const rsx_kdnode *nodes, int32_t id) {
    const int4 raw = *reinterpret_cast<const int4 *>(nodes + id);   // one 16-B load
    rsx_kdnode nd;
    nd.type = raw.x;
    nd.count = raw.y;
    nd.u.leaf.first_item = raw.z;
    nd.u.leaf.pad = raw.w;
    return nd;
}

// Correctly rounded n / d with the d-only part of the division hoisted out of the traversal loop.
// hipcc expands an IEEE f64 division into v_div_scale, v_rcp_f64, two Newton steps on the reciprocal, q0 = n*y,
// r = fma(-d, q0, n), v_div_fmas (= fma(r, y, q0) when no scaling is in effect) and v_div_fixup. The reciprocal refinement
// depends on d alone, and a ray divides by the same three direction components at every KD node, so it is computed once per ray
// space (refine_rcp) and the per-node work shrinks to mul + 2 fma. The shortcut is taken only when neither operand is anywhere
// near the exponent ranges where v_div_scale / v_div_fixup intervene; otherwise the plain division runs. tests/test_gpu_parity.py
// (test_exact_division) checks bit equality against `/` on the device over 2^28 operand pairs including exact and near-tie cases.
#ifndef RSX_FAST_DIV
#define RSX_FAST_DIV 1
#endif

__device__ __forceinline__ double refine_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double f0 = __builtin_fma(-d, r, 1.0);
    const double y1 = __builtin_fma(r, f0, r);
    const double f2 = __builtin_fma(-d, y1, 1.0);
    return __builtin_fma(y1, f2, y1);
}

__device__ __forceinline__ bool div_operand_safe(double x) {          // |x| in [2^-300, 2^300] (false for NaN): two compares
    const double a = __builtin_fabs(x);
    return a >= 0x1p-300 && a <= 0x1p+300;
}

__device__ __forceinline__ double exact_div(double n, double d, double y, bool d_safe) {
#if RSX_FAST_DIV
    if (d_safe && div_operand_safe(n)) {
        const double q0 = n * y;
        const double r = __builtin_fma(-d, q0, n);
        return __builtin_fma(r, y, q0);
    }
    if (d_safe && n == 0.0) return n * y;          // signed zero with the quotient's sign (the correction step would lose it)
#endif
    return n / d;
}

struct AxisDiv {               // per ray space: refined reciprocals of the three direction components
    double yx, yy, yz;
    int safe;                  // bit k: component k may take the shortcut
};

__device__ __forceinline__ AxisDiv axis_div(const Ray &r) {
    AxisDiv a;
    a.yx = refine_rcp(r.dx); a.yy = refine_rcp(r.dy); a.yz = refine_rcp(r.dz);
    a.safe = (div_operand_safe(r.dx) ? 1 : 0) | (div_operand_safe(r.dy) ? 2 : 0) | (div_operand_safe(r.dz) ? 4 : 0);
    return a;
}

// One KD branch step — KDTree3DCore._trace_branch, core/math/spatial/kdtree3d.pyx:626-700.
// Returns the next node; pushes (far, tmax) when both children are crossed.
__device__ __forceinline__ int32_t branch_step(const rsx_kdnode &nd, int32_t node, double o, double d, double y, bool d_safe, double tmin,
                                               double &tmax, const Stack &st, int32_t &sp) {
    const double split = nd.u.split;
    const int32_t lower = node + 1, upper = nd.count;
    // the two rare cases — a ray parallel to the plane, operands outside the range of the hoisted-reciprocal quotient — sit behind
    // wave-level tests: a scalar jump over them in the common case instead of an exec-mask save / restore around each
    if (__builtin_expect(__any(d == 0), 0)) {
        if (d == 0) return o < split ? lower : upper;
    }
    const double num = split - o;
#if RSX_FAST_DIV
    const double q0 = num * y;
    const double rem = __builtin_fma(-d, q0, num);
    double plane = __builtin_fma(rem, y, q0);
    const bool zero = num == 0.0;
    plane = zero ? q0 : plane;                          // 0 * y: signed zero with the quotient's sign
    const bool exact = d_safe & (div_operand_safe(num) | zero);
    if (__builtin_expect(__any(!exact), 0)) {
        if (!exact) plane = num / d;
    }
#else
    const double plane = num / d;
#endif
    const bool below = o < split || (o == split && d < 0);
    const int32_t near_id = below ? lower : upper, far_id = below ? upper : lower;
    if (plane > tmax || plane <= 0) return near_id;
    if (plane < tmin) return far_id;
    stack_push(st, sp, far_id, tmax);
    ++sp;
    tmax = plane;
    return near_id;
}

// Walk from `node` down to a leaf. Nodes are loaded as (node, node+1) pairs: the lower child is always the next record of the
// pre-order array, so stepping into it costs no dependent load (its own successor is fetched in the shadow of the step's arithmetic).
__device__ __forceinline__ int32_t world_step(double split, int32_t lower, int32_t upper, double o, double d, double &tmin, double &tmax, const Stack &st, int32_t &sp,
                                              int32_t cull_bits = 0, double t_cull = 0.0);

// PLAIN: the branch steps use the hardware division outright (world tree: no refined reciprocals are kept for it, and branch_step's
// shortcut would be computed only to be discarded at every step)
template <bool PLAIN = false>
__device__ __forceinline__ rsx_kdnode descend(const rsx_kdnode *nodes, int32_t &node, const Ray &r, const AxisDiv &ad, double tmin, double &tmax,
                                              const Stack &st, int32_t &sp, unsigned long long *util = nullptr, double t_cull = 0.0) {
    rsx_kdnode nd = load_node(nodes, node), nx = load_node(nodes, node + 1);
    while (nd.type >= 0) {
        UTIL_COUNT(util, 4)
        const int axis = nd.type & 3;                   // (world nodes: bits 2, 3 = the cull bits of world_step)
        int32_t next;
        if constexpr (PLAIN) next = world_step(nd.u.split, node + 1, nd.count, sel3(axis, r.ox, r.oy, r.oz), sel3(axis, r.dx, r.dy, r.dz), tmin, tmax, st, sp,
                                               nd.type >> 2, t_cull);
        else next = branch_step(nd, node, sel3(axis, r.ox, r.oy, r.oz), sel3(axis, r.dx, r.dy, r.dz), sel3(axis, ad.yx, ad.yy, ad.yz),
                                         (ad.safe >> axis) & 1, tmin, tmax, st, sp);
        if (next == node + 1) nd = nx; else nd = load_node(nodes, next);
        nx = load_node(nodes, next + 1);
        node = next;
    }
    return nd;
}


// One branch step of the world tree (kdtree3d.pyx:626-700: plane = (split - origin[axis]) / direction[axis], the plain division).
// cull_bits (bit 0 / 1: every item below the lower / upper child is a wide primitive, tagged in the device copy of the world nodes)
// and t_cull (the nearest of the ray's wide answers): a near child whose leaves all end before t_cull cannot accept anything — a
// leaf accepts t <= min(max_distance, its tmax) <= plane < t_cull <= every candidate — so the walk goes straight to the far child,
// which begins at the plane exactly as when it is popped after the near side.
__device__ __forceinline__ int32_t world_step(double split, int32_t lower, int32_t upper, double o, double d, double &tmin, double &tmax, const Stack &st, int32_t &sp,
                                              int32_t cull_bits, double t_cull) {
    if (__builtin_expect(__any(d == 0), 0)) {
        if (d == 0) return o < split ? lower : upper;
    }
    const double plane = (split - o) / d;
    const bool below = o < split || (o == split && d < 0);
    const int32_t near_id = below ? lower : upper, far_id = below ? upper : lower;
    if (plane > tmax || plane <= 0) return near_id;
    if (plane < tmin) return far_id;
    if (((below ? cull_bits : cull_bits >> 1) & 1) && plane < t_cull) { tmin = plane; return far_id; }
    stack_push(st, sp, far_id, tmax);
    ++sp;
    tmax = plane;
    return near_id;
}

