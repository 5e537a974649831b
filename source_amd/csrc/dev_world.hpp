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
            csg_first(e, idx, r, rec);
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
// Primitive.hit of a wave-uniform sphere / box / cylinder (BoundPrimitive gate included): first root t >= 0 and its (face, axis | type)
// packed as (a0 + 1) | (a1 + 1) << 4, or t = -1 when the ray misses. A box under a translate-only transform divides by the world ray's
// own direction components: l.d == r.d wherever a component is non-zero (1 * x + 0 * y + 0 * z), so the gates' 1.0 / d are the box's own.
__device__ __forceinline__ void analytic_first_root(const rsx_primitive *prims, UPrim up, int32_t uidx, bool want, const Ray &r, double rx, double ry, double rz,
                                                    double &t, int32_t &faces) {
    t = -1.0; faces = 0;
    const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
    double f, b;
    const bool gate = want && aabb_rcp(lo, hi, r, rx, ry, rz, f, b);            // BoundPrimitive.hit gate (boundprimitive.pyx:42-51)
#if RSX_UTIL_PROF == 2
    (void)f;
#endif
    if (!gate) return;
    const int32_t type = up->type;
    const Ray l = to_local_uniform(up, r);
    Roots roots;
    roots.n = 0;
    if (type == RSX_PRIM_BOX) {
        const RSX_CONST_AS double *m = up->to_local;
        const bool identity = m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[5] == 1.0 && m[6] == 0.0 &&
                              m[8] == 0.0 && m[9] == 0.0 && m[10] == 1.0;
        const double prm[6] = {up->params[0], up->params[1], up->params[2], up->params[3], up->params[4], up->params[5]};
        box_roots_uniform(prm, l, identity, rx, ry, rz, roots);
    } else if (type == RSX_PRIM_SPHERE) sphere_roots_uniform(up->params[0], l, roots);
    else if (type == RSX_PRIM_CYLINDER) cylinder_roots(prims[uidx], l, roots);
    if (roots.n > 0) { t = roots.t[0]; faces = (roots.a0[0] + 1) | ((roots.a1[0] + 1) << 4); }
}

// UNIFORM_ITEMS: the leaf items are walked one wave-uniform primitive at a time (primary rays: the lanes of a wave sit in the same
// world leaf, so a leaf item is one primitive for all of them and its record comes in over the scalar data path). Scattered rays —
// the daughters of the path kernel, arbitrary query batches — meet a different primitive in every lane: there each lane tests its
// own item and only the meshes are grouped (measured with the uniform walk on the path kernel: Cornell box 59 -> 76 ms per pass).
// Wide primitives (DScene::wide), WIDE_N of them answered before the traversal: two where registers are short (the primary-ray
// kernel and the query batches at three waves per SIMD, every CSG instantiation — with eight the CSG path kernels spill and the
// prism slice goes 54 -> 76 ms); eight in the path kernel of scenes without CSG — its per-lane leaf items run at a fraction of the
// wave's width (a Cornell-box round spent 60 % of its world time there), while eight wave-uniform tests up front run at full width
// over scalar-loaded records: Cornell box 57.7 -> 51.8 ms per pass.
struct WideSet8 {
    double t[8];
    uint32_t faces[2];         // 8 bits per slot
};
template <int N>
__device__ __forceinline__ void wide_lookup(const WideSet8 &w, int slot, double &t, int32_t &faces) {
    t = w.t[0]; uint32_t f = w.faces[0];
#pragma unroll
    for (int j = 1; j < N; ++j) if (slot == j) { t = w.t[j]; f = j < 4 ? w.faces[0] >> (8 * j) : w.faces[1] >> (8 * (j - 4)); }
    faces = (int32_t)(f & 255u);
}
template <int N>
__device__ __forceinline__ int wide_slot(const DScene &sc, int32_t idx) {     // slot of primitive idx in the first N wide entries, or -1
    int slot = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) if (idx == sc.wide[j]) slot = j;
    return slot;
}

#ifndef RSX_WIDE_ALL_SHORTCUT
#define RSX_WIDE_ALL_SHORTCUT 1      // worlds whose every primitive is answered before the walk (eight slots) skip the walk: world_trace_wave
#endif
#ifndef RSX_WORLD_CULL
#define RSX_WORLD_CULL 1
#endif
#ifndef RSX_WIDE_PER_LANE
#define RSX_WIDE_PER_LANE 1           // eight-slot kernels: the box slots' roots per lane, after wave-uniform gates (world_trace_wave)
#endif
#ifndef RSX_CSG_MAILBOX
#define RSX_CSG_MAILBOX 4
#endif
#ifndef RSX_CSG_WIDE
#define RSX_CSG_WIDE 4                // analytic wide slots of the CSG kernels that answer the wide CSG solids before the traversal (2 or 4)
#endif
template <bool CSG, bool FASTONLY = false, int STAGE_MIN = RSX_STAGE_MIN, bool UNIFORM_ITEMS = false, int WIDE_N = 2, int CSG_MAILBOX = RSX_CSG_MAILBOX,
          bool MESHES = true>
__device__ __forceinline__ bool world_trace_wave(bool valid, const DScene &sc, const Ray &r, const Stack &st, const Stack &mesh_stack, NodeSt *csg_state, Hit &best,
                                 uint32_t &work, unsigned long long *phase_acc = nullptr) {
    best.prim = -1;
    double tmin = 0, tmax = 0;
    const double rx = 1.0 / r.dx, ry = 1.0 / r.dy, rz = 1.0 / r.dz;
    bool active = valid && aabb_rcp(sc.wlower, sc.wupper, r, rx, ry, rz, tmin, tmax);
    // The world tree's branch steps use the plain division. Feeding exact_div with the gates' own 1.0 / d (the correctly rounded
    // reciprocal satisfies the same correction step as the refined one: rsx_selftest_exact_division checks both forms against `/`)
    // was measured: 41.1 -> 43.5 ms on configs[2] — the hardware division is ~11 instructions, the shortcut with its operand-range
    // tests is no shorter and costs registers.
    AxisDiv ad;
    ad.yx = ad.yy = ad.yz = 0.0; ad.safe = 0;
    // Primitives that sit in several world leaves (the reference tests a primitive again in every leaf the ray visits,
    // kdtree.pyx:99-116; the answer is the same each time): their first root is computed once, here. configs[2]: the floor box and
    // the enclosing emitter are met 4.4 times per primary ray.
    static_assert(WIDE_N == 2 || WIDE_N == 8 || (CSG && WIDE_N == RSX_CSG_WIDE), "two tagged copies of the world nodes exist: for two slots and for eight (CSG scenes: RSX_CSG_WIDE)");
    // (CSG kernels with the mailbox: the copy whose cull bits count the CSG primitives answered in the round before the traversal)
    constexpr bool CSG_ANSWERED = CSG && FASTONLY && CSG_MAILBOX >= 4 && !UNIFORM_ITEMS;
    static_assert(!CSG_ANSWERED || WIDE_N == RSX_CSG_WIDE, "the CSG-answered copy of the world nodes is tagged for RSX_CSG_WIDE slots");
    static_assert(!CSG || CSG_ANSWERED || WIDE_N == 2, "the other kernels of a CSG scene walk the two-slot copy");
    const rsx_kdnode *wnodes = (WIDE_N == 8 || CSG_ANSWERED) ? sc.wnodes_scatter : sc.wnodes;
    WideSet8 wide;
#pragma unroll
    for (int j = 0; j < 8; ++j) wide.t[j] = -1.0;
    wide.faces[0] = wide.faces[1] = 0;
#if RSX_PHASE_PROF == 2
    const unsigned long long ph2_wd0 = clock64();
#endif
    // Scattered rays (eight slots): a ray inside a room passes the BoundPrimitive gate of two or three of its walls, yet a wave-uniform
    // test per slot finds the roots of all eight boxes for as long as ANY lane passes each gate — a fifth of the lanes on each.
    // So the gates of the box slots run wave-uniform (scalar-loaded boxes, full width) and leave each lane a mask of the boxes it
    // has to answer; the roots are then found round by round, every lane on ITS next box (its record read per lane). Same
    // arithmetic per (ray, box) as analytic_first_root: the gate, Point3D / Vector3D.transform, Box.hit's slabs.
    uint32_t lane_boxes = 0;
#if RSX_PHASE_PROF == 3
    const unsigned long long ph3_w0 = clock64();
#endif
#pragma unroll
    for (int j = 0; j < WIDE_N; ++j) {
        if (sc.wide[j] >= 0) {                              // (wave-uniform)
            const UPrim up = uniform_prim(sc.prims_uniform, sc.wide[j]);
            if (WIDE_N == 8 && RSX_WIDE_PER_LANE && up->type == RSX_PRIM_BOX) {
                const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
                double f, b;
                if (active && aabb_rcp(lo, hi, r, rx, ry, rz, f, b)) lane_boxes |= 1u << j;
                continue;
            }
            int32_t f = 0;
            analytic_first_root(sc.prims, up, sc.wide[j], active, r, rx, ry, rz, wide.t[j], f);
            wide.faces[j >> 2] |= (uint32_t)f << (8 * (j & 3));
        }
    }
#if RSX_PHASE_PROF == 3
    const unsigned long long ph3_w1 = clock64();
    if (phase_acc) phase_acc[9] += ph3_w1 - ph3_w0;
#endif
    if constexpr (WIDE_N == 8 && RSX_WIDE_PER_LANE) {
        while (__any(lane_boxes != 0u)) {
#if RSX_PHASE_PROF == 3
            if (phase_acc) { phase_acc[11] += 1; phase_acc[12] += __popcll(__ballot(lane_boxes != 0u)); }
#endif
            const bool have = lane_boxes != 0u;
            const int slot = have ? __builtin_ctz(lane_boxes) : 0;
            lane_boxes &= lane_boxes - 1u;
            int32_t idx = sc.wide[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) if (slot == q) idx = sc.wide[q];
            if (have) {
                const rsx_primitive &p = sc.prims[idx];
                const double *m = p.to_local;
                Ray l;
                double w = 1.0;                                               // (to_local_uniform's shortcut: an affine matrix has w == 1 exactly)
                if (!(m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0)) { w = m[12] * r.ox + m[13] * r.oy + m[14] * r.oz + m[15]; w = 1.0 / w; }
                l.ox = (m[0] * r.ox + m[1] * r.oy + m[2] * r.oz + m[3]) * w;
                l.oy = (m[4] * r.ox + m[5] * r.oy + m[6] * r.oz + m[7]) * w;
                l.oz = (m[8] * r.ox + m[9] * r.oy + m[10] * r.oz + m[11]) * w;
                l.dx = m[0] * r.dx + m[1] * r.dy + m[2] * r.dz;
                l.dy = m[4] * r.dx + m[5] * r.dy + m[6] * r.dz;
                l.dz = m[8] * r.dx + m[9] * r.dy + m[10] * r.dz;
                l.maxd = r.maxd;
                Roots roots;
                box_roots(p, l, roots);
                if (roots.n > 0) {
                    const uint32_t f = (uint32_t)((roots.a0[0] + 1) | ((roots.a1[0] + 1) << 4));
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (slot == q) wide.t[q] = roots.t[0];
                    if (slot < 4) wide.faces[0] |= f << (8 * slot); else wide.faces[1] |= f << (8 * (slot - 4));
                }
            }
        }
    }
#if RSX_PHASE_PROF == 3
    if (phase_acc) phase_acc[10] += clock64() - ph3_w1;
#endif
    // the nearest of those answers: subtrees that hold wide primitives only and end before it are not entered (world_step)
    double t_cull = INFINITY;
#pragma unroll
    for (int j = 0; j < WIDE_N; ++j) if (wide.t[j] >= 0.0 && wide.t[j] < t_cull) t_cull = wide.t[j];
    // (the prefill round below leaves out a solid whose padded box the ray enters BEYOND that answer: every hit of the solid lies inside its
    // box, hence farther than a hit the traversal is certain to meet — Primitive.hit is not asked, the mailbox records a miss)
    const double t_wide_near = RSX_PREFILL_CULL ? t_cull : INFINITY;
#if RSX_WORLD_CULL == 0
    t_cull = -INFINITY;
#endif
#if RSX_PHASE_PROF == 2
    phase_acc[7] += clock64() - ph2_wd0;
#endif
    constexpr int MB_N = CSG_MAILBOX > 0 ? CSG_MAILBOX : 1;
    int32_t mb_idx[MB_N], mb_leaf[MB_N], mb_next = 0;
    uint32_t mb_meta[MB_N];
    double mb_t[MB_N];
#pragma unroll
    for (int j = 0; j < MB_N; ++j) { mb_idx[j] = -1; mb_leaf[j] = 0; mb_meta[j] = 0; mb_t[j] = -1.0; }
    // CSG primitives that sit in several leaves (DScene::wide_csg) are answered in a round of their own before the traversal, all
    // lanes at once, through the item loop below (one copy of the evaluator): left to the leaves, every lane meets them in a
    // different round and the wave runs the evaluator once per round for a few lanes each.
    bool prefill = false;
    if constexpr (CSG && FASTONLY && CSG_MAILBOX >= 4 && !UNIFORM_ITEMS) prefill = sc.wide_csg[0] >= 0;
    // Coherent waves (UNIFORM_ITEMS): the last CSG primitive the wave evaluated and every lane's answer. A CSG solid that sits in several
    // world leaves is met again leaf after leaf by a ray that went through its box without hitting it (the holes of demos/csg.py's
    // solids); Primitive.hit(ray) does not depend on the leaf, so the answer is kept instead of found again (kdtree.pyx:105-111 calls
    // hit() once per leaf item: same ray, same result).
    int32_t last_csg = -1;                                   // (wave-uniform)
    double last_t = 0.0;
    int32_t last_leaf = 0;
    uint32_t last_meta = 0;                                  // bits 0..7 a0, 8..15 a1, 16..23 flags, 28..29: 0 not asked, 1 hit, 2 no hit, 3 stream merge
#if RSX_WIDE_ALL_SHORTCUT
    // A world all of whose primitives sit in the eight slots (DScene::all_wide8: a Cornell box, a furnace) has been answered completely
    // before the walk; what the walk would add is WHICH answer a ray takes: kdtree.pyx:99-116 accepts answer t_j in the first leaf on the ray
    // that lists j and whose range reaches t_j (`t_j <= min(max_distance, tmax)`, the last leaf's tmax being the world box's `back`), the
    // nearer answer before the farther — the point at t_j lies inside j's bounding box with BOX_PADDING = 1e-9 to spare on every side, so the
    // leaf whose range holds t_j overlaps that box and lists j (positions move by ~1e-12 of rounding at the coordinates this is enabled for,
    // |world bounds| <= 1e4), and a leaf that holds the farther answer but not the nearer lies behind one that held the nearer. So a ray takes
    // its nearest eligible answer; only when the nearest is shared by two primitives (the leaf's item order decides) does it walk.
    if constexpr (!CSG && WIDE_N == 8) {
        if (sc.all_wide8) {                                  // (wave-uniform)
            const double reach = r.maxd < tmax ? r.maxd : tmax;
            double bt = INFINITY;
            int slot = -1;
            bool tie = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = wide.t[j] >= 0.0 && wide.t[j] <= reach;
                if (ok && wide.t[j] == bt) tie = true;
                if (ok && wide.t[j] < bt) { bt = wide.t[j]; slot = j; tie = false; }
            }
            if (active && !tie && slot >= 0) {
                int32_t idx = sc.wide[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) if (slot == q) idx = sc.wide[q];
                const uint32_t f = (slot < 4 ? wide.faces[0] >> (8 * slot) : wide.faces[1] >> (8 * (slot - 4))) & 255u;
                best.prim = idx; best.t = bt; best.a0 = (int32_t)(f & 15u) - 1; best.a1 = (int32_t)(f >> 4) - 1;
                best.u = best.v = best.w = 0.0f;
            }
            active = active && tie;                          // (the decided rays are done; a tied ray walks)
            if (!__any(active)) return best.prim >= 0;
        }
    }
#endif
    int32_t node = 0, sp = 0;
    // The prefill round, packed. Asked solid by solid the wave ran the evaluator 3.3 times per segment round of a prism pass for 21 asking
    // lanes of 38 live ones each (tools/path_prof.py prism, round 5) — 67 (ray, solid) questions, one wave's worth, in three turns.
    // Here the questions are dealt out 64 to a turn: lane L of turn i answers question 64 i + L (questions in solid order, then lane order:
    // the source lane is the rank-th set bit of that solid's gate mask), fetches the asking lane's ray through the cross-lane network,
    // evaluates it with the per-lane evaluator — the same function on the same bits as before — and the asking lane fetches the answer
    // back into its mailbox. Same answers, same mailbox order (solid 0 .. 3); the in-loop prefill round below is then skipped.
    if constexpr (CSG && FASTONLY && CSG_MAILBOX >= 4 && !UNIFORM_ITEMS && RSX_PREFILL_PACK != 0) {
        if (prefill) {
            const int lane = threadIdx.x % WAVE;
            const int n_solids = (sc.wide_csg[1] < 0) ? 1 : (sc.wide_csg[2] < 0) ? 2 : (sc.wide_csg[3] < 0) ? 3 : 4;      // (wave-uniform)
            unsigned long long gmask[4] = {0ULL, 0ULL, 0ULL, 0ULL};
            bool asks[4] = {false, false, false, false};
            int base[5] = {0, 0, 0, 0, 0};
            uint32_t askbits = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < n_solids) {
                    const UPrim up = uniform_prim(sc.prims_uniform, sc.wide_csg[k]);
                    const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
                    double f, b;
                    // (... and not a solid whose padded box the ray enters beyond the nearest analytic answer, RSX_PREFILL_CULL: every hit of it lies
                    // inside the box, so it is never the nearest — with the questions packed, one question less is work less)
                    if (aabb_rcp(lo, hi, r, rx, ry, rz, f, b) && active && !(f > t_wide_near)) askbits |= 1u << k;      // BoundPrimitive.hit gate (boundprimitive.pyx:42-51)
                }
            }
            // A solid that half the wave asks about (the coherent first segments of the camera's rays) is cheaper answered for all its lanes at
            // once by the wave-wide evaluator — tree and operand records over the scalar data path — than as questions of a packed turn, which
            // costs about twice a wave-wide run: such solids are answered here, one loop trip each, and leave the packing.
            if constexpr (RSX_PREFILL_UNIFORM_MIN <= WAVE) {
                for (int k = 0; k < n_solids; ++k) {                         // (wave-uniform trip count; one copy of the evaluator)
                    const bool ask = ((askbits >> k) & 1u) != 0u;
                    if (__popcll(__ballot(ask)) < RSX_PREFILL_UNIFORM_MIN) continue;
                    const int32_t uidx = k == 0 ? sc.wide_csg[0] : k == 1 ? sc.wide_csg[1] : k == 2 ? sc.wide_csg[2] : sc.wide_csg[3];
                    const RSX_CONST_AS CsgFast *flat = sc.csgfast_uniform ? (const RSX_CONST_AS CsgFast *)(unsigned long long)(sc.csgfast_uniform + uidx) : nullptr;
                    if (flat == nullptr || flat->n_leaves <= 0 || mesh_stack.lds_levels < 2 * flat->n_leaves) continue;   // (left to the packed turns: the redo pass)
                    Hit found;
                    found.prim = -1; found.t = 0; found.a0 = found.a1 = 0; found.leaf = 0; found.flags = 0;
                    found.u = found.v = found.w = 0.0f; found.hx = found.hy = found.hz = 0.0;
                    const int fast = csg_fast_hit_uniform(sc.csgfast_uniform, sc.prims_uniform, sc.prims, uidx, ask, r, mesh_stack, found);
                    if (ask) {
                        const bool got = fast == 1;
                        const uint32_t meta = ((uint32_t)found.a0 & 0xffu) | (((uint32_t)found.a1 & 0xffu) << 8) | (found.flags << 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (j == k) { mb_t[j] = got ? found.t : -1.0; mb_leaf[j] = got ? found.leaf : 0; mb_meta[j] = got ? meta : 0u; }
                        if (fast < 0) work |= 0x80000000u;                    // the redo pass traces this ray again
                        askbits &= ~(1u << k);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asks[k] = ((askbits >> k) & 1u) != 0u;
                gmask[k] = __ballot(asks[k]);
                base[k + 1] = base[k] + __popcll(gmask[k]);
            }
            const int n_questions = base[4];
#if RSX_PHASE_PROF == 3
            if (phase_acc) { phase_acc[20] += 1; phase_acc[21] += n_questions; phase_acc[22] += (n_questions + WAVE - 1) / WAVE; phase_acc[23] += __popcll(__ballot(active)); }
#endif
            const unsigned long long below = (1ULL << lane) - 1ULL;
            auto pull = [&](int from, double v) {                           // the value lane `from` holds (every lane takes part)
                const long long bits = __double_as_longlong(v);
                const int lo32 = __builtin_amdgcn_ds_bpermute(from << 2, (int)(bits & 0xffffffffLL)), hi32 = __builtin_amdgcn_ds_bpermute(from << 2, (int)(bits >> 32));
                return __longlong_as_double(((long long)hi32 << 32) | (long long)(uint32_t)lo32);
            };
            for (int turn = 0; turn * WAVE < n_questions; ++turn) {
                const int q_mine = turn * WAVE + lane;
                const bool answering = q_mine < n_questions;
                const int kq = q_mine >= base[3] ? 3 : q_mine >= base[2] ? 2 : q_mine >= base[1] ? 1 : 0;
                int rank = q_mine - (kq == 3 ? base[3] : kq == 2 ? base[2] : kq == 1 ? base[1] : 0);
                unsigned long long m = kq == 3 ? gmask[3] : kq == 2 ? gmask[2] : kq == 1 ? gmask[1] : gmask[0];
                int src = 0;                                                 // the rank-th set bit of m (binary search on popcounts)
#pragma unroll
                for (int width = 32; width >= 1; width >>= 1) {
                    const unsigned long long low = m & ((1ULL << width) - 1ULL);
                    const int c = __popcll(low);
                    const bool upper = rank >= c;
                    rank = upper ? rank - c : rank;
                    src = upper ? src + width : src;
                    m = upper ? m >> width : low;
                }
                if (!answering) src = lane;
                Ray rq;
                rq.ox = pull(src, r.ox); rq.oy = pull(src, r.oy); rq.oz = pull(src, r.oz);
                rq.dx = pull(src, r.dx); rq.dy = pull(src, r.dy); rq.dz = pull(src, r.dz); rq.maxd = pull(src, r.maxd);
                Hit ans;
                ans.prim = -1; ans.t = 0; ans.a0 = ans.a1 = 0; ans.leaf = 0; ans.flags = 0; ans.u = ans.v = ans.w = 0.0f; ans.hx = ans.hy = ans.hz = 0.0;
                int code = 0;                                                // 1 hit, 0 miss, 2 the redo pass
                if (answering) {
                    const int32_t qidx = kq == 3 ? sc.wide_csg[3] : kq == 2 ? sc.wide_csg[2] : kq == 1 ? sc.wide_csg[1] : sc.wide_csg[0];
                    bool needs_stream = false;
                    primitive_first_hit<CSG, FASTONLY>(sc, qidx, sc.prims[qidx], rq, mesh_stack, csg_state, ans, needs_stream);
                    code = needs_stream ? 2 : ans.prim >= 0 ? 1 : 0;
                }
                const uint32_t ans_meta = ((uint32_t)ans.a0 & 0xffu) | (((uint32_t)ans.a1 & 0xffu) << 8) | (ans.flags << 16);
                // the asking lanes take their answers home: solid k's question of lane L is number base[k] + (set bits of gmask[k] below L)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q_k = base[k] + __popcll(gmask[k] & below);
                    const bool here = asks[k] && (q_k >> 6) == turn;
                    const int from = here ? (q_k & (WAVE - 1)) : lane;
                    const double t_k = pull(from, ans.t);
                    const int meta_k = __builtin_amdgcn_ds_bpermute(from << 2, (int)ans_meta), leaf_k = __builtin_amdgcn_ds_bpermute(from << 2, ans.leaf),
                              code_k = __builtin_amdgcn_ds_bpermute(from << 2, code);
                    if (here) {
                        mb_t[k] = code_k == 1 ? t_k : -1.0; mb_leaf[k] = code_k == 1 ? leaf_k : 0; mb_meta[k] = code_k == 1 ? (uint32_t)meta_k : 0u;
                        if (code_k == 2) work |= 0x80000000u;                  // top bit of the cost counter: trace this ray again with the stream merge
                    }
                }
            }
            // every live lane's mailbox now names the solids in their order, asked or not (a failed gate is an answer too)
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < n_solids && active) { mb_idx[k] = sc.wide_csg[k]; }
            if (active) mb_next = n_solids == CSG_MAILBOX ? 0 : n_solids;
            work += 16u * (uint32_t)n_solids;
            prefill = false;
            if constexpr (CSG_ANSWERED) {                    // the nearest answer now includes those of the CSG primitives
#pragma unroll
                for (int j = 0; j < 4; ++j) if (mb_idx[j] >= 0 && mb_t[j] >= 0.0 && mb_t[j] < t_cull) t_cull = mb_t[j];
#if RSX_WORLD_CULL == 0
                t_cull = -INFINITY;
#endif
            }
        }
    }
#if RSX_WIDE_ALL_SHORTCUT
    // The same short cut for the fast forms of a CSG scene (DScene::all_answered_csg: a prism scene's seven primitives): the analytic
    // primitives are in the wide slots, the solids were answered in the prefill round above (the mailbox; a solid left out because its
    // box begins beyond the nearest answer cannot be the nearest either) — a ray takes its nearest eligible answer; a ray whose nearest
    // is shared walks, and so does one that waits for the redo pass (an answer of it is missing).
    if constexpr (CSG_ANSWERED && CSG_MAILBOX >= 4) {
        if (sc.all_answered_csg && !prefill) {               // (wave-uniform)
            const double reach = r.maxd < tmax ? r.maxd : tmax;
            double bt = INFINITY;
            int pick = -1;                                    // 0 .. WIDE_N - 1: a wide slot; 8 + k: mailbox entry k
            bool tie = false;
#pragma unroll
            for (int j = 0; j < WIDE_N; ++j) {
                const bool ok = sc.wide[j] >= 0 && wide.t[j] >= 0.0 && wide.t[j] <= reach;
                if (ok && wide.t[j] == bt) tie = true;
                if (ok && wide.t[j] < bt) { bt = wide.t[j]; pick = j; tie = false; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = mb_idx[k] >= 0 && mb_t[k] >= 0.0 && mb_t[k] <= reach;
                if (ok && mb_t[k] == bt) tie = true;
                if (ok && mb_t[k] < bt) { bt = mb_t[k]; pick = 8 + k; tie = false; }
            }
            const bool decided = active && !tie && !(work >> 31);
            if (decided && pick >= 0) {
                if (pick < 8) {
                    int32_t idx = sc.wide[0];
                    uint32_t f = wide.faces[0];
#pragma unroll
                    for (int q = 1; q < WIDE_N; ++q) if (pick == q) { idx = sc.wide[q]; f = q < 4 ? wide.faces[0] >> (8 * q) : wide.faces[1] >> (8 * (q - 4)); }
                    f &= 255u;
                    best.prim = idx; best.t = bt; best.a0 = (int32_t)(f & 15u) - 1; best.a1 = (int32_t)(f >> 4) - 1;
                    best.u = best.v = best.w = 0.0f;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (pick == 8 + k) {
                            best.prim = mb_idx[k]; best.t = mb_t[k]; best.leaf = mb_leaf[k];
                            best.a0 = (int32_t)(int8_t)(mb_meta[k] & 0xff); best.a1 = (int32_t)(int8_t)((mb_meta[k] >> 8) & 0xff);
                            best.flags = mb_meta[k] >> 16; best.u = best.v = best.w = 0.0f; best.hx = best.hy = best.hz = 0.0;
                        }
                    }
                }
            }
            active = active && !decided;
        }
    }
#endif
    while (__any(active)) {
        double distance = 0;
        int32_t count = 0;
        const int32_t *items = sc.witems;
        if (prefill) {
            if (active) count = (sc.wide_csg[1] < 0) ? 1 : (sc.wide_csg[2] < 0) ? 2 : (sc.wide_csg[3] < 0) ? 3 : 4;
        }
#if RSX_PHASE_PROF == 2 || RSX_PHASE_PROF == 3
        const unsigned long long ph2_d0 = clock64();
#endif
        if (active) { UTIL_COUNT(phase_acc, 0) }
        if (active && !prefill) {
#if RSX_UTIL_PROF == 2
            const rsx_kdnode nd = descend(wnodes, node, r, ad, tmin, tmax, st, sp, phase_acc);
#else
            const rsx_kdnode nd = descend<true>(wnodes, node, r, ad, tmin, tmax, st, sp, nullptr, t_cull);
#endif
            distance = r.maxd < tmax ? r.maxd : tmax;
            items += nd.u.leaf.first_item;
            count = nd.count;
            // a leaf whose items are all wide primitives brings its item list in the node itself (rsx_scene_create): the answers are
            // in registers, so the visit is two compares per item — in list order, `<=` as in the item loop below (kdtree.pyx:113)
            const int32_t tag = nd.u.leaf.pad;
            if (tag < 0) {
                const int n_tagged = (tag >> 28) & 7;          // bit 31: tagged; bits 28..30: items (<= 6); 3 bits per item: its wide slot
#pragma unroll
                for (int j = 0; j < (WIDE_N == 2 ? 2 : 6); ++j) {
                    if (j < n_tagged) {
                        const int slot = (tag >> (3 * j)) & 7;
                        double t;
                        int32_t faces;
                        wide_lookup<WIDE_N>(wide, slot, t, faces);
                        if (t >= 0.0 && t <= distance) {
                            distance = t;
                            int32_t prim = sc.wide[0];
#pragma unroll
                            for (int q = 1; q < WIDE_N; ++q) if (slot == q) prim = sc.wide[q];
                            best.prim = prim; best.t = t; best.a0 = (faces & 15) - 1; best.a1 = (faces >> 4) - 1;
                            best.u = best.v = best.w = 0.0f;
                        }
                    }
                }
                count = 0;
            }
        }
#if RSX_PHASE_PROF == 2
        phase_acc[4] += clock64() - ph2_d0;
#elif RSX_PHASE_PROF == 3
        if (phase_acc) { phase_acc[5] += clock64() - ph2_d0; phase_acc[8] += 1; }
#endif
        if constexpr (!UNIFORM_ITEMS) {
        for (int32_t k = 0; __any(k < count); ++k) {
            const bool have = k < count;
            int32_t idx = 0;
            if (prefill) idx = k == 0 ? sc.wide_csg[0] : k == 1 ? sc.wide_csg[1] : k == 2 ? sc.wide_csg[2] : sc.wide_csg[3];
            else if (have) idx = items[k];
            const rsx_primitive &p = sc.prims[idx];
            Hit cand;
            cand.prim = -1;
            work += CSG ? 16 : 4;
            const int slot = wide_slot<WIDE_N>(sc, idx);          // >= 0: answered before the traversal began
            const bool is_wide = slot >= 0;
            if (have && is_wide) {
                double t;
                int32_t faces;
                wide_lookup<WIDE_N>(wide, slot, t, faces);
                if (t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
            }
            // A CSG primitive sits in several leaves (its box is that of the whole solid) and Primitive.hit(ray) does not depend on
            // the leaf: the answer of the first visit is kept per lane (CSG_MAILBOX entries, round robin) and later leaves reuse it.
            bool cached = false;
            if constexpr (CSG && FASTONLY && CSG_MAILBOX > 0) {
                if (have && !is_wide && is_csg(p.type)) {
#pragma unroll
                    for (int j = 0; j < CSG_MAILBOX; ++j) {
                        if (mb_idx[j] == idx) {
                            cached = true;
                            if (mb_t[j] >= 0.0) {
                                cand.prim = idx; cand.t = mb_t[j]; cand.leaf = mb_leaf[j];
                                cand.a0 = (int32_t)(int8_t)(mb_meta[j] & 0xff); cand.a1 = (int32_t)(int8_t)((mb_meta[j] >> 8) & 0xff);
                                cand.flags = mb_meta[j] >> 16; cand.u = cand.v = cand.w = 0.0f; cand.hx = cand.hy = cand.hz = 0.0;
                            }
                        }
                    }
                }
            }
            double f, b;
            bool gate = have && !is_wide && !cached && aabb_rcp(p.box_lower, p.box_upper, r, rx, ry, rz, f, b);   // BoundPrimitive.hit gate
            if (prefill && f > t_wide_near) gate = false;
            const bool is_mesh = MESHES && gate && p.type == RSX_PRIM_MESH;
            // Mesh primitives are traced one primitive at a time with everything about the primitive wave-uniform (matrix, mesh
            // descriptor, array bases: scalar loads, SGPRs): a wave that straddles several instances takes one turn per instance.
            // MESHES false (the host knows the scene holds no mesh): the walk and the triangle test are not instantiated — on the
            // path kernels they held the register peak though no path ever entered them (Cornell box 28.3 -> 24.6 ms per pass).
            if constexpr (MESHES) {
                unsigned long long todo = __ballot(is_mesh);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int32_t uidx = __builtin_amdgcn_readlane(idx, leader);
                    const bool mine = is_mesh && idx == uidx;
                    todo &= ~__ballot(mine);
                    const UPrim up = uniform_prim(sc.prims_uniform, uidx);
                    Ray l = r;
                    if (mine) l = to_local_uniform(up, r);
                    const UMesh um = (UMesh)(unsigned long long)(sc.meshes + up->mesh);
                    MeshHit mh;
                    if (mesh_trace_wave<STAGE_MIN>(mine, um, l, mesh_stack, mh, work, phase_acc)) {
                        cand.prim = idx; cand.t = (double)mh.t; cand.a0 = mh.tri; cand.a1 = 0; cand.u = mh.u; cand.v = mh.v; cand.w = mh.w;
                    }
                }
            }
            // The prefill round asks every lane about the SAME solid (sc.wide_csg[k]): the wave-wide evaluator answers it — the flattened tree,
            // every node's and leaf's box, matrix and parameters over the scalar data path instead of a 376-byte record per lane and operand —
            // with the lanes' own rays; same operations on the same values per lane as csg_fast_hit. Round 5's phase profile of a prism
            // pass (tools/path_prof.py prism): 0.77 of the path kernel's time is world_trace_wave and 0.88 of that this round's evaluations.
            bool prefilled = false;
            if constexpr (CSG && FASTONLY && RSX_PREFILL_UNIFORM && !(CSG_MAILBOX >= 4 && !UNIFORM_ITEMS && RSX_PREFILL_PACK != 0)) {
                if (prefill) {
                    const int32_t uidx = __builtin_amdgcn_readfirstlane(idx);
                    const RSX_CONST_AS CsgFast *flat = sc.csgfast_uniform ? (const RSX_CONST_AS CsgFast *)(unsigned long long)(sc.csgfast_uniform + uidx) : nullptr;
                    if (flat != nullptr && flat->n_leaves > 0 && mesh_stack.lds_levels >= 2 * flat->n_leaves) {      // (wave-uniform)
                        prefilled = true;
                        const bool ask = gate && !is_mesh;
#if RSX_PHASE_PROF == 3
                        if (phase_acc) { phase_acc[20] += 1; phase_acc[21] += __popcll(__ballot(ask)); phase_acc[22] += __any(ask) ? 1 : 0; phase_acc[23] += __popcll(__ballot(active)); }
#endif
                        if (__any(ask)) {
                            Hit found;
                            found.prim = -1; found.t = 0; found.a0 = found.a1 = 0; found.leaf = 0; found.flags = 0;
                            found.u = found.v = found.w = 0.0f; found.hx = found.hy = found.hz = 0.0;
                            const int fast = csg_fast_hit_uniform(sc.csgfast_uniform, sc.prims_uniform, sc.prims, uidx, ask, r, mesh_stack, found);
                            if (ask) {
                                if (fast == 1) cand = found;
                                if (fast < 0) work |= 0x80000000u;                    // (as below: the redo pass traces this ray again)
                            }
                        }
                    }
                }
            }
            if (gate && !is_mesh && !prefilled) {
                bool needs_stream = false;
                primitive_first_hit<CSG, FASTONLY>(sc, idx, p, r, mesh_stack, csg_state, cand, needs_stream);
                if (FASTONLY && needs_stream) work |= 0x80000000u;            // top bit of the cost counter: trace this ray again with the stream merge
            }
            if constexpr (CSG && FASTONLY && CSG_MAILBOX > 0) {
                if (have && !is_wide && !cached && is_csg(p.type)) {            // (a failed gate is an answer too)
                    const bool got = cand.prim >= 0;
                    const uint32_t meta = ((uint32_t)cand.a0 & 0xffu) | (((uint32_t)cand.a1 & 0xffu) << 8) | (cand.flags << 16);
#pragma unroll
                    for (int j = 0; j < CSG_MAILBOX; ++j) {
                        if (mb_next == j) { mb_idx[j] = idx; mb_t[j] = got ? cand.t : -1.0; mb_leaf[j] = got ? cand.leaf : 0; mb_meta[j] = got ? meta : 0u; }
                    }
                    mb_next = mb_next + 1 == CSG_MAILBOX ? 0 : mb_next + 1;
                }
            }
            if (!prefill && cand.prim >= 0 && cand.t <= distance) { distance = cand.t; best = cand; }   // `<=`: later item wins ties
        }
        if (prefill) {
            prefill = false;
            if constexpr (CSG_ANSWERED) {                    // the nearest answer now includes those of the CSG primitives
#pragma unroll
                for (int j = 0; j < 4; ++j) if (mb_idx[j] >= 0 && mb_t[j] >= 0.0 && mb_t[j] < t_cull) t_cull = mb_t[j];
#if RSX_WORLD_CULL == 0
                t_cull = -INFINITY;
#endif
            }
            continue;
        }
        } else {
        for (int32_t k = 0; __any(k < count); ++k) {
            const bool have = k < count;
#if RSX_UTIL_PROF == 2
            if (have) { UTIL_COUNT(phase_acc, 2) }
#endif
            const int32_t idx = have ? items[k] : 0;
            Hit cand;
            cand.prim = -1;
            work += CSG ? 16 : 4;
            // Leaf items are processed one PRIMITIVE at a time with everything about the primitive wave-uniform: bounding box,
            // type, matrix, parameters and mesh descriptor come in over the scalar data path (SGPRs, no vector loads of the 300-byte
            // record per lane). Coherent waves meet one primitive per leaf item; a wave that straddles several takes one turn each.
            unsigned long long todo = __ballot(have);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int32_t uidx = __builtin_amdgcn_readlane(idx, leader);
                const bool mine = have && idx == uidx;
                todo &= ~__ballot(mine);
                if (uidx == sc.wide[0] || uidx == sc.wide[1]) {                     // answered before the traversal began
                    double t;
                    int32_t faces;
                    wide_lookup<2>(wide, uidx == sc.wide[0] ? 0 : 1, t, faces);
                    if (mine && t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
                    continue;
                }
                const UPrim up = uniform_prim(sc.prims_uniform, uidx);
                const int32_t type = up->type;
                const bool analytic = type != RSX_PRIM_MESH && !(CSG && is_csg(type));
                bool gate = mine;
                if (!analytic) {
                    const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
                    double f, b;
                    gate = mine && aabb_rcp(lo, hi, r, rx, ry, rz, f, b);           // BoundPrimitive.hit gate (boundprimitive.pyx:42-51)
                    if (!__any(gate)) continue;
                }
                if (type == RSX_PRIM_MESH) {
#ifdef RSX_ABLATE_MESH
                    continue;                                 // (timing ablation: results are wrong)
#endif
#if RSX_PHASE_PROF == 2
                    const unsigned long long ph2_m0 = clock64();
#endif
                    Ray l = r;
                    if (gate) l = to_local_uniform(up, r);
                    const UMesh um = (UMesh)(unsigned long long)(sc.meshes + up->mesh);
                    MeshHit mh;
                    if (mesh_trace_wave<STAGE_MIN>(gate, um, l, mesh_stack, mh, work, phase_acc)) {
                        cand.prim = idx; cand.t = (double)mh.t; cand.a0 = mh.tri; cand.a1 = 0; cand.u = mh.u; cand.v = mh.v; cand.w = mh.w;
                    }
#if RSX_PHASE_PROF == 2
                    phase_acc[2] += clock64() - ph2_m0; phase_acc[5] += 1;
#endif
                    continue;
                }
#if RSX_UTIL_PROF == 2
                if (gate) { UTIL_COUNT(phase_acc, 6) }
#endif
                if (CSG && is_csg(type)) {
                    // the state-free evaluator for the whole wave at once, the tree and its operands' records over the scalar data path
                    // (csg_fast_hit_uniform); a lane it cannot answer goes the way csg_fast_hit's -1 goes
                    const RSX_CONST_AS CsgFast *flat = sc.csgfast ? (const RSX_CONST_AS CsgFast *)(unsigned long long)(sc.csgfast + uidx) : nullptr;
                    bool answered = false;
#ifndef RSX_NO_UNIFORM_CSG
                    if (flat != nullptr && flat->n_leaves > 0 && mesh_stack.lds_levels >= 2 * flat->n_leaves) {
                        if (uidx != last_csg) { last_csg = uidx; last_meta = 0; }
                        const bool ask = gate && (last_meta >> 28) == 0u;
                        if (__any(ask)) {
                            Hit found;
                            found.prim = -1; found.t = 0; found.a0 = found.a1 = 0; found.leaf = 0; found.flags = 0;
                            const int fast = csg_fast_hit_uniform(sc.csgfast, sc.prims_uniform, sc.prims, uidx, ask, r, mesh_stack, found);
                            if (ask) {
                                last_t = found.t; last_leaf = found.leaf;
                                last_meta = ((uint32_t)found.a0 & 0xffu) | (((uint32_t)found.a1 & 0xffu) << 8) | ((found.flags & 0xffu) << 16) |
                                            ((fast == 1 ? 1u : fast == 0 ? 2u : 3u) << 28);
                            }
                        }
                        const uint32_t state = last_meta >> 28;
                        if (gate && state == 1u) {
                            cand.prim = idx; cand.t = last_t;
                            cand.a0 = (int32_t)(int8_t)(last_meta & 0xffu); cand.a1 = (int32_t)(int8_t)((last_meta >> 8) & 0xffu);
                            cand.u = cand.v = cand.w = 0.0f;
                            cand.leaf = last_leaf; cand.flags = (last_meta >> 16) & 0xffu;
                            cand.hx = cand.hy = cand.hz = 0.0;
                        }
                        if (gate && state == 3u) {
                            if constexpr (FASTONLY) work |= 0x80000000u;           // top bit of the cost counter: trace this ray again with the stream merge
                            else {
                                bool needs_stream = false;
                                primitive_first_hit<CSG, false>(sc, idx, sc.prims[idx], r, mesh_stack, csg_state, cand, needs_stream);
                            }
                        }
                        answered = true;
                    }
#endif
                    if (!answered && gate) {
                        bool needs_stream = false;
                        primitive_first_hit<CSG, FASTONLY>(sc, idx, sc.prims[idx], r, mesh_stack, csg_state, cand, needs_stream);
                        if (FASTONLY && needs_stream) work |= 0x80000000u;        // top bit of the cost counter: trace this ray again with the stream merge
                    }
                    continue;
                }
                {                                                                   // sphere / box / cylinder: Primitive.hit, first root
                    double t;
                    int32_t faces;
                    analytic_first_root(sc.prims, up, uidx, gate, r, rx, ry, rz, t, faces);
                    if (t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
                }
            }
            if (cand.prim >= 0 && cand.t <= distance) { distance = cand.t; best = cand; }   // `<=`: later item wins ties
        }
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

template <bool CSG, bool MESHES = true>
__device__ __forceinline__ void finalise(const DScene &sc, const Ray &r, const Hit &h, Geom &g) {
    const rsx_primitive &p = sc.prims[h.prim];
    if constexpr (CSG) {
        if (is_csg(p.type)) { csg_geom(sc, r, h, g); return; }
    }
    const Ray l = to_local(p, r);
    if (MESHES && p.type == RSX_PRIM_MESH) mesh_geom(sc.meshes[p.mesh], l, h.t, h.a0, h.u, h.v, h.w, g);
    else analytic_geom(p, l, h.t, h.a0, h.a1, g);
}

