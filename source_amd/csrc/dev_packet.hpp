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

// The scene as the packet walk reads it: where it lies in the kernel-argument segment, field by field at the point of use (scalar
// loads that hit the constant cache) — held by value, its pointers and tables sat in scalar registers through the whole unit and the
// registers the walk needed were spilled around them.
typedef const RSX_CONST_AS DScene *PScene;

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

// Stack of the packet: entry `sp` (wave-uniform) = the node to visit + each lane's tmax there. In LDS: ONE id per level, for every level
// (every lane writes the same word), and tmax[level][lane] (f64) for the first levels; deeper tmax rows spill to the wave's global region.
// 4 world + 14 mesh levels are 9.3 KB per wave: sixteen waves — four per SIMD — fit a CU's 160 KB (packet_lds_bytes, wave_stacks_packet).
#define PKT_WORLD_LDS_LEVELS 4
#define PKT_MESH_LDS_LEVELS 14
// the lane index, formed where it is used and opaque to the optimiser: an address built from it cannot be hoisted out of the walk (the
// per-lane 64-bit addresses of the rarely used global stack rows and of the record ring sat in six registers through every unit)
__device__ __forceinline__ int lane_here() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}
// The node ids of the pending entries live in ONE vector register of the wave — lane L holds the id of stack level L (world levels first,
// the mesh walk's behind them: `base`) — written with v_writelane and read with v_readlane: one instruction each where an LDS word cost
// two moves, the address arithmetic, the write and, on the way back, a read, a wait and a readfirstlane. (wdepth + mdepth <= 64: render()
// takes the packet kernel only then.)
#ifndef RSX_PKT_CLUSTERS
#define RSX_PKT_CLUSTERS 1         // (2: a second level, the members' own boxes for clusters of at most four — measured: 20.93 against 20.99 ms, not worth its code) units none of whose rays comes near a primitive that is not answered up front skip the world walk (world_trace_packet)
#endif
#ifndef RSX_PKT_ASM
#define RSX_PKT_ASM 1              // the descent (branch steps down to a leaf) as hand-written wave-level code: packet_descend
#endif
struct IdStack { int v; int base; };
__device__ __forceinline__ void pstack_push(const Stack &st, IdStack &ids, int32_t sp, int32_t id, double t) {
    const int lane = (int)(threadIdx.x % WAVE);
    {   // (lane select through m0: a VOP3 may name one scalar register, the id is the other)
        const int at = sp + ids.base;
        int keep;
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(ids.v), "=&s"(keep) : "s"(id), "s"(at));
    }
    if (__builtin_expect(sp < st.lds_levels, 1)) *reinterpret_cast<double *>(smem + st.lds_t + (sp * WAVE + lane) * 8) = t;    // (scalar branch)
    else reinterpret_cast<double *>(st.gt)[(sp - st.lds_levels) * WAVE + lane_here()] = t;
}
__device__ __forceinline__ void pstack_pop(const Stack &st, const IdStack &ids, int32_t sp, int32_t &id, double &t) {
    const int lane = (int)(threadIdx.x % WAVE);
    if (__builtin_expect(sp < st.lds_levels, 1)) t = *reinterpret_cast<const double *>(smem + st.lds_t + (sp * WAVE + lane) * 8);
    else t = __builtin_nontemporal_load(reinterpret_cast<const double *>(st.gt) + (sp - st.lds_levels) * WAVE + lane_here());
    id = __builtin_amdgcn_readlane(ids.v, sp + ids.base);
}

__host__ __device__ __forceinline__ int packet_world_levels(int wdepth) { return wdepth < PKT_WORLD_LDS_LEVELS ? wdepth : PKT_WORLD_LDS_LEVELS; }
__host__ __device__ __forceinline__ int packet_mesh_levels(int mdepth) { return mdepth < PKT_MESH_LDS_LEVELS ? mdepth : PKT_MESH_LDS_LEVELS; }
// Bytes of one wave's global spill region — the stack rows that do not fit LDS, of whichever walk the launch runs: the per-lane walks
// keep (wlds, mlds) levels in LDS and spill 12 B per lane and level, the packet keeps its own (packet_*_levels: fewer than mlds when the
// flattened CSG trees asked for more than PKT_MESH_LDS_LEVELS) and spills 8 B. plan() allocates this stride, both carvers step by it.
__host__ __device__ __forceinline__ size_t spill_wave_bytes(int wdepth, int wlds, int mdepth, int mlds) {
    const int per_lane = (wdepth - wlds) + (mdepth - mlds), packet = (wdepth - packet_world_levels(wdepth)) + (mdepth - packet_mesh_levels(mdepth));
    const size_t a = (size_t)(per_lane > 0 ? per_lane : 0) * WAVE * 12, b = (size_t)(packet > 0 ? packet : 0) * WAVE * 8;
    const size_t need = a > b ? a : b;
    return need > 0 ? need : (size_t)WAVE * 12;
}
// csg_rows > 0 (scenes with CSG solids in the state-free evaluator's form): that many rows of (f64 root, i32 face / axis / exit) per lane
// behind the stacks — where csg_fast_hit_uniform leaves the operands' roots (two per leaf of the biggest tree)
__host__ __device__ __forceinline__ size_t packet_stack_bytes(int wdepth, int mdepth) {
    const size_t levels = (size_t)(packet_world_levels(wdepth) + packet_mesh_levels(mdepth));
    return (levels * WAVE * 8 + (size_t)(wdepth + mdepth) * 4 + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t packet_lds_bytes(int wdepth, int mdepth, int csg_rows = 0) {       // per wave
    return packet_stack_bytes(wdepth, mdepth) + (size_t)csg_rows * WAVE * 12;
}
// carve the wave's LDS region and global spill region (spill_wave_bytes) into the packet's world stack and mesh stack
__device__ __forceinline__ void wave_stacks_packet(const DScene &sc, Stack &ws, Stack &ms) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int pw = packet_world_levels(sc.wdepth), pm = packet_mesh_levels(sc.mdepth);
    const uint32_t base = (uint32_t)wave * (uint32_t)packet_lds_bytes(sc.wdepth, sc.mdepth, sc.csg_fast_rows);
    const size_t gwave = (size_t)blockIdx.x * (blockDim.x / WAVE) + wave;
    char *gt = sc.spill + gwave * spill_wave_bytes(sc.wdepth, sc.wlds, sc.mdepth, sc.mlds);
    ws.stage = nullptr; ms.stage = nullptr;
    ws.lds_t = base; ws.lds_id = base + (uint32_t)(pw + pm) * WAVE * 8; ws.gt = gt; ws.gid = nullptr; ws.lds_levels = pw;
    ms.lds_t = base + (uint32_t)pw * WAVE * 8; ms.lds_id = ws.lds_id + (uint32_t)sc.wdepth * 4;
    ms.gt = gt + (size_t)(sc.wdepth - pw) * WAVE * 8; ms.gid = nullptr; ms.lds_levels = pm;
}

#define PKT_EMPTY (-INFINITY)
#ifndef RSX_STEP_CARRY_IN
#define RSX_STEP_CARRY_IN 1      // the lanes-with-a-range mask is handed from step to step (and re-formed once per descent) instead of a vector compare per step: 23.24 -> 23.15 ms
#endif
#ifndef RSX_STEP_UNIFORM_ORIGIN
#define RSX_STEP_UNIFORM_ORIGIN 1
#endif
#ifndef RSX_STEP_PREFETCH
#define RSX_STEP_PREFETCH 0
#endif

// a value that is the same in every lane, declared so to the compiler (it then lives in scalar registers)
__device__ __forceinline__ unsigned long long pkt_uniform64(unsigned long long v) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

#ifdef RSX_PKT_PROF
// tuning builds: wave-level event counts of the packet walk, summed into g_pkt by every wave (printed by rsx_synchronize)
__device__ unsigned long long g_pkt[16];
#define PKT_COUNT(slot, n) { pkc[slot] += (uint32_t)(n); }
#define PKT_ARG , uint32_t *pkc
#define PKT_PASS , pkc
#else
#define PKT_COUNT(slot, n)
#define PKT_ARG
#define PKT_PASS
#endif
enum { PKC_UNITS = 0, PKC_WSTEPS, PKC_WDIVS, PKC_WLEAVES, PKC_WITEMS, PKC_MVISITS, PKC_MSTEPS, PKC_MLEAVES, PKC_TRIS, PKC_PUSHES, PKC_POPS, PKC_DEFERS, PKC_MSLOW, PKC_N };

__device__ __forceinline__ bool pkt_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ULL; }

// Wave-level facts about the rays' space, fixed for one tree walk.
typedef unsigned long long lanemask;
struct PacketSpace {
    int fast;                  // bit k: on axis k the hoisted-reciprocal quotient needs no per-node range test (see packet_space)
    lanemask neg[3];           // lanes whose direction component k is negative
};

// exact_div's shortcut (dev_common.hpp) is valid while numerator and divisor stay clear of the exponent ranges where the hardware
// division rescales: |d| in [2^-300, 2^300] (AxisDiv::safe) and |num| in [2^-300, 2^300] or num == 0. Here num = split - o with every
// split inside the tree's bounds, so the numerator's test can be made ONCE per walk instead of at every node: with |o| >= 2^-240 a
// non-zero difference split - o is at least an ulp of the smaller operand's binade and never below 2^-300 (with o == 0 it is the split
// itself: the host notes whether any split is that small); with |o| and the bounds below 2^299 it never exceeds 2^300. An axis that
// fails this keeps the per-node test.
__device__ __forceinline__ PacketSpace packet_space(const Ray &r, const AxisDiv &ad, const double *lo, const double *hi, bool want, int splits) {
    PacketSpace ps;
    const double o[3] = {r.ox, r.oy, r.oz};
    int fast = 0;
    if (splits & 1) {                                    // (bit 0: every split lies inside [lo, hi]; bit 1: no split is closer to 0 than 2^-240)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = __builtin_fabs(o[k]);
            // an origin coordinate of exactly 0 (a camera on an axis plane) makes the numerator the split itself
            const bool origin_ok = (a >= 0x1p-240 && a <= 0x1p+299) || (a == 0.0 && (splits & 2));
            const bool ok = ((ad.safe >> k) & 1) && origin_ok && __builtin_fabs(lo[k]) <= 0x1p+299 && __builtin_fabs(hi[k]) <= 0x1p+299;
            if (!pkt_any(want && !ok)) fast |= 1 << k;
        }
    }
    ps.fast = fast;
    ps.neg[0] = __builtin_amdgcn_ballot_w64(r.dx < 0.0); ps.neg[1] = __builtin_amdgcn_ballot_w64(r.dy < 0.0); ps.neg[2] = __builtin_amdgcn_ballot_w64(r.dz < 0.0);
    return ps;
}

// One branch node for the packet, on split axis AXIS (the caller dispatches on the node's wave-uniform axis, so the lane's origin,
// direction and reciprocal components are named, not selected). In: the lane's range [tmin, tmax] (tmax == PKT_EMPTY: none).
// Out: the node to go to and the lane's range there; the other child, when some lane enters it too, is pushed.
// WORLD: plain division (skipped when no lane approaches the plane) and the cull of world_step (cull bits in nd.type >> 2, t_cull);
// else the hoisted-reciprocal quotient (dev_common.hpp: exact_div), range-tested per node only where packet_space could not vouch for it.
// Lane predicates are kept as 64-bit lane masks (what a vector compare writes anyway): their algebra then runs on the scalar unit, and a
// mask goes back into a select with inverse_ballot at no cost.
__device__ __forceinline__ lanemask pkt_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool pkt_lanes(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

template <bool WORLD>
__device__ __forceinline__ int32_t packet_step_axis(const UNode &nd, int32_t node, double o, double d, double y, bool fast, lanemask m_neg, lanemask &m_in,
                                                    double &tmin, double &tmax, const Stack &st, IdStack &ids, int32_t &sp, double t_cull PKT_ARG) {
    const double split = unode_split(nd);
    const int32_t lower = node + 1, upper = nd.count;
    const double num = split - o;
    // the reference's quotient (split - origin) / direction, kdtree3d.pyx:672 (exact_div, dev_common.hpp)
    const double q0 = num * y;
    const double rem = __builtin_fma(-d, q0, num);
    double plane = __builtin_fma(rem, y, q0);
    if (__builtin_expect(!fast, 0)) {
        const bool exact = div_operand_safe(d) && (div_operand_safe(num) || num == 0.0);
        if (pkt_any(!exact)) { if (!exact) plane = num / d; }
        PKT_COUNT(PKC_MSLOW, 1)
    }
    // kdtree3d.pyx:661-700 in one form. A ray parallel to the plane (d == 0) has quotient +-inf or NaN: "near only" below, and its near
    // child is `origin < split ? lower : upper` (:664) — which is what below_split (:675) gives when d < 0 is false.
    const lanemask m_lt = pkt_mask(o < split), m_eq = pkt_mask(o == split);
    // crosses the plane inside its range: not (:686) "plane > max_range or plane <= 0" (a lane without a range: plane <= -inf fails)
    const lanemask m_cross = pkt_mask(plane > 0.0) & pkt_mask(plane <= tmax);
    lanemask m_far = m_cross & pkt_mask(plane < tmin);                     // (:690) far child only
    // (the packet's rays leave ONE point — the pinhole, or its image in a mesh's space: the callers hold the origin in scalar registers —
    // and all 64 lanes are active in the walks, so m_lt is empty or full by construction: testing it for that cost every step a 64-bit
    // add and a vector compare, 24.1 -> 23.2 ms on configs[2] without them. RSX_STEP_UNIFORM_ORIGIN=0 restores the test)
#if RSX_STEP_UNIFORM_ORIGIN
    if (__builtin_expect(m_eq == 0ULL, 1)) {
#else
    if (__builtin_expect(m_eq == 0ULL && (m_lt == 0ULL || m_lt == ~0ULL), 1)) {
#endif
        // Every lane has the same near child — rays from one origin that does not lie on the plane: the visit order is a scalar fact.
        const bool lower_near = m_lt != 0ULL;
        const int32_t near_id = lower_near ? lower : upper, far_id = lower_near ? upper : lower;
        if constexpr (WORLD) {
            // world_step's cull: the near subtree holds wide primitives only and ends before the nearest wide answer — straight to the far child
            if ((nd.type >> (lower_near ? 2 : 3)) & 1) {
                const lanemask m_cull = m_cross & ~m_far & pkt_mask(plane < t_cull);
                tmin = pkt_lanes(m_cull) ? plane : tmin;
                m_far |= m_cull;
            }
        }
        const lanemask m_want_near = m_in & ~m_far;
        if (m_want_near) {
            if (m_cross) { pstack_push(st, ids, sp, far_id, pkt_lanes(m_cross) ? tmax : PKT_EMPTY); ++sp; PKT_COUNT(PKC_PUSHES, 1) }
            const double keep = pkt_lanes(m_cross) ? plane : tmax;         // (near-side lanes that cross go on to the plane, the others keep their range)
            tmax = pkt_lanes(m_want_near) ? keep : PKT_EMPTY;
            m_in = m_want_near;
            return near_id;
        }
        tmax = pkt_lanes(m_cross) ? tmax : PKT_EMPTY;                      // nobody enters the near child: the crossing lanes all enter the far one only
        m_in = m_cross;
        return far_id;
    }
    // The general form: lanes may differ in their near child (an origin exactly on the plane, rays with origins of their own).
    const lanemask m_lower_near = m_lt | (m_eq & m_neg);
    lanemask m_both = m_cross & ~m_far;
    if constexpr (WORLD) {
        const int32_t cull_bits = nd.type >> 2;
        const lanemask m_cullable = m_both & (((cull_bits & 1) ? m_lower_near : 0ULL) | ((cull_bits & 2) ? ~m_lower_near : 0ULL));
        if (m_cullable) {
            const lanemask m_cull = m_cullable & pkt_mask(plane < t_cull);
            tmin = pkt_lanes(m_cull) ? plane : tmin;
            m_far |= m_cull; m_both &= ~m_cull;
        }
    }
    lanemask b_up = m_both & ~m_lower_near;
    if (b_up != 0ULL && (m_both & m_lower_near) != 0ULL) {
        // both-crossing lanes disagree on the near child: the upper-first lanes come back to this node alone, after the others are
        // through with it
        pstack_push(st, ids, sp, node, pkt_lanes(b_up) ? tmax : PKT_EMPTY);
        ++sp;
        tmax = pkt_lanes(b_up) ? PKT_EMPTY : tmax;
        m_in &= ~b_up; m_both &= ~b_up;
        b_up = 0ULL;
        PKT_COUNT(PKC_DEFERS, 1)
    }
    const lanemask m_single_lower = m_far ^ m_lower_near;                  // the one child of a lane that enters one: far_only ? far : near
    const lanemask m_want_lower = m_in & (m_both | m_single_lower), m_want_upper = m_in & (m_both | ~m_single_lower);
    const bool upper_first = b_up != 0ULL;
    const int32_t first = upper_first ? upper : lower, second = upper_first ? lower : upper;
    const lanemask m_want_f = upper_first ? m_want_upper : m_want_lower, m_want_s = upper_first ? m_want_lower : m_want_upper;
    if (m_want_f) {
        if (m_want_s) { pstack_push(st, ids, sp, second, pkt_lanes(m_want_s) ? tmax : PKT_EMPTY); ++sp; PKT_COUNT(PKC_PUSHES, 1) }
        const double keep = pkt_lanes(m_both) ? plane : tmax;              // (a both-crossing lane's near child is `first` by construction)
        tmax = pkt_lanes(m_want_f) ? keep : PKT_EMPTY;
        m_in = m_want_f;
        return first;
    }
    tmax = pkt_lanes(m_want_s) ? tmax : PKT_EMPTY;
    m_in = m_want_s;
    return second;
}

template <bool WORLD>
__device__ __forceinline__ int32_t packet_step(const UNode &nd, int32_t node, const Ray &r, const AxisDiv &ad, const PacketSpace &ps, double &tmin, double &tmax,
                                               const Stack &st, IdStack &ids, int32_t &sp, double t_cull, lanemask &m_carry PKT_ARG) {
    const int axis = nd.type & 3;                                          // (wave-uniform: a scalar branch)
#if RSX_STEP_CARRY_IN
    lanemask &m_in = m_carry;                                              // the lanes with a range: what the last step (or pop) left, not a compare per step
#else
    lanemask m_in = pkt_mask(tmax != PKT_EMPTY);
    (void)m_carry;
#endif
    if (axis == 0) return packet_step_axis<WORLD>(nd, node, r.ox, r.dx, ad.yx, (ps.fast & 1) != 0, ps.neg[0], m_in, tmin, tmax, st, ids, sp, t_cull PKT_PASS);
    if (axis == 1) return packet_step_axis<WORLD>(nd, node, r.oy, r.dy, ad.yy, (ps.fast & 2) != 0, ps.neg[1], m_in, tmin, tmax, st, ids, sp, t_cull PKT_PASS);
    return packet_step_axis<WORLD>(nd, node, r.oz, r.dz, ad.yz, (ps.fast & 4) != 0, ps.neg[2], m_in, tmin, tmax, st, ids, sp, t_cull PKT_PASS);
}

#if RSX_PKT_ASM
// ---------------------------------------------------------------------------------------------------
// The descent by hand: every branch step from `node` down to the next leaf in ONE block of wave-level code. It is packet_step_axis's short
// form (every lane has the same near child; the hoisted-reciprocal quotient needs no range test) operation for operation — the same
// quotient (v_mul, v_fma, v_fma = exact_div), the same three range compares, the same child order, pushes and ranges per lane — and it
// hands the one step it does not cover back to packet_step: an origin exactly on the split plane, or a push beyond the LDS levels
// (flag 1: node / nd / masks untouched for that step). Compiled from C++ the same step cost 23 vector + 48 scalar instructions:
// the backend structurizes the whole unit loop, so the three-way axis dispatch and every early return became flag registers and
// flag branches, the node record was copied between register quads, the push went through two LDS words with their address moves.
// Here: 17 vector / LDS + ~30 scalar with a push, and nothing spilled inside.
//  * lane predicates are 64-bit masks in scalar registers; a select is a move under exec = mask (whole 64-bit operands);
//  * a lane without a range is a lane outside `m_in` — its tmax is whatever it was — and PKT_EMPTY is written once on the way out;
//  * near child: num = split - origin is the same number in every lane, `num > 0` (origin < split) fills vcc or leaves it empty;
//  * the node record sits in s[52:55] by name (an asm operand's halves cannot be named otherwise).
// WORLD: world_step's cull (node type bits 2 / 3, t_cull) as in packet_step_axis.
// (Measured and taken out again: both children's records asked for before the step's arithmetic — two s_load_dwordx4 at the top, a scalar
// select at the bottom: configs[2] 21.68 ms against 21.57 without. The walk waits for issue slots, not for its node records.)
typedef int pkt_i4 __attribute__((ext_vector_type(4)));
#define PKT_ASM_STEP(TAG, O, D, Y, CULL) \
    "Lax" TAG "%=:\n" \
    "v_mov_b64 %[num], " O "\n" \
    "v_add_f64 %[num], s[54:55], -%[num]\n" \
    "v_mul_f64 %[q], %[num], " Y "\n" \
    "v_fma_f64 %[pl], -" D ", %[q], %[num]\n" \
    "v_fma_f64 %[pl], %[pl], " Y ", %[q]\n" \
    "v_cmp_eq_f64 vcc, 0, %[num]\n" \
    "s_cbranch_vccnz Lslow%=\n" \
    "v_cmp_lt_f64 vcc, 0, %[num]\n" \
    "v_cmp_lt_f64 %[gt0], 0, %[pl]\n" \
    "v_cmp_le_f64 %[le], %[pl], %[tmax]\n" \
    "v_cmp_lt_f64 %[far], %[pl], %[tmin]\n" \
    "s_and_b64 %[cross], %[gt0], %[le]\n" \
    "s_and_b64 %[cross], %[cross], %[min]\n" \
    "s_and_b64 %[far], %[far], %[cross]\n" \
    "s_add_i32 %[lower], %[node], 1\n" \
    "s_cmp_lg_u64 vcc, 0\n" \
    "s_cselect_b32 %[nearid], %[lower], s53\n" \
    "s_cselect_b32 %[farid], s53, %[lower]\n" \
    CULL(TAG) \
    "s_andn2_b64 %[want], %[min], %[far]\n" \
    "s_cbranch_scc0 Lfaronly" TAG "%=\n" \
    "s_cmp_lg_u64 %[cross], 0\n" \
    "s_cbranch_scc0 Lnopush" TAG "%=\n" \
    "s_cmp_ge_i32 %[sp], %[levels]\n" \
    "s_cbranch_scc1 Lslow%=\n" \
    "v_mov_b64 %[p], %[ninf]\n" \
    "s_mov_b64 exec, %[cross]\n" \
    "v_mov_b64 %[p], %[tmax]\n" \
    "s_mov_b64 exec, %[ex]\n" \
    "v_lshl_add_u32 %[addr], %[sp], 9, %[lds]\n" \
    "ds_write_b64 %[addr], %[p]\n" \
    "s_add_i32 m0, %[sp], %[idbase]\n" \
    "s_nop 0\n" \
    "v_writelane_b32 %[ids], %[farid], m0\n" \
    "s_add_i32 %[sp], %[sp], 1\n" \
    "Lnopush" TAG "%=:\n" \
    "s_mov_b64 exec, %[cross]\n" \
    "v_mov_b64 %[tmax], %[pl]\n" \
    "s_mov_b64 exec, %[ex]\n" \
    "s_mov_b64 %[min], %[want]\n" \
    "s_mov_b32 %[node], %[nearid]\n" \
    "s_branch Lnext%=\n" \
    "Lfaronly" TAG "%=:\n" \
    "s_mov_b64 %[min], %[cross]\n" \
    "s_mov_b32 %[node], %[farid]\n" \
    "s_branch Lnext%=\n"
#define PKT_ASM_NOCULL(TAG)
#define PKT_ASM_CULL(TAG) \
    "s_cselect_b32 %[t0], 4, 8\n" \
    "s_and_b32 %[t0], %[t0], s52\n" \
    "s_cbranch_scc0 Lnocull" TAG "%=\n" \
    "v_cmp_lt_f64 %[gt0], %[pl], %[tcull]\n" \
    "s_andn2_b64 %[le], %[cross], %[far]\n" \
    "s_and_b64 %[gt0], %[gt0], %[le]\n" \
    "s_mov_b64 exec, %[gt0]\n" \
    "v_mov_b64 %[tmin], %[pl]\n" \
    "s_mov_b64 exec, %[ex]\n" \
    "s_or_b64 %[far], %[far], %[gt0]\n" \
    "Lnocull" TAG "%=:\n"
// Around the steps: the walk between two leaves that have work. A mesh leaf without triangles (most leaves a packet meets) and every
// "leave this leaf, pop until some lane has a range again" are transitions of a few instructions each; compiled, each cost ~45 vector and
// ~80 scalar instructions of flags, register moves, an LDS word for the id with its wait and readfirstlane, a vector compare + ballot per
// popped entry — per mesh visit about as much as the forty steps themselves. So the block below runs from "at a node" (or, `enter` set,
// "done with the leaf the walk is at") to the next leaf with work, the end of the walk, or a transition it hands back:
//   flag 0: at a leaf with work (a world leaf; a mesh leaf with triangles) — node / nd are the leaf's
//   flag 1: the branch step at `node` is one for packet_step (origin on the plane, push beyond the LDS levels)
//   flag 2: the stack is empty: the walk is over
//   flag 3: the next entry lies beyond the LDS levels: packet_pop fetches it (the leaf has been left: tmin is up to date)
// Leaving a leaf: tmin = tmax for the lanes that had a range in it (the next range of a ray begins where this one ended); a popped entry
// gives its lanes a range unless they are `done` (their ray found its hit) or the entry holds PKT_EMPTY for them (v_cmp_class: -inf).
#define PKT_ASM_HEAD \
    "s_mov_b64 %[ex], exec\n" \
    "s_mov_b32 %[m0save], m0\n" \
    "s_mov_b32 %[steps], 0\n" \
    "s_cmp_lg_u32 %[enter], 0\n" \
    "s_cbranch_scc1 Lpop%=\n" \
    "s_cmp_lt_i32 s52, 0\n" \
    "s_cbranch_scc1 Lleaf%=\n" \
    "Ltop%=:\n" \
    "s_and_b32 %[t0], s52, 3\n" \
    "s_cmp_eq_u32 %[t0], 0\n" \
    "s_cbranch_scc1 Laxx%=\n" \
    "s_cmp_eq_u32 %[t0], 1\n" \
    "s_cbranch_scc1 Laxy%=\n"
#define PKT_ASM_LEAF_WORLD "s_branch Lwork%=\n"
#define PKT_ASM_LEAF_MESH \
    "s_cmp_eq_u32 s53, 0\n" \
    "s_cbranch_scc0 Lwork%=\n"
#define PKT_ASM_TAIL(LEAF) \
    "Lnext%=:\n" \
    "s_add_i32 %[steps], %[steps], 1\n" \
    "Lload%=:\n" \
    "s_lshl_b32 %[t0], %[node], 4\n" \
    "s_load_dwordx4 s[52:55], %[nodes], %[t0]\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "s_cmp_lt_i32 s52, 0\n" \
    "s_cbranch_scc0 Ltop%=\n" \
    "Lleaf%=:\n" \
    LEAF \
    "Lpop%=:\n" \
    "s_mov_b64 exec, %[min]\n" \
    "v_mov_b64 %[tmin], %[tmax]\n" \
    "s_mov_b64 exec, %[ex]\n" \
    "Lpop2%=:\n" \
    "s_cmp_eq_u32 %[sp], 0\n" \
    "s_cbranch_scc1 Ldone%=\n" \
    "s_sub_i32 %[t0], %[sp], 1\n" \
    "s_cmp_ge_i32 %[t0], %[levels]\n" \
    "s_cbranch_scc1 Lslowpop%=\n" \
    "s_mov_b32 %[sp], %[t0]\n" \
    "v_lshl_add_u32 %[addr], %[sp], 9, %[lds]\n" \
    "ds_read_b64 %[p], %[addr]\n" \
    "s_add_i32 m0, %[sp], %[idbase]\n" \
    "s_nop 0\n" \
    "v_readlane_b32 %[node], %[ids], m0\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_cmp_class_f64 %[gt0], %[p], 4\n" \
    "v_mov_b64 %[tmax], %[p]\n" \
    "s_andn2_b64 %[min], %[notdone], %[gt0]\n" \
    "s_cbranch_scc0 Lpop2%=\n" \
    "s_branch Lload%=\n" \
    "Lwork%=:\n" \
    "s_mov_b32 %[flag], 0\n" \
    "s_branch Lend%=\n" \
    "Lslow%=:\n" \
    "s_mov_b32 %[flag], 1\n" \
    "s_branch Lend%=\n" \
    "Ldone%=:\n" \
    "s_mov_b32 %[flag], 2\n" \
    "s_mov_b64 %[min], 0\n" \
    "s_branch Lend%=\n" \
    "Lslowpop%=:\n" \
    "s_mov_b32 %[flag], 3\n" \
    "Lend%=:\n" \
    "s_andn2_b64 exec, %[ex], %[min]\n" \
    "v_mov_b64 %[tmax], %[ninf]\n" \
    "s_mov_b64 exec, %[ex]\n" \
    "s_mov_b32 m0, %[m0save]\n"

// In: node, nd = its record (a branch or a leaf), the lanes with a range `m_in` (tmax == PKT_EMPTY elsewhere), the lanes still looking for a
// hit `m_notdone`, `enter` (see above). Out: the flag; node / nd / m_in / tmin / tmax / sp / ids as the walk left them. `steps`: branch steps.
template <bool WORLD>
__device__ __forceinline__ int packet_descend(const rsx_kdnode *nodes, int32_t &node, UNode &nd, const Ray &r, const AxisDiv &ad, lanemask &m_in, double &tmin, double &tmax,
                                              const Stack &st, IdStack &ids, int32_t &sp, double t_cull, lanemask m_notdone, int enter, int32_t &steps_out) {
    pkt_i4 rec;
    rec.x = nd.type; rec.y = nd.count; rec.z = (int)nd.lo; rec.w = (int)nd.hi;
    const unsigned long long base = (unsigned long long)nodes;
    const unsigned long long ninf = 0xfff0000000000000ULL;
    const uint32_t lds = st.lds_t + (uint32_t)(threadIdx.x % WAVE) * 8u;
    const int levels = st.lds_levels, idbase = ids.base;
    enter = __builtin_amdgcn_readfirstlane(enter);                              // (wave-uniform by construction: say so)
    double num, q, pl, p;
    int addr, flag, lower, nearid, farid, t0, m0save, steps;
    unsigned long long gt0, le, cross, far, want, ex;
#define PKT_ASM_OPERANDS \
                     : [node] "+s"(node), [rec] "+{s[52:55]}"(rec), [min] "+s"(m_in), [tmax] "+v"(tmax), [tmin] "+v"(tmin), [sp] "+s"(sp), [ids] "+v"(ids.v), \
                       [flag] "=&s"(flag), [steps] "=&s"(steps), [num] "=&v"(num), [q] "=&v"(q), [pl] "=&v"(pl), [p] "=&v"(p), [addr] "=&v"(addr), \
                       [gt0] "=&s"(gt0), [le] "=&s"(le), [cross] "=&s"(cross), [far] "=&s"(far), [want] "=&s"(want), [ex] "=&s"(ex), \
                       [lower] "=&s"(lower), [nearid] "=&s"(nearid), [farid] "=&s"(farid), [t0] "=&s"(t0), [m0save] "=&s"(m0save) \
                     : [nodes] "s"(base), [ox] "s"(r.ox), [oy] "s"(r.oy), [oz] "s"(r.oz), [dx] "v"(r.dx), [dy] "v"(r.dy), [dz] "v"(r.dz), \
                       [yx] "v"(ad.yx), [yy] "v"(ad.yy), [yz] "v"(ad.yz), [levels] "s"(levels), [idbase] "s"(idbase), [lds] "v"(lds), [ninf] "s"(ninf), \
                       [tcull] "v"(t_cull), [notdone] "s"(m_notdone), [enter] "s"(enter) \
                     : "vcc", "scc", "memory"
    if constexpr (WORLD) {
        asm volatile(PKT_ASM_HEAD
                     PKT_ASM_STEP("z", "%[oz]", "%[dz]", "%[yz]", PKT_ASM_CULL)
                     PKT_ASM_STEP("x", "%[ox]", "%[dx]", "%[yx]", PKT_ASM_CULL)
                     PKT_ASM_STEP("y", "%[oy]", "%[dy]", "%[yy]", PKT_ASM_CULL)
                     PKT_ASM_TAIL(PKT_ASM_LEAF_WORLD)
                     PKT_ASM_OPERANDS);
    } else {
        asm volatile(PKT_ASM_HEAD
                     PKT_ASM_STEP("z", "%[oz]", "%[dz]", "%[yz]", PKT_ASM_NOCULL)
                     PKT_ASM_STEP("x", "%[ox]", "%[dx]", "%[yx]", PKT_ASM_NOCULL)
                     PKT_ASM_STEP("y", "%[oy]", "%[dy]", "%[yy]", PKT_ASM_NOCULL)
                     PKT_ASM_TAIL(PKT_ASM_LEAF_MESH)
                     PKT_ASM_OPERANDS);
    }
#undef PKT_ASM_OPERANDS
    nd.type = rec.x; nd.count = rec.y; nd.lo = (uint32_t)rec.z; nd.hi = (uint32_t)rec.w;
    steps_out = steps;
    return flag;
}
#endif

// Pops until some lane has a range again. `done` lanes (their ray found its hit) discard theirs. False: the stack is empty.
__device__ __forceinline__ bool packet_pop(const Stack &st, const IdStack &ids, int32_t &sp, int32_t &node, double &tmax, bool done PKT_ARG) {
    while (sp > 0) {
        --sp;
        double t;
        pstack_pop(st, ids, sp, node, t);
        PKT_COUNT(PKC_POPS, 1)
        tmax = done ? PKT_EMPTY : t;
        if (pkt_any(tmax != PKT_EMPTY)) return true;
    }
    return false;
}

// Camera-relative leaf records (k_camera_relative, dev_render.hpp): the rays of a pinhole pass leave ONE point, so stage 1 of the triangle
// test — vertices minus origin, mesh.pyx:633-643 — gives the same nine numbers for every ray of the pass. They are computed once per
// (mesh instance, leaf item) before the pass and the walk reads them instead of the vertices: 27 of the ~75 vector instructions of a
// missed triangle go, the load stays a scalar one. A walk uses them only after it has compared its rays' origin with the one the
// records were made for (RelInfo::o), bit for bit.
struct RelInfo {
    long long offset;          // first record of the instance in DScene::rel (three float4 per leaf item), -1: none
    double o[3];               // the ray origin in the instance's space they are relative to
};

struct RelJob { int32_t prim, mesh; long long offset, n_items; };

// Fills the camera-relative records of one mesh instance per blockIdx.y. The origin is formed exactly as the rays' own: camera_ray's
// expressions for the pinhole in world space (dev_render.hpp), then Point3D.transform into the instance's space (to_local; for an affine
// matrix its w is exactly 1 and x * (1.0 / 1.0) == x, so it agrees with to_local_uniform's shortcut) — and the walk re-checks it anyway.
__global__ void k_camera_relative(DScene sc, rsx_camera cam, const RelJob *jobs, float4 *rel, RelInfo *info) {
    const RelJob job = jobs[blockIdx.y];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double *m = cam.to_root;
    double wq = m[12] * 0.0 + m[13] * 0.0 + m[14] * 0.0 + m[15];
    wq = 1.0 / wq;
    Ray r;
    r.ox = (m[0] * 0.0 + m[1] * 0.0 + m[2] * 0.0 + m[3]) * wq;
    r.oy = (m[4] * 0.0 + m[5] * 0.0 + m[6] * 0.0 + m[7]) * wq;
    r.oz = (m[8] * 0.0 + m[9] * 0.0 + m[10] * 0.0 + m[11]) * wq;
    r.dx = r.dy = 0.0; r.dz = 1.0; r.maxd = 0.0;
    const Ray l = to_local(sc.prims[job.prim], r);
    if (i == 0) { info[job.prim].offset = job.offset; info[job.prim].o[0] = l.ox; info[job.prim].o[1] = l.oy; info[job.prim].o[2] = l.oz; }
    if (i >= job.n_items) return;
    const float4 *rec = sc.meshes[job.mesh].leaf + 4 * (size_t)i;
    const TriVerts t = tri_translate(l.ox, l.oy, l.oz, rec[0], rec[1], rec[2]);
    float4 *dst = rel + 3 * (size_t)(job.offset + i);
    // by component: (x1 x2 x3 id)(y1 y2 y3 id)(z1 z2 z3 id) — the axis permutation of a ray space is then a choice of ROWS
    const float id = rec[3].x;
    dst[0] = make_float4(t.x1, t.x2, t.x3, id); dst[1] = make_float4(t.y1, t.y2, t.y3, id); dst[2] = make_float4(t.z1, t.z2, t.z3, id);
}

// One leaf: `count` camera-relative records (three float4 each, by component: x1 x2 x3 id | y1 y2 y3 id | z1 z2 z3 id) against every ray
// of the wave, in leaf order, strict `<` (mesh.pyx:520-563). SHARED_AXES: the rays also share the axis permutation (ix, iy, iz
// wave-uniform): stage 2 of the test is then WHICH ROWS are loaded — no instruction at all; otherwise every lane selects its own.
template <bool SHARED_AXES>
__device__ __forceinline__ void packet_leaf_scan(const TriRay &q, int ix, int iy, int iz, const float4 *records, int32_t count, double &distance, int32_t &closest,
                                                 float &bu, float &bv, float &bw) {
    const RSX_CONST_AS PodF4 *rec = (const RSX_CONST_AS PodF4 *)(unsigned long long)records;
    const int r0 = SHARED_AXES ? ix : 0, r1 = SHARED_AXES ? iy : 1, r2 = SHARED_AXES ? iz : 2;
    float4 a = load_f4_u(rec + r0), b = load_f4_u(rec + r1), c = load_f4_u(rec + r2);
    for (int32_t k = 0; k < count; ++k) {
        const int32_t kn = k + 1 < count ? k + 1 : k;
        const float4 na = load_f4_u(rec + 3 * kn + r0), nb = load_f4_u(rec + 3 * kn + r1), nc = load_f4_u(rec + 3 * kn + r2);
        TriVerts p;
        if constexpr (SHARED_AXES) { p.x1 = a.x; p.x2 = a.y; p.x3 = a.z; p.y1 = b.x; p.y2 = b.y; p.y3 = b.z; p.z1 = c.x; p.z2 = c.y; p.z3 = c.z; }
        else {
            TriVerts t;
            t.x1 = a.x; t.x2 = a.y; t.x3 = a.z; t.y1 = b.x; t.y2 = b.y; t.y3 = b.z; t.z1 = c.x; t.z2 = c.y; t.z3 = c.z;
            p = tri_permute(q.ix, q.iy, q.iz, t);
        }
        float ht, hu, hv, hw;
        if (tri_finish(q.sx, q.sy, q.sz, q.maxd, p, ht, hu, hv, hw) && (double)ht < distance) {
            distance = (double)ht; closest = __float_as_int(a.w); bu = hu; bv = hv; bw = hw;
        }
        a = na; b = nb; c = nc;
    }
}

// aabb_rcp (dev_common.hpp; BoundingBox3D.intersect, boundingbox.pyx:180-245) for a packet whose rays agree in the sign of every direction
// component, none of them zero: slab_rcp's `d != 0` / `d > 0` arms are resolved once for the wave (neg_* scalar), the rest is the same
// operations in the same order. `uniform` false (wave-uniform): the general form.
__device__ __forceinline__ bool aabb_rcp_signed(bool uniform, bool neg_x, bool neg_y, bool neg_z, const double *lo, const double *hi, const Ray &r,
                                                double rx, double ry, double rz, double &front, double &back) {
    if (!uniform) return aabb_rcp(lo, hi, r, rx, ry, rz, front, back);
    front = -INFINITY;
    back = INFINITY;
    double a, b;
    a = ((neg_x ? hi[0] : lo[0]) - r.ox) * rx; b = ((neg_x ? lo[0] : hi[0]) - r.ox) * rx;
    if (a > front) front = a;
    if (b < back) back = b;
    a = ((neg_y ? hi[1] : lo[1]) - r.oy) * ry; b = ((neg_y ? lo[1] : hi[1]) - r.oy) * ry;
    if (a > front) front = a;
    if (b < back) back = b;
    a = ((neg_z ? hi[2] : lo[2]) - r.oz) * rz; b = ((neg_z ? lo[2] : hi[2]) - r.oz) * rz;
    if (a > front) front = a;
    if (b < back) back = b;
    if (front > back) return false;
    if (front < 0.0 && back < 0.0) return false;
    return true;
}

// MeshData.trace (mesh.pyx:506-563) for the packet: `m` and the ray space are wave-uniform, `want` = the lane's ray passed the
// BoundPrimitive gate. Leaves of any size are walked the same way: every record comes in once over the scalar data path (the next one
// while this one is tested) and every lane with a range tests it — in leaf order, strict `<`: the reference's own loop.
// (Measured and not adopted — the mesh walk as a real call, `noinline`, so that the world walk's state would sit in callee-saved
// registers across a visit instead of being spilled piecemeal: configs[2] 27.4 -> 33.3 ms. The call's own traffic — arguments through
// vector registers and memory, uniform values re-established with readfirstlane, the callee's saves — cost more than the spills.)
__device__ __forceinline__ bool mesh_trace_packet(PScene sc, int32_t prim, bool want, UMesh m, const Ray &r, const Stack &st, IdStack &ids, MeshHit &out, int32_t &work PKT_ARG) {
    const rsx_kdnode *nodes = m->nodes;
    const float4 *leaf = m->leaf;
    const AxisDiv ad = axis_div(r);
    PacketSpace ps;
    double tmin = 0, tmax = 0;
    {
        const double lo[3] = {m->lower[0], m->lower[1], m->lower[2]}, hi[3] = {m->upper[0], m->upper[1], m->upper[2]};
        const double rx = exact_div(1.0, r.dx, ad.yx, ad.safe & 1), ry = exact_div(1.0, r.dy, ad.yy, (ad.safe >> 1) & 1),
                     rz = exact_div(1.0, r.dz, ad.yz, (ad.safe >> 2) & 1);
        ps = packet_space(r, ad, lo, hi, want, m->splits_bounded);
        const bool uniform = (ps.neg[0] == 0ULL || ps.neg[0] == ~0ULL) && (ps.neg[1] == 0ULL || ps.neg[1] == ~0ULL) && (ps.neg[2] == 0ULL || ps.neg[2] == ~0ULL) &&
                             !pkt_any(r.dx == 0.0 || r.dy == 0.0 || r.dz == 0.0);
        if (!(want && aabb_rcp_signed(uniform, ps.neg[0] != 0ULL, ps.neg[1] != 0ULL, ps.neg[2] != 0ULL, lo, hi, r, rx, ry, rz, tmin, tmax))) tmax = PKT_EMPTY;       // kdtree3d.pyx:589-607
    }
    if (!pkt_any(tmax != PKT_EMPTY)) return false;
    PKT_COUNT(PKC_MVISITS, 1)
    const TriRay q = tri_ray(r);
    // camera-relative records of this instance, if they were made for exactly these rays' origin; and do the rays share their dominant
    // axis and winding (rays through one pixel nearly always do)?
    bool relative = false, shared_axes = false;
    const float4 *rel = nullptr;
    int ax = 0, ay = 0, az = 0;
    if (sc->rel_info) {
        const RSX_CONST_AS RelInfo *ri = (const RSX_CONST_AS RelInfo *)(unsigned long long)(sc->rel_info + prim);
        const long long off = ri->offset;
        if (off >= 0) {
            const double fx = ri->o[0], fy = ri->o[1], fz = ri->o[2];
            relative = !pkt_any(want && (r.ox != fx || r.oy != fy || r.oz != fz));
            rel = sc->rel + 3 * (size_t)off;
            const int first = __ffsll((long long)__builtin_amdgcn_ballot_w64(want)) - 1;
            const int axes = q.ix | (q.iy << 2) | (q.iz << 4);
            const int faxes = __builtin_amdgcn_readlane(axes, first);
            shared_axes = !pkt_any(want && axes != faxes);
            ax = faxes & 3; ay = (faxes >> 2) & 3; az = (faxes >> 4) & 3;
        }
    }
    bool hit = false;
    int32_t node = 0, sp = 0;
    lanemask m_have = pkt_mask(tmax != PKT_EMPTY);                             // (the steps and the pops hand it on)
    UNode nd = load_node_u(nodes, node);
    int enter = 0;                                                             // 1: the leaf the walk is at has been dealt with
#if RSX_PKT_ASM
    const bool by_hand = ps.fast == 7;                                         // (wave-uniform)
#else
    const bool by_hand = false;
#endif
    for (;;) {
        // to the next leaf with triangles (packet_descend's contract; the compiled form of the same transitions serves ray spaces whose
        // quotients need the per-node range test)
        int flag;
#ifdef RSX_ASM_MARKS
        asm volatile("; MARK mesh steps begin");
#endif
#if RSX_PKT_ASM
        if (__builtin_expect(by_hand, 1)) {
            int32_t steps;
            flag = packet_descend<false>(nodes, node, nd, r, ad, m_have, tmin, tmax, st, ids, sp, 0.0, ~pkt_mask(hit), enter, steps);
            work += steps;
            PKT_COUNT(PKC_MSTEPS, steps)
        } else
#endif
        {
            if (enter || (nd.type < 0 && nd.count <= 0)) { if (tmax != PKT_EMPTY) tmin = tmax; flag = 3; }   // the next range of a ray begins where this one ended
            else flag = nd.type >= 0 ? 1 : 0;
        }
#ifdef RSX_ASM_MARKS
        asm volatile("; MARK mesh steps end");
#endif
        enter = 0;
        if (flag == 1) {
            node = packet_step<false>(nd, node, r, ad, ps, tmin, tmax, st, ids, sp, 0.0, m_have PKT_PASS);
            nd = load_node_u(nodes, node);
            work += 1;
            PKT_COUNT(PKC_MSTEPS, 1)
            continue;
        }
        if (flag == 3) {
            if (!packet_pop(st, ids, sp, node, tmax, hit PKT_PASS)) break;
            m_have = pkt_mask(tmax != PKT_EMPTY);
            nd = load_node_u(nodes, node);
            continue;
        }
        if (flag == 2) break;
#ifdef PKT_ABLATE_TRIS
        const int32_t count = 0;                                               // (timing ablation: results are wrong)
#else
        const int32_t count = nd.count;
#endif
        if (count > 0) {                                                       // _trace_leaf, mesh.pyx:520-563
            double distance = r.maxd < tmax ? r.maxd : tmax;                   // (no range: -inf, nothing is accepted)
            int32_t closest = -1;
            float bu = 0, bv = 0, bw = 0;
            work += count;
            PKT_COUNT(PKC_MLEAVES, 1)
            PKT_COUNT(PKC_TRIS, count)
            if (relative) {
                const float4 *records = rel + 3 * (size_t)nd.lo;
                if (shared_axes) packet_leaf_scan<true>(q, ax, ay, az, records, count, distance, closest, bu, bv, bw);
                else packet_leaf_scan<false>(q, 0, 0, 0, records, count, distance, closest, bu, bv, bw);
            } else {
                // (no camera-relative records for these rays: the whole test per ray)
                const RSX_CONST_AS PodF4 *rec = (const RSX_CONST_AS PodF4 *)(unsigned long long)(leaf + 4 * (size_t)nd.lo);
                float4 a = load_f4_u(rec), b = load_f4_u(rec + 1), c = load_f4_u(rec + 2);
                int32_t tri = __float_as_int(rec[3].x);
                for (int32_t k = 0; k < count; ++k) {
                    const int32_t kn = k + 1 < count ? k + 1 : k;
                    const float4 na = load_f4_u(rec + 4 * kn), nb = load_f4_u(rec + 4 * kn + 1), nc = load_f4_u(rec + 4 * kn + 2);
                    const int32_t ntri = __float_as_int(rec[4 * kn + 3].x);
                    float ht, hu, hv, hw;
                    if (tri_test(q, a, b, c, ht, hu, hv, hw) && (double)ht < distance) { distance = (double)ht; closest = tri; bu = hu; bv = hv; bw = hw; }
                    a = na; b = nb; c = nc; tri = ntri;
                }
            }
            if (closest >= 0) { out.u = bu; out.v = bv; out.w = bw; out.t = (float)distance; out.tri = closest; hit = true; }
        }
        enter = 1;
    }
    return hit;
}

// First root of a PLAIN box (DScene::wide_plain: no transform, so the ray in the box's space IS the world ray) for a packet whose rays
// agree in the sign of every direction component, none of them zero — the rays of one pixel, nearly always. With uniform signs which
// bound of a slab is the near one is a scalar fact: box_slab_rcp's / slab_rcp's `d != 0`, `d > 0` arms are resolved once for the wave
// and both the BoundPrimitive gate (boundingbox.pyx:180-245, on the padded bounding box blo / bhi) and Box.hit (box.pyx:149-200, on the
// box's own extents lo / hi) become straight lines of the same operations in the same order — the products (bound - origin) *
// (1.0 / direction), merged with the reference's strict comparisons in axis order (the first axis wins a tie).
// Out as analytic_first_root: t = -1, faces = 0 when there is no root.
__device__ __forceinline__ void plain_box_first_root(const double (&blo)[3], const double (&bhi)[3], const double (&lo)[3], const double (&hi)[3], bool want,
                                                     const Ray &r, double rx, double ry, double rz, bool neg_x, bool neg_y, bool neg_z, double &t, int32_t &faces) {
    t = -1.0; faces = 0;
    {   // the gate: aabb_rcp on the bounding box
        double front = -INFINITY, back = INFINITY, a, b;
        a = ((neg_x ? bhi[0] : blo[0]) - r.ox) * rx; b = ((neg_x ? blo[0] : bhi[0]) - r.ox) * rx;
        if (a > front) front = a;
        if (b < back) back = b;
        a = ((neg_y ? bhi[1] : blo[1]) - r.oy) * ry; b = ((neg_y ? blo[1] : bhi[1]) - r.oy) * ry;
        if (a > front) front = a;
        if (b < back) back = b;
        a = ((neg_z ? bhi[2] : blo[2]) - r.oz) * rz; b = ((neg_z ? blo[2] : bhi[2]) - r.oz) * rz;
        if (a > front) front = a;
        if (b < back) back = b;
        want = want && !(front > back) && !(front < 0.0 && back < 0.0);
    }
    // Box.hit: faces packed as analytic_first_root returns them, (face + 1) | (axis + 1) << 4
    const double nx = ((neg_x ? hi[0] : lo[0]) - r.ox) * rx, fx = ((neg_x ? lo[0] : hi[0]) - r.ox) * rx;
    const double ny = ((neg_y ? hi[1] : lo[1]) - r.oy) * ry, fy = ((neg_y ? lo[1] : hi[1]) - r.oy) * ry;
    const double nz = ((neg_z ? hi[2] : lo[2]) - r.oz) * rz, fz = ((neg_z ? lo[2] : hi[2]) - r.oz) * rz;
    const int32_t near_x = ((neg_x ? UPPER_FACE : LOWER_FACE) + 1) | (1 << 4), far_x = ((neg_x ? LOWER_FACE : UPPER_FACE) + 1) | (1 << 4);
    const int32_t near_y = ((neg_y ? UPPER_FACE : LOWER_FACE) + 1) | (2 << 4), far_y = ((neg_y ? LOWER_FACE : UPPER_FACE) + 1) | (2 << 4);
    const int32_t near_z = ((neg_z ? UPPER_FACE : LOWER_FACE) + 1) | (3 << 4), far_z = ((neg_z ? LOWER_FACE : UPPER_FACE) + 1) | (3 << 4);
    double near_t = -INFINITY, far_t = INFINITY;
    int32_t nf = 0, ff = 0;                                // (NO_FACE, axis -1)
    if (nx > near_t) { near_t = nx; nf = near_x; }
    if (fx < far_t) { far_t = fx; ff = far_x; }
    if (ny > near_t) { near_t = ny; nf = near_y; }
    if (fy < far_t) { far_t = fy; ff = far_y; }
    if (nz > near_t) { near_t = nz; nf = near_z; }
    if (fz < far_t) { far_t = fz; ff = far_z; }
    // pick_roots with the ray's own reach
    if (want && !(near_t > far_t) && !(near_t > r.maxd || far_t < 0.0)) {
        if (near_t >= 0.0) { t = near_t; faces = nf; }
        else if (far_t <= r.maxd) { t = far_t; faces = ff; }
    }
}

// World.hit for the packet (kdtree.pyx:73-122, boundprimitive.pyx:42-51): world_trace_wave<false, false, 1, true> with the walk above.
// Leaf items are wave-uniform by construction (the wave is in ONE leaf); wide primitives, leaf tags and the cull as there.
// CSG: the scene has CSG solids (k_render_trace<true, 1, ..., PACKET>): a solid in the state-free evaluator's form is answered for the whole
// wave by csg_fast_hit_uniform; a lane it cannot answer — or any lane that meets a solid without that form — raises `needs_merge` and is
// traced again by the redo pass, as in the per-lane kernel.
template <bool CSG = false>
__device__ __forceinline__ bool world_trace_packet(bool valid, PScene sc, const Ray &r, const Stack &st, const Stack &mesh_stack, Hit &best, uint32_t &work_out,
                                                   bool &needs_merge PKT_ARG) {
    int32_t work = 0;                  // the unit's cost (traversal rounds): wave-uniform, counted on the scalar unit
    best.prim = -1;
    needs_merge = false;
    // the evaluator's rows lie behind the wave's stacks (packet_lds_bytes); the last solid evaluated and every lane's answer are kept
    // across leaves (see world_trace_wave)
    Stack ev;
    ev.stage = nullptr; ev.gt = nullptr; ev.gid = nullptr;
    ev.lds_levels = CSG ? sc->csg_fast_rows : 0;
    ev.lds_t = (uint32_t)(((st.lds_id + (uint32_t)(sc->wdepth + sc->mdepth) * 4u) + 15u) & ~15u);
    ev.lds_id = ev.lds_t + (uint32_t)ev.lds_levels * WAVE * 8u;
    int32_t last_csg = -1;
    double last_t = 0.0;
    int32_t last_leaf = 0;
    uint32_t last_meta = 0;
    double tmin = 0, tmax = 0;
    // 1.0 / d per axis (BoundingBox3D.intersect, boundingbox.pyx:180-245) and the branch steps' quotients both come from the refined
    // reciprocals: exact_div(1, d) and exact_div(split - o, d) are the correctly rounded quotients (dev_common.hpp)
    AxisDiv ad = axis_div(r);
    // (1.0 / d is two instructions away from the refined reciprocal: it is formed again wherever a box is tested — RCP3 — instead of
    // living in six registers through the mesh walks; the asm keeps the compiler from merging the copies back into one long-lived value)
    const bool all_safe = !pkt_any(ad.safe != 7);
#define RCP3 \
    double rcp_one = 1.0; \
    asm volatile("" : "+s"(rcp_one)); \
    double rx = __builtin_fma(__builtin_fma(-r.dx, ad.yx, rcp_one), ad.yx, ad.yx), \
           ry = __builtin_fma(__builtin_fma(-r.dy, ad.yy, rcp_one), ad.yy, ad.yy), \
           rz = __builtin_fma(__builtin_fma(-r.dz, ad.yz, rcp_one), ad.yz, ad.yz); \
    if (__builtin_expect(!all_safe, 0)) {   /* (a real branch, its operands laundered: as a select the three divisions were hoisted to the \
                                               top of the unit and their quotients AND partial results sat in twelve registers through the walk) */ \
        double ux = r.dx, uy = r.dy, uz = r.dz; \
        asm volatile("" : "+v"(ux), "+v"(uy), "+v"(uz)); \
        rx = 1.0 / ux; ry = 1.0 / uy; rz = 1.0 / uz; \
    }
    bool enters;
    const double wlo[3] = {sc->wlower[0], sc->wlower[1], sc->wlower[2]}, whi[3] = {sc->wupper[0], sc->wupper[1], sc->wupper[2]};
    // (signs of the direction components: wave-uniform and non-zero for the rays of a pixel, except where a pixel straddles an axis)
    const lanemask m_negx = pkt_mask(r.dx < 0.0), m_negy = pkt_mask(r.dy < 0.0), m_negz = pkt_mask(r.dz < 0.0);
    const bool signs_uniform = (m_negx == 0ULL || m_negx == ~0ULL) && (m_negy == 0ULL || m_negy == ~0ULL) && (m_negz == 0ULL || m_negz == ~0ULL) &&
                               !pkt_any(r.dx == 0.0 || r.dy == 0.0 || r.dz == 0.0);
    const bool neg_x = m_negx != 0ULL, neg_y = m_negy != 0ULL, neg_z = m_negz != 0ULL;
    { RCP3 enters = valid && aabb_rcp_signed(signs_uniform, neg_x, neg_y, neg_z, wlo, whi, r, rx, ry, rz, tmin, tmax); }
    if (!enters) tmax = PKT_EMPTY;
    if (!pkt_any(enters)) { work_out = 0; return false; }
    const rsx_kdnode *wnodes = sc->wnodes;
    WideSet8 wide;
#pragma unroll
    for (int j = 0; j < 8; ++j) wide.t[j] = -1.0;
    wide.faces[0] = wide.faces[1] = 0;
#ifdef RSX_ASM_MARKS
    asm volatile("; MARK wide begin");
#endif
#ifndef PKT_ABLATE_WIDE                                  // (timing ablation: results are wrong)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (sc->wide[j] >= 0) {
            int32_t f = 0;
            RCP3
            if (signs_uniform && ((sc->wide_plain >> j) & 1)) {
                const UPrim wp = uniform_prim(sc->prims_uniform, sc->wide[j]);
                const double blo[3] = {wp->box_lower[0], wp->box_lower[1], wp->box_lower[2]}, bhi[3] = {wp->box_upper[0], wp->box_upper[1], wp->box_upper[2]};
                const double plo[3] = {wp->params[0], wp->params[1], wp->params[2]}, phi[3] = {wp->params[3], wp->params[4], wp->params[5]};
                plain_box_first_root(blo, bhi, plo, phi, enters, r, rx, ry, rz, neg_x, neg_y, neg_z, wide.t[j], f);
            } else
            analytic_first_root(sc->prims, uniform_prim(sc->prims_uniform, sc->wide[j]), sc->wide[j], enters, r, rx, ry, rz, wide.t[j], f);
            wide.faces[0] |= (uint32_t)f << (8 * j);
        }
    }
#endif
#ifdef RSX_ASM_MARKS
    asm volatile("; MARK wide end");
#endif
    double t_cull = INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j) if (wide.t[j] >= 0.0 && wide.t[j] < t_cull) t_cull = wide.t[j];
#if RSX_WORLD_CULL == 0
    t_cull = -INFINITY;
#endif
#if RSX_PKT_CLUSTERS
    if constexpr (!CSG) {
    // The short cut. The two wide slots are answered; every other world primitive lies inside one of up to four cluster boxes
    // (DScene::cluster_lo / hi: unions of the primitives' own bounding boxes). A ray that enters none of them fails the BoundPrimitive gate
    // of every such primitive — the gate's slab products are monotone in the bounds, so a box inside a box the ray misses is missed — and
    // its walk would meet the two answers only: kdtree.pyx:99-116 accepts wide answer t_j in the first leaf on the ray that lists j and
    // whose range reaches t_j (`t_j <= min(max_distance, tmax)`; the last leaf's tmax is the world box's `back`), the nearer answer first:
    // the point at t_j lies in j's bounding box with BOX_PADDING = 1e-9 to spare on every side (box.pyx:37, sphere.pyx:38 ...), so the leaf
    // whose range holds t_j overlaps that box and lists j — rounding moves positions by ~1e-12 at the coordinates this is enabled for
    // (|world bounds| <= 1e4, rsx_scene_create) — and a leaf that holds the farther answer but not the nearer one lies behind a leaf that
    // held the nearer. When NO ray of the unit enters a cluster box the unit is finished here: the nearer eligible answer per ray. Equal
    // answers (the leaf's item order would decide) send the unit through the walk. configs[2]: 0.4 - 0.6 of the units.
    // (scenes without CSG solids only: the CSG packet kernel sits at 212 registers / two waves, and in such scenes the solids' boxes are most of the view)
    if (!CSG && sc->pkt_clusters >= 0) {
        bool near_something = false;
        RCP3
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < sc->pkt_clusters) {
                const double lo[3] = {sc->cluster_lo[c][0], sc->cluster_lo[c][1], sc->cluster_lo[c][2]}, hi[3] = {sc->cluster_hi[c][0], sc->cluster_hi[c][1], sc->cluster_hi[c][2]};
                double f, b;
                const bool in_cluster = enters && aabb_rcp_signed(signs_uniform, neg_x, neg_y, neg_z, lo, hi, r, rx, ry, rz, f, b);
#if RSX_PKT_CLUSTERS >= 2
                // (second level: the gates themselves, for the lanes inside a cluster of at most four primitives)
                const int n_members = sc->cluster_members[c];
                if (n_members == 0) near_something = near_something || in_cluster;
                else if (pkt_any(in_cluster)) {
                    for (int m = 0; m < n_members; ++m) {
                        const double mlo[3] = {sc->member_lo[c][m][0], sc->member_lo[c][m][1], sc->member_lo[c][m][2]}, mhi[3] = {sc->member_hi[c][m][0], sc->member_hi[c][m][1], sc->member_hi[c][m][2]};
                        near_something = near_something || (in_cluster && aabb_rcp_signed(signs_uniform, neg_x, neg_y, neg_z, mlo, mhi, r, rx, ry, rz, f, b));
                    }
                }
#else
                near_something = near_something || in_cluster;
#endif
            }
        }
        if (!pkt_any(near_something)) {
            const double reach = r.maxd < tmax ? r.maxd : tmax;                // (no range: -inf)
            const bool c0 = wide.t[0] >= 0.0 && wide.t[0] <= reach, c1 = wide.t[1] >= 0.0 && wide.t[1] <= reach;
            if (!pkt_any(c0 && c1 && wide.t[0] == wide.t[1])) {
                const bool second = c1 && (!c0 || wide.t[1] < wide.t[0]);
                if (c0 || c1) {
                    const int32_t faces = (int32_t)((wide.faces[0] >> (second ? 8 : 0)) & 255u);
                    best.prim = second ? sc->wide[1] : sc->wide[0]; best.t = second ? wide.t[1] : wide.t[0];
                    best.a0 = (faces & 15) - 1; best.a1 = (faces >> 4) - 1;
                    best.u = best.v = best.w = 0.0f;
                }
                work_out = 1;
                return best.prim >= 0;
            }
        }
    }
    }
#endif
    const PacketSpace ps = packet_space(r, ad, wlo, whi, valid, sc->wsplits_bounded);      // (for the walk only: behind the short cut)
    int32_t node = 0, sp = 0;
    IdStack ids;
    ids.v = 0; ids.base = 0;                               // (world levels: lanes 0 .. wdepth - 1, a mesh walk's behind them)
    lanemask m_have = pkt_mask(tmax != PKT_EMPTY);
    UNode nd = load_node_u(wnodes, node);
    int enter = 0;
#if RSX_PKT_ASM
    const bool by_hand = ps.fast == 7;
#else
    const bool by_hand = false;
#endif
    for (;;) {
        int flag;                                          // (packet_descend's contract; every world leaf has work)
#if RSX_PKT_ASM
        if (__builtin_expect(by_hand, 1)) {
            int32_t steps;
            flag = packet_descend<true>(wnodes, node, nd, r, ad, m_have, tmin, tmax, st, ids, sp, t_cull, ~pkt_mask(best.prim >= 0), enter, steps);
            work += steps;
            PKT_COUNT(PKC_WSTEPS, steps)
        } else
#endif
        {
            if (enter) { if (tmax != PKT_EMPTY) tmin = tmax; flag = 3; }
            else flag = nd.type >= 0 ? 1 : 0;
        }
        enter = 0;
        if (flag == 1) {
            node = packet_step<true>(nd, node, r, ad, ps, tmin, tmax, st, ids, sp, t_cull, m_have PKT_PASS);
            nd = load_node_u(wnodes, node);
            work += 1;
            PKT_COUNT(PKC_WSTEPS, 1)
            continue;
        }
        if (flag == 3) {
            work = __builtin_amdgcn_readfirstlane(work);
            if (!packet_pop(st, ids, sp, node, tmax, best.prim >= 0 PKT_PASS)) break;
            m_have = pkt_mask(tmax != PKT_EMPTY);
            nd = load_node_u(wnodes, node);
            continue;
        }
        if (flag == 2) break;
        double distance = r.maxd < tmax ? r.maxd : tmax;                       // (no range: -inf, `t <= distance` fails)
        PKT_COUNT(PKC_WLEAVES, 1)
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
                        best.prim = slot == 1 ? sc->wide[1] : sc->wide[0]; best.t = t; best.a0 = (faces & 15) - 1; best.a1 = (faces >> 4) - 1;
                        best.u = best.v = best.w = 0.0f;
                    }
                }
            }
        } else {
            const int32_t count = nd.count;
            const RSX_CONST_AS int32_t *items = (const RSX_CONST_AS int32_t *)(unsigned long long)(sc->witems + nd.lo);
            for (int32_t k = 0; k < count; ++k) {
                const int32_t idx = items[k];                                  // (scalar load)
                work += 4;
                PKT_COUNT(PKC_WITEMS, 1)
                Hit cand;
                cand.prim = -1;
                const bool in = tmax != PKT_EMPTY;
                if (idx == sc->wide[0] || idx == sc->wide[1]) {                  // answered before the traversal began
                    double t;
                    int32_t faces;
                    wide_lookup<2>(wide, idx == sc->wide[0] ? 0 : 1, t, faces);
                    if (t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
                } else {
                    const UPrim up = uniform_prim(sc->prims_uniform, idx);
                    const int32_t type = up->type;
                    RCP3
                    if (type == RSX_PRIM_MESH) {
                        const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
                        double f, b;
                        const bool gate = in && aabb_rcp_signed(signs_uniform, neg_x, neg_y, neg_z, lo, hi, r, rx, ry, rz, f, b);      // BoundPrimitive.hit gate (boundprimitive.pyx:42-51)
#ifdef PKT_ABLATE_MESH
                        if (false) {                                           // (timing ablation: results are wrong)
#else
                        if (pkt_any(gate)) {
#endif
                            Ray l = to_local_uniform(up, r);           // (every lane: the lanes of one space then share one origin ...
                            l.ox = readlane_f64(l.ox, 0); l.oy = readlane_f64(l.oy, 0); l.oz = readlane_f64(l.oz, 0);   // ... which then lives in scalar registers)
                            const UMesh um = (UMesh)(unsigned long long)(sc->meshes + up->mesh);
                            MeshHit mh;
                            ids.base = sc->wdepth;                             // (the mesh walk's ids lie behind the world's in the same register)
                            if (mesh_trace_packet(sc, idx, gate, um, l, mesh_stack, ids, mh, work PKT_PASS)) {
                                cand.prim = idx; cand.t = (double)mh.t; cand.a0 = mh.tri; cand.a1 = 0; cand.u = mh.u; cand.v = mh.v; cand.w = mh.w;
                            }
                            ids.base = 0;
                        }
                    } else if (CSG && is_csg(type)) {
                        const double lo[3] = {up->box_lower[0], up->box_lower[1], up->box_lower[2]}, hi[3] = {up->box_upper[0], up->box_upper[1], up->box_upper[2]};
                        double f, b;
                        const bool gate = in && aabb_rcp_signed(signs_uniform, neg_x, neg_y, neg_z, lo, hi, r, rx, ry, rz, f, b);      // BoundPrimitive.hit gate (boundprimitive.pyx:42-51)
                        if (pkt_any(gate)) {
                            const CsgFast *table = sc->csgfast;
                            const RSX_CONST_AS CsgFast *flat = table ? (const RSX_CONST_AS CsgFast *)(unsigned long long)(table + idx) : nullptr;
                            if (flat != nullptr && flat->n_leaves > 0 && ev.lds_levels >= 2 * flat->n_leaves) {
                                if (idx != last_csg) { last_csg = idx; last_meta = 0; }
                                const bool ask = gate && (last_meta >> 28) == 0u;
                                if (pkt_any(ask)) {
                                    Hit found;
                                    found.prim = -1; found.t = 0; found.a0 = found.a1 = 0; found.leaf = 0; found.flags = 0;
                                    const int fast = csg_fast_hit_uniform(table, sc->prims_uniform, sc->prims, idx, ask, r, ev, found);
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
                                if (gate && state == 3u) needs_merge = true;
                            } else if (gate) needs_merge = true;               // a solid the evaluator does not serve: the redo pass
                        }
                    } else {                                                   // sphere / box / cylinder: Primitive.hit, first root
                        double t;
                        int32_t faces;
                        analytic_first_root(sc->prims, up, idx, in, r, rx, ry, rz, t, faces);
                        if (t >= 0.0) { cand.prim = idx; cand.t = t; cand.a0 = (faces & 15) - 1; cand.a1 = (faces >> 4) - 1; cand.u = cand.v = cand.w = 0.0f; }
                    }
                }
                if (cand.prim >= 0 && cand.t <= distance) { distance = cand.t; best = cand; }   // `<=`: later item wins ties (kdtree.pyx:113)
            }
        }
        work = __builtin_amdgcn_readfirstlane(work);
        enter = 1;
    }
    work_out = (uint32_t)work;
    return best.prim >= 0;
#undef RCP3
}
