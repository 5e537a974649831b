// dev_wavefront.hpp — part of librsx's single device translation unit (included by rsx_device.hip, after dev_render.hpp).
// Path-traced passes level by level (round 5): the paths of a pass live in a slot-stable array (two cache lines per path) and move from
// one launch of k_wf_level to the next through lists of slot numbers, one list per material arm. A launch takes the paths that wait for
// an arm — 64 that all need the SAME arm per wave — runs the arm (Material.evaluate_surface, material.pxd:36-47: hit geometry, the volumes
// around the segment's origin, Lambert / Dielectric / null surface, the daughter's roulette and the term it leaves), walks the
// daughter ray's segment (Ray.trace's world.hit, ray.pyx:391) and files the path by the material that segment met. Same operations per
// path as k_render_trace_path, in the same order — random numbers, sample record and term list are keyed by (pixel, sample), so which
// lane of which launch renders a segment never shows in the result. What changes is who waits for whom: in the one-kernel form a
// wave ran every arm one after the other for whichever of its lanes needed it (0.42 of the issued vector lanes did work on the
// Cornell box); here an arm sees full waves.
//
// Lists without contended atomics: a list is WF_SUB sub-lists with a counter each; a wave appends to the sub-list of its own number
// modulo WF_SUB (one atomic per wave, chunk and arm, spread over WF_SUB addresses per arm — the first form of this file kept one
// counter per arm and spent two thirds of its time queueing on four addresses: ~17 ns per returned atomic and address, 10^6 of them per
// level), and the next launch maps its chunk numbers onto the (arm, sub-list) segments through a prefix table in LDS.
#pragma once

#define WF_KEYS 5
// (two lists for Lambert surfaces in a world with important primitives: the paths whose next scattering draw takes the important-path
// branch of ContinuousBSDF.evaluate_surface, material.pyx:327-352 — a cone sample of ~600 instructions — and the paths that take the cosine
// lobe: the draw is a function of (pixel, sample, depth), known when the path is filed. Both lists run the same arm.)
enum { WF_NULL = 0, WF_LAMBERT = 1, WF_DIELECTRIC = 2, WF_END = 3, WF_LAMBERT_IMPORTANT = 4 };
#ifndef WF_SUB
#define WF_SUB 64
#endif
#define WF_SEGS (WF_KEYS * WF_SUB)
#ifndef RSX_WF_MIN_WAVES
#define RSX_WF_MIN_WAVES 2
#endif

// A path between two levels. Line 1: the ray of the segment just walked and where the path's term list stands; line 2: what the walk
// found (CSG scenes keep the full Hit record in an array of its own) and what the path's random numbers and sample record are keyed by.
struct WfPath {
    double ox, oy, oz, dx, dy, dz;     // Ray (max_distance is always infinite on this path: ray.pyx:506-534 spawns daughters without one)
    int32_t blk, pos, depth, segments;
    double t;                          // Hit, without the CSG fields
    int32_t prim, a0, a1;
    float u, v, w;
    int32_t record;
    uint32_t rng_pixel_lo;
    uint64_t rng_sample;
    double weight;                     // the sample record's projection weight
    uint32_t path_spawned, pad;
};
static_assert(sizeof(WfPath) == 128 && offsetof(WfPath, t) == 64, "two cache lines per path");

struct WfStore {
    WfPath *paths;                     // [n]
    Hit *csg_hits;                     // [n] CSG scenes: the whole Hit record of the path's current segment
    const uint32_t *list_in;           // [WF_SEGS][sub_stride] slots filed by the previous level, by arm and sub-list
    uint32_t *list_out;                // ... by this level
    const uint32_t *cnt_in;            // [WF_SEGS] lengths of list_in's sub-lists; null at level 0 (every slot of the chunk starts a path)
    uint32_t *cnt_out;                 // [WF_SEGS] zero when the launch starts
    long long n;                       // slots of this chunk = 64 x its units
    long long first_unit;              // the chunk's first 64-ray unit (natural order: unit_pixel)
    uint32_t sub_stride, pad;
};

__device__ __forceinline__ void wf_stage_scene(const DScene &sc_arg, DScene &sc, const RSX_CONST_AS RenderParams *q, bool staged, bool csg) {
    if (staged || q->world_lds > 0) {                      // as k_render_trace_path: the world tree behind the traversal stacks, then the primitive records
        int4 *dst = reinterpret_cast<int4 *>(smem + q->world_lds);
        const int4 *src = reinterpret_cast<const int4 *>(sc_arg.wnodes_scatter);
        for (int i = threadIdx.x; i < sc_arg.n_wnodes; i += blockDim.x) dst[i] = src[i];
        int32_t *idst = reinterpret_cast<int32_t *>(dst + sc_arg.n_wnodes);
        for (int i = threadIdx.x; i < sc_arg.n_witems; i += blockDim.x) idst[i] = sc_arg.witems[i];
        sc.wnodes = sc.wnodes_scatter = reinterpret_cast<const rsx_kdnode *>(dst);
        sc.witems = idst;
        if (staged) {
            long long *pdst = reinterpret_cast<long long *>(smem + q->prims_lds);
            const long long *psrc = reinterpret_cast<const long long *>(sc_arg.prims);
            const int n8 = sc_arg.n_prims * (int)(sizeof(rsx_primitive) / 8);
            for (int i = threadIdx.x; i < n8; i += blockDim.x) pdst[i] = psrc[i];
            sc.prims = reinterpret_cast<const rsx_primitive *>(pdst);
            if (csg && sc_arg.csgfast) {
                long long *fdst = pdst + n8;
                const long long *fsrc = reinterpret_cast<const long long *>(sc_arg.csgfast);
                const int f8 = sc_arg.n_prims * (int)(sizeof(CsgFast) / 8);
                for (int i = threadIdx.x; i < f8; i += blockDim.x) fdst[i] = fsrc[i];
                sc.csgfast = reinterpret_cast<const CsgFast *>(fdst);
            }
        }
    }
}

// a path is over: its sample record is complete (as the last lines of k_render_trace_path's round)
__device__ __forceinline__ void wf_finish(Sample *samples, const PathStore &ps, int32_t record, int32_t blk, int32_t pos, double weight, double end_a, int32_t end_table) {
    Sample smp;
    smp.a = end_a; smp.weight = weight; smp.table = end_table; smp.pad = pos;
    samples[record] = smp;
    ps.tail[record] = blk;
}

// The material arm of one path segment for the lanes of a wave that all wait for the same arm (`arm`: WF_NULL / WF_LAMBERT /
// WF_DIELECTRIC / WF_END): k_render_trace_path's material stage — hit geometry, the volumes around the segment's origin, the surface,
// the daughter's roulette and the term it leaves — on the path registers `p`. On return `active` says whether the path goes on (p.r is
// then the daughter ray); a path that ended has its sample record written.
struct WfRegs {
    Ray r;
    int32_t blk, record;
    int pos, depth, segments;
    uint32_t rng_pixel_lo;
    uint64_t rng_sample;
    unsigned int path_spawned;
    double weight;
};
template <bool CSG, int MODE, bool VOLS, bool MESHES = true>
__device__ __forceinline__ void wf_arm(const DScene &sc, const RSX_CONST_AS RenderParams *q, const PathStore &ps, Sample *samples, const Stack &ms, volatile uint32_t *arena_wave,
                                       int arm, const Hit &hit, int ray_unit, int ray_slot, WfRegs &p_, bool &active, long long &spawned) {
    Ray &r = p_.r;
    int32_t &blk = p_.blk; const int32_t record = p_.record;
    int &pos = p_.pos, &depth = p_.depth, &segments = p_.segments;
    const uint32_t rng_pixel_lo = p_.rng_pixel_lo;
    const uint64_t rng_sample = p_.rng_sample;
    unsigned int &path_spawned = p_.path_spawned;
    const double weight = p_.weight;
    auto push = [&](double a, double b, int32_t table, int32_t kind) {
        const bool full = pos == PATH_BLOCK;
        const unsigned int nb = arena_block(ps, full, arena_wave);       // (dev_render.hpp: blocks reserved per wave)
        bool room = true;
        if (full) {
            if (nb >= ps.arena_blocks) { atomicOr(ps.flags, 1u); room = false; }
            else {
                PathTerm link;
                link.a = 0; link.b = 0; link.table = (int32_t)blk; link.kind = TERM_LINK;
                blk = (int32_t)(ps.n_records + nb);
                ps.pool[(long long)blk * PATH_BLOCK] = link;
                pos = 1;
            }
        }
        if (room) {
            PathTerm t;
            t.a = a; t.b = b; t.table = table; t.kind = kind;
            ps.pool[(long long)blk * PATH_BLOCK + pos] = t;
            ++pos;
        }
    };
    auto roulette = [&]() -> int {                                         // ray.pyx:382-388
        if (depth < q->ray_min_depth) return 1;
        if (depth >= q->ray_max_depth) return 0;
        double k1, k2;
        philox2(q->seed, (uint64_t)rng_pixel_lo, rng_sample | ((uint64_t)(2 * depth) << 48), k1, k2);
        return k1 < q->ray_extinction_prob ? 0 : 2;
    };
    const bool was_active = active;
    bool abandoned = false;
    double end_a = 0.0;
    int32_t end_table = -1;
    if (active) {
        const rsx_primitive &p = sc.prims[hit.prim];
        const rsx_material mat = q->materials[p.material];
        Geom g;
        finalise<CSG, MESHES>(sc, r, hit, g);
        double hx, hy, hz;                                            // hit_point.transform(primitive_to_world)
        xform_point(p.to_root, g.hit[0], g.hit[1], g.hit[2], hx, hy, hz);
        // volume emitters containing this segment's origin (Ray._sample_volumes, ray.pyx:422-455), newest first: the list is replayed backwards
        double v_len[PATH_VOL_OVERLAP] = {0, 0, 0, 0}, v_scale[PATH_VOL_OVERLAP] = {0, 0, 0, 0};
        int32_t v_table[PATH_VOL_OVERLAP] = {0, 0, 0, 0}, v_kind[PATH_VOL_OVERLAP] = {0, 0, 0, 0};
        int n_vol = 0;
        bool contains_needs_stream = false;
        if constexpr (VOLS) if (q->n_vol_emitters) world_contains_each<CSG, MODE == 1, MESHES>(sc, r.ox, r.oy, r.oz, ms, contains_needs_stream, [&](int32_t idx) {
            const int32_t vm_id = sc.prims[idx].material;
            const int32_t vt = q->materials[vm_id].type;
            return vt == RSX_MAT_UNIFORM_VOLUME_EMITTER || (vt == RSX_MAT_DIELECTRIC && q->materials[vm_id].light_dir[2] == 0.0);
        }, [&](int32_t idx) {
            const rsx_primitive &vp = sc.prims[idx];
            const rsx_material vm = q->materials[vp.material];
            double length;
            bool skip = false;
            if (vm.type == RSX_MAT_DIELECTRIC) {                      // dielectric.pyx:300-328: world-space length
                const double vx = r.ox - hx, vy = r.oy - hy, vz = r.oz - hz;
                length = sqrt(vx * vx + vy * vy + vz * vz);
            } else {
                double sx, sy, sz, ex, ey, ez;
                xform_point(vp.to_local, hx, hy, hz, sx, sy, sz);
                xform_point(vp.to_local, r.ox, r.oy, r.oz, ex, ey, ez);
                const double vx = sx - ex, vy = sy - ey, vz = sz - ez;
                length = sqrt(vx * vx + vy * vy + vz * vz);
                skip = length == 0;                                   // homogeneous.pyx:92-94
            }
            if (!skip) {
                if (n_vol == PATH_VOL_OVERLAP) atomicOr(ps.flags, 4u);    // more volumes at a point than the registers keep: the pass is traced again by the one-kernel REWALK form
#pragma unroll
                for (int j = PATH_VOL_OVERLAP - 1; j > 0; --j) { v_len[j] = v_len[j - 1]; v_scale[j] = v_scale[j - 1]; v_table[j] = v_table[j - 1]; v_kind[j] = v_kind[j - 1]; }
                v_len[0] = length; v_scale[0] = vm.scale; v_table[0] = vm.table; v_kind[0] = vm.type == RSX_MAT_DIELECTRIC ? TERM_ATTEN : TERM_VOL;
                ++n_vol;
            }
        });
#pragma unroll
        for (int j = 0; j < PATH_VOL_OVERLAP; ++j) if (j < n_vol) push(v_len[j], v_scale[j], v_table[j], v_kind[j]);
        if constexpr (MODE == 1) {
            if (contains_needs_stream) {                              // a CSG volume without a flattened program: redo pass
                atomicOr(q->redo_mask + ray_unit, 1ULL << ray_slot);
                spawned -= (long long)path_spawned;
                abandoned = true;
                active = false;
            }
        }
        ++segments;
        double scatter1 = 0.0, scatter2 = 0.0;
        if (!abandoned && segments < PATH_MAX_SEGMENTS && (arm == WF_LAMBERT || arm == WF_DIELECTRIC))
            philox2(q->seed, (uint64_t)rng_pixel_lo, rng_sample | ((uint64_t)(2 * depth + 1) << 48), scatter1, scatter2);
        bool daughter = false, lambert_term = false;
        double term_a = 1.0, term_b = 1.0;
        if (abandoned) {}
        else if (segments >= PATH_MAX_SEGMENTS) { atomicOr(ps.flags, 2u); active = false; }
        else if (arm == WF_NULL) {                                    // null surface: carry on from the far side (material.pyx:118-147)
            // (selects, not a pointer into the record: a pointer chosen at run time sends the whole Geom to scratch)
            const double fx = g.exiting ? g.outside[0] : g.inside[0], fy = g.exiting ? g.outside[1] : g.inside[1], fz = g.exiting ? g.outside[2] : g.inside[2];
            xform_point(p.to_root, fx, fy, fz, r.ox, r.oy, r.oz);
            ++spawned; ++path_spawned;
        } else if (arm == WF_LAMBERT) {                               // lambert.pyx:76-104 under ContinuousBSDF.evaluate_surface, material.pyx:286-361
            // (the order of k_render_trace_path's Lambert arm, dev_render.hpp: the draw before the surface frame, world_to_surface formed, used
            // and dropped before surface_to_world, the daughter's origin last — the same operations on the same values)
            const bool mis = q->n_important > 0;
            double h1, h2, sx = 0.0, sy = 0.0, sz = 0.0, wx = 0.0, wy = 0.0, wz = 0.0, pdf_important = 0.0;
            bool from_important = false;
            if (mis) {
                const double choose = scatter1, pick = scatter2;
                philox2(q->seed, (uint64_t)rng_pixel_lo | (1ULL << 63), rng_sample | ((uint64_t)(2 * depth + 1) << 48), h1, h2);
                from_important = choose < q->important_path_weight;
                ImportantPick picked;
                picked.dx = picked.dy = picked.dz = picked.distance = picked.radius = 0.0; picked.cone = false;
                if (from_important) picked = important_pick(q->important, q->n_important, hx, hy, hz, pick);
                double sn, cs;
                portable_sincos(2.0 * M_PI * (from_important && picked.cone ? h1 : h2), sn, cs);
                if (from_important) important_direction(picked, h1, h2, sn, cs, wx, wy, wz);
                else {
                    const double rad = sqrt(h1);
                    sx = rad * cs; sy = rad * sn;
                    const double sz2 = 1.0 - sx * sx - sy * sy;
                    sz = sqrt(sz2 > 0 ? sz2 : 0);
                }
            } else {
                double sn, cs;
                h1 = scatter1; h2 = scatter2;
                const double rad = sqrt(h1);
                portable_sincos(2.0 * M_PI * h2, sn, cs);
                sx = rad * cs; sy = rad * sn;
                const double sz2 = 1.0 - sx * sx - sy * sy;
                sz = sqrt(sz2 > 0 ? sz2 : 0);
            }
            double nx = g.normal[0], ny = g.normal[1], nz = g.normal[2];
            if (g.exiting) { nx = -nx; ny = -ny; nz = -nz; }
            double ux = nx, uy = ny, uz = nz;
            normalise3(ux, uy, uz);
            double vx = 1, vy = 0, vz = 0;
            if (fabs(ux * vx + uy * vy + uz * vz) > 0.5) { vx = 0; vy = 1; }
            const double m = ux * vx + uy * vy + uz * vz;
            double tx = vx - m * ux, ty = vy - m * uy, tz = vz - m * uz;
            normalise3(tx, ty, tz);
            const double bx = ny * tz - ty * nz, by = nz * tx - tz * nx, bz = nx * ty - tx * ny;    // normal.cross(tangent)
            if (mis && from_important) {
                const double *wtp = p.to_local;
                double wts[9];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    wts[0 + j] = tx * wtp[j] + ty * wtp[4 + j] + tz * wtp[8 + j] + 0.0 * wtp[12 + j];
                    wts[3 + j] = bx * wtp[j] + by * wtp[4 + j] + bz * wtp[8 + j] + 0.0 * wtp[12 + j];
                    wts[6 + j] = nx * wtp[j] + ny * wtp[4 + j] + nz * wtp[8 + j] + 0.0 * wtp[12 + j];
                }
                sx = wts[0] * wx + wts[1] * wy + wts[2] * wz;
                sy = wts[3] * wx + wts[4] * wy + wts[5] * wz;
                sz = wts[6] * wx + wts[7] * wy + wts[8] * wz;
            }
            double dirx, diry, dirz;
            {
                const double *a = p.to_root;
                double stw[9];
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    stw[3 * ii + 0] = a[4 * ii] * tx + a[4 * ii + 1] * ty + a[4 * ii + 2] * tz + a[4 * ii + 3] * 0.0;
                    stw[3 * ii + 1] = a[4 * ii] * bx + a[4 * ii + 1] * by + a[4 * ii + 2] * bz + a[4 * ii + 3] * 0.0;
                    stw[3 * ii + 2] = a[4 * ii] * nx + a[4 * ii + 1] * ny + a[4 * ii + 2] * nz + a[4 * ii + 3] * 0.0;
                }
                dirx = stw[0] * sx + stw[1] * sy + stw[2] * sz;
                diry = stw[3] * sx + stw[4] * sy + stw[5] * sz;
                dirz = stw[6] * sx + stw[7] * sy + stw[8] * sz;
            }
            if (mis) {
                if (!from_important) { wx = dirx; wy = diry; wz = dirz; }
                pdf_important = important_pdf(q->important, q->n_important, hx, hy, hz, wx, wy, wz);
            }
            const double pdf = sz >= 0.0 ? M_1_PI * sz : 0.0;         // HemisphereCosineSampler.pdf
            const double pdf_all = mis ? q->important_path_weight * pdf_important + (1 - q->important_path_weight) * pdf : pdf;
            const double rcp = 1.0 / pdf_all;                         // div_scalar (spectrum.pyx:459-467)
            if (pdf == 0.0) { push(pdf, rcp, mat.table, TERM_LAMBERT); active = false; }
            else {
                const double fx = g.exiting ? g.inside[0] : g.outside[0], fy = g.exiting ? g.inside[1] : g.outside[1], fz = g.exiting ? g.inside[2] : g.outside[2];
                xform_point(p.to_root, fx, fy, fz, r.ox, r.oy, r.oz);
                r.dx = dirx; r.dy = diry; r.dz = dirz;
                daughter = true; lambert_term = true; term_a = pdf; term_b = rcp;
            }
        } else if (arm == WF_DIELECTRIC) {                            // dielectric.pyx:159-262
            double ix, iy, iz;
            xform_vector(p.to_local, r.dx, r.dy, r.dz, ix, iy, iz);
            normalise3(ix, iy, iz);
            double nx = g.normal[0], ny = g.normal[1], nz = g.normal[2];
            normalise3(nx, ny, nz);
            const double c1 = -(nx * ix + ny * iy + nz * iz);
            const bool inside = c1 < 0.0;
            const double n1 = inside ? mat.scale : mat.light_dir[0], n2 = inside ? mat.light_dir[0] : mat.scale;
            const bool transmission_only = mat.light_dir[1] != 0.0;
            const double gamma = n1 / n2;
            const double c2s = 1 - (gamma * gamma) * (1 - c1 * c1);
            bool reflect = true;
            double ox = 0, oy = 0, oz = 0;
            if (c2s > 0) {
                const double temp = inside ? gamma * c1 + sqrt(c2s) : gamma * c1 - sqrt(c2s);
                ox = gamma * ix + temp * nx; oy = gamma * iy + temp * ny; oz = gamma * iz + temp * nz;
                const double ci = c1, ct = -(nx * ox + ny * oy + nz * oz);
                const double ra = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct), rb = (n1 * ct - n2 * ci) / (n1 * ct + n2 * ci);
                const double reflectivity = 0.5 * (ra * ra + rb * rb);
                const double transmission = 1 - reflectivity;
                if (transmission_only) reflect = false;
                else reflect = !(scatter1 < transmission);
            }
            if (reflect && transmission_only) active = false;         // total internal reflection without a reflected ray: zero spectrum
            else {
                if (reflect) {
                    const double temp = 2 * c1;
                    ox = ix + temp * nx; oy = iy + temp * ny; oz = iz + temp * nz;
                }
                const bool from_inside = reflect == inside;           // reflect: the side the ray came from; transmit: the far side
                const double fx = from_inside ? g.inside[0] : g.outside[0], fy = from_inside ? g.inside[1] : g.outside[1], fz = from_inside ? g.inside[2] : g.outside[2];
                xform_point(p.to_root, fx, fy, fz, r.ox, r.oy, r.oz);
                xform_vector(p.to_root, ox, oy, oz, r.dx, r.dy, r.dz);
                daughter = true;
            }
        } else {                                                      // optical/ray.pyx:391-393: the path ends at this surface
            if (mat.type == RSX_MAT_UNIFORM_EMITTER) { end_a = mat.scale; end_table = mat.table; }
            else if (mat.type == RSX_MAT_DEBUG_LIGHT && mat.scale != 0.0) {
                double lx, ly, lz;
                xform_vector(p.to_local, -mat.light_dir[0], -mat.light_dir[1], -mat.light_dir[2], lx, ly, lz);
                const double dot = lx * g.normal[0] + ly * g.normal[1] + lz * g.normal[2];
                end_a = mat.scale * (dot > 0 ? dot : 0.0);
                end_table = mat.table;
            }
            active = false;
        }
        if (daughter) {                                               // ray.pyx:380-388: the daughter exists (and counts) before its roulette
            ++depth;
            ++spawned; ++path_spawned;
            const int alive = roulette();
            if (!alive) active = false;
            if (lambert_term || alive == 2) push(term_a, term_b, mat.table, !lambert_term ? TERM_NORM : alive == 2 ? TERM_LAMBERT_NORM : TERM_LAMBERT);
        }
    }
    if (was_active && !active && !abandoned) wf_finish(samples, ps, record, blk, pos, weight, end_a, end_table);
}
template <bool CSG, int MODE = 0, bool VOLS = true, bool STAGED = false, bool MESHES = true>
__global__ __launch_bounds__(WG_THREADS, RSX_WF_MIN_WAVES) void k_wf_level(DScene sc_arg, RenderParams rp, Sample *samples, WfStore wf, PathStore ps) {
    static_assert(!CSG || MODE == 1, "CSG scenes: the staged form is the fast pass (state-free evaluator); the redo pass is k_render_trace_path<true, 2>");
    __shared__ uint32_t seg_first[WF_SEGS + 1];            // first chunk of every (arm, sub-list) segment of list_in
    __shared__ uint32_t seg_count[WF_SEGS];                // ... and its length
    __shared__ uint32_t arena_res[2 * WG_WAVES];           // arena_block: the blocks each wave has reserved
    if (threadIdx.x < 2 * WG_WAVES) arena_res[threadIdx.x] = 0;
    const unsigned long long rp_bits = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DScene) + 7) & ~(size_t)7);
    const RSX_CONST_AS RenderParams *q = (const RSX_CONST_AS RenderParams *)rp_bits;
    (void)rp;
    DScene sc = sc_arg;
    wf_stage_scene(sc_arg, sc, q, STAGED, CSG);
    const bool first = wf.cnt_in == nullptr;               // (wave-uniform: a kernel argument)
    if (!first) {
        // sub-list lengths, then the running sum of their chunk counts (one wave, 64 segments at a time)
        for (int k = threadIdx.x; k < WF_SEGS; k += blockDim.x) seg_count[k] = wf.cnt_in[k];
        __syncthreads();
        if (threadIdx.x < WAVE) {
            uint32_t carry = 0;
            for (int base = 0; base < WF_SEGS; base += WAVE) {
                const int k = base + (int)threadIdx.x;
                uint32_t incl = k < WF_SEGS ? (seg_count[k] + WAVE - 1) / WAVE : 0u;
                for (int o = 1; o < WAVE; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += up; }
                if (k < WF_SEGS) seg_first[k + 1] = carry + incl;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
            }
            if (threadIdx.x == 0) seg_first[0] = 0;
        }
    }
    __syncthreads();
    Stack st, ms;
    wave_stacks(sc, st, ms);
    NodeSt csg_state[1];
    const int lane = threadIdx.x % WAVE;
    const long long n_chunks = first ? (wf.n + WAVE - 1) / WAVE : (long long)seg_first[WF_SEGS];
    const long long n_waves = (long long)gridDim.x * WG_WAVES;
    const long long my_wave = (long long)blockIdx.x * WG_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint32_t my_sub = (uint32_t)(my_wave % WF_SUB);
    long long spawned = 0;
#if RSX_PHASE_PROF == 3
    // (tuning builds: s_memtime per phase of an iteration, summed over the waves into the rsx_debug_unit_times buffer — [0] list entry and
    // path record arrive, [1] material arm, [2] walk, [3] filing, [4] iterations, [5] live lanes, [6] lanes that go on)
    unsigned long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WF_STAMP(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = clock64(); pp[k] += now_ - pp_mark; pp_mark = now_; }
#else
#define WF_STAMP(k)
#endif
    // which sub-list holds chunk c, and which of its entries lane 0 takes: seg_first[lo] <= c < seg_first[lo + 1]
    auto locate = [&](long long c, int &lo, uint32_t &i0) {
        lo = 0;
        int hi = WF_SEGS;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((long long)seg_first[mid] <= c) lo = mid; else hi = mid; }
        lo = __builtin_amdgcn_readfirstlane(lo);
        i0 = (uint32_t)(c - (long long)seg_first[lo]) * WAVE;
    };
    // The slot numbers of a wave's NEXT chunk are requested while it works on the current one. (Touching the two lines of those paths
    // before the current chunk's walk, to have them in the L2 when they are wanted, was measured and dropped: -0.9 k clocks at the top
    // of an iteration, +2.6 k in the walk, whose few scratch reloads then wait behind the touches on the in-order memory counter.)
    int lo_next = 0;
    bool active_next = false;
    uint32_t slot_next = 0;
    if (!first && my_wave < n_chunks) {
        uint32_t i0;
        locate(my_wave, lo_next, i0);
        active_next = i0 + (uint32_t)lane < seg_count[lo_next];
        slot_next = active_next ? wf.list_in[(size_t)lo_next * wf.sub_stride + i0 + lane] : 0u;
    }
    for (long long c = my_wave; c < n_chunks; c += n_waves) {
        int key = 0;                                           // (wave-uniform: the list this chunk's paths waited in)
        bool active;
        uint32_t slot;
        bool have_next = false;
        if (first) {
            slot = (uint32_t)(c * WAVE + lane);
            active = (long long)slot < wf.n;
        } else {
            key = lo_next / WF_SUB; active = active_next; slot = slot_next;
            have_next = c + n_waves < n_chunks;
            if (have_next) {
                uint32_t i0;
                locate(c + n_waves, lo_next, i0);
                active_next = i0 + (uint32_t)lane < seg_count[lo_next];
                slot_next = active_next ? wf.list_in[(size_t)lo_next * wf.sub_stride + i0 + lane] : 0u;
            }
        }
        const int arm = key == WF_LAMBERT_IMPORTANT ? WF_LAMBERT : key;      // the material arm the chunk's paths wait for
#if RSX_PHASE_PROF == 3
        unsigned long long pp_mark = clock64();
#endif
        WfPath &path = wf.paths[slot];
        WfRegs pr;
        Ray &r = pr.r;
        r.ox = r.oy = r.oz = 0; r.dx = r.dy = 0; r.dz = 1; r.maxd = INFINITY;
        int32_t &blk = pr.blk, &record = pr.record;
        int &pos = pr.pos, &depth = pr.depth, &segments = pr.segments;
        uint32_t &rng_pixel_lo = pr.rng_pixel_lo;
        uint64_t &rng_sample = pr.rng_sample;
        unsigned int &path_spawned = pr.path_spawned;
        double &weight = pr.weight;
        blk = 0; record = 0; pos = 0; depth = 0; segments = 0; rng_pixel_lo = 0; rng_sample = 0; path_spawned = 0; weight = 0;
        const int ray_unit = (int)(wf.first_unit + (long long)(slot / WAVE)), ray_slot = (int)(slot % WAVE);    // where the path's primary ray came from
        if (first) {
            const UnitPixel px = unit_pixel(q, wf.first_unit + c, lane);
            active = active && px.valid;
            if (active) {                                       // PinholeCamera._generate_rays, as k_render_trace_path's refill
                rng_pixel_lo = (uint32_t)px.ix * (uint32_t)q->cam.ny + (uint32_t)px.iy; rng_sample = q->sample_offset + (uint64_t)px.s;
                double u1, u2;
                if (q->rng_mode == RSX_RNG_STREAM) { u1 = q->uniforms[2 * (px.k * q->spp + px.s)]; u2 = q->uniforms[2 * (px.k * q->spp + px.s) + 1]; }
                else philox2(q->seed, (uint64_t)rng_pixel_lo, rng_sample, u1, u2);
                camera_ray(q, px.ix, px.iy, u1, u2, r, weight);
                record = (int32_t)(px.slot * q->spp + px.s);
                blk = record; path_spawned = 1;
                ++spawned;
            }
        } else if (active) {
            r.ox = path.ox; r.oy = path.oy; r.oz = path.oz; r.dx = path.dx; r.dy = path.dy; r.dz = path.dz;
            blk = path.blk; pos = path.pos; depth = path.depth; segments = path.segments;
            record = path.record; rng_pixel_lo = path.rng_pixel_lo; rng_sample = path.rng_sample; weight = path.weight; path_spawned = path.path_spawned;
        }
        WF_STAMP(0)
#if RSX_PHASE_PROF == 3
        pp[4] += 1; pp[5] += __popcll(__ballot(active));
#endif
        // ---- the material arm of the segment the previous level walked ----
        if (!first) {
            Hit hit;
            hit.prim = -1;
            if (active) {
                if constexpr (CSG) hit = wf.csg_hits[slot];
                else { hit.t = path.t; hit.prim = path.prim; hit.a0 = path.a0; hit.a1 = path.a1; hit.u = path.u; hit.v = path.v; hit.w = path.w; hit.leaf = 0; hit.flags = 0; hit.hx = hit.hy = hit.hz = 0; }
            }
            wf_arm<CSG, MODE, VOLS, MESHES>(sc, q, ps, samples, ms, arena_res + 2 * (threadIdx.x / WAVE), arm, hit, ray_unit, ray_slot, pr, active, spawned);
        }
        // ---- the next segment: Ray.trace's world.hit for every path that goes on ----
        WF_STAMP(1)
#if RSX_PHASE_PROF == 3
        pp[6] += __popcll(__ballot(active));
#endif
        int next_key = -1;
        if (__any(active)) {
            Hit hit;
            uint32_t work = 0;
            const bool got = world_trace_wave<CSG, MODE == 1, RSX_STAGE_MIN, false, !CSG ? 8 : MODE == 1 && RSX_CSG_MAILBOX >= 4 ? RSX_CSG_WIDE : 2, RSX_CSG_MAILBOX, MESHES>(active, sc, r, st, ms, csg_state, hit, work);
            if (active) {
                bool abandoned = false;
                if constexpr (MODE == 1) abandoned = (work >> 31) != 0;
                if (abandoned) {                                              // this path needs the stream merge: the redo pass traces it again
                    atomicOr(q->redo_mask + ray_unit, 1ULL << ray_slot);
                    spawned -= (long long)path_spawned;
                } else if (!got) wf_finish(samples, ps, record, blk, pos, weight, 0.0, -1);    // new_spectrum(): no volume pass for a segment that hits nothing
                else {
                    const rsx_material mat = q->materials[sc.prims[hit.prim].material];
                    next_key = mat.type == RSX_MAT_LAMBERT ? WF_LAMBERT : mat.type == RSX_MAT_DIELECTRIC ? WF_DIELECTRIC :
                               (mat.type == RSX_MAT_NULL || mat.type == RSX_MAT_UNIFORM_VOLUME_EMITTER) ? WF_NULL : WF_END;
                    if (next_key == WF_LAMBERT && q->n_important > 0) {       // the arm's `choose` draw (material.pyx:327-331), drawn here as well: it only picks the list
                        double choose, unused;
                        philox2(q->seed, (uint64_t)rng_pixel_lo, rng_sample | ((uint64_t)(2 * depth + 1) << 48), choose, unused);
                        if (choose < q->important_path_weight) next_key = WF_LAMBERT_IMPORTANT;
                    }
                    // A path that ends at this surface and has no volume around its last segment to add (no material of the scene has
                    // a volume contribution) is finished here: its arm — optical/ray.pyx:391-393 — is two loads.
                    if (next_key == WF_END && (!VOLS || q->n_vol_emitters == 0) && segments + 1 < PATH_MAX_SEGMENTS && mat.type != RSX_MAT_DEBUG_LIGHT) {
                        const bool emits = mat.type == RSX_MAT_UNIFORM_EMITTER;
                        wf_finish(samples, ps, record, blk, pos, weight, emits ? mat.scale : 0.0, emits ? mat.table : -1);
                        next_key = -1;
                    } else {
                        path.ox = r.ox; path.oy = r.oy; path.oz = r.oz; path.dx = r.dx; path.dy = r.dy; path.dz = r.dz;
                        path.blk = blk; path.pos = pos; path.depth = depth; path.segments = segments;
                        if constexpr (CSG) wf.csg_hits[slot] = hit;
                        else { path.t = hit.t; path.prim = hit.prim; path.a0 = hit.a0; path.a1 = hit.a1; path.u = hit.u; path.v = hit.v; path.w = hit.w; }
                        path.path_spawned = path_spawned;
                        if (first) { path.record = record; path.rng_pixel_lo = rng_pixel_lo; path.rng_sample = rng_sample; path.weight = weight; path.pad = 0; }
                    }
                }
            }
        }
        WF_STAMP(2)
        // file the paths that go on: lane k < WF_KEYS reserves list k's entries (ONE atomic instruction per chunk for all the lists)
        {
            unsigned long long mk[WF_KEYS];
#pragma unroll
            for (int k = 0; k < WF_KEYS; ++k) mk[k] = __ballot(next_key == k);
            unsigned int want = 0;
#pragma unroll
            for (int k = 0; k < WF_KEYS; ++k) if (lane == k) want = (unsigned int)__popcll(mk[k]);
            unsigned int base = 0;
            if (want) base = atomicAdd(wf.cnt_out + lane * WF_SUB + my_sub, want);
            int bad = 0;
            if (want && base + want > wf.sub_stride) { atomicOr(ps.flags, 8u); bad = 1; }      // (cannot happen: render() sizes the sub-lists for it)
            unsigned long long mine = 0;
            unsigned int my_base = 0;
            int my_bad = 0;
#pragma unroll
            for (int k = 0; k < WF_KEYS; ++k) {
                const unsigned int bk = (unsigned int)__builtin_amdgcn_readlane((int)base, k);
                const int badk = __builtin_amdgcn_readlane(bad, k);
                if (next_key == k) { mine = mk[k]; my_base = bk; my_bad = badk; }
            }
            if (next_key >= 0 && !my_bad)
                wf.list_out[((size_t)next_key * WF_SUB + my_sub) * wf.sub_stride + my_base + (unsigned int)__popcll(mine & ((1ULL << lane) - 1ULL))] = slot;
        }
        WF_STAMP(3)
    }
#if RSX_PHASE_PROF == 3
    if (lane == 0 && q->unit_times) for (int k = 0; k < 8; ++k) atomicAdd(q->unit_times + 32 + k + (first ? 8 : 0), pp[k]);
#endif
#undef WF_STAMP
    for (int o = 32; o > 0; o >>= 1) spawned += __shfl_xor(spawned, o);
    if (lane == 0 && spawned != 0) atomicAdd(reinterpret_cast<unsigned long long *>(ps.flags) + 1, (unsigned long long)spawned);
}

// The paths still alive after the levels go to the one-kernel form's drain launch (k_render_trace_path, PathStore::drain): a few
// paths trapped by total internal reflection run for hundreds of segments, a launch per segment is no way to walk them. They
// wait for a material arm, and the drain launch starts its paths with a walk: the arm is not lost — the one-kernel form's round
// is walk, then arm, and a handed-on path's ray is the segment whose hit is already known: it is walked again (same ray, same hit).
__global__ void k_wf_to_queue(WfStore wf, PathStore ps) {
    __shared__ unsigned int seg_base[WF_SEGS + 1];
    if (threadIdx.x == 0) { unsigned int s = 0; for (int k = 0; k < WF_SEGS; ++k) { seg_base[k] = s; s += wf.cnt_in[k]; } seg_base[WF_SEGS] = s; }
    __syncthreads();
    const unsigned int n = seg_base[WF_SEGS];
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = WF_SEGS;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_base[mid] <= i) lo = mid; else hi = mid; }
        const uint32_t slot = wf.list_in[(size_t)lo * wf.sub_stride + (i - seg_base[lo])];
        const WfPath &p = wf.paths[slot];
        PathState s;
        s.r.ox = p.ox; s.r.oy = p.oy; s.r.oz = p.oz; s.r.dx = p.dx; s.r.dy = p.dy; s.r.dz = p.dz; s.r.maxd = INFINITY;
        s.smp.a = 0.0; s.smp.weight = p.weight; s.smp.table = -1; s.smp.pad = 0;
        s.record = p.record; s.blk = p.blk; s.rng_pixel = (uint64_t)p.rng_pixel_lo; s.rng_sample = p.rng_sample;
        s.path_spawned = p.path_spawned; s.pos = p.pos; s.depth = p.depth; s.segments = p.segments;
        s.ray_unit = (int32_t)(wf.first_unit + (long long)(slot / WAVE)); s.ray_slot = (int32_t)(slot % WAVE); s.pad = 0;
        ps.queue[i] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { ps.queue_count[0] = n; ps.queue_count[1] = 0; }
}
