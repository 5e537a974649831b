// dev_render.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// observe(): ray generation + trace + shading kernel, unit scheduling, Welford / frame-merge kernels.
#pragma once

// ---------------------------------------------------------------------------------------------------
// observe(): sample generation + trace + shading  ->  per-sample records; then per-(pixel,bin) Welford
// ---------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011): counter = (pixel, sample), key = seed
__device__ __forceinline__ void philox2(uint64_t seed, uint64_t pixel, uint64_t sample, double &u1, double &u2) {
    uint32_t c0 = (uint32_t)pixel, c1 = (uint32_t)(pixel >> 32), c2 = (uint32_t)sample, c3 = (uint32_t)(sample >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint64_t a = ((uint64_t)c1 << 32) | c0, b = ((uint64_t)c3 << 32) | c2;
    u1 = (double)(a >> 11) * (1.0 / 9007199254740992.0);
    u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

// clock the unit costs are measured with (only ever compared within one lane's cost array)
// Unit cost that steers the longest-first schedule = the wave's own count of traversal rounds (`work`): free and deterministic.
// Reading s_memrealtime / s_memtime around every unit was measured on configs[2]: the reads serialise chip-wide (~9 ns each
// whatever the occupancy) and doubled the kernel time of a 4.2 M-unit pass.
#ifndef RSX_LPT_MAX_UNITS
#define RSX_LPT_MAX_UNITS (1 << 18)   // passes with more 64-ray units than this are not re-ordered
#endif
// Per-unit timestamps (rsx_debug_unit_times: start, end, workgroup of every 64-ray unit) exist in tuning builds only (-DRSX_UNIT_STAMPS=1,
// and whenever a profile build asks for its counters): carried through the walk, the start stamp and the buffer tests cost the
// per-lane kernels registers in every production pass.
#ifndef RSX_UNIT_STAMPS
#define RSX_UNIT_STAMPS (RSX_PHASE_PROF || RSX_UTIL_PROF)
#endif
struct RenderParams {
    rsx_camera cam;
    const rsx_material *materials;
    const int32_t *tasks;      // device [n_tasks,2] or null
    const double *uniforms;    // device or null
    long long n_tasks;
    int32_t rect[4];
    int32_t spp, rng_mode;
    uint64_t seed, sample_offset;
    int32_t ray_max_depth, ray_min_depth;      // Ray.max_depth / extinction_min_depth / extinction_prob: path kernel only
    double ray_extinction_prob;
    const rsx_important_sphere *important;     // ImportanceManager spheres (world.pyx:47-128) or n_important == 0
    int32_t n_important, passes;
    double important_path_weight;
    int32_t prims_lds;                         // > 0: byte offset of the LDS copy of the primitive records (+ CSG programs) in the path kernel
    int32_t n_vol_emitters, world_lds;         // materials with a volume contribution (0: the per-segment world.contains() pass is skipped);
                                               // world_lds > 0: byte offset in the workgroup's LDS where the path kernel stages the world tree
    int32_t bank_lds;                          // path kernel: byte offset in the workgroup's LDS of its waves' ray banks (RAY_BANK_BYTES each; k_render_trace_path)
    unsigned long long *redo_mask;    // [n_units] CSG scenes: lanes of each unit the fast pass could not finish (ties -> stream merge), or null
    uint32_t *unit_cost;              // [n_units] measured duration of each unit in this launch (100 MHz ticks), feeds the next launch's order
    const uint32_t *unit_order;       // work list: ticket k of a list processes unit unit_order[k]
    const uint32_t *seg;              // [10] begin offsets of the shared heavy list and the 8 per-XCD lists in unit_order (+ end)
    int32_t measure_cost;             // 1: record unit costs (small, tail-bound passes); 0: large passes keep the natural order
    unsigned long long *unit_times;   // optional [n_units,12]: wall_clock64 start, end, (xcc<<16 | cu) per 64-ray unit (tuning aid)
    double origin[3];                 // the pinhole in world space: camera_origin(cam) formed once on the host (the same IEEE operations every ray made for itself)
};

// Point3D(0, 0, 0).transform(camera.to_root) — point.pyx:253-284 with the reference's products by zero kept (a non-finite matrix entry
// makes the same NaN). Host and device form it with the same correctly rounded operations (-ffp-contract=off on both sides).
__host__ __device__ inline void camera_origin(const rsx_camera &cam, double *o) {
    const double *m = cam.to_root;
    double wq = m[12] * 0.0 + m[13] * 0.0 + m[14] * 0.0 + m[15];
    wq = 1.0 / wq;
    o[0] = (m[0] * 0.0 + m[1] * 0.0 + m[2] * 0.0 + m[3]) * wq;
    o[1] = (m[4] * 0.0 + m[5] * 0.0 + m[6] * 0.0 + m[7]) * wq;
    o[2] = (m[8] * 0.0 + m[9] * 0.0 + m[10] * 0.0 + m[11]) * wq;
}

// per-sample record consumed by k_accumulate: x[bin] = (a * table[bin]) * weight
struct Sample {
    double a, weight;
    int32_t table, pad;
};

__device__ __forceinline__ void task_pixel(const RenderParams &rp, long long k, int &ix, int &iy) {
    if (rp.tasks) { ix = rp.tasks[2 * k]; iy = rp.tasks[2 * k + 1]; }
    else { const int w = rp.rect[2] - rp.rect[0]; ix = rp.rect[0] + (int)(k % w); iy = rp.rect[1] + (int)(k / w); }
}

// The rays of a pass are numbered g = pixel * spp + sample, with the pixels taken tile by tile (8x8 tiles, row-major inside a
// tile; the entries of the list in task mode), and unit u renders rays 64u .. 64u + 63. So a unit is always the most coherent bundle
// the pass offers: one 8x8 tile of one sample index at 1 spp, ONE pixel at 64 spp, at most two neighbouring pixels at the
// reference's default 100 spp. Rays through one pixel differ only by sub-pixel jitter, so the lanes of a wave walk nearly the same
// nodes and leaves (an 8x8 tile per sample index, whatever the spp, had 27 % of its lanes busy in the KD steps on configs[2]).
struct UnitPixel {
    long long k;               // task index (row-major in rect mode)
    long long slot;            // sample-record slot of the pixel (x-major in rect mode, like the frame)
    int ix, iy, s;
    bool valid;
};

__device__ __forceinline__ UnitPixel unit_pixel(const RSX_CONST_AS RenderParams *q, long long unit, int lane) {
    UnitPixel px;
    const uint32_t spp = (uint32_t)q->spp;
    const unsigned long long g = (unsigned long long)unit * WAVE + (unsigned)lane;
    unsigned long long pixel;
    if (g < (1ULL << 32)) { pixel = (uint32_t)g / spp; px.s = (int)((uint32_t)g % spp); }      // 32-bit divide when it fits
    else { pixel = g / spp; px.s = (int)(g % spp); }
    if (q->tasks) {
        px.k = (long long)pixel;
        px.valid = px.k < q->n_tasks;
        if (!px.valid) px.k = 0;
        px.ix = q->tasks[2 * px.k]; px.iy = q->tasks[2 * px.k + 1];
        px.slot = px.k;
    } else {
        const int w = q->rect[2] - q->rect[0], h = q->rect[3] - q->rect[1];
        const int tiles_x = (w + 7) >> 3;
        const long long tile = (long long)(pixel >> 6);
        const int within = (int)(pixel & 63);
        const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
        const int lx = (tx << 3) + (within & 7), ly = (ty << 3) + (within >> 3);
        px.valid = lx < w && ly < h;
        px.ix = q->rect[0] + (px.valid ? lx : 0); px.iy = q->rect[1] + (px.valid ? ly : 0);
        px.k = px.valid ? (long long)ly * w + lx : 0;
        px.slot = px.valid ? (long long)lx * h + ly : 0;
    }
    return px;
}

// PinholeCamera._generate_rays (pinhole.pyx:169-204) + RectangleSampler3D.sample (surface3d.pyx:197-198) + the observer's transform to
// world space (observer.pyx:403-404: origin (0,0,0) and the direction through camera.to_root). u1 is the FIRST draw of the sample:
// the reference build draws the y jitter first, then x (C argument evaluation order of new_point3d(...) under gcc).
__device__ __forceinline__ void camera_ray(const RSX_CONST_AS RenderParams *q, int ix, int iy, double u1, double u2, Ray &r, double &weight) {
    const double delta = q->cam.image_delta, half = 0.5 * delta;
    const double pixel_x = q->cam.image_start_x - delta * ((double)ix + 0.5);
    const double pixel_y = q->cam.image_start_y - delta * ((double)iy + 0.5);
    double dx = (u2 * delta - half) + pixel_x, dy = (u1 * delta - half) + pixel_y, dz = 0.0 + 1.0;
    normalise3(dx, dy, dz);
    weight = dz;
    const RSX_CONST_AS double *m = q->cam.to_root;
    r.ox = q->origin[0]; r.oy = q->origin[1]; r.oz = q->origin[2];           // (camera_origin, once per pass on the host)
    r.dx = m[0] * dx + m[1] * dy + m[2] * dz;
    r.dy = m[4] * dx + m[5] * dy + m[6] * dz;
    r.dz = m[8] * dx + m[9] * dy + m[10] * dz;
    r.maxd = INFINITY;
}

// StatsArray _add_sample / _combine_samples — core/math/statsarray.pyx:743-859.
// Every division here is by a small positive integer whose refined reciprocal is shared by the two divisions of consecutive
// Welford steps (IntRcp); the quotient is formed by exact_div — bit-identical to `/` (see refine_rcp /
// rsx_selftest_exact_division). At 64 samples/pixel x 15 bins the accumulate kernel is VALU-bound on this recurrence
// (4.0e9 updates in 8.7 ms on configs[2]); staging the sample records through LDS was measured and changed nothing.
// the lanes whose x is +0.0, as a lane mask (written out: through ballot the compiler turns the class test into a value and compares it again)
__device__ __forceinline__ unsigned long long mask_plus_zero(double x) {
    unsigned long long m;
    asm("v_cmp_class_f64_e64 %0, %1, 64" : "=s"(m) : "v"(x));
    return m;
}

struct IntRcp {
    double d, y;
    __device__ __forceinline__ explicit IntRcp(int n) : d((double)n), y(refine_rcp((double)n)) {}
    __device__ __forceinline__ IntRcp(double d_, double y_) : d(d_), y(y_) {}
    __device__ __forceinline__ double div(double numer) const { return exact_div(numer, d, y, d > 0.0); }
};

// n -> n + 1 samples; `by_n` = IntRcp(n + 1), `by_nm1` = IntRcp(n) (the divisors of the update)
__device__ __forceinline__ void add_sample(double x, double &m, double &v, int &n, const IntRcp &by_n, const IntRcp &by_nm1) {
    if (n == 0) { n = 1; m = x; v = 0; return; }
    const double pm = m, pv = v;
    const int pn = n > 1 ? n : 2;
    n += 1;
    m = pm + by_n.div(x - pm);
    v = by_nm1.div(pv * (pn - 1) + (x - pm) * (x - m));
}

// `rcp(n)` = IntRcp(n): computed (v_rcp_f64 and two refinement steps per divisor), or read from a table when the counts are wave-uniform
template <class Rcp>
__device__ __forceinline__ void combine_samples_with(double mx, double vx, int nx, double my, double vy, int ny, double &mt, double &vt, int &nt, Rcp rcp) {
    if (nx < ny) { const int ti = nx; nx = ny; ny = ti; double td = mx; mx = my; my = td; td = vx; vx = vy; vy = td; }
    if (nx > 1 && ny > 1) {
        nt = nx + ny;
        const IntRcp by_nt = rcp(nt);
        mt = by_nt.div(nx * mx + ny * my);
        vx = rcp(nx).div((nx - 1) * vx);
        vy = rcp(ny).div((ny - 1) * vy);
        vt = by_nt.div(nx * (mx * mx + vx) + ny * (my * my + vy)) - mt * mt;
        vt = rcp(nt - 1).div(nt * vt);
        return;
    }
    if (nx == 0 && ny == 0) { nt = 0; mt = 0; vt = 0; }
    else if (nx == 1) {
        if (ny == 0) { nt = 1; mt = mx; vt = 0; }
        else { nt = 2; mt = 0.5 * (mx + my); const double temp = mx - mt; vt = 2 * temp * temp; }
    } else if (nx > 1) {
        nt = nx; mt = mx; vt = vx;
        if (ny == 1) add_sample(my, mt, vt, nt, rcp(nx + 1), rcp(nx));
    } else { nt = 0; mt = 0; vt = 0; }
}

__device__ __forceinline__ void combine_samples(double mx, double vx, int nx, double my, double vy, int ny, double &mt, double &vt, int &nt) {
    combine_samples_with(mx, vx, nx, my, vy, ny, mt, vt, nt, [](int n) { return IntRcp(n); });
}

// The same merge for a wave whose lanes all hold the same two counts — every pixel of an ordinary render has seen the same passes —
// and counts within the table `consts` ([i] = {(double)i, refine_rcp(i)}, k_fill_acc_consts): the divisors and their reciprocals come
// over the scalar data path and the case analysis is scalar control flow. Same operations on the same values; anything else (an
// adaptive sampler's uneven counts, long renders) takes combine_samples. `limit` = entries in the table.
__device__ __forceinline__ void combine_samples_uniform(double mx, double vx, int nx, double my, double vy, int ny, double &mt, double &vt, int &nt,
                                                        const RSX_CONST_AS double *consts, int limit) {
    const int nxu = __builtin_amdgcn_readfirstlane(nx), nyu = __builtin_amdgcn_readfirstlane(ny);
    const bool same = __builtin_amdgcn_ballot_w64(nx != nxu || ny != nyu) == 0ULL;
    if (consts != nullptr && same && nxu >= 0 && nyu >= 0 && (long long)nxu + nyu + 1 < limit)
        combine_samples_with(mx, vx, nxu, my, vy, nyu, mt, vt, nt, [consts](int n) { return IntRcp(consts[2 * n], consts[2 * n + 1]); });
    else combine_samples(mx, vx, nx, my, vy, ny, mt, vt, nt);
}

// ---------------------------------------------------------------------------------------------------
// Welford in the trace kernel (fused passes)
// ---------------------------------------------------------------------------------------------------
// Passes that run alone on the context stream, whose 64-ray units hold whole pixels (64 % spp == 0) and whose materials are the
// closed-form ones, do not hand their sample records to k_accumulate: every wave keeps the records of its last FUSE_UNITS units in a
// private global ring (L2-resident: 6 KB per wave, rewritten every four units), and after the fourth unit runs the per-(pixel, bin)
// Welford recurrence and the frame merge itself — the chains of four units fill the wave (4 pixels x 15 bins = 60 lanes at 64 spp).
// The records are staged through the wave's own LDS region (idle between units: the traversal stacks are empty), as
// a[256] | weight[256] | table[256] | reciprocals[spp + 2] | spectral tables. Same operations in the same order as k_accumulate
// (statsarray.pyx:743-776 per sample, then combine_samples into the frame): frames are bit-identical to the two-kernel form.
// What it buys: the 6.4 GB sample-record round trip through HBM and the 5.8 ms accumulate kernel of a 2048^2 x 64 spp pass go. What
// it costs: the recurrence (34 instructions per sample and bin) moves into a kernel that is already bound by instruction issue —
// measured 41.9 ms against 34.9 + 5.8 = 40.8 ms for the two kernels, so the form is opt-in (RSX_FUSE=1, rsx_device.hip).
#define FUSE_UNITS 4
#define ACC_CONSTS_ENTRIES (65536 + 3)   // entries of the scalar-path table {(double)i, refine_rcp(i)}: Welford steps up to ACC_RCP_TABLE_MAX, frame merges up to 65536 samples
#ifndef TICKET_BATCH
#define TICKET_BATCH 4
#endif
struct FuseParams {
    Sample *ring;                       // [n_waves][FUSE_UNITS * WAVE]
    const double *tables;               // [n_tables, bins]
    double *fmean, *fvar; int32_t *fn;  // frame [nx, ny, frame_bins]
    double sensitivity;
    int32_t n_tables, bins, power, ny, frame_bins, slice_offset;
    int32_t lds_bytes, tables_in_lds;   // per-wave LDS region size; 1: the spectral tables fit behind the staged records
    const double *consts;               // [i] = {(double)i, refine_rcp(i)} (k_fill_acc_consts)
    int32_t passes, pass_spp;           // rsx_render_desc.passes: a pixel's spp = passes * pass_spp samples are `passes` chains, merged one after the other
};

// MODE 0: everything in one kernel. CSG scenes run two passes instead: MODE 1 has only the state-free CSG evaluator (csg_fast_hit), so
// it fits several waves per SIMD; rays it cannot finish (exact ties between operand roots, operands with mesh leaves) are listed in
// redo_mask and traced again by MODE 2, which carries the reference's stream merge (one wave per SIMD, usually nothing to do).
// STAGE_MIN: see mesh_trace_wave (1 for passes with more than one sample per pixel).
// The chains of the units a wave has gathered in its ring: Welford over each pixel's spp samples per bin, merged into the frame.
template <bool TABLES_IN_LDS, bool MULTI>      // MULTI: the pixel's samples are several passes (rsx_render_desc.passes), merged one after the other
__device__ __forceinline__ void fused_chains(const RSX_CONST_AS RenderParams *q, const FuseParams &fz, int u0, int u1, int u2, int u3, int n_units, uint32_t lds_base) {
    const int lane = threadIdx.x % WAVE;
    const int spp = q->spp, bins = fz.bins, ppu = WAVE / spp;             // pixels per unit
    const double *l_a = reinterpret_cast<const double *>(smem + lds_base), *l_w = l_a + FUSE_UNITS * WAVE;
    const int32_t *l_tab = reinterpret_cast<const int32_t *>(l_w + FUSE_UNITS * WAVE);
    const double *l_tables = reinterpret_cast<const double *>(l_tab + FUSE_UNITS * WAVE) + (spp + 2);
    const double scale = fz.power ? fz.sensitivity : 1.0;                 // x * 1.0 is x, bit for bit: one multiply instead of a branch per sample
    const double *tables = TABLES_IN_LDS ? l_tables : fz.tables;
    const int chains = n_units * ppu * bins;
    for (int c = lane; c < chains; c += WAVE) {
        const int pix = (int)((uint32_t)c / (uint32_t)bins), b = c - pix * bins;
        const int u = (int)((uint32_t)pix / (uint32_t)ppu), within = pix - u * ppu;
        const int unit_u = u == 0 ? u0 : u == 1 ? u1 : u == 2 ? u2 : u3;
        const UnitPixel px = unit_pixel(q, unit_u, within * spp);
        if (!px.valid) continue;
        const int base = u * WAVE + within * spp;
        // x = (a * table[bin]) * weight [* sensitivity] — optical/ray.pyx:391-393, observer.pyx:408; absorbers (table < 0) give 0
        auto value = [&](int i) {
            const int32_t table = l_tab[base + i];
            double x = table < 0 ? 0.0 : l_a[base + i] * tables[(table < 0 ? 0 : table) * bins + b];
            x = x * l_w[base + i];
            return x * scale;
        };
        if constexpr (!MULTI && TABLES_IN_LDS) {
            // The staged form (fused_flush): l_tab holds each sample's BYTE OFFSET of its table row in the wave's LDS copy — absorbers point at
            // the all-zero row behind the tables and carry a = +0.0, so x = (a * t) * weight * scale is (+0.0 * +0.0) * weight * scale, the
            // reference's product for an empty spectrum, without a select — and the step is written so that it costs what its arithmetic
            // costs: round 5 counted ~40 vector instructions per step where the recurrence needs 22 (a 32-bit multiply and a clamp for the
            // table address, compare + two selects for absorbers, six register moves of the operand rotation, the lane predicate of the range
            // test turned into a value and back). Two steps per trip: the operands of step i + 1 load into the registers step i - 1 used.
            const RSX_CONST_AS double *consts = (const RSX_CONST_AS double *)(unsigned long long)fz.consts;
            const uint32_t row0 = (uint32_t)(reinterpret_cast<const char *>(l_tables) - reinterpret_cast<const char *>(smem)) + (uint32_t)b * 8u;
            auto t_at = [&](int32_t off) { return *reinterpret_cast<const double *>(smem + row0 + (uint32_t)off); };
            const double *p_a = l_a + base, *p_w = l_w + base;
            const int32_t *p_o = l_tab + base;
            const lanemask act = pkt_mask(true);                              // the lanes that hold a chain
            // the frame cell is asked for now and merged after the chain (its round trip to HBM runs under the recurrence)
            const size_t f = ((size_t)px.ix * fz.ny + px.iy) * fz.frame_bins + fz.slice_offset + b;
            const double f_m = fz.fmean[f], f_v = fz.fvar[f];
            const int f_n = fz.fn[f];
            double m = ((p_a[0] * t_at(p_o[0])) * p_w[0]) * scale, v = 0;
            auto step = [&](int i, double a_c, double w_c, double t_c) {      // _add_sample (statsarray.pyx:743-776) for sample i of the chain
                double x = a_c * t_c;
                x = x * w_c;
                x = x * scale;
                const double dm = consts[2 * i], ym = consts[2 * i + 1], dn = consts[2 * i + 2], yn = consts[2 * i + 3];
                const double cc = i == 1 ? 1.0 : consts[2 * i - 2];
                const double pm = m, pv = v;
                const double n1 = x - pm;
                const double q1 = __builtin_fma(__builtin_fma(-dn, n1 * yn, n1), yn, n1 * yn);
                const double m1 = pm + q1;
                const double n2 = pv * cc + n1 * (x - m1);
                const double q2 = __builtin_fma(__builtin_fma(-dm, n2 * ym, n2), ym, n2 * ym);
                const double a1 = __builtin_fabs(n1), a2 = __builtin_fabs(n2);
                // exact_div's operand test as lane masks: their algebra and the branch are scalar
                const lanemask ok1 = (pkt_mask(a1 >= 0x1p-300) & pkt_mask(a1 <= 0x1p+300)) | mask_plus_zero(n1);
                const lanemask ok2 = (pkt_mask(a2 >= 0x1p-300) & pkt_mask(a2 <= 0x1p+300)) | mask_plus_zero(n2);
                if (__builtin_expect((ok1 & ok2) != act, 0)) {
                    m = pm + exact_div(n1, dn, yn, true);
                    v = exact_div(pv * cc + n1 * (x - m), dm, ym, true);
                } else { m = m1; v = q2; }
            };
            // operands of sample 1 in set A, the offset of sample 2; samples past the chain's end are clamped to its last (read, never used)
            const int last = spp - 1;
            auto at = [&](int i) { return i < last ? i : last; };
            double aA = p_a[at(1)], wA = p_w[at(1)], tA = t_at(p_o[at(1)]);
            int32_t o_n = p_o[at(2)];
            int i = 1;
            for (; i + 1 < spp; i += 2) {
                const double aB = p_a[at(i + 1)], wB = p_w[at(i + 1)], tB = t_at(o_n);
                const int32_t o_b = p_o[at(i + 2)];
                step(i, aA, wA, tA);
                aA = p_a[at(i + 2)]; wA = p_w[at(i + 2)]; tA = t_at(o_b);
                o_n = p_o[at(i + 3)];
                step(i + 1, aB, wB, tB);
            }
            if (i < spp) step(i, aA, wA, tA);
            if (v < 0) v = 0;                                                 // statsarray.pyx:649-650
            double mt, vt;
            int nt;
            combine_samples_uniform(f_m, f_v, f_n, m, v, spp, mt, vt, nt, consts, ACC_CONSTS_ENTRIES);
            fz.fmean[f] = mt; fz.fvar[f] = vt; fz.fn[f] = nt;
        } else if constexpr (!MULTI) {
            double m = value(0), v = 0;
            const RSX_CONST_AS double *consts = (const RSX_CONST_AS double *)(unsigned long long)fz.consts;
            // the operands of sample i + 1 (and the table id of sample i + 2, which the table read of i + 1 needs) are requested while the
            // recurrence of sample i runs: read at the point of use, a step was three LDS round trips in a row (table id -> a, table row ->
            // weight, behind a branch on the id) in front of twelve dependent f64 operations
            auto at = [&](int i) { return base + (i < spp ? i : spp - 1); };
            int32_t tab_n = l_tab[at(1)], tab_nn = l_tab[at(2)];
            double a_n = l_a[at(1)], w_n = l_w[at(1)], t_n = tables[(tab_n < 0 ? 0 : tab_n) * bins + b];
            for (int i = 1; i < spp; ++i) {                                   // _add_sample, as k_accumulate's step(): divisors over the scalar data path,
                const int32_t tab_c = tab_n;                                  // both quotients by exact_div's shortcut, one wave-level range test per step
                const double a_c = a_n, w_c = w_n, t_c = t_n;
                tab_n = tab_nn; tab_nn = l_tab[at(i + 2)];
                a_n = l_a[at(i + 1)]; w_n = l_w[at(i + 1)]; t_n = tables[(tab_n < 0 ? 0 : tab_n) * bins + b];
                double x = a_c * t_c;                                         // x = (a * table[bin]) * weight [* sensitivity], absorbers (table < 0) give 0
                x = tab_c < 0 ? 0.0 : x;
                x = x * w_c;
                x = x * scale;
                const double dm = consts[2 * i], ym = consts[2 * i + 1], dn = consts[2 * i + 2], yn = consts[2 * i + 3];
                const double cc = i == 1 ? 1.0 : consts[2 * i - 2];
                const double pm = m, pv = v;
                const double n1 = x - pm;
                const double q1 = __builtin_fma(__builtin_fma(-dn, n1 * yn, n1), yn, n1 * yn);
                const double m1 = pm + q1;
                const double n2 = pv * cc + n1 * (x - m1);
                const double q2 = __builtin_fma(__builtin_fma(-dm, n2 * ym, n2), ym, n2 * ym);
                const double a1 = __builtin_fabs(n1), a2 = __builtin_fabs(n2);
                const lanemask okm = ((pkt_mask(a1 >= 0x1p-300) & pkt_mask(a1 <= 0x1p+300)) | mask_plus_zero(n1)) & ((pkt_mask(a2 >= 0x1p-300) & pkt_mask(a2 <= 0x1p+300)) | mask_plus_zero(n2));
                if (__builtin_expect(okm != pkt_mask(true), 0)) {
                    m = pm + exact_div(n1, dn, yn, true);
                    v = exact_div(pv * cc + n1 * (x - m), dm, ym, true);
                } else { m = m1; v = q2; }
            }
            const size_t f = ((size_t)px.ix * fz.ny + px.iy) * fz.frame_bins + fz.slice_offset + b;
            if (v < 0) v = 0;                                                 // statsarray.pyx:649-650
            double mt, vt;
            int nt;
            combine_samples(fz.fmean[f], fz.fvar[f], fz.fn[f], m, v, spp, mt, vt, nt);
            fz.fmean[f] = mt; fz.fvar[f] = vt; fz.fn[f] = nt;
        } else {
            const RSX_CONST_AS double *consts = (const RSX_CONST_AS double *)(unsigned long long)fz.consts;
            const size_t f = ((size_t)px.ix * fz.ny + px.iy) * fz.frame_bins + fz.slice_offset + b;
            double fm = fz.fmean[f], fv = fz.fvar[f];
            int fcount = fz.fn[f];
            const int pass_spp = fz.pass_spp;
            for (int first = 0; first < spp; first += pass_spp) {             // (one pass, or the rsx_render_desc.passes of this call in their order)
                double m = value(first), v = 0;
                for (int i = 1; i < pass_spp; ++i) {                          // _add_sample, as k_accumulate's step(): divisors over the scalar data path,
                    const double x = value(first + i);                        // both quotients by exact_div's shortcut, one wave-level range test per step
                    const double dm = consts[2 * i], ym = consts[2 * i + 1], dn = consts[2 * i + 2], yn = consts[2 * i + 3];
                    const double cc = i == 1 ? 1.0 : consts[2 * i - 2];
                    const double pm = m, pv = v;
                    const double n1 = x - pm;
                    const double q1 = __builtin_fma(__builtin_fma(-dn, n1 * yn, n1), yn, n1 * yn);
                    const double m1 = pm + q1;
                    const double n2 = pv * cc + n1 * (x - m1);
                    const double q2 = __builtin_fma(__builtin_fma(-dm, n2 * ym, n2), ym, n2 * ym);
                    const double a1 = __builtin_fabs(n1), a2 = __builtin_fabs(n2);
                    const lanemask okm = ((pkt_mask(a1 >= 0x1p-300) & pkt_mask(a1 <= 0x1p+300)) | mask_plus_zero(n1)) & ((pkt_mask(a2 >= 0x1p-300) & pkt_mask(a2 <= 0x1p+300)) | mask_plus_zero(n2));
                    if (__builtin_expect(okm != pkt_mask(true), 0)) {
                        m = pm + exact_div(n1, dn, yn, true);
                        v = exact_div(pv * cc + n1 * (x - m), dm, ym, true);
                    } else { m = m1; v = q2; }
                }
                if (v < 0) v = 0;                                             // statsarray.pyx:649-650
                double mt, vt;
                int nt;
                combine_samples_uniform(fm, fv, fcount, m, v, pass_spp, mt, vt, nt, consts, ACC_CONSTS_ENTRIES);
                fm = mt; fv = vt; fcount = nt;
            }
            fz.fmean[f] = fm; fz.fvar[f] = fv; fz.fn[f] = fcount;
        }
    }
}

template <bool MULTI = false>
__device__ __forceinline__ void fused_flush(const RSX_CONST_AS RenderParams *q, const FuseParams &fz, const Sample *ring, int u0, int u1, int u2, int u3, int n_units,
                                            uint32_t lds_base) {
    const int lane = threadIdx.x % WAVE;
    const int spp = q->spp;
    double *l_a = reinterpret_cast<double *>(smem + lds_base), *l_w = l_a + FUSE_UNITS * WAVE;
    int32_t *l_tab = reinterpret_cast<int32_t *>(l_w + FUSE_UNITS * WAVE);
    double *l_rcp = reinterpret_cast<double *>(l_tab + FUSE_UNITS * WAVE), *l_tables = l_rcp + (spp + 2);
    // the ring was written by this wave's own lanes (earlier units): same CU, same vector L1 — workgroup-scope ordering is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // (one pass per call with the tables in LDS: the chains read row OFFSETS, and absorbers the zero row behind the tables — fused_chains)
    const bool offsets = !MULTI && fz.tables_in_lds;
    const int32_t row_bytes = fz.bins * 8, zero_row = fz.n_tables * row_bytes;
    for (int u = 0; u < n_units; ++u) {
        const Sample smp = ring[u * WAVE + lane];
        const bool absorber = smp.table < 0;
        l_a[u * WAVE + lane] = offsets && absorber ? 0.0 : smp.a; l_w[u * WAVE + lane] = smp.weight;
        l_tab[u * WAVE + lane] = offsets ? (absorber ? zero_row : smp.table * row_bytes) : smp.table;
    }
    if (fz.tables_in_lds) {
        for (int e = lane; e < fz.n_tables * fz.bins; e += WAVE) l_tables[e] = fz.tables[e];
        if (offsets) for (int e = lane; e < fz.bins; e += WAVE) l_tables[fz.n_tables * fz.bins + e] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (fz.tables_in_lds) fused_chains<true, MULTI>(q, fz, u0, u1, u2, u3, n_units, lds_base);
    else fused_chains<false, MULTI>(q, fz, u0, u1, u2, u3, n_units, lds_base);
    __builtin_amdgcn_wave_barrier();
}

// The flush as a real call (packet kernel): once per FUSE_UNITS units, everything it needs is wave-uniform and re-read from the
// kernel-argument segment inside (the render parameters and the FuseParams lie there, second and fifth argument of k_render_trace) —
// inlined into the unit loop its chains' registers competed with the walk's and cost the kernel sixteen more spilled registers.
// (MULTI — several passes per call — is a kernel of its own: as a run-time branch in here the larger callee cost the one-pass kernel
// 22 more spilled registers around the call and configs[2] 27.2 -> 28.4 ms)
// (the kernel-argument layout those offsets assume — k_render_trace(DScene, RenderParams, Sample *, unsigned long long *, FuseParams): every
// argument 8-byte aligned, so rp lies at align8(sizeof(DScene)) and fz two pointers behind align8(end of rp))
static_assert(alignof(DScene) <= 8 && alignof(RenderParams) == 8 && alignof(FuseParams) == 8 && sizeof(Sample *) == 8 && sizeof(unsigned long long *) == 8,
              "fused_flush_call / k_render_trace locate rp and fz in the kernel-argument segment by these alignments");
template <bool MULTI>
__device__ __attribute__((noinline)) void fused_flush_call(unsigned long long rp_bits_, const Sample *ring_, int u0_, int u1_, int u2_, int u3_, int n_units_, uint32_t lds_base_) {
    const Sample *ring = (const Sample *)pkt_uniform64((unsigned long long)ring_);
    const int u0 = __builtin_amdgcn_readfirstlane(u0_), u1 = __builtin_amdgcn_readfirstlane(u1_), u2 = __builtin_amdgcn_readfirstlane(u2_),
              u3 = __builtin_amdgcn_readfirstlane(u3_), n_units = __builtin_amdgcn_readfirstlane(n_units_);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base_);
    const unsigned long long rp_bits = pkt_uniform64(rp_bits_);           // where the render parameters lie in the kernel-argument segment
    const RSX_CONST_AS RenderParams *q = (const RSX_CONST_AS RenderParams *)rp_bits;
    const RSX_CONST_AS FuseParams *fq = (const RSX_CONST_AS FuseParams *)(rp_bits + ((sizeof(RenderParams) + 7) & ~(size_t)7) + 16);   // (+ samples, ticket)
    FuseParams fz;
    fz.ring = fq->ring; fz.tables = fq->tables; fz.fmean = fq->fmean; fz.fvar = fq->fvar; fz.fn = fq->fn; fz.sensitivity = fq->sensitivity;
    fz.n_tables = fq->n_tables; fz.bins = fq->bins; fz.power = fq->power; fz.ny = fq->ny; fz.frame_bins = fq->frame_bins; fz.slice_offset = fq->slice_offset;
    fz.lds_bytes = fq->lds_bytes; fz.tables_in_lds = fq->tables_in_lds; fz.consts = fq->consts; fz.passes = fq->passes; fz.pass_spp = fq->pass_spp;
    fused_flush<MULTI>(q, fz, ring, u0, u1, u2, u3, n_units, lds_base);
}

// PACKET: the unit's rays walk the trees together (dev_packet.hpp) — passes whose units hold a few pixels' samples.
template <bool CSG, int MODE = 0, int STAGE_MIN = RSX_STAGE_MIN, int FUSED = 0, bool PACKET = false>     // FUSED: 0 no, 1 yes, 2 yes with several passes per call (packet kernel)
__global__ __launch_bounds__(WG_THREADS, !CSG ? (PACKET ? RSX_PACKET_MIN_WAVES : RSX_MIN_WAVES_PER_SIMD) : MODE == 1 ? (PACKET ? RSX_PACKET_CSG_MIN_WAVES : RSX_CSGFAST_MIN_WAVES) : RSX_CSG_MIN_WAVES)
void k_render_trace(DScene sc, RenderParams rp, Sample *samples, unsigned long long *ticket, FuseParams fz) {
    Stack st, ms;
    if constexpr (PACKET) wave_stacks_packet(sc, st, ms); else wave_stacks(sc, st, ms);
    const int lane = threadIdx.x % WAVE;
    int fuse_u0 = 0, fuse_u1 = 0, fuse_u2 = 0, fuse_u3 = 0, fuse_n = 0;           // wave-uniform: the units whose records wait in the ring
    Sample *fuse_ring = nullptr;
    if constexpr (FUSED != 0) fuse_ring = fz.ring + ((size_t)blockIdx.x * (WG_THREADS / WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE))) * (FUSE_UNITS * WAVE);
    NodeSt csg_state[CSG && MODE != 1 ? CSG_MAX_SLOTS : 1];
    long long redo_unit = (long long)blockIdx.x * (WG_THREADS / WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));   // MODE 2: units are strided over the waves
    // Work is handed out from eight longest-first lists, one per XCD (k_order_units): a wave drains the list of the XCD it runs on
    // first, so one L2 only ever sees an eighth of the image's geometry, and steals from the other lists when its own is empty.
    const int my_xcd = xcc_id();
    int victim = -1;                   // -1: the shared list of expensive units comes first (longest-processing-time-first), then the XCD lists
    int32_t batch_next = 0, batch_end = 0;     // (wave-uniform; list positions stay below 2^26) entries of the current ticket not yet rendered
    for (;;) {
        // Render parameters are re-read from the kernel-argument segment at every use site of the unit loop (the pointer is
        // laundered through an empty asm): hoisted out of the loop, the camera matrix and friends sat in ~30 vector registers
        // through the whole traversal, where registers decide how many waves fit a SIMD. `rp` itself is only named for its layout.
        (void)rp;
        // (`rp` is read where it lies in the kernel-argument segment — second argument, after `sc` — so that no private copy is made)
        unsigned long long rp_bits = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DScene) + 7) & ~(size_t)7);
        asm volatile("" : "+s"(rp_bits));
        const RSX_CONST_AS RenderParams *q = (const RSX_CONST_AS RenderParams *)rp_bits;
        long long tk = -1;
        unsigned long long redo_lanes = ~0ULL;
        int unit;
        if constexpr (MODE == 2) {
            const long long n_units = q->seg[9];
            // (the wave looks at the masks of its next 64 units at once, one per lane: nearly every mask is zero — a grazing ray here and
            // there — and one dependent scalar load per unit made the launch 0.12 ms of a 4.5 ms configs[3] pass with nothing to do)
            const long long stride = (long long)gridDim.x * (WG_THREADS / WAVE);
            while (redo_unit < n_units) {
                const long long mine = redo_unit + (long long)lane * stride;
                const unsigned long long found = __ballot(mine < n_units && q->redo_mask[mine] != 0ULL);
                if (found) { redo_unit += (long long)(__ffsll((long long)found) - 1) * stride; break; }
                redo_unit += (long long)WAVE * stride;
            }
            if (redo_unit >= n_units) break;
            unit = __builtin_amdgcn_readfirstlane((int)redo_unit);
            redo_lanes = q->redo_mask[unit];
            redo_unit += (long long)gridDim.x * (WG_THREADS / WAVE);
        } else {
            // A ticket is TICKET_BATCH consecutive entries of a list: the counters are one address per list, an atomic on a contended
            // address costs the wave a microsecond, and a pass of 4.2 M units through nine counters was bounded by them alone when the
            // units were cheap (7.8 ms for ray generation + record store without any traversal). Large passes only: a small pass is
            // tail-bound and hands its units out one by one.
            // (pinned to scalar registers: left to itself the compiler kept these counters as 64-bit values in six vector registers
            // through the walk)
            batch_next = __builtin_amdgcn_readfirstlane(batch_next); batch_end = __builtin_amdgcn_readfirstlane(batch_end);
            if (batch_next < batch_end) tk = batch_next++;
            else while (victim < 8) {
                const int list = victim < 0 ? 0 : 1 + ((my_xcd + victim) & 7);
                const long long begin = q->seg[list], end = q->seg[list + 1];
                const unsigned long long take = (long long)q->seg[9] > (long long)RSX_LPT_MAX_UNITS ? (unsigned long long)TICKET_BATCH : 1ULL;
                unsigned long long mine = 0;
                if (lane == 0) mine = atomicAdd(ticket + 16 * list, take);
                const long long got = begin + (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mine >> 32)) << 32) |
                                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)mine));
                if (got < end) { tk = got; batch_next = (int32_t)got + 1; batch_end = (int32_t)(got + (long long)take < end ? got + (long long)take : end); break; }
                ++victim;
            }
            if (tk < 0) break;
            unit = __builtin_amdgcn_readfirstlane((int)(q->unit_order[tk] & 0x3ffffffu));   // wave-uniform: keep it scalar
        }
        // (unit timestamps — rsx_debug_unit_times — are a per-lane-walk tuning aid: a pass that has them never takes the packet kernel)
        const unsigned long long t_start = RSX_UNIT_STAMPS && !PACKET && q->unit_times ? wall_clock64() : 0ULL;
#if RSX_PHASE_PROF == 2
        const unsigned long long ph2_u0 = clock64();
#endif
#if RSX_PHASE_PROF
        unsigned long long phase_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#elif RSX_UTIL_PROF
        unsigned long long *phase_acc = q->unit_times ? q->unit_times + 12 * (long long)unit + 3 : nullptr;   // caller zeroes the buffer
#else
        unsigned long long *phase_acc = nullptr;
#endif
        const UnitPixel px = unit_pixel(q, unit, lane);
        const bool valid = px.valid && ((redo_lanes >> lane) & 1ULL);
        // PinholeCamera._generate_rays, pinhole.pyx:169-204 + RectangleSampler3D.sample, surface3d.pyx:197-198
        double u1, u2;
        if (q->rng_mode == RSX_RNG_STREAM) { u1 = q->uniforms[2 * (px.k * q->spp + px.s)]; u2 = q->uniforms[2 * (px.k * q->spp + px.s) + 1]; }
        else philox2(q->seed, (uint64_t)px.ix * (uint64_t)q->cam.ny + (uint64_t)px.iy, q->sample_offset + (uint64_t)px.s, u1, u2);
        Ray r;
        double weight;
        camera_ray(q, px.ix, px.iy, u1, u2, r, weight);
        Hit hit;
        uint32_t work = 0;
        if constexpr (PACKET) {
            // the projection weight goes to its record NOW (its two registers are then free through the walk); a and table follow the walk
            if constexpr (FUSED != 0) fuse_ring[fuse_n * WAVE + lane_here()].weight = weight;
            else if (px.valid) samples[px.slot * q->spp + px.s].weight = weight;
            r.ox = readlane_f64(r.ox, 0); r.oy = readlane_f64(r.oy, 0); r.oz = readlane_f64(r.oz, 0);     // (the pinhole: one origin, kept in scalar registers)
        }
#if RSX_PHASE_PROF == 2
        const unsigned long long ph2_w0 = clock64();
        phase_acc[0] = ph2_w0 - ph2_u0;
#endif
        bool got;
        bool packet_redo = false;                                          // PACKET, CSG: this lane met a solid the state-free evaluator could not answer
        if constexpr (PACKET) { static_assert(!CSG || MODE == 1, "CSG scenes: the packet walk is the fast pass; the redo pass walks per lane");
            unsigned long long sc_bits = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();       // (`sc` is the first kernel argument)
            asm volatile("" : "+s"(sc_bits));
            const PScene scq = (PScene)sc_bits; 
#ifdef RSX_PKT_PROF
            uint32_t pkc[PKC_N];
            for (int c = 0; c < PKC_N; ++c) pkc[c] = 0;
            pkc[PKC_UNITS] = 1;
            got = world_trace_packet<CSG>(valid, scq, r, st, ms, hit, work, packet_redo, pkc);
            if (lane == 0) for (int c = 0; c < PKC_N; ++c) atomicAdd(&g_pkt[c], (unsigned long long)pkc[c]);
#elif defined(PKT_ABLATE_TRACE)
            got = false; hit.prim = -1;                                    // (timing ablation: ray generation and the record store alone)
#else
            got = world_trace_packet<CSG>(valid, scq, r, st, ms, hit, work, packet_redo);
#endif
        }
        else got = world_trace_wave<CSG, MODE == 1, STAGE_MIN, true>(valid, sc, r, st, ms, csg_state, hit, work, phase_acc);
#if RSX_PHASE_PROF == 2
        const unsigned long long ph2_w1 = clock64();
        phase_acc[1] = ph2_w1 - ph2_w0;
#endif
        // the unit's pixel bookkeeping is recomputed rather than carried through the traversal (`unit` is laundered so that the
        // compiler cannot merge this with the computation above)
        // (readfirstlane first: both are wave-uniform by construction, but the "+s" constraint fails in the backend — "illegal VGPR to SGPR
        // copy" — whenever the optimiser has decided to carry one of them in a vector register across the walk)
        unit = __builtin_amdgcn_readfirstlane(unit);
        rp_bits = pkt_uniform64(rp_bits);
        asm volatile("" : "+s"(unit));
        asm volatile("" : "+s"(rp_bits));
        const RSX_CONST_AS RenderParams *q2 = (const RSX_CONST_AS RenderParams *)rp_bits;
        const bool redo = MODE == 1 && ((work >> 31) != 0 || packet_redo);
        work &= 0x7fffffffu;
        if (q2->measure_cost && MODE != 2 && lane == 0) {
            unsigned long long c = (unsigned long long)work;
            if (c > 0x7fffffffULL) c = 0x7fffffffULL;
            q2->unit_cost[unit] = (uint32_t)c;
        }
        if (RSX_UNIT_STAMPS && !PACKET && q2->unit_times && lane == 0) {
            q2->unit_times[12 * unit] = t_start;
            q2->unit_times[12 * unit + 1] = wall_clock64();
            q2->unit_times[12 * unit + 2] = ((unsigned long long)blockIdx.x << 8) | (threadIdx.x / WAVE);
#if RSX_PHASE_PROF
#if RSX_PHASE_PROF == 2
            phase_acc[6] = clock64() - ph2_u0;                 // (the store of the sample record follows: not included)
#endif
            for (int ph = 0; ph < 8; ++ph) q2->unit_times[12 * unit + 3 + ph] = phase_acc[ph];
#endif
        }
        const UnitPixel px2 = unit_pixel(q2, unit, lane);
        if constexpr (MODE == 1) {
            const unsigned long long again = __ballot(redo && px2.valid);
            if (lane == 0) q2->redo_mask[unit] = again;
            if (redo) continue;
        }
        if constexpr (MODE == 2) { if (!((q2->redo_mask[unit] >> lane) & 1ULL)) continue; }
        if (FUSED == 0 && !px2.valid) continue;
        Sample smp;
        smp.a = 0.0; smp.weight = PACKET ? 0.0 : weight; smp.table = -1; smp.pad = 0;
        if (got) {                                                                 // optical/ray.pyx:391-393
            const rsx_primitive &p = sc.prims[hit.prim];
            const rsx_material mat = q2->materials[p.material];
            if (mat.type == RSX_MAT_UNIFORM_EMITTER) { smp.a = mat.scale; smp.table = mat.table; }   // emitter/uniform.pyx:67-81
            else if (mat.type == RSX_MAT_DEBUG_LIGHT) {                      // debug.pyx:67-79
                if (mat.scale != 0.0) {
                    Geom g;
                    finalise<CSG>(sc, r, hit, g);
                    double lx, ly, lz;
                    xform_vector(p.to_local, -mat.light_dir[0], -mat.light_dir[1], -mat.light_dir[2], lx, ly, lz);
                    const double dot = lx * g.normal[0] + ly * g.normal[1] + lz * g.normal[2];
                    smp.a = mat.scale * (dot > 0 ? dot : 0.0);
                    smp.table = mat.table;
                }
            }
        }
        if constexpr (FUSED != 0 && PACKET) {
            Sample *rec = fuse_ring + fuse_n * WAVE + lane_here();         // (weight: stored before the walk)
            rec->a = smp.a; rec->table = smp.table; rec->pad = 0;
        }
        if constexpr (FUSED != 0) {
            if constexpr (!PACKET) fuse_ring[fuse_n * WAVE + lane] = smp;  // (lanes of pixels outside the frame write a record nobody reads)
            if (fuse_n == 0) fuse_u0 = unit; else if (fuse_n == 1) fuse_u1 = unit; else if (fuse_n == 2) fuse_u2 = unit; else fuse_u3 = unit;
            if (++fuse_n == FUSE_UNITS) {
                if constexpr (PACKET) fused_flush_call<FUSED == 2>(rp_bits, fuse_ring, fuse_u0, fuse_u1, fuse_u2, fuse_u3, FUSE_UNITS, st.lds_t);
                else fused_flush(q2, fz, fuse_ring, fuse_u0, fuse_u1, fuse_u2, fuse_u3, FUSE_UNITS, st.lds_t);
                fuse_n = 0;
            }
        } else if constexpr (PACKET) { Sample *rec = samples + px2.slot * q2->spp + px2.s; rec->a = smp.a; rec->table = smp.table; rec->pad = 0; }
        else samples[px2.slot * q2->spp + px2.s] = smp;
    }
    if constexpr (FUSED != 0) {
        if (fuse_n) {
            const unsigned long long rp_bits = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DScene) + 7) & ~(size_t)7);
            if constexpr (PACKET) fused_flush_call<FUSED == 2>(rp_bits, fuse_ring, fuse_u0, fuse_u1, fuse_u2, fuse_u3, fuse_n, st.lds_t);
            else fused_flush((const RSX_CONST_AS RenderParams *)rp_bits, fz, fuse_ring, fuse_u0, fuse_u1, fuse_u2, fuse_u3, fuse_n, st.lds_t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Paths: transparent boundaries, volume emission and Lambert scattering
// ---------------------------------------------------------------------------------------------------
// NullMaterial / UniformVolumeEmitter surfaces let the ray through (NullSurface.evaluate_surface, material.pyx:118-147: daughter
// ray from the far side of the boundary, same direction, depth unchanged, keep_alive); every segment of the path adds the emission
// of the volume emitters that contain the segment's origin times the segment length in the emitter's space (Ray._sample_volumes,
// ray.pyx:422-455; HomogeneousVolumeEmitter.evaluate_volume, homogeneous.pyx:55-102); a Lambert surface scatters one cosine-weighted
// daughter ray (lambert.pyx:76-104 under ContinuousBSDF.evaluate_surface, material.pyx:286-361; Russian roulette per ray.pyx:382-388).
//
// The reference evaluates a path innermost ray first while its recursion unwinds, per spectral bin. Here the trace kernel walks the
// path forward and leaves a list of PathTerm records next to the ray's Sample; k_accumulate replays the list backwards per bin with
// the reference's own operations in the reference's own order, so a frame equals the oracle's bit for bit. The list lives in
// 16-slot blocks: the first block of every ray is preallocated (block id = record index); a ray that needs more takes blocks
// from a shared arena (one atomic per 15 terms), each linked back to its predecessor through slot 0. Scenes without such
// materials never run this kernel (k_render_trace is untouched).
struct PathTerm {
    double a, b;               // VOL: segment length, emitter scale.  LAMBERT: pdf, 1 / pdf.  ATTEN: segment length
    int32_t table;             // spectral table row (emission / reflectivity).  LINK: id of the previous block
    int32_t kind;
};
enum { TERM_VOL = 0, TERM_LAMBERT = 1, TERM_LAMBERT_NORM = 2, TERM_LINK = 3, TERM_NORM = 4, TERM_ATTEN = 5 };
// LAMBERT_NORM: the daughter survived roulette, its result is scaled first.  NORM: roulette scaling alone (daughter of a dielectric
// surface, which itself leaves the spectrum unchanged).  ATTEN: a = world-space segment length inside a dielectric, table = transmission
#define PATH_BLOCK 16
#define PATH_MAX_SEGMENTS (1 << 20)        // guard against a path that never ends (each Lambert bounce and each null surface is a segment)
#define PATH_VOL_OVERLAP 4                 // volume emitters that may contain one point

// sin and cos of phi in [0, 2 pi]: the oracle's portable_sincos (oracle/rsx_oracle.c) operation for operation — quadrant by Cody-Waite
// reduction against a two-term pi/2, minimax polynomials on [-pi/4, pi/4], IEEE + and * only (no FMA: -ffp-contract=off).
__device__ __forceinline__ void portable_sincos(double phi, double &sn, double &cs) {
    const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11, TWO_OVER_PI = 6.36619772367581382433e-01;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double kf = floor(phi * TWO_OVER_PI + 0.5);
    const double r = (phi - kf * PIO2_HI) - kf * PIO2_LO;
    const double z = r * r;
    const double ps = r + (r * z) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double pc = (1.0 - 0.5 * z) + (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    switch ((int)kf & 3) {
    case 0: sn = ps; cs = pc; break;
    case 1: sn = pc; cs = -ps; break;
    case 2: sn = -ps; cs = -pc; break;
    default: sn = -pc; cs = ps; break;
    }
}

// asin on [0, 1]: the oracle's portable_asin operation for operation (rational approximation below 0.5, half-angle identity with a
// split square root above).
__device__ __forceinline__ double portable_asin(double x) {
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, PIO4_HI = 7.85398163397448278999e-01;
    const double P0 = 1.66666666666666657415e-01, P1 = -3.25565818622400915405e-01, P2 = 2.01212532134862925881e-01,
                 P3 = -4.00555345006794114027e-02, P4 = 7.91534994289814532176e-04, P5 = 3.47933107596021167570e-05;
    const double Q1 = -2.40339491173441421878e+00, Q2 = 2.02094576023350569471e+00, Q3 = -6.88283971605453293030e-01,
                 Q4 = 7.70381505559019352791e-02;
    if (x >= 1.0) return x * PIO2_HI + x * PIO2_LO;
    if (x < 0.5) {
        if (x < 7.450580596923828e-09) return x;
        const double t = x * x;
        const double p = t * (P0 + t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5)))));
        const double q = 1.0 + t * (Q1 + t * (Q2 + t * (Q3 + t * Q4)));
        return x + x * (p / q);
    }
    double w = 1.0 - x;
    double t = w * 0.5;
    double p = t * (P0 + t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5)))));
    double q = 1.0 + t * (Q1 + t * (Q2 + t * (Q3 + t * Q4)));
    const double s = sqrt(t);
    if (x >= 0.975) {
        w = p / q;
        t = PIO2_HI - (2.0 * (s + s * w) - PIO2_LO);
    } else {
        w = __longlong_as_double(__double_as_longlong(s) & (long long)0xFFFFFFFF00000000ULL);
        const double c = (t - w * w) / (s + w);
        const double r = p / q;
        p = 2.0 * s * r - (PIO2_LO - 2.0 * c);
        q = PIO4_HI - 2.0 * w;
        t = PIO4_HI - (p - q);
    }
    return t;
}

// x ** y for x > 0: the oracle's portable_pow (oracle/rsx_oracle.c) operation for operation — IEEE + - * / only, double-double
// logarithm (2 atanh((m - 1) / (m + 1)) + e ln 2), Dekker product with y, Cody-Waite reduction and a degree-14 Taylor polynomial for
// the exponential. The Beer-Lambert attenuation of a tinted dielectric (dielectric.pyx:325-326) is therefore the same bits on both
// sides; against a correctly rounded pow it is good to about one unit in the last place.
__device__ __forceinline__ void dd_two_sum(double a, double b, double &s, double &e) { const double t = a + b; const double bb = t - a; e = (a - (t - bb)) + (b - bb); s = t; }
__device__ __forceinline__ void dd_split(double a, double &hi, double &lo) { const double c = 134217729.0 * a; const double h = c - (c - a); hi = h; lo = a - h; }
__device__ __forceinline__ void dd_two_prod(double a, double b, double &p, double &e) {
    double ah, al, bh, bl;
    dd_split(a, ah, al); dd_split(b, bh, bl);
    const double pr = a * b;
    e = ((ah * bh - pr) + ah * bl + al * bh) + al * bl; p = pr;
}
__device__ __noinline__ double portable_pow(double x, double y) {
    if (!(x > 0.0) || !(x < INFINITY) || !(y == y) || !(fabs(y) < INFINITY)) return pow(x, y);
    if (y == 0.0 || x == 1.0) return 1.0;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff);
    if (e == 0) { const double xs = x * 18014398509481984.0; bits = (unsigned long long)__double_as_longlong(xs); e = (int)((bits >> 52) & 0x7ff) - 54; }
    e -= 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)bits);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double num = m - 1.0;
    double den_h, den_l; dd_two_sum(m, 1.0, den_h, den_l);
    const double s_h = num / den_h;
    double ph, pl; dd_two_prod(s_h, den_h, ph, pl);
    const double s_l = (((num - ph) - pl) - s_h * den_l) / den_h;
    double z_h, z_l; dd_two_prod(s_h, s_h, z_h, z_l);
    z_l += 2.0 * s_h * s_l;
    double c_h, c_l; dd_two_prod(s_h, z_h, c_h, c_l);
    c_l += s_h * z_l + s_l * z_h;
    const double THIRD_H = 3.33333333333333314830e-01, THIRD_L = 1.85037170770859413132e-17;
    double q_h, q_l; dd_two_prod(c_h, THIRD_H, q_h, q_l);
    q_l += c_h * THIRD_L + c_l * THIRD_H;
    const double z = z_h;
    const double poly = 1.0 / 5.0 + z * (1.0 / 7.0 + z * (1.0 / 9.0 + z * (1.0 / 11.0 + z * (1.0 / 13.0 + z * (1.0 / 15.0 + z * (1.0 / 17.0
                        + z * (1.0 / 19.0 + z * (1.0 / 21.0 + z * (1.0 / 23.0)))))))));
    const double tail = 2.0 * ((c_h * z) * poly) + 2.0 * (q_l + s_l);
    double a_h, a_l; dd_two_sum(2.0 * s_h, 2.0 * q_h, a_h, a_l);
    double l_h, l_l; dd_two_sum(a_h, a_l + tail, l_h, l_l);
    const double LN2_H = 6.93147180559945286227e-01, LN2_L = 2.31904681384629955842e-17;
    double eh, el; dd_two_prod((double)e, LN2_H, eh, el);
    el += (double)e * LN2_L;
    double t_h, t_l; dd_two_sum(eh, l_h, t_h, t_l);
    t_l += el + l_l;
    double L_h, L_l; dd_two_sum(t_h, t_l, L_h, L_l);
    double p_h, p_l; dd_two_prod(y, L_h, p_h, p_l);
    p_l += y * L_l;
    if (p_h > 709.8) return INFINITY;
    if (p_h < -745.2) return 0.0;
    const double LN2_CW_H = 6.93147180369123816490e-01, LN2_CW_L = 1.90821492927058770002e-10;
    const double kf = floor(p_h * 1.44269504088896338700e+00 + 0.5);
    const double r = ((p_h - kf * LN2_CW_H) - kf * LN2_CW_L) + p_l;
    const double ex = 1.0 + r * (1.0 + r * (1.0 / 2.0 + r * (1.0 / 6.0 + r * (1.0 / 24.0 + r * (1.0 / 120.0 + r * (1.0 / 720.0 + r * (1.0 / 5040.0
                      + r * (1.0 / 40320.0 + r * (1.0 / 362880.0 + r * (1.0 / 3628800.0 + r * (1.0 / 39916800.0 + r * (1.0 / 479001600.0
                      + r * (1.0 / 6227020800.0 + r * (1.0 / 87178291200.0))))))))))))));
    const int k = (int)kf;
    const int k1 = k / 2, k2 = k - k1;
    const double f1 = __longlong_as_double((long long)((unsigned long long)(k1 + 1023) << 52));
    const double f2 = __longlong_as_double((long long)((unsigned long long)(k2 + 1023) << 52));
    return (ex * f1) * f2;
}

// ImportanceManager.sample (world.pyx:150-188): pick = the selection uniform, (ua, ub) = the direction pair in the reference's draw order
#ifndef RSX_IMPORTANT_INLINE
#define RSX_IMPORTANT_INLINE 1
#endif
#if RSX_IMPORTANT_INLINE
#define RSX_IMP_INLINE __forceinline__
#else
#define RSX_IMP_INLINE
#endif
// ImportanceManager.sample in two halves, so that the caller finds sin / cos of the azimuth ONCE for the lanes that sample an important
// sphere and the lanes that sample the cosine lobe (both are 2 pi times one of the draws): important_pick chooses the sphere and says
// which draw the azimuth takes (`cone`: the first; a point inside the sphere, like the cosine lobe, the second); important_direction
// finishes with sin / cos of that azimuth. Same operations per lane as the one-piece form had.
struct ImportantPick {
    double dx, dy, dz, distance, radius;
    bool cone;
};
__device__ RSX_IMP_INLINE ImportantPick important_pick(const rsx_important_sphere *spheres, int n, double ox, double oy, double oz, double pick) {
    int index = 0;
    while (index < n - 1 && !(pick < spheres[index].cdf)) ++index;            // find_index(cdf, u) + 1
    const rsx_important_sphere sp = spheres[index];
    ImportantPick k;
    k.dx = sp.centre[0] - ox; k.dy = sp.centre[1] - oy; k.dz = sp.centre[2] - oz;
    k.distance = sqrt(k.dx * k.dx + k.dy * k.dy + k.dz * k.dz);
    k.radius = sp.radius;
    k.cone = !(k.distance == 0 || k.distance < sp.radius);
    return k;
}
__device__ RSX_IMP_INLINE void important_direction(const ImportantPick &k, double ua, double ub, double sn, double cs, double &wx, double &wy, double &wz) {
    double dx = k.dx, dy = k.dy, dz = k.dz;
    if (!k.cone) {                                                            // vector_sphere (azimuth 2 pi ub)
        const double z = 1.0 - 2.0 * ua;
        const double r2 = 1.0 - z * z;
        const double r = sqrt(r2 > 0 ? r2 : 0);
        wx = r * cs; wy = r * sn; wz = z;
        return;
    }
    const double angular_radius = portable_asin(k.radius / k.distance);
    double theta = angular_radius * 180 / M_PI;                               // vector_cone_uniform(degrees)
    theta *= 0.017453292519943295;
    double cos_theta, unused;                                                 // (azimuth 2 pi ua)
    portable_sincos(theta, unused, cos_theta);
    const double z = ub * (1 - cos_theta) + cos_theta;
    const double r2 = 1.0 - z * z;
    const double r = sqrt(r2 > 0 ? r2 : 0);
    const double sx = r * cs, sy = r * sn, sz = z;
    normalise3(dx, dy, dz);
    double nx = dx, ny = dy, nz = dz;                                         // direction.orthogonal()
    normalise3(nx, ny, nz);
    double vx = 1, vy = 0, vz = 0;
    if (fabs(nx * vx + ny * vy + nz * vz) > 0.5) { vx = 0; vy = 1; }
    const double m = nx * vx + ny * vy + nz * vz;
    double ux = vx - m * nx, uy = vy - m * ny, uz = vz - m * nz;
    normalise3(ux, uy, uz);
    const double rx = uy * dz - dy * uz, ry = uz * dx - dz * ux, rz = ux * dy - dx * uy;   // the cimported rotate_basis: up.cross(forward)
    wx = rx * sx + ux * sy + dx * sz;
    wy = ry * sx + uy * sy + dy * sz;
    wz = rz * sx + uz * sy + dz * sz;
}

// ImportanceManager.pdf (world.pyx:190-230)
__device__ RSX_IMP_INLINE double important_pdf(const rsx_important_sphere *spheres, int n, double ox, double oy, double oz, double wx, double wy, double wz) {
    double pdf_all = 0;
    for (int i = 0; i < n; ++i) {
        const rsx_important_sphere sp = spheres[i];
        double ax = sp.centre[0] - ox, ay = sp.centre[1] - oy, az = sp.centre[2] - oz;
        const double distance = sqrt(ax * ax + ay * ay + az * az);
        double solid_angle;
        if (distance == 0 || distance < sp.radius) solid_angle = 4 * M_PI;
        else {
            const double t = sp.radius / distance;
            const double angular_radius_cos = sqrt(1 - t * t);
            normalise3(ax, ay, az);
            if (wx * ax + wy * ay + wz * az < angular_radius_cos) continue;
            solid_angle = 2 * M_PI * (1 - angular_radius_cos);
        }
        const double pdf_sphere = 1 / solid_angle;
        pdf_all += sp.weight * pdf_sphere;
    }
    return pdf_all;
}

// world.contains(point) in leaf order (kdtree3d.pyx:736-792, kdtree.pyx:126-162): calls f(primitive index) for every world
// primitive that passes `want` (a side-effect-free filter evaluated BEFORE the containment test, which for a mesh is a ray cast) and
// whose bounding box and surface contain the point
// FASTONLY (fast pass of a CSG scene): CSG primitives are tested with the flattened program; one without a program makes the
// caller abandon the path to the redo pass (needs_stream).
template <bool CSG, bool FASTONLY, bool MESHES = true, typename W, typename F>
__device__ __forceinline__ void world_contains_each(const DScene &sc, double px, double py, double pz, const Stack &ms, bool &needs_stream, W want, F f) {
    if (!aabb_contains(sc.wlower, sc.wupper, px, py, pz)) return;
    int32_t node = 0;
    rsx_kdnode nd = load_node(sc.wnodes, node);
    while (nd.type >= 0) {
        node = sel3(nd.type & 3, px, py, pz) < nd.u.split ? node + 1 : nd.count;
        nd = load_node(sc.wnodes, node);
    }
    // (no `continue` and no early return in the bodies handed in: hipcc 7.2 miscompiles divergent loops that lanes leave in the middle —
    // tests/toolchain/divergent_loop_exit.hip reproduces it — and round 2 held this one together with an atomic it did not need)
    for (int32_t k = 0; k < nd.count; ++k) {
        const int32_t idx = sc.witems[nd.u.leaf.first_item + k];
        const rsx_primitive &p = sc.prims[idx];
        if (want(idx)) {
            bool in;
            if constexpr (CSG && FASTONLY) {
                if (is_csg(p.type)) {
                    if (sc.csgfast && sc.csgfast[idx].n_leaves > 0) in = csg_fast_contains(sc, idx, px, py, pz, ms);
                    else { in = false; needs_stream = true; }
                } else in = aabb_contains(p.box_lower, p.box_upper, px, py, pz) && leaf_contains<MESHES>(sc, p, px, py, pz, ms);
            } else if constexpr (CSG) in = node_contains(sc, idx, px, py, pz, ms);
            else in = aabb_contains(p.box_lower, p.box_upper, px, py, pz) && leaf_contains<MESHES>(sc, p, px, py, pz, ms);
            if (in) f(idx);
        }
    }
}

struct PathStore {
    PathTerm *pool;            // [(n_records + arena_blocks) * PATH_BLOCK]
    int32_t *tail;             // [n_records] last block of each ray's list (Sample.pad = slots used in it)
    long long n_records;
    unsigned int arena_blocks;
    unsigned int *arena_next;  // bump allocator over the arena
    unsigned int *flags;       // bit 0: arena exhausted, bit 1: segment guard, bit 2: more than PATH_VOL_OVERLAP emitters at a point
    struct PathState *queue;   // paths handed on by waves that ran out of new rays (see k_render_trace_path), or null
    unsigned int *queue_count; // [0] paths in the queue, [1] paths taken out of it again
    unsigned int queue_cap;
    int32_t drain;             // 1: this launch takes its paths from the queue instead of the camera
};

// A path between two segments — everything a lane carries from one round of k_render_trace_path to the next.
struct PathState {
    Ray r;
    Sample smp;
    long long record, blk;
    uint64_t rng_pixel, rng_sample;
    unsigned long long path_spawned;
    int32_t pos, depth, segments, ray_unit, ray_slot, pad;
};
#ifndef PATH_DONATE_MAX
#define PATH_DONATE_MAX 32          // a wave out of new rays with this many live paths or fewer hands them on and retires
#endif

// Term blocks beyond a path's first come out of a shared arena. One RETURNED atomic per block — all on one address, and behind the
// in-order vector-memory counter, so the lane waits for every store it issued before — stalled nearly every round of a wave (64 lanes, a
// block every 15 terms: some lane needs one in 99 % of the rounds). A wave reserves ARENA_BATCH blocks at a time and hands them out
// from a counter pair in LDS (`res` = {next, end} of this wave, zero at kernel start): the atomic runs every ~16th round. Called by
// whatever lanes are active at the call site; `need` marks the ones that want a block. Which block a list continues in never shows
// in a result. (What a batch has left when the next is reserved is abandoned: the arena holds two blocks per path.)
#define ARENA_BATCH 64
__device__ __forceinline__ unsigned int arena_block(const PathStore &ps, bool need, volatile uint32_t *res) {
    const unsigned long long m = __ballot(need);
    unsigned int base = 0;
    const int lane = threadIdx.x % WAVE;
    const int leader = m ? __ffsll((long long)m) - 1 : 0;
    if (m != 0ULL && lane == leader) {
        const unsigned int want = (unsigned int)__popcll(m);
        unsigned int next = res[0], end = res[1];
        if (next + want > end) {
            next = atomicAdd(ps.arena_next, (unsigned int)ARENA_BATCH);
            end = next + ARENA_BATCH;
        }
        base = next;
        res[0] = next + want; res[1] = end;
    }
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
    return base + (unsigned int)__popcll(m & ((1ULL << lane) - 1ULL));
}

// Lanes are refilled: a lane whose path has ended takes the next ray of the wave's current 64-ray unit (a new unit when that one is
// used up), so a wave keeps all its lanes on live paths instead of waiting for the longest of 64 (path lengths are geometric: the
// longest of 64 is several times the mean). Which lane renders which ray never shows in the result — random numbers, sample record
// and term list are keyed by (pixel, sample).
// CSG scenes run it twice like k_render_trace: MODE 1 carries only the state-free CSG evaluator; a path that meets a ray it cannot
// finish (exact tie between operand roots, operand with mesh leaves) is abandoned and its (unit, slot) bit set in redo_mask; MODE 2,
// with the reference's stream merge, then traces the abandoned paths from their primary ray again (their term lists restart in the
// ray's own block; arena blocks of the abandoned attempt are simply not linked any more).
#ifndef RSX_PATH_MIN_WAVES
#define RSX_PATH_MIN_WAVES 2
#endif
#ifndef RSX_REDO_MIN_WAVES
// The redo pass of a CSG scene (the stream merge) usually finds a handful of paths, so its own speed does not matter; when it can
// START does: built for one wave per SIMD (256 + 143 registers) each of its workgroups needs a CU whose SIMDs have drained completely,
// behind the persistent workgroups of the other slices in flight, and the slice's k_accumulate waits behind it (2.4 ms on average
// per configs[4] slice). Capped for two waves per SIMD (181 registers spilled, 7 KB scratch) it moves in as soon as one path wave
// retires: configs[4] 1.72 -> 1.67 s per pass.
#define RSX_REDO_MIN_WAVES 2
#endif
// VOLS = the scene has materials with a volume contribution (volume emitters, dielectrics): only then is the per-segment
// world.contains() pass compiled in (its CSG form, a depth-unrolled recursion, costs the CSG instantiation a wave per SIMD).
// REWALK: the instantiation a pass is traced again with when a path met more overlapping volumes than the registers keep
// (PATH_VOL_OVERLAP): it produces the older terms by walking world.contains() again. Compiled into the ordinary instantiation the
// second copy of the enumeration slowed every path pass by 30 % without ever running (and out of line by 20 %: the call ABI).
// STAGED: the primitive records and CSG programs are read from an LDS copy (scenes of a few dozen primitives; see render()). A
// template parameter, not a run-time choice: a pointer that may be either makes every record access of a big scene a flat load
// (prism-sized scene unstaged 35 -> 38 ms, Cornell box 42.5 -> 44.5 ms).
// QUEUE: the instantiation can hand paths on and drain them (PathStore::queue / drain) — the forms the overlapping slices of an observe() run;
// compiled into every form the hand-over cost a pass that never uses it 3 % (Cornell box 41.1 -> 42.7 ms: six more spilled registers).
// MESHES false: the form for scenes without a mesh primitive (render() picks it). The wave-cooperative mesh walk and its triangle
// test held the register peak of every path instantiation (235 registers live there against ~200 elsewhere) whether or not a path
// ever reached a mesh; without them the plain form fits three waves per SIMD (168 registers, 328 B scratch: Cornell box 28.3 ->
// 24.6 ms per pass, two waves 25.9), the CSG forms stay at two (three waves spill the flattened evaluator: prism pass 333 -> 487 ms)
// and gain from the shorter code alone (357 -> 333 ms).
#ifndef RSX_PATH_NOMESH_MIN_WAVES
#define RSX_PATH_NOMESH_MIN_WAVES 3
#endif
// The ray bank. A path wave refills the lanes whose paths ended from the unit it holds a ticket for — seven lanes of 64 in a Cornell box
// round — and ran the whole of a camera ray for them every round (pixel of the ray's number, its Philox draw, PinholeCamera._generate_rays,
// the transform to world space: ~350 vector instructions for a tenth of the lanes, 0.16 of the kernel). Here the wave makes the 64 rays of a
// unit at once when it takes the unit, all lanes busy, and leaves them in LDS; a refill is a handful of loads. Same arithmetic per ray.
#define RAY_BANK_BYTES (WAVE * (4 * 8 + 3 * 4))      // per wave: direction[3][64], weight[64] (f64); record[64] (-1: no ray), pixel key[64], sample[64] (i32)
template <bool CSG, int MODE = 0, bool VOLS = true, bool REWALK = false, bool STAGED = false, bool QUEUE = false, bool MESHES = true>
__global__ __launch_bounds__(WG_THREADS, CSG && MODE == 2 ? RSX_REDO_MIN_WAVES : CSG && MODE != 1 ? 1 : !CSG && !MESHES ? RSX_PATH_NOMESH_MIN_WAVES : RSX_PATH_MIN_WAVES) void k_render_trace_path(DScene sc_arg, RenderParams rp, Sample *samples, unsigned long long *ticket, PathStore ps) {
    __shared__ uint32_t arena_res[2 * WG_WAVES];           // arena_block: the blocks each wave has reserved
    if (threadIdx.x < 2 * WG_WAVES) arena_res[threadIdx.x] = 0;
    __syncthreads();
    DScene sc = sc_arg;
    if (STAGED || rp.world_lds > 0) {                      // stage the world tree behind the traversal stacks (see render(); STAGED implies it)
        int4 *dst = reinterpret_cast<int4 *>(smem + rp.world_lds);
        const int4 *src = reinterpret_cast<const int4 *>(CSG && MODE != 1 ? sc_arg.wnodes : sc_arg.wnodes_scatter);   // (the copy annotated for this kernel: world_trace_wave)
        for (int i = threadIdx.x; i < sc_arg.n_wnodes; i += blockDim.x) dst[i] = src[i];
        int32_t *idst = reinterpret_cast<int32_t *>(dst + sc_arg.n_wnodes);
        for (int i = threadIdx.x; i < sc_arg.n_witems; i += blockDim.x) idst[i] = sc_arg.witems[i];
        __syncthreads();
        sc.wnodes = sc.wnodes_scatter = reinterpret_cast<const rsx_kdnode *>(dst);     // (contains() walks read type / count / first item only)
        sc.witems = idst;
        if constexpr (STAGED) {                            // ... and the primitive records and CSG programs the lanes read one by one
            long long *pdst = reinterpret_cast<long long *>(smem + rp.prims_lds);
            const long long *psrc = reinterpret_cast<const long long *>(sc_arg.prims);
            const int n8 = sc_arg.n_prims * (int)(sizeof(rsx_primitive) / 8);
            for (int i = threadIdx.x; i < n8; i += blockDim.x) pdst[i] = psrc[i];
            sc.prims = reinterpret_cast<const rsx_primitive *>(pdst);
            if (CSG && sc_arg.csgfast) {
                long long *fdst = pdst + n8;
                const long long *fsrc = reinterpret_cast<const long long *>(sc_arg.csgfast);
                const int f8 = sc_arg.n_prims * (int)(sizeof(CsgFast) / 8);
                for (int i = threadIdx.x; i < f8; i += blockDim.x) fdst[i] = fsrc[i];
                sc.csgfast = reinterpret_cast<const CsgFast *>(fdst);
            }
            __syncthreads();
        }
    }
    Stack st, ms;
    wave_stacks(sc, st, ms);
    const int lane = threadIdx.x % WAVE;
    double *const bank_d = reinterpret_cast<double *>(smem + rp.bank_lds + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)) * RAY_BANK_BYTES);   // [4][WAVE]
    int32_t *const bank_i = reinterpret_cast<int32_t *>(bank_d + 4 * WAVE);                                                                              // [3][WAVE]
    NodeSt csg_state[CSG && MODE != 1 ? CSG_MAX_SLOTS : 1];
    int ray_unit = 0, ray_slot = 0;                        // where the lane's current ray came from (MODE 1: to flag it for the redo pass)
    unsigned long long path_spawned = 0;
    const unsigned long long rp_bits = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DScene) + 7) & ~(size_t)7);
    const RSX_CONST_AS RenderParams *q = (const RSX_CONST_AS RenderParams *)rp_bits;
    (void)rp;
    const int my_xcd = xcc_id();
    int victim = -1;
    unsigned long long redo_batch = 0ULL;                  // MODE 2: the entries of the wave's current 64-entry ticket whose units have rays to redo,
    long long redo_base = 0;                               // and the ticket's first list position (wave-uniform)
    long long unit = -1;                                   // wave-uniform: the unit rays are being taken from, and how many are gone
    int cursor = WAVE;
    bool exhausted = false;
    // per-lane path state
    bool active = false;
    // (kept narrow: the kernel lives at 256 registers, and every value that stays live through the walk is a spill elsewhere — record and
    // block ids stay below 2^31 (render()), a pixel index below 2^32, and of the sample record only the weight is known before the
    // path ends: the emitter's (a, table) are found in its last round)
    Ray r;
    double smp_weight = 0;
    int32_t record = 0, blk = 0;
    uint32_t rng_pixel_lo = 0;
    uint64_t rng_sample = 0;
    int pos = 0, depth = 0, segments = 0;
    unsigned int spawned = 0;                              // rays this lane traced or spawned (the reference's ray_count statistic)
    uint32_t work = 0;
    r.ox = r.oy = r.oz = 0; r.dx = r.dy = 0; r.dz = 1; r.maxd = INFINITY;
#if RSX_PHASE_PROF == 3
    unsigned long long pp_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pp_mark = clock64(), pp_t2 = 0, pp_t3 = 0;
#endif
    auto push = [&](double a, double b, int32_t table, int32_t kind) {
        const bool full = pos == PATH_BLOCK;
        const unsigned int nb = arena_block(ps, full, arena_res + 2 * (threadIdx.x / WAVE));
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
    // Russian roulette of a freshly spawned daughter (ray.pyx:382-388): 0 = extinguished, 1 = alive, 2 = alive and its result is
    // scaled by 1 / (1 - extinction_prob)
    auto roulette = [&]() -> int {
        if (depth < q->ray_min_depth) return 1;
        if (depth >= q->ray_max_depth) return 0;
        double k1, k2;
        philox2(q->seed, (uint64_t)rng_pixel_lo, rng_sample | ((uint64_t)(2 * depth) << 48), k1, k2);
        return k1 < q->ray_extinction_prob ? 0 : 2;
    };
    for (;;) {
        // ---- refill idle lanes ----
        unsigned long long idle = __ballot(!active);
        if (QUEUE && __builtin_expect(ps.drain != 0, 0)) {
            // second launch of a pass: the paths the first launch's retiring waves handed on, packed 64 to a wave again
            while (idle && !exhausted) {
                const int n_idle = __popcll(idle);
                unsigned int start = 0;
                if (lane == 0) start = atomicAdd(ps.queue_count + 1, (unsigned int)n_idle);
                start = (unsigned int)__builtin_amdgcn_readfirstlane((int)start);
                const unsigned int filled = __builtin_amdgcn_readfirstlane((int)ps.queue_count[0]);      // (final: the first launch has ended)
                const int take = start >= filled ? 0 : (filled - start < (unsigned int)n_idle ? (int)(filled - start) : n_idle);
                const int rank = __popcll(idle & ((1ULL << lane) - 1ULL));
                if (!active && rank < take) {
                    const PathState stt = ps.queue[start + (unsigned int)rank];
                    r = stt.r; smp_weight = stt.smp.weight; record = (int32_t)stt.record; blk = (int32_t)stt.blk; rng_pixel_lo = (uint32_t)stt.rng_pixel; rng_sample = stt.rng_sample;
                    pos = stt.pos; depth = stt.depth; segments = stt.segments; ray_unit = stt.ray_unit;
                    if constexpr (MODE == 1) { path_spawned = stt.path_spawned; ray_slot = stt.ray_slot; }       // (what a path leaves to the redo pass)
                    active = true;
                }
                if (take < n_idle) exhausted = true;
                idle = __ballot(!active);
            }
        }
        while (idle && !exhausted) {
            if (cursor >= WAVE) {
                long long tk = -1;
                if constexpr (MODE == 2) {
                    // The redo pass visits the units whose mask is not zero — a few in a thousand. One ticket per unit made every wave pay a
                    // returned atomic on one of nine contended addresses (~17 ns each, serialised: 131 072 units of an 8-pass slice = 2.2 ms
                    // in which the launch's workgroups hold their 256-register places) plus two dependent loads, to learn "nothing to do".
                    // A ticket is 64 list entries here: the wave reads their 64 masks at once, one per lane, and walks the set bits.
                    for (;;) {
                        if (redo_batch) { tk = redo_base + (__ffsll((long long)redo_batch) - 1); redo_batch &= redo_batch - 1ULL; break; }
                        bool refilled = false;
                        while (victim < 8) {
                            const int list = victim < 0 ? 0 : 1 + ((my_xcd + victim) & 7);
                            const long long begin = q->seg[list], end = q->seg[list + 1];
                            unsigned long long mine = 0;
                            if (lane == 0) mine = atomicAdd(ticket + 16 * list, (unsigned long long)WAVE);
                            const long long got = begin + (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mine >> 32)) << 32) |
                                                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)mine));
                            if (got < end) {
                                unsigned long long m = 0ULL;
                                if (got + lane < end) m = q->redo_mask[q->unit_order[got + lane] & 0x3ffffffu];
                                redo_batch = __ballot(m != 0ULL); redo_base = got; refilled = true;
                                break;
                            }
                            ++victim;
                        }
                        if (!refilled) break;
                    }
                } else
                while (victim < 8) {
                    const int list = victim < 0 ? 0 : 1 + ((my_xcd + victim) & 7);
                    const long long begin = q->seg[list], end = q->seg[list + 1];
                    unsigned long long mine = 0;
                    if (lane == 0) mine = atomicAdd(ticket + 16 * list, 1ULL);
                    const long long got = begin + (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mine >> 32)) << 32) |
                                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)mine));
                    if (got < end) { tk = got; break; }
                    ++victim;
                }
                if (tk < 0) { exhausted = true; break; }
                unit = __builtin_amdgcn_readfirstlane((int)(q->unit_order[tk] & 0x3ffffffu));
                cursor = 0;
                if constexpr (MODE == 2) { if (q->redo_mask[unit] == 0ULL) { cursor = WAVE; continue; } }
                {   // the unit's 64 rays into the wave's bank, ray `lane` by lane `lane`
                    const UnitPixel px = unit_pixel(q, unit, lane);
                    const uint32_t key = (uint32_t)px.ix * (uint32_t)q->cam.ny + (uint32_t)px.iy;
                    double u1, u2;
                    if (q->rng_mode == RSX_RNG_STREAM) { u1 = q->uniforms[2 * (px.k * q->spp + px.s)]; u2 = q->uniforms[2 * (px.k * q->spp + px.s) + 1]; }
                    else philox2(q->seed, (uint64_t)key, q->sample_offset + (uint64_t)px.s, u1, u2);
                    Ray cr;
                    double weight;
                    camera_ray(q, px.ix, px.iy, u1, u2, cr, weight);
                    bank_d[lane] = cr.dx; bank_d[WAVE + lane] = cr.dy; bank_d[2 * WAVE + lane] = cr.dz; bank_d[3 * WAVE + lane] = weight;
                    bank_i[lane] = px.valid ? (int32_t)(px.slot * q->spp + px.s) : -1; bank_i[WAVE + lane] = (int32_t)key; bank_i[2 * WAVE + lane] = px.s;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            const int n_idle = __popcll(idle);
            const int take = n_idle < WAVE - cursor ? n_idle : WAVE - cursor;
            const int rank = __popcll(idle & ((1ULL << lane) - 1ULL));
            if (!active && rank < take) {
                const int e = cursor + rank;
                const int32_t rec = bank_i[e];
                bool wanted = rec >= 0;
                if constexpr (MODE == 2) wanted = wanted && ((q->redo_mask[unit] >> e) & 1ULL);
                if (wanted) {
                    ray_unit = (int)unit; ray_slot = e; path_spawned = 0;
                    rng_pixel_lo = (uint32_t)bank_i[WAVE + e]; rng_sample = q->sample_offset + (uint64_t)bank_i[2 * WAVE + e];
                    r.ox = q->origin[0]; r.oy = q->origin[1]; r.oz = q->origin[2];      // (camera_ray's)
                    r.dx = bank_d[e]; r.dy = bank_d[WAVE + e]; r.dz = bank_d[2 * WAVE + e]; r.maxd = INFINITY;
                    smp_weight = bank_d[3 * WAVE + e];
                    record = rec;
                    blk = record; pos = 0; depth = 0; segments = 0;
                    ++spawned; ++path_spawned;
                    active = true;
                }
            }
            cursor += take;
            idle = __ballot(!active);
        }
        // Out of new rays with only a few paths left: a trapped path (total internal reflection inside glass, hundreds of segments) would
        // keep this wave — and with it the workgroup's place on its CU — for milliseconds at one live lane in 64. The wave hands its
        // paths on and retires; a second, small launch (ps.drain) walks the handed-on paths of all waves, packed 64 to a wave, while the
        // places this launch gave back take the next pass's workgroups. Which wave walks a path never shows in the result (random
        // numbers, sample record and term list are keyed by pixel and sample).
        if (QUEUE && __builtin_expect(exhausted && ps.queue != nullptr && !ps.drain, 0)) {
            const unsigned long long live = __ballot(active);
            const int n_live = __popcll(live);
            if (n_live > 0 && n_live <= PATH_DONATE_MAX) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(ps.queue_count, (unsigned int)n_live);
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (base + (unsigned int)n_live <= ps.queue_cap) {
                    if (active) {
                        PathState stt;
                        stt.r = r; stt.smp.a = 0.0; stt.smp.weight = smp_weight; stt.smp.table = -1; stt.smp.pad = 0;
                        stt.record = record; stt.blk = blk; stt.rng_pixel = (uint64_t)rng_pixel_lo; stt.rng_sample = rng_sample;
                        stt.path_spawned = 0; stt.ray_slot = 0;
                        if constexpr (MODE == 1) { stt.path_spawned = path_spawned; stt.ray_slot = ray_slot; }
                        stt.pos = pos; stt.depth = depth; stt.segments = segments; stt.ray_unit = ray_unit; stt.pad = 0;
                        ps.queue[base + (unsigned int)__popcll(live & ((1ULL << lane) - 1ULL))] = stt;
                        active = false;
                    }
                } else if (lane == 0) atomicSub(ps.queue_count, (unsigned int)n_live);      // (cannot happen: the queue holds every wave's share)
            }
        }
        if (!__any(active)) break;
        // ---- one segment of every live path ----
        const bool was_active = active;
        Hit hit;
        work = 0;
        double end_a = 0.0;                                                   // the sample record's emitter, if this round ends the path at one
        int32_t end_table = -1;
#if RSX_PHASE_PROF == 3
        const unsigned long long pp0 = clock64();
        pp_acc[0] += pp0 - pp_mark;                           // refill
        pp_acc[6] += __popcll(__ballot(active)); pp_acc[7] += 1;
#endif
#if RSX_PHASE_PROF == 3
        const bool got = world_trace_wave<CSG, MODE == 1, RSX_STAGE_MIN, false, !CSG ? 8 : MODE == 1 && RSX_CSG_MAILBOX >= 4 ? RSX_CSG_WIDE : 2, RSX_CSG_MAILBOX, MESHES>(active, sc, r, st, ms, csg_state, hit, work, pp_acc);
#else
        const bool got = world_trace_wave<CSG, MODE == 1, RSX_STAGE_MIN, false, !CSG ? 8 : MODE == 1 && RSX_CSG_MAILBOX >= 4 ? RSX_CSG_WIDE : 2, RSX_CSG_MAILBOX, MESHES>(active, sc, r, st, ms, csg_state, hit, work);
#endif
#if RSX_PHASE_PROF == 3
        const unsigned long long pp1 = clock64();
        pp_acc[1] += pp1 - pp0;                               // world_trace_wave
#endif
        bool abandoned = false;
        if constexpr (MODE == 1) {
            if (active && (work >> 31)) {                                     // this path needs the stream merge: hand it to the redo pass
                atomicOr(q->redo_mask + ray_unit, 1ULL << ray_slot);
                spawned -= (unsigned int)path_spawned;
                abandoned = true;
                active = false;
            }
        }
        if (active && !got) active = false;                               // new_spectrum(): no volume pass for a segment that hits nothing
        if (active) {
            const rsx_primitive &p = sc.prims[hit.prim];
            const rsx_material mat = q->materials[p.material];
            Geom g;
            finalise<CSG, MESHES>(sc, r, hit, g);
            double hx, hy, hz;                                                // hit_point.transform(primitive_to_world)
            xform_point(p.to_root, g.hit[0], g.hit[1], g.hit[2], hx, hy, hz);
#if RSX_PHASE_PROF == 3
            pp_t2 = clock64();
#endif
            double *const park = reinterpret_cast<double *>(st.stage) + lane;          // [6][WAVE]: see parked_point below
            static_assert(STAGE_BYTES >= 6 * WAVE * 8, "the parked points fit the staging area");
            auto park_points = [&]() {
#pragma unroll
                for (int k = 0; k < 3; ++k) { park[k * WAVE] = g.inside[k]; park[(3 + k) * WAVE] = g.outside[k]; }
            };
            if constexpr (!MESHES) park_points();                              // (no mesh walk: nothing else uses the staging area, the points leave the registers before the volume pass)
            // volume emitters containing this segment's origin: found in world.contains() order, pushed newest first because the
            // list is replayed backwards
            double v_len[PATH_VOL_OVERLAP] = {0, 0, 0, 0}, v_scale[PATH_VOL_OVERLAP] = {0, 0, 0, 0};     // (initialised: the shift below reads every slot)
            int32_t v_table[PATH_VOL_OVERLAP] = {0, 0, 0, 0}, v_kind[PATH_VOL_OVERLAP] = {0, 0, 0, 0};
            int n_vol = 0;
            bool contains_needs_stream = false;
            if constexpr (VOLS) if (q->n_vol_emitters) world_contains_each<CSG, MODE == 1, MESHES>(sc, r.ox, r.oy, r.oz, ms, contains_needs_stream, [&](int32_t idx) {
                const int32_t vm_id = sc.prims[idx].material;                         // every other evaluate_volume leaves the spectrum unchanged
                const int32_t vt = q->materials[vm_id].type;                          // (light_dir[2] != 0: a dielectric of unit transmission — render())
                return vt == RSX_MAT_UNIFORM_VOLUME_EMITTER || (vt == RSX_MAT_DIELECTRIC && q->materials[vm_id].light_dir[2] == 0.0);
            }, [&](int32_t idx) {
                const rsx_primitive &vp = sc.prims[idx];
                const rsx_material vm = q->materials[vp.material];
                double length;
                bool skip = false;
                if (vm.type == RSX_MAT_DIELECTRIC) {                          // dielectric.pyx:300-328: world-space length
                    const double vx = r.ox - hx, vy = r.oy - hy, vz = r.oz - hz;  // start_point.vector_to(end_point)
                    length = sqrt(vx * vx + vy * vy + vz * vz);
                } else {
                    double sx, sy, sz, ex, ey, ez;
                    xform_point(vp.to_local, hx, hy, hz, sx, sy, sz);
                    xform_point(vp.to_local, r.ox, r.oy, r.oz, ex, ey, ez);
                    const double vx = sx - ex, vy = sy - ey, vz = sz - ez;    // end.vector_to(start)
                    length = sqrt(vx * vx + vy * vy + vz * vz);
                    skip = length == 0;                                       // homogeneous.pyx:92-94: nothing to add (no early return: see world_contains_each)
                }
                if (!skip) {
                    // More volumes at this point than the registers keep (the newest PATH_VOL_OVERLAP): the pass is traced again by the REWALK
                    // instantiation (flag bit 2), which fetches the older ones again below.
                    if constexpr (!REWALK) { if (n_vol == PATH_VOL_OVERLAP) atomicOr(ps.flags, 4u); }
#pragma unroll
                    for (int j = PATH_VOL_OVERLAP - 1; j > 0; --j) { v_len[j] = v_len[j - 1]; v_scale[j] = v_scale[j - 1]; v_table[j] = v_table[j - 1]; v_kind[j] = v_kind[j - 1]; }
                    v_len[0] = length; v_scale[0] = vm.scale; v_table[0] = vm.table; v_kind[0] = vm.type == RSX_MAT_DIELECTRIC ? TERM_ATTEN : TERM_VOL;
                    ++n_vol;
                }
            });
#pragma unroll
            for (int j = 0; j < PATH_VOL_OVERLAP; ++j) if (j < n_vol) push(v_len[j], v_scale[j], v_table[j], v_kind[j]);
            if constexpr (VOLS && REWALK) {
                // the terms beyond the newest four: the same world.contains() enumeration walked again, once per missing term, oldest
                // last — the list is replayed backwards, so the push order is newest first
                for (int want = n_vol - PATH_VOL_OVERLAP - 1; want >= 0; --want) {
                    int seen = 0;
                    bool dummy = false;
                    world_contains_each<CSG, MODE == 1, MESHES>(sc, r.ox, r.oy, r.oz, ms, dummy, [&](int32_t idx) {
                        const int32_t vm_id = sc.prims[idx].material;
                        const int32_t vt = q->materials[vm_id].type;
                        return vt == RSX_MAT_UNIFORM_VOLUME_EMITTER || (vt == RSX_MAT_DIELECTRIC && q->materials[vm_id].light_dir[2] == 0.0);
                    }, [&](int32_t idx) {
                        const rsx_primitive &vp = sc.prims[idx];
                        const rsx_material vm = q->materials[vp.material];
                        double length;
                        if (vm.type == RSX_MAT_DIELECTRIC) {
                            const double vx = r.ox - hx, vy = r.oy - hy, vz = r.oz - hz;
                            length = sqrt(vx * vx + vy * vy + vz * vz);
                        } else {
                            double sx, sy, sz, ex, ey, ez;
                            xform_point(vp.to_local, hx, hy, hz, sx, sy, sz);
                            xform_point(vp.to_local, r.ox, r.oy, r.oz, ex, ey, ez);
                            const double vx = sx - ex, vy = sy - ey, vz = sz - ez;
                            length = sqrt(vx * vx + vy * vy + vz * vz);
                        }
                        if (vm.type == RSX_MAT_DIELECTRIC || length != 0) {
                            if (seen++ == want) push(length, vm.scale, vm.table, vm.type == RSX_MAT_DIELECTRIC ? TERM_ATTEN : TERM_VOL);
                        }
                    });
                }
            }
            if constexpr (MODE == 1) {
                if (contains_needs_stream) {                                  // a CSG volume without a flattened program: redo pass
                    atomicOr(q->redo_mask + ray_unit, 1ULL << ray_slot);
                    spawned -= (unsigned int)path_spawned;
                    abandoned = true;
                    active = false;
                }
            }
#if RSX_PHASE_PROF == 3
            pp_t3 = clock64();
#endif
            ++segments;
            // What the scattering materials' arms have in common runs once, outside them (the arms run one after the other for
            // whatever lanes each has): the scattering draw of this depth before them — Lambert's (h1, h2) or, with important
            // primitives, its (choose, pick); the Dielectric's reflect / transmit draw — and, behind them, the daughter's roulette
            // and the term it leaves in the list.
            double scatter1 = 0.0, scatter2 = 0.0;
            if (!abandoned && segments < PATH_MAX_SEGMENTS && (mat.type == RSX_MAT_LAMBERT || mat.type == RSX_MAT_DIELECTRIC))
                philox2(q->seed, (uint64_t)rng_pixel_lo, rng_sample | ((uint64_t)(2 * depth + 1) << 48), scatter1, scatter2);
#if RSX_PHASE_PROF == 3
            pp_acc[18] += clock64() - pp_t3;
#endif
            bool daughter = false, lambert_term = false;
            double term_a = 1.0, term_b = 1.0;                                // (a = b = 1: the replay multiplies every term's a and b in)
            // The intersection's inside and outside points wait in LDS for the arm that takes one of them: twelve registers through every arm
            // otherwise — what the compiler spilled to scratch. The wave's leaf staging area is free here: the walk is over, the volume pass too
            // (park_points, above).
            if constexpr (MESHES) park_points();                               // (forms with the mesh walk: the volume pass above stages leaves there)
            auto parked_point = [&](bool outside, double &x, double &y, double &z) {
                const double *at = park + (outside ? 3 * WAVE : 0);
                x = at[0]; y = at[WAVE]; z = at[2 * WAVE];
            };
            if (abandoned) {}
            else if (segments >= PATH_MAX_SEGMENTS) { atomicOr(ps.flags, 2u); active = false; }
            else if (mat.type == RSX_MAT_NULL || mat.type == RSX_MAT_UNIFORM_VOLUME_EMITTER) {      // null surface: carry on from the far side
                double fx, fy, fz;
                parked_point(g.exiting, fx, fy, fz);
                xform_point(p.to_root, fx, fy, fz, r.ox, r.oy, r.oz);
                ++spawned; ++path_spawned;
            } else if (mat.type == RSX_MAT_DIELECTRIC) {                      // dielectric.pyx:159-262
#if RSX_PHASE_PROF == 3
                const unsigned long long ppd0 = clock64();
#endif
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
                if (reflect && transmission_only) active = false;             // total internal reflection without a reflected ray: zero spectrum
                else {
                    if (reflect) {
                        const double temp = 2 * c1;
                        ox = ix + temp * nx; oy = iy + temp * ny; oz = iz + temp * nz;
                    }
                    double fx, fy, fz;
                    parked_point(reflect ? !inside : inside, fx, fy, fz);         // reflected: from the side the ray came from; transmitted: the far side
                    xform_point(p.to_root, fx, fy, fz, r.ox, r.oy, r.oz);
                    xform_vector(p.to_root, ox, oy, oz, r.dx, r.dy, r.dz);
                    daughter = true;
                }
#if RSX_PHASE_PROF == 3
                pp_acc[15] += clock64() - ppd0; pp_acc[16] += 1;
#endif
            } else if (mat.type != RSX_MAT_LAMBERT) {                         // emitters and absorbers end the path
                if (mat.type == RSX_MAT_UNIFORM_EMITTER) { end_a = mat.scale; end_table = mat.table; }
                else if (mat.type == RSX_MAT_DEBUG_LIGHT && mat.scale != 0.0) {
                    double lx, ly, lz;
                    xform_vector(p.to_local, -mat.light_dir[0], -mat.light_dir[1], -mat.light_dir[2], lx, ly, lz);
                    const double dot = lx * g.normal[0] + ly * g.normal[1] + lz * g.normal[2];
                    end_a = mat.scale * (dot > 0 ? dot : 0.0);
                    end_table = mat.table;
                }
                active = false;
            } else {                                                          // RSX_MAT_LAMBERT (the last arm: nothing behind it reads the intersection record)
#if RSX_PHASE_PROF == 3
                const unsigned long long ppl0 = clock64();
#endif
                // The draw first, the surface frame behind it (the order of independent operations is free: the sampling arithmetic — two
                // Philox draws, portable_sincos, the cone's portable_asin, the CDF walk — holds the kernel's register peak, and the frame's
                // three vectors and its matrix are not needed before the direction is carried into it).
                // Outgoing direction: HemisphereCosineSampler.sample (solidangle.pyx:228-233) or, in a world with important primitives,
                // the important-path / BSDF mixture of ContinuousBSDF.evaluate_surface (material.pyx:327-352)
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
                    double sn, cs;                                            // one azimuth for every lane: 2 pi h1 (a cone) or 2 pi h2 (inside a sphere; the cosine lobe)
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
                // w_reflection_origin and the surface frame (_generate_surface_transforms, material.pyx:393-422; Normal3D.orthogonal,
                // normal.pyx:346-370); the normal faces the incident side
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
                    // s_outgoing = w_outgoing.transform(primitive_to_surface.mul(world_to_primitive))
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
                // surface_to_world = primitive_to_world.mul(surface_to_primitive) (affinematrix.pyx:255-273), rotation part, and
                // direction = s_outgoing.transform(surface_to_world) — for a direction drawn from the cosine lobe the same operations as the
                // w_outgoing the important pdf is asked about (material.pyx:343), so that is this vector
                {
                    const double *a = p.to_root;
                    double stw[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        stw[3 * i + 0] = a[4 * i] * tx + a[4 * i + 1] * ty + a[4 * i + 2] * tz + a[4 * i + 3] * 0.0;
                        stw[3 * i + 1] = a[4 * i] * bx + a[4 * i + 1] * by + a[4 * i + 2] * bz + a[4 * i + 3] * 0.0;
                        stw[3 * i + 2] = a[4 * i] * nx + a[4 * i + 1] * ny + a[4 * i + 2] * nz + a[4 * i + 3] * 0.0;
                    }
                    r.dx = stw[0] * sx + stw[1] * sy + stw[2] * sz;
                    r.dy = stw[3] * sx + stw[4] * sy + stw[5] * sz;
                    r.dz = stw[6] * sx + stw[7] * sy + stw[8] * sz;
                }
                if (mis) {
                    if (!from_important) { wx = r.dx; wy = r.dy; wz = r.dz; }
                    pdf_important = important_pdf(q->important, q->n_important, hx, hy, hz, wx, wy, wz);
                }
                const double pdf = sz >= 0.0 ? M_1_PI * sz : 0.0;             // HemisphereCosineSampler.pdf
                const double pdf_all = mis ? q->important_path_weight * pdf_important + (1 - q->important_path_weight) * pdf : pdf;
                const double rcp = 1.0 / pdf_all;                             // div_scalar (spectrum.pyx:459-467)
                if (pdf == 0.0) { push(pdf, rcp, mat.table, TERM_LAMBERT); active = false; }   // zero spectrum, then * (1 / 0)
                else {
                    double fx, fy, fz;                                        // w_reflection_origin: the point on the incident side, fetched now
                    parked_point(!g.exiting, fx, fy, fz);
                    xform_point(p.to_root, fx, fy, fz, r.ox, r.oy, r.oz);
                    daughter = true; lambert_term = true; term_a = pdf; term_b = rcp;
                }
#if RSX_PHASE_PROF == 3
                pp_acc[13] += clock64() - ppl0; pp_acc[14] += 1;
#endif
            }
#if RSX_PHASE_PROF == 3
            const unsigned long long ppt0 = clock64();
#endif
            if (daughter) {                                                   // ray.pyx:380-388: the daughter exists (and counts) before its roulette
                ++depth;
                ++spawned; ++path_spawned;
                const int alive = roulette();
                if (!alive) active = false;
                // Lambert: the term is left whatever the roulette says; Dielectric: only the roulette's 1 / (1 - p)
                if (lambert_term || alive == 2) push(term_a, term_b, mat.table, !lambert_term ? TERM_NORM : alive == 2 ? TERM_LAMBERT_NORM : TERM_LAMBERT);
            }
#if RSX_PHASE_PROF == 3
            pp_acc[17] += clock64() - ppt0;
#endif
        }
#if RSX_PHASE_PROF == 3
        {
            const unsigned long long pp4 = clock64();
            if (pp_t2) { pp_acc[2] += pp_t2 - pp1; pp_acc[3] += pp_t3 - pp_t2; pp_acc[4] += pp4 - pp_t3; } else pp_acc[4] += pp4 - pp1;
            pp_t2 = pp_t3 = 0;
            pp_mark = pp4;
        }
#endif
        if (was_active && !active && !abandoned) {                            // path over: its record is complete
#ifndef RSX_NO_PATHCOST
            if (q->measure_cost) atomicMax(q->unit_cost + ray_unit, (uint32_t)segments + 1u);   // the unit's longest path: next pass's schedule
#endif
            Sample smp;
            smp.a = end_a; smp.weight = smp_weight; smp.table = end_table; smp.pad = pos;
            samples[record] = smp;
            ps.tail[record] = (int32_t)blk;
        }
    }
#if RSX_PHASE_PROF == 3
    if (lane == 0 && q->unit_times) for (int k = 0; k < 24; ++k) atomicAdd(q->unit_times + k, pp_acc[k]);
#endif
    // ray statistics (Ray.ray_count, ray.pyx:536-547: the primary ray and every daughter spawned)
    // (a lane of the drain launch that abandons a handed-on path takes off rays other lanes counted: the lane's count is signed)
    unsigned long long spawned_wave = (unsigned long long)(long long)(int)spawned;
    for (int o = 32; o > 0; o >>= 1) spawned_wave += __shfl_xor(spawned_wave, o);
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(ps.flags) + 1, spawned_wave);
}

// Self-test of exact_div(): bit equality with the compiler's IEEE division over pseudo-random and adversarial operand pairs.
__global__ void k_selftest_division(unsigned long long n, unsigned long long seed, unsigned long long *mismatches) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long i = gid; i < n; i += stride) {
        double u1, u2;
        philox2(seed, i, 0, u1, u2);
        double a, b;
        philox2(seed, i, 1, a, b);
        double num, den;
        switch (i & 7) {
        case 0: num = (u1 - 0.5) * 4.0; den = (u2 - 0.5) * 2.0; break;                      // plane distances: (split - o) / d
        case 1: num = ldexp(u1 - 0.5, (int)(a * 120) - 60); den = ldexp(u2 - 0.5, (int)(b * 120) - 60); break;
        case 2: den = (u2 - 0.5) * 2.0; num = den * (double)(long long)(u1 * 4096.0 - 2048.0); break;   // exact quotients
        case 3: den = 1.0 + u2 * 0x1p-30; num = 1.0 + u1 * 0x1p-30; break;                 // quotients hugging 1 (near ties)
        case 4: den = (double)(1 + (long long)(u2 * 1e6)); num = (double)(long long)(u1 * 2e6 - 1e6); break;   // small integers
        case 5: num = ldexp(u1 - 0.5, (int)(a * 1200) - 600); den = ldexp(u2 - 0.5, (int)(b * 1200) - 600); break;   // wide exponents (guard path)
        case 6: num = (i & 8) ? 0.0 : -0.0; den = (u2 - 0.5); break;                        // zero numerators
        default: den = u2 * 1e-3 + 1e-300 * a; num = u1 - 0.5; break;                       // tiny denominators
        }
        if (den == 0.0) continue;
        const double want = num / den;
        const double got = exact_div(num, den, refine_rcp(den), div_operand_safe(den));
        if (__double_as_longlong(want) != __double_as_longlong(got)) { ++bad; atomicAdd(mismatches + 1 + (i & 7), 1ULL); }
        // the world-level branch steps use the ray's own 1.0 / d (kept for the box gates) as the reciprocal
        const double got2 = exact_div(num, den, 1.0 / den, div_operand_safe(den));
        if (__double_as_longlong(want) != __double_as_longlong(got2)) { ++bad; atomicAdd(mismatches + 1 + (i & 7), 1ULL); }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Longest-processing-time-first schedule for the next pass over the same units: counting sort of the measured unit costs into
// 128 logarithmic buckets, most expensive first. A few silhouette tiles cost 50x the median (grazing rays cross hundreds of KD
// cells); handing them out first lets the cheap bulk fill in behind them instead of leaving one wave to finish alone.
#define ORDER_BUCKETS 128
__device__ __forceinline__ int cost_bucket(uint32_t c) {
    if (c == 0) return 0;
    const int lg = 31 - __clz((int)c);                      // floor(log2 c)
    const int frac = lg >= 2 ? (int)((c >> (lg - 2)) & 3) : 0;   // two mantissa bits -> quarter-octave resolution
    const int b = lg * 4 + frac;
    return b < ORDER_BUCKETS ? b : ORDER_BUCKETS - 1;
}

#ifndef RSX_HEAVY_FACTOR
#define RSX_HEAVY_FACTOR 3ULL
#endif

// which XCD's list a unit belongs to: 4x4-tile blocks (32x32 pixels) are dealt round-robin to the 8 XCDs, so each L2 caches the
// geometry behind an eighth of the image while every XCD still gets a fair share of cheap and expensive regions
__device__ __forceinline__ int unit_xcd(long long unit, int tiles_x, int spp) {
    const long long tile = unit / spp;                       // first pixel of the unit = 64 * unit / spp; 64 pixels per tile
    if (tiles_x > 0) { const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x); return ((tx >> 2) + 3 * (ty >> 2)) & 7; }
    return (int)((tile >> 4) & 7);
}

// list 0: units well above the mean cost (latency-bound stragglers: every XCD takes them first); lists 1..8: the rest, by XCD
__device__ __forceinline__ int unit_list(long long unit, uint32_t c, unsigned long long mean, int tiles_x, int spp) {
    if ((unsigned long long)c > RSX_HEAVY_FACTOR * mean) return 0;
    return 1 + unit_xcd(unit, tiles_x, spp);
}

// The work lists of a pass whose unit costs are not known (a lane's first pass over these units, every pass of more than
// RSX_LPT_MAX_UNITS units): no heavy list, every unit in the list of its XCD, in about natural order. Three small launches over all CUs —
// count per block, scan, scatter — instead of the one-workgroup sort below, which walked 4.2 M units of a configs[2] pass three times
// on one CU (9 ms, a third of the pass it prepared).
#define ORDER_BLOCK_UNITS 16384
__global__ __launch_bounds__(1024) void k_order_natural_count(uint32_t *block_counts, long long n, int tiles_x, int spp) {
    __shared__ unsigned int h[8];
    if (threadIdx.x < 8) h[threadIdx.x] = 0;
    __syncthreads();
    const long long begin = (long long)blockIdx.x * ORDER_BLOCK_UNITS, end = begin + ORDER_BLOCK_UNITS < n ? begin + ORDER_BLOCK_UNITS : n;
    unsigned int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const int x = unit_xcd(i, tiles_x, spp);
#pragma unroll
        for (int k = 0; k < 8; ++k) mine[k] += x == k;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (mine[k]) atomicAdd(&h[k], mine[k]);
    __syncthreads();
    if (threadIdx.x < 8) block_counts[(size_t)blockIdx.x * 8 + threadIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_order_natural_scan(uint32_t *block_counts, uint32_t *seg, int n_blocks) {
    // thread x < 8 walks list x's per-block counts and turns them into offsets inside the list; then the lists are laid end to end
    __shared__ unsigned int totals[8];
    if (threadIdx.x < 8) {
        unsigned int run = 0;
        for (int b = 0; b < n_blocks; ++b) { const unsigned int c = block_counts[(size_t)b * 8 + threadIdx.x]; block_counts[(size_t)b * 8 + threadIdx.x] = run; run += c; }
        totals[threadIdx.x] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        seg[0] = 0;                                          // (list 0, the heavy units: empty)
        for (int x = 0; x < 8; ++x) { seg[1 + x] = run; run += totals[x]; }
        seg[9] = run;
    }
}
__global__ __launch_bounds__(1024) void k_order_natural_scatter(const uint32_t *block_counts, const uint32_t *seg, uint32_t *order, long long n, int tiles_x, int spp) {
    __shared__ unsigned int at[8];
    if (threadIdx.x < 8) at[threadIdx.x] = seg[1 + threadIdx.x] + block_counts[(size_t)blockIdx.x * 8 + threadIdx.x];
    __syncthreads();
    const long long begin = (long long)blockIdx.x * ORDER_BLOCK_UNITS, end = begin + ORDER_BLOCK_UNITS < n ? begin + ORDER_BLOCK_UNITS : n;
    for (long long i = begin + threadIdx.x; i < end; i += blockDim.x) order[atomicAdd(&at[unit_xcd(i, tiles_x, spp)], 1u)] = (uint32_t)i;
}

// One workgroup: counting sort of the units by (list, descending cost bucket). Splitting a heavy unit over several waves was tried
// and dropped: a silhouette tile is bound by its single slowest ray, so parts only multiplied the waves.
// flat != 0 (path passes): only the heavy units are pulled forward (list 0, longest first); the rest keep the natural order of their
// XCD list — sorting ALL units of a path pass by cost scatters neighbouring pixels over the chip and costs more than the tail it removes.
__global__ __launch_bounds__(1024) void k_order_units(uint32_t *cost, uint32_t *order, uint32_t *seg, long long n, int tiles_x, int spp, int flat) {
    __shared__ unsigned int hist[9][ORDER_BUCKETS];
    __shared__ unsigned int offset[9][ORDER_BUCKETS];
    __shared__ unsigned long long total;
    unsigned int *hflat = &hist[0][0];
    for (int b = threadIdx.x; b < 9 * ORDER_BUCKETS; b += blockDim.x) hflat[b] = 0;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned long long part_sum = 0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) part_sum += cost[i];
    atomicAdd(&total, part_sum);
    __syncthreads();
    const unsigned long long mean = total / (unsigned long long)n + 1;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t c = cost[i];
        const int list = unit_list(i, c, mean, tiles_x, spp);
        atomicAdd(&hist[list][flat && list ? 0 : cost_bucket(c)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int x = 0; x < 9; ++x) {
            seg[x] = run;
            for (int b = ORDER_BUCKETS - 1; b >= 0; --b) { offset[x][b] = run; run += hist[x][b]; }
        }
        seg[9] = run;
    }
    __syncthreads();
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t c = cost[i];
        const int list = unit_list(i, c, mean, tiles_x, spp);
        order[atomicAdd(&offset[list][flat && list ? 0 : cost_bucket(c)], 1u)] = (uint32_t)i;
        if (flat) cost[i] = 0u;                            // (path passes measure every pass: consumed here, zero for the next one — no memset)
    }
}

// One thread per (task, bin): sequential Welford over the task's spp samples in sample order
// (SpectralRadiance/PowerPixelProcessor.add_sample, pipeline/spectral/power.pyx:468-486, radiance.pyx:245-263).
// frame == null: write per-task (mean, variance) like _render_pixel packs them; else merge into the
// device-resident frame with the combine_samples law (Pipeline2D.update, power.pyx:424-437).
struct AccumParams {
    const Sample *samples;
    const double *tables;
    const int32_t *tasks;
    long long n_tasks;
    int32_t rect[4];
    int32_t ny, bins, spp, power;
    int32_t n_tables, tables_in_lds;    // tables_in_lds = 0: the spectral tables do not fit the kernel's LDS next to the reciprocals: read from global
    double sensitivity;
    double *mean, *variance;            // per-task outputs [n_tasks, bins] (or null)
    double *fmean, *fvar; int32_t *fn;  // frame [nx, ny, frame_bins] (or null)
    int32_t frame_bins, slice_offset;
    unsigned long long *ticket;         // work tickets of the trace kernel: re-armed here for the next launch
    const struct PathTerm *pool;        // path terms (k_render_trace_path) or null: block b = pool[b * PATH_BLOCK ..]
    const int32_t *tail;                // last block of each sample's list; Sample.pad = slots used in it
    long long n_records;                // block ids below this are the samples' own first blocks, above it arena blocks (slot 0 = link)
    double roulette_norm;               // 1 / (1 - extinction_prob)
    const double *consts;               // [i] = {(double)i, refine_rcp(i)} for i = 0 .. ACC_RCP_TABLE_MAX + 2 (k_fill_acc_consts), or null
    int32_t passes, pad_passes;         // consecutive passes of spp samples per pixel in the records (rsx_render_desc.passes; 1 = the usual case)
    const unsigned int *abort_flags;    // deferred path passes: PathStore::flags of the trace kernel — a pass whose arena ran out (bit 0), that hit the
                                        // segment guard (1) or met too many volumes at a point (2) is left out of the frame and rendered again by the caller
    unsigned int *zero;                 // scheduling state of the lane's trace kernels that this kernel leaves zeroed for their next pass, like the
    long long zero_n;                   // tickets (the redo mask of a CSG path pass): zero_n words, or 0
};

// The tickets (two sets: the second serves the redo pass of a CSG path pass) and the words at `zero` are re-armed here for the lane's next
// pass: every hipMemsetAsync between the launches of an overlapping slice waits ~0.3 ms for a free place on a chip full of persistent
// path workgroups. Stream order: the trace kernels that used them have finished.
__device__ __forceinline__ void accumulate_rearm(const AccumParams &ap, long long gid) {
    if (gid < 18 && ap.ticket) ap.ticket[16 * gid] = 0ULL;
    for (long long z = gid; z < ap.zero_n; z += (long long)gridDim.x * blockDim.x) ap.zero[z] = 0u;
}

__global__ void k_fill_acc_consts(double *consts, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { consts[2 * i] = (double)i; consts[2 * i + 1] = refine_rcp((double)i); }
}

// Thread order: bin fastest, then iy, then ix (rect mode) — the order of the x-major frame and of the sample records the trace
// kernel wrote, so both streams are read and written as contiguous runs. Task-list mode keeps task order.
#define ACC_RCP_TABLE_MAX 4096      // samples per pixel per pass up to which the reciprocal table is kept in LDS
#ifndef ACC_BATCH
#define ACC_BATCH 4                 // sample records whose loads are issued together
#endif
#ifndef RSX_ACC_TOUCH
#define RSX_ACC_TOUCH 0             // path passes: a chunk's term blocks requested together before the walk (k_accumulate) — measured (round 5,
                                    // Cornell box 1024^2 x 16 spp): the replay 5.13 -> 5.72 ms with the touches; the walk's own loads already overlap across the
                                    // eight waves of a SIMD, and the touches only add instructions and a full stop per chunk. Off.
#endif
#ifndef ACC_TERM_BATCH
#define ACC_TERM_BATCH 1            // path passes: terms of a block requested together in k_accumulate's walk — measured on the Cornell box (replay alone):
                                    // 1: 5.06 ms (56 registers, eight waves), 2: 5.21, 3: 5.93 (72, seven), 4: 6.62 (78, six). Off.
#endif
#ifndef RSX_ACC_SHARE_TOUCH
#define RSX_ACC_SHARE_TOUCH 0       // (measured: 5.06 -> 5.05 ms on the Cornell box: the walk does not wait for misses but for every round trip) path passes: the bins of a pixel ask for the chunk's term lines between them before the walk (k_accumulate)
#endif
#ifndef ACC_PATH_CHUNK
#define ACC_PATH_CHUNK 8            // path passes: samples whose term lists a lane walks back to back before the wave meets for their Welford steps
                                    // (Cornell box: 4 -> 5.4 ms, 8 -> 5.2 ms, 16 -> 8.8 ms: the values wait in LDS, 2 KB per sample and workgroup)
#endif

template <bool STAGED, int VOL, bool TAB_LDS = STAGED, bool MULTI = false>   // STAGED = many samples per pixel: LDS tables, batched record loads; else the lean one-shot form.
                                        // VOL = samples carry path terms (k_render_trace_path): 1 = without, 2 = with dielectric attenuation (pow() costs 60 registers)
                                        // TAB_LDS = false with STAGED: the spectral tables do not fit the LDS next to the reciprocals and are read from global
                                        // memory (a template parameter: as a run-time flag the choice cost the recurrence 23 %, 5.8 -> 7.1 ms on configs[2])
__global__ __launch_bounds__(256) void k_accumulate(AccumParams ap) {
    // LDS: refined reciprocals of 1 .. spp (the Welford divisors are the same for every pixel) and the spectral tables
    extern __shared__ __attribute__((aligned(16))) double acc_lds[];
    constexpr bool staged = STAGED;                         // few samples per pixel: not worth a barrier, read the tables from global
    const bool rcp_table = staged && ap.spp <= ACC_RCP_TABLE_MAX;
    const int n_rcp = rcp_table ? ap.spp + 2 : 2;
    double *acc_rcp = acc_lds, *acc_tab = acc_lds + n_rcp;
    constexpr bool tab_lds = STAGED && TAB_LDS;
    if (staged) {
        for (int d = threadIdx.x + 1; d < n_rcp; d += blockDim.x) acc_rcp[d] = refine_rcp((double)d);
        if (tab_lds) for (int e = threadIdx.x; e < ap.n_tables * ap.bins; e += blockDim.x) acc_tab[e] = ap.tables[e];
        __syncthreads();
    }
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = ap.n_tasks * ap.bins;
    accumulate_rearm(ap, gid);
    if (gid >= total) return;
    if (ap.abort_flags && (*ap.abort_flags & 7u)) return;
    long long p;
    int b;
    if (total < (1LL << 31)) { p = (uint32_t)gid / (uint32_t)ap.bins; b = (int)((uint32_t)gid % (uint32_t)ap.bins); }   // 32-bit divide when it fits
    else { p = gid / ap.bins; b = (int)(gid % ap.bins); }
    long long k = p;                                        // task index (row-major in rect mode): addresses the per-task outputs
    int ix, iy;
    if (ap.tasks) { ix = ap.tasks[2 * p]; iy = ap.tasks[2 * p + 1]; }
    else {
        const int w = ap.rect[2] - ap.rect[0], h = ap.rect[3] - ap.rect[1];
        const int lx = (int)((uint32_t)p / (uint32_t)h), ly = (int)((uint32_t)p % (uint32_t)h);
        ix = ap.rect[0] + lx; iy = ap.rect[1] + ly;
        k = (long long)ly * w + lx;
    }
    const Sample *s = ap.samples + p * ap.spp * (MULTI ? ap.passes : 1);   // (passes > 1: the records of a pixel's passes follow one another)
    // One term of a path's list applied to the running value x (newest term first: the reference's recursion unwinding).
    // VOL: += (0 + table * scale) * length (uniform.pyx:129-131, homogeneous.pyx:99-100). LAMBERT: [daughter's roulette normalisation,
    // ray.pyx:399] * reflectivity * pdf * (1 / pdf) (lambert.pyx:101-103, material.pyx:356-358).
    // One straight line for every kind — the lanes of a wave replay the lists of four or five pixels, whose terms differ in kind at
    // every step: as branches each step ran every kind's code in turn (and `||` of two kind tests became branches again: the tests
    // below are arithmetic). A factor a kind does not have is 1.0 — x * 1.0 = x exactly, any x: NORM terms are stored with a = b = 1 —
    // and the emission sum is chosen by a select. 9.2 -> 8.3 ms on the Cornell box.
    auto term_table = [&](const PathTerm &tm) { return tab_lds ? acc_tab[tm.table * ap.bins + b] : ap.tables[tm.table * ap.bins + b]; };
    auto apply_term_with = [&](double x, const PathTerm &tm, const double tv) {
        static_assert(TERM_LAMBERT == 1 && TERM_LAMBERT_NORM == 2 && TERM_NORM == 4 && TERM_ATTEN == 5, "kind tests below");
        const bool has_norm = ((0x14u >> (unsigned)tm.kind) & 1u) != 0u;
        const double emission = 0.0 + tv * tm.b;
        const double with_emission = x + emission * tm.a;
        double y = x * (has_norm ? ap.roulette_norm : 1.0);
        y = y * (tm.kind == TERM_NORM ? 1.0 : tv);
        y = y * tm.a;
        y = y * tm.b;
        y = tm.kind == TERM_VOL ? with_emission : y;
        if constexpr (VOL == 2) {                                  // dielectric.pyx:325-326; pow(1, length) = 1 exactly
            if (tm.kind == TERM_ATTEN) y = tv != 1.0 ? x * portable_pow(tv, tm.a) : x;
        }                                                          // (VOL == 1: no dielectric absorbs, and none left a term — render())
        return y;
    };
    auto apply_term = [&](double x, const PathTerm &tm) { return apply_term_with(x, tm, term_table(tm)); };
    // x = (a * table[bin]) * weight [* sensitivity] — optical/ray.pyx:391-393, observer.pyx:408; absorbers (table < 0) give 0
    auto value = [&](const Sample &smp, long long record) {
        const int e = (smp.table < 0 ? 0 : smp.table) * ap.bins + b;
        const double tab = tab_lds ? acc_tab[e] : ap.tables[e];
        double x = smp.table < 0 ? 0.0 : smp.a * tab;
        if (VOL) {
            long long blk = ap.tail[record];
            int n = smp.pad;
            for (;;) {
                const PathTerm *t = ap.pool + blk * PATH_BLOCK;
                const int first = blk < ap.n_records ? 0 : 1;
                for (int j = n - 1; j >= first; --j) {
                    x = apply_term(x, t[j]);       // (fetching term j - 1 before term j is used: 8.3 -> 8.7 ms)
                }
                if (first == 0) break;
                blk = t[0].table;
                n = PATH_BLOCK;
            }
        }
        x = x * smp.weight;
        if (ap.power) x = x * ap.sensitivity;
        return x;
    };
    // _add_sample (statsarray.pyx:743-776) unrolled over the pass: the first sample sets (m, 0); sample i >= 1 divides by the new
    // count i + 1 and by i, and scales the previous variance by prev_n - 1 with prev_n := 2 when only one sample was held.
    // Records are fetched ACC_BATCH at a time so that their loads are in flight together (one dependent load per sample was the bound).
    long long rec0 = p * ap.spp * (MULTI ? ap.passes : 1);
    double m = 0, v = 0;
    double dm = 1.0;                                        // (double)i, advanced by exact additions
    // The step's divisors, their refined reciprocals and the factor prev_n - 1 depend on the sample index alone — wave-uniform: they come
    // from a table over the scalar data path (ap.consts) and cost the recurrence no vector instruction. The two quotients are formed by
    // exact_div's shortcut for the whole wave at once, and ONE wave-level test per step checks that every lane's numerators were in its
    // range (|n| in [2^-300, 2^300], or +0 whose quotient the shortcut also gets right); a step where some lane's was not — a
    // denormal-sized difference, a -0 — is redone with exact_div proper. Same bits either way (rsx_selftest_welford against the
    // reference's states), 35 -> 24 vector instructions per sample and bin.
    const RSX_CONST_AS double *consts = (const RSX_CONST_AS double *)(unsigned long long)ap.consts;
    const bool fast_steps = rcp_table && ap.consts != nullptr;
    auto quotient_ok = [](double n) {                        // (as lane masks: the algebra and the branch run on the scalar unit)
        const double a = __builtin_fabs(n);
        return (pkt_mask(a >= 0x1p-300) & pkt_mask(a <= 0x1p+300)) | mask_plus_zero(n);
    };
    auto step = [&](double x, int i) {
        if (fast_steps) {
            const double dm_u = consts[2 * i], ym = consts[2 * i + 1], dn = consts[2 * i + 2], yn = consts[2 * i + 3];
            const double c = i == 1 ? 1.0 : consts[2 * i - 2];
            const double pm = m, pv = v;
            const double n1 = x - pm;
            const double q1 = __builtin_fma(__builtin_fma(-dn, n1 * yn, n1), yn, n1 * yn);
            const double m1 = pm + q1;
            const double n2 = pv * c + n1 * (x - m1);
            const double q2 = __builtin_fma(__builtin_fma(-dm_u, n2 * ym, n2), ym, n2 * ym);
            if (__builtin_expect((quotient_ok(n1) & quotient_ok(n2)) != pkt_mask(true), 0)) {
                m = pm + exact_div(n1, dn, yn, true);
                v = exact_div(pv * c + n1 * (x - m), dm_u, ym, true);
            } else { m = m1; v = q2; }
            return;
        }
        const double dn = dm + 1.0, c = i == 1 ? 1.0 : dm - 1.0;
        const double yn = rcp_table ? acc_rcp[i + 1] : refine_rcp(dn), ym = rcp_table ? acc_rcp[i] : refine_rcp(dm);
        const double pm = m, pv = v;
        m = pm + exact_div(x - pm, dn, yn, true);
        v = exact_div(pv * c + (x - pm) * (x - m), dm, ym, true);
        dm = dn;
    };
    if constexpr (VOL != 0 && STAGED) {
        // Path passes. The lanes of a wave are the bins of four or five pixels, and the lists of those pixels' samples differ in length
        // (path lengths are geometric): walked sample by sample, every lane waited for the longest list of every sample and 0.4 of the
        // lane-slots did work. Here every lane walks ITS lists one after the other — a step is one term of whatever sample the lane
        // has reached — over ACC_PATH_CHUNK samples at a time, leaving each sample's value in LDS; the Welford steps of the chunk then
        // run for all lanes together (their divisors depend on the sample index alone and come over the scalar data path).
        // Same operations in the same order per (pixel, bin); the loop has one exit (see the toolchain note in dev_csg.hpp).
        // MULTI (rsx_render_desc.passes = K): the pixel's K * spp records are K passes — per pass the recurrence from its first sample
        // and the frame merge, in pass order, the frame cell in registers between the merges: the frame of K calls.
        double *acc_xs = acc_tab + (tab_lds ? ap.n_tables * ap.bins : 1);      // [ACC_PATH_CHUNK][blockDim.x]
        const int n_pass_v = MULTI ? ap.passes : 1;
        const size_t ff = ((size_t)ix * ap.ny + iy) * ap.frame_bins + ap.slice_offset + b;
        double fm_v = 0, fv_v = 0;
        int fn_v = 0;
        if (MULTI && ap.fmean) { fm_v = ap.fmean[ff]; fv_v = ap.fvar[ff]; fn_v = ap.fn[ff]; }
        for (int pass = 0; pass < n_pass_v; ++pass) {
        const Sample *sp = s + (long long)pass * ap.spp;
        const long long rec_base = (p * n_pass_v + pass) * ap.spp;
        m = 0; v = 0; dm = 1.0;
        for (int chunk0 = 0; chunk0 < ap.spp; chunk0 += ACC_PATH_CHUNK) {
            const int cn = ap.spp - chunk0 < ACC_PATH_CHUNK ? ap.spp - chunk0 : ACC_PATH_CHUNK;     // (wave-uniform)
#if RSX_ACC_SHARE_TOUCH
            // The lanes of a pixel (its bins) walk the same lists, one dependent line after the other: a wave has the lists of four or five
            // pixels in flight and waits out every miss. Before the walk the pixel's lanes ask for the chunk's lists BETWEEN them: the lane of
            // bin b < cn for the line in which sample chunk0 + b's walk begins (its newest term), the lanes behind them for the line before
            // it — one request per lane, the chunk's lists in flight together, one wait; the walk then finds its lines in L2.
            {
                const int which = b < cn ? b : b - cn < cn ? b - cn : -1;                  // (bins < 2 cn: some lines are left to the walk)
                float sink = 0.0f;
                if (which >= 0) {
                    const Sample ts = sp[chunk0 + which];
                    const long long tb = ap.tail[rec_base + chunk0 + which];
                    const int first_t = tb < ap.n_records ? 0 : 1;
                    int slot = ts.pad - 1 - (b < cn ? 0 : 5);                              // (a 128-byte line holds five terms and a third)
                    slot = slot < first_t ? first_t : slot;
                    sink = *reinterpret_cast<const float *>(ap.pool + tb * PATH_BLOCK + slot);
                }
                asm volatile("" :: "v"(sink));
            }
#endif
#if RSX_ACC_TOUCH
            // The chunk's lists are asked for TOGETHER before any of them is walked: the lines of every sample's last block (where its walk
            // begins; a path of up to fifteen terms has no other) — left to the walk, a lane met one miss after the other, three lines per
            // block and sample, each a round trip to HBM with nothing else of the lane in flight.
            {
                float sink = 0.0f;
                for (int qi = 0; qi < cn; ++qi) {
                    const char *tb = reinterpret_cast<const char *>(ap.pool + (long long)ap.tail[rec_base + chunk0 + qi] * PATH_BLOCK);
                    sink += *reinterpret_cast<const float *>(tb) + *reinterpret_cast<const float *>(tb + 128) + *reinterpret_cast<const float *>(tb + 256);
                }
                asm volatile("" :: "v"(sink));
            }
#endif
            int si = 0;
            Sample cur = sp[chunk0], nxt = cur;
            long long blk = ap.tail[rec_base + chunk0], nblk = blk;
            if (cn > 1) { nxt = sp[chunk0 + 1]; nblk = ap.tail[rec_base + chunk0 + 1]; }
            // (the walk keeps a pointer to the term it is at and one to where the block's terms end — slot 0 of an arena block is its link —
            // instead of forming pool + 24 (16 block + slot) at every step)
            int first = blk < ap.n_records ? 0 : 1;
            const PathTerm *tp = ap.pool + blk * PATH_BLOCK + (cur.pad - 1), *floor = ap.pool + blk * PATH_BLOCK + (first - 1);
            double x = cur.table < 0 ? 0.0 : cur.a * (tab_lds ? acc_tab[cur.table * ap.bins + b] : ap.tables[cur.table * ap.bins + b]);
            bool live = true;
            while (live) {
#if ACC_TERM_BATCH > 1
                // (an experiment kept for its measurement: where the block still holds ACC_TERM_BATCH terms they are requested together, then
                // their table entries together, then applied in the list's order — see ACC_TERM_BATCH: slower the wider the batch)
                if (tp - floor >= ACC_TERM_BATCH) {
                    PathTerm tb[ACC_TERM_BATCH];
                    double tvb[ACC_TERM_BATCH];
#pragma unroll
                    for (int j = 0; j < ACC_TERM_BATCH; ++j) tb[j] = tp[-j];
#pragma unroll
                    for (int j = 0; j < ACC_TERM_BATCH; ++j) tvb[j] = term_table(tb[j]);
#pragma unroll
                    for (int j = 0; j < ACC_TERM_BATCH; ++j) x = apply_term_with(x, tb[j], tvb[j]);
                    tp -= ACC_TERM_BATCH;
                } else
#endif
                if (tp > floor) {
                    x = apply_term(x, *tp);                                    // (term j - 1 requested before term j is applied: 5.2 -> 5.5 ms)
                    --tp;
                } else if (first == 1) {                                       // an arena block: slot 0 links to the block before it
                    blk = floor->table;
                    first = blk < ap.n_records ? 0 : 1;
                    tp = ap.pool + blk * PATH_BLOCK + (PATH_BLOCK - 1); floor = ap.pool + blk * PATH_BLOCK + (first - 1);
                } else {                                                       // this sample's list is done
                    x = x * cur.weight;
                    if (ap.power) x = x * ap.sensitivity;
                    acc_xs[si * blockDim.x + threadIdx.x] = x;
                    ++si;
                    if (si < cn) {
                        cur = nxt; blk = nblk;
                        first = blk < ap.n_records ? 0 : 1;
                        tp = ap.pool + blk * PATH_BLOCK + (cur.pad - 1); floor = ap.pool + blk * PATH_BLOCK + (first - 1);
                        x = cur.table < 0 ? 0.0 : cur.a * (tab_lds ? acc_tab[cur.table * ap.bins + b] : ap.tables[cur.table * ap.bins + b]);
                        if (si + 1 < cn) { nxt = sp[chunk0 + si + 1]; nblk = ap.tail[rec_base + chunk0 + si + 1]; }
                    } else live = false;
                }
            }
            for (int i = 0; i < cn; ++i) {
                const double xi = acc_xs[i * blockDim.x + threadIdx.x];
                if (chunk0 + i == 0) { m = xi; v = 0; } else step(xi, chunk0 + i);
            }
        }
        if constexpr (MULTI) {
            if (ap.fmean) {
                if (v < 0) v = 0;                                                 // statsarray.pyx:649-650
                double mt, vt;
                int nt;
                combine_samples_uniform(fm_v, fv_v, fn_v, m, v, ap.spp, mt, vt, nt, consts, ACC_CONSTS_ENTRIES);
                fm_v = mt; fv_v = vt; fn_v = nt;
            }
        }
        }
        if constexpr (MULTI) {
            if (ap.fmean) { ap.fmean[ff] = fm_v; ap.fvar[ff] = fv_v; ap.fn[ff] = fn_v; }
            return;
        }
        if (ap.mean) { ap.mean[k * ap.bins + b] = m; ap.variance[k * ap.bins + b] = v; }
        if (ap.fmean) {
            if (v < 0) v = 0;                                                     // statsarray.pyx:649-650
            double mt, vt;
            int nt;
            combine_samples(ap.fmean[ff], ap.fvar[ff], ap.fn[ff], m, v, ap.spp, mt, vt, nt);
            ap.fmean[ff] = mt; ap.fvar[ff] = vt; ap.fn[ff] = nt;
        }
        return;
    }
    // rsx_render_desc.passes = K > 1 (MULTI): K consecutive passes of spp samples each in this one launch — per pass the recurrence from its
    // first sample and the frame merge, in pass order: the frame of K calls. The frame cell stays in registers between the merges. A kernel
    // of its own: as a run-time trip count the pass loop cost the one-pass kernels 10 - 20 % (Cornell box replay 9.2 -> 11.3 ms).
    const int n_pass = MULTI ? ap.passes : 1;
    size_t f = 0;
    double fm = 0, fv = 0;
    int fcount = 0;
    if (MULTI && ap.fmean) {
        f = ((size_t)ix * ap.ny + iy) * ap.frame_bins + ap.slice_offset + b;
        fm = ap.fmean[f]; fv = ap.fvar[f]; fcount = ap.fn[f];
    }
#if RSX_ACC_TOUCH
    if constexpr (VOL != 0 && MULTI) {                       // the term blocks of (up to sixteen of) the call's records, requested together
        const int n_touch = n_pass * ap.spp < 16 ? n_pass * ap.spp : 16;
        float sink = 0.0f;
        for (int qi = 0; qi < n_touch; ++qi) {
            const char *tb = reinterpret_cast<const char *>(ap.pool + (long long)ap.tail[rec0 + qi] * PATH_BLOCK);
            sink += *reinterpret_cast<const float *>(tb) + *reinterpret_cast<const float *>(tb + 128) + *reinterpret_cast<const float *>(tb + 256);
        }
        asm volatile("" :: "v"(sink));
    }
#endif
    Sample head = s[0];
    for (int pass = 0; pass < n_pass; ++pass, s += ap.spp, rec0 += ap.spp) {
    // (the first record of the NEXT pass is requested before this pass's merge: K one-sample passes are K dependent merges per thread, and
    // a load in front of each was what they waited for — 0.99 -> 0.88 ms for 16 passes of configs[1]; fetching the table entry ahead as
    // well measured 1.10 ms)
    const Sample first = head;
    if (pass + 1 < n_pass) head = s[ap.spp];
    m = value(first, rec0); v = 0; dm = 1.0;
    // ... and the NEXT batch is requested before this one is stepped through: the records are a stream that is read once, from HBM,
    // and a batch's own arithmetic (0.2 us) does not cover that round trip even with eight waves per SIMD.
    int i = 1;
    if (i + ACC_BATCH <= ap.spp) {
        Sample cur[ACC_BATCH];
#pragma unroll
        for (int j = 0; j < ACC_BATCH; ++j) cur[j] = s[i + j];
        for (; i + ACC_BATCH <= ap.spp; i += ACC_BATCH) {
            Sample nxt[ACC_BATCH];
            // (samples without path terms only: the replay of a term list needs the registers itself — Cornell box 9.2 -> 13.1 ms with it.
            // Also measured there and not adopted: the four lists of a batch replayed together, one term of each per turn — 11.6 ms: the
            // replay waits on HBM for the term blocks, 6.4 GB per launch, not on its own dependent multiplications)
            const bool more = VOL == 0 && i + 2 * ACC_BATCH <= ap.spp;     // (wave-uniform)
            if (more) {
#pragma unroll
                for (int j = 0; j < ACC_BATCH; ++j) nxt[j] = s[i + ACC_BATCH + j];
            }
#pragma unroll
            for (int j = 0; j < ACC_BATCH; ++j) step(value(cur[j], rec0 + i + j), i + j);
            if (more) {
#pragma unroll
                for (int j = 0; j < ACC_BATCH; ++j) cur[j] = nxt[j];
            } else if (VOL != 0 && i + 2 * ACC_BATCH <= ap.spp) {
#pragma unroll
                for (int j = 0; j < ACC_BATCH; ++j) cur[j] = s[i + ACC_BATCH + j];
            }
        }
    }
    for (; i < ap.spp; ++i) step(value(s[i], rec0 + i), i);
    if (MULTI && ap.fmean) {
        if (v < 0) v = 0;                                                     // statsarray.pyx:649-650
        double mt, vt;
        int nt;
        combine_samples_uniform(fm, fv, fcount, m, v, ap.spp, mt, vt, nt, consts, ACC_CONSTS_ENTRIES);
        fm = mt; fv = vt; fcount = nt;
    }
    }
    if (ap.mean) { ap.mean[k * ap.bins + b] = m; ap.variance[k * ap.bins + b] = v; }    // (per-task outputs: passes == 1)
    if (MULTI) {
        if (ap.fmean) { ap.fmean[f] = fm; ap.fvar[f] = fv; ap.fn[f] = fcount; }
    } else if (ap.fmean) {
        f = ((size_t)ix * ap.ny + iy) * ap.frame_bins + ap.slice_offset + b;
        if (v < 0) v = 0;                                                     // statsarray.pyx:649-650
        double mt, vt;
        int nt;
        combine_samples(ap.fmean[f], ap.fvar[f], ap.fn[f], m, v, ap.spp, mt, vt, nt);
        ap.fmean[f] = mt; ap.fvar[f] = vt; ap.fn[f] = nt;
    }
}

// XYZPixelProcessor (rgb.pyx:534-562): one thread per (task, channel). Every sample's spectrum, times its projection weight, is
// projected on the channel's resampled CIE curve in bin order (spectrum_to_ciexyz, colour.pyx:176-186: x += delta * sample * curve),
// scaled by the pixel sensitivity and fed to the same Welford recurrence as k_accumulate. The curves are rows xyz_table0 + channel of
// the spectral tables. Outputs are per task: RGBPipeline2D sums them over the spectral slices on the host.
template <int VOL>                      // 0 = no path terms, 1 = path terms, 2 = path terms with absorbing dielectrics (pow), as in k_accumulate
__global__ __launch_bounds__(256) void k_accumulate_xyz(AccumParams ap, int xyz_table0, double delta) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = ap.n_tasks * 3;
    accumulate_rearm(ap, gid);
    if (gid >= total) return;
    const long long p = gid / 3;
    const int c = (int)(gid % 3);
    long long k = p;
    if (!ap.tasks) {
        const int w = ap.rect[2] - ap.rect[0], h = ap.rect[3] - ap.rect[1];
        const int lx = (int)(p / h), ly = (int)(p % h);
        k = (long long)ly * w + lx;
    }
    const Sample *s = ap.samples + p * ap.spp;
    const double *curve = ap.tables + (size_t)(xyz_table0 + c) * ap.bins;
    auto value = [&](const Sample &smp, long long record) {
        double acc = 0;
        for (int b = 0; b < ap.bins; ++b) {
            double x = smp.table < 0 ? 0.0 : smp.a * ap.tables[smp.table * ap.bins + b];
            if (VOL) {
                long long blk = ap.tail[record];
                int n = smp.pad;
                for (;;) {
                    const PathTerm *t = ap.pool + blk * PATH_BLOCK;
                    const int first = blk < ap.n_records ? 0 : 1;
                    for (int j = n - 1; j >= first; --j) {
                        const PathTerm tm = t[j];
                        const double tv = ap.tables[tm.table * ap.bins + b];
                        if (tm.kind == TERM_VOL) { const double emission = 0.0 + tv * tm.b; x = x + emission * tm.a; }
                        else if (tm.kind == TERM_ATTEN) { if constexpr (VOL == 2) { if (tv != 1.0) x = x * portable_pow(tv, tm.a); } }
                        else if (tm.kind == TERM_NORM) x = x * ap.roulette_norm;
                        else {
                            if (tm.kind == TERM_LAMBERT_NORM) x = x * ap.roulette_norm;
                            x = x * tv; x = x * tm.a; x = x * tm.b;
                        }
                    }
                    if (first == 0) break;
                    blk = t[0].table;
                    n = PATH_BLOCK;
                }
            }
            x = x * smp.weight;
            acc += delta * x * curve[b];
        }
        return acc * ap.sensitivity;
    };
    const long long rec0 = p * ap.spp;
    double m = value(s[0], rec0), v = 0;
    double dm = 1.0;
    for (int i = 1; i < ap.spp; ++i) {
        const double x = value(s[i], rec0 + i);
        const double dn = dm + 1.0, cc = i == 1 ? 1.0 : dm - 1.0;
        const double pm = m, pv = v;
        m = pm + exact_div(x - pm, dn, refine_rcp(dn), true);
        v = exact_div(pv * cc + (x - pm) * (x - m), dm, refine_rcp(dm), true);
        dm = dn;
    }
    ap.mean[k * 3 + c] = m; ap.variance[k * 3 + c] = v;
}

// Several ONE-SAMPLE passes of a path-traced call (rsx_render_desc.passes = K with spp = 1: the step of configs[4], 512 one-bin slices x K
// passes), in two kernels. k_accumulate's multi-pass form gives a (pixel, bin) thread the K records of its pixel to replay one after the
// other — K dependent pointer walks per thread, every lane of a wave in a different list (one bin per slice: no two lanes share a
// load): at K = 8 it ran 18 ms per slice against 1.25 ms for one pass, a fifth of the GPU time of a configs[4] step
// (profiles/r05c_c5_kernel_stats.csv). Here every (pixel, bin, record) gets a thread of its own for the replay — K times the lists in
// flight, records and tail ids read as contiguous runs — and leaves its value x in `xs`; k_merge_passes then merges a pixel's K values
// into the frame cell in pass order. Per (pixel, bin) the same operations on the same values in the same order as k_accumulate<false, VOL,
// false, true> with spp = 1 (m = x, v = 0, combine_samples per pass): frames are equal bit for bit
// (test_several_path_passes_per_call_equal_separate_passes).
template <int VOL>
__global__ __launch_bounds__(256) void k_path_values(AccumParams ap, double *xs) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long R = ap.passes, total = ap.n_tasks * ap.bins * R;
    if (gid >= total) return;
    if (ap.abort_flags && (*ap.abort_flags & 7u)) return;
    long long pb;
    int r;
    if (total < (1LL << 31)) { pb = (uint32_t)gid / (uint32_t)R; r = (int)((uint32_t)gid % (uint32_t)R); }
    else { pb = gid / R; r = (int)(gid % R); }
    const long long p = pb / ap.bins;
    const int b = (int)(pb % ap.bins);
    const long long record = p * R + r;
    const Sample smp = ap.samples[record];
    // x = (a * table[bin]) * weight [* sensitivity] with the path's terms applied newest first — value() / apply_term of k_accumulate, tables from global
    const double tab = ap.tables[(smp.table < 0 ? 0 : smp.table) * ap.bins + b];
    double x = smp.table < 0 ? 0.0 : smp.a * tab;
    long long blk = ap.tail[record];
    int n = smp.pad;
    for (;;) {
        const PathTerm *t = ap.pool + blk * PATH_BLOCK;
        const int first = blk < ap.n_records ? 0 : 1;
        for (int j = n - 1; j >= first; --j) {
            const PathTerm tm = t[j];
            const double tv = ap.tables[tm.table * ap.bins + b];
            static_assert(TERM_LAMBERT == 1 && TERM_LAMBERT_NORM == 2 && TERM_NORM == 4 && TERM_ATTEN == 5, "kind tests below");
            const bool has_norm = ((0x14u >> (unsigned)tm.kind) & 1u) != 0u;
            const double emission = 0.0 + tv * tm.b;
            const double with_emission = x + emission * tm.a;
            double y = x * (has_norm ? ap.roulette_norm : 1.0);
            y = y * (tm.kind == TERM_NORM ? 1.0 : tv);
            y = y * tm.a;
            y = y * tm.b;
            y = tm.kind == TERM_VOL ? with_emission : y;
            if constexpr (VOL == 2) {                          // dielectric.pyx:325-326; pow(1, length) = 1 exactly
                if (tm.kind == TERM_ATTEN) y = tv != 1.0 ? x * portable_pow(tv, tm.a) : x;
            }
            x = y;
        }
        if (first == 0) break;
        blk = t[0].table;
        n = PATH_BLOCK;
    }
    x = x * smp.weight;
    if (ap.power) x = x * ap.sensitivity;
    xs[gid] = x;
}

__global__ __launch_bounds__(256) void k_merge_passes(AccumParams ap, const double *xs) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = ap.n_tasks * ap.bins;
    accumulate_rearm(ap, gid);
    if (gid >= total) return;
    if (ap.abort_flags && (*ap.abort_flags & 7u)) return;
    long long p;
    int b;
    if (total < (1LL << 31)) { p = (uint32_t)gid / (uint32_t)ap.bins; b = (int)((uint32_t)gid % (uint32_t)ap.bins); }
    else { p = gid / ap.bins; b = (int)(gid % ap.bins); }
    int ix, iy;
    if (ap.tasks) { ix = ap.tasks[2 * p]; iy = ap.tasks[2 * p + 1]; }
    else {
        const int h = ap.rect[3] - ap.rect[1];
        ix = ap.rect[0] + (int)((uint32_t)p / (uint32_t)h); iy = ap.rect[1] + (int)((uint32_t)p % (uint32_t)h);
    }
    const RSX_CONST_AS double *consts = (const RSX_CONST_AS double *)(unsigned long long)ap.consts;
    const size_t f = ((size_t)ix * ap.ny + iy) * ap.frame_bins + ap.slice_offset + b;
    double fm = ap.fmean[f], fv = ap.fvar[f];
    int fcount = ap.fn[f];
    const double *mine = xs + gid * ap.passes;
    for (int r = 0; r < ap.passes; ++r) {                       // one pass = one sample: (m, v, n) = (x, 0, 1), merged like Pipeline2D.update
        double mt, vt;
        int nt;
        combine_samples_uniform(fm, fv, fcount, mine[r], 0.0, 1, mt, vt, nt, consts, ACC_CONSTS_ENTRIES);
        fm = mt; fv = vt; fcount = nt;
    }
    ap.fmean[f] = fm; ap.fvar[f] = fv; ap.fn[f] = fcount;
}

__global__ __launch_bounds__(256) void k_frame_combine(long long n, double *ma, double *va, int32_t *na, const double *mb,
                                                       const double *vb, const int32_t *nb) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (nb[i] < 1) return;
    double v = vb[i];
    if (v < 0) v = 0;
    double mt, vt;
    int nt;
    combine_samples(ma[i], va[i], na[i], mb[i], v, nb[i], mt, vt, nt);
    ma[i] = mt; va[i] = vt; na[i] = nt;
}

