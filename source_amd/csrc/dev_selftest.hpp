// dev_selftest.hpp — part of librsx's device translation unit (included by rsx_device.hip after dev_render.hpp).
// Device-side known-answer entry points: each runs the very device functions (or kernels) the render path uses on caller-supplied
// operands, so that the reference's golden vectors for them (tests/golden: F2 box slabs, F8 camera rays, F9 Welford states) and the
// oracle's portable math can be compared with the DEVICE directly, not only through rendered frames.
#pragma once

// BoundingBox3D.intersect (core/boundingbox.pyx:180-245): out[n,3] = hit, front, back — through aabb_rcp (hoisted reciprocals: the
// form world_trace_wave and the mesh gate use) and, for comparison, aabb(); a mismatch between the two is counted.
__global__ void k_selftest_aabb(long long n, const double *lower, const double *upper, const double *origin, const double *direction, double *out,
                                unsigned long long *mismatch) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.ox = origin[3 * i]; r.oy = origin[3 * i + 1]; r.oz = origin[3 * i + 2];
    r.dx = direction[3 * i]; r.dy = direction[3 * i + 1]; r.dz = direction[3 * i + 2];
    r.maxd = INFINITY;
    const double rx = 1.0 / r.dx, ry = 1.0 / r.dy, rz = 1.0 / r.dz;
    double f, b, f2, b2;
    const bool hit = aabb_rcp(lower + 3 * i, upper + 3 * i, r, rx, ry, rz, f, b);
    const bool hit2 = aabb(lower + 3 * i, upper + 3 * i, r, f2, b2);
    if (hit != hit2 || __double_as_longlong(f) != __double_as_longlong(f2) || __double_as_longlong(b) != __double_as_longlong(b2)) atomicAdd(mismatch, 1ULL);
    out[3 * i] = hit ? 1.0 : 0.0; out[3 * i + 1] = f; out[3 * i + 2] = b;
}

// Primary rays exactly as k_render_trace / k_render_trace_path generate them (camera_ray): out[n_tasks * spp, 7] = origin, direction, weight
__global__ void k_selftest_camera(RenderParams rp, double *out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rp.n_tasks * rp.spp) return;
    const unsigned long long rp_bits = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    const RSX_CONST_AS RenderParams *q = (const RSX_CONST_AS RenderParams *)rp_bits;
    const long long k = g / rp.spp;
    const int s = (int)(g % rp.spp);
    int ix, iy;
    task_pixel(rp, k, ix, iy);
    double u1, u2;
    if (rp.rng_mode == RSX_RNG_STREAM) { u1 = rp.uniforms[2 * g]; u2 = rp.uniforms[2 * g + 1]; }
    else philox2(rp.seed, (uint64_t)ix * (uint64_t)rp.cam.ny + (uint64_t)iy, rp.sample_offset + (uint64_t)s, u1, u2);
    Ray r;
    double weight;
    camera_ray(q, ix, iy, u1, u2, r, weight);
    double *o = out + 7 * g;
    o[0] = r.ox; o[1] = r.oy; o[2] = r.oz; o[3] = r.dx; o[4] = r.dy; o[5] = r.dz; o[6] = weight;
}

// builds the Sample records k_accumulate consumes from plain values: x = (a * table) * weight with table = 1, weight = 1
__global__ void k_selftest_fill_samples(long long n, const double *x, Sample *s) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sample smp;
    smp.a = x[i]; smp.weight = 1.0; smp.table = 0; smp.pad = 0;
    s[i] = smp;
}

// op 0: portable_pow(a, b) -> out0; op 1: portable_sincos(a) -> (out0, out1); op 2: portable_asin(a) -> out0
__global__ void k_selftest_math(int op, long long n, const double *a, const double *b, double *out0, double *out1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 0) out0[i] = portable_pow(a[i], b[i]);
    else if (op == 1) { double sn, cs; portable_sincos(a[i], sn, cs); out0[i] = sn; out1[i] = cs; }
    else out0[i] = portable_asin(a[i]);
}
