// dev_analytic.hpp — part of librsx's single device translation unit (included by rsx_device.hip, in order).
// Sphere / box / cylinder roots and the Intersection records rebuilt from a hit.
#pragma once

// ---------------------------------------------------------------------------------------------------
// analytic primitives: ordered roots inside [0, max_distance]
//   sphere.pyx:115-159, box.pyx:157-294, cylinder.pyx:148-276, utility.pyx:376-419 (solve_quadratic)
// Each returns 0..2 roots: t[k] plus (a0, a1) = (face, axis|type) needed to rebuild the intersection.
// ---------------------------------------------------------------------------------------------------
#define NO_FACE (-1)
#define LOWER_FACE 0
#define UPPER_FACE 1
#define T_CYLINDER 0
#define T_SLAB 1

struct Roots {
    int n;
    double t[2];
    int32_t a0[2], a1[2];
};

__device__ __forceinline__ bool solve_quadratic(double a, double b, double c, double &t0, double &t1) {
    const double d = b * b - 4 * a * c;
    if (d < 0) return false;
    double q;
    if (b < 0) q = -0.5 * (b - sqrt(d)); else q = -0.5 * (b + sqrt(d));
    t0 = q / a;
    t1 = c / q;
    return true;
}

// shared tail of Sphere/Box/Cylinder.hit: choose closest root and whether a cached further root exists
__device__ __forceinline__ void pick_roots(double near_t, double far_t, int nf, int na, int ff, int fa, double maxd, Roots &out) {
    out.n = 0;
    if (near_t > far_t) return;                                              // (never true for the sphere's sorted roots)
    if (near_t > maxd || far_t < 0.0) return;
    if (near_t >= 0.0) {
        out.t[0] = near_t; out.a0[0] = nf; out.a1[0] = na; out.n = 1;
        if (far_t <= maxd) { out.t[1] = far_t; out.a0[1] = ff; out.a1[1] = fa; out.n = 2; }
    } else if (far_t <= maxd) {
        out.t[0] = far_t; out.a0[0] = ff; out.a1[0] = fa; out.n = 1;
    }
}

__device__ void sphere_roots(const rsx_primitive &p, const Ray &l, Roots &out) {
    out.n = 0;
    const double radius = p.params[0];
    const double a = l.dx * l.dx + l.dy * l.dy + l.dz * l.dz;
    const double b = 2 * (l.dx * l.ox + l.dy * l.oy + l.dz * l.oz);
    const double c = l.ox * l.ox + l.oy * l.oy + l.oz * l.oz - radius * radius;
    double t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return;
    if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
    pick_roots(t0, t1, 0, 0, 0, 0, l.maxd, out);
}

__device__ __forceinline__ void box_slab(int axis, double o, double d, double lo, double hi, double &near_t, double &far_t,
                                         int &nf, int &ff, int &na, int &fa) {
    double tmin, tmax;
    int fmin, fmax;
    const double inf = INFINITY;
    if (d != 0.0) {
        const double rcp = 1.0 / d;
        if (d > 0) { tmin = (lo - o) * rcp; tmax = (hi - o) * rcp; fmin = LOWER_FACE; fmax = UPPER_FACE; }
        else       { tmin = (hi - o) * rcp; tmax = (lo - o) * rcp; fmin = UPPER_FACE; fmax = LOWER_FACE; }
    } else {
        if (o < lo)      { tmin = -inf; tmax = -inf; }
        else if (o > hi) { tmin = inf;  tmax = inf; }
        else             { tmin = -inf; tmax = inf; }
        fmin = NO_FACE; fmax = NO_FACE;
    }
    if (tmin > near_t) { near_t = tmin; nf = fmin; na = axis; }
    if (tmax < far_t)  { far_t = tmax;  ff = fmax; fa = axis; }
}

__device__ void box_roots(const rsx_primitive &p, const Ray &l, Roots &out) {
    double near_t = -INFINITY, far_t = INFINITY;
    int nf = NO_FACE, ff = NO_FACE, na = -1, fa = -1;
    box_slab(0, l.ox, l.dx, p.params[0], p.params[3], near_t, far_t, nf, ff, na, fa);
    box_slab(1, l.oy, l.dy, p.params[1], p.params[4], near_t, far_t, nf, ff, na, fa);
    box_slab(2, l.oz, l.dz, p.params[2], p.params[5], near_t, far_t, nf, ff, na, fa);
    pick_roots(near_t, far_t, nf, na, ff, fa, l.maxd, out);
}

// The same three functions for a wave-uniform primitive (world_trace_wave): the parameters come in as scalars, and a box whose
// to_local rotation block is the identity (floors, walls, enclosing emitters: translate-only transforms) divides by the world
// ray's own direction components, for which the caller already holds 1.0 / d (identity_rcp: the same quotients, not recomputed).
__device__ __forceinline__ void box_slab_rcp(int axis, double o, double d, double rcp, double lo, double hi, double &near_t, double &far_t,
                                             int &nf, int &ff, int &na, int &fa) {
    double tmin, tmax;
    int fmin, fmax;
    const double inf = INFINITY;
    if (d != 0.0) {
        if (d > 0) { tmin = (lo - o) * rcp; tmax = (hi - o) * rcp; fmin = LOWER_FACE; fmax = UPPER_FACE; }
        else       { tmin = (hi - o) * rcp; tmax = (lo - o) * rcp; fmin = UPPER_FACE; fmax = LOWER_FACE; }
    } else {
        if (o < lo)      { tmin = -inf; tmax = -inf; }
        else if (o > hi) { tmin = inf;  tmax = inf; }
        else             { tmin = -inf; tmax = inf; }
        fmin = NO_FACE; fmax = NO_FACE;
    }
    if (tmin > near_t) { near_t = tmin; nf = fmin; na = axis; }
    if (tmax < far_t)  { far_t = tmax;  ff = fmax; fa = axis; }
}

__device__ __forceinline__ void box_roots_uniform(const double (&prm)[6], const Ray &l, bool identity_rcp, double rx, double ry, double rz, Roots &out) {
    double near_t = -INFINITY, far_t = INFINITY;
    int nf = NO_FACE, ff = NO_FACE, na = -1, fa = -1;
    if (!identity_rcp) { rx = 1.0 / l.dx; ry = 1.0 / l.dy; rz = 1.0 / l.dz; }          // (wave-uniform branch)
    box_slab_rcp(0, l.ox, l.dx, rx, prm[0], prm[3], near_t, far_t, nf, ff, na, fa);
    box_slab_rcp(1, l.oy, l.dy, ry, prm[1], prm[4], near_t, far_t, nf, ff, na, fa);
    box_slab_rcp(2, l.oz, l.dz, rz, prm[2], prm[5], near_t, far_t, nf, ff, na, fa);
    pick_roots(near_t, far_t, nf, na, ff, fa, l.maxd, out);
}

__device__ __forceinline__ void sphere_roots_uniform(double radius, const Ray &l, Roots &out) {
    out.n = 0;
    const double a = l.dx * l.dx + l.dy * l.dy + l.dz * l.dz;
    const double b = 2 * (l.dx * l.ox + l.dy * l.oy + l.dz * l.oz);
    const double c = l.ox * l.ox + l.oy * l.oy + l.oz * l.oz - radius * radius;
    double t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return;
    if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
    pick_roots(t0, t1, 0, 0, 0, 0, l.maxd, out);
}

__device__ void cylinder_roots(const rsx_primitive &p, const Ray &l, Roots &out) {
    out.n = 0;
    const double radius = p.params[0], height = p.params[1];
    double near_t, far_t, t0, t1;
    int nf = NO_FACE, ff = NO_FACE, nt, ft, f0, f1;
    if (l.dx == 0 && l.dy == 0) {
        if ((l.ox * l.ox + l.oy * l.oy) <= (radius * radius)) { near_t = -INFINITY; far_t = INFINITY; nt = -1; ft = -1; }
        else return;
    } else {
        const double a = l.dx * l.dx + l.dy * l.dy;
        const double b = 2.0 * (l.dx * l.ox + l.dy * l.oy);
        const double c = l.ox * l.ox + l.oy * l.oy - radius * radius;
        if (!solve_quadratic(a, b, c, t0, t1)) return;
        if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
        near_t = t0; far_t = t1; nt = T_CYLINDER; ft = T_CYLINDER;
    }
    if (l.dz != 0.0) {
        const double temp = 1.0 / l.dz;
        if (l.dz > 0) { t0 = -l.oz * temp; t1 = (height - l.oz) * temp; f0 = LOWER_FACE; f1 = UPPER_FACE; }
        else          { t0 = (height - l.oz) * temp; t1 = -l.oz * temp; f0 = UPPER_FACE; f1 = LOWER_FACE; }
        if (t0 > near_t) { near_t = t0; nf = f0; nt = T_SLAB; }
        if (t1 < far_t)  { far_t = t1;  ff = f1; ft = T_SLAB; }
    }
    pick_roots(near_t, far_t, nf, nt, ff, ft, l.maxd, out);
}

// ---------------------------------------------------------------------------------------------------
// intersection records (Intersection / MeshIntersection) rebuilt from a Hit
//   sphere.pyx:170-200, box.pyx:296-342, cylinder.pyx:287-354, mesh.pyx:718-800
// geom = hit_point, inside_point, outside_point, normal (primitive-local space)
// ---------------------------------------------------------------------------------------------------
#define PRIM_EPS 1e-9
#define MESH_EPS 1e-6

__device__ __forceinline__ void normalise3(double &x, double &y, double &z) {
    double t = x * x + y * y + z * z;
    t = 1.0 / sqrt(t);
    x *= t; y *= t; z *= t;
}

__device__ __forceinline__ double box_interior_offset(double hit, double lo, double hi) {
    if (fabs(hit - lo) < PRIM_EPS) return PRIM_EPS;
    if (fabs(hit - hi) < PRIM_EPS) return -PRIM_EPS;
    return 0.0;
}

struct Geom {
    double hit[3], inside[3], outside[3], normal[3];
    bool exiting;
};

__device__ void analytic_geom(const rsx_primitive &p, const Ray &l, double t, int a0, int a1, Geom &g) {
    g.hit[0] = l.ox + t * l.dx; g.hit[1] = l.oy + t * l.dy; g.hit[2] = l.oz + t * l.dz;
    if (p.type == RSX_PRIM_SPHERE) {
        g.normal[0] = g.hit[0]; g.normal[1] = g.hit[1]; g.normal[2] = g.hit[2];
        normalise3(g.normal[0], g.normal[1], g.normal[2]);
        for (int k = 0; k < 3; ++k) {
            const double delta = PRIM_EPS * g.normal[k];
            g.inside[k] = g.hit[k] - delta; g.outside[k] = g.hit[k] + delta;
        }
    } else if (p.type == RSX_PRIM_BOX) {
        g.normal[0] = 0; g.normal[1] = 0; g.normal[2] = 0;
        const double s = a0 == LOWER_FACE ? -1.0 : 1.0;
        if (a1 == 0) g.normal[0] = s; else if (a1 == 1) g.normal[1] = s; else if (a1 == 2) g.normal[2] = s;
        for (int k = 0; k < 3; ++k) {
            g.inside[k] = g.hit[k] + box_interior_offset(g.hit[k], p.params[k], p.params[3 + k]);
            g.outside[k] = g.hit[k] + PRIM_EPS * g.normal[k];
        }
    } else {  // cylinder
        const double radius = p.params[0], height = p.params[1];
        double off[3];
        if (a1 == T_CYLINDER) {
            g.normal[0] = g.hit[0]; g.normal[1] = g.hit[1]; g.normal[2] = 0;
            normalise3(g.normal[0], g.normal[1], g.normal[2]);
            off[0] = -PRIM_EPS * g.normal[0]; off[1] = -PRIM_EPS * g.normal[1];
        } else {
            g.normal[0] = 0; g.normal[1] = 0; g.normal[2] = a0 == LOWER_FACE ? -1.0 : 1.0;
            off[0] = 0; off[1] = 0;
            if (g.hit[0] != 0.0 && g.hit[1] != 0.0) {
                double length = sqrt(g.hit[0] * g.hit[0] + g.hit[1] * g.hit[1]);
                if ((length - radius) < PRIM_EPS) {
                    length = 1.0 / length;
                    off[0] = -PRIM_EPS * length * g.hit[0]; off[1] = -PRIM_EPS * length * g.hit[1];
                }
            }
        }
        if (fabs(g.hit[2]) < PRIM_EPS) off[2] = PRIM_EPS;
        else if (fabs(g.hit[2] - height) < PRIM_EPS) off[2] = -PRIM_EPS;
        else off[2] = 0;
        for (int k = 0; k < 3; ++k) { g.inside[k] = g.hit[k] + off[k]; g.outside[k] = g.hit[k] + PRIM_EPS * g.normal[k]; }
    }
    g.exiting = (l.dx * g.normal[0] + l.dy * g.normal[1] + l.dz * g.normal[2]) >= 0.0;
}

// the `exiting` flag analytic_geom would compute, without the points it does not need (same normal, same dot product)
__device__ __forceinline__ bool analytic_exiting(int32_t type, const Ray &l, double t, int a0, int a1) {
    double nx = 0, ny = 0, nz = 0;
    if (type == RSX_PRIM_SPHERE) {
        nx = l.ox + t * l.dx; ny = l.oy + t * l.dy; nz = l.oz + t * l.dz;
        normalise3(nx, ny, nz);
    } else if (type == RSX_PRIM_BOX) {
        const double s = a0 == LOWER_FACE ? -1.0 : 1.0;
        if (a1 == 0) nx = s; else if (a1 == 1) ny = s; else if (a1 == 2) nz = s;
    } else if (a1 == T_CYLINDER) {
        nx = l.ox + t * l.dx; ny = l.oy + t * l.dy; nz = 0;
        normalise3(nx, ny, nz);
    } else nz = a0 == LOWER_FACE ? -1.0 : 1.0;
    return (l.dx * nx + l.dy * ny + l.dz * nz) >= 0.0;
}
__device__ __forceinline__ bool analytic_exiting(const rsx_primitive &p, const Ray &l, double t, int a0, int a1) { return analytic_exiting(p.type, l, t, a0, a1); }

// MeshData.calc_intersection / _intersection_normal. `t` is the LOCAL distance from l's origin.
__device__ void mesh_geom(const DMesh &m, const Ray &l, double t, int32_t tri, float u, float v, float w, Geom &g) {
    const float4 q2 = m.tris[3 * (size_t)tri + 2];
    const double fx = (double)q2.y, fy = (double)q2.z, fz = (double)q2.w;
    g.hit[0] = l.ox + l.dx * t; g.hit[1] = l.oy + l.dy * t; g.hit[2] = l.oz + l.dz * t;
    g.inside[0] = g.hit[0] - fx * MESH_EPS; g.inside[1] = g.hit[1] - fy * MESH_EPS; g.inside[2] = g.hit[2] - fz * MESH_EPS;
    g.outside[0] = g.hit[0] + fx * MESH_EPS; g.outside[1] = g.hit[1] + fy * MESH_EPS; g.outside[2] = g.hit[2] + fz * MESH_EPS;
    if (m.smoothing && m.vnormals) {
        const int32_t n1 = m.nidx[3 * (size_t)tri], n2 = m.nidx[3 * (size_t)tri + 1], n3 = m.nidx[3 * (size_t)tri + 2];
        for (int k = 0; k < 3; ++k) {   // f32 arithmetic, then widened (mesh.pyx:783-787)
            const float nk = u * m.vnormals[3 * (size_t)n1 + k] + v * m.vnormals[3 * (size_t)n2 + k] + w * m.vnormals[3 * (size_t)n3 + k];
            g.normal[k] = (double)nk;
        }
    } else {
        g.normal[0] = fx; g.normal[1] = fy; g.normal[2] = fz;
    }
    normalise3(g.normal[0], g.normal[1], g.normal[2]);
    g.exiting = (l.dx * fx + l.dy * fy + l.dz * fz) > 0.0;
}

