// rsx_hostwalk.cpp — host side of librsx: World.hit / World.contains for ONE ray or point at a time.
//
// SURVEY.md 8(b), "Who calls it": `World.hit (n = 1 -> CPU lib)`. A caller that probes the scene ray by ray (World.hit(ray), Ray.trace of a
// single ray, LoggingRay, the reference's demos/core/ray_intersection_hitpoints.py) pays a kernel launch and two PCIe round trips per ray
// on the device path — 69 us against the reference's ~1.2 us (world.pyx:125-146 from Python). This file answers those calls on the
// host: an explicit-stack walk over the SAME flattened arrays the device scene is built from (rsx_scene_desc), operation for operation
// what the kernels do (and therefore what the reference does: every function cites it). It is product code for the single-ray API
// only: observe(), the batch queries and bench.py never come here, nothing under oracle/ is used. CSG solids are answered by the
// reference's own stream merge (csg.pyx:132-234): hit() / next_intersection() of the operands, each with the cached state the reference
// keeps on its primitive objects (one record per primitive, per calling thread).
//
// Compiled with -ffp-contract=off like the device side; IEEE `/` and sqrt; the mesh test keeps the reference's mixed precision.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rsx.h"
#include "rsx_internal.h"

namespace {

struct HRay { double ox, oy, oz, dx, dy, dz, maxd; };

struct HostMesh {
    std::vector<float> vertices, vnormals, fnormals;
    std::vector<int32_t> triangles;
    int32_t n_triangles = 0, stride = 3, smoothing = 0, closed = 0;
    std::vector<rsx_kdnode> nodes;
    std::vector<int32_t> items;
    double lower[3], upper[3];
    int32_t depth = 0;
};

inline bool is_csg_type(int32_t t) { return t == RSX_PRIM_UNION || t == RSX_PRIM_INTERSECT || t == RSX_PRIM_SUBTRACT; }

}  // namespace

struct rsx_host_scene {
    std::vector<rsx_primitive> prims;
    int32_t n_world = 0;
    std::vector<HostMesh> meshes;
    std::vector<rsx_kdnode> wnodes;
    std::vector<int32_t> witems;
    double wlower[3], wupper[3];
    int32_t wdepth = 0;
    bool has_csg = false;
};

namespace {

inline double sel3(int i, double x, double y, double z) { return i == 0 ? x : (i == 1 ? y : z); }
inline float sel3f(int i, float x, float y, float z) { return i == 0 ? x : (i == 1 ? y : z); }

// Point3D.transform / Vector3D.transform — core/math/point.pyx:253-284, vector.pyx:339-369
inline void xform_point(const double *m, double x, double y, double z, double &ox, double &oy, double &oz) {
    double w = m[12] * x + m[13] * y + m[14] * z + m[15];
    w = 1.0 / w;
    ox = (m[0] * x + m[1] * y + m[2] * z + m[3]) * w;
    oy = (m[4] * x + m[5] * y + m[6] * z + m[7]) * w;
    oz = (m[8] * x + m[9] * y + m[10] * z + m[11]) * w;
}
inline void xform_vector(const double *m, double x, double y, double z, double &ox, double &oy, double &oz) {
    ox = m[0] * x + m[1] * y + m[2] * z;
    oy = m[4] * x + m[5] * y + m[6] * z;
    oz = m[8] * x + m[9] * y + m[10] * z;
}
inline HRay to_local(const rsx_primitive &p, const HRay &r) {
    HRay l;
    xform_point(p.to_local, r.ox, r.oy, r.oz, l.ox, l.oy, l.oz);
    xform_vector(p.to_local, r.dx, r.dy, r.dz, l.dx, l.dy, l.dz);
    l.maxd = r.maxd;
    return l;
}

// BoundingBox3D._slab / intersect — core/boundingbox.pyx:180-245
inline void slab(double o, double d, double lo, double hi, double &front, double &back) {
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
inline bool aabb(const double *lo, const double *hi, const HRay &r, double &front, double &back) {
    front = -INFINITY;
    back = INFINITY;
    slab(r.ox, r.dx, lo[0], hi[0], front, back);
    slab(r.oy, r.dy, lo[1], hi[1], front, back);
    slab(r.oz, r.dz, lo[2], hi[2], front, back);
    if (front > back) return false;
    if (front < 0.0 && back < 0.0) return false;
    return true;
}
inline bool aabb_contains(const double *lo, const double *hi, double x, double y, double z) {
    if (x < lo[0] || x > hi[0]) return false;
    if (y < lo[1] || y > hi[1]) return false;
    if (z < lo[2] || z > hi[2]) return false;
    return true;
}

// KDTree3DCore._trace / _trace_branch — core/math/spatial/kdtree3d.pyx:589-700, as a loop over an explicit stack of (far node, its
// tmax): the far range's tmin is the tmax of the leaf that was just exhausted. `leaf(node, tmin, tmax)` returns true to stop the walk.
struct KdStack {
    std::vector<int32_t> id;
    std::vector<double> t;
    explicit KdStack(int depth) : id((size_t)depth + 2), t((size_t)depth + 2) {}
};
template <typename Leaf>
inline bool kd_walk(const rsx_kdnode *nodes, const double *lower, const double *upper, const HRay &r, KdStack &st, Leaf leaf) {
    double tmin, tmax;
    if (!aabb(lower, upper, r, tmin, tmax)) return false;               // kdtree3d.pyx:589-607 (max_distance is not looked at here)
    int32_t node = 0;
    size_t sp = 0;
    for (;;) {
        while (nodes[node].type >= 0) {
            const rsx_kdnode &nd = nodes[node];
            const int axis = nd.type & 3;
            const double o = sel3(axis, r.ox, r.oy, r.oz), d = sel3(axis, r.dx, r.dy, r.dz), split = nd.u.split;
            const int32_t lower_id = node + 1, upper_id = nd.count;
            if (d == 0) { node = o < split ? lower_id : upper_id; continue; }
            const double plane = (split - o) / d;
            const bool below = o < split || (o == split && d < 0);
            const int32_t near_id = below ? lower_id : upper_id, far_id = below ? upper_id : lower_id;
            if (plane > tmax || plane <= 0) node = near_id;
            else if (plane < tmin) node = far_id;
            else {
                if (sp >= st.id.size()) { st.id.resize(2 * st.id.size()); st.t.resize(2 * st.t.size()); }
                st.id[sp] = far_id; st.t[sp] = tmax; ++sp;
                tmax = plane;
                node = near_id;
            }
        }
        if (leaf(nodes[node], tmin, tmax)) return true;
        if (sp == 0) return false;
        --sp;
        tmin = tmax;
        node = st.id[sp]; tmax = st.t[sp];
    }
}

// ---- analytic primitives: sphere.pyx:115-159, box.pyx:157-294, cylinder.pyx:148-276, utility.pyx:376-419 ----
enum { NO_FACE = -1, LOWER_FACE = 0, UPPER_FACE = 1, T_CYLINDER = 0, T_SLAB = 1 };
struct Roots { int n; double t[2]; int32_t a0[2], a1[2]; };

inline bool solve_quadratic(double a, double b, double c, double &t0, double &t1) {
    const double d = b * b - 4 * a * c;
    if (d < 0) return false;
    double q;
    if (b < 0) q = -0.5 * (b - std::sqrt(d)); else q = -0.5 * (b + std::sqrt(d));
    t0 = q / a;
    t1 = c / q;
    return true;
}
inline void pick_roots(double near_t, double far_t, int nf, int na, int ff, int fa, double maxd, Roots &out) {
    out.n = 0;
    if (near_t > far_t) return;
    if (near_t > maxd || far_t < 0.0) return;
    if (near_t >= 0.0) {
        out.t[0] = near_t; out.a0[0] = nf; out.a1[0] = na; out.n = 1;
        if (far_t <= maxd) { out.t[1] = far_t; out.a0[1] = ff; out.a1[1] = fa; out.n = 2; }
    } else if (far_t <= maxd) {
        out.t[0] = far_t; out.a0[0] = ff; out.a1[0] = fa; out.n = 1;
    }
}
inline void sphere_roots(const rsx_primitive &p, const HRay &l, Roots &out) {
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
inline void box_slab(int axis, double o, double d, double lo, double hi, double &near_t, double &far_t, int &nf, int &ff, int &na, int &fa) {
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
inline void box_roots(const rsx_primitive &p, const HRay &l, Roots &out) {
    double near_t = -INFINITY, far_t = INFINITY;
    int nf = NO_FACE, ff = NO_FACE, na = -1, fa = -1;
    box_slab(0, l.ox, l.dx, p.params[0], p.params[3], near_t, far_t, nf, ff, na, fa);
    box_slab(1, l.oy, l.dy, p.params[1], p.params[4], near_t, far_t, nf, ff, na, fa);
    box_slab(2, l.oz, l.dz, p.params[2], p.params[5], near_t, far_t, nf, ff, na, fa);
    pick_roots(near_t, far_t, nf, na, ff, fa, l.maxd, out);
}
inline void cylinder_roots(const rsx_primitive &p, const HRay &l, Roots &out) {
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

// ---- intersection records: sphere.pyx:170-200, box.pyx:296-342, cylinder.pyx:287-354, mesh.pyx:718-800 ----
const double PRIM_EPS = 1e-9, MESH_EPS = 1e-6;
struct Geom { double hit[3], inside[3], outside[3], normal[3]; bool exiting; };

inline void normalise3(double &x, double &y, double &z) {
    double t = x * x + y * y + z * z;
    t = 1.0 / std::sqrt(t);
    x *= t; y *= t; z *= t;
}
inline double box_interior_offset(double hit, double lo, double hi) {
    if (std::fabs(hit - lo) < PRIM_EPS) return PRIM_EPS;
    if (std::fabs(hit - hi) < PRIM_EPS) return -PRIM_EPS;
    return 0.0;
}
void analytic_geom(const rsx_primitive &p, const HRay &l, double t, int a0, int a1, Geom &g) {
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
                double length = std::sqrt(g.hit[0] * g.hit[0] + g.hit[1] * g.hit[1]);
                if ((length - radius) < PRIM_EPS) {
                    length = 1.0 / length;
                    off[0] = -PRIM_EPS * length * g.hit[0]; off[1] = -PRIM_EPS * length * g.hit[1];
                }
            }
        }
        if (std::fabs(g.hit[2]) < PRIM_EPS) off[2] = PRIM_EPS;
        else if (std::fabs(g.hit[2] - height) < PRIM_EPS) off[2] = -PRIM_EPS;
        else off[2] = 0;
        for (int k = 0; k < 3; ++k) { g.inside[k] = g.hit[k] + off[k]; g.outside[k] = g.hit[k] + PRIM_EPS * g.normal[k]; }
    }
    g.exiting = (l.dx * g.normal[0] + l.dy * g.normal[1] + l.dz * g.normal[2]) >= 0.0;
}

// ---- mesh: raysect/primitive/mesh/mesh.pyx:506-713 (MeshData.trace / _trace_leaf / _hit_triangle, watertight test) ----
struct MeshHit { float u, v, w, t; int32_t tri; };
struct TriRay { double ox, oy, oz, maxd; float sx, sy, sz; int ix, iy, iz; };

inline TriRay tri_ray(const HRay &r) {                                     // _calc_rayspace_transform, mesh.pyx:566-610
    TriRay q;
    int ix, iy, iz;
    const double ax = std::fabs(r.dx), ay = std::fabs(r.dy), az = std::fabs(r.dz);
    if (ax > ay && ax > az) { ix = 1; iy = 2; iz = 0; }
    else if (ay > ax && ay > az) { ix = 2; iy = 0; iz = 1; }
    else { ix = 0; iy = 1; iz = 2; }
    const float rdz = (float)sel3(iz, r.dx, r.dy, r.dz);
    if (rdz < 0.0f) { const int tmp = ix; ix = iy; iy = tmp; }
    q.sz = (float)(1.0 / (double)rdz);
    q.sx = (float)(sel3(ix, r.dx, r.dy, r.dz) * (double)q.sz);
    q.sy = (float)(sel3(iy, r.dx, r.dy, r.dz) * (double)q.sz);
    q.ix = ix; q.iy = iy; q.iz = iz;
    q.ox = r.ox; q.oy = r.oy; q.oz = r.oz; q.maxd = r.maxd;
    return q;
}
inline bool tri_test(const TriRay &q, const float *v1, const float *v2, const float *v3, float &ht, float &hu, float &hv, float &hw) {   // _hit_triangle, mesh.pyx:616-713
    float p1[3], p2[3], p3[3];                              // vertices relative to the ray origin: f32 vertex minus f64 origin, rounded to f32
    p1[0] = (float)((double)v1[0] - q.ox); p1[1] = (float)((double)v1[1] - q.oy); p1[2] = (float)((double)v1[2] - q.oz);
    p2[0] = (float)((double)v2[0] - q.ox); p2[1] = (float)((double)v2[1] - q.oy); p2[2] = (float)((double)v2[2] - q.oz);
    p3[0] = (float)((double)v3[0] - q.ox); p3[1] = (float)((double)v3[1] - q.oy); p3[2] = (float)((double)v3[2] - q.oz);
    const float a1 = sel3f(q.ix, p1[0], p1[1], p1[2]), b1 = sel3f(q.iy, p1[0], p1[1], p1[2]), c1 = sel3f(q.iz, p1[0], p1[1], p1[2]);
    const float a2 = sel3f(q.ix, p2[0], p2[1], p2[2]), b2 = sel3f(q.iy, p2[0], p2[1], p2[2]), c2 = sel3f(q.iz, p2[0], p2[1], p2[2]);
    const float a3 = sel3f(q.ix, p3[0], p3[1], p3[2]), b3 = sel3f(q.iy, p3[0], p3[1], p3[2]), c3 = sel3f(q.iz, p3[0], p3[1], p3[2]);
    const float x1 = a1 - q.sx * c1, x2 = a2 - q.sx * c2, x3 = a3 - q.sx * c3;
    const float y1 = b1 - q.sy * c1, y2 = b2 - q.sy * c2, y3 = b3 - q.sy * c3;
    float u = x3 * y2 - y3 * x2, v = x1 * y3 - y1 * x3, w = x2 * y1 - y2 * x1;
    if (u == 0.0f || v == 0.0f || w == 0.0f) {              // on an edge: once more in f64 (mesh.pyx:668-680)
        u = (float)((double)x3 * (double)y2 - (double)y3 * (double)x2);
        v = (float)((double)x1 * (double)y3 - (double)y1 * (double)x3);
        w = (float)((double)x2 * (double)y1 - (double)y2 * (double)x1);
    }
    if ((u < 0.0f || v < 0.0f || w < 0.0f) && (u > 0.0f || v > 0.0f || w > 0.0f)) return false;
    const float det = u + v + w;
    if (det == 0.0f) return false;
    const float z1 = q.sz * c1, z2 = q.sz * c2, z3 = q.sz * c3;
    const float t = u * z1 + v * z2 + w * z3;
    if (det > 0.0f) { if (t < 0.0f || (double)t > q.maxd * (double)det) return false; }
    else            { if (t > 0.0f || (double)t < q.maxd * (double)det) return false; }
    const float rdet = (float)(1.0 / (double)det);
    ht = t * rdet; hu = u * rdet; hv = v * rdet; hw = w * rdet;
    return true;
}
bool mesh_trace(const HostMesh &m, const HRay &r, MeshHit &out) {
    const TriRay q = tri_ray(r);
    KdStack st(m.depth);
    bool found = false;
    kd_walk(m.nodes.data(), m.lower, m.upper, r, st, [&](const rsx_kdnode &nd, double, double tmax) {
        // _trace_leaf, mesh.pyx:520-563: items in leaf order, strict `<` keeps the first of equal distances
        double distance = r.maxd < tmax ? r.maxd : tmax;
        int32_t closest = -1;
        float bu = 0, bv = 0, bw = 0;
        for (int32_t k = 0; k < nd.count; ++k) {
            const int32_t tri = m.items[(size_t)nd.u.leaf.first_item + k];
            const int32_t *ix = m.triangles.data() + (size_t)tri * m.stride;
            float ht, hu, hv, hw;
            if (tri_test(q, &m.vertices[3 * (size_t)ix[0]], &m.vertices[3 * (size_t)ix[1]], &m.vertices[3 * (size_t)ix[2]], ht, hu, hv, hw) && (double)ht < distance) {
                distance = (double)ht; closest = tri; bu = hu; bv = hv; bw = hw;
            }
        }
        if (closest < 0) return false;
        out.u = bu; out.v = bv; out.w = bw; out.t = (float)distance; out.tri = closest;
        found = true;
        return true;
    });
    return found;
}
void mesh_geom(const HostMesh &m, const HRay &l, double t, int32_t tri, float u, float v, float w, Geom &g) {   // mesh.pyx:718-800
    const double fx = (double)m.fnormals[3 * (size_t)tri], fy = (double)m.fnormals[3 * (size_t)tri + 1], fz = (double)m.fnormals[3 * (size_t)tri + 2];
    g.hit[0] = l.ox + l.dx * t; g.hit[1] = l.oy + l.dy * t; g.hit[2] = l.oz + l.dz * t;
    g.inside[0] = g.hit[0] - fx * MESH_EPS; g.inside[1] = g.hit[1] - fy * MESH_EPS; g.inside[2] = g.hit[2] - fz * MESH_EPS;
    g.outside[0] = g.hit[0] + fx * MESH_EPS; g.outside[1] = g.hit[1] + fy * MESH_EPS; g.outside[2] = g.hit[2] + fz * MESH_EPS;
    if (m.smoothing && !m.vnormals.empty() && m.stride == 6) {
        const int32_t *ix = m.triangles.data() + (size_t)tri * m.stride;
        for (int k = 0; k < 3; ++k) {                          // f32 arithmetic, then widened (mesh.pyx:783-787)
            const float nk = u * m.vnormals[3 * (size_t)ix[3] + k] + v * m.vnormals[3 * (size_t)ix[4] + k] + w * m.vnormals[3 * (size_t)ix[5] + k];
            g.normal[k] = (double)nk;
        }
    } else { g.normal[0] = fx; g.normal[1] = fy; g.normal[2] = fz; }
    normalise3(g.normal[0], g.normal[1], g.normal[2]);
    g.exiting = (l.dx * fx + l.dy * fy + l.dz * fz) > 0.0;
}

// ---- CSG: raysect/primitive/csg.pyx:132-234 (hit / next_intersection / _identify_intersection / _closest_intersection), the operators'
// _valid_intersection tables (:326-348 Union, :424-446 Intersect, :526-548 Subtract) and Subtract._modify_intersection (:550-568) ----
// An Intersection as an operand hands it up: the record in the operand's own space (points, normal, exiting), lifted one level at a time.
struct Isect {
    bool valid = false;
    double t = 0;
    int32_t prim = -1;         // the primitive whose space the record is in (a CSG node once lifted)
    int32_t leaf = -1;         // the operand leaf that produced the root
    int32_t tri = -1;
    float u = 0, v = 0, w = 0;
    Geom g;
};
// What the reference keeps on the primitive objects between hit() and next_intersection(): the cached second root of a sphere / box /
// cylinder (sphere.pyx:150-157 ...), a mesh's continuation ray (mesh.pyx:1240-1275), a CSG node's stream heads (csg.pyx:150-155) and
// BoundPrimitive's "was the primitive reached" (boundprimitive.pyx:38-60).
struct PState {
    bool further = false; double next_t = 0; HRay c_local{}; int32_t c_a0 = 0, c_a1 = 0;
    bool seek_next = false; HRay next_local{}; double ray_distance = 0;
    HRay cache_ray{}; Isect cache_a, cache_b; bool cache_last_is_a = false, cache_invalid = true;
    bool tested = false;
};
struct CsgEval { const rsx_host_scene &sc; std::vector<PState> &st; };

void prim_first(CsgEval &e, int32_t idx, const HRay &ray, Isect &out);
void prim_next(CsgEval &e, int32_t idx, Isect &out);
bool bound_contains(const rsx_host_scene &sc, int32_t idx, double px, double py, double pz);

inline void analytic_record(const rsx_primitive &p, int32_t idx, const HRay &l, double t, int32_t a0, int32_t a1, Isect &out) {
    out.valid = true; out.t = t; out.prim = idx; out.leaf = idx; out.tri = -1; out.u = out.v = out.w = 0.0f;
    analytic_geom(p, l, t, a0, a1, out.g);
}
inline void mesh_record(CsgEval &e, int32_t idx, const HostMesh &m, const HRay &local, const MeshHit &mh, Isect &out) {   // Mesh._process_intersection, mesh.pyx:1240-1275
    PState &s = e.st[(size_t)idx];
    out.valid = true; out.prim = idx; out.leaf = idx; out.tri = mh.tri; out.u = mh.u; out.v = mh.v; out.w = mh.w;
    out.t = (double)mh.t;
    mesh_geom(m, local, out.t, mh.tri, mh.u, mh.v, mh.w, out.g);
    s.seek_next = true;
    s.next_local.ox = out.g.hit[0] + local.dx * MESH_EPS; s.next_local.oy = out.g.hit[1] + local.dy * MESH_EPS; s.next_local.oz = out.g.hit[2] + local.dz * MESH_EPS;
    s.next_local.dx = local.dx; s.next_local.dy = local.dy; s.next_local.dz = local.dz;
    s.next_local.maxd = local.maxd - out.t - MESH_EPS;
    out.t += s.ray_distance;
    s.ray_distance = out.t + MESH_EPS;
}
// BoundPrimitive.hit / next_intersection (boundprimitive.pyx:42-60)
inline void bound_first(CsgEval &e, int32_t idx, const HRay &ray, Isect &out) {
    const rsx_primitive &p = e.sc.prims[(size_t)idx];
    double f, b;
    out.valid = false;
    if (aabb(p.box_lower, p.box_upper, ray, f, b)) { e.st[(size_t)idx].tested = true; prim_first(e, idx, ray, out); return; }
    e.st[(size_t)idx].tested = false;
}
inline void bound_next(CsgEval &e, int32_t idx, Isect &out) {
    out.valid = false;
    if (e.st[(size_t)idx].tested) prim_next(e, idx, out);
}
inline bool csg_accepts(int32_t type, const Isect &a, const Isect &b, bool closest_is_a) {
    const bool inside_a = a.valid && a.g.exiting, inside_b = b.valid && b.g.exiting;
    if (type == RSX_PRIM_UNION) return (!inside_a && !inside_b) || (inside_a && !inside_b && closest_is_a) || (!inside_a && inside_b && !closest_is_a);
    if (type == RSX_PRIM_INTERSECT) return (inside_a && inside_b) || (inside_a && !inside_b && !closest_is_a) || (!inside_a && inside_b && closest_is_a);
    return (!inside_a && !inside_b && closest_is_a) || (inside_a && !inside_b) || (inside_a && inside_b && !closest_is_a);
}
inline int csg_nearer(const Isect &a, const Isect &b) {       // _closest_intersection, csg.pyx:226-234: 1 = a, 0 = b (b wins a tie), -1 = neither
    if (!a.valid) return b.valid ? 0 : -1;
    if (!b.valid || a.t < b.t) return 1;
    return 0;
}
void csg_merge(CsgEval &e, int32_t idx, const HRay &ray, Isect &a, Isect &b, Isect &out) {     // _identify_intersection, csg.pyx:181-224
    const rsx_primitive &p = e.sc.prims[(size_t)idx];
    PState &s = e.st[(size_t)idx];
    out.valid = false;
    for (int closest = csg_nearer(a, b); closest >= 0; closest = csg_nearer(a, b)) {
        const Isect &c = closest ? a : b;
        if (csg_accepts(p.type, a, b, closest != 0)) {
            if (!(c.t <= ray.maxd)) return;
            s.cache_ray = ray; s.cache_a = a; s.cache_b = b; s.cache_last_is_a = closest != 0; s.cache_invalid = false;
            Isect r = c;
            if (p.type == RSX_PRIM_SUBTRACT && !closest) {                    // the surface of the subtracted solid, seen from the other side
                for (int k = 0; k < 3; ++k) { const double tmp = r.g.inside[k]; r.g.inside[k] = r.g.outside[k]; r.g.outside[k] = tmp; r.g.normal[k] = -r.g.normal[k]; }
                r.g.exiting = !r.g.exiting;
            }
            const rsx_primitive &src = e.sc.prims[(size_t)r.prim];            // from the operand's space into this node's (csg.pyx:198-208)
            double x, y, z;
            xform_point(src.to_root, r.g.hit[0], r.g.hit[1], r.g.hit[2], x, y, z); r.g.hit[0] = x; r.g.hit[1] = y; r.g.hit[2] = z;
            xform_point(src.to_root, r.g.inside[0], r.g.inside[1], r.g.inside[2], x, y, z); r.g.inside[0] = x; r.g.inside[1] = y; r.g.inside[2] = z;
            xform_point(src.to_root, r.g.outside[0], r.g.outside[1], r.g.outside[2], x, y, z); r.g.outside[0] = x; r.g.outside[1] = y; r.g.outside[2] = z;
            const double *mi = src.to_local;                                  // Normal3D.transform_with_inverse: the inverse's transpose
            x = mi[0] * r.g.normal[0] + mi[4] * r.g.normal[1] + mi[8] * r.g.normal[2];
            y = mi[1] * r.g.normal[0] + mi[5] * r.g.normal[1] + mi[9] * r.g.normal[2];
            z = mi[2] * r.g.normal[0] + mi[6] * r.g.normal[1] + mi[10] * r.g.normal[2];
            r.g.normal[0] = x; r.g.normal[1] = y; r.g.normal[2] = z;
            r.prim = idx;
            out = r;
            return;
        }
        if (closest) bound_next(e, p.child_a, a); else bound_next(e, p.child_b, b);
    }
}
void prim_first(CsgEval &e, int32_t idx, const HRay &ray, Isect &out) {
    const rsx_primitive &p = e.sc.prims[(size_t)idx];
    PState &s = e.st[(size_t)idx];
    out.valid = false;
    if (is_csg_type(p.type)) {                                                // CSGPrimitive.hit, csg.pyx:132-155
        s.cache_invalid = true;
        HRay local = to_local(p, ray);
        local.maxd = INFINITY;
        Isect a, b;
        bound_first(e, p.child_a, local, a);
        if (p.type != RSX_PRIM_UNION && !a.valid) return;                     // Intersect / Subtract: nothing without a root of A (:421, :523)
        bound_first(e, p.child_b, local, b);
        csg_merge(e, idx, ray, a, b, out);
        return;
    }
    const HRay l = to_local(p, ray);
    if (p.type == RSX_PRIM_MESH) {                                            // Mesh.hit, mesh.pyx:1178-1211
        s.ray_distance = 0;
        MeshHit mh;
        const HostMesh &m = e.sc.meshes[(size_t)p.mesh];
        if (mesh_trace(m, l, mh)) { mesh_record(e, idx, m, l, mh, out); return; }
        s.seek_next = false;
        return;
    }
    Roots roots;
    roots.n = 0;
    s.further = false;
    if (p.type == RSX_PRIM_SPHERE) sphere_roots(p, l, roots);
    else if (p.type == RSX_PRIM_BOX) box_roots(p, l, roots);
    else if (p.type == RSX_PRIM_CYLINDER) cylinder_roots(p, l, roots);
    if (roots.n == 0) return;
    if (roots.n == 2) { s.further = true; s.next_t = roots.t[1]; s.c_local = l; s.c_a0 = roots.a0[1]; s.c_a1 = roots.a1[1]; }
    analytic_record(p, idx, l, roots.t[0], roots.a0[0], roots.a1[0], out);
}
void prim_next(CsgEval &e, int32_t idx, Isect &out) {
    const rsx_primitive &p = e.sc.prims[(size_t)idx];
    PState &s = e.st[(size_t)idx];
    out.valid = false;
    if (is_csg_type(p.type)) {                                                // CSGPrimitive.next_intersection, csg.pyx:160-179
        if (s.cache_invalid) return;
        Isect a = s.cache_a, b = s.cache_b;
        const HRay ray = s.cache_ray;
        if (s.cache_last_is_a) bound_next(e, p.child_a, a); else bound_next(e, p.child_b, b);
        csg_merge(e, idx, ray, a, b, out);
        return;
    }
    if (p.type == RSX_PRIM_MESH) {                                            // Mesh.next_intersection, mesh.pyx:1213-1238
        if (!s.seek_next) return;
        const HostMesh &m = e.sc.meshes[(size_t)p.mesh];
        const HRay local = s.next_local;
        MeshHit mh;
        if (mesh_trace(m, local, mh)) { mesh_record(e, idx, m, local, mh, out); return; }
        s.seek_next = false;
        return;
    }
    if (!s.further) return;
    s.further = false;
    analytic_record(p, idx, s.c_local, s.next_t, s.c_a0, s.c_a1, out);
}
std::vector<PState> &csg_states(const rsx_host_scene &sc) {   // one set of records per calling thread, as large as the biggest scene it has seen
    thread_local std::vector<PState> st;
    if (st.size() < sc.prims.size()) st.resize(sc.prims.size());
    return st;
}

struct HHit { int32_t prim; double t; int32_t a0, a1; float u, v, w; Isect csg; };

// World.hit — core/scenegraph/world.pyx:125-146, core/acceleration/kdtree.pyx:73-122, boundprimitive.pyx:42-51
bool world_hit(const rsx_host_scene &sc, const HRay &r, HHit &best) {
    best.prim = -1;
    KdStack st(sc.wdepth);
    kd_walk(sc.wnodes.data(), sc.wlower, sc.wupper, r, st, [&](const rsx_kdnode &nd, double, double tmax) {
        double distance = r.maxd < tmax ? r.maxd : tmax;
        for (int32_t k = 0; k < nd.count; ++k) {
            const int32_t idx = sc.witems[(size_t)nd.u.leaf.first_item + k];
            const rsx_primitive &p = sc.prims[(size_t)idx];
            double f, b;
            if (!aabb(p.box_lower, p.box_upper, r, f, b)) continue;         // BoundPrimitive.hit gate
            const HRay l = to_local(p, r);
            HHit cand;
            cand.prim = -1;
            if (is_csg_type(p.type)) {
                CsgEval e{sc, csg_states(sc)};
                e.st[(size_t)idx].tested = true;
                prim_first(e, idx, r, cand.csg);
                if (cand.csg.valid) { cand.prim = idx; cand.t = cand.csg.t; cand.a0 = cand.csg.tri; cand.a1 = 0; cand.u = cand.csg.u; cand.v = cand.csg.v; cand.w = cand.csg.w; }
            } else if (p.type == RSX_PRIM_MESH) {
                MeshHit mh;
                if (mesh_trace(sc.meshes[(size_t)p.mesh], l, mh)) { cand.prim = idx; cand.t = (double)mh.t; cand.a0 = mh.tri; cand.a1 = 0; cand.u = mh.u; cand.v = mh.v; cand.w = mh.w; }
            } else {
                Roots roots;
                roots.n = 0;
                if (p.type == RSX_PRIM_SPHERE) sphere_roots(p, l, roots);
                else if (p.type == RSX_PRIM_BOX) box_roots(p, l, roots);
                else if (p.type == RSX_PRIM_CYLINDER) cylinder_roots(p, l, roots);
                if (roots.n > 0) { cand.prim = idx; cand.t = roots.t[0]; cand.a0 = roots.a0[0]; cand.a1 = roots.a1[0]; cand.u = cand.v = cand.w = 0.0f; }
            }
            if (cand.prim >= 0 && cand.t <= distance) { distance = cand.t; best = cand; }   // `<=`: the later item wins ties (kdtree.pyx:113)
        }
        return best.prim >= 0;
    });
    return best.prim >= 0;
}

// primitive.contains(): sphere.pyx:202-214, box.pyx:344-361, cylinder.pyx:356-372, mesh.pyx:1277-1297 (+802-830)
bool leaf_contains(const rsx_host_scene &sc, const rsx_primitive &p, double px, double py, double pz) {
    double qx, qy, qz;
    xform_point(p.to_local, px, py, pz, qx, qy, qz);
    if (is_csg_type(p.type)) {                                                // Union / Intersect / Subtract.contains, csg.pyx:350-353, 448-451, 570-573
        const bool a = bound_contains(sc, p.child_a, qx, qy, qz);
        if (p.type == RSX_PRIM_UNION) return a || bound_contains(sc, p.child_b, qx, qy, qz);
        if (p.type == RSX_PRIM_INTERSECT) return a && bound_contains(sc, p.child_b, qx, qy, qz);
        return a && !bound_contains(sc, p.child_b, qx, qy, qz);
    }
    if (p.type == RSX_PRIM_SPHERE) return (qx * qx + qy * qy + qz * qz) <= p.params[0] * p.params[0];
    if (p.type == RSX_PRIM_BOX) return aabb_contains(p.params, p.params + 3, qx, qy, qz);
    if (p.type == RSX_PRIM_CYLINDER) return (0.0 <= qz && qz <= p.params[1]) && ((qx * qx + qy * qy) <= (p.params[0] * p.params[0]));
    if (p.type == RSX_PRIM_MESH) {
        const HostMesh &m = sc.meshes[(size_t)p.mesh];
        if (!m.closed) return false;
        HRay zr;
        zr.ox = qx; zr.oy = qy; zr.oz = qz; zr.dx = 0; zr.dy = 0; zr.dz = 1; zr.maxd = INFINITY;
        MeshHit mh;
        if (mesh_trace(m, zr, mh)) return m.fnormals[3 * (size_t)mh.tri + 2] > 0.0f;
    }
    return false;
}

bool bound_contains(const rsx_host_scene &sc, int32_t idx, double px, double py, double pz) {   // BoundPrimitive.contains, boundprimitive.pyx:62-66
    const rsx_primitive &p = sc.prims[(size_t)idx];
    return aabb_contains(p.box_lower, p.box_upper, px, py, pz) && leaf_contains(sc, p, px, py, pz);
}

int kd_depth(const rsx_kdtree &kd) {                          // deepest leaf of the pre-order array (bounds the stack)
    int deepest = 0;
    std::vector<std::pair<int32_t, int>> todo{{0, 0}};
    while (!todo.empty()) {
        const auto [id, depth] = todo.back();
        todo.pop_back();
        if (id < 0 || id >= kd.n_nodes) continue;
        deepest = depth > deepest ? depth : deepest;
        if (kd.nodes[id].type >= 0) { todo.push_back({id + 1, depth + 1}); todo.push_back({kd.nodes[id].count, depth + 1}); }
    }
    return deepest;
}

}  // namespace

extern "C" int rsx_host_scene_create(const rsx_scene_desc *desc, rsx_host_scene **out) {
    if (!desc || !out) return rsx_fail(RSX_EINVAL, "rsx_host_scene_create: null argument");
    if (desc->n_primitives < 0 || desc->n_world < 0 || desc->n_world > desc->n_primitives || desc->n_meshes < 0)
        return rsx_fail(RSX_EINVAL, "rsx_host_scene_create: counts out of range");
    rsx_host_scene *sc = new (std::nothrow) rsx_host_scene();
    if (!sc) return rsx_fail(RSX_ENOMEM, "rsx_host_scene_create: out of memory");
    try {
        sc->prims.assign(desc->primitives, desc->primitives + desc->n_primitives);
        sc->n_world = desc->n_world;
        for (int32_t i = 0; i < desc->n_primitives; ++i) {
            const rsx_primitive &p = sc->prims[(size_t)i];
            if (is_csg_type(p.type)) sc->has_csg = true;
            if (p.type == RSX_PRIM_MESH && (p.mesh < 0 || p.mesh >= desc->n_meshes)) { delete sc; return rsx_fail(RSX_EINVAL, "rsx_host_scene_create: primitive %d names mesh %d of %d", i, p.mesh, desc->n_meshes); }
        }
        sc->meshes.resize((size_t)desc->n_meshes);
        for (int32_t i = 0; i < desc->n_meshes; ++i) {
            const rsx_meshdata &md = desc->meshes[i];
            HostMesh &m = sc->meshes[(size_t)i];
            m.vertices.assign(md.vertices, md.vertices + 3 * (size_t)md.n_vertices);
            m.triangles.assign(md.triangles, md.triangles + (size_t)md.tri_stride * (size_t)md.n_triangles);
            if (md.vertex_normals) m.vnormals.assign(md.vertex_normals, md.vertex_normals + 3 * (size_t)md.n_normals);
            m.fnormals.assign(md.face_normals, md.face_normals + 3 * (size_t)md.n_triangles);
            m.n_triangles = md.n_triangles; m.stride = md.tri_stride; m.smoothing = md.smoothing; m.closed = md.closed;
            m.nodes.assign(md.kd.nodes, md.kd.nodes + md.kd.n_nodes);
            m.items.assign(md.kd.items, md.kd.items + md.kd.n_items);
            std::memcpy(m.lower, md.kd.lower, sizeof(m.lower)); std::memcpy(m.upper, md.kd.upper, sizeof(m.upper));
            m.depth = kd_depth(md.kd);
        }
        sc->wnodes.assign(desc->world_kd.nodes, desc->world_kd.nodes + desc->world_kd.n_nodes);
        sc->witems.assign(desc->world_kd.items, desc->world_kd.items + desc->world_kd.n_items);
        std::memcpy(sc->wlower, desc->world_kd.lower, sizeof(sc->wlower)); std::memcpy(sc->wupper, desc->world_kd.upper, sizeof(sc->wupper));
        sc->wdepth = kd_depth(desc->world_kd);
    } catch (const std::bad_alloc &) { delete sc; return rsx_fail(RSX_ENOMEM, "rsx_host_scene_create: out of memory"); }
    *out = sc;
    return RSX_OK;
}

extern "C" void rsx_host_scene_free(rsx_host_scene *scene) { delete scene; }

extern "C" int rsx_hit_host(const rsx_host_scene *scene, int64_t n, const double *origin, const double *direction, const double *max_distance,
                            int32_t *prim, double *t, uint8_t *exiting, int32_t *tri, float *uvw, double *geom) {
    if (!scene || n < 0 || (n > 0 && (!origin || !direction || !prim))) return rsx_fail(RSX_EINVAL, "rsx_hit_host: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        HRay r;
        r.ox = origin[3 * i]; r.oy = origin[3 * i + 1]; r.oz = origin[3 * i + 2];
        r.dx = direction[3 * i]; r.dy = direction[3 * i + 1]; r.dz = direction[3 * i + 2];
        r.maxd = max_distance ? max_distance[i] : INFINITY;
        HHit h;
        const bool hit = world_hit(*scene, r, h);
        prim[i] = hit ? h.prim : -1;
        const bool csg = hit && is_csg_type(scene->prims[(size_t)h.prim].type);
        const bool mesh = hit && (csg ? h.csg.tri >= 0 : scene->prims[(size_t)h.prim].type == RSX_PRIM_MESH);   // (a CSG node hands its operand's MeshIntersection on)
        // (a mesh reports `t + Mesh._ray_distance`, mesh.pyx:1240-1275 — + 0.0 for a first hit: a root at -0.0 leaves as +0.0, the geometry keeps the root as found)
        if (t) t[i] = hit ? (mesh && !csg ? h.t + 0.0 : h.t) : NAN;
        if (tri) tri[i] = mesh ? h.a0 : -1;
        if (uvw) { uvw[3 * i] = mesh ? h.u : 0.0f; uvw[3 * i + 1] = mesh ? h.v : 0.0f; uvw[3 * i + 2] = mesh ? h.w : 0.0f; }
        if (exiting || geom) {
            Geom g;
            g.exiting = false;
            if (hit) {
                const rsx_primitive &p = scene->prims[(size_t)h.prim];
                const HRay l = to_local(p, r);
                if (csg) g = h.csg.g;
                else if (mesh) mesh_geom(scene->meshes[(size_t)p.mesh], l, h.t, h.a0, h.u, h.v, h.w, g);
                else analytic_geom(p, l, h.t, h.a0, h.a1, g);
            }
            if (exiting) exiting[i] = hit && g.exiting ? 1 : 0;
            if (geom) {
                double *o = geom + 12 * i;
                for (int k = 0; k < 3; ++k) {
                    o[k] = hit ? g.hit[k] : NAN; o[3 + k] = hit ? g.inside[k] : NAN;
                    o[6 + k] = hit ? g.outside[k] : NAN; o[9 + k] = hit ? g.normal[k] : NAN;
                }
            }
        }
    }
    return RSX_OK;
}

// One ray with two pointers (a foreign-function call pays per argument): in = {origin xyz, direction xyz, max_distance},
// out = {primitive id (-1: none), t, exiting, triangle (-1: not a mesh), u, v, w, hit xyz, inside xyz, outside xyz, normal xyz}
extern "C" int rsx_hit_host_one(const rsx_host_scene *scene, const double *in, double *out) {
    if (!scene || !in || !out) return rsx_fail(RSX_EINVAL, "rsx_hit_host_one: null argument");
    int32_t prim = -1, tri = -1;
    double t = NAN;
    uint8_t exiting = 0;
    float uvw[3] = {0, 0, 0};
    const int rc = rsx_hit_host(scene, 1, in, in + 3, in + 6, &prim, &t, &exiting, &tri, uvw, out + 7);
    if (rc) return rc;
    out[0] = (double)prim; out[1] = t; out[2] = (double)exiting; out[3] = (double)tri; out[4] = (double)uvw[0]; out[5] = (double)uvw[1]; out[6] = (double)uvw[2];
    return RSX_OK;
}

// World.contains — kdtree3d.pyx:736-792, kdtree.pyx:126-162; inside[i, j] = primitive j of World.primitives contains point i
extern "C" int rsx_contains_host(const rsx_host_scene *scene, int64_t n, const double *points, uint8_t *inside) {
    if (!scene || n < 0 || (n > 0 && (!points || !inside))) return rsx_fail(RSX_EINVAL, "rsx_contains_host: bad arguments");
    const int32_t nw = scene->n_world;
    for (int64_t i = 0; i < n; ++i) {
        const double px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
        for (int32_t j = 0; j < nw; ++j) inside[i * nw + j] = 0;
        if (scene->wnodes.empty() || !aabb_contains(scene->wlower, scene->wupper, px, py, pz)) continue;
        int32_t node = 0;
        while (scene->wnodes[(size_t)node].type >= 0) {
            const rsx_kdnode &nd = scene->wnodes[(size_t)node];
            node = sel3(nd.type & 3, px, py, pz) < nd.u.split ? node + 1 : nd.count;
        }
        const rsx_kdnode &leaf = scene->wnodes[(size_t)node];
        for (int32_t k = 0; k < leaf.count; ++k) {
            const int32_t idx = scene->witems[(size_t)leaf.u.leaf.first_item + k];
            const rsx_primitive &p = scene->prims[(size_t)idx];
            const bool in = aabb_contains(p.box_lower, p.box_upper, px, py, pz) && leaf_contains(*scene, p, px, py, pz);
            inside[i * nw + idx] = in ? 1 : 0;
        }
    }
    return RSX_OK;
}
