/*
 * rsx.h — C-ABI of librsx, the MI355X (gfx950) ray/primitive intersection, KD-tree traversal
 * and spectral-accumulation library that sits under Raysect's Python API.
 *
 * Everything here is plain C: pointers, sizes, POD structs. No torch / C++ types cross the boundary.
 * Ownership: the caller owns every buffer it passes; the library owns what hides behind the opaque
 * handles (rsx_ctx, rsx_scene, rsx_kd) and frees it in the matching *_free call.
 * Errors: every call returns RSX_OK (0) or a negative RSX_E* code; rsx_last_error() gives the text
 * (thread-local). No exceptions cross the ABI. One rsx_ctx per GPU; calls on one ctx are serialised
 * by the caller (mirrors the reference's one-process-one-world model).
 *
 * Each entry point names the reference interface (file:line under raysect/source) it replaces.
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSX_OK            0
#define RSX_EINVAL       -1   /* bad argument */
#define RSX_ENODEV       -2   /* no usable gfx950 device / HIP failure at init */
#define RSX_EHIP         -3   /* HIP runtime error (text in rsx_last_error) */
#define RSX_ENOMEM       -4
#define RSX_EUNSUPPORTED -5   /* something outside the device path (CSG nesting beyond 64 levels, librccl missing, ...) */

/* ---- primitive and material enumerations ------------------------------------------------------ */
enum {
    RSX_PRIM_SPHERE = 0,    /* raysect/primitive/sphere.pyx   params[0]=radius                       */
    RSX_PRIM_BOX = 1,       /* raysect/primitive/box.pyx      params[0..2]=lower, params[3..5]=upper */
    RSX_PRIM_CYLINDER = 2,  /* raysect/primitive/cylinder.pyx params[0]=radius, params[1]=height     */
    RSX_PRIM_MESH = 3,      /* raysect/primitive/mesh/mesh.pyx  mesh = index into rsx_scene_desc.meshes */
    RSX_PRIM_UNION = 4,     /* raysect/primitive/csg.pyx:262-383 */
    RSX_PRIM_INTERSECT = 5, /* raysect/primitive/csg.pyx:386-468 */
    RSX_PRIM_SUBTRACT = 6,  /* raysect/primitive/csg.pyx:471-599 */
    RSX_PRIM_NULL = 7       /* csg.pyx NullPrimitive: never hit, contains nothing */
};

enum {
    RSX_MAT_ABSORBER = 0,        /* optical/material/absorber.pyx:37-55  -> zero spectrum                     */
    RSX_MAT_UNIFORM_EMITTER = 1, /* optical/material/emitter/uniform.pyx:36-88 -> table[bin] * scale         */
    RSX_MAT_DEBUG_LIGHT = 2,     /* optical/material/debug.pyx:41-79 -> scale*max(0,-L_local.n) * table[bin] */
    /* Transparent boundaries: the ray carries on from the far side of the surface with its depth unchanged and Russian roulette
     * disabled (NullSurface / NullMaterial.evaluate_surface, optical/material/material.pyx:118-178), and every segment of the path
     * collects the volume emission of the primitives that contain the segment's origin (Ray._sample_volumes, optical/ray.pyx:422-455;
     * HomogeneousVolumeEmitter.evaluate_volume, emitter/homogeneous.pyx:55-102): spectrum[bin] += (table[bin] * scale) * length. */
    RSX_MAT_NULL = 3,                    /* NullMaterial: null surface, no volume                                   */
    RSX_MAT_UNIFORM_VOLUME_EMITTER = 4,  /* UniformVolumeEmitter (emitter/uniform.pyx:91-131): null surface + table[bin] * scale per unit length */
    /* Lambert (optical/material/lambert.pyx:40-112 under ContinuousBSDF.evaluate_surface, material.pyx:286-361, without important
     * primitives): one cosine-weighted daughter ray (HemisphereCosineSampler, core/math/sampler/solidangle.pyx:208-238) from the
     * incident side, depth + 1, Russian roulette per Ray.trace (optical/ray.pyx:382-388);
     * spectrum = trace(daughter) * table[bin] * pdf / pdf. Stochastic: RSX_RNG_PHILOX only. */
    RSX_MAT_LAMBERT = 5,
    /* Dielectric (optical/material/dielectric.pyx:125-328): scale = refractive index inside, light_dir[0] = index outside (both
     * SpectralFunction.average over the slice's wavelength range), light_dir[1] != 0 = transmission_only, table = transmission per
     * metre. Surface: refracted or reflected daughter chosen with probability(transmission) (Fresnel, unpolarised), total internal
     * reflection when 1 - (n1/n2)^2 (1 - cos^2) <= 0; depth + 1, Russian roulette. Volume: every segment starting inside the
     * primitive scales the spectrum by pow(table[bin], world-space segment length). Stochastic: RSX_RNG_PHILOX only. */
    RSX_MAT_DIELECTRIC = 6
};

/* ---- flattened KD-tree (raysect/core/math/spatial/kdtree3d.pxd:38-43 `kdnode`, 32 B -> 16 B) -- */
typedef struct rsx_kdnode {
    int32_t type;   /* -1 = leaf, 0/1/2 = split axis                      */
    int32_t count;  /* leaf: item count; branch: index of the upper child (lower child is id+1) */
    union {
        double split;                                   /* branch: split plane coordinate */
        struct { int32_t first_item; int32_t pad; } leaf; /* leaf: offset of its ids in items[] */
    } u;
} rsx_kdnode;

typedef struct rsx_kdtree {
    const rsx_kdnode *nodes;
    const int32_t *items;
    int32_t n_nodes;
    int32_t n_items;
    int32_t max_depth;      /* depth cap the tree was built with (bounds the traversal stack) */
    int32_t pad;
    double lower[3];        /* KDTree3DCore.bounds */
    double upper[3];
} rsx_kdtree;

/* ---- mesh data (raysect/primitive/mesh/mesh.pxd:43-60 MeshData), shared by instances ----------- */
typedef struct rsx_meshdata {
    const float *vertices;        /* [n_vertices,3] f32 */
    const int32_t *triangles;     /* [n_triangles,tri_stride] (after the degenerate filter) */
    const float *vertex_normals;  /* [n_normals,3] or NULL */
    const float *face_normals;    /* [n_triangles,3] f32 (mesh.pyx:428-462) */
    int32_t n_vertices;
    int32_t n_triangles;
    int32_t n_normals;
    int32_t tri_stride;           /* 3, or 6 when vertex-normal indices follow */
    int32_t smoothing;
    int32_t closed;
    rsx_kdtree kd;                /* over triangle AABBs (mesh.pyx:464-504) */
} rsx_meshdata;

/* ---- primitive table --------------------------------------------------------------------------
 * primitives[0 .. n_world) are World.primitives in registration order: that index is the
 * "primitive id" reported by hits (world KD item id, core/acceleration/kdtree.pyx:52-55).
 * CSG operands follow; their transforms/boxes are relative to the owning CSG node (CSGRoot). */
typedef struct rsx_primitive {
    int32_t type;
    int32_t material;     /* index into the render call's material table (world primitives only) */
    int32_t mesh;         /* RSX_PRIM_MESH: index into meshes[] */
    int32_t child_a;      /* CSG: primitive indices of operands A and B */
    int32_t child_b;
    int32_t pad;          /* ignored on input (the library's device copy keeps per-record flags here: bit 0 = to_local keeps directions, bit 1 = affine) */
    double params[6];
    double to_local[16];  /* Node.to_local(): root -> primitive space, row-major 4x4 */
    double to_root[16];   /* Node.to_root()  */
    double box_lower[3];  /* BoundPrimitive.box = primitive.bounding_box() in root space */
    double box_upper[3];
} rsx_primitive;

typedef struct rsx_scene_desc {
    const rsx_primitive *primitives;
    const rsx_meshdata *meshes;
    int32_t n_primitives;
    int32_t n_world;
    int32_t n_meshes;
    int32_t pad;
    rsx_kdtree world_kd;  /* _PrimitiveKDTree over the n_world primitive boxes */
} rsx_scene_desc;

/* ---- render inputs ---------------------------------------------------------------------------- */
typedef struct rsx_camera {   /* optical/observer/imaging/pinhole.pyx:148-204 */
    int32_t nx, ny;
    double image_delta, image_start_x, image_start_y;
    double to_root[16];
    double sensitivity;       /* _pixel_sensitivity(); used only by the power pipeline */
} rsx_camera;

typedef struct rsx_material {
    int32_t type;
    int32_t table;            /* row of tables[] holding SpectralFunction.sample(min,max,bins) */
    double scale;             /* emitter scale / light intensity */
    double light_dir[3];      /* RSX_MAT_DEBUG_LIGHT: normalised world-space light direction */
} rsx_material;

#define RSX_RNG_STREAM 0  /* uniforms[] supplied by caller: 2 per sample in task order (reference MT19937-64 parity) */
#define RSX_RNG_PHILOX 1  /* on-device Philox4x32-10 keyed by (seed; pixel, sample): order/shard independent      */
/* Philox counter = (pixel, sample | draw << 48): draw 0 = the camera's sub-pixel jitter; for a ray of depth d >= 1 draw 2d is its
 * Russian-roulette uniform and draw 2d + 1 the pair of uniforms of the scattering event that spawns the depth d + 1 ray (draw 1
 * for the primary ray's). Paths therefore reproduce whatever wave, GPU or rank renders them. */

/* One important primitive of the world's ImportanceManager (optical/scenegraph/world.pyx:47-230): world-space bounding sphere
 * (primitive.bounding_sphere()), cumulative selection probability (_calculate_cdf) and importance / total_importance. */
typedef struct rsx_important_sphere {
    double centre[3];
    double radius;
    double cdf;
    double weight;
} rsx_important_sphere;

typedef struct rsx_render_desc {
    rsx_camera camera;
    const rsx_material *materials;  /* one per world primitive material id */
    const double *tables;           /* [n_tables, bins] f64, this slice's bins */
    const int32_t *tasks;           /* [n_tasks,2] (ix,iy) or NULL -> rect tile in iy-outer/ix-inner order */
    const double *uniforms;         /* RSX_RNG_STREAM: [n_tasks*spp*2] */
    int64_t n_tasks;
    int32_t rect[4];                /* x0,y0,x1,y1 when tasks == NULL */
    int32_t n_materials;
    int32_t n_tables;
    int32_t bins;                   /* bins in this spectral slice */
    int32_t spp;                    /* pixel_samples */
    int32_t power;                  /* 1: SpectralPower (sample*sensitivity), 0: SpectralRadiance */
    int32_t rng_mode;
    uint64_t seed;
    uint64_t sample_offset;         /* RSX_RNG_PHILOX: first sample counter (sample-sharded ranks use rank*spp) */
    /* secondary rays (optical/ray.pyx:72-106; observer defaults observer.pyx:114-127). Only read when a material spawns them. */
    int32_t ray_max_depth;            /* Ray.max_depth */
    int32_t ray_extinction_min_depth; /* Ray.extinction_min_depth */
    double ray_extinction_prob;       /* Ray.extinction_prob */
    /* multiple importance sampling of ContinuousBSDF materials (material.pyx:327-352): active when n_important > 0, i.e. when
     * ray.importance_sampling and world.has_important_primitives(). Philox draws of a scattering event at depth d: block 2d+1 =
     * (probability(important_path_weight), sphere selection), the same block with bit 63 of the pixel word set = the direction pair. */
    const rsx_important_sphere *important;
    int32_t n_important;
    int32_t passes;                   /* 0 / 1: one pass. K > 1 (rsx_render_pinhole_frame, RSX_RNG_PHILOX; path-traced scenes included): K consecutive
                                       * passes of spp samples per pixel in this one call — the frame K calls with sample_offset advanced by spp
                                       * each time would leave (Observer.observe() called K times, observer.pyx:265-309), bit for bit. */
    double important_path_weight;     /* Ray.important_path_weight */
} rsx_render_desc;

typedef struct rsx_ctx rsx_ctx;
typedef struct rsx_scene rsx_scene;
typedef struct rsx_kd rsx_kd;

/* ---- library / device ------------------------------------------------------------------------- */
/* Opens HIP device `device_ordinal` (must be gfx950). No reference analogue (the reference is CPU-only). */
int rsx_init(int device_ordinal, rsx_ctx **out);
void rsx_free(rsx_ctx *ctx);
const char *rsx_last_error(void);
const char *rsx_version(void);
/* Launch all kernels of this ctx on an external HIP stream (e.g. torch's current stream); NULL = own stream. */
int rsx_set_stream(rsx_ctx *ctx, void *hip_stream);
int rsx_synchronize(rsx_ctx *ctx);
/* Non-blocking: *idle = 1 when everything issued on the context (its stream and its render lanes) has completed, 0 while work is in
 * flight. The host layer uses it to decide when to submit the small passes it is holding back (HipEngine.auto_batch): an idle device is
 * handed what there is, a busy one lets the batch grow. */
int rsx_idle(rsx_ctx *ctx, int32_t *idle);
/* Duration (ms, HIP events on the launch stream) of the most recent kernel launched through this ctx. */
int rsx_last_kernel_ms(rsx_ctx *ctx, float *ms);
/* Durations (ms) of the two kernels of the most recent render call: sample trace and per-bin accumulation. */
int rsx_last_render_ms(rsx_ctx *ctx, float *trace_ms, float *accumulate_ms);
/* Per-call kernel durations of the last n render calls (oldest first). The library keeps a ring of HIP event
 * triples, so K asynchronous renders can be timed individually with a single synchronisation at the end. */
int rsx_render_history(rsx_ctx *ctx, int32_t n, float *trace_ms, float *accumulate_ms);
/* Device self-test: the traversal's hoisted-reciprocal division must equal the IEEE quotient bit for bit; counts mismatches
 * over n pseudo-random / adversarial operand pairs. */
int rsx_selftest_exact_division(rsx_ctx *ctx, uint64_t n, uint64_t seed, uint64_t *mismatches);
/* Tuning aid: when dev_buffer != NULL the next render calls write, per 64-ray work unit, {start, end} wall_clock64 ticks
 * (100 MHz) and the (workgroup << 8 | wave) that processed it into dev_buffer[n_units][12] (u64; slots 3.. are per-phase
 * cycle counters in RSX_PHASE_PROF builds). NULL switches it off. Tuning builds only (-DRSX_UNIT_STAMPS=1 / -DRSX_PHASE_PROF / -DRSX_UTIL_PROF):
 * a production build compiles the stamps out and answers a non-NULL buffer with RSX_EUNSUPPORTED. */
int rsx_debug_unit_times(rsx_ctx *ctx, void *dev_buffer);
/* Timeline of the last n render calls: t[n][4] = trace begin, trace end, merge begin, merge end in ms since the first of them. */
int rsx_render_timeline(rsx_ctx *ctx, int32_t n, float *t);
/* Device allocation helpers so non-torch callers can keep frames resident in HBM. */
int rsx_dev_alloc(rsx_ctx *ctx, size_t bytes, void **dptr);
int rsx_dev_free(rsx_ctx *ctx, void *dptr);
int rsx_dev_upload(rsx_ctx *ctx, void *dptr, const void *host, size_t bytes);
int rsx_dev_download(rsx_ctx *ctx, void *host, const void *dptr, size_t bytes);
int rsx_dev_memset(rsx_ctx *ctx, void *dptr, int value, size_t bytes);

/* ---- host-side builders (CPU in the reference too) --------------------------------------------- */
/* SAH KD-tree build, node-for-node identical to KDTree3DCore.__init__/_build/_split
 * (raysect/core/math/spatial/kdtree3d.pyx:126-486). aabbs = [n,6] (lower xyz, upper xyz); item id = row. */
int rsx_kd_build(const double *aabbs, int32_t n, int32_t max_depth, int32_t min_items,
                 double hit_cost, double empty_bonus, rsx_kd **out);
int rsx_kd_info(const rsx_kd *kd, rsx_kdtree *view);   /* view points into kd-owned memory */
/* Threads the host builders (rsx_kd_build, the mesh preprocessing) start: the smaller of the process's CPU affinity and its cgroup CPU
 * quota (cpu.max), RSX_HOST_THREADS overrides; their OpenMP workers sleep when a region ends. The reference builds on one thread
 * (kdtree3d.pyx:126-486); what this replaces is the OpenMP default — every hardware thread a container can see, spinning — which got
 * a process with a 16-core quota on a 256-thread node throttled for 60 - 90 ms at a time after every build. */
int rsx_host_team_size(void);
void rsx_kd_free(rsx_kd *kd);
/* Serialise exactly like KDTree3DCore.save() (kdtree3d.pyx:864-912); returns bytes written or needed. */
int64_t rsx_kd_serialise(const rsx_kd *kd, int32_t min_items, double hit_cost, double empty_bonus,
                         uint8_t *out, int64_t capacity);

/* MeshData.__init__ preprocessing (raysect/primitive/mesh/mesh.pyx:363-504):
 * filter degenerate triangles in place (returns the new count), face normals, padded triangle AABBs. */
int32_t rsx_mesh_filter_triangles(const float *vertices, int32_t *triangles, int32_t n_triangles, int32_t stride);
int rsx_mesh_face_normals(const float *vertices, const int32_t *triangles, int32_t n_triangles, int32_t stride, float *out);
int rsx_mesh_triangle_aabbs(const float *vertices, const int32_t *triangles, int32_t n_triangles, int32_t stride, double *out);
/* MeshData.bounding_box(to_world) (mesh.pyx:835-859): out = lower xyz, upper xyz */
int rsx_mesh_world_bbox(const float *vertices, int32_t n_vertices, const double *to_world, double *out);

/* MT19937-64 exactly as raysect/core/math/random.pyx:99-265 (seed() -> init_by_array64 of 312 words). */
typedef struct rsx_mt { uint64_t mt[312]; int32_t mti; int32_t pad; } rsx_mt;
void rsx_mt_seed_words(rsx_mt *st, const uint64_t *key, uint64_t key_length);
void rsx_mt_uniform(rsx_mt *st, int64_t n, double *out);

/* ---- scene ------------------------------------------------------------------------------------ */
/* Uploads the flattened scenegraph; replaces Accelerator.build (core/acceleration/accelerator.pxd:37-41,
 * kdtree.pyx:166-168) + the per-primitive state the reference keeps in Python objects. */
int rsx_scene_create(rsx_ctx *ctx, const rsx_scene_desc *desc, rsx_scene **out);
void rsx_scene_free(rsx_scene *scene);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* World.hit for a batch of rays (core/scenegraph/world.pyx:125-146 -> kdtree.pyx:170-175).
 * Host buffers. prim[i] = -1 on a miss. Optional outputs may be NULL.
 *   tri/uvw: MeshIntersection extras (mesh.pyx:85-135), -1 / 0 for non-mesh hits
 *   geom[n,12]: hit_point, inside_point, outside_point, normal in primitive-local space
 *               (core/intersection.pyx:47-54) */
int rsx_hit_batch(rsx_scene *scene, int64_t n, const double *origin, const double *direction,
                  const double *max_distance, int32_t *prim, double *t, uint8_t *exiting,
                  int32_t *tri, float *uvw, double *geom);
/* Same with every pointer a device pointer; asynchronous on the ctx stream. */
int rsx_hit_batch_dev(rsx_scene *scene, int64_t n, const double *origin, const double *direction,
                      const double *max_distance, int32_t *prim, double *t, uint8_t *exiting,
                      int32_t *tri, float *uvw, double *geom);

/* Primitive.hit + repeated next_intersection() on ONE primitive (sphere.pyx:115-168, box.pyx:157-232,
 * cylinder.pyx:148-285, csg.pyx:132-179, mesh.pyx:1178-1238): up to max_roots ordered roots per ray.
 * counts[n]; t[n,max_roots]; exiting[n,max_roots]. Optional (NULL to skip): geometry[n,max_roots,12] = hit, inside,
 * outside points and normal of each root in primitive space (intersection.pyx:36-106), triangle[n,max_roots] and
 * uvw[n,max_roots,3] for roots on a mesh surface (MeshIntersection, mesh.pyx:85-135; -1 / 0 elsewhere). Host buffers. */
int rsx_roots_batch(rsx_scene *scene, int32_t primitive, int64_t n, const double *origin, const double *direction,
                    const double *max_distance, int32_t max_roots, int32_t *counts, double *t, uint8_t *exiting,
                    double *geometry, int32_t *triangle, float *uvw);

/* World.contains for a batch of points (world.pyx:149-168 -> kdtree.pyx:126-162):
 * inside[n, n_world] = 1 where world primitive j contains point i. Host buffers. */
int rsx_contains_batch(rsx_scene *scene, int64_t n, const double *points, uint8_t *inside);

/* ---- one ray, one point: the host side (SURVEY.md 8b "Who calls it: World.hit (n = 1 -> CPU lib)") ---------------------------------
 * World.hit(ray) / World.contains(point) / Ray.trace of ONE ray from Python (core/scenegraph/world.pyx:125-168 over
 * core/acceleration/kdtree.pyx:73-162): a device round trip per ray costs ~70 us, the reference answers in ~1 us. These entry points
 * answer such calls on the host from the same flattened arrays the device scene is created from — the same operations in the same
 * order as the kernels (hit ids, distances, barycentrics and geometry are the device's, bit for bit). They exist for the single-ray
 * API only: render calls, batch queries and the bench never use them. CSG solids are answered by the reference's stream merge
 * (csg.pyx:132-234, 326-599) with per-thread operand state. No device is needed. Arguments as rsx_hit_batch / rsx_contains_batch (host
 * pointers; max_distance, t, exiting, tri, uvw, geom may be NULL). */
typedef struct rsx_host_scene rsx_host_scene;
int rsx_host_scene_create(const rsx_scene_desc *desc, rsx_host_scene **out);   /* copies what it needs from desc */
void rsx_host_scene_free(rsx_host_scene *scene);
int rsx_hit_host(const rsx_host_scene *scene, int64_t n, const double *origin, const double *direction, const double *max_distance,
                 int32_t *prim, double *t, uint8_t *exiting, int32_t *tri, float *uvw, double *geom);
/* one ray through two pointers: in[7] = origin, direction, max_distance; out[19] = primitive id (-1: none), t, exiting, triangle (-1: not a
 * mesh), u, v, w, then hit / inside / outside point and normal (12 values, as rsx_hit_batch's geom) */
int rsx_hit_host_one(const rsx_host_scene *scene, const double *in, double *out);
int rsx_contains_host(const rsx_host_scene *scene, int64_t n, const double *points, uint8_t *inside);

/* One spectral slice of Observer.observe(): _render_pixel for every task (observer.pyx:363-419) —
 * pinhole ray generation, Ray.trace (optical/ray.pyx:338-401) with closed-form materials, per-pixel
 * Welford accumulation (statsarray.pyx:743-776). Outputs per task: mean[n_tasks,bins], variance[n_tasks,bins]
 * (the tuple _render_pixel returns). Host buffers unless *_dev. ray_count may be NULL. */
int rsx_render_pinhole(rsx_scene *scene, const rsx_render_desc *desc, double *mean, double *variance,
                       uint64_t *ray_count);
/* Fused form: results are merged straight into a device-resident frame with the combine_samples law
 * (Pipeline2D.update, optical/observer/pipeline/spectral/power.pyx:424-437; statsarray.pyx:623-668,780-859).
 * frame_* are device pointers to [nx,ny,frame_bins] (x-major, as StatsArray3D); the slice occupies
 * bins [slice_offset, slice_offset+desc->bins). desc->tasks/uniforms/materials/tables are host pointers.
 * Asynchronous on the ctx stream: the frame is complete after rsx_synchronize() (or any later call on the ctx). */
int rsx_render_pinhole_frame(rsx_scene *scene, const rsx_render_desc *desc, double *frame_mean,
                             double *frame_variance, int32_t *frame_samples, int32_t frame_bins,
                             int32_t slice_offset, uint64_t *ray_count);
/* The spectral slices of one observe() (observer.pyx:289-300: one engine.run per slice) touch disjoint bins of the frame, so their
 * passes need not wait for one another. Between rsx_defer_path_checks(ctx, 1) and rsx_collect_path_checks, rsx_render_pinhole_frame
 * calls of path-traced scenes (scattering / refracting / volume materials) return without the end-of-pass round trip: the passes
 * overlap on the device, ray_count comes back as UINT64_MAX (= this call was deferred), and a pass that could not finish (term arena exhausted, more volumes at a point
 * than the fast kernel keeps) leaves the frame untouched. rsx_collect_path_checks waits for the passes, switches nothing off, and
 * reports: failed_calls[0 .. *n_failed) = indices (0 = first deferred call since the last collect / switch-on) of the passes the
 * caller must issue again with deferral off — they then take the ordinary retry path — and *ray_count = rays of all deferred
 * passes. A path that crossed more surfaces than the build allows makes it return RSX_EUNSUPPORTED as the undeferred call does. */
int rsx_defer_path_checks(rsx_ctx *ctx, int32_t on);
int rsx_collect_path_checks(rsx_ctx *ctx, int32_t *failed_calls, int32_t capacity, int32_t *n_failed, uint64_t *ray_count);

/* How the render calls of path-traced scenes (Ray.trace with scattering materials, optical/ray.pyx:338-455) are scheduled on the device.
 * mode 1: level by level — one launch per path segment over lists of live paths filed by the material arm they wait for (a wave runs
 * ONE arm on 64 paths that all need it, then walks their daughters' segments); mode 0: one persistent kernel in which every lane carries a path from its camera ray to its end; mode -1:
 * the library's default (the one-kernel form, which measures faster on MI355X today; $RSX_WAVEFRONT=1 makes it mode 1 for calls of at
 * least $RSX_WF_MIN_PATHS paths that no other call overlaps).
 * Scenes without a mesh primitive run forms of both that leave the wave-cooperative mesh walk out (its registers cost the plain
 * one-kernel form a wave per SIMD: Cornell box 28.3 -> 25.1 ms per pass); mode 2 / 3 = mode 0 / 1 with the general forms kept (A/B, tests).
 * min_paths < 0 keeps the default threshold. Frames are identical either way: the random numbers, sample record and term list of a
 * path are keyed by (pixel, sample), never by the lane or launch that renders it. */
int rsx_set_path_stages(rsx_ctx *ctx, int32_t mode, int64_t min_paths);

/* XYZPixelProcessor (optical/observer/pipeline/rgb.pyx:534-562) for one spectral slice: every sample's spectrum (times its
 * projection weight) is projected on the CIE XYZ curves resampled over the slice (spectrum_to_ciexyz, optical/colour.pyx:158-187:
 * sum over bins of delta_wavelength * sample[bin] * resampled_xyz[bin, c]), times camera.sensitivity, and the three channels go
 * through the Welford accumulator. resampled_xyz: host [desc->bins, 3]; mean / variance: host [n_tasks, 3] — what
 * XYZPixelProcessor.pack_results() returns per pixel; RGBPipeline2D.update/finalise (rgb.pyx:249-289) stay on the host. */
int rsx_render_pinhole_xyz(rsx_scene *scene, const rsx_render_desc *desc, const double *resampled_xyz, double delta_wavelength,
                           double *mean, double *variance, uint64_t *ray_count);

/* StatsArray3D.combine_samples applied elementwise to two frames resident on the device
 * (statsarray.pyx:780-859): a <- combine(a, b). Used to merge passes / sample-sharded ranks. */
int rsx_frame_combine_dev(rsx_ctx *ctx, int64_t n, double *mean_a, double *var_a, int32_t *n_a,
                          const double *mean_b, const double *var_b, const int32_t *n_b);

/* ---- device-side known-answer entry points -----------------------------------------------------------------------------------
 * Each runs the device functions / kernels of the render path on caller-supplied operands so that the reference's golden vectors can
 * be checked against the DEVICE directly (tests/test_gpu_parity.py::test_device_known_answers), not only through rendered frames.
 * BoundingBox3D.intersect (core/boundingbox.pyx:180-245): result[n,3] = hit, front, back; *mismatches counts rays on which the
 * hoisted-reciprocal form the traversal uses and the plain form disagree (must be 0). */
int rsx_selftest_aabb(rsx_ctx *ctx, int64_t n, const double *lower, const double *upper, const double *origin, const double *direction,
                      double *result, uint64_t *mismatches);
/* PinholeCamera._generate_rays + the observer's transform to world space (pinhole.pyx:169-204, observer.pyx:403-404) as the render
 * kernels perform them: rays[n_tasks * spp, 7] = origin, direction, projection weight. desc: camera, tasks (required), spp, rng. */
int rsx_selftest_camera_rays(rsx_ctx *ctx, const rsx_render_desc *desc, double *rays);
/* StatsArray _add_sample (statsarray.pyx:743-776) through k_accumulate: x[n_chains, spp] -> (mean, variance)[n_chains] after the
 * spp samples; both instantiations of the kernel run and must agree. */
int rsx_selftest_welford(rsx_ctx *ctx, int64_t n_chains, int32_t spp, const double *x, double *mean, double *variance);
/* The portable math of the path kernels: op 0 = pow(a, b) -> out0 (Beer-Lambert attenuation, dielectric.pyx:325-326),
 * op 1 = (sin a, cos a) -> (out0, out1), op 2 = asin(a) -> out0 (solidangle.pyx:228-233, world.pyx:150-188). */
int rsx_selftest_math(rsx_ctx *ctx, int32_t op, int64_t n, const double *a, const double *b, double *out0, double *out1);

/* ---- multi-GPU: the spectral framebuffer over RCCL / xGMI (one process per GPU) ---------------------------------------------
 * Replaces the result queue of MulticoreEngine (raysect/core/workflow.py:201-251: workers send per-task (mean, variance) blocks, the
 * parent folds them into the frame, pipeline/spectral/power.pyx:424-437). Rays never cross GPUs; the frames meet once per render.
 * librccl is dlopen'ed by the first rsx_comm_* call (RSX_EUNSUPPORTED if it cannot be loaded); nothing else in librsx needs it.
 * Bootstrap: rank 0 calls rsx_comm_unique_id and hands the 128 bytes to every rank by any means (file, socket, MPI, a
 * torch.distributed store); then every rank calls rsx_comm_create (collective, blocks until all ranks arrive). All calls run on the
 * ctx stream; frame pointers are DEVICE pointers (the frames rsx_render_pinhole_frame accumulates into). */
typedef struct rsx_comm rsx_comm;
#define RSX_COMM_ID_BYTES 128
int rsx_comm_unique_id(void *id128);
int rsx_comm_create(rsx_ctx *ctx, int32_t n_ranks, int32_t rank, const void *id128, rsx_comm **out);
void rsx_comm_free(rsx_comm *comm);
int rsx_comm_barrier(rsx_comm *comm);                       /* all ranks' ctx streams have drained up to here */
int rsx_comm_max_f64(rsx_comm *comm, double *value);        /* in place: max over ranks of a host double (step timing) */
/* Tile sharding (SURVEY.md 8e default): rank r rendered frame elements [shard_begin[r], shard_begin[r+1]) of the x-major arrays — a
 * column tile [x0, x1) x [0, ny) is the contiguous run [x0 * ny * bins, x1 * ny * bins). In place; afterwards every rank holds the
 * whole frame, bit-identical to a one-GPU render (no arithmetic in the collective). shard_begin: host [n_ranks + 1]. */
int rsx_allgather_frame(rsx_comm *comm, double *frame_mean, double *frame_variance, int32_t *frame_samples, const int64_t *shard_begin);
/* Sample sharding: every rank rendered the whole n-element frame with its own samples; afterwards every rank holds the
 * StatsArray3D.combine_samples fold (statsarray.pyx:780-859) of the ranks' frames in rank order — deterministic and identical on
 * every rank. Routed as reduce-scatter + all-gather (xGMI is point to point: 2 (W-1)/W frames per rank instead of W-1). */
int rsx_allreduce_frame(rsx_comm *comm, double *frame_mean, double *frame_variance, int32_t *frame_samples, int64_t n);
/* The segment of an n-element frame that rank `rank` owns in rsx_allreduce_frame's reduce-scatter: ceil(n / n_ranks) elements, cut at n
 * (rank == n_ranks gives the end, length 0). Pure host arithmetic — exported so that it can be checked without a GPU. */
int rsx_frame_segment(int64_t n, int32_t n_ranks, int32_t rank, int64_t *offset, int64_t *length);
/* Slice sharding (the slice loop of observer.pyx:299-340 split over the GPUs): rank r rendered the spectral slices that
 * fill bins [bin_begin[r], bin_begin[r+1]) of every pixel of the [n_pixels, bins] frame arrays. In place; afterwards every rank holds
 * every bin, bit-identical to a one-GPU render (packed bin planes over direct sends / receives, no arithmetic). bin_begin: host
 * [n_ranks + 1], bin_begin[0] = 0, bin_begin[n_ranks] = bins. */
int rsx_allgather_bins(rsx_comm *comm, double *frame_mean, double *frame_variance, int32_t *frame_samples, int64_t n_pixels, int32_t bins,
                       const int32_t *bin_begin);
/* How many ranks RCCL says the communicator spans (ncclCommCount). */
int rsx_comm_size(rsx_comm *comm, int32_t *n_ranks);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H */
