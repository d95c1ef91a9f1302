/*
 * amber_oracle.h -- C interface of the CPU ORACLE for the amber path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / reported CPU baseline.  The product (amber_amd/) never links or calls it.
 *
 * The oracle is a CPU restatement (own code, C++17, g++) of the reference algorithm
 *   src/amber/rendering/algorithm_pt.cc:112-160  and everything it calls
 * (SURVEY.md section 8(a) rows a1..a20).  Every function in amber_oracle.cc cites the
 * reference file:line it follows.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"): the reference has no tests, golden vectors or
 * fixtures for this path and cannot be built in this image (its hot path
 * includes boost/operators.hpp and boost/optional.hpp; boost is not installed and writing
 * stand-ins for missing headers is not allowed).  The only reference outputs available are the
 * whole-image known answers recorded in SURVEY.md section 8(c) / BASELINE.md section 2 (unmodified
 * reference, g++ 11.4, seed 12345, one thread).  The oracle's "MT-stream / BVH / libm" mode is
 * checked bit-for-bit against those (tests/test_oracle_pin.py).  mt19937_64 itself is pinned by
 * the C++ standard's 10000th-output known answer.
 */
#ifndef AMBER_ORACLE_H
#define AMBER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flat scene description (plain data; same layout as include/amber_hip.h) ---------- */

enum { ORACLE_PRIM_TRIANGLE = 0, ORACLE_PRIM_SPHERE = 1, ORACLE_PRIM_DISK = 2, ORACLE_PRIM_CYLINDER = 3 };
enum {
  ORACLE_MAT_LAMBERTIAN = 0, ORACLE_MAT_PHONG = 1, ORACLE_MAT_SPECULAR = 2,
  ORACLE_MAT_REFRACTION = 3, ORACLE_MAT_DIFFUSE_LIGHT = 4, ORACLE_MAT_EYE = 5
};

typedef struct {
  uint32_t kind;       /* ORACLE_PRIM_* */
  uint32_t material;   /* index into materials */
  float    p[9];       /* triangle: v0,v1,v2 ; sphere: center, radius ; disk: center,normal,radius ; cylinder: center,normal,radius,height */
} oracle_object;

typedef struct {
  uint32_t kind;       /* ORACLE_MAT_* */
  float    rho[3];     /* kd / ks / radiance ; (1,1,1) for refraction and eye */
  float    param;      /* phong exponent / refraction ior */
} oracle_material;

typedef struct {
  float    transform[16];   /* row-major Matrix4 */
  float    focal_length, focus_distance, radius;
  uint32_t n_blades;
} oracle_thin_lens;

typedef struct {
  uint32_t width, height;
  float    scene_width, scene_height;
} oracle_sensor;

/* modes */
enum { ORACLE_ACCEL_BVH = 0, ORACLE_ACCEL_LIST = 1,
       /* The List scan's answer (acceleration_list.h:51-68: the closest hit over ALL objects, the lower insertion index on a distance tie)
        * computed through the reference's tree with conservative culling: boxes that contain every point at which the reference's
        * binary32 primitive tests can accept a ray, and no order-dependent early exit (amber_oracle.cc, BVH::CastCons).  It exists because
        * the reference's own two accelerations disagree on large scenes (its BVH loses grazing hits its List keeps) and the plain scan
        * is O(n) per ray; tests/test_oracle_conservative_bvh.py proves it equal to the plain scan.  A scene created with this mode
        * can be switched between all three (oracle_scene_set_accel). */
       ORACLE_ACCEL_BVH_CONS = 2 };
/* Phong rejection sampling (material_phong.cc:81-106) gives up re-sampling at this attempt; same constant on the device */
#define ORACLE_PHONG_MAX_TRIES 1024
/* OR into `accel` of oracle_scene_create: append the aperture blades AFTER the objects, the order cli::ImportScene
 * produces (import.cc:155-157); default is blades first (cornel_box.cc:62-64). */
enum { ORACLE_BLADES_LAST = 0x100 };
/* LIBM: the live glibc calls the reference makes (sincosf, powf, pow).  PORTABLE: round 1's own + - * / forms (kept to
 * measure the distance between the two).  GLIBC: the restatement of glibc 2.35's x86-64 FMA-variant sincosf / powf that
 * the gfx950 engine executes (amber_oracle.cc "GLIBC mode"); equal to LIBM on every FMA-capable x86-64 host for every
 * argument the path can produce (tests/test_math_modes.py), and independent of the host's libm. */
enum { ORACLE_MATH_LIBM = 0, ORACLE_MATH_PORTABLE = 1, ORACLE_MATH_GLIBC = 2 };

typedef struct oracle_scene oracle_scene;

/* The reference's etude::CornelBox (src/amber/etude/cornel_box.cc:38-204). */
oracle_scene* oracle_scene_cornell_box(float focal_length, float aperture_radius, uint32_t n_blades, int accel);
/* Arbitrary scene: the lens' aperture triangles are PREPENDED as objects 0..n_blades-1 with an
 * Eye material appended to the material table (mirrors cornel_box.cc:62-64). */
oracle_scene* oracle_scene_create(const oracle_object* objects, uint32_t n_objects,
                                  const oracle_material* materials, uint32_t n_materials,
                                  const oracle_thin_lens* lens, int accel);
void          oracle_scene_destroy(oracle_scene*);
/* Switches the acceleration Scene::Cast uses: LIST always; BVH needs a scene created with BVH or BVH_CONS; BVH_CONS one created with
 * BVH_CONS.  Returns 0, or -1 when the scene lacks what the mode needs.  Not thread-safe against running renders. */
int           oracle_scene_set_accel(oracle_scene*, int accel);

/* introspection (for cross-checking the product's own flattening) */
uint32_t oracle_scene_object_count(const oracle_scene*);
uint32_t oracle_scene_material_count(const oracle_scene*);
void     oracle_scene_get_object(const oracle_scene*, uint32_t i, oracle_object* out, float normal_out[3]);
void     oracle_scene_get_material(const oracle_scene*, uint32_t i, oracle_material* out, float* r0_out);
void     oracle_scene_get_lens(const oracle_scene*, float origin[3], float global_[9], float local_[9],
                               float* focus_distance, float* sensor_distance, float* p_area);
/* BVH stats: nodes, leaves, max depth (reference BVH, acceleration_bvh.h:134-312) */
void     oracle_scene_bvh_stats(const oracle_scene*, uint32_t* n_nodes, uint32_t* n_leaves, uint32_t* max_depth);
/* objects_ after the build (position -> insertion index) ; FNV-1a64 over a pre-order walk of the tree (tag, box, leaf range) */
void     oracle_scene_bvh_order(const oracle_scene*, uint32_t* order_out /* n_objects */);
uint64_t oracle_scene_bvh_digest(const oracle_scene*);

typedef struct {
  uint64_t casts;    /* Scene::Cast calls (= "rays", algorithm_pt.cc:139) */
  uint64_t hits;     /* casts that hit something */
  uint64_t paths;
} oracle_counters;

/* O1: MT-stream mode -- restates PathTracing::Render with ONE thread and the reference's
 * sampler (mt19937_64 seeded with `seed`, stream running across pixels and passes), the
 * binary-counter Accumulator and Mean.  out_rgb: W*H*3 floats, index (x + y*W)*3 + c. */
void oracle_render_mt(const oracle_scene*, const oracle_sensor*, uint64_t seed, uint32_t spp,
                      int math, float* out_rgb, oracle_counters* counters);

/* XorShift mode -- per-(pixel,sample) sampler: seed = hash(global_seed, pixel, sample).
 * Adds, for every pixel, the path measurements of samples [first_sample, first_sample+n) to sum_rgb
 * (W*H*3, caller zero-initialises before the first pass): sequential f32 sums over chunks of `chunk`
 * samples, chunk sums added in order (the engine's AMBER_ACCUM_CHUNK; 0 = a single chunk).
 * Rows [y0, y1) only.  n_threads > 1 splits rows across threads (results are identical). */
void oracle_render_xorshift(const oracle_scene*, const oracle_sensor*, uint64_t global_seed,
                            uint32_t first_sample, uint32_t n_samples, uint32_t y0, uint32_t y1,
                            int math, uint32_t max_depth, uint32_t n_threads, uint32_t chunk,
                            float* sum_rgb, oracle_counters* counters);

/* Light tracing (algorithm_lt.cc:112-163) in XorShift mode; see amber_oracle.cc.  No reference output exists for it:
 * it is pinned only through the functions it shares with the pinned path tracer. */
uint64_t oracle_render_lt_xorshift(const oracle_scene*, const oracle_sensor*, uint64_t global_seed, uint32_t first_sample,
                                   uint32_t n_samples, int math, uint32_t max_depth, float* sum_rgb, oracle_counters* counters,
                                   uint32_t* records, uint64_t max_records);

/* Path signatures in XorShift mode for rows [y0, y1), samples [first_sample, first_sample + n): sig[((y - y0) * W + x) * n + k]
 * = FNV-1a-32 over the object index of every cast of the path (0xffffffff = miss) in the low word -- the path's discrete
 * history: two paths "diverged" iff these differ -- and FNV-1a-32 over the bits of every hit distance in the high word. */
void oracle_path_signatures(const oracle_scene*, const oracle_sensor*, uint64_t global_seed, uint32_t first_sample,
                            uint32_t n_samples, uint32_t y0, uint32_t y1, int math, uint32_t max_depth, uint32_t n_threads,
                            uint64_t* sig);

/* per-path trace in XorShift mode (for path-level parity tests) */
typedef struct {
  int32_t  object;       /* object index in oracle object order, -1 = miss */
  float    t;
  float    pos[3];
  float    weight[3];    /* path weight BEFORE this hit's scatter is applied */
  float    measurement[3]; /* running measurement after this hit */
} oracle_bounce;
/* returns number of casts performed; writes at most max_bounces records */
uint32_t oracle_trace_path(const oracle_scene*, const oracle_sensor*, uint64_t global_seed,
                           uint32_t px, uint32_t py, uint32_t sample, int math, uint32_t max_depth,
                           oracle_bounce* out, uint32_t max_bounces, float eye_ray_out[7]);

/* ---- known-answer entry points for unit parity tests --------------------------------- */
/* closest hit of one ray against the scene; returns object index or -1 */
int32_t oracle_cast(const oracle_scene*, const float origin[3], const float dir[3],
                    float* t, float pos[3], float normal[3]);
/* Closest hits of n rays (origins / dirs: 3 floats each) through acceleration `accel` of the scene (see oracle_scene_set_accel for
 * what a scene can serve), on n_threads threads: object index (-1 = miss) and distance (NaN = miss).  ORACLE_ACCEL_LIST is the plain
 * scan of acceleration_list.h:51-68 over every object; for scenes of many spheres it runs blocked (objects outer, rays inner) with
 * the sign of the discriminant -- SolveQuadratic's first exit, algebra.h:33-34, same binary32 operations -- evaluated eight spheres
 * at a time, and Intersect() called for the survivors: the same function of the ray, 1e11 pairs in seconds. */
void oracle_cast_many(const oracle_scene*, int accel, uint64_t n, const float* origins, const float* dirs, uint32_t n_threads,
                      int32_t* object_out, float* t_out);
/* The rays Scene::Cast is called with while rows [y0, y1), samples [first_sample, first_sample + n_samples) are rendered in XorShift
 * mode (every bounce of every path, in path order): at most max_rays are stored; returns the number cast. */
uint64_t oracle_collect_rays(const oracle_scene*, const oracle_sensor*, uint64_t global_seed, uint32_t first_sample, uint32_t n_samples,
                             uint32_t y0, uint32_t y1, int math, uint32_t max_depth, uint64_t max_rays, float* origins, float* dirs);
/* Why two accelerations of one scene disagree on a path (scene created with ORACLE_ACCEL_BVH_CONS).  Traces path (px, py, sample)
 * with acceleration accel_a; at every cast also asks accel_b.  At the FIRST cast on which the two differ (object or distance bits)
 * it fills `out` and stops:
 *   out[0] cast number (1-based), out[1] object of a, out[2] object of b, out[3] object of the plain List scan over all objects,
 *   out[4] bits of a's distance, out[5] bits of b's distance, out[6] bits of List's distance,
 *   out[7] 1 if the ray passes the reference's slab test (aabb.cc:28-62, t_max = FLT_MAX) on the GEOMETRIC box (Primitive::BoundingBox) of List's object, else 0,
 *   out[8] 1 if a's and b's distances are equal bit for bit (an exact tie between two objects), else 0,
 *   out[9] the same slab test limited to List's hit distance (t_max = that distance): 0 means the accepted hit lies IN FRONT OF the point
 *          where the ray enters the object's geometric box -- a BVH that searches a far child only up to the nearer hit, or returns the
 *          near child's hit when it lies before the far child's box (acceleration_bvh.h:386-391), cannot find it.
 * Returns 1 if a difference was found, 0 if the whole path agrees. */
int oracle_classify_path(const oracle_scene*, const oracle_sensor*, uint64_t global_seed, uint32_t px, uint32_t py, uint32_t sample,
                         int math, uint32_t max_depth, int accel_a, int accel_b, uint32_t out[10]);
/* single-primitive intersection (Primitive::Intersect); returns 1 on finite hit */
int oracle_intersect(const oracle_object* obj, const float origin[3], const float dir[3],
                     float* t, float pos[3], float normal[3]);
/* AABB slab test, prelude::Intersect (aabb.cc:28-62) */
int oracle_aabb_intersect(const float bmin[3], const float bmax[3], const float origin[3],
                          const float dir[3], float t_max, float* t_in, float* t_out);
/* material sampling: uniforms consumed in order from u[]; returns number consumed */
uint32_t oracle_sample_material(const oracle_material* m, const float normal[3], const float dir_out[3],
                                const double* u, uint32_t n_u, int math, float dir_in[3], float weight[3]);
void oracle_radiance(const oracle_material* m, const float normal[3], const float dir_out[3], float out[3]);
/* eye ray: 5 uniforms in draw order (blade, tri-u, tri-v, jitter-Y, jitter-X -- g++ order) */
void oracle_eye_ray(const oracle_scene*, const oracle_sensor*, uint32_t px, uint32_t py, const double u[5],
                    int math, float origin[3], float normal[3], float dir[3], float* weight, uint32_t* blade);

/* samplers */
void     oracle_mt_uniforms(uint64_t seed, uint32_t n, uint64_t* raw, double* u, float* uf);
uint64_t oracle_xorshift_seed(uint64_t global_seed, uint32_t pixel, uint32_t sample);
void     oracle_xorshift_uniforms(uint64_t state, uint32_t n, double* u);

/* portable math (bit-identical on CPU and gfx950) vs libm */
void  oracle_sincos(float phi, int math, float* s, float* c);
float oracle_pow(float x, float y, int math);

/* Number of arguments on which two math modes differ in any output bit.
 * kind 0: sincos(phi) for phi = (2.0f * pi_f) * (k * 2^-24), k in [k0, k1) -- the path's phi (sampling.h:245).
 * kind 1: pow(k * 2^-24, y), k in [k0, k1) -- CosinePower's pow(r0, 1 / (e + 1)) (sampling.h:279).
 * kind 2: sincos of n = k1 - k0 random binary32 arguments in (-120, 120) ; kind 3: pow of random positive (x, y) pairs
 *         with y scaled by `y`; both seeded by k0.  first_bad receives the first differing argument (if any). */
uint64_t oracle_math_compare(int kind, int mode_a, int mode_b, uint32_t k0, uint32_t k1, float y, float first_bad[2]);
/* pow(x, n) of a binary32 x in double for n = 4, 5 (lens_thin.cc:92-93,146, material_refraction.cc:271-275) */
double oracle_pow_i(float x, int n, int math);

/* Output stage: Filmic -> Gamma(2.2) -> 8-bit, postprocess/filmic.cc:30-66 + gamma.cc:36-52 as application.cc:98-108 chains
 * them, with the host's powf.  rgb: w*h*3 floats in Image order; out: w*h*3 bytes, same order (NOT mirrored). */
void oracle_tonemap(const float* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb8);

/* image helpers */
uint64_t oracle_fnv1a64(const void* data, uint64_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif
