/*
 * amber_hip.h -- C ABI of the MI355X (gfx950) path-tracing engine.
 *
 * This is the drop-in boundary for amber's unidirectional path tracer.  The reference side of
 * the boundary is
 *     rendering::Algorithm<RGB>::Render(scene, sensor, context)
 *         /root/reference/include/amber/rendering/algorithm.h:40-45
 * as implemented by PathTracing<RGB>
 *         /root/reference/src/amber/rendering/algorithm_pt.cc:82-160.
 * The reference has no FFI; an integrator inside amber would be a C++ class deriving from
 * Algorithm<RGB> that flattens the scene and calls the functions below (INTEGRATION.md shows
 * that adapter; amber_amd/csrc/amber/ holds a complete one).  Plain pointers and sizes only.
 *
 * Threading: a handle is owned by ONE host thread at a time (one handle per GPU).
 * Errors: every int-returning function returns AMBER_OK (0) or a negative AMBER_E* code and
 * records a message retrievable with amber_hip_last_error() (thread-local).
 * There is NO CPU fallback: without a usable HIP device amber_hip_pt_create fails with
 * AMBER_ENODEVICE.
 */
#ifndef AMBER_HIP_H
#define AMBER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the libraries are built with -fvisibility=hidden: what these headers declare is what they export */

#define AMBER_HIP_ABI_VERSION 3   /* 3 (round 5): the known-answer / signature entry points, engine WAVEFRONT and AMBER_PT_FLAG_BVH_POOL moved to the lab library
                                    (amber_hip_lab.h); AMBER_PT_FLAG_BVH_ITEMS added.  2 (round 4): lt ranges, stream; a stream-ordered read of
                                    amber_hip_pt_device_framebuffer() needs amber_hip_pt_sync() first when a launch may have run out of record slots */

/* Accumulation granule: within a render pass the samples of a pixel are summed sequentially in chunks of
 * AMBER_ACCUM_CHUNK consecutive samples (starting at first_sample), and the chunk sums are added to the
 * framebuffer in chunk order.  Part of the numerical contract (the oracle restates it).  8 keeps the work queue
 * fine-grained: with 32 the tail of an 8-way sharded render cost 10 % (EXPERIMENTS.md, multi-GPU). */
#define AMBER_ACCUM_CHUNK 8u

enum {
  AMBER_OK = 0,
  AMBER_EINVAL = -1,     /* bad argument / malformed flat scene */
  AMBER_ENODEVICE = -2,  /* no HIP device / device index out of range */
  AMBER_EHIP = -3,       /* a HIP runtime call failed (message has hipGetErrorString) */
  AMBER_ENOMEM = -4
};

/* ---- flattened scene -------------------------------------------------------------------
 * Produced by scene::Scene::Flatten() (amber_amd/csrc/amber/scene.h).  Object order is the
 * scene's insertion order: closest-hit ties are resolved toward the LOWER index, which is the
 * semantics of the reference's List acceleration (acceleration_list.h:51-68).
 */
enum { AMBER_PRIM_TRIANGLE = 0, AMBER_PRIM_SPHERE = 1, AMBER_PRIM_DISK = 2, AMBER_PRIM_CYLINDER = 3 };
enum {
  AMBER_MAT_LAMBERTIAN = 0,    /* material_lambertian.cc:61-70   rho = kd               */
  AMBER_MAT_PHONG = 1,         /* material_phong.cc:81-106       rho = ks, param = exponent */
  AMBER_MAT_SPECULAR = 2,      /* material_specular.cc:62-70     rho = ks               */
  AMBER_MAT_REFRACTION = 3,    /* material_refraction.cc:177-220 param = ior, r0 = Fresnel(ior) */
  AMBER_MAT_DIFFUSE_LIGHT = 4, /* material_diffuse_light.h:127-194 rho = radiance       */
  AMBER_MAT_EYE = 5            /* material_eye.h:146-155 (aperture pass-through)        */
};

typedef struct {
  uint32_t kind;       /* AMBER_PRIM_* */
  uint32_t material;   /* index into AmberFlatScene.materials */
  /* triangle: v0[3] v1[3] v2[3] normal[3]   (normal = Normalize(Cross(v1-v0, v2-v0)),
   *                                          primitive_triangle.cc:59-68, computed by the host)
   * sphere:   center[3] radius
   * disk:     center[3] normal[3] radius
   * cylinder: center[3] normal[3] radius height */
  float    p[12];
} AmberFlatObject;

typedef struct {
  uint32_t kind;       /* AMBER_MAT_* */
  float    rho[3];
  float    param;
  float    r0;
} AmberFlatMaterial;

/* lens (all derived values computed by the host object model).
 * kind AMBER_LENS_THIN:    lens_thin.cc:32-57.
 * kind AMBER_LENS_PINHOLE: lens_pinhole.cc:31-106; sensor_distance as given, n_blades = 1 (the degenerate aperture
 *                          triangle origin/origin/origin that the reference inserts into the scene), focus_distance and
 *                          p_area unused. */
enum { AMBER_LENS_THIN = 0, AMBER_LENS_PINHOLE = 1 };
typedef struct {
  float    origin[3];
  float    global_[9];         /* Matrix3, row-major */
  float    local_[9];          /* global_.Inverse() */
  float    focus_distance;
  float    sensor_distance;    /* 1 / (1/focal_length - 1/focus_distance) */
  float    p_area;             /* 1 / (blade area * n_blades) */
  uint32_t n_blades;
  uint32_t first_blade_object; /* objects[first_blade_object + i] is aperture blade i */
  uint32_t kind;               /* AMBER_LENS_* */
} AmberFlatThinLens;

/* One entry per SurfaceType::Light object, in the order of scene::LightSet (sorted by power, light_set.h:61-82):
 * cumulative power, pdf_area = Sum(Irradiance) / total power, Irradiance = radiance * pi.  Used by light tracing only. */
typedef struct {
  uint32_t object;        /* index into objects */
  float    cum_power;
  float    pdf_area;
  float    irradiance[3];
} AmberFlatLight;

typedef struct {
  const AmberFlatObject*   objects;
  uint32_t                 n_objects;
  const AmberFlatMaterial* materials;
  uint32_t                 n_materials;
  AmberFlatThinLens        lens;
  const AmberFlatLight*    lights;      /* may be NULL when n_lights == 0 */
  uint32_t                 n_lights;
} AmberFlatScene;

/* rendering::Sensor, sensor.h:34-92 / application.cc:89-94 */
typedef struct {
  uint32_t width, height;
  float    scene_width, scene_height;
} AmberSensor;

typedef struct {
  uint64_t seed;        /* global seed of the per-(pixel,sample) XorShift sampler */
  uint32_t max_depth;   /* 0 = Russian roulette only (reference behaviour, algorithm_pt.cc:137-157) */
  int32_t  device;      /* HIP device ordinal */
  uint32_t row_begin;   /* this handle renders framebuffer rows [row_begin, row_end) -- multi-GPU sharding */
  uint32_t row_end;     /* 0,0 = all rows */
  void*    stream;      /* hipStream_t to launch on; NULL = a NON-BLOCKING stream owned by the handle (it does not synchronise
                           with the legacy default stream: order other work after it with amber_hip_pt_sync or by enqueueing
                           on amber_hip_pt_stream); NULL + AMBER_PT_FLAG_NULL_STREAM in `reserved` = the legacy default stream */
  uint32_t engine;      /* AMBER_ENGINE_* */
  uint32_t stripe_rows; /* 0: every row of [row_begin,row_end).  S > 0: only the rows y with                */
  uint32_t stripe_period; /* (y - row_begin) % stripe_period < S  (interleaved stripes: rank r of N uses     */
  uint32_t reserved;    /* row_begin = r*S, row_end = height, stripe_period = N*S).  Local rows are compact.      */
                        /* `reserved` carries AMBER_PT_FLAG_* bits (0 = none). row_begin == row_end != 0: empty band.   */
} AmberPtParams;
enum {
  AMBER_PT_FLAG_NULL_STREAM = 1u,
  AMBER_PT_FLAG_BVH_POOL = 2u,    /* LAB BUILD ONLY (the product answers AMBER_EINVAL): engine BVH scheduled with the per-wave ray pool
                                     (pt_bvh_pool_kernel).  Same results bit for bit; measured slower on the 1M-sphere scene. */
  AMBER_PT_FLAG_BVH_ITEMS = 4u    /* engine BVH: always pt_bvh_megakernel.  By default a tree of depth <= 12 (scenes of a few hundred
                                     objects) renders with the path-granular kernel and a one-shot per-lane traversal
                                     (pt_megakernel<ENGINE_BVH>): same results bit for bit, faster on shallow trees. */
};

/* All engines are the same persistent work-queue kernel; they differ in how a lane finds its closest hit.
 * Engines 1-4 return the List-semantics answer (closest finite hit, ties to the lower object index: acceleration_list.h:51-68).
 * REFERENCE_BVH returns the hit the reference's command line finds: Cast through the reference's own BVH
 * (acceleration_bvh.h:134-403), which differs from List on distance ties and on hits the reference's traversal loses
 * (INTEGRATION.md section 3). */
enum {
  AMBER_ENGINE_AUTO = 0,       /* TWO_PHASE when the scene has <= 80 objects, BVH otherwise */
  AMBER_ENGINE_LIST = 1,       /* exact test of every object, wave-uniform scan (object data in SGPRs) */
  AMBER_ENGINE_TWO_PHASE = 2,  /* conservative wave-uniform candidate filter, then exact tests of the candidates only; <= 128 objects
                                  (beyond 32 the objects are dealt into groups of 32, one filter program each) */
  AMBER_ENGINE_BVH = 3,        /* host-built flattened 2-wide BVH, per-lane traversal with an LDS stack, exact leaf tests */
  AMBER_ENGINE_WAVEFRONT = 4,  /* LAB BUILD ONLY (the product answers AMBER_EINVAL): streaming formulation -- SoA ray queues in HBM, one launch
                                  per bounce, ballot/prefix-sum compaction; closest hit as AUTO.  Same results; kept to measure that design. */
  /* 5 is reserved (the library's own id of the two-phase engine over groups of 32 objects; create answers AMBER_EINVAL) */
  AMBER_ENGINE_REFERENCE_BVH = 6  /* the reference's own tree, built at create as acceleration_bvh.h:134-312 builds it (same topology, boxes and
                                  object order) and walked per lane in the order of BVH::Node::Cast (:340-403) with the reference's slab test
                                  (aabb.cc:28-62): ties and lost hits exactly as the reference's command line resolves them.  Never chosen by AUTO: 1.1x (1M spheres) to 2x
                                  (the Cornell box) the time of the engine AUTO picks (unquantised 64-byte nodes, large leaves, a stack in global
                                  memory), and the build sorts every node four times like the reference does (1M objects: under 2 s). */
};

#pragma GCC visibility push(hidden)            /* the handle is opaque: its members (and their constructors) are not part of the ABI */
typedef struct amber_hip_pt amber_hip_pt;
#pragma GCC visibility pop

/* Uploads the flattened scene to HBM and allocates the band framebuffer (zeroed). */
int  amber_hip_pt_create(const AmberFlatScene* scene, const AmberSensor* sensor,
                         const AmberPtParams* params, amber_hip_pt** out);
/* Adds, for every pixel of the band, the path measurements of samples [first_sample, first_sample + n_samples)
 * to the device framebuffer (binary32; order: see AMBER_ACCUM_CHUNK).  Asynchronous. */
int  amber_hip_pt_render_pass(amber_hip_pt*, uint32_t first_sample, uint32_t n_samples);
/* Zeroes the device framebuffer and the ray counter (asynchronous). */
int  amber_hip_pt_clear(amber_hip_pt*);
/* Waits for the handle's stream. */
int  amber_hip_pt_sync(amber_hip_pt*);
/* Copies the band framebuffer (the handle's rows in increasing y, width*3 floats per row, RGB sums --
 * NOT divided by the sample count) and the ray count (Scene::Cast calls) to the host.  Synchronises. */
int  amber_hip_pt_download(amber_hip_pt*, float* rgb_sum, uint64_t* ray_count);
/* Device pointer of the band framebuffer (float, rows*width*3) for zero-copy hand-off to RCCL. */
int  amber_hip_pt_device_framebuffer(amber_hip_pt*, void** dptr, uint64_t* n_floats);
/* The hipStream_t every launch and copy of this handle is enqueued on (NULL = the legacy default stream): work that
 * consumes the device framebuffer (an RCCL gather, a peer copy) is ordered after the render by enqueueing it there. */
int  amber_hip_pt_stream(amber_hip_pt*, void** stream);
/* Number of framebuffer rows this handle owns (after striping). */
int  amber_hip_pt_local_rows(amber_hip_pt*, uint32_t* n_rows);
/* Per-launch timing of the dominant kernel, measured with hipEvents on the handle's stream:
 * number of timed launches since create/clear and their total duration. */
int  amber_hip_pt_kernel_time(amber_hip_pt*, uint32_t* n_launches, double* total_ms);
void amber_hip_pt_destroy(amber_hip_pt*);

/* ---- light tracing (rendering::LightTracing, algorithm_lt.cc:112-163) on the same kernels -----------
 * Traces width*height light paths for every pass in [first_sample, first_sample + n_samples) (path i of pass s is seeded
 * by (seed, i, s)) and returns the splats they make on the sensor, sorted in the reference's accumulation order
 * (pass, path, bounce).  value = weight * response / image.Size() is ready to be added to pixel `pixel`.
 * Synchronous.  AMBER_ENOMEM if more than `capacity` splats were produced (n_out then holds the number needed). */
typedef struct { uint32_t path, sample, bounce, pixel; float rgb[3]; uint32_t pad; } AmberSplat;
int amber_hip_lt_trace(amber_hip_pt*, uint32_t first_sample, uint32_t n_samples, AmberSplat* out, uint32_t capacity,
                       uint32_t* n_out, uint64_t* ray_count);
/* The same for the light paths [path_begin, path_end) only: light paths are independent, so the path index shards across
 * devices the way pixels do for path tracing -- every device traces its range of every pass, the caller merges the sorted
 * lists (HipLightTracing::Render; reference: algorithm_lt.cc:82-95 parallelises lt exactly like pt). */
int amber_hip_lt_trace_range(amber_hip_pt*, uint32_t first_sample, uint32_t n_samples, uint32_t path_begin, uint32_t path_end,
                             AmberSplat* out, uint32_t capacity, uint32_t* n_out, uint64_t* ray_count);

const char* amber_hip_last_error(void);
int         amber_hip_abi_version(void);
/* Arithmetic of sin / cos / pow the library was built with: AMBER_MATH_GLIBC (the product) executes glibc 2.35's
 * binary32 sincosf / powf -- the functions the reference calls (sampling.h:249-250, 279) -- operation for operation;
 * AMBER_MATH_PORTABLE (-DAMBER_BUILD_PORTABLE_MATH, measurement builds only) round 1's + - * / forms. */
enum { AMBER_MATH_PORTABLE = 1, AMBER_MATH_GLIBC = 2 };
int         amber_hip_math_mode(void);
int         amber_hip_device_count(void);

/* The known-answer entry points of the tests (amber_hip_kat_*), amber_hip_pt_signatures, engine WAVEFRONT and AMBER_PT_FLAG_BVH_POOL belong to
 * the LAB build, libamber_hip_lab.so: include/amber_hip_lab.h.  libamber_hip.so exports exactly what this header declares (plus the C shim of
 * the host object model, include/amber_host.h, and that model's C++ classes, amber_amd/csrc/amber/). */

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
