// dev_scene.h -- part of pt_device.h (included from there, in order; not a stand-alone header): diagnostic macros of the stamped builds; the device-side scene layout (records in HBM, DevScene).
#pragma once

namespace amber_dev {

// Diagnostic build only (-DAMBER_STAMPS, tools/stamps.py): s_memtime stamps around the sections of one loop
// iteration, summed per wave.  Never compiled into libamber_hip.so; the stamped build's run time is not quoted.
#ifdef AMBER_STAMPS
struct StampCtx { unsigned long long last; unsigned long long acc[8]; };
__device__ __forceinline__ void StampAt(StampCtx* c, int k) {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  c->acc[k] += t - c->last;
  c->last = t;
  __builtin_amdgcn_sched_barrier(0);
}
#define AMBER_STAMP_PARAM , StampCtx* stamp_ctx
#define AMBER_STAMP_ARG , stamp_ctx
#define AMBER_STAMP(k) StampAt(stamp_ctx, k)
// Engine BVH in the same build: lane / wave-trip counters of the divergent loops instead of clocks (tools/bvh_counters.py).
// acc[2k] += 1 on every lane that executes the site, acc[2k+1] += 1 on the first active lane only (wave-level trips).
__device__ __forceinline__ void CountAt(StampCtx* c, int k) {
  if (!c) return;
  const unsigned long long m = __ballot(true);
  c->acc[2 * k] += 1ull;
  if (__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)) == 0u) c->acc[2 * k + 1] += 1ull;
}
#ifdef AMBER_BVH_CLOCKS   /* the same sites as clocks: where a wave's time goes (tools/bvh_counters.py --clocks) */
#define AMBER_COUNT(k)
#define AMBER_CLK(k) do { if (stamp_ctx) StampAt(stamp_ctx, k); } while (0)
#else
#define AMBER_COUNT(k) CountAt(stamp_ctx, k)
#define AMBER_CLK(k)
#endif
/* -DAMBER_COUNT_ALT: slots 2 and 3 count leaf-phase executions and ray swaps instead of rounds and shading calls */
#ifdef AMBER_COUNT_ALT
#define AMBER_COUNT_ROUNDS(k)
#define AMBER_COUNT_LEAVES(k) AMBER_COUNT(k)
#else
#define AMBER_COUNT_ROUNDS(k) AMBER_COUNT(k)
#define AMBER_COUNT_LEAVES(k)
#endif
#define AMBER_STAMP_PARAM_OPT , StampCtx* stamp_ctx = nullptr
#else
#define AMBER_COUNT(k)
#define AMBER_COUNT_ROUNDS(k)
#define AMBER_COUNT_LEAVES(k)
#define AMBER_CLK(k)
#define AMBER_STAMP_PARAM_OPT
#define AMBER_STAMP_PARAM
#define AMBER_STAMP_ARG
#define AMBER_STAMP(k)
#endif

// ---------------------------------------------------------------------------------------------
// device-side scene layout (HBM, read through the scalar cache when the index is wave-uniform)
// ---------------------------------------------------------------------------------------------
struct alignas(16) DevObject {   // 64 B
  float a[3];   uint32_t kind;       // triangle v0 | sphere/disk/cylinder centre
  float e1[3];  float radius;        // triangle E1 = v1 - v0 | disk/cylinder normal
  float e2[3];  float height;        // triangle E2 = v2 - v0
  float n[3];   uint32_t material;   // triangle normal
};
struct alignas(16) DevMaterial {  // 32 B
  uint32_t kind; float rho[3];
  float param; float r0;
  float aux0, aux1;               // per-material constants of the sampling code, computed once on the host with the same binary32
};                                //   operations: Phong 1 / (e + 1) and (e + 2) / (e + 1); Refraction 1 / ior
struct alignas(16) DevBlade {     // 64 B: aperture triangle with explicit vertices (SampleSurfacePoint)
  float v0[3], v1[3], v2[3], n[3];
  int32_t slot;                   // the blade's slot in the two-phase filter program (-1 if it is not a filtered triangle)
  int32_t pad[3];
};
// Phase-A program of the two-phase closest hit (conservative candidate filter): coplanar
// triangles share one plane record, so the plane hit point is computed once per plane.  All wave-uniform, scalar-loaded.
struct alignas(16) DevPlane {       // 32 B: one s_load_dwordx8
  float n[3]; float d0;             // unit normal, n.v0
  float kt;                         // tolerance of the t >= kEPS test   (multiplied by |1/(n.d)|)
  float ktol;                       // barycentric tolerance, max over the plane's triangles (same factor)
  uint32_t n_tris;                  // single-triangle records that follow the pair records; bit 31: same normal vector as the previous plane
  uint32_t n_pairs;                 // parallelogram records: ONE DevTriFilter, TWO consecutive candidate bits
};
// A single triangle: u, v = barycentric coordinates of v1, v2.  A parallelogram pair (two coplanar triangles that
// share an edge, fourth corner D = A + C - B): row A is the coordinate "beta" of the first triangle's unshared
// corner B, row B the coordinate "alpha" of a shared corner; with gamma = 1 - alpha - beta the second triangle's
// coordinates are (1 - gamma, -beta, 1 - alpha), so both minima come from one pair of affine evaluations.
struct alignas(16) DevTriFilter {   // 32 B: one s_load_dwordx8.  Candidate bits follow the program order.
  float c[4][2];                    // (u, v) = c[0] P.x + c[1] P.y + c[2] P.z + c[3] for P on the plane, the two rows interleaved.
};                                  // (Evaluating the pair with three v_pk_fma_f32 on the SGPR pairs was measured: 6 fewer VALU
                                    //  instructions per record, 1.3 % SLOWER -- as the packed forms tried in round 1.)
struct alignas(16) DevSphereFilter { // 32 B
  float c[3]; float r2;
  float ktol; uint32_t pad[3];
};
// Two-phase engine on 33 .. 128 objects (round 5): the objects are dealt into GROUPS of <= 32, each with a Phase-A program of its own over the
// shared record arrays; the closest hit runs group after group (the (t, index) rule makes the order irrelevant).  Group g's objects occupy the
// LDS slots [32 g, 32 g + n_objects).  Group 0 is described by DevScene's own fields as well (pixel_mask_kernel and the 32-object engine read those).
struct alignas(16) DevFilterGroup {  // 48 B, scalar-loaded
  uint32_t plane_first, n_planes, n_simple_planes, tri_first;      // offsets into DevScene.planes / tri_filters (records)
  uint32_t sphere_first, n_sphere_filters, always_mask, n_prog_tris;
  uint32_t n_objects, pad[3];
};
// Flattened 2-wide BVH node (engine BVH, bvh_build.h): both children's (padded) boxes live in the parent.
// Child reference: >= 0 inner node index; < 0 leaf: -(ref+1) = first*8 + count into prim_index (count <= 7).
// The planes are grouped so that the slab test runs on packed FMAs (v_pk_fma_f32): (x, y) pairs of every corner
// against (1/d.x, 1/d.y), and the four z planes in two pairs against (1/d.z, 1/d.z).
struct alignas(16) DevBvhNode {   // 64 B
  float lxy[4];                   // left child:  min.x min.y max.x max.y
  float rxy[4];                   // right child: min.x min.y max.x max.y
  float z[4];                     // left min.z, left max.z, right min.z, right max.z
  int32_t left, right, pad0, pad1;
};
// The node the DEVICE traverses: the same 2-wide tree with both children's boxes quantised to 16 bits per plane
// (plane = bvh_gmin + value * bvh_step, min planes rounded down, max planes up; since round 3 `value` is a binary16 number in
// [-1, 1] around the scene centre -- bvh_build.h, AMBER_BVH_F16 -- before that an integer of a uniform grid), 32 bytes = two
// 16-byte loads per visit instead of four.  The traversal of the 1M-sphere scene is bound by the per-CU rate at which the
// vector L1 looks up the distinct lines a wave's lanes ask for (rocprofv3: TCP busy 92 %, TA busy 70 %, VALU issue 41 %:
// profiles/r02_config3_memory_counters.txt), so bytes -- i.e. load instructions -- per visit are what counts.
// (Uniform grid: step 1 / 65535 of the scene extent, 0.3 % of the smallest sphere of config 3; binary16: 2^-11 of the coordinate.)
struct alignas(16) DevBvhNodeQ {  // 32 B
  uint32_t w[6];                  // 16-bit planes, one word per axis: L.min.x|L.max.x<<16, L.y, L.z, R.x, R.y, R.z (a rotation by 16 swaps entry and exit)
  int32_t left, right;            // >= 0 inner node index; < 0 leaf: -(ref+1) = first*16 + all_triangles*8 + all_spheres*4 + count (count <= 3)
};
// AMBER_BVH_WIDE builds (measurement: VERDICT r01 item 4 asked for a 4-wide tree): the 2-wide tree collapsed to up to four
// children per node, boxes on the same grid, 64 bytes = four 16-byte loads per visit.  An absent child has the reference
// -1 (a leaf of zero objects) and an inverted box.
struct alignas(16) DevBvhNodeQ4 {  // 64 B
  uint32_t w[12];                  // child k, axis a: w[3k + a] = min | max << 16
  int32_t child[4];
};
#ifndef AMBER_BVH_WIDE
#define AMBER_BVH_WIDE 0
#endif
// Light-tracing source record: one per DiffuseLight object, sorted by power (scene/light_set.h:61-82).
struct alignas(16) DevLight {     // 96 B
  uint32_t kind; int32_t slot;      // primitive kind ; filter-program slot of a light triangle (-1 otherwise)
  float cum_power; float pdf_area;  // cumulative power ; Sum(Irradiance) / total power (light_set.h:107-111)
  float irr[3]; float pad0;         // Irradiance = radiance * pi (material_diffuse_light.h:118-125)
  float p[12];                      // triangle v0 v1 v2 normal | sphere centre r | disk centre normal r | cylinder centre normal r h
  float pad1[4];
};
// One splat of a light path onto the sensor (algorithm_lt.cc:141-147)
struct DevSplat { uint32_t path, sample, bounce, pixel; float rgb[3]; uint32_t pad; };   // 32 B
struct DevLens {
  float origin[3];
  float global_[9];
  float local_[9];
  float focus_distance, sensor_distance, p_area;
  float neg_fd_over_sd;       // -focus_distance / sensor_distance (lens_thin.cc:87)
  float neg_sd_over_fd;       // -sensor_distance / focus_distance (lens_thin.cc:118)
  float size_over_area;       // sensor.Size() / sensor.SceneArea() in float (lens_thin.cc:145)
  double sd2;                 // std::pow(sensor_distance_, 2) in double (lens_thin.cc:146)
  uint32_t n_blades;
  float n_blades_f;
  uint32_t kind;              // 0 thin lens, 1 pinhole
  float inv_scene_area;       // 1 / sensor.SceneArea() (lens_pinhole.cc:101)
  float edge_tol;             // barycentric distance from a blade's boundary below which an aperture sample may also lie in ANOTHER blade for the exact
                              // test: 1e-3, or more when the blades are small against the binary32 grid of their world coordinates (>= 0.34: always)
};
// Engine REFERENCE_BVH (ref_bvh_build.h): an inner node of the reference's own tree with BOTH children's boxes as the reference stores them
// (binary32, unpadded); a child reference >= 0 is a node, < 0 is leaf -(reference + 1); a leaf is a run of the sorted object order.
struct alignas(16) DevRefNode { float lmin[3], lmax[3], rmin[3], rmax[3]; int32_t left, right; uint32_t pad[2]; };   // 64 B
struct DevRefLeaf { uint32_t first, count; };
struct DevSensor {
  uint32_t w, h;
  float wf, hf, sw, sh;
  float size_f;               // float(width * height): image.Size() (algorithm_lt.cc:146)
};
struct DevScene {
  const DevObject* __restrict__ objects;
  const DevMaterial* __restrict__ materials;
  const DevBlade* __restrict__ blades;
  const DevPlane* __restrict__ planes;
  const DevTriFilter* __restrict__ tri_filters;
  const DevSphereFilter* __restrict__ sphere_filters;
  uint32_t n_planes, n_sphere_filters;
  uint32_t n_simple_planes;    // the first planes of the program, an even number: slabs of two parallel planes with one pair record each
  uint32_t always_mask;        // program slots that are always candidates (disks, cylinders, degenerate triangles)
  uint32_t blade_mask;         // program slots of the aperture blades (primary rays: decided per ray, not per pixel)
  uint32_t n_prog_tris;        // program slots [0, n_prog_tris) are filtered triangles, then spheres, then the rest
  const DevObject* __restrict__ prog_objects;   // objects in program order, kind |= scene index << 8 (| 0x80: a filtered triangle) (staged to LDS)
  const DevFilterGroup* __restrict__ groups;    // engine TWO_PHASE_N: n_groups records (null otherwise)
  uint32_t n_groups;                            // 1 for the 32-object engine
  uint32_t n_lds_objects;                       // records of prog_objects (n_objects, or 32 * n_groups)
  const DevBvhNodeQ* __restrict__ bvh_nodes;    // engine BVH: quantised 2-wide nodes
  const DevBvhNodeQ4* __restrict__ bvh_nodes4;  // AMBER_BVH_WIDE builds: the collapsed 4-wide nodes (else null)
  float bvh_gmin[3], bvh_step[3];               // plane = bvh_gmin + value * bvh_step (binary16 planes: scene centre, half extent)
  float bvh_reach[3];                           // max(|bounds_min - x|, |bounds_max - x|) over x in the bounds, per axis = extent (slab rounding slack)
  const float4* __restrict__ bvh_spheres;       // (centre, radius) of every object in leaf order (zeros for non-spheres): leaves of spheres only test from here
  const float4* __restrict__ bvh_tris;          // three float4 per object in leaf order, {A.xyz E1.x} {E1.yz E2.xy} {E2.z, scene index, 0, 0} (zeros for non-triangles): leaves of triangles only
  const uint32_t* __restrict__ bvh_prims;       // leaf order -> object index
  const DevObject* __restrict__ bvh_objects;    // the object records in leaf order (HitRec.slot of engine BVH indexes this array)
  int32_t bvh_root;                             // child reference of the whole scene
  float fp_center[3];                           // two-phase filter: rays whose origin is farther than fp_reach (max norm) from here,
  float fp_reach;                               // or with |d| > 2, bypass the filter (all objects become candidates)
  float fp_tmax;                                // no ray of the model hits anything beyond t = fp_tmax / |d| (Phase-A distance pruning)
  float bvh_center[3];                          // centre and half diagonal of the scene bounds (per-ray box margin, BvhBegin)
  float bvh_half_diag;
  float bvh_inv_rmin;                           // 1 / smallest sphere radius ; 0 when the scene has no spheres
  const DevLight* __restrict__ lights;          // light tracing
  uint32_t n_lights;
  float total_power;
  uint32_t n_objects;
  uint32_t max_depth;
  const DevLens* __restrict__ lens;              // device memory, read with LoadLens() where a path starts / ends
  DevSensor sensor;
  // engine REFERENCE_BVH (the last fields: the other engines' kernels never read them).  It also uses bvh_objects / bvh_prims (the objects in the
  // order the reference's build leaves them in) and bvh_root (a DevRefNode index, or a leaf)
  const DevRefNode* __restrict__ ref_nodes;
  const DevRefLeaf* __restrict__ ref_leaves;
  uint2* ref_stack;                              // traversal stack, [level][thread of the grid]: {child reference, bits of the child's entry distance}
  uint32_t ref_stack_stride;                     // threads of the grid the stack was allocated for
};

#define AMBER_PHONG_MAX_TRIES 1024
enum { PRIM_TRIANGLE = 0, PRIM_SPHERE = 1, PRIM_DISK = 2, PRIM_CYLINDER = 3 };
enum { MAT_LAMBERTIAN = 0, MAT_PHONG = 1, MAT_SPECULAR = 2, MAT_REFRACTION = 3, MAT_DIFFUSE_LIGHT = 4, MAT_EYE = 5 };

}  // namespace amber_dev
