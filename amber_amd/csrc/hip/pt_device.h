// pt_device.h -- gfx950 device functions of the path-tracing engine.
//
// Arithmetic contract (DESIGN.md "Numerics"): every float operation below is a single IEEE
// binary32 operation in the order the reference performs it; the translation unit is built with
// -ffp-contract=off (the reference's x86 build has no FMA) and hipcc's default correctly rounded
// f32 divide/sqrt and un-flushed subnormals.  No OCML math call is on the path: sin / cos / pow
// execute glibc 2.35's binary32 sincosf / powf -- double-precision kernels, fused multiply-adds
// where the x86-64 FMA variant has them, one rounding to binary32 at the end -- operation for
// operation (section "sin / cos / pow" below), so the CPU oracle and the live libm the reference
// calls produce the same bits.  (-DAMBER_BUILD_PORTABLE_MATH builds round 1's own + - * / forms
// instead; a measurement build, never the product.)
//
// Reference citations are relative to /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// ---------------------------------------------------------------------------------------------
// Vector3 (include/amber/prelude/vector3.h:36-342): component-wise ops, scalar splat
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
// Wave-uniform reads of scene records go through the constant address space: hipcc then emits scalar loads
// (s_load_dwordx4/x8/x16, operands in SGPRs) even though the kernel also stores to global memory in its loop.
typedef const uint32_t __attribute__((address_space(4)))* ConstWords;
typedef float F2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cw_f(ConstWords w, int k) { return __uint_as_float(w[k]); }
__device__ __forceinline__ F2 cw_f2(ConstWords w, int k) { return F2{cw_f(w, k), cw_f(w, k + 1)}; }
__device__ __forceinline__ V3 cw_v3(ConstWords w, int k) { return V3{cw_f(w, k), cw_f(w, k + 1), cw_f(w, k + 2)}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 v) { return V3{s * v.x, s * v.y, s * v.z}; }
__device__ __forceinline__ V3 operator*(V3 v, float s) { return V3{v.x * s, v.y * s, v.z * s}; }
__device__ __forceinline__ V3 operator/(V3 v, float s) { return V3{v.x / s, v.y / s, v.z / s}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float Dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
__device__ __forceinline__ float SquaredLength(V3 v) { return Dot(v, v); }
// __builtin_sqrtf lowers to the correctly rounded sequence (v_sqrt_f32 + one-ulp fix-up); __fsqrt_rn does NOT on ROCm 7.2
__device__ __forceinline__ float Sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ V3 Normalize(V3 v) { const float l = Sqrt(SquaredLength(v)); return V3{v.x / l, v.y / l, v.z / l}; }
__device__ __forceinline__ V3 Cross(V3 u, V3 v) {
  return V3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
// std::max({x,y,z}) (vector3.h:276-281): first element wins unless a later one is strictly greater
__device__ __forceinline__ float Max3(V3 v) { float m = v.x; if (m < v.y) m = v.y; if (m < v.z) m = v.z; return m; }
__device__ __forceinline__ float Abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ void OrthonormalBasis(V3 w, V3& u, V3& v) {      // vector3.h:330-342
  const bool xs = Abs(w.x) < Abs(w.y);
  u = Normalize(Cross(w, xs ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f)));
  v = Normalize(Cross(w, u));
}
__device__ __forceinline__ V3 MatMul(const float* e, V3 v) {               // matrix3.h:101-109
  return v3(e[0] * v.x + e[1] * v.y + e[2] * v.z, e[3] * v.x + e[4] * v.y + e[5] * v.z, e[6] * v.x + e[7] * v.y + e[8] * v.z);
}
__device__ __forceinline__ bool IsFinite(float x) { return Abs(x) < __builtin_inff(); }   // false for NaN and inf

// ---------------------------------------------------------------------------------------------
// per-(pixel,sample) XorShift sampler (DESIGN.md "Sampler"): splitmix64-hashed seed, Marsaglia
// xorshift64 (13,7,17), uniform = top 24 bits * 2^-24 -- exact in binary32, never 1.0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t SplitMix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t XorShiftSeed(uint64_t hashed_global_seed, uint32_t pixel, uint32_t sample) {
  const uint64_t key = (static_cast<uint64_t>(pixel) << 32) | sample;
  const uint64_t s = SplitMix64(hashed_global_seed ^ key);
  return s ? s : 0x9E3779B97F4A7C15ull;
}
__device__ __forceinline__ float Uniform(uint64_t& s) {
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return static_cast<float>(static_cast<uint32_t>(s >> 40)) * 0x1p-24f;
}

// ---------------------------------------------------------------------------------------------
// sin / cos / pow.  The reference calls glibc: g++ -O2 merges std::cos(phi), std::sin(phi) (sampling.h:249-250, 283-284)
// into ONE sincosf call, std::pow(r0, 1 / (e + 1)) (sampling.h:279) is powf.  glibc 2.35's binary32 functions are
// double-precision kernels (the algorithms and tables of ARM's optimized-routines: s_sincosf.c, sincosf_poly.h, e_powf.c,
// e_powf_log2_data.c, e_exp2f_data.c) whose x86-64 FMA variant -- the one every FMA-capable host selects -- fuses every
// a*b+c of the source and rounds to binary32 once at the end.  gfx950 has IEEE binary64 mul/fma, so the engine executes
// exactly those operations: bit-identical to the live libm for every argument the path can produce (the oracle's
// GLIBC mode is the same restatement, proven equal to libm.so.6 over the whole argument set in tests/test_math_modes.py).
// Domain of SinCos: |x| < 120 (the path needs [0, 2 pi]); NaN beyond.
// -DAMBER_BUILD_PORTABLE_MATH builds round 1's + - * / forms instead (Cephes-style; 29 % of the 2^24 possible phi differ from
// glibc in the last bit) -- kept only to measure the distance between the two.
// ---------------------------------------------------------------------------------------------
#ifdef AMBER_BUILD_PORTABLE_MATH
#define AMBER_MATH_MODE 1
__device__ __forceinline__ void SinCos(float x, float& s_out, float& c_out) {
  const float FOPI = 1.27323954473516f;
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  int j = static_cast<int>(FOPI * x);
  j += (j & 1);
  const float y = static_cast<float>(j);
  const float r = ((x - y * DP1) - y * DP2) - y * DP3;
  const float z = r * r;
  const float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * r + r;
  const float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
                   - 0.5f * z + 1.0f;
  const int q = (j >> 1) & 3;
  const float s = (q & 1) ? pc : ps;
  const float c = (q & 1) ? ps : pc;
  s_out = (q & 2) ? -s : s;
  c_out = (q == 1 || q == 2) ? -c : c;
}
__device__ __forceinline__ float Pow(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
  if (x == 1.0f) return 1.0f;
  uint32_t bits = __float_as_uint(x);
  int e = static_cast<int>((bits >> 23) & 0xff);
  if (e == 0) { x = x * 16777216.0f; bits = __float_as_uint(x); e = static_cast<int>((bits >> 23) & 0xff) - 24; }
  e -= 127;
  float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.0f;
  const float s = f / (2.0f + f);
  const float z = s * s;
  float p = 0.0909090909f;
  p = p * z + 0.111111111f;
  p = p * z + 0.142857143f;
  p = p * z + 0.2f;
  p = p * z + 0.333333333f;
  p = p * z + 1.0f;
  const float ln_m = 2.0f * s * p;
  const float log2x = static_cast<float>(e) + ln_m * 1.44269504f;
  const float w = y * log2x;
  if (w >= 128.0f) return __builtin_inff();
  if (w < -149.0f) return 0.0f;
  const float nf = __builtin_floorf(w + 0.5f);
  const float g = w - nf;
  const float t = g * 0.693147181f;
  float q = 1.98412698e-4f;
  q = q * t + 1.38888889e-3f;
  q = q * t + 8.33333333e-3f;
  q = q * t + 4.16666667e-2f;
  q = q * t + 1.66666667e-1f;
  q = q * t + 0.5f;
  q = q * t + 1.0f;
  q = q * t + 1.0f;
  int n = static_cast<int>(nf);
  if (n < -126) { q = q * __uint_as_float(static_cast<uint32_t>(n + 126 + 127) << 23); n = -126; }
  return q * __uint_as_float(static_cast<uint32_t>(n + 127) << 23);
}
#else
#define AMBER_MATH_MODE 2
__device__ const double kGlibcLog2Tab[16][2] = {   // __powf_log2_data.tab: {invc, logc}
  {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
  {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
  {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
  {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
  {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
  {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
  {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
__device__ const uint64_t kGlibcExp2Tab[32] = {    // __exp2f_data.tab
  0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
  0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
  0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
  0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
  0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
  0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
// __sincosf (s_sincosf.c) for |y| < 120.  The source's two short paths are folded into reduce_fast, which computes the
// same values there: for |y| < 0.75 it finds n = 0 and x - 0 * hpi = x exactly.  Its second coefficient table (n & 2)
// holds the negated cosine coefficients and sign[] = {1, -1, -1, 1} flips x: round-to-nearest is symmetric, so the
// first table's results with the signs applied afterwards are the same bits.
__device__ __forceinline__ void SinCos(float y, float& s_out, float& c_out) {
  const double x0 = static_cast<double>(y);
  const double r = x0 * 0x1.45f306dc9c883p+23;                                  // hpi_inv = 2^24 * 2 / pi
  const int n = (static_cast<int>(r) + 0x800000) >> 24;                          // nearest multiple of pi / 2
  const double x = __builtin_fma(-static_cast<double>(n), 0x1.921fb54442d18p+0, x0);
  const double x2 = x * x;
  const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);    // sincosf_poly (sysdeps/x86_64/fpu/sincosf_poly.h)
  const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);
  const double x3 = x2 * x, x4 = x2 * x2;
  const double x5 = x2 * x3, x6 = x2 * x4;
  const double s = __builtin_fma(x3, -0x1.555545995a603p-3, x);
  const double c = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
  float sv = static_cast<float>(__builtin_fma(x5, s1, s));
  float cv = static_cast<float>(__builtin_fma(x6, c2, c));
  if ((n + 1) & 2) sv = -sv;                                                     // sign[n & 3]
  if (n & 2) cv = -cv;                                                           // table[1]
  const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;                     // abstop12
  if (top < 0x398u) { sv = y; cv = 1.0f; }                                       // |y| < 2^-12: sin = y, cos = 1 (n = 0)
  s_out = (n & 1) ? cv : sv;
  c_out = (n & 1) ? sv : cv;
  if (!(top < 0x42fu)) { s_out = __builtin_nanf(""); c_out = __builtin_nanf(""); }   // |y| >= 120, inf, NaN: outside the restated domain
}
// __powf (e_powf.c) for x >= +0 and finite y (CosinePower's r0^(1/(e+1)): r0 in [0, 1), y > 0); negative or
// non-finite arguments follow glibc's special cases.
__device__ __forceinline__ int GlibcCheckInt(uint32_t iy) {
  const int e = static_cast<int>(iy >> 23 & 0xffu);
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
// A double constant that is the ADDEND of an fma must sit in a VGPR pair (one SGPR operand per VOP3 instruction).  Left to
// itself the compiler materialises such constants once, outside the persistent loop, and -- under the kernels' register
// caps -- spills them to scratch, reloading them on every evaluation (pt_megakernel: three 8-byte scratch loads per Pow).
// Passing the constant through an empty asm next to its use makes it two v_mov instead.
#define AMBER_NEAR_CONSTANT(bits64_) NearConstantBits<static_cast<uint32_t>((bits64_) & 0xffffffffull), static_cast<uint32_t>((bits64_) >> 32)>()
template <uint32_t kLo, uint32_t kHi>
__device__ __forceinline__ double NearConstantBits() {
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "n"(kLo), "n"(kHi));
  return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
}
__device__ __forceinline__ float Pow(float x, float y) {
  uint32_t sign_bias = 0u;
  uint32_t ix = __float_as_uint(x);
  const uint32_t iy = __float_as_uint(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
    if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {                               // y is 0, inf or NaN
      if (2u * iy == 0u) return 1.0f;
      if (ix == 0x3f800000u) return 1.0f;
      if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
      if (2u * ix == 2u * 0x3f800000u) return 1.0f;
      if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
      return y * y;
    }
    if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {                               // x is 0, inf or NaN
      float x2 = x * x;
      if ((ix & 0x80000000u) && GlibcCheckInt(iy) == 1) x2 = -x2;
      return (iy & 0x80000000u) ? 1.0f / x2 : x2;
    }
    if (ix & 0x80000000u) {                                                     // finite x < 0
      const int yint = GlibcCheckInt(iy);
      if (yint == 0) return __builtin_nanf("");
      if (yint == 1) sign_bias = 1u << 16;
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) { ix = __float_as_uint(__uint_as_float(ix) * 0x1p23f); ix &= 0x7fffffffu; ix -= 23u << 23; }   // subnormal x
  }
  // log2_inline: x = 2^k z, z in [OFF, 2 OFF), one of 16 subintervals with centre c; log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const uint32_t top = tmp & 0xff800000u;
  const double z = static_cast<double>(__uint_as_float(ix - top));
  const int k = static_cast<int32_t>(top) >> 23;
  const double2 t_log = *reinterpret_cast<const double2*>(kGlibcLog2Tab[i]);
  const double r = __builtin_fma(z, t_log.x, -1.0);
  const double y0 = t_log.y + static_cast<double>(k);
  const double r2 = r * r;
  double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, AMBER_NEAR_CONSTANT(0xbfd71969a075c67aull) /* -0x1.71969a075c67ap-2 */);
  const double pp = __builtin_fma(0x1.ec70a6ca7baddp-2, r, AMBER_NEAR_CONSTANT(0xbfe7154748bef6c8ull) /* -0x1.7154748bef6c8p-1 */);
  const double r4 = r2 * r2;
  double q = __builtin_fma(0x1.71547652ab82bp0, r, y0);
  q = __builtin_fma(pp, r2, q);
  yy = __builtin_fma(yy, r4, q);
  const double ylogx = static_cast<double>(y) * yy;
  if ((static_cast<uint64_t>(__double_as_longlong(ylogx)) >> 47 & 0xffffull) >= (0x405f800000000000ull >> 47)) {   // |y log2 x| >= 126
    const float sgn = sign_bias ? -1.0f : 1.0f;
    if (ylogx > 0x1.fffffffd1d571p+6) return sgn * __builtin_inff();
    if (ylogx <= -150.0) return sgn * 0.0f;
    if (ylogx < -149.0) return sgn * 0x1p-149f;
  }
  // exp2_inline: x = k/32 + r, 2^x = 2^(k/32) * 2^r
  const double shift = 0x1.8p+52 / 32;
  double kd = ylogx + shift;
  const uint64_t ki = static_cast<uint64_t>(__double_as_longlong(kd));
  kd -= shift;
  const double rr = ylogx - kd;
  uint64_t t = kGlibcExp2Tab[ki & 31ull];
  t += (ki + sign_bias) << 47;
  const double sc = __longlong_as_double(static_cast<long long>(t));
  const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, AMBER_NEAR_CONSTANT(0x3fcebfce50fac4f3ull) /* 0x1.ebfce50fac4f3p-3 */);
  const double rr2 = rr * rr;
  double e = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
  e = __builtin_fma(zz, rr2, e);
  return static_cast<float>(e * sc);
}
#endif
// std::pow(float, int) promotes to double (C++11): glibc's double pow.  x*x is exact (24 + 24 bits), so (x*x)^2 is the
// correctly rounded x^4; x^5 is formed from the exact (hi, lo) pair of x^4 and rounds once too (portable build: two
// roundings, as round 1).  glibc's pow is not correctly rounded: it differs from these in the last bit of the DOUBLE for
// 1e-3 of the arguments, which never survived the conversion to binary32 in 2e8 trials.
__device__ __forceinline__ double Pow4(float x) { const double d = x; const double d2 = d * d; return d2 * d2; }
#ifdef AMBER_BUILD_PORTABLE_MATH
__device__ __forceinline__ double Pow5(float x) { const double d = x; const double d2 = d * d; return (d2 * d2) * d; }
#else
__device__ __forceinline__ double Pow5(float x) {
  const double d = x, d2 = d * d;
  const double h = d2 * d2, l = __builtin_fma(d2, d2, -h);
  const double p = h * d, pl = __builtin_fma(h, d, -p);
  return p + __builtin_fma(l, d, pl);
}
#endif

// ---------------------------------------------------------------------------------------------
// closest hit -- List semantics (acceleration_list.h:51-68): scan objects in insertion order,
// keep a hit iff it is finite and STRICTLY closer.  The object index is wave-uniform, so object
// data arrives through scalar loads and sits in SGPRs.
// ---------------------------------------------------------------------------------------------
#define AMBER_KEPS 1e-6f   // (t < kEPS) <=> (t <= 1e-6f) ; (t > kEPS) <=> (t > 1e-6f): 1e-6f < 1e-6L < nextafterf(1e-6f)

struct HitRec { float t, u, v; int idx; int slot; };   // idx: object index (scene order); slot: index into the array the engine scans

// algebra.h:31-52
__device__ __forceinline__ bool SolveQuadratic(float a, float b, float c, float& alpha, float& beta) {
  const float d = b * b - 4.0f * a * c;
  if (d < 0.0f) return false;
  const float sqrt_d = Sqrt(d);
  alpha = -b - sqrt_d;
  beta = -b + sqrt_d;
  if (Abs(alpha) < Abs(beta)) { alpha = c / beta * 2.0f; beta = beta / (2.0f * a); }
  else { beta = c / alpha * 2.0f; alpha = alpha / (2.0f * a); }
  return true;
}

// A hit replaces the best iff it is finite and strictly closer; kTie additionally lets an equally distant hit
// of a LOWER object index win, which makes the result independent of the order in which candidates are visited
// (the List scan visits indices in ascending order, where strict < alone already gives the lower index).
template <bool kTie>
__device__ __forceinline__ bool Closer(float t, int i, const HitRec& best) {
  if (!IsFinite(t)) return false;
  if (t < best.t) return true;
  return kTie && t == best.t && i < best.idx;
}

template <bool kTie>
__device__ __forceinline__ void IntersectTriangle(V3 A, V3 E1, V3 E2, int i, int slot, V3 o, V3 d, HitRec& best) {   // primitive_triangle.cc:97-128
  const V3 P = Cross(d, E2);
  const float det = Dot(P, E1);
  const V3 T = o - A;
  const float u = Dot(P, T) / det;
  if (!(u > 1.0f || u < 0.0f)) {
    const V3 Q = Cross(T, E1);
    const float v = Dot(Q, d) / det;
    if (!(v > 1.0f || v < 0.0f) && !(u + v > 1.0f)) {
      const float t = Dot(Q, E2) / det;
      if (!(t <= AMBER_KEPS) && Closer<kTie>(t, i, best)) { best.t = t; best.u = u; best.v = v; best.idx = i; best.slot = slot; }
    }
  }
}
template <bool kTie>
__device__ __forceinline__ void IntersectSphere(V3 A, float radius, int i, int slot, V3 o, V3 d, HitRec& best) {      // primitive_sphere.cc:75-107
  const V3 co = A - o;
  const float b = -2.0f * Dot(co, d);
  const float c = SquaredLength(co) - radius * radius;
  float alpha, beta;
  if (SolveQuadratic(1.0f, b, c, alpha, beta)) {
    float t;
    bool ok = true;
    if (alpha > AMBER_KEPS) t = alpha; else if (beta > AMBER_KEPS) t = beta; else { ok = false; t = 0.f; }
    if (ok && Closer<kTie>(t, i, best)) { best.t = t; best.idx = i; best.slot = slot; }
  }
}
template <bool kTie>
__device__ __forceinline__ void IntersectDisk(V3 A, V3 N, float radius, int i, int slot, V3 o, V3 d, HitRec& best) {  // primitive_disk.cc:94-114
  const float cos_theta = Dot(d, N);
  if (!(cos_theta == 0.0f)) {
    const float t = Dot(A - o, N) / cos_theta;
    if (!(t <= AMBER_KEPS)) {
      const float sq = SquaredLength(o + t * d - A);
      if (!(sq > radius * radius) && Closer<kTie>(t, i, best)) { best.t = t; best.idx = i; best.slot = slot; }
    }
  }
}
template <bool kTie>
__device__ __forceinline__ void IntersectCylinder(V3 A, V3 N, float radius, float height, int i, int slot, V3 o, V3 d, HitRec& best) {   // primitive_cylinder.cc:100-142
  const V3 OC = A - o;
  const V3 uu = d - Dot(d, N) * N;
  const V3 vv = OC - Dot(OC, N) * N;
  const float a = SquaredLength(uu);
  const float b = -2.0f * Dot(uu, vv);
  const float c = SquaredLength(vv) - radius * radius;
  float alpha, beta;
  if (SolveQuadratic(a, b, c, alpha, beta)) {
    bool ok = false; float t = 0.f;
    if (alpha > AMBER_KEPS) {
      const float h = Dot(alpha * d - OC, N);
      if (h >= 0.0f && h <= height) { ok = true; t = alpha; }
    }
    if (!ok && beta > AMBER_KEPS) {
      const float h = Dot(beta * d - OC, N);
      if (h >= 0.0f && h <= height) { ok = true; t = beta; }
    }
    if (ok && Closer<kTie>(t, i, best)) { best.t = t; best.idx = i; best.slot = slot; }
  }
}

template <bool kTie>
__device__ __forceinline__ void IntersectObject(const DevObject& ob, uint32_t kind, int i, int slot, V3 o, V3 d, HitRec& best) {
  const V3 A = ld3(ob.a);
  if (kind == PRIM_TRIANGLE) IntersectTriangle<kTie>(A, ld3(ob.e1), ld3(ob.e2), i, slot, o, d, best);
  else if (kind == PRIM_SPHERE) IntersectSphere<kTie>(A, ob.radius, i, slot, o, d, best);
  else if (kind == PRIM_DISK) IntersectDisk<kTie>(A, ld3(ob.e1), ob.radius, i, slot, o, d, best);
  else IntersectCylinder<kTie>(A, ld3(ob.e1), ob.radius, ob.height, i, slot, o, d, best);
}

// Engine LIST: every object, exact test, wave-uniform index (object data in SGPRs).
__device__ __forceinline__ void ClosestHitList(const DevScene& sc, V3 o, V3 d, HitRec& best) {
  best.t = 3.402823466e+38f; best.u = 0.f; best.v = 0.f; best.idx = -1; best.slot = -1;   // Acceleration::Cast(ray, FLT_MAX)
  const int n = static_cast<int>(sc.n_objects);
  const ConstWords base = (ConstWords)(sc.objects);
  for (int i = 0; i < n; ++i) {
    const ConstWords w = base + i * 16;                 // DevObject = 16 dwords
    const uint32_t kind = w[3];
    const V3 A = cw_v3(w, 0);
    if (kind == PRIM_TRIANGLE) IntersectTriangle<false>(A, cw_v3(w, 4), cw_v3(w, 8), i, i, o, d, best);
    else if (kind == PRIM_SPHERE) IntersectSphere<false>(A, cw_f(w, 7), i, i, o, d, best);
    else if (kind == PRIM_DISK) IntersectDisk<false>(A, cw_v3(w, 4), cw_f(w, 7), i, i, o, d, best);
    else IntersectCylinder<false>(A, cw_v3(w, 4), cw_f(w, 7), cw_f(w, 11), i, i, o, d, best);
  }
}

// Engine TWO-PHASE (n_objects <= 32).
//  Phase A, all lanes, wave-uniform object index, approximate arithmetic (FMA, v_rcp): a CONSERVATIVE filter that
//  sets bit i of a per-lane mask unless object i certainly cannot pass the reference's exact test.  For a triangle:
//  intersect the supporting plane, evaluate the barycentrics of the plane point as affine functions, and keep the
//  object when they are inside [0,1] up to a tolerance that scales with 1/|n.d| (the conditioning of the
//  reference's own arithmetic), when the ray is nearly parallel to the plane, or when anything is NaN.
//  Phase B, per lane: the exact reference-arithmetic test (the same functions as engine LIST) on the candidates
//  only, object records gathered from the LDS copy; the (t, index) tie rule makes the visiting order irrelevant.
//  The result is identical to ClosestHitList (tests: full-image and per-ray equality of both engines).
#define AMBER_GRAZING 1e-3f
#ifndef AMBER_SHARED_WEIGHT_QUOTIENT
#define AMBER_SHARED_WEIGHT_QUOTIENT 1
#endif
// The Phase-A program one call works through: the whole scene's (the 32-object engine: DevScene's own fields, compile-time `first`) or one group's.
struct FilterView {
  ConstWords planes, tris, spheres;
  int n_planes, n_simple_planes, n_sphere_filters;
  uint32_t always_mask, n_prog_tris, n_objects;
  int slot_base;                                             // LDS slot of the view's first object
};
__device__ __forceinline__ FilterView SceneView(const DevScene& sc) {
  return FilterView{(ConstWords)(sc.planes), (ConstWords)(sc.tri_filters), (ConstWords)(sc.sphere_filters), static_cast<int>(sc.n_planes), static_cast<int>(sc.n_simple_planes),
                    static_cast<int>(sc.n_sphere_filters), sc.always_mask, sc.n_prog_tris, sc.n_objects, 0};
}
template <bool kMulti, bool kFirst>
__device__ __forceinline__ void ClosestHitTwoPhaseView(const DevScene& sc, const FilterView fv, const DevObject* lds_objects, V3 o_world, V3 d, int origin_slot, HitRec& best AMBER_STAMP_PARAM,
                                                       const bool use_premask, const uint32_t premask);
__device__ __forceinline__ void ClosestHitTwoPhase(const DevScene& sc, const DevObject* lds_objects, V3 o_world, V3 d, int origin_slot, HitRec& best AMBER_STAMP_PARAM,
                                                   const bool use_premask = false, const uint32_t premask = 0u) {
  ClosestHitTwoPhaseView<false, true>(sc, SceneView(sc), lds_objects, o_world, d, origin_slot, best AMBER_STAMP_ARG, use_premask, premask);
}
// Engine TWO_PHASE_N: every group in turn; the primary rounds' pixel masks describe group 0 only (the other groups run Phase A for eye rays too).
__device__ __forceinline__ void ClosestHitTwoPhaseGroups(const DevScene& sc, const DevObject* lds_objects, V3 o_world, V3 d, int origin_slot, HitRec& best AMBER_STAMP_PARAM,
                                                         const bool use_premask = false, const uint32_t premask = 0u) {
  ClosestHitTwoPhaseView<true, true>(sc, SceneView(sc), lds_objects, o_world, d, origin_slot < 32 ? origin_slot : -1, best AMBER_STAMP_ARG, use_premask, premask);
  const int n_groups = static_cast<int>(sc.n_groups);
  ConstWords gw = (ConstWords)(sc.groups);
  for (int g = 1; g < n_groups; ++g) {
    ConstWords w = gw + g * 12;                               // DevFilterGroup = 12 dwords
    const FilterView fv{(ConstWords)(sc.planes) + w[0] * 8u, (ConstWords)(sc.tri_filters) + w[3] * 8u, (ConstWords)(sc.sphere_filters) + w[4] * 8u,
                        static_cast<int>(w[1]), static_cast<int>(w[2]), static_cast<int>(w[5]), w[6], w[7], w[8], g * 32};
    ClosestHitTwoPhaseView<true, false>(sc, fv, lds_objects, o_world, d, origin_slot, best AMBER_STAMP_ARG, false, 0u);
  }
}
// kMulti = false: the 32-object engine -- the code of rounds 2-4, operand for operand (slot base 0, the kind byte unmasked: a 1.2 % slower config-2 kernel was
// the price of sharing ONE instantiation with the grouped engine, tools/ab_lib.py across the round's commits).
template <bool kMulti, bool kFirst>
__device__ __forceinline__ void ClosestHitTwoPhaseView(const DevScene& sc, const FilterView fv, const DevObject* lds_objects_all, V3 o_world, V3 d, int origin_slot_all, HitRec& best AMBER_STAMP_PARAM,
                                                       const bool use_premask, const uint32_t premask) {
  if (kFirst) { best.t = 3.402823466e+38f; best.u = 0.f; best.v = 0.f; best.idx = -1; best.slot = -1; }
  const int slot_base = kMulti ? fv.slot_base : 0;
  const DevObject* lds_objects = kMulti ? lds_objects_all + slot_base : lds_objects_all;
  const int origin_slot = kFirst ? origin_slot_all : ((origin_slot_all >= slot_base && origin_slot_all < slot_base + 32) ? origin_slot_all - slot_base : -1);
  constexpr uint32_t kKindMask = kMulti ? 0x7fu : 0xffu;       // the grouped engine's LDS records flag filtered triangles in bit 7 of `kind`
  uint32_t cand = kMulti ? fv.always_mask : sc.always_mask;      // (kMulti = false reads the scene record where rounds 2-4 read it: nothing of `fv` is live)
  // use_premask (WAVE-UNIFORM): the candidates are already known -- a primary round of pt_megakernel, whose 64 eye rays take them
  // from their pixel's mask (pixel_mask_kernel: every object some ray of the pixel's beam can hit, computed once per handle) --
  // so Phase A, a third of the kernel, is skipped for the ray that every path starts with.
  if (use_premask) cand |= premask;
  else {
    const V3 o = v3(o_world.x - sc.fp_center[0], o_world.y - sc.fp_center[1], o_world.z - sc.fp_center[2]);   // Phase A runs in centred coordinates (filter_build.h)
    // Pruning by distance.  A triangle that the filter finds hit with a MARGIN -- inside by the same tolerance that
    // otherwise widens it, beyond kEPS by the distance tolerance, not grazing -- is certain to pass the reference's exact
    // test at a distance below t' + tolerance; `t_upper` is the least such bound seen so far, and a later triangle whose
    // distance is certainly larger (t' - tolerance > t_upper) cannot be the closest hit and is not made a candidate.
    // One-sided (nothing is remembered per object), so the program evaluates likely occluders first (filter_build.h).
    // t_upper starts at the largest distance at which a ray of the model (origin and objects inside the model box,
    // checked below) can hit anything: that bound also removes the planes a ray runs nearly PARALLEL to -- their plane
    // point lies hundreds of scene sizes away -- which used to be kept with all their triangles whatever the distance
    // (1e-3 of the rays per plane: one lane in most waves, and every lane of the wave waited for its extra exact tests).
    // Every comparison is false on NaN: nothing is pruned and nothing is certain.
    float t_upper = sc.fp_tmax * __builtin_amdgcn_rsqf(d.x * d.x + d.y * d.y + d.z * d.z);
    if (!(t_upper == t_upper)) t_upper = 3.402823466e+38f;
    ConstWords pl = kMulti ? fv.planes : (ConstWords)(sc.planes);
    ConstWords tr = kMulti ? fv.tris : (ConstWords)(sc.tri_filters);
    const int n_planes = kMulti ? fv.n_planes : static_cast<int>(sc.n_planes);
    uint32_t bit = 1u;                                       // candidate bit of the next program triangle (SALU)
    float nd = 0.f, no = 0.f, rc = 0.f;
#define AMBER_PLANE_NORMAL() \
        nd = __builtin_fmaf(cw_f(pl, 0), d.x, __builtin_fmaf(cw_f(pl, 1), d.y, cw_f(pl, 2) * d.z)); \
        no = __builtin_fmaf(cw_f(pl, 0), o.x, __builtin_fmaf(cw_f(pl, 1), o.y, cw_f(pl, 2) * o.z)); \
        rc = __builtin_amdgcn_rcpf(nd);
#define AMBER_PLANE_SETUP() \
      const float tp = (cw_f(pl, 3) - no) * rc; \
      const float rho = Abs(rc); \
      const float Px = __builtin_fmaf(tp, d.x, o.x), Py = __builtin_fmaf(tp, d.y, o.y), Pz = __builtin_fmaf(tp, d.z, o.z); \
      const float kr = cw_f(pl, 4) * rho;                   /* distance tolerance of this plane for this ray */ \
      const bool t_ok = (tp >= AMBER_KEPS - kr) && !(tp - kr > t_upper);                  /* beyond kEPS, and not certainly behind a certain hit */ \
      /* nearly parallel: the in-plane coordinates are not trusted (every triangle of the plane stays a candidate), the */ \
      /* distance still is, down to |n.d| = 1e-6; below that, or NaN, everything is kept */ \
      const bool degenerate = !(Abs(nd) >= 1e-6f); \
      const bool grazing = !(Abs(nd) >= AMBER_GRAZING); \
      const bool t_sure = (tp - kr > AMBER_KEPS) && !grazing; \
      const float ptol = cw_f(pl, 5) * rho; \
      const float mtol = grazing ? -3.402823466e+38f : -ptol;                             /* grazing: any in-plane position passes */ \
      bool plane_hit = false;
#ifndef AMBER_NO_CERTAIN_HITS
#define AMBER_PLANE_HIT(m_) plane_hit |= (m_) >= ptol
#else
#define AMBER_PLANE_HIT(m_)
#endif
#define AMBER_PAIR_RECORD() { \
        const float b = __builtin_fmaf(cw_f(tr, 0), Px, __builtin_fmaf(cw_f(tr, 2), Py, __builtin_fmaf(cw_f(tr, 4), Pz, cw_f(tr, 6)))); \
        const float a = __builtin_fmaf(cw_f(tr, 1), Px, __builtin_fmaf(cw_f(tr, 3), Py, __builtin_fmaf(cw_f(tr, 5), Pz, cw_f(tr, 7)))); \
        const float ba = b + a;                             /* 1 - gamma: the second triangle's first coordinate */ \
        const float g = 1.0f - ba; \
        const float m1 = __builtin_fminf(__builtin_fminf(b, a), g); \
        const float m2 = __builtin_fminf(__builtin_fminf(-b, 1.0f - a), ba); \
        const bool keep1 = (!(m1 < mtol) && t_ok) || degenerate;                          /* NaN coordinates -> keep */ \
        const bool keep2 = (!(m2 < mtol) && t_ok) || degenerate; \
        cand |= keep1 ? bit : 0u; \
        cand |= keep2 ? (bit << 1) : 0u; \
        AMBER_PLANE_HIT(__builtin_fmaxf(m1, m2)); }
    const int n_simple = kMulti ? fv.n_simple_planes : static_cast<int>(sc.n_simple_planes);
    int p = 0;
    for (; p < n_simple; p += 2) {                          // slabs (filter_build.h): two parallel planes of one parallelogram pair each
      AMBER_PLANE_NORMAL();
      { AMBER_PLANE_SETUP(); AMBER_PAIR_RECORD(); if (plane_hit && t_sure) t_upper = __builtin_fminf(t_upper, tp + kr); }
      pl += 8; tr += 8; bit <<= 2;
      { AMBER_PLANE_SETUP(); AMBER_PAIR_RECORD(); if (plane_hit && t_sure) t_upper = __builtin_fminf(t_upper, tp + kr); }
      pl += 8; tr += 8; bit <<= 2;
    }
    for (; p < n_planes; ++p, pl += 8) {                    // DevPlane = 8 dwords
      if (!(pl[6] & 0x80000000u)) { AMBER_PLANE_NORMAL(); } // a plane parallel to the previous one (same stored normal) reuses n.d, n.o and 1 / n.d
      AMBER_PLANE_SETUP();
      const int nt = static_cast<int>(pl[6] & 0x7fffffffu), np = static_cast<int>(pl[7]);
      for (int k = 0; k < np; ++k, tr += 8, bit <<= 2) {    // parallelogram pairs: one record, two candidate bits
        AMBER_PAIR_RECORD();
      }
      for (int k = 0; k < nt; ++k, tr += 8, bit <<= 1) {    // DevTriFilter = 8 dwords
        const float u = __builtin_fmaf(cw_f(tr, 0), Px, __builtin_fmaf(cw_f(tr, 2), Py, __builtin_fmaf(cw_f(tr, 4), Pz, cw_f(tr, 6))));
        const float v = __builtin_fmaf(cw_f(tr, 1), Px, __builtin_fmaf(cw_f(tr, 3), Py, __builtin_fmaf(cw_f(tr, 5), Pz, cw_f(tr, 7))));
        const float w = 1.0f - u - v;
        const float m = __builtin_fminf(__builtin_fminf(u, v), w);
        const bool keep = (!(m < mtol) && t_ok) || degenerate;
        cand |= keep ? bit : 0u;
        AMBER_PLANE_HIT(m);
      }
      if (plane_hit && t_sure) t_upper = __builtin_fminf(t_upper, tp + kr);
    }
#undef AMBER_PLANE_SETUP
#undef AMBER_PLANE_NORMAL
#undef AMBER_PAIR_RECORD
#undef AMBER_PLANE_HIT
    ConstWords sp = kMulti ? fv.spheres : (ConstWords)(sc.sphere_filters);
    const int ns = kMulti ? fv.n_sphere_filters : static_cast<int>(sc.n_sphere_filters);
    for (int k = 0; k < ns; ++k, sp += 8, bit <<= 1) {      // DevSphereFilter = 8 dwords
      const float cx = cw_f(sp, 0) - o.x, cy = cw_f(sp, 1) - o.y, cz = cw_f(sp, 2) - o.z;
      const float bb = __builtin_fmaf(cx, d.x, __builtin_fmaf(cy, d.y, cz * d.z));       // co.d
      const float c2 = __builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz));          // |co|^2
      const float r2 = cw_f(sp, 3);
      const float cc = c2 - r2;                                                           // |co|^2 - r^2
      const float tol = cw_f(sp, 4) * (c2 + r2);
      const bool miss = (__builtin_fmaf(bb, bb, -cc) < -tol) || (bb < 0.0f && cc > tol);  // no real root | both roots behind
      cand |= miss ? 0u : bit;                                                            // NaN -> keep
    }
  }
  // The filter's tolerances are derived for rays of the scene: origin within the model box, |d| <= 2 (DESIGN.md section
  // 5).  Anything else -- possible only when a scene hands the reference non-unit normals, whose sphere test then
  // reports "hits" far outside the scene -- skips the filter: every object becomes a candidate for the exact tests.
  if (!use_premask) {
    const float ex = Abs(o_world.x - sc.fp_center[0]), ey = Abs(o_world.y - sc.fp_center[1]), ez = Abs(o_world.z - sc.fp_center[2]);
    const bool in_model = __builtin_fmaxf(__builtin_fmaxf(ex, ey), ez) <= sc.fp_reach && (d.x * d.x + d.y * d.y + d.z * d.z) <= 4.0f;   // NaN -> false
    if (!in_model) { const uint32_t n_here = kMulti ? fv.n_objects : sc.n_objects; cand = n_here >= 32u ? 0xffffffffu : ((1u << n_here) - 1u); }
  }
  AMBER_STAMP(2);
  // Self trip.  A ray that leaves a triangle always re-selects that triangle in Phase A (t' ~ 0), and the exact test
  // then rejects it at its LAST step, t < kEPS (primitive_triangle.cc:122-125), after two wasted divisions.  All lanes
  // that carry such a self candidate evaluate just t -- the same operations the full test performs -- in one
  // common trip; "t <= kEPS" means the full test would reject whatever u and v are, so the bit is cleared.
  // Otherwise (t above kEPS, or NaN) the bit stays and the full exact test decides below.
  {
    const bool has_self = origin_slot >= 0 && ((cand >> (origin_slot & 31)) & 1u) != 0u;
    if (has_self) {
      const DevObject& ob = lds_objects[origin_slot];
      const V3 A = ld3(ob.a), E1 = ld3(ob.e1), E2 = ld3(ob.e2);
      const float det = Dot(Cross(d, E2), E1);
      const float t = Dot(Cross(o_world - A, E1), E2) / det;
      if (t <= AMBER_KEPS) cand &= ~(1u << origin_slot);
    }
  }
  // Phase B: filtered triangles first, then everything else (keeps the per-lane kind branch out of the hot loop)
  const uint32_t n_tris_here = kMulti ? fv.n_prog_tris : sc.n_prog_tris;
  const uint32_t tri_bits = n_tris_here >= 32u ? 0xffffffffu : ((1u << n_tris_here) - 1u);
  uint32_t mt = cand & tri_bits;
#ifdef AMBER_STAMPS
  while (__any(mt != 0u)) {                                  // diagnostic build: all lanes stay in the loop so that lane 0 can count
    stamp_ctx->acc[7] += 1ull + (static_cast<unsigned long long>(__popcll(__ballot(mt != 0u))) << 32);   // lo: wave trips, hi: lane tests
    if (mt != 0u) {
#else
  while (mt != 0u) {                                         // per lane; the wave leaves the loop with its last lane
    {
#endif
      const int slot = __builtin_ctz(mt);
      mt &= mt - 1u;
      const DevObject& ob = lds_objects[slot];
      IntersectTriangle<true>(ld3(ob.a), ld3(ob.e1), ld3(ob.e2), static_cast<int>(ob.kind >> 8), slot_base + slot, o_world, d, best);
    }
  }
  uint32_t mo = cand & ~tri_bits;
  while (mo != 0u) {
    const int slot = __builtin_ctz(mo);
    mo &= mo - 1u;
    const DevObject& ob = lds_objects[slot];
    IntersectObject<true>(ob, ob.kind & kKindMask, static_cast<int>(ob.kind >> 8), slot_base + slot, o_world, d, best);
  }
}

// Engine BVH: per-lane traversal of the flattened 2-wide BVH, near child first, far child pushed on a per-lane
// stack held in LDS (layout [level][thread]: conflict-free for ds_read/write_b32).  Boxes are padded on the host
// and the slab test ignores NaN axes (v_min/v_max return the non-NaN operand), so culling is conservative;
// every object of a visited leaf gets the exact reference test with the (t, index) tie rule.  Result = List.
#ifndef AMBER_BVH_STACK
#define AMBER_BVH_STACK 32
#endif
// Slab test of one child box.  The caller evaluates the six plane parameters t = (plane - o) / d as one FMA each
// (q * A + B, BvhTrav) -- already sorted into the three ENTRY and the three EXIT planes of this ray (the sign of d decides
// which of an axis' two planes is which, and a ray knows it once: BvhOperands), and already widened: the entry parameters
// are lowered and the exit parameters raised by the ray's slack (rounding of the FMA form, direction-length drift), folded
// into B.  So no per-axis min/max and no slack arithmetic is left here: 5 instructions per box instead of 15.
// Culling only has to be conservative.  Axes the ray is parallel to arrive as NaN planes, which max3/min3 skip; the lower
// clamp of the entry is -slack (the widened form of t >= 0), passed as neg_slack.  NaN anywhere -> treated as a hit.
__device__ __forceinline__ void SlabDecide(float nx, float ny, float nz, float fx, float fy, float fz, float neg_slack, float t_best, bool& hit, float& t_in) {
  const float tn = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, neg_slack));
  const float tf = __builtin_fminf(__builtin_fminf(fx, fy), fz);
  t_in = tn;
  hit = !(tn > tf) && !(tn > t_best);
}

// Fallback of engine BVH (traversal stack overflow; cannot happen with the builder's depth cap): scan the leaf-order
// array.  The (t, scene index) tie rule makes the visiting order irrelevant, so this equals ClosestHitList.
__device__ __forceinline__ void ClosestHitLeafList(const DevScene& sc, V3 o, V3 d, HitRec& best) {
  best.t = 3.402823466e+38f; best.u = 0.f; best.v = 0.f; best.idx = -1; best.slot = -1;
  for (uint32_t k = 0; k < sc.n_objects; ++k) {
    const DevObject& ob = sc.bvh_objects[k];
    IntersectObject<true>(ob, ob.kind, static_cast<int>(sc.bvh_prims[k]), static_cast<int>(k), o, d, best);
  }
}

// Traversal state of one ray.  It lives in registers (+ the lane's LDS stack) so that a traversal can be suspended
// while other lanes of the wave are shaded (pt_bvh_megakernel) and resumed afterwards.
// Slab parameter of a quantised plane q on axis c:  t = (gmin + q*step - o) / d  is evaluated as ONE fma, q * A + B, with
// A = step / d and B = (gmin - o) / d per ray.  B exists twice: b_in for the plane the ray ENTERS the slab through (the min
// plane if d > 0, the max plane otherwise), b_out for the other; everything that widens a box for this ray is folded in --
// the margin E (direction-length drift, BvhOperands) and the slack of the t interval.  rot (0 or 16 per axis) rotates a
// node word (min | max << 16) so that the entry plane sits in the low half.
struct BvhTrav {
  V3 A;                // step / d
  V3 b_in, b_out;      // entry planes: (gmin - o -+ E) / d - slack of the axis;  exit planes: ... + slack of the axis
  uint32_t rot[3];     // 16 where d < 0
  float neg_slack;     // -(smallest axis slack): lower clamp of the entry parameter
  int32_t cur;         // >= 0 inner node, < 0 leaf reference, AMBER_BVH_DONE finished
  int32_t pend;        // postponed leaf reference (< 0), 0 = none (BvhRound)
  int sp;              // entries on the lane's stack
  bool overflow;       // the stack was too small (cannot happen with the builder's depth cap): fall back to the list scan
};
#define AMBER_BVH_DONE 0x7fffffff
#define AMBER_BVH_REL_SLACK 9.5367431640625e-07f   /* 2^-20 */

// Slab-test operands of a ray.
__device__ __forceinline__ void BvhOperands(const DevScene& sc, V3 o, V3 d, BvhTrav& tr) {
  V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  // Direction length.  The reference never renormalises sampled directions (vector3.h:236-239) and its sphere test
  // assumes |d| = 1 (a = 1 in SolveQuadratic, primitive_sphere.cc:80-83): with |d|^2 = 1 + delta it accepts a ray whose
  // closest approach p to the centre, at distance s along the ray, satisfies p^2 <= r^2 + delta * s^2 (+ rounding, which
  // the sphere's box already covers).  delta is 1e-7 .. 5e-6 on ordinary paths (oracle_direction_length_stats) and
  // anything at all in scenes with non-unit disk normals.  The ray must therefore be allowed to miss a sphere's box by
  // E = sqrt(r^2 + delta s^2) - r <= min(delta * S^2 / r_min, sqrt(delta) * S), where S = |o - scene centre| + half the
  // scene diagonal bounds s (no point of the scene is farther from the origin) and r_min is the smallest sphere radius
  // (no spheres: E = 0).  Widening every box by E costs nothing per node: it moves into the two offset vectors.
  // t itself is measured in units of |d|: the reference's sphere distance and the geometric entry into the sphere's
  // box differ by a relative |delta|, which joins the slack as the absolute term 2 |delta| * S / |d|.
  const float len2 = d.x * d.x + d.y * d.y + d.z * d.z;
  const float delta = len2 - 1.0f;
  float E = 0.0f, slack_len = 0.0f;
  if (sc.bvh_inv_rmin > 0.0f && !(Abs(delta) <= 2.0e-8f)) {                          // NaN delta: nan_ray in BvhBegin
    const float cx = o.x - sc.bvh_center[0], cy = o.y - sc.bvh_center[1], cz = o.z - sc.bvh_center[2];
    const float S = 1.001f * (__builtin_amdgcn_sqrtf(cx * cx + cy * cy + cz * cz) + sc.bvh_half_diag);
    const float dpos = delta > 0.0f ? delta : 0.0f;
    E = 1.001f * __builtin_fminf(dpos * S * S * sc.bvh_inv_rmin, __builtin_amdgcn_sqrtf(dpos) * S);
    slack_len = 2.0f * Abs(delta) * S * __builtin_amdgcn_rsqf(__builtin_fminf(len2, 1.0f));   // t <= S / |d|
  }
  // An axis the ray is (almost) parallel to: |1 / d| beyond 1e12, or infinite.  Dropping it from the slab test (rounds 1-2) is
  // conservative but lets the ray walk every box of the sheet it lies in -- 40 000 nodes, 8 000 wave rounds, on the 1M-sphere
  // scene (tools/traversal_rounds_secondary.py).  It stays in with 1 / d clamped to +-1e12: the plane parameters keep their signs
  // and only shrink in magnitude, so an entry parameter that was positive is still a lower bound of itself and a negative one stays
  // non-positive; an exit parameter may now come out too SMALL, and that is repaired by raising the axis' slack by T, a bound on
  // the t of any hit (no point of the scene is farther than S from the origin).  In distance that slack is 3e-6 of the scene.
  const float kInvMax = 1.0e12f;
  const bool flat_x = !(Abs(inv.x) <= kInvMax), flat_y = !(Abs(inv.y) <= kInvMax), flat_z = !(Abs(inv.z) <= kInvMax);   // NaN d: nan_ray in BvhBegin
  float t_far = 0.0f;
  if (flat_x || flat_y || flat_z) {
    const float cx = o.x - sc.bvh_center[0], cy = o.y - sc.bvh_center[1], cz = o.z - sc.bvh_center[2];
    t_far = 1.001f * (__builtin_amdgcn_sqrtf(cx * cx + cy * cy + cz * cz) + sc.bvh_half_diag) * __builtin_amdgcn_rsqf(__builtin_fminf(len2, 1.0f)) * 1.001f;
    if (flat_x) inv.x = __builtin_copysignf(kInvMax, d.x);
    if (flat_y) inv.y = __builtin_copysignf(kInvMax, d.y);
    if (flat_z) inv.z = __builtin_copysignf(kInvMax, d.z);
  }
  V3 A = v3(sc.bvh_step[0] * inv.x, sc.bvh_step[1] * inv.y, sc.bvh_step[2] * inv.z);
  V3 bmn = v3((sc.bvh_gmin[0] - (o.x + E)) * inv.x, (sc.bvh_gmin[1] - (o.y + E)) * inv.y, (sc.bvh_gmin[2] - (o.z + E)) * inv.z);
  V3 bmx = v3((sc.bvh_gmin[0] - (o.x - E)) * inv.x, (sc.bvh_gmin[1] - (o.y - E)) * inv.y, (sc.bvh_gmin[2] - (o.z - E)) * inv.z);
  // Rounding of q * A + B: A and B carry three roundings each and the fma one more -- at most 2^-22 of
  // (|gmin - o| + E + scene extent) / |d| ON THE AXIS (`mg`), which also bounds |t| of every plane of the axis; the axis' slack
  // takes 2^-20 of that.  2^-16 costs 13 % on config 3: the term scales with |o / d|.  The slack is PER AXIS since round 3:
  // one value for the ray (the maximum over the axes, rounds 1-2) lets an almost axis-parallel ray -- |d.y| = 4e-7 on the
  // middle rows of a frame: mg.y = 7.5e6, slack 7 in units of t, more than the whole scene -- switch off the culling of the
  // OTHER two axes and of the closest hit so far: such a ray walked every node its plane y = o.y touches, 39 000 of them
  // against 60 for its neighbours (tools/traversal_rounds.py: 0.1 % of the eye rays of those rows took 450 .. 7 800 wave rounds,
  // the median 10).  The error bound never needed it: it is a bound on the axis' own plane parameters.
  V3 mg = v3((Abs(sc.bvh_gmin[0] - o.x) + E + sc.bvh_reach[0]) * Abs(inv.x), (Abs(sc.bvh_gmin[1] - o.y) + E + sc.bvh_reach[1]) * Abs(inv.y),
             (Abs(sc.bvh_gmin[2] - o.z) + E + sc.bvh_reach[2]) * Abs(inv.z));
  // An axis whose operands are still not finite (huge origin or scene) is taken OUT of the slab test by making
  // them NaN: fma(q, NaN, NaN) = NaN, which min/max skip.  Leaving +-inf in would not be conservative (inf - inf), and an
  // infinite value must not reach a slack either: an infinite slack makes every box "hit", and an axis-parallel ray
  // then walks the whole tree -- 2 M nodes, 0.75 s for one lane, found on config 3.
  const float kNaN = __builtin_nanf("");
  if (!(Abs(A.x) < 3.0e38f) || !(Abs(bmn.x) < 3.0e38f) || !(Abs(bmx.x) < 3.0e38f) || !(mg.x < 3.0e38f)) { A.x = kNaN; bmn.x = kNaN; bmx.x = kNaN; mg.x = kNaN; }
  if (!(Abs(A.y) < 3.0e38f) || !(Abs(bmn.y) < 3.0e38f) || !(Abs(bmx.y) < 3.0e38f) || !(mg.y < 3.0e38f)) { A.y = kNaN; bmn.y = kNaN; bmx.y = kNaN; mg.y = kNaN; }
  if (!(Abs(A.z) < 3.0e38f) || !(Abs(bmn.z) < 3.0e38f) || !(Abs(bmx.z) < 3.0e38f) || !(mg.z < 3.0e38f)) { A.z = kNaN; bmn.z = kNaN; bmx.z = kNaN; mg.z = kNaN; }
  // The subtraction / addition of a slack rounds once more, by at most 2^-24 of |B| + slack <= 2^-24 * (mg + mg * 2^-20 + slack_len):
  // inside what the axis' slack has to spare (2^-20 - 2^-22 of mg, and half of slack_len, which is twice the bound it stands for).
  const V3 slack = v3(AMBER_BVH_REL_SLACK * mg.x + slack_len + (flat_x ? t_far : 0.0f), AMBER_BVH_REL_SLACK * mg.y + slack_len + (flat_y ? t_far : 0.0f),
                      AMBER_BVH_REL_SLACK * mg.z + slack_len + (flat_z ? t_far : 0.0f));   // NaN on a NaN axis
  const bool nx = A.x < 0.0f, ny = A.y < 0.0f, nz = A.z < 0.0f;                       // NaN axes: either order, the planes are NaN
  tr.b_in = v3((nx ? bmx.x : bmn.x) - slack.x, (ny ? bmx.y : bmn.y) - slack.y, (nz ? bmx.z : bmn.z) - slack.z);
  tr.b_out = v3((nx ? bmn.x : bmx.x) + slack.x, (ny ? bmn.y : bmx.y) + slack.y, (nz ? bmn.z : bmx.z) + slack.z);
  tr.rot[0] = nx ? 16u : 0u; tr.rot[1] = ny ? 16u : 0u; tr.rot[2] = nz ? 16u : 0u;
  // Lower clamp of the entry parameter, the widened form of t >= 0.  Any value <= 0 is conservative (the exit parameters are
  // raised: a box the ray really enters at t >= 0 has every computed exit >= 0); the smallest of the axes' slacks keeps a margin.
  float smin = __builtin_fminf(__builtin_fminf(slack.x, slack.y), slack.z);           // fmin skips the NaN axes
  if (!(smin == smin)) smin = slack_len;                                              // no axis takes part
  tr.neg_slack = -smin;
  tr.A = A;
}

__device__ __forceinline__ void BvhBegin(const DevScene& sc, V3 o, V3 d, BvhTrav& tr, HitRec& best) {
  best.t = 3.402823466e+38f; best.u = 0.f; best.v = 0.f; best.idx = -1; best.slot = -1;
  BvhOperands(sc, o, d, tr);
  // A ray with a NaN component cannot hit anything: every exact test forms dot products over all components of o and
  // d, so every t it computes is NaN, and Closer() never accepts a NaN t (the List scan returns "no hit" too).
  const bool nan_ray = !(o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z);
  tr.cur = nan_ray ? AMBER_BVH_DONE : sc.bvh_root; tr.pend = 0; tr.sp = 0; tr.overflow = false;
}

// primitive_sphere.cc:75-107 from the compact (centre, radius) records of a leaf of 1..3 spheres, in two stages.
// Stage 1, all spheres of the leaf (loads issued together): the coefficients b, c and the sign of the discriminant -- 20
// instructions each.  Stage 2, only for the spheres whose discriminant is not negative (a third of them on config 3): the
// roots -- a square root and two divisions, 50 instructions -- the reference's choice of the root and the (t, index) rule;
// the object index (tie rule, reported hit) is fetched only when the distance can win.  A lane runs stage 2 once per
// surviving sphere, so a wave makes as many trips through it as its worst lane has survivors (mostly one) instead of one
// per sphere of the leaf.  The operations on each sphere are those of IntersectSphere; only their grouping differs.
#define AMBER_IDX_LAZY 0x7ffffffe   /* HitRec.idx of engine BVH: "slot is valid, the scene index has not been looked up" (> every real index) */
__device__ __forceinline__ void BvhResolveIndex(const DevScene& sc, HitRec& h) {
  if (h.idx == AMBER_IDX_LAZY) h.idx = static_cast<int>(sc.bvh_prims[h.slot]);
}
__device__ __forceinline__ void SphereCoefficients(float4 s, V3 o, V3 d, float& b, float& c) {
  const V3 co = v3(s.x, s.y, s.z) - o;
  b = -2.0f * Dot(co, d);
  c = SquaredLength(co) - s.w * s.w;
}
__device__ __forceinline__ void IntersectSphereLeaf(const DevScene& sc, uint32_t first, uint32_t count, V3 o, V3 d, HitRec& best AMBER_STAMP_PARAM_OPT) {
  const float4* sp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.bvh_spheres) + (first << 4));   // uniform base + 32-bit offset
  const float4 s0 = sp[0], s1 = sp[count > 1u ? 1u : 0u], s2 = sp[count > 2u ? 2u : 0u];
  float b0, c0, b1, c1, b2, c2;
  SphereCoefficients(s0, o, d, b0, c0); SphereCoefficients(s1, o, d, b1, c1); SphereCoefficients(s2, o, d, b2, c2);
  // algebra.h:33-34: "if (d < 0) return false" -- a NaN discriminant goes on (and ends as a NaN distance, which no hit accepts)
  uint32_t todo = (!(b0 * b0 - 4.0f * 1.0f * c0 < 0.0f) ? 1u : 0u) | (count > 1u && !(b1 * b1 - 4.0f * 1.0f * c1 < 0.0f) ? 2u : 0u) |
                  (count > 2u && !(b2 * b2 - 4.0f * 1.0f * c2 < 0.0f) ? 4u : 0u);
  while (todo) {
    AMBER_COUNT(1);
    const uint32_t k = (todo & 1u) ? 0u : ((todo & 2u) ? 1u : 2u);
    todo &= todo - 1u;
    const float b = k == 0u ? b0 : (k == 1u ? b1 : b2), c = k == 0u ? c0 : (k == 1u ? c1 : c2);
    float alpha, beta;
    if (SolveQuadratic(1.0f, b, c, alpha, beta)) {
      float t;
      bool ok = true;
      if (alpha > AMBER_KEPS) t = alpha; else if (beta > AMBER_KEPS) t = beta; else { ok = false; t = 0.f; }
      if (ok && IsFinite(t) && !(t > best.t)) {
        // The scene index of a sphere matters only for the tie rule: a strictly closer hit wins whatever its index, so the
        // dependent load of bvh_prims[] -- one more memory round trip per accepted hit, on the traversal's critical path --
        // is left out and the index marked unknown (AMBER_IDX_LAZY); an exact tie, or a consumer that reports the object
        // (traces, signatures, known-answer kernels: BvhResolveIndex), looks it up then.
        if (t < best.t) { best.t = t; best.idx = AMBER_IDX_LAZY; best.slot = static_cast<int>(first + k); }
        else {
          BvhResolveIndex(sc, best);
          const int i = static_cast<int>(sc.bvh_prims[first + k]);
          if (i < best.idx) { best.idx = i; best.slot = static_cast<int>(first + k); }
        }
      }
    }
  }
}

// primitive_triangle.cc:97-128 from the compact 48-byte records of a leaf of 1..3 triangles, in two stages (round 5; the sphere
// leaf's scheme).  A mesh scene used to pay, per leaf triangle and for every lane of the wave, a 64-byte object record, a dependent
// index load and the full test with its three IEEE divisions (~ 36 of its ~ 85 vector instructions) -- although three of four
// leaf triangles are missed.
// Stage 1, all triangles of the leaf (loads issued together): the reference's own numerators -- det = (d x E2).E1, a = (d x E2).T,
// b = (T x E1).d, c = (T x E1).E2, each operation the reference's, so u = a / det, v = b / det, t = c / det are its quotients -- and a
// CONSERVATIVE rejection on approximate quotients (v_rcp_f32, relative error < 4e-7, against a margin of 1e-5): a triangle is dropped
// only where the exact test certainly fails (u or v outside [0, 1], u + v > 1, t <= kEPS, or t beyond the closest hit so far);
// anything NaN, and a determinant too small for v_rcp_f32, survives.
// Stage 2, survivors only, one per trip: the three IEEE divisions of the kept numerators, the reference's comparisons in its order,
// the (t, index) rule.  A wave makes as many trips through it as its worst lane has survivors (mostly one).
#define AMBER_TRI_REJECT_EPS 1.0e-5f
__device__ __forceinline__ void TriangleNumerators(float4 t0, float4 t1, float4 t2, V3 o, V3 d, float& det, float& a, float& b, float& c) {
  const V3 A = v3(t0.x, t0.y, t0.z), E1 = v3(t0.w, t1.x, t1.y), E2 = v3(t1.z, t1.w, t2.x);
  const V3 P = Cross(d, E2);
  det = Dot(P, E1);
  const V3 T = o - A;
  a = Dot(P, T);
  const V3 Q = Cross(T, E1);
  b = Dot(Q, d);
  c = Dot(Q, E2);
}
__device__ __forceinline__ bool TriangleSurvives(float det, float a, float b, float c, float t_best) {
  const float inv = __builtin_amdgcn_rcpf(det);
  const float uh = a * inv, vh = b * inv, th = c * inv;
  const bool reject = uh < -AMBER_TRI_REJECT_EPS || uh > 1.0f + AMBER_TRI_REJECT_EPS || vh < -AMBER_TRI_REJECT_EPS || vh > 1.0f + AMBER_TRI_REJECT_EPS ||
                      uh + vh > 1.0f + AMBER_TRI_REJECT_EPS || th < AMBER_KEPS * (1.0f - AMBER_TRI_REJECT_EPS) || th > t_best * (1.0f + AMBER_TRI_REJECT_EPS);
  return !reject || !(Abs(det) >= 1.0e-30f);                // tiny (denormal) or NaN determinant: v_rcp_f32 is not trusted, the exact test decides
}
__device__ __forceinline__ void IntersectTriangleLeaf(const DevScene& sc, uint32_t first, uint32_t count, V3 o, V3 d, HitRec& best AMBER_STAMP_PARAM_OPT) {
  const float4* tp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.bvh_tris) + first * 48u);   // uniform base + 32-bit offset
  const uint32_t k1 = count > 1u ? 3u : 0u, k2 = count > 2u ? 6u : 0u;
  const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2], q0 = tp[k1], q1 = tp[k1 + 1u], q2 = tp[k1 + 2u], r0 = tp[k2], r1 = tp[k2 + 1u], r2 = tp[k2 + 2u];
  float det0, a0, b0, c0, det1, a1, b1, c1, det2, a2, b2, c2;
  TriangleNumerators(p0, p1, p2, o, d, det0, a0, b0, c0);
  TriangleNumerators(q0, q1, q2, o, d, det1, a1, b1, c1);
  TriangleNumerators(r0, r1, r2, o, d, det2, a2, b2, c2);
  uint32_t todo = (TriangleSurvives(det0, a0, b0, c0, best.t) ? 1u : 0u) | (count > 1u && TriangleSurvives(det1, a1, b1, c1, best.t) ? 2u : 0u) |
                  (count > 2u && TriangleSurvives(det2, a2, b2, c2, best.t) ? 4u : 0u);
  while (todo) {
    AMBER_COUNT(1);
    const uint32_t k = (todo & 1u) ? 0u : ((todo & 2u) ? 1u : 2u);
    todo &= todo - 1u;
    const float det = k == 0u ? det0 : (k == 1u ? det1 : det2), a = k == 0u ? a0 : (k == 1u ? a1 : a2);
    const float b = k == 0u ? b0 : (k == 1u ? b1 : b2), c = k == 0u ? c0 : (k == 1u ? c1 : c2);
    const float u = a / det, v = b / det, t = c / det;       // primitive_triangle.cc:104-120 (the reference forms v and t only when the earlier tests pass: no side effects)
    if (!(u > 1.0f || u < 0.0f) && !(v > 1.0f || v < 0.0f) && !(u + v > 1.0f) && !(t <= AMBER_KEPS) && IsFinite(t) && !(t > best.t)) {
      const int i = static_cast<int>(__float_as_uint(k == 0u ? p2.y : (k == 1u ? q2.y : r2.y)));
      bool take = t < best.t;
      if (!take) { BvhResolveIndex(sc, best); take = i < best.idx; }     // exact tie: the lower scene index (Closer<true>)
      if (take) { best.t = t; best.u = u; best.v = v; best.idx = i; best.slot = static_cast<int>(first + k); }
    }
  }
}

// A traversal advances in rounds of two phases.
// N-phase (BvhDescend, per lane): descend through at most AMBER_BVH_DESCENT_BUDGET inner nodes.  A leaf reached while the
// lane has none set aside is POSTPONED (tr.pend) and the walk goes on with the next subtree from the stack; a second leaf
// stops the lane.
// S-phase: the postponed leaves' objects get their exact tests -- once the wave has collected enough of them, or nobody can
// descend any more -- so that the expensive tests run with many lanes (counters of the undeferred form, 1M spheres: 8.0
// sphere tests per ray with 21 % of the lanes, 29.4 node visits with 46 %).  Postponing never changes the result: culling
// against a larger best.t is still conservative and the (t, index) rule makes the closest hit independent of the order.
// Unbounded descent ("while-while") makes every stopped lane wait for the slowest descent of the wave, one node per round
// ("if-if") interleaves too finely; round 1, 64-byte nodes, config 3 at 128 spp: budget 2 -> 195 ms, 3 -> 175, 4 -> 169,
// 5 -> 164, 6 -> 165, 8 -> 173, unbounded -> 189.
#ifndef AMBER_BVH_DESCENT_BUDGET
#define AMBER_BVH_DESCENT_BUDGET 5
#endif
// The same for the one-shot traversal (ClosestHitBvh: pt_megakernel<ENGINE_BVH>, the known-answer kernels), where nothing is resumed
#ifndef AMBER_ONE_SHOT_BVH_BUDGET
#define AMBER_ONE_SHOT_BVH_BUDGET AMBER_BVH_DESCENT_BUDGET
#endif
// Where a lane keeps the far children it has not visited yet.
//  BvhStackLds     the whole stack in LDS, [level][thread of the workgroup] (conflict-free ds_read/write_b32); a push beyond
//                  `cap` sets the overflow flag (the caller then falls back to the list scan; cannot happen with the
//                  builder's depth cap).
//  BvhStackHybrid  pt_bvh_pool_kernel: the first `lds_levels` levels in LDS, [level][lane of the wave]; deeper levels in
//                  global memory, [level][thread of the grid] (coalesced; one lane's store -> load of the same address is
//                  ordered like scratch memory is).  1M-sphere scene: a ray pushes 6.9 entries, 1.1 % of them at depth >= 8.
struct BvhStackLds {
  int32_t* base; int cap;                              // base = lds_stack + threadIdx.x; every kernel that uses it runs 256 threads
  __device__ __forceinline__ void push(int& sp, int32_t v, bool& overflow) const { if (sp < cap) { base[sp * 256] = v; ++sp; } else overflow = true; }
  __device__ __forceinline__ int32_t pop(int& sp) const { --sp; return base[sp * 256]; }
};
typedef int32_t __attribute__((address_space(3)))* LdsInts;
typedef int32_t __attribute__((address_space(1)))* GlobalInts;
struct BvhStackHybrid {
  LdsInts lds;               // WAVE-UNIFORM (SGPR): the wave's [level][lane] block
  GlobalInts glob;           // WAVE-UNIFORM: column of the wave's lane 0 in the global levels
  uint32_t glob_stride; int lds_levels, cap;
  // The pointers carry their address spaces on purpose.  With generic pointers the compiler turns "LDS level or global
  // level" into a select between two POINTERS and one flat_load (generic address, both memory counters, a full s_waitcnt)
  // on every pop, and spills the operands of the address it then keeps -- measured on the 1M-sphere scene: every traversal
  // trip 1.85x slower.  The lane index is recomputed where it is used (opaque to the optimiser, or it is hoisted and spilled).
  static __device__ __forceinline__ uint32_t LaneId() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  }
  __device__ __forceinline__ void push(int& sp, int32_t v, bool& overflow) const {
    if (sp < lds_levels) lds[sp * 64 + static_cast<int>(LaneId())] = v;
    else if (sp < cap) glob[static_cast<uint32_t>(sp - lds_levels) * glob_stride + LaneId()] = v;
    else { overflow = true; return; }
    ++sp;
  }
  __device__ __forceinline__ int32_t pop(int& sp) const {
    --sp;
    if (sp < lds_levels) return lds[sp * 64 + static_cast<int>(LaneId())];
    return glob[static_cast<uint32_t>(sp - lds_levels) * glob_stride + LaneId()];
  }
};
// The value of a plane's 16 bits (bvh_build.h): a binary16 number in [-1, 1] -- the conversion folds into the fma that consumes it
// (v_fma_mix_f32 with op_sel on the half of the word) -- or, in -DAMBER_BVH_F16=0 builds, an integer of the 16-bit grid.
#ifndef AMBER_BVH_F16
#define AMBER_BVH_F16 1
#endif
#if AMBER_BVH_F16
#define AMBER_PLANE_VALUE(bits16) static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(bits16)))
#else
#define AMBER_PLANE_VALUE(bits16) static_cast<float>(bits16)
#endif
template <class Stack, int kBudget = AMBER_BVH_DESCENT_BUDGET>
__device__ __forceinline__ void BvhDescend(const DevScene& sc, const Stack& stack, BvhTrav& tr, const float t_best AMBER_STAMP_PARAM_OPT) {
  int32_t cur = tr.cur, pend = tr.pend;
  int sp = tr.sp;
  AMBER_CLK(4);
#define AMBER_BVH_PARK() \
  if (cur < 0 && pend == 0) { pend = cur; if (sp > 0) cur = stack.pop(sp); else cur = AMBER_BVH_DONE; }
  AMBER_BVH_PARK();
  int budget = kBudget;
  while (cur >= 0 && cur != AMBER_BVH_DONE && budget-- > 0) {
    AMBER_COUNT(0);
#if AMBER_BVH_WIDE
    const uint4* nd = reinterpret_cast<const uint4*>(sc.bvh_nodes4 + cur);
    const uint4 p0 = nd[0], p1 = nd[1], p2 = nd[2], cr = nd[3];
#define AMBER_ROT(wd, c) __builtin_amdgcn_alignbit((wd), (wd), tr.rot[c])
#define AMBER_QLO(wd) AMBER_PLANE_VALUE((wd) & 0xffffu)
#define AMBER_QHI(wd) AMBER_PLANE_VALUE((wd) >> 16)
#define AMBER_CHILD(wx_, wy_, wz_, hit_, key_) { \
      const uint32_t wx = AMBER_ROT(wx_, 0), wy = AMBER_ROT(wy_, 1), wz = AMBER_ROT(wz_, 2); \
      const float nx = __builtin_fmaf(AMBER_QLO(wx), tr.A.x, tr.b_in.x), ny = __builtin_fmaf(AMBER_QLO(wy), tr.A.y, tr.b_in.y), nz = __builtin_fmaf(AMBER_QLO(wz), tr.A.z, tr.b_in.z); \
      const float fx = __builtin_fmaf(AMBER_QHI(wx), tr.A.x, tr.b_out.x), fy = __builtin_fmaf(AMBER_QHI(wy), tr.A.y, tr.b_out.y), fz = __builtin_fmaf(AMBER_QHI(wz), tr.A.z, tr.b_out.z); \
      float tin; SlabDecide(nx, ny, nz, fx, fy, fz, tr.neg_slack, t_best, hit_, tin); key_ = hit_ ? tin : 3.402823466e+38f; }
    bool h0, h1, h2, h3; float k0, k1, k2, k3;
    AMBER_CHILD(p0.x, p0.y, p0.z, h0, k0); AMBER_CHILD(p0.w, p1.x, p1.y, h1, k1);
    AMBER_CHILD(p1.z, p1.w, p2.x, h2, k2); AMBER_CHILD(p2.y, p2.z, p2.w, h3, k3);
#undef AMBER_CHILD
#undef AMBER_ROT
#undef AMBER_QLO
#undef AMBER_QHI
    int32_t r0 = static_cast<int32_t>(cr.x), r1 = static_cast<int32_t>(cr.y), r2 = static_cast<int32_t>(cr.z), r3 = static_cast<int32_t>(cr.w);
    const int n_hit = (h0 ? 1 : 0) + (h1 ? 1 : 0) + (h2 ? 1 : 0) + (h3 ? 1 : 0);
    // nearest first: sort the four (entry, reference) pairs (misses carry FLT_MAX) -- network (0,1)(2,3)(0,2)(1,3)(1,2)
#define AMBER_CSWAP(ka, ra, kb, rb) { const bool sw = kb < ka; const float kt = sw ? kb : ka; kb = sw ? ka : kb; ka = kt; const int32_t rt = sw ? rb : ra; rb = sw ? ra : rb; ra = rt; }
    AMBER_CSWAP(k0, r0, k1, r1); AMBER_CSWAP(k2, r2, k3, r3); AMBER_CSWAP(k0, r0, k2, r2); AMBER_CSWAP(k1, r1, k3, r3); AMBER_CSWAP(k1, r1, k2, r2);
#undef AMBER_CSWAP
    if (n_hit > 0) {
      // the others go on the stack, the farthest first
      if (n_hit > 3) stack.push(sp, r3, tr.overflow);
      if (n_hit > 2) stack.push(sp, r2, tr.overflow);
      if (n_hit > 1) stack.push(sp, r1, tr.overflow);
      cur = r0;
    } else if (sp > 0) {
      cur = stack.pop(sp);
    } else {
      cur = AMBER_BVH_DONE;
    }
#else
    // uniform base + 32-bit byte offset (the tree is < 4 GB): global_load with an SGPR base, no 64-bit address arithmetic per visit
    const uint4* nd = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.bvh_nodes) + (static_cast<uint32_t>(cur) << 5));
    const uint4 p = nd[0], q = nd[1];
    const int32_t left = static_cast<int32_t>(q.z), right = static_cast<int32_t>(q.w);
    // rotate each axis word so that the entry plane is the low half, convert (v_cvt_f32_u32 with a half-word select), one fma
#define AMBER_ROT(wd, c) __builtin_amdgcn_alignbit((wd), (wd), tr.rot[c])
#define AMBER_QLO(wd) AMBER_PLANE_VALUE((wd) & 0xffffu)
#define AMBER_QHI(wd) AMBER_PLANE_VALUE((wd) >> 16)
    const uint32_t wlx = AMBER_ROT(p.x, 0), wly = AMBER_ROT(p.y, 1), wlz = AMBER_ROT(p.z, 2), wrx = AMBER_ROT(p.w, 0), wry = AMBER_ROT(q.x, 1), wrz = AMBER_ROT(q.y, 2);
    const float lnx = __builtin_fmaf(AMBER_QLO(wlx), tr.A.x, tr.b_in.x), lny = __builtin_fmaf(AMBER_QLO(wly), tr.A.y, tr.b_in.y), lnz = __builtin_fmaf(AMBER_QLO(wlz), tr.A.z, tr.b_in.z);
    const float lfx = __builtin_fmaf(AMBER_QHI(wlx), tr.A.x, tr.b_out.x), lfy = __builtin_fmaf(AMBER_QHI(wly), tr.A.y, tr.b_out.y), lfz = __builtin_fmaf(AMBER_QHI(wlz), tr.A.z, tr.b_out.z);
    const float rnx = __builtin_fmaf(AMBER_QLO(wrx), tr.A.x, tr.b_in.x), rny = __builtin_fmaf(AMBER_QLO(wry), tr.A.y, tr.b_in.y), rnz = __builtin_fmaf(AMBER_QLO(wrz), tr.A.z, tr.b_in.z);
    const float rfx = __builtin_fmaf(AMBER_QHI(wrx), tr.A.x, tr.b_out.x), rfy = __builtin_fmaf(AMBER_QHI(wry), tr.A.y, tr.b_out.y), rfz = __builtin_fmaf(AMBER_QHI(wrz), tr.A.z, tr.b_out.z);
#undef AMBER_ROT
#undef AMBER_QLO
#undef AMBER_QHI
    bool hl, hr; float tl, tr_;
    SlabDecide(lnx, lny, lnz, lfx, lfy, lfz, tr.neg_slack, t_best, hl, tl);
    SlabDecide(rnx, rny, rnz, rfx, rfy, rfz, tr.neg_slack, t_best, hr, tr_);
    if (hl && hr) {
      const bool left_first = !(tr_ < tl);
      const int32_t near_ = left_first ? left : right, far_ = left_first ? right : left;
      stack.push(sp, far_, tr.overflow);              // overflow: stay correct anyway (list scan at the end)
      cur = near_;
    } else if (hl) {
      cur = left;
    } else if (hr) {
      cur = right;
    } else if (sp > 0) {
      cur = stack.pop(sp);
    } else {
      cur = AMBER_BVH_DONE;
    }
#endif
    AMBER_BVH_PARK();
  }
#undef AMBER_BVH_PARK
  AMBER_CLK(2);
  tr.cur = cur; tr.sp = sp; tr.pend = pend;
}

// Exact tests of one leaf by the lane that owns the ray.
__device__ __forceinline__ void BvhLeafPrivate(const DevScene& sc, int32_t leaf, V3 o, V3 d, HitRec& best AMBER_STAMP_PARAM_OPT) {
  AMBER_COUNT_LEAVES(2);
  const uint32_t ref = static_cast<uint32_t>(-(leaf + 1));
  const uint32_t first = ref >> 4, count = ref & 3u;
  if (ref & 4u) {                                           // spheres only: one 16-byte record each
    IntersectSphereLeaf(sc, first, count, o, d, best AMBER_STAMP_ARG);
  } else if (ref & 8u) {                                    // triangles only: one 48-byte record each
    IntersectTriangleLeaf(sc, first, count, o, d, best AMBER_STAMP_ARG);
  } else {
    BvhResolveIndex(sc, best);                                     // the tie rule of Closer<true> compares real indices
    for (uint32_t k = 0; k < count; ++k) {
      const uint32_t oi = sc.bvh_prims[first + k];                 // scene index (tie rule, reported hit); independent of ...
      const DevObject& ob = sc.bvh_objects[first + k];             // ... the record itself, stored in leaf order: no dependent load
      IntersectObject<true>(ob, ob.kind, static_cast<int>(oi), static_cast<int>(first + k), o, d, best);
    }
  }
}

// S-phase: the postponed leaves are tested once a third of the lanes still traversing hold one, or nobody can descend
// (config 3 at 128 spp: no postponing 142.3 ms; 1/1 168.7, 3/4 146.8, 1/2 140.4, 1/3 137.5, 1/4 138.8, 1/6 140.5).
// (Tried: dealing the collected sphere tests out one per lane through ds_bpermute, every lane of the wave working -- 13
// trips with 68 % of the lanes instead of 29 with 30 %, but 160 ms: the longer dependent chain per round costs more.)
__device__ __forceinline__ void BvhLeafPhase(const DevScene& sc, V3 o, V3 d, BvhTrav& tr, HitRec& best AMBER_STAMP_PARAM_OPT) {
  const bool has_pend = tr.pend != 0;
  const uint32_t n_lanes = static_cast<uint32_t>(__popcll(__ballot(true)));
  const uint32_t n_pend = static_cast<uint32_t>(__popcll(__ballot(has_pend)));
  const bool nobody_descends = __ballot(tr.cur >= 0 && tr.cur != AMBER_BVH_DONE) == 0ull;
  if (has_pend && (n_pend * 3u >= n_lanes || nobody_descends)) { BvhLeafPrivate(sc, tr.pend, o, d, best AMBER_STAMP_ARG); tr.pend = 0; }
  AMBER_CLK(3);
}

// One round for a lane on its own; returns false when the traversal is complete.
template <class Stack, int kBudget = AMBER_BVH_DESCENT_BUDGET>
__device__ __forceinline__ bool BvhRoundOn(const DevScene& sc, const Stack& stack, V3 o, V3 d, BvhTrav& tr, HitRec& best AMBER_STAMP_PARAM_OPT) {
  BvhDescend<Stack, kBudget>(sc, stack, tr, best.t AMBER_STAMP_ARG);
  BvhLeafPhase(sc, o, d, tr, best AMBER_STAMP_ARG);
  return tr.cur != AMBER_BVH_DONE || tr.pend != 0;
}
__device__ __forceinline__ bool BvhRound(const DevScene& sc, int32_t* lds_stack, V3 o, V3 d, BvhTrav& tr, HitRec& best, const int stack_cap = AMBER_BVH_STACK AMBER_STAMP_PARAM_OPT) {
  const BvhStackLds stack{lds_stack + threadIdx.x, stack_cap};
  return BvhRoundOn(sc, stack, o, d, tr, best AMBER_STAMP_ARG);
}

__device__ __forceinline__ void ClosestHitBvh(const DevScene& sc, int32_t* lds_stack, V3 o, V3 d, HitRec& best, const int stack_cap = AMBER_BVH_STACK) {
  BvhTrav tr;
  BvhBegin(sc, o, d, tr, best);
  const BvhStackLds stack{lds_stack + threadIdx.x, stack_cap};
  while (BvhRoundOn<BvhStackLds, AMBER_ONE_SHOT_BVH_BUDGET>(sc, stack, o, d, tr, best)) {}
  if (__any(tr.overflow)) { if (tr.overflow) ClosestHitLeafList(sc, o, d, best); }
  BvhResolveIndex(sc, best);                                       // callers of this form report the object
}

// Engine REFERENCE_BVH: BVH::Node::Cast (acceleration_bvh.h:340-403) on the reference's own tree, without recursion.
//
// The recursion passes a bound `distance` down and hands hits up; every hit it accepts is strictly closer than the bound it was searched
// under, and a far child is searched under the near child's hit distance (:393-396) -- so the bound in force at any moment is the closest
// hit found so far in the whole cast, and one running `best` serves every level.  A node whose two children are both entered (:374-383)
// visits the nearer one (left_in < right_in ? left : right) and comes back for the other: that one goes on the stack with its entry
// distance max(left_in, right_in).  When it is popped the recursion's three cases are one comparison:
//   near subtree found nothing  -> best is still the bound the far child's box was accepted under, t_in <= bound: visit (:384-386)
//   near hit in front of the far box (best.t < t_in) -> return the near hit: skip (:387-389)
//   otherwise -> search the far child under the near hit's distance (:391-400).
// A leaf scans its objects in the order the build left them, a hit replacing the best iff it is strictly closer (:343-355): on equal
// distances the FIRST object met wins -- not the lower scene index of the List rule the other engines implement.
// The slab test is aabb.cc:28-62 operation for operation: reciprocal direction by IEEE division, (plane - origin) * reciprocal,
// _mm_min_ps / _mm_max_ps (the SECOND operand when one is NaN), std::max / std::min over {0 | distance, x, y, z} left to right.
__device__ __forceinline__ float SseMin(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float SseMax(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ bool ReferenceSlab(const float* mn, const float* mx, V3 o, V3 inv, float distance, float& t_in) {
  const float t0x = (mn[0] - o.x) * inv.x, t0y = (mn[1] - o.y) * inv.y, t0z = (mn[2] - o.z) * inv.z;
  const float t1x = (mx[0] - o.x) * inv.x, t1y = (mx[1] - o.y) * inv.y, t1z = (mx[2] - o.z) * inv.z;
  const float nx = SseMin(t0x, t1x), ny = SseMin(t0y, t1y), nz = SseMin(t0z, t1z);
  const float fx = SseMax(t0x, t1x), fy = SseMax(t0y, t1y), fz = SseMax(t0z, t1z);
  float t_min = 0.0f;                                    // std::max({t_min, ..}): the running value is replaced iff it is < the next
  if (t_min < nx) t_min = nx;
  if (t_min < ny) t_min = ny;
  if (t_min < nz) t_min = nz;
  float t_max = distance;                                // std::min({t_max, ..}): replaced iff the next is < it
  if (fx < t_max) t_max = fx;
  if (fy < t_max) t_max = fy;
  if (fz < t_max) t_max = fz;
  t_in = t_min;
  return t_min <= t_max;
}
__device__ __forceinline__ void ClosestHitReferenceBvh(const DevScene& sc, V3 o, V3 d, HitRec& best) {
  best.t = 3.402823466e+38f; best.u = 0.f; best.v = 0.f; best.idx = -1; best.slot = -1;   // Acceleration::Cast(ray, max()), acceleration.h:46-51
  const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  uint2* const stack = sc.ref_stack + (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x);
  const size_t stride = sc.ref_stack_stride;
  int sp = 0;
  // The newest entry stays in registers (top_ref / top_in; top_ref == kNone: none): a pop right after a push -- the common case, a leaf or a miss
  // below the node that pushed -- costs no memory round trip; an entry reaches the global stack only when a second one is pushed on top of it.
  constexpr uint32_t kNone = 0x7fffffffu;
  uint32_t top_ref = kNone; float top_in = 0.0f;
  int32_t cur = sc.bvh_root;                             // the root is cast without a test of its own box (:152-156)
  for (;;) {
    if (cur >= 0) {
      const uint4* nd = reinterpret_cast<const uint4*>(sc.ref_nodes + cur);
      const uint4 w0 = nd[0], w1 = nd[1], w2 = nd[2], w3 = nd[3];
      const float lmin[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)}, lmax[3] = {__uint_as_float(w0.w), __uint_as_float(w1.x), __uint_as_float(w1.y)};
      const float rmin[3] = {__uint_as_float(w1.z), __uint_as_float(w1.w), __uint_as_float(w2.x)}, rmax[3] = {__uint_as_float(w2.y), __uint_as_float(w2.z), __uint_as_float(w2.w)};
      const int32_t left = static_cast<int32_t>(w3.x), right = static_cast<int32_t>(w3.y);
      float lin, rin;
      const bool lh = ReferenceSlab(lmin, lmax, o, inv, best.t, lin);
      const bool rh = ReferenceSlab(rmin, rmax, o, inv, best.t, rin);
      if (lh && rh) {
        const bool left_near = lin < rin;
        const uint32_t far_ = static_cast<uint32_t>(left_near ? right : left);
        if (top_ref != kNone) { stack[static_cast<size_t>(sp) * stride] = make_uint2(top_ref, __float_as_uint(top_in)); ++sp; }
        top_ref = far_; top_in = left_near ? rin : lin;                                                        // std::max(left_in, right_in)
        cur = left_near ? left : right;
        continue;
      }
      if (lh) { cur = left; continue; }
      if (rh) { cur = right; continue; }
    } else {
      // (best.idx holds the leaf-order slot until the end: strict < needs no index, and the scene index is one load for the winner)
      const DevRefLeaf lf = sc.ref_leaves[-(cur + 1)];
      const uint32_t last = lf.first + (lf.count & 0x3fffffffu);
      if (lf.count & 0x80000000u) {                      // spheres only: the 16-byte records of engine BVH's sphere leaves, same operands
        for (uint32_t k = lf.first; k < last; ++k) {
          const float4 sp4 = sc.bvh_spheres[k];
          IntersectSphere<false>(v3(sp4.x, sp4.y, sp4.z), sp4.w, static_cast<int>(k), static_cast<int>(k), o, d, best);
        }
      } else if (lf.count & 0x40000000u) {               // triangles only: {A.xyz E1.x} {E1.yz E2.xy} {E2.z ..}
        for (uint32_t k = lf.first; k < last; ++k) {
          // (a conservative rejection on v_rcp_f32 quotients in front of the divisions, as in engine BVH's leaves, was measured neutral here:
          //  terrain 44.1 -> 44.8 ms, room 26.7 -> 27.8 -- this walk is bound by its node visits, EXPERIMENTS.md round 5)
          const float4 t0 = sc.bvh_tris[3u * k], t1 = sc.bvh_tris[3u * k + 1u], t2 = sc.bvh_tris[3u * k + 2u];
          IntersectTriangle<false>(v3(t0.x, t0.y, t0.z), v3(t0.w, t1.x, t1.y), v3(t1.z, t1.w, t2.x), static_cast<int>(k), static_cast<int>(k), o, d, best);
        }
      } else {
        for (uint32_t k = lf.first; k < last; ++k) {
          const DevObject& ob = sc.bvh_objects[k];
          IntersectObject<false>(ob, ob.kind, static_cast<int>(k), static_cast<int>(k), o, d, best);
        }
      }
    }
    bool more = false;
    for (;;) {
      uint32_t ref; float fin;
      if (top_ref != kNone) { ref = top_ref; fin = top_in; top_ref = kNone; }
      else if (sp > 0) { --sp; const uint2 e = stack[static_cast<size_t>(sp) * stride]; ref = e.x; fin = __uint_as_float(e.y); }
      else break;
      if (!(best.t < fin)) { cur = static_cast<int32_t>(ref); more = true; break; }
    }
    if (!more) break;
  }
  if (best.slot >= 0) best.idx = static_cast<int>(sc.bvh_prims[best.slot]);
}

enum { ENGINE_LIST = 1, ENGINE_TWO_PHASE = 2, ENGINE_BVH = 3, ENGINE_TWO_PHASE_N = 5, ENGINE_REF_BVH = 6 };   // (4 is the public WAVEFRONT; 5 = two-phase over groups of 32 objects)

template <int kEngine>
__device__ __forceinline__ void ClosestHit(const DevScene& sc, const DevObject* lds_objects, int32_t* lds_stack, V3 o, V3 d, int origin_slot, HitRec& best AMBER_STAMP_PARAM,
                                           const bool use_premask = false, const uint32_t premask = 0u, const int bvh_stack_cap = AMBER_BVH_STACK) {
  if (kEngine == ENGINE_TWO_PHASE) ClosestHitTwoPhase(sc, lds_objects, o, d, origin_slot, best AMBER_STAMP_ARG, use_premask, premask);
  else if (kEngine == ENGINE_TWO_PHASE_N) ClosestHitTwoPhaseGroups(sc, lds_objects, o, d, origin_slot, best AMBER_STAMP_ARG, use_premask, premask);
  else if (kEngine == ENGINE_BVH) ClosestHitBvh(sc, lds_stack, o, d, best, bvh_stack_cap);
  else if (kEngine == ENGINE_REF_BVH) ClosestHitReferenceBvh(sc, o, d, best);
  else ClosestHitList(sc, o, d, best);
  AMBER_STAMP(3);
}

// position / normal of the winning hit, evaluated exactly as the reference's Intersect() does
template <uint32_t kKindMask = 0xffu>
__device__ __forceinline__ void ResolveHit(const DevObject* objects, const HitRec& h, V3 o, V3 d, V3& pos, V3& normal, uint32_t& material) {
  const DevObject* ob = objects + h.slot;
  const uint32_t kind = ob->kind & kKindMask;         // the LDS image of the two-phase engine tags kind with index << 8; the grouped engine's also with bit 7 (a filtered triangle)
  material = ob->material;
  const V3 A = ld3(ob->a);
  if (kind == PRIM_TRIANGLE) {
    pos = A + h.u * ld3(ob->e1) + h.v * ld3(ob->e2);     // primitive_triangle.cc:127
    normal = ld3(ob->n);
  } else if (kind == PRIM_SPHERE) {
    pos = o + h.t * d;                                   // primitive_sphere.cc:91-95
    normal = Normalize(o + h.t * d - A);
  } else if (kind == PRIM_DISK) {
    pos = o + h.t * d; normal = ld3(ob->e1);
  } else {
    const V3 N = ld3(ob->e1);
    const float hh = Dot(h.t * d - (A - o), N);
    pos = o + h.t * d;
    normal = Normalize(o + h.t * d - A - hh * N);
  }
}

// Copies the object records into the workgroup's LDS image (two-phase engine only; n_objects <= 32).
#define AMBER_MAX_LDS_OBJECTS 32
#define AMBER_MAX_GROUP_OBJECTS 128          /* engine TWO_PHASE_N: four groups of 32 */
template <bool kGroups = false>                          // kGroups: the grouped engine's image, 32 slots per group (sc.n_lds_objects records); else the scene's n_objects -- the
__device__ __forceinline__ void StageObjects(const DevScene& sc, DevObject* lds_objects) {   // 32-object kernel must not read one more field of the scene record: its SGPRs are spilled as it is
  const uint32_t n_dwords = (kGroups ? sc.n_lds_objects : sc.n_objects) * (sizeof(DevObject) / 4u);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.prog_objects);
  uint32_t* dst = reinterpret_cast<uint32_t*>(lds_objects);
  for (uint32_t k = threadIdx.x; k < n_dwords; k += blockDim.x) dst[k] = src[k];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// materials (src/amber/scene/material_*.cc)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 PerfectReflection(V3 incident, V3 normal, float signed_cos) {   // geometry.h:38-47
  return (2.0f * signed_cos) * normal - incident;
}
__device__ __forceinline__ V3 HemispherePSA(V3 w, uint64_t& rng) {          // sampling.h:234-265
  V3 u, v; OrthonormalBasis(w, u, v);
  const float r0 = Uniform(rng);
  const float r1 = Uniform(rng);
  const float cos_theta = Sqrt(r0);
  const float sin_theta = Sqrt(1.0f - r0);
  const float phi = 2.0f * 3.14159274f * r1;
  float sp, cp; SinCos(phi, sp, cp);
  return u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
}
__device__ __forceinline__ V3 CosinePower(V3 w, float exponent, uint64_t& rng) {   // sampling.h:267-300
  V3 u, v; OrthonormalBasis(w, u, v);
  const float r0 = Uniform(rng);
  const float r1 = Uniform(rng);
  const float cos_theta = Pow(r0, 1.0f / (exponent + 1.0f));
  const float sin_theta = Sqrt(1.0f - cos_theta * cos_theta);
  const float phi = 2.0f * 3.14159274f * r1;
  float sp, cp; SinCos(phi, sp, cp);
  return u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
}

// Scene::Radiance -> DiffuseLight::Radiance (material_diffuse_light.h:127-139)
__device__ __forceinline__ V3 Radiance(const DevMaterial& m, V3 normal, V3 dir_out) {
  if (m.kind != MAT_DIFFUSE_LIGHT) return v3(0.f, 0.f, 0.f);
  if (Dot(dir_out, normal) <= 0.0f) return v3(0.f, 0.f, 0.f);
  return ld3(m.rho);
}

// Scene::SampleLight -> Material::SampleLight (+ rho forwarders, material_basic.h:233-245, 327-338)
__device__ __forceinline__ void SampleLight(const DevMaterial& m, V3 normal, V3 dir_out, uint64_t& rng, V3& dir_in, V3& weight) {
  const V3 rho = ld3(m.rho);
  const uint32_t kind = m.kind;
#define AMBER_COS_O() Dot(dir_out, normal)
#define AMBER_MIRROR(c_) PerfectReflection(dir_out, normal, c_)
  if (kind == MAT_LAMBERTIAN || kind == MAT_PHONG) {
    // Lambertian (material_lambertian.cc:61-70, HemispherePSA sampling.h:234-265) and Phong (material_phong.cc:81-106,
    // CosinePower sampling.h:267-300) share the lobe construction -- orthonormal basis, two uniforms, sin/cos of phi,
    // the three-term combination -- and differ only in the lobe axis and in cos(theta).  One code path serves both,
    // so a wave with lanes on both materials pays for the shared part once; each lane still executes exactly the
    // operations of its own material (Phong re-samples until the direction is on the side of dir_out).
    const bool phong = kind == MAT_PHONG;
    const float signed_cos_o = AMBER_COS_O();
    const V3 w = phong ? AMBER_MIRROR(signed_cos_o) : (signed_cos_o > 0.0f ? normal : -normal);
    V3 u, v; OrthonormalBasis(w, u, v);                  // CosinePower rebuilds the same basis on every attempt
    // The reference's Phong loop re-samples forever when no direction of the lobe lies on dir_out's side (possible with a
    // normal that is not of unit length); a kernel must terminate, so attempt AMBER_PHONG_MAX_TRIES is accepted as it
    // is (the oracle does the same; with a proper normal at least half of the lobe is acceptable: probability 2^-1024).
    for (int attempt = 1;; ++attempt) {
      const float r0 = Uniform(rng);
      const float r1 = Uniform(rng);
      float cos_theta, sin_theta;
      if (phong) {
        cos_theta = Pow(r0, m.aux0);                     // r0 ^ (1 / (e + 1))
        sin_theta = Sqrt(1.0f - cos_theta * cos_theta);
      } else {
        cos_theta = Sqrt(r0);
        sin_theta = Sqrt(1.0f - r0);
      }
      const float phi = 2.0f * 3.14159274f * r1;
      float sp, cp; SinCos(phi, sp, cp);
      const V3 di = u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
      if (!phong) { dir_in = di; weight = 1.0f * rho; break; }
      const float signed_cos_i = Dot(di, normal);
      if (signed_cos_o * signed_cos_i <= 0.0f && attempt < AMBER_PHONG_MAX_TRIES) continue;
      dir_in = di;
      weight = (m.aux1 * Abs(signed_cos_i)) * rho;      // (e + 2) / (e + 1) * |cos|
      break;
    }
  } else if (kind == MAT_SPECULAR) {                     // material_specular.cc:62-70
    dir_in = AMBER_MIRROR(AMBER_COS_O());
    weight = 1.0f * rho;
  } else if (kind == MAT_REFRACTION) {                   // material_refraction.cc:177-220
    const float signed_cos_alpha = AMBER_COS_O();
    const float ior = signed_cos_alpha > 0.0f ? m.aux0 : m.param;   // 1 / ior when entering
    const float squared_cos_beta = 1.0f - (1.0f - signed_cos_alpha * signed_cos_alpha) * (ior * ior);
    const V3 dir_r = AMBER_MIRROR(signed_cos_alpha);
    if (squared_cos_beta < 0.0f) {
      dir_in = dir_r; weight = 1.0f * rho;
    } else {
      const float cos_alpha = Abs(signed_cos_alpha);
      const float cos_beta = Sqrt(squared_cos_beta);
      const V3 dir_t = (-ior) * dir_out + ((signed_cos_alpha < 0.0f ? 1.0f : -1.0f) * cos_beta + ior * signed_cos_alpha) * normal;
      // Schlick (material_refraction.cc:271-275): r0 + (1 - r0) * pow(1 - cos, 5) evaluated in double
      const float rho_r = static_cast<float>(static_cast<double>(m.r0) + static_cast<double>(1.0f - m.r0) * Pow5(1.0f - cos_alpha));
      const float rho_t = (1.0f - rho_r) * (ior * ior);
      const float rho_s = rho_r + rho_t;
      const float p_r = (rho_r / rho_s + 0.5f) / 2.0f;
      const float p_t = (rho_t / rho_s + 0.5f) / 2.0f;
      // one division for both outcomes: the lane's own operands are selected first (the same operation on the same values)
      const bool reflect = Uniform(rng) < p_r;
      dir_in = reflect ? dir_r : dir_t;
      weight = ((reflect ? rho_r : rho_t) / (reflect ? p_r : p_t)) * rho;
    }
  } else if (kind == MAT_EYE) {                          // material_eye.h:146-155
    dir_in = -dir_out; weight = v3(1.f, 1.f, 1.f);
  } else {                                               // DiffuseLight: Scatter() (material_diffuse_light.h:185-194)
    dir_in = v3(0.f, 0.f, 0.f); weight = v3(0.f, 0.f, 0.f);
  }
#undef AMBER_COS_O
#undef AMBER_MIRROR
}

// Scene::SampleImportance: identical to SampleLight for the symmetric forwarders, Eye and DiffuseLight
// (material_basic.h:340-351); BasicRefraction::SampleImportance (material_refraction.cc:222-263) drops the ior^2
// radiance scaling and uses p = (rho + 0.5) / 2.
__device__ __forceinline__ void SampleImportance(const DevMaterial& m, V3 normal, V3 dir_out, uint64_t& rng, V3& dir_in, V3& weight) {
  if (m.kind != MAT_REFRACTION) { SampleLight(m, normal, dir_out, rng, dir_in, weight); return; }
  const V3 rho = ld3(m.rho);
  const float signed_cos_alpha = Dot(dir_out, normal);
  const float ior = signed_cos_alpha > 0.0f ? m.aux0 : m.param;   // 1 / ior when entering
  const float squared_cos_beta = 1.0f - (1.0f - signed_cos_alpha * signed_cos_alpha) * (ior * ior);
  const V3 dir_r = PerfectReflection(dir_out, normal, signed_cos_alpha);
  if (squared_cos_beta < 0.0f) { dir_in = dir_r; weight = 1.0f * rho; return; }
  const float cos_alpha = Abs(signed_cos_alpha);
  const float cos_beta = Sqrt(squared_cos_beta);
  const V3 dir_t = (-ior) * dir_out + ((signed_cos_alpha < 0.0f ? 1.0f : -1.0f) * cos_beta + ior * signed_cos_alpha) * normal;
  const float rho_r = static_cast<float>(static_cast<double>(m.r0) + static_cast<double>(1.0f - m.r0) * Pow5(1.0f - cos_alpha));
  const float rho_t = 1.0f - rho_r;
  const float p_r = (rho_r + 0.5f) / 2.0f;
  const float p_t = (rho_t + 0.5f) / 2.0f;
  const bool reflect = Uniform(rng) < p_r;                 // one division for both outcomes (as in SampleLight)
  dir_in = reflect ? dir_r : dir_t;
  weight = ((reflect ? rho_r : rho_t) / (reflect ? p_r : p_t)) * rho;
}

// LightSet::GenerateRay (light_set.h:84-104) + Primitive::SampleSurfacePoint (primitive_*.cc) + HemispherePSA.
__device__ __forceinline__ void GenerateLightRay(const DevScene& sc, uint64_t& rng, V3& origin, V3& dir, V3& weight, int& origin_slot) {
  const float x = Uniform(rng) * sc.total_power;                       // prelude::Uniform(powers_.back(), sampler)
  uint32_t pos = 0;
  while (pos + 1u < sc.n_lights && sc.lights[pos].cum_power < x) ++pos; // std::lower_bound (clamped to the last light)
  const DevLight* L = sc.lights + pos;
  const uint32_t kind = L->kind;
  V3 normal;
  if (kind == PRIM_TRIANGLE) {                                          // primitive_triangle.cc:136-150
    float u = Uniform(rng), v = Uniform(rng);
    if (u + v >= 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    origin = (1.0f - u - v) * ld3(L->p) + u * ld3(L->p + 3) + v * ld3(L->p + 6);
    normal = ld3(L->p + 9);
  } else if (kind == PRIM_SPHERE) {                                     // primitive_sphere.cc:115-122, SphereSA sampling.h:185-199
    const float r0 = Uniform(rng) * (1.0f - (-1.0f)) + (-1.0f);
    const float r1 = Uniform(rng);
    const float sin_theta = Sqrt(1.0f - r0 * r0);
    float sp, cp; SinCos(2.0f * 3.14159274f * r1, sp, cp);
    normal = v3(r0 * cp, r0 * sp, sin_theta);
    origin = ld3(L->p) + L->p[3] * normal;
  } else if (kind == PRIM_DISK) {                                       // primitive_disk.cc:122-136
    const float radius = Sqrt(Uniform(rng) * (L->p[6] * L->p[6]));
    const V3 N = ld3(L->p + 3);
    V3 u, v; OrthonormalBasis(N, u, v);
    float ay, ax; SinCos(Uniform(rng) * 6.28318548f, ay, ax);           // Circle: theta = Uniform<T>(2 * kPI, sampler)
    origin = ld3(L->p) + (u * ax + v * ay) * radius;
    normal = N;
  } else {                                                              // primitive_cylinder.cc:150-164
    const float height = Uniform(rng) * L->p[7];
    const V3 N = ld3(L->p + 3);
    V3 u, v; OrthonormalBasis(N, u, v);
    float ay, ax; SinCos(Uniform(rng) * 6.28318548f, ay, ax);
    const V3 n = u * ax + v * ay;
    origin = ld3(L->p) + N * height + n * L->p[6];
    normal = Normalize(n);
  }
  dir = HemispherePSA(normal, rng);
  weight = ld3(L->irr) / L->pdf_area;                                   // object.Irradiance() / PDFArea(object)
  origin_slot = L->slot;
}

// The lens record (40 dwords) is needed once per path, not per bounce: it is read from constant memory next to its
// use.  The empty asm makes the pointer opaque per use, otherwise the compiler hoists the 40 scalar loads out of the
// persistent loop, keeps them live across the whole bounce loop and spills SGPRs to VGPR lanes in the hot code.
__device__ __forceinline__ DevLens LoadLens(const DevScene& sc) {
  ConstWords w = (ConstWords)(sc.lens);
  asm volatile("" : "+s"(w));
  union { DevLens lens; uint32_t words[sizeof(DevLens) / 4]; } u;
#pragma unroll
  for (unsigned k = 0; k < sizeof(DevLens) / 4; ++k) u.words[k] = w[k];
  return u.lens;
}

// Lens::Response for Ray(position, direction_out) (scene/scene.h:299-307, lens_thin.cc:109-130, lens_pinhole.cc:70-85,
// Sensor::ResponsePixel sensor.cc:46-59).  Returns false when the ray does not reach the sensor.
__device__ __forceinline__ bool LensResponse(const DevScene& sc, V3 position, V3 direction_out, uint32_t& pixel, float& value) {
  const DevLens L = LoadLens(sc);
  const V3 direction = MatMul(L.local_, direction_out);
  float sx, sy;
  if (L.kind == 1u) {
    const V3 point = (L.sensor_distance / direction.z) * direction;
    sx = point.x; sy = point.y; value = 1.0f;
  } else {
    if (direction.z >= 0.0f) return false;
    const V3 aperture_point = MatMul(L.local_, position - ld3(L.origin));
    const V3 sensor_point = L.neg_sd_over_fd * aperture_point + (L.sensor_distance / direction.z) * direction;
    sx = sensor_point.x; sy = sensor_point.y;
    value = static_cast<float>(Pow4(Normalize(sensor_point - aperture_point).z / direction.z));
  }
  const float uvx = sx / sc.sensor.sw + 0.5f, uvy = sy / sc.sensor.sh + 0.5f;
  const float mn = uvy < uvx ? uvy : uvx, mx = uvx < uvy ? uvy : uvx;  // std::min / std::max of (x, y)
  if (mn < 0.0f || mx >= 1.0f) return false;
  uint32_t ix = static_cast<uint32_t>(uvx * sc.sensor.wf), iy = static_cast<uint32_t>(uvy * sc.sensor.hf);
  if (ix > sc.sensor.w - 1u) ix = sc.sensor.w - 1u;
  if (iy > sc.sensor.h - 1u) iy = sc.sensor.h - 1u;
  pixel = ix + iy * sc.sensor.w;
  return true;
}

// ---------------------------------------------------------------------------------------------
// eye ray: BasicThin::GenerateRay (lens_thin.cc:70-107) + Sensor::PixelBound::Uniform
// (sensor.cc:111-120, jitter draw order Y then X -- the g++ order the reference outputs were made with)
// ---------------------------------------------------------------------------------------------
// near_edge (optional): the aperture sample lies within DevLens.edge_tol (barycentric) of its blade's boundary -- only then can the exact
// test of ANOTHER blade accept the ray's own origin (pt_megakernel's primary rounds: which blades are candidates).
__device__ __forceinline__ void GenerateEyeRay(const DevScene& sc, uint32_t px, uint32_t py, uint64_t& rng,
                                               V3& origin, V3& dir, float& weight, int& origin_slot, bool* near_edge = nullptr) {
  const DevLens L = LoadLens(sc);
  if (L.kind == 1u) {                                      // BasicPinhole::GenerateRay lens_pinhole.cc:48-68
    const float jy = Uniform(rng);
    const float jx = Uniform(rng);
    const float uvx = (static_cast<float>(px) + jx) / sc.sensor.wf;
    const float uvy = (static_cast<float>(py) + jy) / sc.sensor.hf;
    const V3 sensor_point = v3((uvx - 0.5f) * sc.sensor.sw, (uvy - 0.5f) * sc.sensor.sh, L.sensor_distance);
    const V3 ray_dir = Normalize(MatMul(L.global_, -sensor_point));
    // PDFDirection lens_pinhole.cc:93-106 (binary32 throughout; no sensor.Size() factor, unlike the thin lens)
    const V3 dl = MatMul(L.local_, ray_dir);
    const V3 point = (L.sensor_distance / dl.z) * dl;
    const float geometry_factor = dl.z * dl.z / SquaredLength(point);
    const float pdf_dir = L.inv_scene_area / geometry_factor;
    origin = ld3(L.origin); dir = ray_dir;
    weight = 1.0f / 1.0f / pdf_dir;                        // 1 / PDFArea (= kDiracDelta) / PDFDirection
    origin_slot = -1;
    if (near_edge) *near_edge = true;                      // (the pinhole's degenerate blade: keep every blade bit)
    return;
  }
  const float fpos = __builtin_floorf(Uniform(rng) * L.n_blades_f);
  uint32_t pos = static_cast<uint32_t>(fpos);
  if (pos > L.n_blades - 1) pos = L.n_blades - 1;
  const DevBlade* bl = sc.blades + pos;
  float u = Uniform(rng);
  float v = Uniform(rng);
  if (u + v >= 1.0f) { u = 1.0f - u; v = 1.0f - v; }
  if (near_edge) *near_edge = !(u >= L.edge_tol && v >= L.edge_tol && u + v <= 1.0f - L.edge_tol);
  const V3 ap_origin = (1.0f - u - v) * ld3(bl->v0) + u * ld3(bl->v1) + v * ld3(bl->v2);   // primitive_triangle.cc:136-150
  const V3 aperture_point = MatMul(L.local_, ap_origin - ld3(L.origin));
  const float jy = Uniform(rng);
  const float jx = Uniform(rng);
  const float uvx = (static_cast<float>(px) + jx) / sc.sensor.wf;
  const float uvy = (static_cast<float>(py) + jy) / sc.sensor.hf;
  const V3 sensor_point = v3((uvx - 0.5f) * sc.sensor.sw, (uvy - 0.5f) * sc.sensor.sh, L.sensor_distance);
  const V3 direction = Normalize(L.neg_fd_over_sd * sensor_point - aperture_point);
  const double factor = Pow4(Normalize(sensor_point - aperture_point).z / direction.z);
  const V3 ray_dir = Normalize(MatMul(L.global_, direction));
  const V3 dloc = MatMul(L.local_, ray_dir);
  const float pdf_dir = static_cast<float>(static_cast<double>(L.size_over_area) * L.sd2 / Pow4(dloc.z));
  origin = ap_origin; dir = ray_dir;
  origin_slot = bl->slot;                                  // the eye ray starts ON this aperture triangle
  weight = static_cast<float>(factor / static_cast<double>(L.p_area) / static_cast<double>(pdf_dir));
}

// ---------------------------------------------------------------------------------------------
// one bounce of PathTracing::Thread::Render (algorithm_pt.cc:137-157).
// Returns true if the path continues (o, d, weight updated), false if it ended.
// ---------------------------------------------------------------------------------------------
struct Bounce { int object; float t; V3 pos; V3 weight_before; };

// kLight = false: PathTracing::Thread::Render (algorithm_pt.cc:137-157).  kLight = true: LightTracing::Thread::Render
// (algorithm_lt.cc:134-162): a hit on an Eye surface splats weight * response / image.Size() instead of collecting
// emitted radiance, and the material is sampled with SampleImportance.
struct SplatSink { DevSplat* records; unsigned int* count; uint32_t capacity; uint32_t path, sample; float size_f; };

template <bool kTrace, int kEngine, bool kLight>
__device__ __forceinline__ bool PathShade(const DevScene& sc, const DevObject* lds_objects, const HitRec& h, V3& o, V3& d, V3& weight, V3& measurement,
                                          uint64_t& rng, uint32_t& casts, int& origin_slot, Bounce* trace AMBER_STAMP_PARAM, const SplatSink* sink);

template <bool kTrace, int kEngine, bool kLight = false>
__device__ __forceinline__ bool PathStep(const DevScene& sc, const DevObject* lds_objects, int32_t* lds_stack, V3& o, V3& d, V3& weight, V3& measurement,
                                         uint64_t& rng, uint32_t& casts, int& origin_slot, Bounce* trace AMBER_STAMP_PARAM, const SplatSink* sink = nullptr,
                                         const bool use_premask = false, const uint32_t premask = 0u, const int bvh_stack_cap = AMBER_BVH_STACK) {
  HitRec h;
  ClosestHit<kEngine>(sc, lds_objects, lds_stack, o, d, origin_slot, h AMBER_STAMP_ARG, use_premask, premask, bvh_stack_cap);
  return PathShade<kTrace, kEngine, kLight>(sc, lds_objects, h, o, d, weight, measurement, rng, casts, origin_slot, trace AMBER_STAMP_ARG, sink);
}

// Everything of a bounce after the closest-hit query (algorithm_pt.cc:140-157): h is the result of Scene::Cast.
template <bool kTrace, int kEngine, bool kLight>
__device__ __forceinline__ bool PathShade(const DevScene& sc, const DevObject* lds_objects, const HitRec& h, V3& o, V3& d, V3& weight, V3& measurement,
                                          uint64_t& rng, uint32_t& casts, int& origin_slot, Bounce* trace AMBER_STAMP_PARAM, const SplatSink* sink) {
  casts++;
  if (h.idx < 0) {
    if (kTrace) { trace->object = -1; trace->t = __builtin_nanf(""); trace->pos = v3(0, 0, 0); trace->weight_before = v3(0, 0, 0); }
    return false;
  }
  V3 pos, normal; uint32_t mat;
  ResolveHit<kEngine == ENGINE_TWO_PHASE_N ? 0x7fu : 0xffu>((kEngine == ENGINE_TWO_PHASE || kEngine == ENGINE_TWO_PHASE_N) ? lds_objects : ((kEngine == ENGINE_BVH || kEngine == ENGINE_REF_BVH) ? sc.bvh_objects : sc.objects), h, o, d, pos, normal, mat);
  const DevMaterial m = sc.materials[mat];
  const V3 dir_out = -d;
  if (kTrace) { trace->object = h.idx; trace->t = h.t; trace->pos = pos; trace->weight_before = weight; }
  V3 dir_in, sw;
  if (kLight) {
    if (m.kind == MAT_EYE) {                                                 // algorithm_lt.cc:141-147
      uint32_t pixel; float value;
      if (LensResponse(sc, pos, dir_out, pixel, value)) {
        const V3 add = (weight * value) / sink->size_f;
        const unsigned int k = atomicAdd(sink->count, 1u);
        if (k < sink->capacity) {
          DevSplat& r = sink->records[k];
          r.path = sink->path; r.sample = sink->sample; r.bounce = casts; r.pixel = pixel;
          r.rgb[0] = add.x; r.rgb[1] = add.y; r.rgb[2] = add.z; r.pad = 0u;
        }
      }
    }
    AMBER_STAMP(4);
    SampleImportance(m, normal, dir_out, rng, dir_in, sw);
  } else {
    measurement = measurement + weight * Radiance(m, normal, dir_out);      // algorithm_pt.cc:144
    AMBER_STAMP(4);
    SampleLight(m, normal, dir_out, rng, dir_in, sw);                        // :145-146
  }
  AMBER_STAMP(5);
  float p_rr = 0.9375f;                                                      // std::min<real_type>(kRussianRoulette, Max(w)) :148-149
  const float mw = Max3(sw);
  if (mw < p_rr) p_rr = mw;
  if (Uniform(rng) >= p_rr) return false;                                    // :151-153
  if (sc.max_depth && casts >= sc.max_depth) return false;                   // build-side extension (BASELINE config 5)
  o = pos; d = dir_in;                                                       // :155 Ray(pos, UnitVector3) -- no renormalisation
  if (kEngine == ENGINE_TWO_PHASE_N) origin_slot = (lds_objects[h.slot].kind & 0x80u) ? h.slot : -1;      // a filtered triangle of its group (bit 7 of the LDS record)
  else origin_slot = (kEngine == ENGINE_TWO_PHASE && static_cast<uint32_t>(h.slot) < sc.n_prog_tris) ? h.slot : -1;
#if AMBER_SHARED_WEIGHT_QUOTIENT
  {
    // :156  weight *= scatter.Weight() / p -- three binary32 divisions by the same p.  A component of the scatter weight that has the
    // bits of the first one has the first one's quotient, and +0 / p is +0 (p > 0 here: a path with p = 0 has ended above): grey, white,
    // mirror, glass and single-channel materials need ONE division.  The other two run only if some lane of the wave holds a weight
    // that is neither (wave-uniform branch; the same IEEE quotients either way).
    const uint32_t bx = __float_as_uint(sw.x), by = __float_as_uint(sw.y), bz = __float_as_uint(sw.z);
    const float qx = sw.x / p_rr;
    float qy = by == bx ? qx : 0.0f, qz = bz == bx ? qx : 0.0f;
    const bool hard_y = !(by == bx || (by == 0u && p_rr > 0.0f)), hard_z = !(bz == bx || (bz == 0u && p_rr > 0.0f));
    if (__ballot(hard_y || hard_z) != 0ull) {
      if (hard_y) qy = sw.y / p_rr;
      if (hard_z) qz = sw.z / p_rr;
    }
    weight = weight * v3(qx, qy, qz);
  }
#else
  weight = weight * (sw / p_rr);                                             // :156
#endif
  return true;
}

}  // namespace amber_dev
