// pt_host.hip -- the ONE translation unit of the gfx950 path-tracing engine: the handle, the host side of every launch and the C ABI
// (include/amber_hip.h).  The kernels live in files of their own and are included here, where they are launched:
//   pt_args.h              RenderArgs; record emission (accumulation without owners)
//   pt_megakernel.inc      pt_megakernel: persistent waves, work unit = ONE PATH (q = band pixel * n_samples + sample), a wave claims 1024 paths with
//                          one atomicAdd; lanes are decoupled from pixels through the wave's LDS pool; 64 fresh paths of one pixel start together
//                          (primary round, candidates from the per-pixel masks).  Engines LIST / TWO_PHASE (<= 32 objects; groups of 32 up to 128), engine BVH on trees of depth <= 12
//                          (one-shot per-lane traversal, stack in LDS), light tracing.
//   pt_bvh_megakernel.inc  pt_bvh_megakernel: engine BVH on deep trees -- lanes own (pixel, chunk of 8 samples) items, the closest hit is a RESUMABLE
//                          per-lane traversal of the 2-wide binary16-plane tree advanced in wave rounds until a batch of lanes has finished; per-item
//                          sums, reduce_partials_kernel.
//   pt_records.inc         records {q, rgb} -> path order -> the per-pixel sums of the numerical contract (rec_rank / scan / place, reduce_flagged);
//                          pixel_mask_kernel (candidates of a pixel block's eye rays).
// LAB BUILD (-DAMBER_LAB -> libamber_hip_lab.so; include/amber_hip_lab.h): the schedulers that were measured and lost but stay provably equal
// (bvh_pool.inc, wavefront.inc, bvh_stream.inc), the known-answer kernels and entry points the tests use (lab_kernels.inc, lab_api.inc) and the
// signature instantiations of the product kernels.  libamber_hip.so is built WITHOUT it and exports only the documented ABI.
//
// Replaces: PathTracing<RGB>::Thread::operator() / Render
//           (/root/reference/src/amber/rendering/algorithm_pt.cc:112-160).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/amber_hip.h"
#ifdef AMBER_LAB
#include "../../../include/amber_hip_lab.h"
#endif
#include "pt_device.h"
#include "bvh_build.h"
#include "filter_build.h"
#include "ref_bvh_build.h"

using namespace amber_dev;

namespace {

thread_local std::string g_last_error;

int Fail(int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return Fail(AMBER_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

// ------------------------------------------------------------------------------------------------
// render kernel
// ------------------------------------------------------------------------------------------------
#include "pt_args.h"
#include "pt_megakernel.inc"
#include "pt_records.inc"
#include "pt_bvh_megakernel.inc"
#ifdef AMBER_LAB
#include "bvh_pool.inc"
#include "bvh_stream.inc"
#include "wavefront.inc"
#include "lab_kernels.inc"
#endif

}  // namespace

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct amber_hip_pt {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  DevScene scene{};
  DevObject* d_objects = nullptr;
  DevMaterial* d_materials = nullptr;
  DevBlade* d_blades = nullptr;
  DevPlane* d_planes = nullptr;
  DevTriFilter* d_tri_filters = nullptr;
  DevSphereFilter* d_sphere_filters = nullptr;
  DevObject* d_prog_objects = nullptr;
  DevFilterGroup* d_groups = nullptr;       // engine TWO_PHASE_N: one record per group of 32 objects
  DevBvhNodeQ* d_bvh_nodes = nullptr;
  DevBvhNodeQ4* d_bvh_nodes4 = nullptr;      // AMBER_BVH_WIDE builds only
  float4* d_bvh_spheres = nullptr;
  float4* d_bvh_tris = nullptr;
  uint32_t* d_bvh_prims = nullptr;
  DevObject* d_bvh_objects = nullptr;
  DevRefNode* d_ref_nodes = nullptr;        // engine REFERENCE_BVH: the reference's own tree (ref_bvh_build.h) ...
  DevRefLeaf* d_ref_leaves = nullptr;
  uint2* d_ref_stack = nullptr;             // ... and its traversal stack, [level][thread of the largest grid]
  bool two_phase = false;
  uint32_t bvh_depth = 0;                   // depth of the flattened tree (selects the traversal-stack size)
  float* d_fb = nullptr;
  unsigned long long* d_rays = nullptr;
  unsigned int* d_next = nullptr;
  unsigned long long* d_stamps = nullptr;
  DevLight* d_lights = nullptr;
  DevLens* d_lens = nullptr;
  DevSplat* d_splats = nullptr;
  unsigned int* d_splat_count = nullptr;
  uint32_t splat_capacity = 0;
  uint64_t hashed_seed_lt = 0;
  float* d_partial = nullptr;               // pt_bvh_megakernel: per-item sums
  size_t partial_floats = 0;
  bool bvh_pool = false;                    // engine BVH renders with pt_bvh_pool_kernel (AMBER_PT_FLAG_BVH_POOL / AMBER_BVH_POOL=1) instead of pt_bvh_megakernel
  uint32_t bvh_shade_batch = AMBER_BVH_SHADE_BATCH;   // pt_bvh_megakernel's shading batch for this scene (BvhShadeBatch)
  bool bvh_paths = false;                   // engine BVH on a shallow tree (depth <= AMBER_PATH_BVH_STACK): pt_megakernel<ENGINE_BVH>, the path-granular scheduler (AMBER_BVH_PATHS=0/1 overrides)
  // path-granular accumulation (RenderPassPaths): bitmap, records in arrival order, measurements in path order, ranks
  uint32_t* d_flags = nullptr;  size_t flag_words = 0;   bool flags_dirty = true;   // dirty: must be cleared before the next launch
  uint32_t* d_touched = nullptr; size_t touched_words = 0;
  uint4* d_records = nullptr;   float* d_sorted = nullptr;   uint32_t rec_capacity = 0;
  unsigned int* d_launch_ctl = nullptr;      // [0] queue head of the path kernels, [1] = *d_rec_count, [2..3] = *d_rays_launch
  unsigned int* d_rec_count = nullptr;
  unsigned long long* d_rays_launch = nullptr;
  uint32_t* d_excl = nullptr;   uint32_t* d_block_sum = nullptr;   uint32_t rank_pixels = 0;
  unsigned int* h_rec_count = nullptr;      // pinned: the record counter of the launch in flight
  hipEvent_t pending_event = nullptr;
  bool pending = false, pending_checked = true;   // a launch whose record counter has not been looked at yet; checked = it cannot have run out of slots
  uint32_t pending_first = 0, pending_n = 0;
  bool density_known = false;
  double test_density_scale = 0;            // AMBER_TEST_RECORD_DENSITY_SCALE, read once at create (0 = off): a test hook that mis-sizes the record buffer
  double rec_density = 0;                   // record slots used per path, as the last launch measured it
  int32_t* d_bvh_stack = nullptr;  size_t bvh_stack_ints = 0;     // pt_bvh_pool_kernel: deep traversal-stack levels
  float* d_carried = nullptr;      size_t carried_floats = 0;     // ... and carried measurements
  unsigned long long* d_sig = nullptr;  uint64_t sig_paths = 0;   // amber_hip_pt_signatures
  uint32_t* d_pixel_mask = nullptr;  bool pixel_mask_on = true;   // two-phase engine: primary-ray candidates per band pixel
  // pixel_mask_kernel is enqueued on the render stream by create (0.07 ms since the masks are per 4 x 4 block of pixels: it no longer pays to
  // overlap it -- a second stream costs a millisecond of host time to create, more than the kernel it would hide)
  bool pixel_mask_ready = false;            // the kernel has been enqueued in front of everything that reads d_pixel_mask
  std::vector<uint32_t> prog_order;         // two-phase engine: scene index of the object in filter-program slot k (the bit positions of the masks)
  std::vector<DevPlane> host_planes;        // ... and its plane records (pixel_mask_kernel's wave-uniform tests are made on the host)
  uint32_t lens_kind = 0; float lens_sensor_distance = 0, lens_focus_distance = 0, lens_origin[3] = {0, 0, 0};   // host copies of the lens constants pixel_mask_kernel's arguments derive from
  float aperture_rect[4][3] = {};           // world corners of the blades' bounding rectangle in the lens plane (pixel_mask_kernel)
  int n_cus = 256;
  uint32_t row_begin = 0, row_end = 0, stripe_rows = 0, stripe_period = 0, local_rows = 0;
  uint64_t seed = 0, hashed_seed = 0;
  uint32_t engine = AMBER_ENGINE_LIST;      // as requested / resolved: LIST, TWO_PHASE, BVH or WAVEFRONT
  uint32_t hit_engine = AMBER_ENGINE_LIST;  // closest-hit engine the kernels are instantiated with
  float* d_wf = nullptr; size_t wf_bytes = 0;   // WAVEFRONT: queues + meas + counts in one allocation
  uint32_t n_materials = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // pool of event pairs, one per timed launch in flight
  size_t events_used = 0;
  uint32_t timed_launches = 0;     // launches already folded into timed_ms
  double timed_ms = 0;
};

namespace {

uint64_t HostSplitMix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, (n ? n : 1) * sizeof(T)); }
};

int StartPixelMasks(amber_hip_pt* h, float* timing);      // defined with the launch code below
uint32_t ResidentBlocksPerCu(uint32_t hit_engine, uint32_t bvh_depth);

constexpr uint32_t kHitTwoPhaseN = 5;                // amber_hip_pt.hit_engine: the two-phase engine over groups of 32 objects (device: ENGINE_TWO_PHASE_N); not a public engine id
constexpr uint32_t kTwoPhaseAutoObjects = 80;        // AUTO picks the grouped two-phase engine up to this many objects: the Cornell box plus small quads, 1024^2 @ 128 spp
                                                     // (tools/object_count_curve.py, profiles/r05_object_count_curve.txt): 33 objects 8.5 ms against 16.5 for engine BVH, 64: 13.7 / 16.9,
                                                     // 73: 15.1 / 16.9, 89: 17.6 / 15.5 -- the curves cross near 80

int ValidateScene(const AmberFlatScene* s, const AmberSensor* sensor) {
  if (!s || !sensor) return Fail(AMBER_EINVAL, "null scene or sensor");
  if (!s->objects || s->n_objects == 0) return Fail(AMBER_EINVAL, "scene has no objects");
  if (s->n_objects >= (1u << 26)) return Fail(AMBER_EINVAL, "too many objects (engine BVH addresses its 48-byte leaf records with 32-bit byte offsets: fewer than 2^26 objects)");
  if (!s->materials || s->n_materials == 0) return Fail(AMBER_EINVAL, "scene has no materials");
  if (sensor->width == 0 || sensor->height == 0) return Fail(AMBER_EINVAL, "empty sensor");
  if (static_cast<uint64_t>(sensor->width) * sensor->height >= (1ull << 32)) return Fail(AMBER_EINVAL, "sensor too large");
  for (uint32_t i = 0; i < s->n_objects; i++) {
    if (s->objects[i].kind > AMBER_PRIM_CYLINDER) return Fail(AMBER_EINVAL, "object " + std::to_string(i) + ": unknown primitive kind");
    if (s->objects[i].material >= s->n_materials) return Fail(AMBER_EINVAL, "object " + std::to_string(i) + ": material index out of range");
  }
  for (uint32_t i = 0; i < s->n_materials; i++)
    if (s->materials[i].kind > AMBER_MAT_EYE) return Fail(AMBER_EINVAL, "material " + std::to_string(i) + ": unknown kind");
  if (s->n_lights && !s->lights) return Fail(AMBER_EINVAL, "n_lights > 0 but lights is null");
  for (uint32_t i = 0; i < s->n_lights; i++)
    if (s->lights[i].object >= s->n_objects) return Fail(AMBER_EINVAL, "light object index out of range");
  const AmberFlatThinLens& L = s->lens;
  if (L.n_blades == 0) return Fail(AMBER_EINVAL, "lens has no aperture blades");
  if (L.kind > AMBER_LENS_PINHOLE) return Fail(AMBER_EINVAL, "unknown lens kind");
  if (static_cast<uint64_t>(L.first_blade_object) + L.n_blades > s->n_objects) return Fail(AMBER_EINVAL, "aperture blade objects out of range");
  for (uint32_t i = 0; i < L.n_blades; i++)
    if (s->objects[L.first_blade_object + i].kind != AMBER_PRIM_TRIANGLE) return Fail(AMBER_EINVAL, "aperture blade is not a triangle");
  return AMBER_OK;
}

}  // namespace

extern "C" {

const char* amber_hip_last_error(void) { return g_last_error.c_str(); }
int amber_hip_abi_version(void) { return AMBER_HIP_ABI_VERSION; }
int amber_hip_math_mode(void) { return AMBER_MATH_MODE; }
int amber_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int amber_hip_pt_create(const AmberFlatScene* s, const AmberSensor* sensor, const AmberPtParams* params, amber_hip_pt** out) {
  if (!out || !params) return Fail(AMBER_EINVAL, "null argument");
  *out = nullptr;
  int rc = ValidateScene(s, sensor);
  if (rc != AMBER_OK) return rc;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return Fail(AMBER_ENODEVICE, "no HIP device available (this engine has no CPU fallback)");
  if (params->device < 0 || params->device >= n_dev) return Fail(AMBER_ENODEVICE, "device ordinal out of range");
  uint32_t rb = params->row_begin, re = params->row_end;
  if (rb == 0 && re == 0) re = sensor->height;
  // rb == re (other than 0,0) is an EMPTY band: a rank beyond the number of stripes (distributed.stripe_partition) still
  // creates a handle, whose render_pass / clear / download are no-ops, so that it can take part in the gather
  if (rb > re || re > sensor->height) return Fail(AMBER_EINVAL, "bad row band");
  uint32_t local_rows = re - rb;
  if (params->stripe_rows) {
    if (params->stripe_period < params->stripe_rows) return Fail(AMBER_EINVAL, "stripe_period must be >= stripe_rows");
    const uint32_t q = (re - rb) / params->stripe_period, rem = (re - rb) % params->stripe_period;
    local_rows = q * params->stripe_rows + (rem < params->stripe_rows ? rem : params->stripe_rows);
  }
  if (params->engine > AMBER_ENGINE_REFERENCE_BVH || params->engine == 5u) return Fail(AMBER_EINVAL, "unknown engine");
#ifndef AMBER_LAB
  if (params->engine == AMBER_ENGINE_WAVEFRONT) return Fail(AMBER_EINVAL, "engine WAVEFRONT (the streaming formulation, kept for measurement) is part of the lab build, libamber_hip_lab.so");
  if (params->reserved & AMBER_PT_FLAG_BVH_POOL) return Fail(AMBER_EINVAL, "AMBER_PT_FLAG_BVH_POOL (pt_bvh_pool_kernel, kept for measurement) is part of the lab build, libamber_hip_lab.so");
#endif
  if (params->engine == AMBER_ENGINE_REFERENCE_BVH && !amber_refbvh::CentresAreOrdered(s->objects, s->n_objects))
    return Fail(AMBER_EINVAL, "AMBER_ENGINE_REFERENCE_BVH: an object's centre is NaN (the reference's build sorts objects by centre; std::sort is undefined on NaN keys)");
  if (params->engine == AMBER_ENGINE_TWO_PHASE && s->n_objects > AMBER_MAX_GROUP_OBJECTS)
    return Fail(AMBER_EINVAL, "AMBER_ENGINE_TWO_PHASE supports at most 128 objects");

  HIP_TRY(hipSetDevice(params->device));
  auto* h = new amber_hip_pt();
  h->device = params->device;
  h->row_begin = rb; h->row_end = re; h->local_rows = local_rows;
  h->stripe_rows = params->stripe_rows; h->stripe_period = params->stripe_rows ? params->stripe_period : 0;
  h->seed = params->seed; h->hashed_seed = HostSplitMix64(params->seed);
  h->hashed_seed_lt = HostSplitMix64(params->seed + 0x6C74ull);     // light paths use streams of their own
  if (params->stream) { h->stream = static_cast<hipStream_t>(params->stream); }
  else if (params->reserved & AMBER_PT_FLAG_NULL_STREAM) { h->stream = nullptr; }     // the legacy default stream, on request
  else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return Fail(AMBER_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    h->own_stream = true;
  }

  // ---- flatten -> device layout
  std::vector<DevObject> objs(s->n_objects);
  for (uint32_t i = 0; i < s->n_objects; i++) {
    const AmberFlatObject& f = s->objects[i];
    DevObject& o = objs[i];
    std::memset(&o, 0, sizeof o);
    o.kind = f.kind; o.material = f.material;
    o.a[0] = f.p[0]; o.a[1] = f.p[1]; o.a[2] = f.p[2];
    if (f.kind == AMBER_PRIM_TRIANGLE) {
      // E1 = v1 - v0, E2 = v2 - v0 (primitive_triangle.cc:100-101): same binary32 subtraction the
      // reference performs per intersection, hoisted to scene upload.
      for (int c = 0; c < 3; c++) {
        volatile float e1 = f.p[3 + c] - f.p[c];
        volatile float e2 = f.p[6 + c] - f.p[c];
        o.e1[c] = e1; o.e2[c] = e2; o.n[c] = f.p[9 + c];
      }
    } else if (f.kind == AMBER_PRIM_SPHERE) {
      o.radius = f.p[3];
    } else {
      o.e1[0] = f.p[3]; o.e1[1] = f.p[4]; o.e1[2] = f.p[5]; o.radius = f.p[6]; o.height = f.p[7];
    }
  }
  std::vector<DevMaterial> mats(s->n_materials);
  for (uint32_t i = 0; i < s->n_materials; i++) {
    const AmberFlatMaterial& f = s->materials[i];
    DevMaterial& m = mats[i];
    std::memset(&m, 0, sizeof m);
    m.kind = f.kind; m.rho[0] = f.rho[0]; m.rho[1] = f.rho[1]; m.rho[2] = f.rho[2]; m.param = f.param; m.r0 = f.r0;
    // constants the reference recomputes for every sample (material_phong.cc:92-105, material_refraction.cc:181-183): the same
    // binary32 operations, once (volatile: no wider intermediate, no reassociation)
    volatile float e1 = f.param + 1.0f, e2 = f.param + 2.0f;
    if (f.kind == AMBER_MAT_PHONG) { volatile float a = 1.0f / e1, b = e2 / e1; m.aux0 = a; m.aux1 = b; }
    else if (f.kind == AMBER_MAT_REFRACTION) { volatile float a = 1.0f / f.param; m.aux0 = a; }
  }
  const AmberFlatThinLens& L = s->lens;
  std::vector<DevBlade> blades(L.n_blades);
  for (uint32_t i = 0; i < L.n_blades; i++) {
    const AmberFlatObject& f = s->objects[L.first_blade_object + i];
    for (int c = 0; c < 3; c++) { blades[i].v0[c] = f.p[c]; blades[i].v1[c] = f.p[3 + c]; blades[i].v2[c] = f.p[6 + c]; blades[i].n[c] = f.p[9 + c]; }
    blades[i].slot = -1; blades[i].pad[0] = blades[i].pad[1] = blades[i].pad[2] = 0;
  }

  // AUTO: <= 32 objects the two-phase engine; up to kTwoPhaseAutoObjects its grouped form (one Phase-A program per 32 objects: cheaper than a per-lane
  // tree traversal while the groups are few -- tools/object_count_curve.py); beyond that engine BVH.  Asked for explicitly, two-phase takes up to 128 objects.
  uint32_t two_phase_auto = kTwoPhaseAutoObjects;
  { const char* ev = std::getenv("AMBER_TWO_PHASE_MAX_OBJECTS"); if (ev && std::atoi(ev) >= 0) two_phase_auto = std::min<uint32_t>(AMBER_MAX_GROUP_OBJECTS, static_cast<uint32_t>(std::atoi(ev))); }   // measurement hook
  const uint32_t auto_hit = s->n_objects <= AMBER_MAX_LDS_OBJECTS ? AMBER_ENGINE_TWO_PHASE : (s->n_objects <= two_phase_auto ? kHitTwoPhaseN : AMBER_ENGINE_BVH);
  h->engine = params->engine != AMBER_ENGINE_AUTO ? params->engine : (auto_hit == kHitTwoPhaseN ? static_cast<uint32_t>(AMBER_ENGINE_TWO_PHASE) : auto_hit);
  h->hit_engine = h->engine == AMBER_ENGINE_WAVEFRONT ? (s->n_objects <= AMBER_MAX_LDS_OBJECTS ? AMBER_ENGINE_TWO_PHASE : AMBER_ENGINE_BVH) : h->engine;
  if (h->hit_engine == AMBER_ENGINE_TWO_PHASE && s->n_objects > AMBER_MAX_LDS_OBJECTS) h->hit_engine = kHitTwoPhaseN;
  h->two_phase = h->hit_engine == AMBER_ENGINE_TWO_PHASE || h->hit_engine == kHitTwoPhaseN;
  // engine BVH has two schedulers with identical results: the default is the faster one on the 1M-sphere
  // scene (pt_bvh_megakernel, 74 ms at 64 spp against 79); the environment overrides the flag either way (A/B tools)
#ifdef AMBER_LAB
  h->bvh_pool = (params->reserved & AMBER_PT_FLAG_BVH_POOL) != 0u;
  { const char* ev = std::getenv("AMBER_BVH_POOL"); if (ev && (ev[0] == '0' || ev[0] == '1')) h->bvh_pool = ev[0] == '1'; }
#endif
  amber_bvh::FlatBvh bvh;
  if (h->hit_engine == AMBER_ENGINE_BVH) {
    bvh = amber_bvh::BuildBvh(objs);
    const bool debug_bvh = std::getenv("AMBER_DEBUG_BVH") != nullptr;
    auto report = [&](const char* what) {
      const amber_bvh::BvhQuality q = amber_bvh::MeasureBvh(bvh.nodes, bvh.root_ref);
      std::fprintf(stderr, "amber_hip: BVH %s: SAH inner-node term %.3f, leaf term %.3f (x objects %.3f), leaf volume / scene volume %.3f; %u inner nodes, %u leaves, %u levels\n",
                   what, q.inner_area, q.leaf_area, q.leaf_object_area, q.leaf_volume, q.inner, q.leaves, q.depth);
    };
    if (debug_bvh) report("as built");
    h->bvh_paths = !h->bvh_pool && !(params->reserved & AMBER_PT_FLAG_BVH_ITEMS) && h->engine != AMBER_ENGINE_WAVEFRONT && bvh.depth <= static_cast<uint32_t>(AMBER_PATH_BVH_STACK);
    // The shading batch of pt_bvh_megakernel.  While a wave collects finished lanes they idle through the rounds of the others, and a round
    // over triangle leaves costs about twice a round over sphere leaves (45 against 20 vector instructions per leaf object before any
    // root / quotient), so idle lanes are dearer in a mesh: tools/shade_batch_sweep.py (profiles/r05_shade_batch_sweep.txt) -- 1M spheres
    // best at 52 (49.7 ms at 64 spp; 40: 52.3), 1M-triangle terrain at 32 (62.1; 40: 63.8; 52: 68.5), 82k-triangle room at 36-44 (32.8; 52: 33.9).
    {
      size_t n_triangles = 0;                                   // (every scene has a few: the aperture blades)
      for (const DevObject& ob : objs) n_triangles += (ob.kind & 0xffu) == AMBER_PRIM_TRIANGLE ? 1u : 0u;
      h->bvh_shade_batch = 2u * n_triangles > objs.size() ? 40u : static_cast<uint32_t>(AMBER_BVH_SHADE_BATCH);   // a mesh: 40; mostly spheres (disks, cylinders): 52
    }
    { const char* ev = std::getenv("AMBER_BVH_SHADE_BATCH"); if (ev && std::atoi(ev) >= 1 && std::atoi(ev) <= 64) h->bvh_shade_batch = static_cast<uint32_t>(std::atoi(ev)); }   // measurement hook
    { const char* ev = std::getenv("AMBER_BVH_PATHS"); if (ev && ev[0] == '0') h->bvh_paths = false; }
    { const char* ev = std::getenv("AMBER_BVH_PATHS_MAX_DEPTH"); if (ev && static_cast<uint32_t>(std::atoi(ev)) < bvh.depth) h->bvh_paths = false; }   // measurement hook
    if (debug_bvh) std::fprintf(stderr, "amber_hip: BVH of %u objects: %zu nodes, depth %u; scheduler %s, shading batch %u\n", s->n_objects, bvh.nodes.size(), bvh.depth,
                                h->bvh_pool ? "pt_bvh_pool_kernel" : (h->bvh_paths ? "pt_megakernel<ENGINE_BVH>" : "pt_bvh_megakernel"), h->bvh_shade_batch);
  }
  amber_refbvh::FlatTree ref_tree;
  uint32_t ref_depth = 0;
  if (h->hit_engine == AMBER_ENGINE_REFERENCE_BVH) {
    const amber_refbvh::Tree tree = amber_refbvh::Build(s->objects, s->n_objects);
    if (tree.too_deep) { amber_hip_pt_destroy(h); return Fail(AMBER_EINVAL, "AMBER_ENGINE_REFERENCE_BVH: the reference's recursive build goes deeper than " + std::to_string(amber_refbvh::kMaxDepth) + " levels on this scene"); }
    ref_tree = amber_refbvh::Flatten(tree, s->objects);
    ref_depth = tree.depth;
    bvh.prim_index = tree.order;                               // the object arrays of engine BVH, in the reference's order
    if (std::getenv("AMBER_DEBUG_BVH"))
      std::fprintf(stderr, "amber_hip: reference BVH of %u objects: %u inner nodes, %u leaves (largest %u objects), depth %u\n", s->n_objects, tree.n_inner, tree.n_leaves, tree.largest_leaf, tree.depth);
  }
  amber_filter::FilterProgram fprog;
  float fp_center[3] = {0, 0, 0}, fp_reach = 0;
  {
    // model box of the two-phase filter: bounds of every object and of the lens, doubled
    double lo[3] = {L.origin[0], L.origin[1], L.origin[2]}, hi[3] = {L.origin[0], L.origin[1], L.origin[2]};
    for (const DevObject& ob : objs) {
      const amber_bvh::Box bx = amber_bvh::ObjectBox(ob);
      for (int c = 0; c < 3; c++) { lo[c] = std::min<double>(lo[c], bx.mn[c]); hi[c] = std::max<double>(hi[c], bx.mx[c]); }
    }
    double reach = 0;
    for (int c = 0; c < 3; c++) { fp_center[c] = static_cast<float>(0.5 * (lo[c] + hi[c])); reach = std::max(reach, 0.5 * (hi[c] - lo[c])); }
    fp_reach = static_cast<float>(std::min(3.0e38, 2.0 * reach + 1e-3));
  }
  std::vector<amber_filter::FilterProgram> more_progs;          // engine TWO_PHASE_N: the programs of groups 1, 2, ... (fprog is group 0's)
  if (h->hit_engine == kHitTwoPhaseN) {
    // groups of 32 in scene order, the aperture blades first (the primary rounds' masks and the blades' own slots live in group 0)
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < L.n_blades; i++) order.push_back(L.first_blade_object + i);
    for (uint32_t i = 0; i < s->n_objects; i++) if (i < L.first_blade_object || i >= L.first_blade_object + L.n_blades) order.push_back(i);
    for (size_t first = 0; first < order.size(); first += 32) {
      const std::vector<uint32_t> members(order.begin() + first, order.begin() + std::min(order.size(), first + 32));
      if (first == 0) amber_filter::BuildFilterProgram(objs, fp_center, fprog, &members);
      else { more_progs.emplace_back(); amber_filter::BuildFilterProgram(objs, fp_center, more_progs.back(), &members); }
    }
  } else if (h->two_phase) amber_filter::BuildFilterProgram(objs, fp_center, fprog);
  if (h->two_phase && std::getenv("AMBER_DEBUG_FILTER")) {     // diagnostic: shape of the Phase-A program
    uint32_t pairs = 0, singles = 0;
    uint32_t shared = 0;
    for (const DevPlane& pl : fprog.planes) { pairs += pl.n_pairs; singles += pl.n_tris & 0x7fffffffu; shared += pl.n_tris >> 31; }
    std::fprintf(stderr, "amber_hip: filter program%s: %zu planes (%u share the previous plane's normal), %u pair records, %u single records, %zu spheres, always mask %#x; %zu group(s) of <= 32 objects\n",
                 more_progs.empty() ? "" : " of group 0", fprog.planes.size(), shared, pairs, singles, fprog.spheres.size(), fprog.always_mask, more_progs.size() + 1);
  }
  h->prog_order = fprog.order;
  h->host_planes = fprog.planes;
  for (uint32_t i = 0; i < L.n_blades; i++)            // filter-program slot of every aperture blade (self-candidate trip)
    for (uint32_t k = 0; k < fprog.n_prog_tris; k++)
      if (fprog.order[k] == L.first_blade_object + i) blades[i].slot = static_cast<int32_t>(k);
  std::vector<DevLight> lights(s->n_lights);
  for (uint32_t i = 0; i < s->n_lights; i++) {
    const AmberFlatLight& fl = s->lights[i];
    const AmberFlatObject& fo = s->objects[fl.object];
    DevLight& dl = lights[i];
    std::memset(&dl, 0, sizeof dl);
    dl.kind = fo.kind; dl.slot = -1; dl.cum_power = fl.cum_power; dl.pdf_area = fl.pdf_area;
    for (int c = 0; c < 3; c++) dl.irr[c] = fl.irradiance[c];
    for (int c = 0; c < 12; c++) dl.p[c] = fo.p[c];
    for (uint32_t k = 0; k < fprog.n_prog_tris; k++)
      if (fprog.order[k] == fl.object) dl.slot = static_cast<int32_t>(k);
    for (size_t g = 0; g < more_progs.size(); g++)              // LDS slots of group g + 1 start at 32 (g + 1)
      for (uint32_t k = 0; k < more_progs[g].n_prog_tris; k++)
        if (more_progs[g].order[k] == fl.object) dl.slot = static_cast<int32_t>(32u * (g + 1) + k);
  }
  // every group's records behind each other; the LDS image: 32 slots per group, kind |= scene index << 8 | 0x80 for a filtered triangle
  std::vector<DevPlane> all_planes = fprog.planes;
  std::vector<DevTriFilter> all_tris = fprog.tris;
  std::vector<DevSphereFilter> all_spheres = fprog.spheres;
  std::vector<DevFilterGroup> groups;
  std::vector<DevObject> prog(more_progs.empty() ? fprog.order.size() : 32u * (more_progs.size() + 1));
  if (!prog.empty()) std::memset(prog.data(), 0, prog.size() * sizeof(DevObject));
  auto place = [&](const amber_filter::FilterProgram& fp, size_t base) {
    for (size_t k = 0; k < fp.order.size(); k++) { prog[base + k] = objs[fp.order[k]]; prog[base + k].kind |= fp.order[k] << 8 | (!more_progs.empty() && k < fp.n_prog_tris ? 0x80u : 0u); }   // (the flag only in the grouped engine's image)
  };
  place(fprog, 0);
  if (!more_progs.empty()) {
    auto record = [&](const amber_filter::FilterProgram& fp, size_t plane_first, size_t tri_first, size_t sphere_first) {
      DevFilterGroup g{};
      g.plane_first = static_cast<uint32_t>(plane_first); g.n_planes = static_cast<uint32_t>(fp.planes.size()); g.n_simple_planes = fp.n_simple_planes;
      g.tri_first = static_cast<uint32_t>(tri_first); g.sphere_first = static_cast<uint32_t>(sphere_first); g.n_sphere_filters = static_cast<uint32_t>(fp.spheres.size());
      g.always_mask = fp.always_mask; g.n_prog_tris = fp.n_prog_tris; g.n_objects = static_cast<uint32_t>(fp.order.size());
      groups.push_back(g);
    };
    record(fprog, 0, 0, 0);
    for (size_t g = 0; g < more_progs.size(); g++) {
      const amber_filter::FilterProgram& fp = more_progs[g];
      record(fp, all_planes.size(), all_tris.size(), all_spheres.size());
      all_planes.insert(all_planes.end(), fp.planes.begin(), fp.planes.end());
      all_tris.insert(all_tris.end(), fp.tris.begin(), fp.tris.end());
      all_spheres.insert(all_spheres.end(), fp.spheres.begin(), fp.spheres.end());
      place(fp, 32u * (g + 1));
    }
  }

  auto cleanup = [&](int code, const std::string& msg) { amber_hip_pt_destroy(h); return Fail(code, msg); };
#define HIP_TRY_H(expr)                                                                            \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(AMBER_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

  HIP_TRY_H(hipMalloc(&h->d_objects, objs.size() * sizeof(DevObject)));
  HIP_TRY_H(hipMalloc(&h->d_materials, mats.size() * sizeof(DevMaterial)));
  HIP_TRY_H(hipMalloc(&h->d_blades, blades.size() * sizeof(DevBlade)));
  HIP_TRY_H(hipMalloc(&h->d_planes, (all_planes.size() + 1) * sizeof(DevPlane)));
  HIP_TRY_H(hipMalloc(&h->d_tri_filters, (all_tris.size() + 1) * sizeof(DevTriFilter)));
  HIP_TRY_H(hipMalloc(&h->d_sphere_filters, (all_spheres.size() + 1) * sizeof(DevSphereFilter)));
  if (!all_planes.empty()) HIP_TRY_H(hipMemcpy(h->d_planes, all_planes.data(), all_planes.size() * sizeof(DevPlane), hipMemcpyHostToDevice));
  if (!all_tris.empty()) HIP_TRY_H(hipMemcpy(h->d_tri_filters, all_tris.data(), all_tris.size() * sizeof(DevTriFilter), hipMemcpyHostToDevice));
  HIP_TRY_H(hipMalloc(&h->d_prog_objects, (prog.size() + 1) * sizeof(DevObject)));
  if (!prog.empty()) HIP_TRY_H(hipMemcpy(h->d_prog_objects, prog.data(), prog.size() * sizeof(DevObject), hipMemcpyHostToDevice));
  if (!groups.empty()) {
    HIP_TRY_H(hipMalloc(&h->d_groups, groups.size() * sizeof(DevFilterGroup)));
    HIP_TRY_H(hipMemcpy(h->d_groups, groups.data(), groups.size() * sizeof(DevFilterGroup), hipMemcpyHostToDevice));
  }
  HIP_TRY_H(hipMalloc(&h->d_lights, (lights.size() + 1) * sizeof(DevLight)));
  if (!lights.empty()) HIP_TRY_H(hipMemcpy(h->d_lights, lights.data(), lights.size() * sizeof(DevLight), hipMemcpyHostToDevice));
  // engine BVH: quantised nodes, leaf-order permutation, object records and compact sphere records in leaf order
  amber_bvh::QuantizedBvh qbvh = amber_bvh::QuantizeBvh(bvh.nodes, bvh.root_ref, [&](uint32_t slot) { return objs[bvh.prim_index[slot]].kind & 0xffu; });
  HIP_TRY_H(hipMalloc(&h->d_bvh_nodes, (qbvh.nodes.size() + 1) * sizeof(DevBvhNodeQ)));
  HIP_TRY_H(hipMalloc(&h->d_bvh_prims, (bvh.prim_index.size() + 1) * sizeof(uint32_t)));
  if (!qbvh.nodes.empty()) HIP_TRY_H(hipMemcpy(h->d_bvh_nodes, qbvh.nodes.data(), qbvh.nodes.size() * sizeof(DevBvhNodeQ), hipMemcpyHostToDevice));
#if AMBER_BVH_WIDE
  {
    amber_bvh::QuantizedBvh4 q4 = amber_bvh::CollapseBvh4(bvh.nodes, bvh.root_ref, qbvh, [&](uint32_t slot) { return objs[bvh.prim_index[slot]].kind & 0xffu; });
    HIP_TRY_H(hipMalloc(&h->d_bvh_nodes4, (q4.nodes.size() + 1) * sizeof(DevBvhNodeQ4)));
    if (!q4.nodes.empty()) HIP_TRY_H(hipMemcpy(h->d_bvh_nodes4, q4.nodes.data(), q4.nodes.size() * sizeof(DevBvhNodeQ4), hipMemcpyHostToDevice));
    qbvh.root_ref = q4.root_ref;
  }
#endif
  if (!bvh.prim_index.empty()) HIP_TRY_H(hipMemcpy(h->d_bvh_prims, bvh.prim_index.data(), bvh.prim_index.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  {
    std::vector<DevObject> leaf_order(bvh.prim_index.size());
    std::vector<float4> leaf_spheres(bvh.prim_index.size());
    for (size_t k = 0; k < leaf_order.size(); k++) {
      const DevObject& ob = objs[bvh.prim_index[k]];
      leaf_order[k] = ob;
      leaf_spheres[k] = (ob.kind & 0xffu) == AMBER_PRIM_SPHERE ? make_float4(ob.a[0], ob.a[1], ob.a[2], ob.radius) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // compact triangle records of the leaves (IntersectTriangleLeaf): three float4 per leaf-order slot, only when the scene has triangles
    bool any_tri = false;
    for (const DevObject& ob : leaf_order) any_tri = any_tri || (ob.kind & 0xffu) == AMBER_PRIM_TRIANGLE;
    std::vector<float4> leaf_tris(any_tri ? 3 * leaf_order.size() : 0);
    for (size_t k = 0; any_tri && k < leaf_order.size(); k++) {
      const DevObject& ob = leaf_order[k];
      if ((ob.kind & 0xffu) != AMBER_PRIM_TRIANGLE) { leaf_tris[3 * k] = leaf_tris[3 * k + 1] = leaf_tris[3 * k + 2] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      float idx; const uint32_t scene_index = bvh.prim_index[k]; std::memcpy(&idx, &scene_index, 4);
      leaf_tris[3 * k] = make_float4(ob.a[0], ob.a[1], ob.a[2], ob.e1[0]);
      leaf_tris[3 * k + 1] = make_float4(ob.e1[1], ob.e1[2], ob.e2[0], ob.e2[1]);
      leaf_tris[3 * k + 2] = make_float4(ob.e2[2], idx, 0.f, 0.f);
    }
    HIP_TRY_H(hipMalloc(&h->d_bvh_tris, (leaf_tris.size() + 3) * sizeof(float4)));
    if (!leaf_tris.empty()) HIP_TRY_H(hipMemcpy(h->d_bvh_tris, leaf_tris.data(), leaf_tris.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY_H(hipMalloc(&h->d_bvh_objects, (leaf_order.size() + 1) * sizeof(DevObject)));
    HIP_TRY_H(hipMalloc(&h->d_bvh_spheres, (leaf_spheres.size() + 1) * sizeof(float4)));
    if (!leaf_order.empty()) HIP_TRY_H(hipMemcpy(h->d_bvh_objects, leaf_order.data(), leaf_order.size() * sizeof(DevObject), hipMemcpyHostToDevice));
    if (!leaf_spheres.empty()) HIP_TRY_H(hipMemcpy(h->d_bvh_spheres, leaf_spheres.data(), leaf_spheres.size() * sizeof(float4), hipMemcpyHostToDevice));
  }
  if (!all_spheres.empty()) HIP_TRY_H(hipMemcpy(h->d_sphere_filters, all_spheres.data(), all_spheres.size() * sizeof(DevSphereFilter), hipMemcpyHostToDevice));
  HIP_TRY_H(hipMemcpy(h->d_objects, objs.data(), objs.size() * sizeof(DevObject), hipMemcpyHostToDevice));
  HIP_TRY_H(hipMemcpy(h->d_materials, mats.data(), mats.size() * sizeof(DevMaterial), hipMemcpyHostToDevice));
  HIP_TRY_H(hipMemcpy(h->d_blades, blades.data(), blades.size() * sizeof(DevBlade), hipMemcpyHostToDevice));
  const size_t fb_floats = static_cast<size_t>(local_rows) * sensor->width * 3;
  HIP_TRY_H(hipMalloc(&h->d_fb, (fb_floats ? fb_floats : 1) * sizeof(float)));
  HIP_TRY_H(hipMalloc(&h->d_rays, sizeof(unsigned long long)));
  HIP_TRY_H(hipMalloc(&h->d_next, sizeof(unsigned int)));
#ifdef AMBER_STAMPS
  HIP_TRY_H(hipMalloc(&h->d_stamps, (8 + 4 * AMBER_WAVE_TIME_SLOTS) * sizeof(unsigned long long)));     // 8 section sums, then per wave: start, first claim done, queue empty, end
  HIP_TRY_H(hipMemset(h->d_stamps, 0, (8 + 4 * AMBER_WAVE_TIME_SLOTS) * sizeof(unsigned long long)));
#endif
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, params->device) == hipSuccess && v > 0) h->n_cus = v; }
  HIP_TRY_H(hipMemsetAsync(h->d_fb, 0, fb_floats * sizeof(float), h->stream));
  HIP_TRY_H(hipMemsetAsync(h->d_rays, 0, sizeof(unsigned long long), h->stream));

  if (h->hit_engine == AMBER_ENGINE_REFERENCE_BVH) {
    HIP_TRY_H(hipMalloc(&h->d_ref_nodes, (ref_tree.nodes.size() + 1) * sizeof(DevRefNode)));
    HIP_TRY_H(hipMalloc(&h->d_ref_leaves, (ref_tree.leaves.size() + 1) * sizeof(DevRefLeaf)));
    if (!ref_tree.nodes.empty()) HIP_TRY_H(hipMemcpy(h->d_ref_nodes, ref_tree.nodes.data(), ref_tree.nodes.size() * sizeof(DevRefNode), hipMemcpyHostToDevice));
    if (!ref_tree.leaves.empty()) HIP_TRY_H(hipMemcpy(h->d_ref_leaves, ref_tree.leaves.data(), ref_tree.leaves.size() * sizeof(DevRefLeaf), hipMemcpyHostToDevice));
    // one stack column per thread of the largest grid a launch of this handle uses; a walk from the root pushes at most one entry per level
    const uint64_t threads = static_cast<uint64_t>(h->n_cus) * ResidentBlocksPerCu(h->hit_engine, 0) * 256u;
    const uint64_t bytes = threads * (static_cast<uint64_t>(ref_depth) + 1u) * sizeof(uint2);
    if (bytes > (64ull << 30)) return cleanup(AMBER_ENOMEM, "the reference's BVH of this scene is " + std::to_string(ref_depth) + " levels deep: its traversal stacks would take " + std::to_string(bytes >> 30) + " GiB");
    HIP_TRY_H(hipMalloc(&h->d_ref_stack, bytes));
    h->scene.ref_stack_stride = static_cast<uint32_t>(threads);
  }
  DevScene& sc = h->scene;
  sc.objects = h->d_objects; sc.materials = h->d_materials; sc.blades = h->d_blades;
  sc.planes = h->d_planes; sc.tri_filters = h->d_tri_filters; sc.sphere_filters = h->d_sphere_filters;
  sc.n_planes = static_cast<uint32_t>(fprog.planes.size()); sc.n_simple_planes = fprog.n_simple_planes; sc.n_sphere_filters = static_cast<uint32_t>(fprog.spheres.size());
  sc.bvh_nodes = h->d_bvh_nodes; sc.bvh_nodes4 = h->d_bvh_nodes4; sc.bvh_prims = h->d_bvh_prims; sc.bvh_objects = h->d_bvh_objects; sc.bvh_spheres = h->d_bvh_spheres; sc.bvh_tris = h->d_bvh_tris; sc.bvh_root = h->hit_engine == AMBER_ENGINE_REFERENCE_BVH ? ref_tree.root : qbvh.root_ref;
  sc.ref_nodes = h->d_ref_nodes; sc.ref_leaves = h->d_ref_leaves; sc.ref_stack = h->d_ref_stack;
  for (int c = 0; c < 3; c++) { sc.bvh_gmin[c] = qbvh.gmin[c]; sc.bvh_step[c] = qbvh.step[c]; sc.bvh_reach[c] = qbvh.reach[c]; }
  {
    // per-ray box margin of engine BVH (BvhBegin): centre and half diagonal of the scene bounds, 1 / smallest sphere radius
    double d2 = 0;
    for (int c = 0; c < 3; c++) {
      sc.bvh_center[c] = 0.5f * (bvh.bounds_min[c] + bvh.bounds_max[c]);
      const double e = double(bvh.bounds_max[c]) - bvh.bounds_min[c];
      d2 += e * e;
    }
    sc.bvh_half_diag = static_cast<float>(0.5 * std::sqrt(d2) * 1.0001);
    sc.bvh_inv_rmin = !bvh.has_spheres ? 0.0f : (bvh.min_sphere_radius > 0 ? static_cast<float>(std::min(3.0e38, 1.0001 / bvh.min_sphere_radius)) : 3.0e38f);
  }
  for (int c = 0; c < 3; c++) sc.fp_center[c] = fp_center[c];
  sc.fp_reach = fp_reach;
  // origin within fp_reach (max norm) of the centre, objects within half of that: no two such points are farther apart than
  sc.fp_tmax = static_cast<float>(std::min(3.0e38, 1.7320508 * 1.5 * 1.01 * static_cast<double>(fp_reach)));
  sc.lights = h->d_lights; sc.n_lights = s->n_lights; sc.total_power = s->n_lights ? s->lights[s->n_lights - 1].cum_power : 0.0f;
  sc.n_prog_tris = fprog.n_prog_tris; sc.always_mask = fprog.always_mask; sc.prog_objects = h->d_prog_objects;
  sc.groups = h->d_groups; sc.n_groups = static_cast<uint32_t>(groups.empty() ? 1 : groups.size()); sc.n_lds_objects = static_cast<uint32_t>(prog.size());
  sc.blade_mask = 0u;
  for (const DevBlade& bl : blades) if (bl.slot >= 0 && bl.slot < 32) sc.blade_mask |= 1u << bl.slot;
  {
    // bounding rectangle of the aperture in the lens plane (lens-local x, y; the blades lie in z = 0), inflated, as four world points
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (const DevBlade& bl : blades)
      for (const float* v : {bl.v0, bl.v1, bl.v2}) {
        const double r[3] = {double(v[0]) - L.origin[0], double(v[1]) - L.origin[1], double(v[2]) - L.origin[2]};
        for (int c = 0; c < 2; c++) {
          const double x = L.local_[3 * c] * r[0] + L.local_[3 * c + 1] * r[1] + L.local_[3 * c + 2] * r[2];
          lo[c] = std::min(lo[c], x); hi[c] = std::max(hi[c], x);
        }
      }
    double world_mag = 0;
    for (int c = 0; c < 3; c++) world_mag = std::max({world_mag, std::fabs(double(fp_center[c])) + fp_reach, std::fabs(double(L.origin[c]))});
    for (int c = 0; c < 2; c++) { const double m = 1e-3 * (hi[c] - lo[c]) + 1e-6 + 1e-5 * fp_reach + 32.0 * 5.9604644775390625e-08 * world_mag; lo[c] -= m; hi[c] += m; }
    for (int i = 0; i < 4; i++) {
      const double x = (i & 1) ? hi[0] : lo[0], y = (i & 2) ? hi[1] : lo[1];
      for (int c = 0; c < 3; c++) h->aperture_rect[i][c] = static_cast<float>(L.origin[c] + (L.kind == AMBER_LENS_PINHOLE ? 0.0 : L.global_[3 * c] * x + L.global_[3 * c + 1] * y));
    }
  }
  { const char* ev = std::getenv("AMBER_PIXEL_MASK"); h->pixel_mask_on = !(ev && ev[0] == '0'); }
  if (const char* ts = std::getenv("AMBER_TEST_RECORD_DENSITY_SCALE")) h->test_density_scale = std::atof(ts);   // the environment is read at create only
  sc.n_objects = s->n_objects; sc.max_depth = params->max_depth;
  h->n_materials = s->n_materials;
  DevLens lens{};
  std::memcpy(lens.origin, L.origin, sizeof L.origin);
  std::memcpy(lens.global_, L.global_, sizeof L.global_);
  std::memcpy(lens.local_, L.local_, sizeof L.local_);
  lens.focus_distance = L.focus_distance; lens.sensor_distance = L.sensor_distance; lens.p_area = L.p_area;
  { volatile float q = -L.focus_distance / L.sensor_distance; lens.neg_fd_over_sd = q; }
  {
    // sensor.Size() / sensor.SceneArea(): uint -> float, float*float, float/float (lens_thin.cc:145, sensor.cc:40-50)
    volatile float size_f = static_cast<float>(static_cast<uint64_t>(sensor->width) * sensor->height);
    volatile float area = sensor->scene_width * sensor->scene_height;
    volatile float r = size_f / area;
    lens.size_over_area = r;
  }
  lens.sd2 = static_cast<double>(L.sensor_distance) * static_cast<double>(L.sensor_distance);
  lens.n_blades = L.n_blades; lens.n_blades_f = static_cast<float>(L.n_blades);
  lens.kind = L.kind;
  h->lens_kind = L.kind; h->lens_sensor_distance = L.sensor_distance; h->lens_focus_distance = L.focus_distance;
  for (int c = 0; c < 3; c++) h->lens_origin[c] = L.origin[c];
  { volatile float area = sensor->scene_width * sensor->scene_height; volatile float inv = 1.0f / area; lens.inv_scene_area = inv; }
  sc.sensor.w = sensor->width; sc.sensor.h = sensor->height;
  sc.sensor.wf = static_cast<float>(sensor->width); sc.sensor.hf = static_cast<float>(sensor->height);
  sc.sensor.sw = sensor->scene_width; sc.sensor.sh = sensor->scene_height;
  sc.sensor.size_f = static_cast<float>(static_cast<uint64_t>(sensor->width) * sensor->height);
  { volatile float q = -L.sensor_distance / L.focus_distance; lens.neg_sd_over_fd = q; }
  {
    // a ray that starts on blade b is seen by another blade's exact test only if its origin lies within the rounding of WORLD
    // coordinates of that blade: a few ulp of the lens position, in units of the blade's size
    double world_mag = 0, min_edge = 1e300;
    for (int c = 0; c < 3; c++) world_mag = std::max(world_mag, std::fabs(double(L.origin[c])));
    for (const DevBlade& bl : blades) {
      const float* v[3] = {bl.v0, bl.v1, bl.v2};
      for (int k = 0; k < 3; k++) {
        double e2 = 0;
        for (int c = 0; c < 3; c++) { const double e = double(v[k][c]) - v[(k + 1) % 3][c]; e2 += e * e; world_mag = std::max(world_mag, std::fabs(double(v[k][c]))); }
        min_edge = std::min(min_edge, std::sqrt(e2));
      }
    }
    const double tol = min_edge > 0 ? std::max(1e-3, 64.0 * 5.9604644775390625e-08 * world_mag / min_edge) : 1.0;
    lens.edge_tol = static_cast<float>(std::min(1.0, tol));
  }
  HIP_TRY_H(hipMalloc(&h->d_lens, sizeof(DevLens)));
  HIP_TRY_H(hipMemcpy(h->d_lens, &lens, sizeof(DevLens), hipMemcpyHostToDevice));
  sc.lens = h->d_lens;
  {
    const int rc_masks = StartPixelMasks(h, nullptr);          // asynchronous, on the render stream: in front of the handle's first launch
    if (rc_masks != AMBER_OK) { const std::string msg = g_last_error; amber_hip_pt_destroy(h); return Fail(rc_masks, msg); }
  }
  *out = h;
  return AMBER_OK;
}

namespace {
// Hands out the next event pair; when the pool of 64 is used up the finished launches are folded into the running
// totals (one stream synchronisation every 64 launches), so long renders (--spp 0 until expiry) do not grow the pool.
int AcquireEventPair(amber_hip_pt* h, std::pair<hipEvent_t, hipEvent_t>** out) {
  if (h->events_used == 64) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->events_used; i++) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
      h->timed_ms += ms;
    }
    h->timed_launches += static_cast<uint32_t>(h->events_used);
    h->events_used = 0;
  }
  if (h->events_used == h->events.size()) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    h->events.emplace_back(e0, e1);
  }
  *out = &h->events[h->events_used++];
  return AMBER_OK;
}
}  // namespace

namespace {

// Workgroups of the persistent kernels that fit a CU at once: pt_bvh_megakernel is bounded by its LDS traversal stacks
// and VGPRs (5 with 24-entry stacks, else 4); the others are capped to AMBER_MEGAKERNEL_WAVES_PER_SIMD by their launch bounds.
uint32_t ResidentBlocksPerCu(uint32_t hit_engine, uint32_t bvh_depth) {
  return hit_engine == AMBER_ENGINE_BVH ? (bvh_depth <= 24 ? static_cast<uint32_t>(AMBER_BVH_WGS) : 4u) : (AMBER_MEGAKERNEL_WAVES_PER_SIMD > 5 ? static_cast<uint32_t>(AMBER_MEGAKERNEL_WAVES_PER_SIMD) : 5u);   // uncapped: 87 VGPRs -> 5
}

}  // namespace

namespace {
// ---- path-granular launches (pt_megakernel for engines LIST / TWO_PHASE, pt_bvh_pool_kernel for engine BVH) --------------
// One launch = megakernel + rank / scan / place / reduce.  The record buffer is sized from the density the previous launch
// measured (record slots per path), with a margin; the very first launch of a handle is one accumulation chunk, whose buffer
// can hold a record for EVERY path, and doubles as the probe.  A launch that still runs out of slots leaves the framebuffer
// and the ray total untouched (the device-side kernels check the counter); the host notices when it next touches the handle
// (ResolvePending), grows the buffer and repeats exactly that launch, so results never depend on the sizing.  Launches are
// split on chunk boundaries, which leaves the summation order unchanged.
constexpr uint64_t kMaxPathsPerLaunch = 1ull << 30;       // q and its bitmap index stay 32-bit; bitmap 128 MiB
constexpr uint64_t kMaxRecordSlots = 48ull << 20;         // 28 B per slot (record + sorted measurement): 1.3 GiB at most

uint32_t PathBlocks(const amber_hip_pt* h, uint64_t n_paths) {
  uint32_t n_blocks = static_cast<uint32_t>(h->n_cus) * ResidentBlocksPerCu(h->hit_engine, h->bvh_depth);
#ifdef AMBER_LAB
  if (h->hit_engine == AMBER_ENGINE_BVH && !h->bvh_paths) n_blocks = static_cast<uint32_t>(h->n_cus) * static_cast<uint32_t>(AMBER_BVH_POOL_WGS);   // pt_bvh_pool_kernel
#endif
  const uint64_t by_work = (n_paths + 255u) / 256u;
  if (by_work < n_blocks) n_blocks = static_cast<uint32_t>(by_work);
  return n_blocks;
}
// record slots that stay unused because every wave reserves them AMBER_REC_BLOCK at a time, plus a floor
uint64_t RecordSlack(const amber_hip_pt* h) { return static_cast<uint64_t>(h->n_cus) * 8u * 4u * AMBER_REC_BLOCK + 4096u; }

int EnsureRecordCapacity(amber_hip_pt* h, uint64_t slots) {
  if (slots <= h->rec_capacity) return AMBER_OK;
  if (slots > 0xfffffff0ull) return Fail(AMBER_ENOMEM, "record buffer beyond 2^32 slots");
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->d_records) { HIP_TRY(hipFree(h->d_records)); h->d_records = nullptr; }
  if (h->d_sorted) { HIP_TRY(hipFree(h->d_sorted)); h->d_sorted = nullptr; }
  h->rec_capacity = 0;
  hipError_t e = hipMalloc(&h->d_records, slots * sizeof(uint4));
  if (e == hipSuccess) e = hipMalloc(&h->d_sorted, slots * 3u * sizeof(float));
  if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(path records): ") + hipGetErrorString(e));
  h->rec_capacity = static_cast<uint32_t>(slots);
  return AMBER_OK;
}

// Enqueues one launch over samples [first, first + n) of the band.  `sig` != nullptr: the signature variant of the same
// kernel; nothing is reduced (amber_hip_pt_signatures).
int EnsureLaunchCtl(amber_hip_pt* h) {                                       // queue head | record count | rays of the launch: one block, one memset per launch
  if (!h->d_launch_ctl) {
    HIP_TRY(hipMalloc(&h->d_launch_ctl, 4 * sizeof(unsigned int)));
    h->d_rec_count = h->d_launch_ctl + 1;
    h->d_rays_launch = reinterpret_cast<unsigned long long*>(h->d_launch_ctl + 2);
  }
  return AMBER_OK;
}

// Two-phase engine, once per handle: the candidate mask of every band pixel's eye rays (pixel_mask_kernel), enqueued on the render stream by
// create -- in front of the handle's first launch.  The buffer is allocated once and owned by the handle (amber_hip_pt_destroy releases it
// whatever step failed).  `timing` != null (amber_hip_kat_pixel_masks): run the kernel (again) between two events and report its duration.
int StartPixelMasks(amber_hip_pt* h, float* timing) {
  const uint32_t n_pixels = h->local_rows * h->scene.sensor.w;
  if (!(h->two_phase && h->pixel_mask_on) || n_pixels == 0 || (h->pixel_mask_ready && !timing)) return AMBER_OK;
  if (!h->d_pixel_mask) HIP_TRY(hipMalloc(&h->d_pixel_mask, static_cast<size_t>(n_pixels) * sizeof(uint32_t)));
  PixelMaskArgs pm{};
  for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++) pm.ap[i][c] = static_cast<double>(h->aperture_rect[i][c]) - static_cast<double>(h->scene.fp_center[c]);
  pm.inv_w = 1.0 / static_cast<double>(h->scene.sensor.wf); pm.inv_h = 1.0 / static_cast<double>(h->scene.sensor.hf);
  pm.focal_scale = h->lens_kind == 1u ? -(8.0 * static_cast<double>(h->scene.fp_reach) + 1.0) / static_cast<double>(h->lens_sensor_distance)
                                      : static_cast<double>(h->lens_focus_distance) / -static_cast<double>(h->lens_sensor_distance);
  pm.row_begin = h->row_begin; pm.stripe_rows = h->stripe_rows; pm.stripe_period = h->stripe_period; pm.n_pixels = n_pixels;
  pm.block = (h->stripe_rows == 0u || h->stripe_rows % 4u == 0u) ? 4u : 1u;
  { const char* ev = std::getenv("AMBER_PIXEL_MASK_BLOCK"); if (ev && (ev[0] == '1' || ev[0] == '2' || ev[0] == '4') && ev[1] == 0 && (h->stripe_rows == 0u || h->stripe_rows % static_cast<uint32_t>(ev[0] - '0') == 0u)) pm.block = static_cast<uint32_t>(ev[0] - '0'); }   // measurement hook
  pm.local_rows = h->local_rows;
  pm.blocks_x = (h->scene.sensor.w + pm.block - 1u) / pm.block;
  pm.n_blocks = pm.blocks_x * ((h->local_rows + pm.block - 1u) / pm.block);
  {
    // "does plane i cut the aperture rectangle?" -- the kernel's own expressions (binary64, the slack of its first lines), evaluated once
    const DevScene& sc = h->scene;
    const double cx = sc.fp_center[0], cy = sc.fp_center[1], cz = sc.fp_center[2], reach = sc.fp_reach;
    const double world_mag = std::max(std::max(std::fabs(cx), std::max(std::fabs(cy), std::fabs(cz))) + reach,
                                      std::max(std::fabs(double(h->lens_origin[0])), std::max(std::fabs(double(h->lens_origin[1])), std::fabs(double(h->lens_origin[2])))));
    const double slack = 1e-5 * reach + 32.0 * 5.9604644775390625e-08 * world_mag;
    pm.planes_cut_a = h->host_planes.size() > 32 ? 0xffffffffu : 0u;
    for (size_t i = 0; i < h->host_planes.size() && i < 32; i++) {
      const DevPlane& q = h->host_planes[i];
      bool above = true, below = true;
      for (int k = 0; k < 4; k++) {
        const double sa = double(q.n[0]) * pm.ap[k][0] + double(q.n[1]) * pm.ap[k][1] + double(q.n[2]) * pm.ap[k][2] - double(q.d0);
        above = above && sa > 10.0 * slack; below = below && sa < -10.0 * slack;
      }
      if (!(above || below)) pm.planes_cut_a |= 1u << i;
    }
  }
  struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evs;
  if (timing) { HIP_TRY(hipEventCreate(&evs.a)); HIP_TRY(hipEventCreate(&evs.b)); HIP_TRY(hipEventRecord(evs.a, h->stream)); }
  hipLaunchKernelGGL(pixel_mask_kernel, dim3((pm.n_blocks + 15u) / 16u), dim3(256), 0, h->stream, h->scene, pm, h->d_pixel_mask);   // 16 lanes per block of pixels
  HIP_TRY(hipGetLastError());
  if (timing) { HIP_TRY(hipEventRecord(evs.b, h->stream)); HIP_TRY(hipEventSynchronize(evs.b)); HIP_TRY(hipEventElapsedTime(timing, evs.a, evs.b)); }
  h->pixel_mask_ready = true;
  return AMBER_OK;
}

int LaunchPaths(amber_hip_pt* h, uint32_t first, uint32_t n, uint32_t n_pixels, unsigned long long* sig) {
  const uint64_t n_paths = static_cast<uint64_t>(n_pixels) * n;
#ifdef AMBER_LAB
  const bool bvh = h->hit_engine == AMBER_ENGINE_BVH && !h->bvh_paths;       // pt_bvh_pool_kernel (bvh_paths: pt_megakernel<ENGINE_BVH>)
#else
  if (sig) return Fail(AMBER_EINVAL, "path signatures are part of the lab build");
#endif
  const size_t need_words = static_cast<size_t>((n_paths + 31u) / 32u) + 4u;
  if (need_words > h->flag_words) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_flags) { HIP_TRY(hipFree(h->d_flags)); h->d_flags = nullptr; h->flag_words = 0; }
    hipError_t e = hipMalloc(&h->d_flags, need_words * sizeof(uint32_t));
    if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(path bitmap): ") + hipGetErrorString(e));
    h->flag_words = need_words;
    h->flags_dirty = true;
  }
  const size_t touched_words = static_cast<size_t>(n_pixels) / 32u + 2u;
  if (touched_words > h->touched_words) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_touched) { HIP_TRY(hipFree(h->d_touched)); h->d_touched = nullptr; h->touched_words = 0; }
    hipError_t e = hipMalloc(&h->d_touched, touched_words * sizeof(uint32_t));
    if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(touched pixels): ") + hipGetErrorString(e));
    h->touched_words = touched_words;
  }
  if (n_pixels > h->rank_pixels) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_excl) { HIP_TRY(hipFree(h->d_excl)); h->d_excl = nullptr; }
    if (h->d_block_sum) { HIP_TRY(hipFree(h->d_block_sum)); h->d_block_sum = nullptr; }
    h->rank_pixels = 0;
    hipError_t e = hipMalloc(&h->d_excl, static_cast<size_t>(n_pixels) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&h->d_block_sum, (static_cast<size_t>(n_pixels) / 256u + 2u) * sizeof(uint32_t));
    if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(pixel ranks): ") + hipGetErrorString(e));
    h->rank_pixels = n_pixels;
  }
  { const int rc = EnsureLaunchCtl(h); if (rc != AMBER_OK) return rc; }
  if (!h->h_rec_count) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_rec_count), sizeof(unsigned int), hipHostMallocDefault));
  if (!h->pending_event) HIP_TRY(hipEventCreateWithFlags(&h->pending_event, hipEventDisableTiming));
  const uint32_t n_blocks = PathBlocks(h, n_paths);
  {
    // measurements carried across bounces (degenerate paths only): pt_megakernel 3 floats per thread of the grid, pt_bvh_pool_kernel per ray of a wave
    size_t carried = static_cast<size_t>(h->n_cus) * ResidentBlocksPerCu(h->hit_engine, h->bvh_depth) * 256u * 3u * 2u;   // own slots + parked slots
#ifdef AMBER_LAB
    if (bvh) carried = static_cast<size_t>(h->n_cus) * AMBER_BVH_POOL_WGS * 4u * AMBER_BVH_POOL_CARRIED_PER_WAVE;
#endif
    if (carried > h->carried_floats) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->d_carried) { HIP_TRY(hipFree(h->d_carried)); h->d_carried = nullptr; h->carried_floats = 0; }
      hipError_t e = hipMalloc(&h->d_carried, carried * sizeof(float));
      if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(carried measurements): ") + hipGetErrorString(e));
      h->carried_floats = carried;
    }
  }
#ifdef AMBER_LAB
  if (bvh) {
    const size_t stack_ints = static_cast<size_t>(h->n_cus) * AMBER_BVH_POOL_WGS * 256u * AMBER_BVH_POOL_GLOBAL_LEVELS;
    if (stack_ints > h->bvh_stack_ints) {
      if (h->d_bvh_stack) { HIP_TRY(hipFree(h->d_bvh_stack)); h->d_bvh_stack = nullptr; h->bvh_stack_ints = 0; }
      hipError_t e = hipMalloc(&h->d_bvh_stack, stack_ints * sizeof(int32_t));
      if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(traversal stacks): ") + hipGetErrorString(e));
      h->bvh_stack_ints = stack_ints;
    }
  }
#endif
  RenderArgs a{};
  a.pixel_mask = h->pixel_mask_ready ? h->d_pixel_mask : nullptr;
  a.scene = h->scene; a.flags = h->d_flags; a.touched = h->d_touched; a.records = h->d_records; a.rec_count = h->d_rec_count; a.rec_capacity = h->rec_capacity;
  a.ray_count = h->d_rays_launch; a.next_item = h->d_launch_ctl; a.stamps = h->d_stamps; a.hashed_seed = h->hashed_seed;
  a.bvh_stack = h->d_bvh_stack; a.carried = h->d_carried; a.sig = sig;
  a.row_begin = h->row_begin; a.stripe_rows = h->stripe_rows; a.stripe_period = h->stripe_period; a.n_pixels = n_pixels; a.first_sample = first; a.n_samples = n;
  a.n_chunks = (n + AMBER_ACCUM_CHUNK - 1) / AMBER_ACCUM_CHUNK; a.n_items = static_cast<uint32_t>(n_paths);
  HIP_TRY(hipMemsetAsync(h->d_launch_ctl, 0, 4 * sizeof(unsigned int), h->stream));
  // The bitmap is cleared by the reduction itself where it can be (whole words per pixel); the host clears all of it only when a
  // launch left it dirty: the first use, sample counts that are not multiples of 32, signature launches, a launch that ran out of slots.
  if (h->flags_dirty) HIP_TRY(hipMemsetAsync(h->d_flags, 0, h->flag_words * sizeof(uint32_t), h->stream));
  h->flags_dirty = sig != nullptr || (n & 31u) != 0u;
  HIP_TRY(hipMemsetAsync(h->d_touched, 0, touched_words * sizeof(uint32_t), h->stream));
  std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
  { const int rc = AcquireEventPair(h, &evp); if (rc != AMBER_OK) return rc; }
  auto& ev = *evp;
  HIP_TRY(hipEventRecord(ev.first, h->stream));
#ifdef AMBER_LAB
  if (sig) {                                                  // the signature instantiations of the same kernels (amber_hip_pt_signatures)
    if (bvh) hipLaunchKernelGGL((pt_bvh_pool_kernel<true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->bvh_paths) hipLaunchKernelGGL((pt_megakernel<ENGINE_BVH, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == kHitTwoPhaseN) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE_N, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == AMBER_ENGINE_REFERENCE_BVH) hipLaunchKernelGGL((pt_megakernel<ENGINE_REF_BVH, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((pt_megakernel<ENGINE_LIST, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
  } else if (bvh) {
    hipLaunchKernelGGL((pt_bvh_pool_kernel<false>), dim3(n_blocks), dim3(256), 0, h->stream, a);
  } else
#endif
  {
    if (h->bvh_paths) hipLaunchKernelGGL((pt_megakernel<ENGINE_BVH>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == kHitTwoPhaseN) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE_N>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == AMBER_ENGINE_REFERENCE_BVH) hipLaunchKernelGGL((pt_megakernel<ENGINE_REF_BVH>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((pt_megakernel<ENGINE_LIST>), dim3(n_blocks), dim3(256), 0, h->stream, a);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev.second, h->stream));
  if (sig) return AMBER_OK;
  const uint32_t n_rank_blocks = (n_pixels + 255u) / 256u;
  hipLaunchKernelGGL(rec_rank_kernel, dim3(n_rank_blocks), dim3(256), 0, h->stream, h->d_flags, h->d_touched, n_pixels, n, h->d_excl, h->d_block_sum);
  hipLaunchKernelGGL(rec_scan_blocks_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_block_sum, n_rank_blocks, h->d_rec_count, h->rec_capacity, h->d_rays, h->d_rays_launch);
  hipLaunchKernelGGL(rec_place_kernel, dim3(static_cast<uint32_t>(h->n_cus) * 8u), dim3(256), 0, h->stream, h->d_records, h->d_rec_count, h->rec_capacity, h->d_flags,
                     h->d_excl, h->d_block_sum, n, h->d_sorted);
  hipLaunchKernelGGL(reduce_flagged_kernel, dim3(n_rank_blocks), dim3(256), 0, h->stream, h->d_fb, h->d_flags, h->d_touched, h->d_sorted, h->d_excl, h->d_block_sum,
                     h->d_rec_count, h->rec_capacity, n_pixels, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h->h_rec_count, h->d_rec_count, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipEventRecord(h->pending_event, h->stream));
  h->pending = true; h->pending_first = first; h->pending_n = n;
  h->pending_checked = n_paths + RecordSlack(h) <= h->rec_capacity;      // a slot for every path: cannot run out
  return AMBER_OK;
}

// Looks at the record counter of the launch in flight (waits for it), remembers the density, and repeats the launch with a
// larger buffer if it ran out of slots.  Every entry point that reads results, or enqueues work whose order matters, calls it.
int ResolvePending(amber_hip_pt* h) {
  while (h->pending) {
    HIP_TRY(hipEventSynchronize(h->pending_event));
    h->pending = false;
    const uint64_t used = *h->h_rec_count;
    const uint64_t n_paths = static_cast<uint64_t>(h->local_rows) * h->scene.sensor.w * h->pending_n;
    h->rec_density = n_paths ? static_cast<double>(used) / static_cast<double>(n_paths) : 0.0;
    h->density_known = true;
    if (used <= h->rec_capacity) break;
    // out of slots: nothing of that launch reached the framebuffer or the ray total (and its bits are still set)
    h->flags_dirty = true;
    const int rc = EnsureRecordCapacity(h, used + used / 4u + RecordSlack(h));
    if (rc != AMBER_OK) return rc;
    const int rl = LaunchPaths(h, h->pending_first, h->pending_n, h->local_rows * h->scene.sensor.w, nullptr);
    if (rl != AMBER_OK) return rl;
  }
  return AMBER_OK;
}

int RenderPassPaths(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, uint32_t n_pixels) {
  uint64_t max_samples = kMaxPathsPerLaunch / n_pixels / AMBER_ACCUM_CHUNK * AMBER_ACCUM_CHUNK;
  if (max_samples == 0) return Fail(AMBER_EINVAL, "band too large for one launch");
  const uint64_t slack = h->test_density_scale > 0 ? 64u : RecordSlack(h);   // (test hook: no cushion either)
  uint32_t done = 0;
  while (done < n_samples) {
    // the previous launch must stand before the next one adds to the framebuffer (the order of the sums is part of the
    // contract), unless it had a slot for every path
    if (h->pending && (!h->pending_checked || !h->density_known)) { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
    uint32_t n = n_samples - done;
    if (n > max_samples) n = static_cast<uint32_t>(max_samples);
    uint64_t slots;
    if (!h->density_known) {
      // first launch of the handle: a slot for every path, and at most one chunk unless the job is small -- it measures the density
      if (static_cast<uint64_t>(n_pixels) * n > (4ull << 20) && n > AMBER_ACCUM_CHUNK) n = AMBER_ACCUM_CHUNK;
      slots = static_cast<uint64_t>(n_pixels) * n + slack;
    } else {
      double density = h->rec_density;
      if (h->test_density_scale > 0) density *= h->test_density_scale;   // test hook: a wrong estimate must only cost a repeated launch
      const double per_sample = std::max(1e-9, density * 1.5) * static_cast<double>(n_pixels);   // slots one sample of the band needs, with margin
      const uint64_t all = static_cast<uint64_t>(n_pixels) * n;
      const double want = per_sample * n;
      if (want + static_cast<double>(slack) > static_cast<double>(kMaxRecordSlots) && all + slack > kMaxRecordSlots) {
        // a dense scene: shorter launches instead of a larger buffer
        uint64_t fit = static_cast<uint64_t>((static_cast<double>(kMaxRecordSlots) - static_cast<double>(slack)) / per_sample) / AMBER_ACCUM_CHUNK * AMBER_ACCUM_CHUNK;
        if (fit < AMBER_ACCUM_CHUNK) fit = AMBER_ACCUM_CHUNK;
        if (n > fit) n = static_cast<uint32_t>(fit);
      }
      const uint64_t all_n = static_cast<uint64_t>(n_pixels) * n;
      slots = std::min<uint64_t>(all_n, static_cast<uint64_t>(per_sample * n) + 1u) + slack;
    }
    if (slots > h->rec_capacity) {
      if (h->pending) { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }     // the buffer is in use
      const int rc = EnsureRecordCapacity(h, slots);
      if (rc != AMBER_OK) return rc;
    }
    const int rl = LaunchPaths(h, first_sample + done, n, n_pixels, nullptr);
    if (rl != AMBER_OK) return rl;
    done += n;
  }
  return AMBER_OK;
}
}  // namespace

static int RenderPassBvhItems(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, uint32_t n_pixels, unsigned long long* sig);   // internal: not part of the ABI
#ifdef AMBER_LAB
namespace { int RenderPassWavefront(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples); }      // lab_api.inc
#endif

int amber_hip_pt_render_pass(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  if (n_samples == 0) return AMBER_OK;
  if (static_cast<uint64_t>(first_sample) + n_samples > 0xffffffffull) return Fail(AMBER_EINVAL, "sample index overflow");
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t n_pixels = h->local_rows * h->scene.sensor.w;
  if (n_pixels == 0) return AMBER_OK;                    // empty band
#ifdef AMBER_LAB
  if (h->engine == AMBER_ENGINE_WAVEFRONT) return RenderPassWavefront(h, first_sample, n_samples);
#endif
  if (h->hit_engine != AMBER_ENGINE_BVH || h->bvh_pool || h->bvh_paths) return RenderPassPaths(h, first_sample, n_samples, n_pixels);
  return RenderPassBvhItems(h, first_sample, n_samples, n_pixels, nullptr);
}

// sig != null (amber_hip_pt_signatures): one launch of the signature instantiation; nothing reaches the framebuffer or the ray total
static int RenderPassBvhItems(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, uint32_t n_pixels, unsigned long long* sig) {
  // engine BVH, default scheduler (pt_bvh_megakernel: lanes own (pixel, chunk) items).  A launch covers at most kMaxPartialFloats
  // of per-item sums and < 2^31 items; longer passes are split on chunk boundaries, which leaves the summation order unchanged
  const uint64_t kMaxPartialFloats = 768ull << 20;    // 3 GiB
  uint64_t max_chunks = kMaxPartialFloats / (static_cast<uint64_t>(n_pixels) * 3u);
  const uint64_t by_items = 0x7fffffffull / n_pixels;
  if (by_items < max_chunks) max_chunks = by_items;
  if (max_chunks == 0) return Fail(AMBER_EINVAL, "band too large for one launch");
  uint32_t done = 0;
  while (done < n_samples) {
    uint32_t n = n_samples - done;
    const uint64_t cap = max_chunks * AMBER_ACCUM_CHUNK;
    if (n > cap) n = static_cast<uint32_t>(cap);
    const uint32_t n_chunks = (n + AMBER_ACCUM_CHUNK - 1) / AMBER_ACCUM_CHUNK;
    const size_t need = static_cast<size_t>(n_chunks) * n_pixels * 3u;
    if (need > h->partial_floats) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->d_partial) { HIP_TRY(hipFree(h->d_partial)); h->d_partial = nullptr; h->partial_floats = 0; }
      hipError_t e = hipMalloc(&h->d_partial, need * sizeof(float));
      if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(partial sums): ") + hipGetErrorString(e));
      h->partial_floats = need;
    }
    if (sig && n != n_samples) return Fail(AMBER_EINVAL, "too many paths for one signature launch");
    if (sig) { const int rc = EnsureLaunchCtl(h); if (rc != AMBER_OK) return rc; }
    RenderArgs a{};
    a.scene = h->scene; a.partial = h->d_partial; a.ray_count = sig ? h->d_rays_launch : h->d_rays; a.next_item = h->d_next; a.stamps = h->d_stamps; a.hashed_seed = h->hashed_seed;
    a.row_begin = h->row_begin; a.stripe_rows = h->stripe_rows; a.stripe_period = h->stripe_period; a.n_pixels = n_pixels; a.first_sample = first_sample + done; a.n_samples = n;
    a.n_chunks = n_chunks; a.n_items = n_pixels * n_chunks; a.sig = sig; a.shade_batch = h->bvh_shade_batch;
    // persistent workers: one workgroup of 4 waves per CU and resident wave slot, fewer if the queue is short
    uint32_t n_blocks = static_cast<uint32_t>(h->n_cus) * ResidentBlocksPerCu(h->hit_engine, h->bvh_depth);
    const uint32_t by_work = (a.n_items + 255u) / 256u;
    if (by_work < n_blocks) n_blocks = by_work;
    HIP_TRY(hipMemsetAsync(h->d_next, 0, sizeof(unsigned int), h->stream));
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    { const int rc = AcquireEventPair(h, &evp); if (rc != AMBER_OK) return rc; }
    auto& ev = *evp;
    HIP_TRY(hipEventRecord(ev.first, h->stream));
#ifdef AMBER_LAB
    if (sig) {
      if (h->bvh_depth <= 24) hipLaunchKernelGGL((pt_bvh_megakernel<false, 24, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      else hipLaunchKernelGGL((pt_bvh_megakernel<false, AMBER_BVH_STACK, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(ev.second, h->stream));
      return AMBER_OK;
    }
#else
    if (sig) return Fail(AMBER_EINVAL, "path signatures are part of the lab build");
#endif
    if (h->bvh_depth <= 24) hipLaunchKernelGGL((pt_bvh_megakernel<false, 24>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((pt_bvh_megakernel<false, AMBER_BVH_STACK>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.second, h->stream));
    const uint32_t n_elems = n_pixels * 3u;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n_elems + 255u) / 256u), dim3(256), 0, h->stream, h->d_fb, h->d_partial, n_elems, n_chunks);
    HIP_TRY(hipGetLastError());
    done += n;
  }
  return AMBER_OK;
}

int amber_hip_lt_trace_range(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, uint32_t path_begin, uint32_t path_end,
                             AmberSplat* out, uint32_t capacity, uint32_t* n_out, uint64_t* ray_count) {
  if (!h || !n_out || (capacity && !out)) return Fail(AMBER_EINVAL, "null argument");
  *n_out = 0;
  if (ray_count) *ray_count = 0;
  if (h->engine == AMBER_ENGINE_WAVEFRONT) return Fail(AMBER_EINVAL, "light tracing runs on the work-queue kernel (engine auto, list, two_phase or bvh)");
  if (static_cast<uint64_t>(first_sample) + n_samples > 0xffffffffull) return Fail(AMBER_EINVAL, "sample index overflow");
  const uint32_t all_paths = h->scene.sensor.w * h->scene.sensor.h;          // image.Size() light paths per pass
  if (path_begin > path_end || path_end > all_paths) return Fail(AMBER_EINVAL, "bad light-path range");
  const uint32_t n_paths = path_end - path_begin;
  if (n_samples == 0 || n_paths == 0 || h->scene.n_lights == 0) return AMBER_OK;
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  const bool bvh = h->hit_engine == AMBER_ENGINE_BVH;
  const uint32_t dev_capacity = capacity ? capacity : 1u;
  if (dev_capacity > h->splat_capacity) {
    if (h->d_splats) { HIP_TRY(hipFree(h->d_splats)); h->d_splats = nullptr; h->splat_capacity = 0; }
    hipError_t e = hipMalloc(&h->d_splats, static_cast<size_t>(dev_capacity) * sizeof(DevSplat));
    if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(splats): ") + hipGetErrorString(e));
    h->splat_capacity = dev_capacity;
  }
  if (!h->d_splat_count) HIP_TRY(hipMalloc(&h->d_splat_count, sizeof(unsigned int)));
  // One launch numbers its work units with 31 bits -- pt_megakernel: (light path, pass) pairs; pt_bvh_megakernel: (light
  // path, chunk of passes) items -- so a long range of passes is traced in several launches, each a whole number of chunks.
  // Launches run in pass order and each one's splats are sorted, so the concatenation is in (pass, path, bounce) order.
  const uint64_t max_units = 0x7fffffffull / n_paths;
  if (max_units == 0) return Fail(AMBER_EINVAL, "too many light paths for one launch");
  uint64_t max_samples = bvh ? max_units * AMBER_ACCUM_CHUNK : max_units / AMBER_ACCUM_CHUNK * AMBER_ACCUM_CHUNK;
  if (max_samples == 0) max_samples = max_units;               // fewer than a chunk fits: any split is valid for light tracing (nothing is summed on the device)
  uint64_t total = 0;                                         // splats produced (also beyond the caller's capacity)
  unsigned long long rays_before = 0, rays_after = 0;
  HIP_TRY(hipMemcpyAsync(&rays_before, h->d_rays, sizeof rays_before, hipMemcpyDeviceToHost, h->stream));
  uint32_t done = 0;
  while (done < n_samples) {
    uint32_t n = n_samples - done;
    if (n > max_samples) n = static_cast<uint32_t>(max_samples);
    const uint32_t n_chunks = (n + AMBER_ACCUM_CHUNK - 1) / AMBER_ACCUM_CHUNK;
    const uint64_t n_work = bvh ? static_cast<uint64_t>(n_paths) * n_chunks : static_cast<uint64_t>(n_paths) * n;
    HIP_TRY(hipMemsetAsync(h->d_splat_count, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_next, 0, sizeof(unsigned int), h->stream));
    RenderArgs a{};
    a.scene = h->scene; a.ray_count = h->d_rays; a.next_item = h->d_next;
    a.splats = h->d_splats; a.splat_count = h->d_splat_count; a.splat_capacity = dev_capacity; a.hashed_seed = h->hashed_seed_lt;
    a.n_pixels = n_paths; a.first_sample = first_sample + done; a.n_samples = n; a.path_offset = path_begin;
    a.n_chunks = n_chunks; a.n_items = static_cast<uint32_t>(n_work); a.shade_batch = h->bvh_shade_batch;
    uint32_t n_blocks = static_cast<uint32_t>(h->n_cus) * ResidentBlocksPerCu(h->hit_engine, h->bvh_depth);
    const uint32_t by_work = (a.n_items + 255u) / 256u;
    if (by_work < n_blocks) n_blocks = by_work;
      if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      else if (h->hit_engine == kHitTwoPhaseN) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE_N, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      else if (h->hit_engine == AMBER_ENGINE_REFERENCE_BVH) hipLaunchKernelGGL((pt_megakernel<ENGINE_REF_BVH, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (bvh) {
      if (h->bvh_depth <= 24) hipLaunchKernelGGL((pt_bvh_megakernel<true, 24>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      else hipLaunchKernelGGL((pt_bvh_megakernel<true, AMBER_BVH_STACK>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    }
    else hipLaunchKernelGGL((pt_megakernel<ENGINE_LIST, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    unsigned int produced = 0;
    HIP_TRY(hipMemcpyAsync(&produced, h->d_splat_count, sizeof produced, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const uint64_t room = total < capacity ? capacity - total : 0;
    if (produced <= room && produced <= dev_capacity) {
      if (produced) {
        static_assert(sizeof(AmberSplat) == sizeof(DevSplat), "splat layouts must agree");
        AmberSplat* dst = out + total;
        HIP_TRY(hipMemcpy(dst, h->d_splats, static_cast<size_t>(produced) * sizeof(DevSplat), hipMemcpyDeviceToHost));
        // the reference adds the splats of pass s in path order (algorithm_lt.cc:112-123): restore that order
        std::sort(dst, dst + produced, [](const AmberSplat& x, const AmberSplat& y) {
          if (x.sample != y.sample) return x.sample < y.sample;
          if (x.path != y.path) return x.path < y.path;
          return x.bounce < y.bounce;
        });
      }
    }
    total += produced;                                        // keeps counting: the caller learns the capacity it needs
    done += n;
  }
  HIP_TRY(hipMemcpy(&rays_after, h->d_rays, sizeof rays_after, hipMemcpyDeviceToHost));
  if (ray_count) *ray_count = rays_after - rays_before;
  *n_out = total > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(total);
  if (total > capacity) return Fail(AMBER_ENOMEM, "splat buffer too small: " + std::to_string(total) + " splats produced");
  return AMBER_OK;
}

int amber_hip_lt_trace(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, AmberSplat* out, uint32_t capacity,
                       uint32_t* n_out, uint64_t* ray_count) {
  if (!h) return Fail(AMBER_EINVAL, "null argument");
  return amber_hip_lt_trace_range(h, first_sample, n_samples, 0u, h->scene.sensor.w * h->scene.sensor.h, out, capacity, n_out, ray_count);
}

int amber_hip_pt_clear(amber_hip_pt* h) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  const size_t fb_floats = static_cast<size_t>(h->local_rows) * h->scene.sensor.w * 3;
  HIP_TRY(hipMemsetAsync(h->d_fb, 0, fb_floats * sizeof(float), h->stream));
  HIP_TRY(hipMemsetAsync(h->d_rays, 0, sizeof(unsigned long long), h->stream));
  // (no host synchronisation: the two fills are ordered on the handle's stream like everything else; event pairs handed out before
  //  this call have either been read by kernel_time() or are dropped here)
  h->events_used = 0; h->timed_launches = 0; h->timed_ms = 0;
  return AMBER_OK;
}

int amber_hip_pt_sync(amber_hip_pt* h) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return AMBER_OK;
}

int amber_hip_pt_download(amber_hip_pt* h, float* rgb_sum, uint64_t* ray_count) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  const size_t fb_floats = static_cast<size_t>(h->local_rows) * h->scene.sensor.w * 3;
  if (rgb_sum) HIP_TRY(hipMemcpyAsync(rgb_sum, h->d_fb, fb_floats * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  unsigned long long r = 0;
  HIP_TRY(hipMemcpyAsync(&r, h->d_rays, sizeof r, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (ray_count) *ray_count = r;
  return AMBER_OK;
}

int amber_hip_pt_device_framebuffer(amber_hip_pt* h, void** dptr, uint64_t* n_floats) {
  if (!h || !dptr) return Fail(AMBER_EINVAL, "null argument");
  *dptr = h->d_fb;
  if (n_floats) *n_floats = static_cast<uint64_t>(h->local_rows) * h->scene.sensor.w * 3;
  return AMBER_OK;
}

int amber_hip_pt_stream(amber_hip_pt* h, void** stream) {
  if (!h || !stream) return Fail(AMBER_EINVAL, "null argument");
  *stream = h->stream;
  return AMBER_OK;
}

int amber_hip_pt_local_rows(amber_hip_pt* h, uint32_t* n_rows) {
  if (!h || !n_rows) return Fail(AMBER_EINVAL, "null argument");
  *n_rows = h->local_rows;
  return AMBER_OK;
}

int amber_hip_pt_kernel_time(amber_hip_pt* h, uint32_t* n_launches, double* total_ms) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  HIP_TRY(hipStreamSynchronize(h->stream));
  double tot = h->timed_ms;
  for (size_t i = 0; i < h->events_used; i++) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
    tot += ms;
  }
  if (n_launches) *n_launches = h->timed_launches + static_cast<uint32_t>(h->events_used);
  if (total_ms) *total_ms = tot;
  return AMBER_OK;
}

#ifdef AMBER_STAMPS
// diagnostic build only: per wave of the last pt_megakernel launch, wall-clock ticks (100 MHz) at start, after the first claim,
// when it found the queue empty, at its end (tools/wave_times.py)
extern "C" int amber_hip_pt_read_wave_times(amber_hip_pt* h, unsigned long long* out, unsigned int n_waves) {
  if (!h || !h->d_stamps || n_waves > AMBER_WAVE_TIME_SLOTS) return AMBER_EINVAL;
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(out, h->d_stamps + 8, 4ull * n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? AMBER_OK : AMBER_EHIP;
}
extern "C" int amber_hip_pt_read_stamps(amber_hip_pt* h, unsigned long long out[8]) {
  if (!h || !h->d_stamps) return AMBER_EINVAL;
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(out, h->d_stamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? AMBER_OK : AMBER_EHIP;
}
#endif

void amber_hip_pt_destroy(amber_hip_pt* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h->d_objects) (void)hipFree(h->d_objects);
  if (h->d_materials) (void)hipFree(h->d_materials);
  if (h->d_blades) (void)hipFree(h->d_blades);
  if (h->d_planes) (void)hipFree(h->d_planes);
  if (h->d_tri_filters) (void)hipFree(h->d_tri_filters);
  if (h->d_sphere_filters) (void)hipFree(h->d_sphere_filters);
  if (h->d_prog_objects) (void)hipFree(h->d_prog_objects);
  if (h->d_groups) (void)hipFree(h->d_groups);
  if (h->d_bvh_nodes) (void)hipFree(h->d_bvh_nodes);
  if (h->d_bvh_nodes4) (void)hipFree(h->d_bvh_nodes4);
  if (h->d_bvh_spheres) (void)hipFree(h->d_bvh_spheres);
  if (h->d_bvh_tris) (void)hipFree(h->d_bvh_tris);
  if (h->d_bvh_prims) (void)hipFree(h->d_bvh_prims);
  if (h->d_bvh_objects) (void)hipFree(h->d_bvh_objects);
  if (h->d_ref_nodes) (void)hipFree(h->d_ref_nodes);
  if (h->d_ref_leaves) (void)hipFree(h->d_ref_leaves);
  if (h->d_ref_stack) (void)hipFree(h->d_ref_stack);
  if (h->d_wf) (void)hipFree(h->d_wf);
  if (h->d_fb) (void)hipFree(h->d_fb);
  if (h->d_rays) (void)hipFree(h->d_rays);
  if (h->d_next) (void)hipFree(h->d_next);
  if (h->d_stamps) (void)hipFree(h->d_stamps);
  if (h->d_lights) (void)hipFree(h->d_lights);
  if (h->d_lens) (void)hipFree(h->d_lens);
  if (h->d_splats) (void)hipFree(h->d_splats);
  if (h->d_splat_count) (void)hipFree(h->d_splat_count);
  if (h->d_partial) (void)hipFree(h->d_partial);
  if (h->d_flags) (void)hipFree(h->d_flags);
  if (h->d_touched) (void)hipFree(h->d_touched);
  if (h->d_records) (void)hipFree(h->d_records);
  if (h->d_sorted) (void)hipFree(h->d_sorted);
  if (h->d_launch_ctl) (void)hipFree(h->d_launch_ctl);                     // d_rec_count and d_rays_launch point into it
  if (h->d_excl) (void)hipFree(h->d_excl);
  if (h->d_block_sum) (void)hipFree(h->d_block_sum);
  if (h->h_rec_count) (void)hipHostFree(h->h_rec_count);
  if (h->pending_event) (void)hipEventDestroy(h->pending_event);
  if (h->d_bvh_stack) (void)hipFree(h->d_bvh_stack);
  if (h->d_carried) (void)hipFree(h->d_carried);
  if (h->d_sig) (void)hipFree(h->d_sig);
  if (h->d_pixel_mask) (void)hipFree(h->d_pixel_mask);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

}  // extern "C"

#ifdef AMBER_LAB
#include "lab_api.inc"
#endif
