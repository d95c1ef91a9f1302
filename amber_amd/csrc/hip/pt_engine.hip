// pt_engine.hip -- kernels and C ABI (include/amber_hip.h) of the gfx950 path-tracing engine.
//
// Persistent work-queue kernels.
//   pt_megakernel       (<= 32 objects: engines LIST / TWO_PHASE; also light tracing) -- the work unit is ONE PATH, numbered
//                       q = (band pixel) * n_samples + (sample offset) within a launch; a wave claims 1024 consecutive paths with
//                       one atomicAdd.  Lanes are decoupled from pixels: when the wave's pool (LDS) cannot serve a lane, every
//                       lane parks its ray there and all 64 lanes start the next 64 paths together (a PRIMARY ROUND: 64 samples
//                       of one pixel, whose candidate objects come from a per-pixel mask, pixel_mask_kernel); a lane whose path
//                       ended pops a parked ray (ballot + mbcnt rank).  Ray / throughput / sampler state lives in VGPRs; scene
//                       records arrive through scalar loads (wave-uniform indices).
//   pt_bvh_megakernel   (engine BVH, the default for larger scenes; also light tracing) -- lanes own (pixel, chunk of 8 samples)
//                       items; the closest hit is a RESUMABLE per-lane traversal of the 2-wide binary16-plane tree (stack in LDS),
//                       advanced in wave rounds until 52 lanes have finished theirs; those are shaded and restarted.
//   pt_bvh_pool_kernel  (engine BVH, opt-in: AMBER_PT_FLAG_BVH_POOL; bvh_pool.inc) -- path-granular; 64 rays in flight on the
//                       lanes plus a pool of waiting rays per wave in LDS: a lane whose traversal finished swaps its ray for a
//                       waiting one at once, finished rays are shaded in batches.  Bit-identical, measured slower.
// Accumulation without owners (the path-granular kernels): a path that ends with a non-zero measurement appends a record {q, rgb}
// (slots reserved per wave, 64 at a time) and sets bit q of a bitmap; rec_rank_kernel / rec_place_kernel move the records into
// path order and reduce_flagged_kernel forms, per pixel, exactly the sums of the numerical contract;
// pt_bvh_megakernel writes one sum per item and reduce_partials_kernel adds them in chunk order -- the same sums.
// wavefront.inc holds the streaming (SoA queues in HBM, one launch per bounce) formulation, bvh_stream.inc the traversal on
// its own; both kept for measurement.
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
#include "pt_device.h"
#include "bvh_build.h"
#include "filter_build.h"

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
struct RenderArgs {
  DevScene scene;
  float* partial;            // pt_bvh_megakernel: [n_chunks][n_pixels][3] per-item sums of this launch
  uint32_t* flags;           // path-granular kernels: bit q set = path q of this launch ended with a non-zero measurement ...
  uint32_t* touched;         // ... and bit (q / n_samples) of this one: the band pixels that have any record (the reduction skips the others)
  uint4* records;            // ... appended as {q, r, g, b}; q = 0xffffffff marks a reserved slot that was never used
  unsigned int* rec_count;   // slots handed out so far (waves reserve AMBER_REC_BLOCK at a time)
  uint32_t rec_capacity;     // slots of `records`; a launch that needs more is repeated by the host with a larger buffer
  unsigned long long* ray_count;
  unsigned int* next_item;   // work-queue head (zeroed before every launch)
  unsigned long long* stamps; // diagnostic build only (AMBER_STAMPS): 8 section sums
  DevSplat* splats;          // light tracing: splat records, their counter and capacity
  unsigned int* splat_count;
  uint32_t splat_capacity;
  int32_t* bvh_stack;        // pt_bvh_pool_kernel: traversal-stack levels beyond the LDS part, [level][thread of the grid]
  float* carried;            // pt_bvh_pool_kernel: measurements of the (degenerate) paths that carry a non-zero one across bounces, [thread of the grid * kRays/64 ...]
  const uint32_t* pixel_mask; // pt_megakernel, two-phase engine: candidate mask of every band pixel's primary rays (pixel_mask_kernel); null = off
  unsigned long long* sig;   // signature variants: sig[q] = the path's hit-object / hit-distance hashes (amber_hip_pt_signatures)
  uint64_t hashed_seed;      // SplitMix64(global_seed)
  uint32_t row_begin;
  uint32_t stripe_rows, stripe_period;   // 0,0 = contiguous rows
  uint32_t n_pixels;         // pixels of the band
  uint32_t first_sample, n_samples;
  uint32_t n_chunks, n_items;   // pt_bvh_megakernel: items = (pixel, chunk); path-granular kernels: n_items = paths of the launch
  uint32_t path_offset;      // light tracing: index of the first light path of this launch's range (amber_hip_lt_trace_range)
  uint32_t shade_batch;      // pt_bvh_megakernel: lanes that must have finished their traversal before the wave shades (the handle's choice, BvhShadeBatch)
};

// FNV-1a-32 step over the four bytes of v (path signatures: amber_hip_kat_signatures, amber_hip_pt_signatures)
__device__ __forceinline__ uint32_t Fnv32(uint32_t h, uint32_t v) {
#pragma unroll
  for (int k = 0; k < 4; k++) { h ^= (v >> (8 * k)) & 0xffu; h *= 16777619u; }
  return h;
}

// Accumulation without owners, device side.  A path that ends with a measurement other than +0 appends {q, rgb} to the
// record buffer and sets bit q.  Slots are reserved per WAVE, AMBER_REC_BLOCK at a time (one atomicAdd on the shared
// counter per 64 records: a scene in which every second path reaches a light would otherwise put 5e8 atomics on one
// address -- the L2 retires ~88 of those per microsecond); [rec_next, rec_end) is the wave's open block (SGPRs).  Must be
// called with all 64 lanes active.  A slot beyond the buffer is not written: the counter then tells the host to repeat
// the launch with a larger buffer (RenderPassPaths).
#define AMBER_REC_BLOCK 64u
#define AMBER_REC_UNUSED 0xffffffffu
__device__ __forceinline__ void EmitRecords(const RenderArgs& a, bool emit, uint32_t q, V3 meas, uint32_t& rec_next, uint32_t& rec_end) {
  const unsigned long long me = __ballot(emit);
  if (me == 0ull) return;                                     // wave-uniform
  const uint32_t n = static_cast<uint32_t>(__popcll(me));
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(me >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(me), 0u));
  const uint32_t room = rec_end - rec_next;
  uint32_t fresh = 0;
  if (n > room) {                                             // n <= 64 = AMBER_REC_BLOCK: one new block always suffices
    if ((threadIdx.x & 63u) == 0u) fresh = atomicAdd(a.rec_count, AMBER_REC_BLOCK);
    fresh = __builtin_amdgcn_readfirstlane(fresh);
  }
  if (emit) {
    const uint32_t slot = rank < room ? rec_next + rank : fresh + (rank - room);
    if (slot < a.rec_capacity) a.records[slot] = make_uint4(q, __float_as_uint(meas.x), __float_as_uint(meas.y), __float_as_uint(meas.z));
    atomicOr(a.flags + (q >> 5), 1u << (q & 31u));
    const uint32_t p = q / a.n_samples;
    atomicOr(a.touched + (p >> 5), 1u << (p & 31u));
  }
  if (n > room) { rec_next = fresh + (n - room); rec_end = fresh + AMBER_REC_BLOCK; }
  else rec_next += n;
}
// At the end of a wave: the slots of its open block that were never used are marked.
__device__ __forceinline__ void CloseRecords(const RenderArgs& a, uint32_t rec_next, uint32_t rec_end) {
  const uint32_t k = rec_next + (threadIdx.x & 63u);
  if (k < rec_end && k < a.rec_capacity) a.records[k].x = AMBER_REC_UNUSED;
}

// pt_megakernel: persistent waves, work unit = ONE PATH, lanes decoupled from pixels.
//
//  * Paths of a launch are numbered q = plocal * n_samples + k (band pixel, sample offset): 64 consecutive paths are 64
//    samples of one pixel (coherent eye rays, first hits and materials).  A wave claims AMBER_CLAIM_PATHS paths with ONE
//    atomicAdd on the global queue head.
//  * Eye rays are generated 64 at a time, by ALL lanes of the wave, into the wave's pool in LDS (SoA, 64 slots); a lane
//    whose path has ended pops the next ray from the pool (ballot + mbcnt rank), whichever path that is.  Round 1's
//    kernel regenerated in place, i.e. the whole wave executed the eye-ray code for the half of its lanes whose paths had
//    just ended -- 16 % of the kernel at 48 % lane utilisation; here that code runs once per 64 paths with every lane
//    busy, and no lane ever waits: a generation round happens only when the pool cannot serve a lane.
//  * Accumulation.  Lanes no longer own pixels, so a path that ends with a non-zero measurement (it reached a light:
//    2e-5 of the Cornell paths) appends a record {q, rgb} and sets bit q of a bitmap (EmitRecords); the rank / place /
//    reduce kernels below then form, per pixel, exactly the sums of the numerical contract -- samples of a chunk of
//    AMBER_ACCUM_CHUNK in order, chunk sums added to the framebuffer in chunk order -- skipping the
//    +0 terms, which is exact: a running binary32 sum that starts at +0 never becomes -0, and x + (+0) = x for every
//    other x.  HBM per launch: the bitmap (n_paths / 8 bytes, cleared and read once) and 16 B per record.
//  * kLight: the same worker loop traces LIGHT paths (algorithm_lt.cc:112-163): path q = (light path index, pass),
//    nothing is stored at a path's end, Eye hits append splat records instead.
//  * kSig (tests only, amber_hip_pt_signatures): the same kernel also hashes every cast's hit object and hit distance and
//    stores the pair at the path's end -- the PRODUCT kernel's paths, compared with the oracle's one by one.
#define AMBER_WAVE_TIME_SLOTS 8192u          /* diagnostic build: per-wave timestamps behind the 8 section sums of RenderArgs.stamps */
#ifndef AMBER_MEGAKERNEL_WAVES_PER_SIMD
#define AMBER_MEGAKERNEL_WAVES_PER_SIMD 6
#endif
// Paths a wave claims from the global queue head with one atomicAdd.  One L2 word retires ~88 returning atomics per
// microsecond (MI355X_MICROARCH.md, "dequeue"): claiming 64 paths at a time (1.7e7 atomics per config-2 launch) made the
// queue head the bottleneck -- 190 ms per launch; 1024 paths (16 generation rounds) is 15 atomics per microsecond, and the
// tail it can leave on one wave is ~35 iterations (80 us).  Round 3 measured the queue again (tools/launch_fixed_cost.py, kernel time
// = fixed + slope * spp on config 2's frame): a claim costs the wave 4.4 us (512 / 1024 / 2048 / 4096 paths per claim: slope 51.67 /
// 50.97 / 50.60 / 50.21 us per spp, i.e. 0.75 ms of a 52.8-ms launch at 1024), the fixed part of a launch is 0.59 / 0.63 / 0.81 / 1.17 ms
// -- 0.5 ms of it independent of the claim size (9 % of a rank's step when 8 GPUs share config 2) -- and it is NOT the idle -> busy
// transition (8 launches back to back cost the same each, tools/launch_back_to_back.py).  Tried and dropped: guided claims
// ((paths left) / (2 waves), 256 .. 1024, from a fresh load of the queue head): +4 ms per launch -- the extra load of the hot word costs
// more than a claim -- and a static first claim per wave (no start-up burst of 6144 atomics): -0.05 ms, within noise of its cost.
#ifndef AMBER_CLAIM_PATHS
#define AMBER_CLAIM_PATHS 1024u
#endif
// Pool slot: the whole state of a path between two bounces in four 16-byte chunks {o.xyz d.x} {d.yz w.xy} {w.z rng q}
// {casts | carried-flag, origin slot, signature hashes}.
template <bool kLight> struct PoolLayout { static constexpr int kChunks = 4; };

// Traversal-stack entries of pt_megakernel<ENGINE_BVH> (a deeper tree renders with pt_bvh_megakernel): 12 KB of LDS next to the 16-KB
// pool, five workgroups per CU (16 entries: four).  Where the two schedulers cross over (tools/mesh_workloads.py, 1024^2 @ 256 spp, a room
// with a tessellated ball): Cornell's tree of depth 6, 30.7 ms against 44.2 for pt_bvh_megakernel; 1.3 k triangles, depth 12: 52.7 against
// 54.7; 5 k triangles, depth 14 (built with 16 entries): 64.9 against 57.6 -- beyond depth 12 the resumable item kernel wins.
#ifndef AMBER_PATH_BVH_STACK
#define AMBER_PATH_BVH_STACK 12
#endif
template <int kEngine, bool kLight = false, bool kSig = false>
__global__ void __launch_bounds__(256, kEngine == ENGINE_BVH ? 5 : AMBER_MEGAKERNEL_WAVES_PER_SIMD) pt_megakernel(const RenderArgs a) {
  const DevScene& sc = a.scene;
  const uint32_t lane = threadIdx.x & 63u;
  constexpr bool kTwoPhase = kEngine == ENGINE_TWO_PHASE;
  constexpr int kChunks = PoolLayout<kLight>::kChunks;
  __shared__ DevObject lds_objects[kTwoPhase ? AMBER_MAX_LDS_OBJECTS : 1];
  __shared__ uint4 lds_pool[4][64 * kChunks];                 // [wave][slot * kChunks + chunk]
  // engine BVH on a SHALLOW tree (mid-size scenes, RenderPassPaths): the closest hit is one uninterrupted per-lane traversal (ClosestHitBvh),
  // its stack [level][thread] in LDS -- a few dozen node visits differ little between the lanes of a wave, so nothing has to be resumable,
  // and the path-granular scheduling above (coherent primary rounds, no lane waits for a shading batch) is what a small scene gains most from
  __shared__ int32_t lds_stack[kEngine == ENGINE_BVH ? AMBER_PATH_BVH_STACK * 256 : 1];
  if (kTwoPhase) StageObjects(sc, lds_objects);
  const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and the compiler knows it: what derives from it stays in SGPRs
  uint4* pool = lds_pool[wave_in_block];

  uint32_t claim_next = 0, claim_end = 0;    // wave-uniform: paths claimed from the global queue, not yet generated
  uint32_t pool_count = 0;                   // wave-uniform: rays in the pool (slots [0, pool_count))
  uint32_t rec_next = 0, rec_end = 0;        // wave-uniform: the wave's open block of record slots
  bool exhausted = false;                    // wave-uniform: the global queue is empty
  bool retired = false, alive = false;
  uint32_t q = 0;                            // path of this lane
  // The measurement is NOT kept in registers between bounces: emitted radiance is non-zero only on a DiffuseLight, whose
  // scatter weight 0 ends the path, so a path that continues has measurement +0 -- unless a weight has become inf or NaN
  // (inf * 0 = NaN: degenerate scenes), and such a path parks its measurement in global memory and sets `carries`.
  bool carries = false;
  // (its address is formed where it is used, from a thread index the optimiser cannot see through: hoisted out of the
  //  persistent loop, the pointer is the one value the register cap pushes into scratch)
  auto carried_slot = [&]() -> float* {
    const uint32_t t = wave_in_block * 64u + BvhStackHybrid::LaneId();     // = threadIdx.x, from an SGPR and two v_mbcnt
    return a.carried + (static_cast<size_t>(blockIdx.x) * 256u + t) * 3u;
  };
#define AMBER_CARRIED() carried_slot()
  // a carried measurement travels with its ray: parked with it (a second block of slots, one per pool slot of the wave) and handed to
  // the lane that pops the ray -- rays change lanes since the primary rounds
  auto carried_parked = [&](uint32_t pool_slot) -> float* {
    return a.carried + (static_cast<size_t>(gridDim.x) * 256u + (static_cast<size_t>(blockIdx.x) * 4u + wave_in_block) * 64u + pool_slot) * 3u;
  };
  V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f), w = v3(0.f, 0.f, 0.f);
  uint64_t rng = 1;
  uint32_t casts = 0;
  uint32_t rays_wave = 0;                    // wave-uniform (SGPR): rays cast by this wave
  int origin_slot = -1;                      // filter-program slot of the triangle the current ray starts on
  uint32_t sig_obj = 2166136261u, sig_t = 2166136261u;   // kSig only
#ifdef AMBER_STAMPS
  StampCtx stamp_store{}; StampCtx* stamp_ctx = &stamp_store;
  stamp_ctx->last = __builtin_amdgcn_s_memtime();
  unsigned long long* wave_times = a.stamps ? a.stamps + 8 + 4ull * ((blockIdx.x * 4u + wave_in_block) % AMBER_WAVE_TIME_SLOTS) : nullptr;
  if (wave_times && lane == 0) { wave_times[0] = wall_clock64(); wave_times[1] = 0; wave_times[2] = 0; }
#endif

  for (;;) {
    AMBER_STAMP(6);
    const bool need = !alive && !retired;
    const unsigned long long mask = __ballot(need);
    bool primary = false;                                     // wave-uniform: this iteration's rays are 64 fresh paths whose candidates are known (premask)
    uint32_t premask = 0u;
    if (mask) {                                               // wave-uniform
      const uint32_t n_need = static_cast<uint32_t>(__popcll(mask));
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
      // (1) the parked rays first
      const uint32_t take = pool_count < n_need ? pool_count : n_need;
      if (need && rank < take) {                              // pop
        const uint32_t sl = pool_count - 1u - rank;
        const uint4 c0 = pool[sl * kChunks + 0], c1 = pool[sl * kChunks + 1], c2 = pool[sl * kChunks + 2], c3 = pool[sl * kChunks + 3];
        o = v3(__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z));
        d = v3(__uint_as_float(c0.w), __uint_as_float(c1.x), __uint_as_float(c1.y));
        w = v3(__uint_as_float(c1.z), __uint_as_float(c1.w), __uint_as_float(c2.x));
        rng = static_cast<uint64_t>(c2.y) | (static_cast<uint64_t>(c2.z) << 32);
        q = c2.w;
        casts = c3.x & 0x7fffffffu; carries = (c3.x >> 31) != 0u;
        if (!kLight && carries) { const float* from = carried_parked(sl); float* to = AMBER_CARRIED(); to[0] = from[0]; to[1] = from[1]; to[2] = from[2]; }
        origin_slot = static_cast<int>(c3.y);
        if (kSig) { sig_obj = c3.z; sig_t = c3.w; }
        alive = true;
      }
      __builtin_amdgcn_wave_barrier();                        // (compiler fence: the park below overwrites the slots just read)
      pool_count -= take;
      if (take < n_need && !exhausted) {
        // (2) the pool is empty and lanes still wait: a PRIMARY ROUND.  Every lane that holds a ray parks it in the pool;
        // all 64 lanes then start the next 64 paths of the queue TOGETHER -- 64 samples of one pixel: one coherent iteration
        // (first hits, materials) at full width; the parked rays go to whichever lanes lose their path afterwards.
        AMBER_STAMP(0);
        if (claim_next == claim_end) {                        // claim the next block of paths from the global queue
          const uint32_t size = AMBER_CLAIM_PATHS;                         // (claims sized from what is left of the queue were measured neutral and removed: EXPERIMENTS.md, round 4)
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(a.next_item, size);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base >= a.n_items) exhausted = true;
          else { claim_next = base; claim_end = a.n_items - base < size ? a.n_items : base + size; }
#ifdef AMBER_STAMPS
          if (wave_times && lane == 0) { if (exhausted) wave_times[2] = wall_clock64(); else if (wave_times[1] == 0) wave_times[1] = wall_clock64(); }
#endif
        }
        if (!exhausted) {
          const uint32_t n_new = claim_end - claim_next < 64u ? claim_end - claim_next : 64u;
          const uint32_t gbase = claim_next;
          claim_next += n_new;
          const unsigned long long m_alive = __ballot(alive);
          if (alive) {                                        // park (pool_count is 0 here)
            const uint32_t sl = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m_alive >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m_alive), 0u));
            pool[sl * kChunks + 0] = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(d.x));
            pool[sl * kChunks + 1] = make_uint4(__float_as_uint(d.y), __float_as_uint(d.z), __float_as_uint(w.x), __float_as_uint(w.y));
            pool[sl * kChunks + 2] = make_uint4(__float_as_uint(w.z), static_cast<uint32_t>(rng), static_cast<uint32_t>(rng >> 32), q);
            pool[sl * kChunks + 3] = make_uint4(casts | (carries ? 0x80000000u : 0u), static_cast<uint32_t>(origin_slot), sig_obj, sig_t);
            if (!kLight && carries) { const float* from = AMBER_CARRIED(); float* to = carried_parked(sl); to[0] = from[0]; to[1] = from[1]; to[2] = from[2]; }
          }
          pool_count = static_cast<uint32_t>(__popcll(m_alive));
          __builtin_amdgcn_wave_barrier();
          alive = lane < n_new;
          if (alive) {
            q = gbase + lane;
            const uint32_t plocal = q / a.n_samples, k = q - plocal * a.n_samples;
            if (kLight) {
              rng = XorShiftSeed(a.hashed_seed, a.path_offset + plocal, a.first_sample + k);
              GenerateLightRay(sc, rng, o, d, w, origin_slot);
            } else {
              const uint32_t lrow = plocal / sc.sensor.w;
              const uint32_t px = plocal - lrow * sc.sensor.w;
              const uint32_t py = a.row_begin + (a.stripe_rows ? (lrow / a.stripe_rows) * a.stripe_period + lrow % a.stripe_rows : lrow);
              rng = XorShiftSeed(a.hashed_seed, px + py * sc.sensor.w, a.first_sample + k);   // Image index x + y*W (image.h:116-124)
              float ew;
              bool near_edge = false;
              GenerateEyeRay(sc, px, py, rng, o, d, ew, origin_slot, &near_edge);
              w = v3(ew, ew, ew);                             // Leading<RGB>(.., Radiant(weight)) lens_basic.h:139-144
              if (kTwoPhase && a.pixel_mask) {
                // the candidates of this pixel's beam, plus the ray's own aperture blade (the self trip decides it exactly), plus
                // every blade when the aperture sample lies on a blade's boundary (only then can a neighbour's exact test see it)
                premask = a.pixel_mask[plocal] | (origin_slot >= 0 ? 1u << origin_slot : 0u) | (near_edge ? sc.blade_mask : 0u);
              }
            }
            carries = false;
            casts = 0;
            if (kSig) { sig_obj = 2166136261u; sig_t = 2166136261u; }
          }
          primary = !kLight && kTwoPhase && a.pixel_mask != nullptr;
          AMBER_STAMP(1);
        }
      }
      if (!alive && !retired && pool_count == 0u && exhausted) retired = true;   // queue and pool empty: this lane retires
      if (__ballot(!retired) == 0ull) break;
    }

    rays_wave += static_cast<uint32_t>(__popcll(__ballot(alive)));
    bool emit = false;
    V3 meas = v3(0.f, 0.f, 0.f);
    if (alive) {
      if (!kLight && carries) meas = ld3(AMBER_CARRIED());
      if (kLight) {
        const uint32_t plocal = q / a.n_samples;
        const SplatSink sink{a.splats, a.splat_count, a.splat_capacity, a.path_offset + plocal, a.first_sample + (q - plocal * a.n_samples), sc.sensor.size_f};
        alive = PathStep<false, kEngine, true>(sc, lds_objects, lds_stack, o, d, w, meas, rng, casts, origin_slot, nullptr AMBER_STAMP_ARG, &sink, false, 0u, AMBER_PATH_BVH_STACK);
      } else if (kSig) {
        Bounce b;
        alive = PathStep<true, kEngine>(sc, lds_objects, lds_stack, o, d, w, meas, rng, casts, origin_slot, &b AMBER_STAMP_ARG, nullptr, primary, premask, AMBER_PATH_BVH_STACK);
        sig_obj = Fnv32(sig_obj, static_cast<uint32_t>(b.object));
        if (b.object >= 0) sig_t = Fnv32(sig_t, __float_as_uint(b.t));
        if (!alive) a.sig[q] = static_cast<unsigned long long>(sig_obj) | (static_cast<unsigned long long>(sig_t) << 32);
      } else {
        alive = PathStep<false, kEngine>(sc, lds_objects, lds_stack, o, d, w, meas, rng, casts, origin_slot, nullptr AMBER_STAMP_ARG, nullptr, primary, premask, AMBER_PATH_BVH_STACK);
      }
      if (!kLight) {
        const bool nz = (__float_as_uint(meas.x) | __float_as_uint(meas.y) | __float_as_uint(meas.z)) != 0u;   // anything but +0 (RGB)
        emit = !alive && nz;
        if (alive && nz) { float* c = AMBER_CARRIED(); c[0] = meas.x; c[1] = meas.y; c[2] = meas.z; carries = true; }
      }
    }
    if (!kLight) EmitRecords(a, emit, q, meas, rec_next, rec_end);
  }

#undef AMBER_CARRIED
  if (!kLight) CloseRecords(a, rec_next, rec_end);
#ifdef AMBER_STAMPS
  if (lane == 0 && a.stamps) for (int k = 0; k < 8; k++) atomicAdd(a.stamps + k, stamp_ctx->acc[k]);
  if (wave_times && lane == 0) wave_times[3] = wall_clock64();
#endif
  // one atomic per wave for the ray counter
  if (lane == 0 && rays_wave) atomicAdd(a.ray_count, static_cast<unsigned long long>(rays_wave));
}

// ---- from records to ordered per-pixel sums ------------------------------------------------------------------------
// The records of a launch arrive in no order.  The bitmap gives every record its place: record of path q goes to
//   sorted[ base(pixel of q) + (set bits of that pixel below q) ],   base(p) = set bits of all pixels before p,
// so reduce_flagged_kernel can walk a pixel's bits in sample order and read its measurements consecutively.
//   rec_rank_kernel        per pixel: number of set bits, exclusive prefix within the workgroup, workgroup total
//   rec_scan_blocks_kernel one workgroup: exclusive prefix over the workgroup totals; also folds the launch's ray count
//                          into the handle's total
//   rec_place_kernel       per record slot: q -> place, copy the measurement
// All of them, and the reduction, do NOTHING when the launch ran out of record slots (*rec_count > capacity): the host
// then repeats the launch with a larger buffer and the framebuffer and ray total have not been touched.
__device__ __forceinline__ uint32_t PixelBits(const uint32_t* __restrict__ flags, uint32_t q0, uint32_t q1) {   // set bits in [q0, q1)
  if (q0 == q1) return 0u;
  const uint32_t w0 = q0 >> 5, w1 = (q1 - 1u) >> 5;
  const uint32_t lo = 0xffffffffu << (q0 & 31u), hi = 0xffffffffu >> (31u - ((q1 - 1u) & 31u));
  if (w0 == w1) return static_cast<uint32_t>(__popc(flags[w0] & lo & hi));
  uint32_t c = static_cast<uint32_t>(__popc(flags[w0] & lo)) + static_cast<uint32_t>(__popc(flags[w1] & hi));
  for (uint32_t wd = w0 + 1u; wd < w1; ++wd) c += static_cast<uint32_t>(__popc(flags[wd]));
  return c;
}
__global__ void rec_rank_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ touched, uint32_t n_pixels, uint32_t n_samples,
                                uint32_t* __restrict__ excl, uint32_t* __restrict__ block_sum) {
  __shared__ uint32_t part[256];
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  // only a touched pixel's bits are read: on the Cornell box that is 2e-2 of the pixels, and the bitmap is 128 MB per launch
  const bool any = p < n_pixels && ((touched[p >> 5] >> (p & 31u)) & 1u) != 0u;
  const uint32_t c = any ? PixelBits(flags, p * n_samples, p * n_samples + n_samples) : 0u;
  part[threadIdx.x] = c;
  __syncthreads();
  for (uint32_t off = 1; off < 256u; off <<= 1) {             // Hillis-Steele inclusive scan
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  if (p < n_pixels) excl[p] = part[threadIdx.x] - c;
  if (threadIdx.x == 255u) block_sum[blockIdx.x] = part[255];
}
__global__ void rec_scan_blocks_kernel(uint32_t* __restrict__ block_sum, uint32_t n_blocks, const unsigned int* __restrict__ rec_count, uint32_t rec_capacity,
                                       unsigned long long* __restrict__ ray_total, unsigned long long* __restrict__ ray_launch) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) {
    carry = 0u;
    if (*rec_count <= rec_capacity) *ray_total += *ray_launch;            // the launch stands: its rays count
  }
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t c = i < n_blocks ? block_sum[i] : 0u;
    part[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
      const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    if (i < n_blocks) block_sum[i] = carry + part[threadIdx.x] - c;
    __syncthreads();
    if (threadIdx.x == 1023u) carry += part[1023];
    __syncthreads();
  }
}
__global__ void rec_place_kernel(const uint4* __restrict__ records, const unsigned int* __restrict__ rec_count, uint32_t rec_capacity, const uint32_t* __restrict__ flags,
                                 const uint32_t* __restrict__ excl, const uint32_t* __restrict__ block_sum, uint32_t n_samples, float* __restrict__ sorted) {
  const uint32_t n = *rec_count;
  if (n > rec_capacity) return;                               // out of slots: the launch is repeated
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint4 r = records[k];
    if (r.x == AMBER_REC_UNUSED) continue;
    const uint32_t p = r.x / n_samples;
    const uint32_t at = block_sum[p >> 8] + excl[p] + PixelBits(flags, p * n_samples, r.x);
    float* s = sorted + static_cast<size_t>(at) * 3u;
    s[0] = __uint_as_float(r.y); s[1] = __uint_as_float(r.z); s[2] = __uint_as_float(r.w);
  }
}

// fb[pixel] += the measurements of the launch's paths, in the order of the numerical contract: within a chunk of
// AMBER_ACCUM_CHUNK consecutive samples (counted from the launch's first sample) a sequential binary32 sum, chunk sums
// added to the framebuffer value in chunk order.  Only paths whose bit is set contribute a term; every other term is +0,
// and both the chunk sum (starts at +0) and the framebuffer value (cleared to +0) can never be -0, so leaving the +0
// terms out changes no bit.  One thread per band pixel; its measurements are sorted[base ..] in sample order.
__global__ void reduce_flagged_kernel(float* __restrict__ fb, uint32_t* __restrict__ flags, const uint32_t* __restrict__ touched, const float* __restrict__ sorted,
                                      const uint32_t* __restrict__ excl, const uint32_t* __restrict__ block_sum, const unsigned int* __restrict__ rec_count,
                                      uint32_t rec_capacity, uint32_t n_pixels, uint32_t n_samples) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pixels) return;
  if (*rec_count > rec_capacity) return;                       // out of slots: the launch is repeated
  if (((touched[p >> 5] >> (p & 31u)) & 1u) == 0u) return;     // no path of this pixel carried a measurement
  const uint32_t q0 = p * n_samples;                           // n_pixels * n_samples < 2^32 (RenderPassPaths splits launches)
  const float* m = sorted + static_cast<size_t>(block_sum[p >> 8] + excl[p]) * 3u;
  float v0 = fb[3u * p], v1 = fb[3u * p + 1u], v2 = fb[3u * p + 2u];
  if ((n_samples & 31u) == 0u && (32u % AMBER_ACCUM_CHUNK) == 0u) {
    // The pixel owns whole words: walk the SET bits only (a touched pixel of the Cornell box has one or two among 1024) -- the same
    // sums in the same order: samples of a chunk in order from +0, chunk sums onto the pixel in chunk order.  The bit-by-bit loop
    // below took 0.135 ms per launch of a rank's share of config 2 on 8 GPUs, 2 % of its step.  Consumed words are cleared on the
    // way, so that the bitmap is all zero again for the next launch and the host does not have to clear 128 MB per launch.
    const uint32_t w0 = q0 >> 5, nw = n_samples >> 5;
    constexpr uint32_t kChunkMask = AMBER_ACCUM_CHUNK >= 32u ? 0xffffffffu : ((1u << (AMBER_ACCUM_CHUNK & 31u)) - 1u);
    auto consume = [&](uint32_t wd, uint32_t bits) {
      if (bits == 0u) return;
      flags[w0 + wd] = 0u;
      for (uint32_t sh = 0; sh < 32u; sh += AMBER_ACCUM_CHUNK) {
        uint32_t b = (bits >> sh) & kChunkMask;
        if (b == 0u) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        while (b) { b &= b - 1u; s0 = s0 + m[0]; s1 = s1 + m[1]; s2 = s2 + m[2]; m += 3; }
        v0 = v0 + s0; v1 = v1 + s1; v2 = v2 + s2;
      }
    };
    uint32_t wd = 0;
    if ((w0 & 3u) == 0u) {                                     // four words per load: a pixel's 1024 samples are 8 loads issued together, not 32 in turn
      for (; wd + 16u <= nw; wd += 16u) {
        const uint4* f4 = reinterpret_cast<const uint4*>(flags + w0 + wd);
        const uint4 a0 = f4[0], a1 = f4[1], a2 = f4[2], a3 = f4[3];
        if ((a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w | a2.x | a2.y | a2.z | a2.w | a3.x | a3.y | a3.z | a3.w) == 0u) continue;
        consume(wd + 0u, a0.x); consume(wd + 1u, a0.y); consume(wd + 2u, a0.z); consume(wd + 3u, a0.w);
        consume(wd + 4u, a1.x); consume(wd + 5u, a1.y); consume(wd + 6u, a1.z); consume(wd + 7u, a1.w);
        consume(wd + 8u, a2.x); consume(wd + 9u, a2.y); consume(wd + 10u, a2.z); consume(wd + 11u, a2.w);
        consume(wd + 12u, a3.x); consume(wd + 13u, a3.y); consume(wd + 14u, a3.z); consume(wd + 15u, a3.w);
      }
    }
    for (; wd < nw; ++wd) consume(wd, flags[w0 + wd]);
    fb[3u * p] = v0; fb[3u * p + 1u] = v1; fb[3u * p + 2u] = v2;
    return;
  }
  for (uint32_t c0 = 0; c0 < n_samples; c0 += AMBER_ACCUM_CHUNK) {
    const uint32_t c1 = c0 + AMBER_ACCUM_CHUNK < n_samples ? c0 + AMBER_ACCUM_CHUNK : n_samples;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    bool hit = false;
    for (uint32_t k = c0; k < c1; ++k) {
      const uint32_t q = q0 + k;
      if ((flags[q >> 5] >> (q & 31u)) & 1u) {
        s0 = s0 + m[0]; s1 = s1 + m[1]; s2 = s2 + m[2];
        m += 3;
        hit = true;
      }
    }
    if (hit) { v0 = v0 + s0; v1 = v1 + s1; v2 = v2 + s2; }
  }
  fb[3u * p] = v0; fb[3u * p + 1u] = v1; fb[3u * p + 2u] = v2;
  // (a pixel that does not own whole words leaves its bits: the host clears the bitmap before the next launch, LaunchPaths)
}

// ---- per-pixel candidate masks of the primary rays (two-phase engine) ---------------------------------------------------------
// All eye rays of pixel (px, py) lie in one thin BEAM: every ray starts in the aperture (thin lens: lens_thin.cc:70-107; inside the
// rectangle A that bounds the blades in the lens plane) and passes through the image of its jittered sensor point on the focal plane
// (inside the rectangle F: the image of the pixel; the direction is Normalize(-fd/sd * sensor_point - aperture_point), which does not
// depend on where in the aperture the ray starts), or starts at the pinhole (A = one point).  The mask of a pixel has a bit for every
// object of the filter program that SOME ray of that beam can hit -- the closest hit of every one of the pixel's eye rays, whatever its
// sample index, is then found among the mask's objects, and a primary round needs no Phase A.  Conservative by construction:
//  * plane records: the hit points of the beam on a plane form a convex region whenever the plane cuts neither A nor F and no ray is
//    close to parallel to it (before the focal plane the beam is conv(A u F), behind it F + cone(F - A): both convex, and a plane that
//    separates or misses A and F meets them only in edges that lie on CORNER rays a_i -> f_j); barycentric coordinates are affine on
//    the plane, so each one's maximum over the region is attained on one of the 16 corner rays.  A triangle is dropped only if one of
//    its three coordinates stays below -tolerance on all 16, or if every corner ray meets the plane behind its origin.  Planes that cut
//    A or F, or that a corner ray grazes, keep all their triangles.
//  * spheres: kept unless the central ray passes the centre farther away than the radius plus the beam's half width there.
//  * the same tolerances as Phase A (DevPlane.kt / ktol scaled by the largest 1 / |n.d| of the corner rays) plus a position slack of
//    1e-5 of the model size for the difference between a binary32 eye ray and the ideal one; A is inflated by 1e-3 of its size, F by
//    2 % of the pixel.  Aperture blades are NOT part of the mask: a ray's own blade and, on blade boundaries, its neighbours are added
//    per ray where the path starts.
// Double precision throughout, once per handle (the mask depends on scene and sensor only).  Since round 5 one mask serves a BLOCK of 4 x 4 band
// pixels (F = the image of the whole block): a pixel's beam is as wide as the aperture nearly everywhere, so sixteen neighbours share their
// candidates -- config 2: 2.10 candidates per pixel instead of 1.97, the render kernel unchanged (tools/pixel_mask_blocks.py,
// profiles/r05_pixel_mask_blocks.txt), and the kernel takes 0.07 ms instead of 1.0.
struct PixelMaskArgs {
  double ap[4][3];                // corners of the aperture's bounding rectangle in the lens plane (inflated), RELATIVE TO THE FILTER CENTRE; pinhole: the origin
  double inv_w, inv_h;            // 1 / sensor width, height (pixels)
  double focal_scale;             // thin lens: -focus_distance / sensor_distance; pinhole: -(8 reach + 1) / sensor_distance (a point far along the ray)
  uint32_t planes_cut_a;          // bit i: plane i of the filter program passes within the slack of the aperture rectangle (the same for every pixel: tested on the host)
  uint32_t row_begin, stripe_rows, stripe_period, n_pixels;
  uint32_t block;                 // a mask serves a block of block x block band pixels (4, or 1 when the stripes are not whole blocks): the beam of a pixel's eye rays is as
  uint32_t blocks_x, n_blocks;    // wide as the APERTURE nearly everywhere, so 16 pixels share their candidates at a sixteenth of the work (round 5: 1.0 -> < 0.1 ms)
  uint32_t local_rows;
};
// SIXTEEN LANES PER BLOCK OF PIXELS, lane k = corner ray a_(k >> 2) -> f_(k & 3) (round 4: per pixel).  Round 3's kernel ran one thread per pixel with the 16
// directions and the 16 plane points in per-thread arrays: 128 VGPRs, 1 728 B of scratch, 3.7 ms per config-2 handle -- more than the first
// 64 samples per pixel of the render it prepares.  Here a lane keeps ITS ray (3 + 3 doubles); "on all 16 corner rays" is a vote
// (__ballot, masked to the pixel's 16 lanes), the maxima a plane needs are butterfly reductions over the 16 lanes, the aperture
// corners are wave-uniform kernel arguments (SGPRs), and a lane forms only its own corner of F.  Same tests as before with two
// differences, both towards keeping a candidate: a NaN coordinate keeps its triangle (fmax used to skip it), and the gradient norms
// of a record's affine rows are bounded by their 1-norms (no square roots).
__device__ __forceinline__ double Max16(double x) {               // maximum over the 16 lanes of a pixel (lanes 16 g .. 16 g + 15)
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) x = fmax(x, __shfl_xor(x, m, 16));
  return x;
}
__global__ void __launch_bounds__(256) pixel_mask_kernel(const DevScene sc, const PixelMaskArgs pm, uint32_t* __restrict__ out) {
  const uint32_t b_raw = blockIdx.x * 16u + (threadIdx.x >> 4), k = threadIdx.x & 15u;     // 16 pixel blocks per workgroup of 256
  const bool live = b_raw < pm.n_blocks;
  const uint32_t b = live ? b_raw : pm.n_blocks - 1u;              // whole groups of 16 lanes stay active (votes, shuffles); only `live` groups store
  const uint32_t shift = threadIdx.x & 48u;                        // position of the block's 16 lanes in the wave's 64-bit votes
  auto all16 = [&](bool c) { return ((__ballot(c) >> shift) & 0xffffull) == 0xffffull; };
  auto any16 = [&](bool c) { return ((__ballot(c) >> shift) & 0xffffull) != 0ull; };
  const uint32_t brow = b / pm.blocks_x, lrow = brow * pm.block, px = (b - brow * pm.blocks_x) * pm.block;   // first band row and column of the block
  const uint32_t py = pm.row_begin + (pm.stripe_rows ? (lrow / pm.stripe_rows) * pm.stripe_period + lrow % pm.stripe_rows : lrow);   // (a block never straddles two stripes)
  const double span = static_cast<double>(pm.block);              // the block covers pixels [px, px + span) x [py, py + span)
  const DevLens& L = *sc.lens;
  const double cx = sc.fp_center[0], cy = sc.fp_center[1], cz = sc.fp_center[2];
  const double reach = sc.fp_reach;
  // Position slack (world units): 1e-5 of the model size, plus the binary32 grid of WORLD coordinates -- an eye ray is defined by the
  // reference's binary32 arithmetic on world coordinates (blade point, aperture_point = local (point - lens origin), ...), so in a scene
  // that lies 1e4 of its size from the world origin (the fuzzer's --scaled scenes) the actual ray is a few 1e-4 of the scene away from
  // the ideal one.  The same amount widens A (host) and F (below).
  const double world_mag = fmax(fmax(fabs(cx), fmax(fabs(cy), fabs(cz))) + reach, fmax(fabs(double(L.origin[0])), fmax(fabs(double(L.origin[1])), fabs(double(L.origin[2])))));
  const double slack = 1e-5 * reach + 32.0 * 5.9604644775390625e-08 * world_mag;
  // this lane's corner ray: from a_ = A[k >> 2] towards fk = F[k & 3], the corner (k & 1, k & 2) of the pixel's image on the focal plane
  const uint32_t ki = k >> 2, kj = k & 3u;
  double a_[3], fk[3];
  for (int c = 0; c < 3; c++) a_[c] = ki == 0u ? pm.ap[0][c] : (ki == 1u ? pm.ap[1][c] : (ki == 2u ? pm.ap[2][c] : pm.ap[3][c]));
  {
    // (an IEEE binary64 division is 64 SIMD cycles on gfx950: the wave-uniform quotients are kernel arguments)
    const double ux = (static_cast<double>(px) + ((kj & 1u) ? span + 0.02 : -0.02)) * pm.inv_w;
    const double uy = (static_cast<double>(py) + ((kj & 2u) ? span + 0.02 : -0.02)) * pm.inv_h;
    const double sx = (ux - 0.5) * sc.sensor.sw, sy = (uy - 0.5) * sc.sensor.sh, sz = L.sensor_distance;
    // thin lens: the point of the focal plane all rays of this sensor point pass through; pinhole: a point far along the ray
    const double kk = pm.focal_scale;
    const double fl[3] = {kk * sx + ((kk * ((kj & 1u) ? 1.0 : -1.0) >= 0) ? slack : -slack), kk * sy + ((kk * ((kj & 2u) ? 1.0 : -1.0) >= 0) ? slack : -slack), kk * sz};   // outwards by the slack
    for (int c = 0; c < 3; c++) fk[c] = L.origin[c] + L.global_[3 * c] * fl[0] + L.global_[3 * c + 1] * fl[1] + L.global_[3 * c + 2] * fl[2];
    fk[0] -= cx; fk[1] -= cy; fk[2] -= cz;
  }
  double Dk[3];
  {
    const double v[3] = {fk[0] - a_[0], fk[1] - a_[1], fk[2] - a_[2]};
    const double inv_l = 1.0 / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int c = 0; c < 3; c++) Dk[c] = v[c] * inv_l;                  // unit up to two roundings: far inside every tolerance below
  }
  uint32_t mask = 0u, bit = 1u;
  const DevPlane* pl = sc.planes;
  const DevTriFilter* tr = sc.tri_filters;
  for (uint32_t pi = 0; pi < sc.n_planes; ++pi, ++pl) {
    const double n[3] = {pl->n[0], pl->n[1], pl->n[2]}, d0 = pl->d0;
    const uint32_t nt = pl->n_tris & 0x7fffffffu, np = pl->n_pairs;
    // does the plane cut A or F?  (all corners strictly on one side, with the slack; the lanes of a pixel hold every corner of both)
    bool all = false;
    {
      const double sf = n[0] * fk[0] + n[1] * fk[1] + n[2] * fk[2] - d0;
      if (!(all16(sf > 10.0 * slack) || all16(sf < -10.0 * slack))) all = true;
      if ((pm.planes_cut_a >> (pi & 31u)) & 1u) all = true;            // the aperture's side of the same question: pixel-independent
    }
    const double nd = n[0] * Dk[0] + n[1] * Dk[1] + n[2] * Dk[2];
    if (any16(!(fabs(nd) >= 4.0 * AMBER_GRAZING))) all = true;       // a corner ray grazes the plane (or NaN): no convexity argument
    if (any16(nd > 0) && any16(!(nd > 0))) all = true;                // corner rays meet the plane from both sides
    if (all) {                                                         // (uniform over the pixel's 16 lanes)
      for (uint32_t r = 0; r < 2u * np + nt; r++, bit <<= 1) mask |= bit;
      tr += np + nt;
      continue;
    }
    const double inv_nd = 1.0 / nd;
    const double t = (d0 - (n[0] * a_[0] + n[1] * a_[1] + n[2] * a_[2])) * inv_nd;
    const double P[3] = {a_[0] + t * Dk[0], a_[1] + t * Dk[1], a_[2] + t * Dk[2]};
    const double rho_max = Max16(fabs(inv_nd)) * 1.000001, t_max = Max16(t);
    if (t_max < static_cast<double>(AMBER_KEPS) - static_cast<double>(pl->kt) * rho_max - slack) {   // every ray meets the plane behind its origin
      bit <<= (2u * np + nt); tr += np + nt;
      continue;
    }
    const double ptol = static_cast<double>(pl->ktol) * rho_max;
    for (uint32_t r = 0; r < np + nt; ++r, ++tr) {
      // |gradient| of the two affine rows, bounded by their 1-norms
      const double g0 = fabs(double(tr->c[0][0])) + fabs(double(tr->c[1][0])) + fabs(double(tr->c[2][0]));
      const double g1 = fabs(double(tr->c[0][1])) + fabs(double(tr->c[1][1])) + fabs(double(tr->c[2][1]));
      const double tol = ptol + (g0 + g1) * slack + 1e-6;               // barycentric units
      const double u = tr->c[0][0] * P[0] + tr->c[1][0] * P[1] + tr->c[2][0] * P[2] + tr->c[3][0];
      const double v = tr->c[0][1] * P[0] + tr->c[1][1] * P[1] + tr->c[2][1] * P[2] + tr->c[3][1];
      // a triangle is dropped only if ONE of its coordinates is below -tol on all 16 corner rays.
      // pair record: (u, v) = (beta, alpha) of the first triangle; the second one's coordinates are (-beta, 1 - alpha, beta + alpha)
      const bool drop1 = all16(u < -tol) || all16(v < -tol) || all16(1.0 - u - v < -tol);   // a NaN coordinate votes "keep"
      mask |= drop1 ? 0u : bit; bit <<= 1;
      if (r < np) {
        const bool drop2 = all16(-u < -tol) || all16(1.0 - v < -tol) || all16(u + v < -tol);
        mask |= drop2 ? 0u : bit; bit <<= 1;
      }
    }
  }
  // spheres: the central ray against the sphere inflated by the beam's half width at the sphere's depth
  if (sc.n_sphere_filters) {
    double ac[3] = {0, 0, 0}, fc[3] = {0, 0, 0};
    for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++) { ac[c] += 0.25 * pm.ap[i][c]; fc[c] += 0.25 * __shfl(fk[c], i, 16); }   // lanes 0..3 of the pixel hold F[0..3]
    double dc[3] = {fc[0] - ac[0], fc[1] - ac[1], fc[2] - ac[2]};
    const double inv_lc = 1.0 / sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
    for (int c = 0; c < 3; c++) dc[c] *= inv_lc;
    const double inv_dd = 1.0 / (Dk[0] * dc[0] + Dk[1] * dc[1] + Dk[2] * dc[2]);
    const double off = (a_[0] - ac[0]) * dc[0] + (a_[1] - ac[1]) * dc[1] + (a_[2] - ac[2]) * dc[2];
    auto half_width = [&](double tau) {                                // largest distance of a corner ray from the central one in the plane at depth tau
      const double s_ = (tau - off) * inv_dd;
      double e2 = 0;
      for (int c = 0; c < 3; c++) { const double e = a_[c] + s_ * Dk[c] - (ac[c] + tau * dc[c]); e2 += e * e; }
      return sqrt(Max16(fmax(0.0, e2)));                               // (fmax(0, NaN) = 0: a NaN ray is skipped, as the serial loop's fmax did)
    };
    const DevSphereFilter* sp = sc.sphere_filters;
    for (uint32_t r_ = 0; r_ < sc.n_sphere_filters; ++r_, ++sp, bit <<= 1) {
      const double co[3] = {sp->c[0] - ac[0], sp->c[1] - ac[1], sp->c[2] - ac[2]};
      const double r = sqrt(static_cast<double>(sp->r2)) * 1.001 + slack;
      const double tau = co[0] * dc[0] + co[1] * dc[1] + co[2] * dc[2];
      const double d2 = co[0] * co[0] + co[1] * co[1] + co[2] * co[2] - tau * tau;
      const double hw = fmax(half_width(tau - r), half_width(tau + r));
      const double lim = (r + hw) * 1.02 + slack;
      const bool miss = d2 > lim * lim;                                  // NaN -> keep
      mask |= miss ? 0u : bit;
    }
  }
  // lane k stores the block's mask for its pixel (k % block, k / block) of the block, where the band has one
  const uint32_t sx = px + k % pm.block, sr = lrow + k / pm.block;
  if (live && k < pm.block * pm.block && sx < sc.sensor.w && sr < pm.local_rows) out[sr * sc.sensor.w + sx] = mask & ~sc.blade_mask;
}

// Engine BVH worker.  Same work queue, item walk and accumulation order as pt_megakernel, but the closest-hit query
// is a per-lane tree traversal whose length varies by an order of magnitude between the lanes of a wave (1M-sphere
// scene: 24 % VALU lane utilisation when every bounce waits for the slowest lane).  Traversal is therefore
// RESUMABLE (BvhTrav in registers, stack in LDS): the wave keeps running traversal rounds until
// AMBER_BVH_SHADE_BATCH lanes have finished theirs (or nobody traverses any more), shades exactly those lanes --
// hit resolution, material sampling, Russian roulette, regeneration -- starts their next rays and resumes.  Lanes
// never wait for more than a batch to fill; results do not depend on the schedule (paths are independent and a
// lane's samples are still summed in order).
#ifndef AMBER_BVH_SHADE_BATCH
#define AMBER_BVH_SHADE_BATCH 52   // config 3 at 128 spp, one process (descent budget 5): 24 -> 173 ms, 32 -> 172, 40 -> 166, 48 -> 162, 52 -> 160, 56 -> 160.5, 60 -> 164, 64 (wait for all lanes) -> 181
#endif
// kStack: entries of the per-lane LDS stack (chosen from the tree depth at create time).  24 entries = 24 KB per
// workgroup let 5 workgroups share a CU (96 VGPRs), 32 entries only 4: config 3, 128 spp: 193 vs 208 ms.
#ifndef AMBER_BVH_WGS
#define AMBER_BVH_WGS 5
#endif
// kSig (tests only, amber_hip_pt_signatures): the same kernel also hashes every cast's hit object and hit distance and stores the pair
// at the path's end -- config 3's PRODUCT kernel compared with the oracle path by path, like pt_megakernel's signature instantiation.
template <bool kLight, int kStack, bool kSig = false>
__global__ void __launch_bounds__(256, kStack <= 24 && !kSig ? AMBER_BVH_WGS : 4) pt_bvh_megakernel(const RenderArgs a) {
  const DevScene& sc = a.scene;
  const uint32_t lane = threadIdx.x & 63u;
  __shared__ int32_t lds_stack[kStack * 256];
  // the item's running sum is touched once per PATH (2.6 rays): it lives in LDS ([component][thread]: conflict-free), not in three of
  // the 96 registers the traversal loop is short of -- the compiler used to keep it in scratch
  __shared__ float lds_sum[6 * 256];                       // rows 0-2: the item's sum; rows 3-5: the measurement of the path in flight.  (Two more rows for y, z of the
                                                           // throughput -- what the compiler spills next -- make 32 KB per workgroup: five no longer fit a CU, 113 ms instead of 101 at 128 spp)
  float* const my_sum = lds_sum + threadIdx.x;

  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;
  bool lane_done = false, have_item = false, alive = false, traversing = false;
  uint32_t s = 0, s_end = 0, pixel = 0, slot = 0;          // the pixel's x, y are recomputed where a path starts: two registers less
  V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f), w = v3(0.f, 0.f, 0.f);
  uint64_t rng = 1;
  uint32_t casts = 0;
  uint32_t rays_wave = 0;                                   // wave-uniform (SGPR): rays shaded by this wave
  int origin_slot = -1;
  uint32_t sig_obj = 2166136261u, sig_t = 2166136261u;      // kSig only
  BvhTrav tr; tr.A = v3(0.f, 0.f, 0.f); tr.b_in = v3(0.f, 0.f, 0.f); tr.b_out = v3(0.f, 0.f, 0.f); tr.neg_slack = 0.f; tr.rot[0] = tr.rot[1] = tr.rot[2] = 0u;
  tr.cur = AMBER_BVH_DONE; tr.pend = 0; tr.sp = 0; tr.overflow = false;
  HitRec hit; hit.t = 0.f; hit.u = 0.f; hit.v = 0.f; hit.idx = -1; hit.slot = -1;
#ifdef AMBER_STAMPS
  StampCtx stamp_store{}, stamp_shade{}; StampCtx* stamp_ctx = &stamp_store;     // counters here; PathShade's clocks go to a dummy
#define AMBER_SHADE_STAMP_ARG , &stamp_shade
  stamp_ctx->last = __builtin_amdgcn_s_memtime();
#else
#define AMBER_SHADE_STAMP_ARG
#endif

  for (;;) {
    bool need = !alive && s >= s_end && !lane_done;
    if (need && have_item) {                                // item finished: publish its sum
      if (!kLight) {
        float* p = a.partial + static_cast<size_t>(slot) * 3u;
        p[0] = my_sum[0]; p[1] = my_sum[256]; p[2] = my_sum[512];
      }
      have_item = false;
    }
    unsigned long long mask = __ballot(need);
    while (mask) {                                          // hand out work items (see pt_megakernel)
      const uint32_t avail = pool_end - pool_next;
      if (avail == 0) {
        if (exhausted) break;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.next_item, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= a.n_items) { exhausted = true; break; }
        pool_next = base;
        pool_end = base + 64u < a.n_items ? base + 64u : a.n_items;
        continue;
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
      if (need && rank < avail) {
        const uint32_t item = pool_next + rank;
        const uint32_t plocal = item / a.n_chunks, chunk = item - plocal * a.n_chunks;
        const uint32_t lrow = plocal / sc.sensor.w;
        const uint32_t px = plocal - lrow * sc.sensor.w;
        const uint32_t py = a.row_begin + (a.stripe_rows ? (lrow / a.stripe_rows) * a.stripe_period + lrow % a.stripe_rows : lrow);
        pixel = kLight ? a.path_offset + plocal : px + py * sc.sensor.w;      // light tracing: the light path's index
        slot = chunk * a.n_pixels + plocal;
        s = a.first_sample + chunk * AMBER_ACCUM_CHUNK;
        const uint32_t left = a.first_sample + a.n_samples - s;
        s_end = s + (left < AMBER_ACCUM_CHUNK ? left : AMBER_ACCUM_CHUNK);
        my_sum[0] = 0.f; my_sum[256] = 0.f; my_sum[512] = 0.f;
        have_item = true;
        need = false;
      }
      const uint32_t wanted = static_cast<uint32_t>(__popcll(mask));
      pool_next += wanted < avail ? wanted : avail;
      mask = __ballot(need);
    }
    if (need) lane_done = true;
    if (__ballot(!lane_done) == 0ull) break;

    AMBER_CLK(0);
    if (!alive && !lane_done) {                             // regenerate: next sample of the item, start its traversal
      AMBER_COUNT_ROUNDS(3);
      rng = XorShiftSeed(a.hashed_seed, pixel, s);
      if (kLight) {
        GenerateLightRay(sc, rng, o, d, w, origin_slot);
      } else {
        float ew;
        const uint32_t py = pixel / sc.sensor.w;
        GenerateEyeRay(sc, pixel - py * sc.sensor.w, py, rng, o, d, ew, origin_slot);
        w = v3(ew, ew, ew);
      }
      my_sum[768] = 0.f; my_sum[1024] = 0.f; my_sum[1280] = 0.f;
      casts = 0;
      alive = true;
      ++s;
      if (kSig) { sig_obj = 2166136261u; sig_t = 2166136261u; }
      BvhBegin(sc, o, d, tr, hit);
      traversing = true;
    }

    AMBER_CLK(1);
    for (;;) {                                              // traversal rounds until a batch of lanes is ready to shade
      const unsigned long long tm = __ballot(traversing);
      if (tm == 0ull) break;
      if (static_cast<uint32_t>(__popcll(__ballot(alive && !traversing))) >= a.shade_batch) break;
      if (traversing) {
        AMBER_COUNT_ROUNDS(2);
        traversing = BvhRound(sc, lds_stack, o, d, tr, hit, kStack AMBER_STAMP_ARG);
        if (!traversing && tr.overflow) ClosestHitLeafList(sc, o, d, hit);
      }
    }

    AMBER_CLK(6);
    rays_wave += static_cast<uint32_t>(__popcll(__ballot(alive && !traversing)));
    if (alive && !traversing) {                             // shade the lanes whose closest hit is known
      V3 meas = v3(my_sum[768], my_sum[1024], my_sum[1280]);  // the measurement is read and written once per bounce: LDS, not registers held across the traversal
      if (kLight) {
        const SplatSink sink{a.splats, a.splat_count, a.splat_capacity, pixel, s - 1u, sc.sensor.size_f};
        alive = PathShade<false, ENGINE_BVH, true>(sc, nullptr, hit, o, d, w, meas, rng, casts, origin_slot, nullptr AMBER_SHADE_STAMP_ARG, &sink);
      } else if (kSig) {
        BvhResolveIndex(sc, hit);                             // only the signature variant reports object indices
        Bounce b;
        alive = PathShade<true, ENGINE_BVH, false>(sc, nullptr, hit, o, d, w, meas, rng, casts, origin_slot, &b AMBER_SHADE_STAMP_ARG, nullptr);
        sig_obj = Fnv32(sig_obj, static_cast<uint32_t>(b.object));
        if (b.object >= 0) sig_t = Fnv32(sig_t, __float_as_uint(b.t));
        if (!alive) {                                         // path (band pixel, sample s - 1) of this launch
          const uint32_t plocal = slot % a.n_pixels;
          a.sig[static_cast<size_t>(plocal) * a.n_samples + (s - 1u - a.first_sample)] = static_cast<unsigned long long>(sig_obj) | (static_cast<unsigned long long>(sig_t) << 32);
        }
      } else {
        alive = PathShade<false, ENGINE_BVH, false>(sc, nullptr, hit, o, d, w, meas, rng, casts, origin_slot, nullptr AMBER_SHADE_STAMP_ARG, nullptr);
      }
      if (alive) { my_sum[768] = meas.x; my_sum[1024] = meas.y; my_sum[1280] = meas.z; }
      if (alive) { BvhBegin(sc, o, d, tr, hit); traversing = true; }
      else { my_sum[0] = my_sum[0] + meas.x; my_sum[256] = my_sum[256] + meas.y; my_sum[512] = my_sum[512] + meas.z; }   // sequential sum over the item's samples
    }
    AMBER_CLK(5);
  }

#undef AMBER_SHADE_STAMP_ARG
#ifdef AMBER_STAMPS
#ifdef AMBER_BVH_CLOCKS
  if (a.stamps) for (int k = 0; k < 8; k++) atomicAdd(a.stamps + k, stamp_ctx->acc[k] >> 6);      // lane-time / 64
#else
  if (a.stamps) for (int k = 0; k < 8; k++) if (stamp_ctx->acc[k]) atomicAdd(a.stamps + k, stamp_ctx->acc[k]);
#endif
#endif
  if (lane == 0 && rays_wave) atomicAdd(a.ray_count, static_cast<unsigned long long>(rays_wave));
}

// fb[e] += partial[0][e] + partial[1][e] + ... in chunk order (e = pixel*3 + channel of the band)
__global__ void reduce_partials_kernel(float* __restrict__ fb, const float* __restrict__ partial, uint32_t n_elems, uint32_t n_chunks) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elems) return;
  float v = fb[e];
  for (uint32_t c = 0; c < n_chunks; ++c) v = v + partial[static_cast<size_t>(c) * n_elems + e];
  fb[e] = v;
}

#include "bvh_pool.inc"
#include "bvh_stream.inc"
#include "wavefront.inc"

// ------------------------------------------------------------------------------------------------
// known-answer kernels (same device functions)
// ------------------------------------------------------------------------------------------------
template <int kEngine>
__global__ void kat_cast_kernel(const DevScene sc, uint32_t n, const float* org, const float* dir,
                                int32_t* out_obj, float* out_t, float* out_pos, float* out_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = i < n ? i : n - 1;      // keep the object loop wave-uniform for every lane
  constexpr bool kTwoPhase = kEngine == ENGINE_TWO_PHASE;
  __shared__ DevObject lds_objects[kTwoPhase ? AMBER_MAX_LDS_OBJECTS : 1];
  __shared__ int32_t lds_stack[kEngine == ENGINE_BVH ? AMBER_BVH_STACK * 256 : 1];
  if (kTwoPhase) StageObjects(sc, lds_objects);
  const V3 o = ld3(org + 3 * k), d = ld3(dir + 3 * k);
  HitRec h;
#ifdef AMBER_STAMPS
  StampCtx stamp_store{}; StampCtx* stamp_ctx = &stamp_store;
#endif
  ClosestHit<kEngine>(sc, lds_objects, lds_stack, o, d, -1, h AMBER_STAMP_ARG);
  if (i >= n) return;
  out_obj[i] = h.idx;
  if (h.idx < 0) {
    out_t[i] = __builtin_nanf("");
    for (int c = 0; c < 3; c++) { out_pos[3 * i + c] = 0.f; out_n[3 * i + c] = 0.f; }
    return;
  }
  V3 pos, nrm; uint32_t mat;
  ResolveHit(kEngine == ENGINE_TWO_PHASE ? lds_objects : (kEngine == ENGINE_BVH ? sc.bvh_objects : sc.objects), h, o, d, pos, nrm, mat);
  out_t[i] = h.t;
  out_pos[3 * i] = pos.x; out_pos[3 * i + 1] = pos.y; out_pos[3 * i + 2] = pos.z;
  out_n[3 * i] = nrm.x; out_n[3 * i + 1] = nrm.y; out_n[3 * i + 2] = nrm.z;
}

__global__ void kat_sample_kernel(const DevScene sc, uint32_t n, const uint32_t* material, const float* normals,
                                  const float* dirs_out, uint64_t* rng_state, float* out_dir, float* out_w) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const DevMaterial m = sc.materials[material[i]];
  uint64_t rng = rng_state[i];
  V3 di, w;
  SampleLight(m, ld3(normals + 3 * i), ld3(dirs_out + 3 * i), rng, di, w);
  rng_state[i] = rng;
  out_dir[3 * i] = di.x; out_dir[3 * i + 1] = di.y; out_dir[3 * i + 2] = di.z;
  out_w[3 * i] = w.x; out_w[3 * i + 1] = w.y; out_w[3 * i + 2] = w.z;
}

__global__ void kat_eye_kernel(const DevScene sc, uint64_t hashed_seed, uint32_t n, const uint32_t* pixel,
                               const uint32_t* sample, float* out7) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t rng = XorShiftSeed(hashed_seed, pixel[i], sample[i]);
  V3 o, d; float w; int origin_slot;
  GenerateEyeRay(sc, pixel[i] % sc.sensor.w, pixel[i] / sc.sensor.w, rng, o, d, w, origin_slot);
  float* p = out7 + 7 * i;
  p[0] = o.x; p[1] = o.y; p[2] = o.z; p[3] = d.x; p[4] = d.y; p[5] = d.z; p[6] = w;
}

template <int kEngine>
__global__ void kat_trace_kernel(const DevScene sc, uint64_t hashed_seed, uint32_t n, const uint32_t* pixel,
                                 const uint32_t* sample, uint32_t max_bounces, uint32_t* out_records, uint32_t* out_casts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = i < n ? i : n - 1;
  constexpr bool kTwoPhase = kEngine == ENGINE_TWO_PHASE;
  __shared__ DevObject lds_objects[kTwoPhase ? AMBER_MAX_LDS_OBJECTS : 1];
  __shared__ int32_t lds_stack[kEngine == ENGINE_BVH ? AMBER_BVH_STACK * 256 : 1];
  if (kTwoPhase) StageObjects(sc, lds_objects);
  uint64_t rng = XorShiftSeed(hashed_seed, pixel[k], sample[k]);
  V3 o, d; float ew; int origin_slot;
  GenerateEyeRay(sc, pixel[k] % sc.sensor.w, pixel[k] / sc.sensor.w, rng, o, d, ew, origin_slot);
  V3 w = v3(ew, ew, ew), meas = v3(0.f, 0.f, 0.f);
  uint32_t casts = 0;
  bool alive = true;
  // all lanes iterate together so that the scene loop stays wave-uniform; finished lanes idle
  while (__any(alive)) {
    if (alive) {
      Bounce b;
#ifdef AMBER_STAMPS
      StampCtx stamp_store{}; StampCtx* stamp_ctx = &stamp_store;
#endif
      alive = PathStep<true, kEngine>(sc, lds_objects, lds_stack, o, d, w, meas, rng, casts, origin_slot, &b AMBER_STAMP_ARG);
      if (i < n && casts <= max_bounces) {
        uint32_t* r = out_records + (static_cast<size_t>(i) * max_bounces + (casts - 1)) * 11u;
        r[0] = static_cast<uint32_t>(b.object);
        r[1] = __float_as_uint(b.t);
        r[2] = __float_as_uint(b.pos.x); r[3] = __float_as_uint(b.pos.y); r[4] = __float_as_uint(b.pos.z);
        r[5] = __float_as_uint(b.weight_before.x); r[6] = __float_as_uint(b.weight_before.y); r[7] = __float_as_uint(b.weight_before.z);
        r[8] = __float_as_uint(meas.x); r[9] = __float_as_uint(meas.y); r[10] = __float_as_uint(meas.z);
      }
    }
  }
  if (i < n) out_casts[i] = casts;
}

// Path signatures of the handle's band (amber_hip_kat_signatures): thread i traces the path of (band pixel i / n_samples,
// sample first_sample + i % n_samples) with the render kernels' device functions and writes FNV-1a-32 over the object index
// of every cast (low word; 0xffffffff = miss) and over the bits of every hit distance (high word).
template <int kEngine>
__global__ void kat_signature_kernel(const DevScene sc, uint64_t hashed_seed, uint64_t n, uint32_t first_sample, uint32_t n_samples,
                                     uint32_t row_begin, uint32_t stripe_rows, uint32_t stripe_period, unsigned long long* out) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t k = i < n ? i : n - 1;      // keep the scene loops wave-uniform for every lane
  constexpr bool kTwoPhase = kEngine == ENGINE_TWO_PHASE;
  __shared__ DevObject lds_objects[kTwoPhase ? AMBER_MAX_LDS_OBJECTS : 1];
  __shared__ int32_t lds_stack[kEngine == ENGINE_BVH ? AMBER_BVH_STACK * 256 : 1];
  if (kTwoPhase) StageObjects(sc, lds_objects);
  const uint32_t plocal = static_cast<uint32_t>(k / n_samples), smp = first_sample + static_cast<uint32_t>(k % n_samples);
  const uint32_t lrow = plocal / sc.sensor.w, px = plocal - lrow * sc.sensor.w;
  const uint32_t py = row_begin + (stripe_rows ? (lrow / stripe_rows) * stripe_period + lrow % stripe_rows : lrow);
  uint64_t rng = XorShiftSeed(hashed_seed, px + py * sc.sensor.w, smp);
  V3 o, d; float ew; int origin_slot;
  GenerateEyeRay(sc, px, py, rng, o, d, ew, origin_slot);
  V3 w = v3(ew, ew, ew), meas = v3(0.f, 0.f, 0.f);
  uint32_t casts = 0, sig_obj = 2166136261u, sig_t = 2166136261u;
  bool alive = true;
  while (__any(alive)) {
    if (alive) {
      Bounce b;
#ifdef AMBER_STAMPS
      StampCtx stamp_store{}; StampCtx* stamp_ctx = &stamp_store;
#endif
      alive = PathStep<true, kEngine>(sc, lds_objects, lds_stack, o, d, w, meas, rng, casts, origin_slot, &b AMBER_STAMP_ARG);
      sig_obj = Fnv32(sig_obj, static_cast<uint32_t>(b.object));
      if (b.object >= 0) sig_t = Fnv32(sig_t, __float_as_uint(b.t));
    }
  }
  if (i < n) out[i] = static_cast<unsigned long long>(sig_obj) | (static_cast<unsigned long long>(sig_t) << 32);
}

__global__ void kat_math_kernel(int mode, uint32_t n, const float* x, float* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) { float s, c; SinCos(x[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; }
  else if (mode == 1) out[i] = Pow(x[2 * i], x[2 * i + 1]);
  else {                                                    // 2: Pow4, 3: Pow5 -- the double's two words
    const double d = mode == 2 ? Pow4(x[i]) : Pow5(x[i]);
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(d));
    out[2 * i] = __uint_as_float(static_cast<uint32_t>(b)); out[2 * i + 1] = __uint_as_float(static_cast<uint32_t>(b >> 32));
  }
}

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
  DevBvhNodeQ* d_bvh_nodes = nullptr;
  DevBvhNodeQ4* d_bvh_nodes4 = nullptr;      // AMBER_BVH_WIDE builds only
  float4* d_bvh_spheres = nullptr;
  float4* d_bvh_tris = nullptr;
  uint32_t* d_bvh_prims = nullptr;
  DevObject* d_bvh_objects = nullptr;
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

int ValidateScene(const AmberFlatScene* s, const AmberSensor* sensor) {
  if (!s || !sensor) return Fail(AMBER_EINVAL, "null scene or sensor");
  if (!s->objects || s->n_objects == 0) return Fail(AMBER_EINVAL, "scene has no objects");
  if (s->n_objects >= (1u << 27)) return Fail(AMBER_EINVAL, "too many objects (BVH leaf references hold 27-bit offsets)");
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
  if (params->engine > AMBER_ENGINE_WAVEFRONT) return Fail(AMBER_EINVAL, "unknown engine");
  if (params->engine == AMBER_ENGINE_TWO_PHASE && s->n_objects > AMBER_MAX_LDS_OBJECTS)
    return Fail(AMBER_EINVAL, "AMBER_ENGINE_TWO_PHASE supports at most 32 objects");

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

  const uint32_t auto_hit = s->n_objects <= AMBER_MAX_LDS_OBJECTS ? AMBER_ENGINE_TWO_PHASE : AMBER_ENGINE_BVH;
  h->engine = params->engine != AMBER_ENGINE_AUTO ? params->engine : auto_hit;
  h->hit_engine = h->engine == AMBER_ENGINE_WAVEFRONT ? auto_hit : h->engine;
  h->two_phase = h->hit_engine == AMBER_ENGINE_TWO_PHASE;
  // engine BVH has two schedulers with identical results: the default is the faster one on the 1M-sphere
  // scene (pt_bvh_megakernel, 74 ms at 64 spp against 79); the environment overrides the flag either way (A/B tools)
  h->bvh_pool = (params->reserved & AMBER_PT_FLAG_BVH_POOL) != 0u;
  { const char* ev = std::getenv("AMBER_BVH_POOL"); if (ev && (ev[0] == '0' || ev[0] == '1')) h->bvh_pool = ev[0] == '1'; }
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
    // The shading batch of pt_bvh_megakernel.  While a wave collects finished lanes they idle through the rounds of the others, and a round
    // over triangle leaves costs about twice a round over sphere leaves (45 against 20 vector instructions per leaf object before any
    // root / quotient), so idle lanes are dearer in a mesh: tools/shade_batch_sweep.py (profiles/r05_shade_batch_sweep.txt) -- 1M spheres
    // best at 52 (49.7 ms at 64 spp; 40: 52.3), 1M-triangle terrain at 32 (62.1; 40: 63.8; 52: 68.5), 82k-triangle room at 36-44 (32.8; 52: 33.9).
    {
      bool any_triangle = false;
      for (const DevObject& ob : objs) any_triangle = any_triangle || (ob.kind & 0xffu) == AMBER_PRIM_TRIANGLE;
      h->bvh_shade_batch = any_triangle ? 40u : static_cast<uint32_t>(AMBER_BVH_SHADE_BATCH);
    }
    { const char* ev = std::getenv("AMBER_BVH_SHADE_BATCH"); if (ev && std::atoi(ev) >= 1 && std::atoi(ev) <= 64) h->bvh_shade_batch = static_cast<uint32_t>(std::atoi(ev)); }   // measurement hook
    { const char* ev = std::getenv("AMBER_BVH_PATHS"); if (ev && ev[0] == '0') h->bvh_paths = false; }
    { const char* ev = std::getenv("AMBER_BVH_PATHS_MAX_DEPTH"); if (ev && static_cast<uint32_t>(std::atoi(ev)) < bvh.depth) h->bvh_paths = false; }   // measurement hook
    if (std::getenv("AMBER_DEBUG_BVH")) std::fprintf(stderr, "amber_hip: BVH of %u objects: %zu nodes, depth %u\n", s->n_objects, bvh.nodes.size(), bvh.depth);
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
  if (h->two_phase) amber_filter::BuildFilterProgram(objs, fp_center, fprog);
  if (h->two_phase && std::getenv("AMBER_DEBUG_FILTER")) {     // diagnostic: shape of the Phase-A program
    uint32_t pairs = 0, singles = 0;
    uint32_t shared = 0;
    for (const DevPlane& pl : fprog.planes) { pairs += pl.n_pairs; singles += pl.n_tris & 0x7fffffffu; shared += pl.n_tris >> 31; }
    std::fprintf(stderr, "amber_hip: filter program: %zu planes (%u share the previous plane's normal), %u pair records, %u single records, %zu spheres, always mask %#x\n",
                 fprog.planes.size(), shared, pairs, singles, fprog.spheres.size(), fprog.always_mask);
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
  }

  auto cleanup = [&](int code, const std::string& msg) { amber_hip_pt_destroy(h); return Fail(code, msg); };
#define HIP_TRY_H(expr)                                                                            \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(AMBER_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

  HIP_TRY_H(hipMalloc(&h->d_objects, objs.size() * sizeof(DevObject)));
  HIP_TRY_H(hipMalloc(&h->d_materials, mats.size() * sizeof(DevMaterial)));
  HIP_TRY_H(hipMalloc(&h->d_blades, blades.size() * sizeof(DevBlade)));
  HIP_TRY_H(hipMalloc(&h->d_planes, (fprog.planes.size() + 1) * sizeof(DevPlane)));
  HIP_TRY_H(hipMalloc(&h->d_tri_filters, (fprog.tris.size() + 1) * sizeof(DevTriFilter)));
  HIP_TRY_H(hipMalloc(&h->d_sphere_filters, (fprog.spheres.size() + 1) * sizeof(DevSphereFilter)));
  if (!fprog.planes.empty()) HIP_TRY_H(hipMemcpy(h->d_planes, fprog.planes.data(), fprog.planes.size() * sizeof(DevPlane), hipMemcpyHostToDevice));
  if (!fprog.tris.empty()) HIP_TRY_H(hipMemcpy(h->d_tri_filters, fprog.tris.data(), fprog.tris.size() * sizeof(DevTriFilter), hipMemcpyHostToDevice));
  {
    std::vector<DevObject> prog(fprog.order.size());
    for (size_t k = 0; k < prog.size(); k++) { prog[k] = objs[fprog.order[k]]; prog[k].kind |= fprog.order[k] << 8; }
    HIP_TRY_H(hipMalloc(&h->d_prog_objects, (prog.size() + 1) * sizeof(DevObject)));
    if (!prog.empty()) HIP_TRY_H(hipMemcpy(h->d_prog_objects, prog.data(), prog.size() * sizeof(DevObject), hipMemcpyHostToDevice));
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
  if (!fprog.spheres.empty()) HIP_TRY_H(hipMemcpy(h->d_sphere_filters, fprog.spheres.data(), fprog.spheres.size() * sizeof(DevSphereFilter), hipMemcpyHostToDevice));
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

  DevScene& sc = h->scene;
  sc.objects = h->d_objects; sc.materials = h->d_materials; sc.blades = h->d_blades;
  sc.planes = h->d_planes; sc.tri_filters = h->d_tri_filters; sc.sphere_filters = h->d_sphere_filters;
  sc.n_planes = static_cast<uint32_t>(fprog.planes.size()); sc.n_simple_planes = fprog.n_simple_planes; sc.n_sphere_filters = static_cast<uint32_t>(fprog.spheres.size());
  sc.bvh_nodes = h->d_bvh_nodes; sc.bvh_nodes4 = h->d_bvh_nodes4; sc.bvh_prims = h->d_bvh_prims; sc.bvh_objects = h->d_bvh_objects; sc.bvh_spheres = h->d_bvh_spheres; sc.bvh_tris = h->d_bvh_tris; sc.bvh_root = qbvh.root_ref;
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

// Engine WAVEFRONT host loop: batches of <= max_chunks accumulation chunks; per batch generate, then bounce launches
// until the live-ray count read back from the device is zero, then the ordered reduction into the framebuffer.
int RenderPassWavefront(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples) {
  const uint32_t n_pixels = h->local_rows * h->scene.sensor.w;
  const uint64_t kMaxPaths = 1ull << 28;                       // 268 M paths: 2 x 14 GB of queues + 3.2 GB of measurements
  uint64_t max_chunks = kMaxPaths / (static_cast<uint64_t>(n_pixels) * AMBER_ACCUM_CHUNK);
  if (max_chunks == 0) return Fail(AMBER_EINVAL, "band too large for the wavefront engine");
  const uint32_t kMaxBounces = 4096;
  uint32_t done = 0;
  while (done < n_samples) {
    uint32_t n = n_samples - done;
    if (n > max_chunks * AMBER_ACCUM_CHUNK) n = static_cast<uint32_t>(max_chunks * AMBER_ACCUM_CHUNK);
    const uint32_t n_chunks = (n + AMBER_ACCUM_CHUNK - 1) / AMBER_ACCUM_CHUNK;
    const size_t n_paths = static_cast<size_t>(n_chunks) * n_pixels * AMBER_ACCUM_CHUNK;
    // every queue array holds AMBER_WF_SHARDS shards of shard_capacity rays
    const size_t shard_capacity = ((n_paths + AMBER_WF_SHARDS - 1) / AMBER_WF_SHARDS + 63) / 64 * 64 + 64;
    const size_t q_len = shard_capacity * AMBER_WF_SHARDS;
    const size_t n_counts = static_cast<size_t>(kMaxBounces + 2) * AMBER_WF_SHARDS;
    const size_t bytes = (26 * q_len + 3 * n_paths) * 4 + n_counts * sizeof(unsigned int);
    if (bytes > h->wf_bytes) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->d_wf) { HIP_TRY(hipFree(h->d_wf)); h->d_wf = nullptr; h->wf_bytes = 0; }
      hipError_t e = hipMalloc(&h->d_wf, bytes);
      if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(wavefront queues): ") + hipGetErrorString(e));
      h->wf_bytes = bytes;
    }
    float* base = h->d_wf;
    WfQueue q[2];
    for (int k = 0; k < 2; k++) {
      for (int c = 0; c < 9; c++) q[k].f[c] = base + (static_cast<size_t>(k) * 13 + c) * q_len;
      for (int c = 0; c < 4; c++) q[k].u[c] = reinterpret_cast<uint32_t*>(base + (static_cast<size_t>(k) * 13 + 9 + c) * q_len);
    }
    float* meas = base + 26 * q_len;
    unsigned int* counts = reinterpret_cast<unsigned int*>(meas + 3 * n_paths);
    HIP_TRY(hipMemsetAsync(counts, 0, n_counts * sizeof(unsigned int), h->stream));
    WfArgs a;
    a.scene = h->scene; a.meas = meas; a.counts = counts; a.ray_count = h->d_rays; a.hashed_seed = h->hashed_seed;
    a.row_begin = h->row_begin; a.stripe_rows = h->stripe_rows; a.stripe_period = h->stripe_period; a.n_pixels = n_pixels;
    a.first_sample = first_sample + done; a.n_samples = n; a.n_paths = static_cast<uint32_t>(n_paths); a.bounce = 0; a.shard_capacity = static_cast<uint32_t>(shard_capacity);
    const uint32_t n_blocks = 2048u;                          // 8192 waves = 32 per shard
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    { const int rc = AcquireEventPair(h, &evp); if (rc != AMBER_OK) return rc; }
    auto& ev = *evp;
    HIP_TRY(hipEventRecord(ev.first, h->stream));
    a.out = q[0]; a.in = q[1];
    hipLaunchKernelGGL(wf_generate_kernel, dim3(n_blocks), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    uint32_t bounce = 0;
    for (;;) {
      const uint32_t burst = bounce < 16 ? 8 : 16;            // launches enqueued before the live count is read back
      for (uint32_t k = 0; k < burst && bounce < kMaxBounces; k++, bounce++) {
        a.bounce = bounce; a.in = q[bounce & 1]; a.out = q[(bounce + 1) & 1];
        if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL(wf_bounce_kernel<ENGINE_TWO_PHASE>, dim3(n_blocks), dim3(256), 0, h->stream, a);
        else if (h->hit_engine == AMBER_ENGINE_BVH) hipLaunchKernelGGL(wf_bounce_kernel<ENGINE_BVH>, dim3(n_blocks), dim3(256), 0, h->stream, a);
        else hipLaunchKernelGGL(wf_bounce_kernel<ENGINE_LIST>, dim3(n_blocks), dim3(256), 0, h->stream, a);
        HIP_TRY(hipGetLastError());
      }
      unsigned int shard_live[AMBER_WF_SHARDS];
      HIP_TRY(hipMemcpyAsync(shard_live, counts + static_cast<size_t>(bounce) * AMBER_WF_SHARDS, sizeof shard_live, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      unsigned long long live = 0;
      for (unsigned int v : shard_live) live += v;
      if (live == 0) break;
      if (bounce >= kMaxBounces) return Fail(AMBER_EHIP, "wavefront engine: path longer than 4096 bounces");
    }
    HIP_TRY(hipEventRecord(ev.second, h->stream));
    const uint32_t n_elems = n_pixels * 3u;
    hipLaunchKernelGGL(wf_reduce_kernel, dim3((n_elems + 255u) / 256u), dim3(256), 0, h->stream, h->d_fb, meas, n_pixels, n_chunks);
    HIP_TRY(hipGetLastError());
    done += n;
  }
  return AMBER_OK;
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
  const bool bvh = h->hit_engine == AMBER_ENGINE_BVH && !h->bvh_paths;
  uint32_t n_blocks = static_cast<uint32_t>(h->n_cus) * (bvh ? static_cast<uint32_t>(AMBER_BVH_POOL_WGS) : ResidentBlocksPerCu(h->hit_engine, h->bvh_depth));
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
  const bool bvh = h->hit_engine == AMBER_ENGINE_BVH && !h->bvh_paths;       // pt_bvh_pool_kernel (bvh_paths: pt_megakernel<ENGINE_BVH>)
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
    const size_t carried = bvh ? static_cast<size_t>(h->n_cus) * AMBER_BVH_POOL_WGS * 4u * AMBER_BVH_POOL_CARRIED_PER_WAVE
                               : static_cast<size_t>(h->n_cus) * ResidentBlocksPerCu(h->hit_engine, h->bvh_depth) * 256u * 3u * 2u;   // own slots + parked slots
    if (carried > h->carried_floats) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->d_carried) { HIP_TRY(hipFree(h->d_carried)); h->d_carried = nullptr; h->carried_floats = 0; }
      hipError_t e = hipMalloc(&h->d_carried, carried * sizeof(float));
      if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(carried measurements): ") + hipGetErrorString(e));
      h->carried_floats = carried;
    }
  }
  if (bvh) {
    const size_t stack_ints = static_cast<size_t>(h->n_cus) * AMBER_BVH_POOL_WGS * 256u * AMBER_BVH_POOL_GLOBAL_LEVELS;
    if (stack_ints > h->bvh_stack_ints) {
      if (h->d_bvh_stack) { HIP_TRY(hipFree(h->d_bvh_stack)); h->d_bvh_stack = nullptr; h->bvh_stack_ints = 0; }
      hipError_t e = hipMalloc(&h->d_bvh_stack, stack_ints * sizeof(int32_t));
      if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(traversal stacks): ") + hipGetErrorString(e));
      h->bvh_stack_ints = stack_ints;
    }
  }
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
  if (sig) {
    if (bvh) hipLaunchKernelGGL((pt_bvh_pool_kernel<true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->bvh_paths) hipLaunchKernelGGL((pt_megakernel<ENGINE_BVH, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((pt_megakernel<ENGINE_LIST, false, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
  } else {
    if (bvh) hipLaunchKernelGGL((pt_bvh_pool_kernel<false>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->bvh_paths) hipLaunchKernelGGL((pt_megakernel<ENGINE_BVH>), dim3(n_blocks), dim3(256), 0, h->stream, a);
    else if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL((pt_megakernel<ENGINE_TWO_PHASE>), dim3(n_blocks), dim3(256), 0, h->stream, a);
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

int amber_hip_pt_render_pass(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  if (n_samples == 0) return AMBER_OK;
  if (static_cast<uint64_t>(first_sample) + n_samples > 0xffffffffull) return Fail(AMBER_EINVAL, "sample index overflow");
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t n_pixels = h->local_rows * h->scene.sensor.w;
  if (n_pixels == 0) return AMBER_OK;                    // empty band
  if (h->engine == AMBER_ENGINE_WAVEFRONT) return RenderPassWavefront(h, first_sample, n_samples);
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
    if (sig) {
      if (h->bvh_depth <= 24) hipLaunchKernelGGL((pt_bvh_megakernel<false, 24, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      else hipLaunchKernelGGL((pt_bvh_megakernel<false, AMBER_BVH_STACK, true>), dim3(n_blocks), dim3(256), 0, h->stream, a);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(ev.second, h->stream));
      return AMBER_OK;
    }
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
  if (h->d_bvh_nodes) (void)hipFree(h->d_bvh_nodes);
  if (h->d_bvh_nodes4) (void)hipFree(h->d_bvh_nodes4);
  if (h->d_bvh_spheres) (void)hipFree(h->d_bvh_spheres);
  if (h->d_bvh_tris) (void)hipFree(h->d_bvh_tris);
  if (h->d_bvh_prims) (void)hipFree(h->d_bvh_prims);
  if (h->d_bvh_objects) (void)hipFree(h->d_bvh_objects);
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

// ---- KAT entry points -----------------------------------------------------------------------------
int amber_hip_kat_cast(amber_hip_pt* h, uint32_t n, const float* origins, const float* dirs,
                       int32_t* out_object, float* out_t, float* out_pos, float* out_normal) {
  if (!h || !origins || !dirs || !out_object || !out_t || !out_pos || !out_normal) return Fail(AMBER_EINVAL, "null argument");
  if (n == 0) return AMBER_OK;
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<float> d_o, d_d, d_t, d_p, d_n; DevBuf<int32_t> d_i;
  HIP_TRY(d_o.alloc(3 * n)); HIP_TRY(d_d.alloc(3 * n)); HIP_TRY(d_t.alloc(n)); HIP_TRY(d_p.alloc(3 * n)); HIP_TRY(d_n.alloc(3 * n)); HIP_TRY(d_i.alloc(n));
  HIP_TRY(hipMemcpy(d_o.p, origins, 3ull * n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_d.p, dirs, 3ull * n * 4, hipMemcpyHostToDevice));
  if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL(kat_cast_kernel<ENGINE_TWO_PHASE>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, n, d_o.p, d_d.p, d_i.p, d_t.p, d_p.p, d_n.p);
  else if (h->hit_engine == AMBER_ENGINE_BVH) hipLaunchKernelGGL(kat_cast_kernel<ENGINE_BVH>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, n, d_o.p, d_d.p, d_i.p, d_t.p, d_p.p, d_n.p);
  else hipLaunchKernelGGL(kat_cast_kernel<ENGINE_LIST>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, n, d_o.p, d_d.p, d_i.p, d_t.p, d_p.p, d_n.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_object, d_i.p, 4ull * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_t, d_t.p, 4ull * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_pos, d_p.p, 12ull * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_normal, d_n.p, 12ull * n, hipMemcpyDeviceToHost));
  return AMBER_OK;
}

int amber_hip_kat_sample(amber_hip_pt* h, uint32_t n, const uint32_t* material, const float* normals,
                         const float* dirs_out, uint64_t* rng_state, float* out_dir_in, float* out_weight) {
  if (!h || !material || !normals || !dirs_out || !rng_state || !out_dir_in || !out_weight) return Fail(AMBER_EINVAL, "null argument");
  if (n == 0) return AMBER_OK;
  for (uint32_t i = 0; i < n; i++)
    if (material[i] >= h->n_materials) return Fail(AMBER_EINVAL, "material index out of range");
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<uint32_t> d_m; DevBuf<float> d_n, d_d, d_o, d_w; DevBuf<uint64_t> d_r;
  HIP_TRY(d_m.alloc(n)); HIP_TRY(d_n.alloc(3 * n)); HIP_TRY(d_d.alloc(3 * n)); HIP_TRY(d_o.alloc(3 * n)); HIP_TRY(d_w.alloc(3 * n)); HIP_TRY(d_r.alloc(n));
  HIP_TRY(hipMemcpy(d_m.p, material, 4ull * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_n.p, normals, 12ull * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_d.p, dirs_out, 12ull * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_r.p, rng_state, 8ull * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kat_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, n, d_m.p, d_n.p, d_d.p, d_r.p, d_o.p, d_w.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(rng_state, d_r.p, 8ull * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_dir_in, d_o.p, 12ull * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_weight, d_w.p, 12ull * n, hipMemcpyDeviceToHost));
  return AMBER_OK;
}

int amber_hip_kat_eye(amber_hip_pt* h, uint32_t n, const uint32_t* pixel, const uint32_t* sample, float* out7) {
  if (!h || !pixel || !sample || !out7) return Fail(AMBER_EINVAL, "null argument");
  if (n == 0) return AMBER_OK;
  const uint32_t npx = h->scene.sensor.w * h->scene.sensor.h;
  for (uint32_t i = 0; i < n; i++) if (pixel[i] >= npx) return Fail(AMBER_EINVAL, "pixel index out of range");
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<uint32_t> d_p, d_s; DevBuf<float> d_o;
  HIP_TRY(d_p.alloc(n)); HIP_TRY(d_s.alloc(n)); HIP_TRY(d_o.alloc(7 * n));
  HIP_TRY(hipMemcpy(d_p.p, pixel, 4ull * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_s.p, sample, 4ull * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kat_eye_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, d_p.p, d_s.p, d_o.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out7, d_o.p, 28ull * n, hipMemcpyDeviceToHost));
  return AMBER_OK;
}

int amber_hip_kat_trace(amber_hip_pt* h, uint32_t n, const uint32_t* pixel, const uint32_t* sample,
                        uint32_t max_bounces, uint32_t* out_records, uint32_t* out_casts) {
  if (!h || !pixel || !sample || !out_records || !out_casts || max_bounces == 0) return Fail(AMBER_EINVAL, "bad argument");
  if (n == 0) return AMBER_OK;
  const uint32_t npx = h->scene.sensor.w * h->scene.sensor.h;
  for (uint32_t i = 0; i < n; i++) if (pixel[i] >= npx) return Fail(AMBER_EINVAL, "pixel index out of range");
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<uint32_t> d_p, d_s, d_r, d_c;
  const size_t nrec = static_cast<size_t>(n) * max_bounces * 11;
  HIP_TRY(d_p.alloc(n)); HIP_TRY(d_s.alloc(n)); HIP_TRY(d_r.alloc(nrec)); HIP_TRY(d_c.alloc(n));
  HIP_TRY(hipMemcpy(d_p.p, pixel, 4ull * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_s.p, sample, 4ull * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemsetAsync(d_r.p, 0, nrec * 4, h->stream));
  if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL(kat_trace_kernel<ENGINE_TWO_PHASE>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, d_p.p, d_s.p, max_bounces, d_r.p, d_c.p);
  else if (h->hit_engine == AMBER_ENGINE_BVH) hipLaunchKernelGGL(kat_trace_kernel<ENGINE_BVH>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, d_p.p, d_s.p, max_bounces, d_r.p, d_c.p);
  else hipLaunchKernelGGL(kat_trace_kernel<ENGINE_LIST>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, d_p.p, d_s.p, max_bounces, d_r.p, d_c.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_records, d_r.p, nrec * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_casts, d_c.p, 4ull * n, hipMemcpyDeviceToHost));
  return AMBER_OK;
}

int amber_hip_kat_signatures(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, uint64_t* out) {
  if (!h || !out || n_samples == 0) return Fail(AMBER_EINVAL, "bad argument");
  if (static_cast<uint64_t>(first_sample) + n_samples > 0xffffffffull) return Fail(AMBER_EINVAL, "sample index overflow");
  const uint64_t n = static_cast<uint64_t>(h->local_rows) * h->scene.sensor.w * n_samples;
  if (n == 0) return AMBER_OK;
  if ((n + 255) / 256 > 0x7fffffffull) return Fail(AMBER_EINVAL, "too many paths for one call");
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<unsigned long long> d_out;
  HIP_TRY(d_out.alloc(n));
  const dim3 grid(static_cast<uint32_t>((n + 255) / 256));
  if (h->hit_engine == AMBER_ENGINE_TWO_PHASE) hipLaunchKernelGGL(kat_signature_kernel<ENGINE_TWO_PHASE>, grid, dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, first_sample, n_samples, h->row_begin, h->stripe_rows, h->stripe_period, d_out.p);
  else if (h->hit_engine == AMBER_ENGINE_BVH) hipLaunchKernelGGL(kat_signature_kernel<ENGINE_BVH>, grid, dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, first_sample, n_samples, h->row_begin, h->stripe_rows, h->stripe_period, d_out.p);
  else hipLaunchKernelGGL(kat_signature_kernel<ENGINE_LIST>, grid, dim3(256), 0, h->stream, h->scene, h->hashed_seed, n, first_sample, n_samples, h->row_begin, h->stripe_rows, h->stripe_period, d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return AMBER_OK;
}

// Traversal alone (bvh_stream.inc): closest hits of n rays through engine BVH's resumable traversal in a kernel that does nothing
// else, `waves` resident waves per SIMD (4, 5, 6 or 8), idle lanes refilled once `refill_min` of a wave's lanes are idle.
// Returns t (NaN = miss) and the object index per ray, and the kernel time of `repeats` launches (the best one).
int amber_hip_kat_traversal_rate(amber_hip_pt* h, uint32_t n, const float* origins, const float* dirs, uint32_t waves, uint32_t refill_min, uint32_t repeats,
                                 float* out_t, int32_t* out_object, double* best_ms, uint32_t* out_rounds) {
  // one launch walks the ray array `repeats` times (a launch of n * repeats rays); best_ms is the time of that launch
  if (!h || !origins || !dirs || !out_t || !out_object || n == 0) return Fail(AMBER_EINVAL, "bad argument");
  if (h->hit_engine != AMBER_ENGINE_BVH) return Fail(AMBER_EINVAL, "the handle's engine is not BVH");
  if (refill_min == 0 || refill_min > 64) return Fail(AMBER_EINVAL, "refill_min must be in [1, 64]");
  const uint64_t n_virtual = static_cast<uint64_t>(n) * (repeats ? repeats : 1u);
  if (n_virtual > 0xfffffeffull) return Fail(AMBER_EINVAL, "n * repeats must stay below 2^32");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  std::vector<float4> packed(2ull * n);
  for (uint32_t i = 0; i < n; i++) {
    packed[2ull * i] = make_float4(origins[3ull * i], origins[3ull * i + 1], origins[3ull * i + 2], 0.f);
    packed[2ull * i + 1] = make_float4(dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2], 0.f);
  }
  DevBuf<float4> d_rays; DevBuf<float2> d_out; DevBuf<unsigned int> d_next; DevBuf<int32_t> d_stack; DevBuf<uint32_t> d_rounds;
  if (out_rounds) HIP_TRY(d_rounds.alloc(n));
  HIP_TRY(d_rays.alloc(2ull * n)); HIP_TRY(d_out.alloc(n)); HIP_TRY(d_next.alloc(1));
  const uint32_t n_blocks = static_cast<uint32_t>(h->n_cus) * (waves >= 4 && waves <= 8 ? waves : 5u);
  HIP_TRY(d_stack.alloc(static_cast<size_t>(n_blocks) * 256u * AMBER_BVH_STACK));
  HIP_TRY(hipMemcpy(d_rays.p, packed.data(), packed.size() * sizeof(float4), hipMemcpyHostToDevice));
  struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evs;   // released on every exit path
  HIP_TRY(hipEventCreate(&evs.a)); HIP_TRY(hipEventCreate(&evs.b));
  const hipEvent_t e0 = evs.a, e1 = evs.b;
  double best = 1e300;
  const uint32_t nv = static_cast<uint32_t>(n_virtual);
  for (uint32_t rep = 0; rep < 2u; rep++) {                              // a warm-up launch and the measured one
    HIP_TRY(hipMemsetAsync(d_next.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(hipEventRecord(e0, h->stream));
    switch (waves) {
      case 4: hipLaunchKernelGGL((bvh_trace_rate_kernel<4, 24>), dim3(n_blocks), dim3(256), 0, h->stream, h->scene, nv, d_rays.p, d_out.p, d_next.p, d_stack.p, refill_min, n, d_rounds.p); break;
      case 6: hipLaunchKernelGGL((bvh_trace_rate_kernel<6, 24>), dim3(n_blocks), dim3(256), 0, h->stream, h->scene, nv, d_rays.p, d_out.p, d_next.p, d_stack.p, refill_min, n, d_rounds.p); break;
      case 8: hipLaunchKernelGGL((bvh_trace_rate_kernel<8, 16>), dim3(n_blocks), dim3(256), 0, h->stream, h->scene, nv, d_rays.p, d_out.p, d_next.p, d_stack.p, refill_min, n, d_rounds.p); break;
      default: hipLaunchKernelGGL((bvh_trace_rate_kernel<5, 24>), dim3(n_blocks), dim3(256), 0, h->stream, h->scene, nv, d_rays.p, d_out.p, d_next.p, d_stack.p, refill_min, n, d_rounds.p); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1u || ms < best) best = ms;
  }
  std::vector<float2> res(n);
  HIP_TRY(hipMemcpy(res.data(), d_out.p, n * sizeof(float2), hipMemcpyDeviceToHost));
  if (out_rounds) HIP_TRY(hipMemcpy(out_rounds, d_rounds.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<uint32_t> prims(h->scene.n_objects);
  HIP_TRY(hipMemcpy(prims.data(), h->d_bvh_prims, prims.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; i++) {
    int32_t slot; std::memcpy(&slot, &res[i].y, 4);
    out_object[i] = slot < 0 ? -1 : static_cast<int32_t>(prims[slot]);
    out_t[i] = slot < 0 ? std::nanf("") : res[i].x;
  }
  if (best_ms) *best_ms = best;
  return AMBER_OK;
}

int amber_hip_kat_pixel_masks(amber_hip_pt* h, uint32_t* out_mask, uint32_t* out_slot_of_object, uint32_t* out_always_mask, double* kernel_ms) {
  if (!h) return Fail(AMBER_EINVAL, "null handle");
  if (!h->two_phase) return Fail(AMBER_EINVAL, "pixel masks belong to the two-phase engine");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  const uint32_t n_pixels = h->local_rows * h->scene.sensor.w;
  if (out_mask || kernel_ms) {                                 // the slot table and the always-mask below do not need the kernel
    float ms = 0;
    { const int rc = StartPixelMasks(h, kernel_ms ? &ms : nullptr); if (rc != AMBER_OK) return rc; }
    if (!h->pixel_mask_ready) return Fail(AMBER_EINVAL, "pixel masks are switched off (AMBER_PIXEL_MASK=0) or the band is empty");
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (out_mask) HIP_TRY(hipMemcpy(out_mask, h->d_pixel_mask, static_cast<size_t>(n_pixels) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (kernel_ms) *kernel_ms = static_cast<double>(ms);
  }
  if (out_slot_of_object) {
    for (uint32_t i = 0; i < h->scene.n_objects; i++) out_slot_of_object[i] = 0xffffffffu;
    for (size_t k = 0; k < h->prog_order.size(); k++) if (h->prog_order[k] < h->scene.n_objects) out_slot_of_object[h->prog_order[k]] = static_cast<uint32_t>(k);
  }
  if (out_always_mask) *out_always_mask = h->scene.always_mask | h->scene.blade_mask;
  return AMBER_OK;
}

int amber_hip_pt_signatures(amber_hip_pt* h, uint32_t first_sample, uint32_t n_samples, uint64_t* out) {
  if (!h || !out || n_samples == 0) return Fail(AMBER_EINVAL, "bad argument");
  if (static_cast<uint64_t>(first_sample) + n_samples > 0xffffffffull) return Fail(AMBER_EINVAL, "sample index overflow");
  if (h->engine == AMBER_ENGINE_WAVEFRONT) return Fail(AMBER_EINVAL, "signatures come from the work-queue kernels (engines list, two_phase, bvh)");
  const bool bvh_items = h->hit_engine == AMBER_ENGINE_BVH && !h->bvh_pool && !h->bvh_paths;        // pt_bvh_megakernel's signature instantiation
  const uint32_t n_pixels = h->local_rows * h->scene.sensor.w;
  const uint64_t n = static_cast<uint64_t>(n_pixels) * n_samples;
  if (n == 0) return AMBER_OK;
  if (n > kMaxPathsPerLaunch) return Fail(AMBER_EINVAL, "too many paths for one call");
  HIP_TRY(hipSetDevice(h->device));
  { const int rc = ResolvePending(h); if (rc != AMBER_OK) return rc; }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (n > h->sig_paths) {
    if (h->d_sig) { HIP_TRY(hipFree(h->d_sig)); h->d_sig = nullptr; h->sig_paths = 0; }
    hipError_t e = hipMalloc(&h->d_sig, n * sizeof(unsigned long long));
    if (e != hipSuccess) return Fail(AMBER_ENOMEM, std::string("hipMalloc(signatures): ") + hipGetErrorString(e));
    h->sig_paths = n;
  }
  HIP_TRY(hipMemsetAsync(h->d_sig, 0, n * sizeof(unsigned long long), h->stream));
  if (bvh_items) {
    const int rc = RenderPassBvhItems(h, first_sample, n_samples, n_pixels, h->d_sig); if (rc != AMBER_OK) return rc;
  } else {
    { const int rc = EnsureRecordCapacity(h, RecordSlack(h)); if (rc != AMBER_OK) return rc; }   // records of this launch are discarded
    const int rc = LaunchPaths(h, first_sample, n_samples, n_pixels, h->d_sig); if (rc != AMBER_OK) return rc;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out, h->d_sig, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return AMBER_OK;
}

int amber_hip_kat_math(int device, int mode, uint32_t n, const float* x, float* out) {
  if (!x || !out || mode < 0 || mode > 3) return Fail(AMBER_EINVAL, "bad argument");
  if (n == 0) return AMBER_OK;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return Fail(AMBER_ENODEVICE, "no such HIP device");
  HIP_TRY(hipSetDevice(device));
  const size_t n_in = mode == 1 ? 2ull * n : n, n_out = mode == 1 ? n : 2ull * n;
  DevBuf<float> d_x, d_o;
  HIP_TRY(d_x.alloc(n_in)); HIP_TRY(d_o.alloc(n_out));
  HIP_TRY(hipMemcpy(d_x.p, x, n_in * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kat_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, mode, n, d_x.p, d_o.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d_o.p, n_out * 4, hipMemcpyDeviceToHost));
  return AMBER_OK;
}

}  // extern "C"
