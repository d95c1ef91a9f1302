// pt_args.h -- kernel arguments of the render kernels and the record-emission helpers they share.
// Part of the one translation unit pt_host.hip (kernels are launched from there); see its header comment.
#pragma once

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

