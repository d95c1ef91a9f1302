/*
 * amber_hip_lab.h -- entry points of the LAB build (libamber_hip_lab.so = the product's sources compiled with -DAMBER_LAB).
 *
 * Not part of the product: libamber_hip.so exports none of these.  The lab library contains everything the product does -- the same kernels
 * from the same files -- plus (1) the known-answer kernels through which tests/ compare single stages with the oracle, (2) the signature
 * instantiations of the product kernels (per-path hashes of hit objects and distances), (3) the schedulers that were built, measured slower
 * and kept provably equal: engine WAVEFRONT (wavefront.inc: SoA ray queues in HBM, one launch per bounce, ballot / prefix-sum compaction --
 * the north star's formulation), AMBER_PT_FLAG_BVH_POOL (bvh_pool.inc) and the traversal-only kernel (bvh_stream.inc).
 */
#ifndef AMBER_HIP_LAB_H
#define AMBER_HIP_LAB_H

#include "amber_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the libraries are built with -fvisibility=hidden: what these headers declare is what they export */

/* ---- known-answer entry points (same device functions as the render kernels) --------------
 * Used by tests/ to compare individual stages against the oracle.  All buffers are HOST
 * pointers; n items; synchronous. */
/* closest hit: out_object = object index or -1 */
int amber_hip_kat_cast(amber_hip_pt*, uint32_t n, const float* origins /*n*3*/, const float* dirs /*n*3*/,
                       int32_t* out_object, float* out_t, float* out_pos /*n*3*/, float* out_normal /*n*3*/);
/* material sampling with a per-item XorShift state; returns dir_in, weight and the advanced state */
int amber_hip_kat_sample(amber_hip_pt*, uint32_t n, const uint32_t* material /*n*/, const float* normals,
                         const float* dirs_out, uint64_t* rng_state /*n, in/out*/, float* out_dir_in, float* out_weight);
/* eye rays for (pixel index, sample) pairs: out = origin[3] dir[3] weight */
int amber_hip_kat_eye(amber_hip_pt*, uint32_t n, const uint32_t* pixel, const uint32_t* sample, float* out7);
/* full per-path trace: for item i writes up to max_bounces records of
 * {object(int32 as float bits), t, pos[3], weight[3], measurement[3]} (11 x 4 bytes) and the cast count */
int amber_hip_kat_trace(amber_hip_pt*, uint32_t n, const uint32_t* pixel, const uint32_t* sample,
                        uint32_t max_bounces, uint32_t* out_records /*n*max_bounces*11*/, uint32_t* out_casts /*n*/);
/* Path signatures of the handle's rows for samples [first_sample, first_sample + n_samples): out[(band pixel * n_samples) + k]
 * = FNV-1a-32 over the object index of every cast of that path (0xffffffff = miss) in the low word -- two paths have
 * DIVERGED iff these differ -- and FNV-1a-32 over the bits of every hit distance in the high word.  out: host pointer,
 * local_rows * width * n_samples entries. */
int amber_hip_kat_signatures(amber_hip_pt*, uint32_t first_sample, uint32_t n_samples, uint64_t* out);
/* The same signatures from the PRODUCT render kernel (pt_megakernel / pt_bvh_megakernel / pt_bvh_pool_kernel instantiated with the hashing
 * switched on: identical scheduling, work queue, ray pool and device functions), so that the kernel that renders -- not
 * only the per-thread known-answer kernel above -- is compared with the oracle path by path
 * (algorithm_pt.cc:125-160).  Same layout as amber_hip_kat_signatures.  Leaves the framebuffer and the ray count untouched. */
int amber_hip_pt_signatures(amber_hip_pt*, uint32_t first_sample, uint32_t n_samples, uint64_t* out);
/* Engine BVH's traversal in a kernel of its own (no shading, no path state): closest hits of n rays, `waves` resident waves per SIMD
 * (4, 5, 6 or 8), idle lanes refilled once `refill_min` of a wave's 64 lanes are idle; out_t = NaN for a miss; best_ms = the fastest of
 * `repeats` launches; out_rounds (may be NULL): per ray, the number of wave rounds it was in flight for.  A measurement (DESIGN.md
 * section 5) that doubles as a known-answer test of the traversal. */
int amber_hip_kat_traversal_rate(amber_hip_pt*, uint32_t n, const float* origins, const float* dirs, uint32_t waves, uint32_t refill_min,
                                 uint32_t repeats, float* out_t, int32_t* out_object, double* best_ms, uint32_t* out_rounds);
/* Two-phase engine: the per-pixel candidate masks of the primary rays (pixel_mask_kernel, one mask per 4 x 4 block of pixels; for scenes of more than 32
 * objects the masks describe group 0 -- the aperture blades and the first 26 objects).
 * out_mask: one word per band pixel, bit k = the object in filter-program slot k can be hit by SOME eye ray of the pixel (aperture blades
 * excluded: they are added per ray).  out_slot_of_object: n_objects entries, the slot of every scene object (0xffffffff: none).
 * out_always_mask: the slots that are candidates of EVERY ray whatever the pixel (objects the filter program has no record for, and the
 * aperture blades, which a primary ray adds itself).  kernel_ms: duration of the mask kernel (create has run it once; asking for the duration runs it again between two events).
 * Any pointer may be NULL. */
int amber_hip_kat_pixel_masks(amber_hip_pt*, uint32_t* out_mask, uint32_t* out_slot_of_object, uint32_t* out_always_mask, double* kernel_ms);
/* the engine's sin/cos/pow on device: mode 0 = sincos(x[i]) -> out[2i], out[2i+1] ; mode 1 = pow(x[2i], x[2i+1]) -> out[i] ;
 * mode 2 / 3 = x[i]^4 / x[i]^5 in binary64 -> out[2i], out[2i+1] = low, high word of the double */
int amber_hip_kat_math(int device, int mode, uint32_t n, const float* x, float* out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
