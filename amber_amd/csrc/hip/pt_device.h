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

// The device functions live in seven parts, included here in the order they build on each other (one translation unit: pt_host.hip):
//   dev_scene.h        diagnostic macros of the stamped builds; the scene layout in HBM (DevObject ... DevScene)
//   dev_math.h         Vector3, the XorShift sampler, glibc's sincosf / powf kernels restated
//   dev_primitives.h   the reference's primitive tests and hit rule; engine LIST
//   dev_two_phase.h    engine TWO_PHASE: Phase-A filter, Phase-B exact tests, groups of 32 objects
//   dev_bvh.h          engine BVH: quantised 2-wide tree, two-stage leaves, resumable rounds
//   dev_closest_hit.h  engine REFERENCE_BVH; ClosestHit<kEngine>; ResolveHit
//   dev_shading.h      materials, eye ray, light-path start, PathStep / PathShade
#include "dev_scene.h"
#include "dev_math.h"
#include "dev_primitives.h"
#include "dev_two_phase.h"
#include "dev_bvh.h"
#include "dev_closest_hit.h"
#include "dev_shading.h"
