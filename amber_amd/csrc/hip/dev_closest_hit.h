// dev_closest_hit.h -- part of pt_device.h (included from there, in order; not a stand-alone header): engine REFERENCE_BVH (the reference's own tree in the reference's order); the engine switch ClosestHit<kEngine>; ResolveHit.
#pragma once

namespace amber_dev {

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
  constexpr int32_t kDone = 0x7fffffff;
  // next subtree: the newest pending entry whose box the closest hit so far does not lie in front of (the recursion's return path, see above)
  auto next_subtree = [&]() {
    for (;;) {
      uint32_t ref; float fin;
      if (top_ref != kNone) { ref = top_ref; fin = top_in; top_ref = kNone; }
      else if (sp > 0) { --sp; const uint2 e = stack[static_cast<size_t>(sp) * stride]; ref = e.x; fin = __uint_as_float(e.y); }
      else { cur = kDone; return; }
      if (!(best.t < fin)) { cur = static_cast<int32_t>(ref); return; }
    }
  };
  // A lane's own sequence of visits is the recursion's; only how the lanes of a wave INTERLEAVE is a choice: up to AMBER_REF_BVH_BUDGET inner nodes,
  // then (all lanes that have reached one) a leaf -- so that the long leaf scans run with many lanes instead of holding up lanes that only step
  // through a node (one node or one leaf per trip: config 3 at 64 spp 59.7 ms; this form: see EXPERIMENTS.md round 5).
#ifndef AMBER_REF_BVH_BUDGET
#define AMBER_REF_BVH_BUDGET 4
#endif
  for (;;) {
    int budget = AMBER_REF_BVH_BUDGET;
    while (cur >= 0 && cur != kDone && budget-- > 0) {
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
      } else if (lh) {
        cur = left;
      } else if (rh) {
        cur = right;
      } else {
        next_subtree();
      }
    }
    if (cur < 0) {
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
      next_subtree();
    }
    if (cur == kDone) break;
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

}  // namespace amber_dev
