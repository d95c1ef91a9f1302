// dev_two_phase.h -- part of pt_device.h (included from there, in order; not a stand-alone header): engine TWO_PHASE (and its grouped form): the wave-uniform Phase-A filter and the per-lane exact tests of Phase B.
#pragma once

namespace amber_dev {

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

}  // namespace amber_dev
