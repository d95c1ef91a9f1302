// dev_bvh.h -- part of pt_device.h (included from there, in order; not a stand-alone header): engine BVH: per-lane traversal of the quantised 2-wide tree, two-stage sphere and triangle leaves, resumable rounds.
#pragma once

namespace amber_dev {

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

}  // namespace amber_dev
