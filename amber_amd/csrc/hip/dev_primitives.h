// dev_primitives.h -- part of pt_device.h (included from there, in order; not a stand-alone header): the reference's primitive tests (triangle, sphere, disk, cylinder) and the hit rule; engine LIST.
#pragma once

namespace amber_dev {

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

}  // namespace amber_dev
