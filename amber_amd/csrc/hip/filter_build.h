// filter_build.h -- host-side construction of the Phase-A filter program of the two-phase closest hit
// (device side: ClosestHitTwoPhase in dev_two_phase.h).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/amber_hip.h"
#include "pt_device.h"

namespace amber_filter {

using amber_dev::DevObject;
using amber_dev::DevPlane;
using amber_dev::DevSphereFilter;
using amber_dev::DevTriFilter;

// Phase-A program of the two-phase closest hit.  Everything here only has to be CONSERVATIVE: a candidate that
// the filter keeps is decided by the exact reference-arithmetic test, so these values are computed in double and
// rounded once.  Tolerances: the reference's binary32 Moeller-Trumbore has an absolute error of at most
// ~6 eps (|o - v0| + 1.5 |E|) / |cos| in world units along the triangle's plane; with eps = 2^-24 and a safety
// factor the bound used is  c / |n.d|,  c = 32 eps (scene diameter + 1.5 max edge), turned into barycentric units
// with the gradient magnitudes of the affine barycentric maps.  Coplanar triangles (all vertices within 4 eps of
// the scene diameter of the first triangle's plane) share a plane record; their distance to it joins the tolerance.
struct FilterProgram {
  std::vector<DevPlane> planes;
  std::vector<DevTriFilter> tris;
  std::vector<DevSphereFilter> spheres;
  std::vector<uint32_t> order;          // program slot -> scene object index
  uint32_t n_prog_tris = 0, always_mask = 0;
  uint32_t n_simple_planes = 0;         // the first planes of the program (an even number) form slabs: two parallel planes of one parallelogram pair each
};

// `center`: the filter works in coordinates relative to this point (the device subtracts it from the ray origin), so its
// affine maps see magnitudes of the order of the scene size however far the scene lies from the world origin; in
// absolute coordinates u = A.P + a0 cancels two large terms and loses the precision the tolerances assume
// (found by fuzzing with scenes offset by 1e4 of their size).
// `subset` (engine TWO_PHASE_N): the scene indices of the group the program is for, at most 32; null = the scene's first 32 objects.  Tolerances
// always derive from the bounds of ALL objects.
inline void BuildFilterProgram(const std::vector<DevObject>& objs, const float center[3], FilterProgram& fp, const std::vector<uint32_t>* subset = nullptr) {
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, e_max = 0;
  auto grow = [&](double x, double y, double z) {
    const double p[3] = {x, y, z};
    for (int c = 0; c < 3; c++) { if (p[c] < lo[c]) lo[c] = p[c]; if (p[c] > hi[c]) hi[c] = p[c]; }
  };
  auto len3 = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  for (const DevObject& o : objs) {
    if (o.kind == AMBER_PRIM_TRIANGLE) {
      grow(o.a[0], o.a[1], o.a[2]);
      grow(double(o.a[0]) + o.e1[0], double(o.a[1]) + o.e1[1], double(o.a[2]) + o.e1[2]);
      grow(double(o.a[0]) + o.e2[0], double(o.a[1]) + o.e2[1], double(o.a[2]) + o.e2[2]);
      const double E1[3] = {o.e1[0], o.e1[1], o.e1[2]}, E2[3] = {o.e2[0], o.e2[1], o.e2[2]};
      e_max = std::max(e_max, std::max(len3(E1), len3(E2)));
    } else {
      const double r = std::fabs(double(o.radius)) + std::fabs(double(o.height));
      grow(o.a[0] - r, o.a[1] - r, o.a[2] - r); grow(o.a[0] + r, o.a[1] + r, o.a[2] + r);
    }
  }
  const double diam = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  const double eps = 5.9604644775390625e-08;
  const double c = 32.0 * eps * (diam + 1.5 * e_max);
  // per triangle of a group: the three affine coordinate functions (of v0, v1, v2) in double, for pairing
  struct Tri { uint32_t index; double v[3][3]; double coef[3][4]; double g; };   // coef[k] = (gradient, constant) of the coordinate of vertex k
  struct Group { double n[3], d0, kt, ktol; std::vector<Tri> tris; bool flip = false, same_normal = false; };
  std::vector<Group> groups;
  std::vector<uint32_t> sphere_index, always_index;
  const size_t n_members = subset ? std::min<size_t>(subset->size(), 32) : std::min<size_t>(objs.size(), 32);
  for (size_t member = 0; member < n_members; member++) {
    const uint32_t idx = subset ? (*subset)[member] : static_cast<uint32_t>(member);
    const DevObject& o = objs[idx];
    if (o.kind == AMBER_PRIM_TRIANGLE) {
      const double v0[3] = {double(o.a[0]) - center[0], double(o.a[1]) - center[1], double(o.a[2]) - center[2]};   // centred
      const double E1[3] = {o.e1[0], o.e1[1], o.e1[2]}, E2[3] = {o.e2[0], o.e2[1], o.e2[2]};
      const double nr[3] = {E1[1] * E2[2] - E1[2] * E2[1], E1[2] * E2[0] - E1[0] * E2[2], E1[0] * E2[1] - E1[1] * E2[0]};
      const double n2 = nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2];
      if (!(n2 > 1e-60) || !std::isfinite(n2)) { always_index.push_back(idx); continue; }
      const double nl = std::sqrt(n2);
      const double n[3] = {nr[0] / nl, nr[1] / nl, nr[2] / nl};
      const double d0 = n[0] * v0[0] + n[1] * v0[1] + n[2] * v0[2];
      // u = A.(P - v0), A = (E2 x nr)/|nr|^2 ; v = B.(P - v0), B = (nr x E1)/|nr|^2
      const double A[3] = {(E2[1] * nr[2] - E2[2] * nr[1]) / n2, (E2[2] * nr[0] - E2[0] * nr[2]) / n2, (E2[0] * nr[1] - E2[1] * nr[0]) / n2};
      const double B[3] = {(nr[1] * E1[2] - nr[2] * E1[1]) / n2, (nr[2] * E1[0] - nr[0] * E1[2]) / n2, (nr[0] * E1[1] - nr[1] * E1[0]) / n2};
      const double AB[3] = {A[0] + B[0], A[1] + B[1], A[2] + B[2]};
      const double g = std::max(len3(A), std::max(len3(B), len3(AB)));
      const double inv_sin = len3(E1) * len3(E2) / nl;
      // find a plane group that contains all three vertices
      const double verts[3][3] = {{v0[0], v0[1], v0[2]}, {v0[0] + E1[0], v0[1] + E1[1], v0[2] + E1[2]}, {v0[0] + E2[0], v0[1] + E2[1], v0[2] + E2[2]}};
      Group* grp = nullptr; double dist = 0;
      for (Group& gq : groups) {
        double dmax = 0;
        for (const auto& vtx : verts) dmax = std::max(dmax, std::fabs(gq.n[0] * vtx[0] + gq.n[1] * vtx[1] + gq.n[2] * vtx[2] - gq.d0));
        if (dmax <= 4.0 * eps * diam) { grp = &gq; dist = dmax; break; }
      }
      Tri t;
      t.index = idx; t.g = g;
      for (int k = 0; k < 3; k++) for (int c2 = 0; c2 < 3; c2++) t.v[k][c2] = verts[k][c2];
      const double a0 = -(A[0] * v0[0] + A[1] * v0[1] + A[2] * v0[2]), b0 = -(B[0] * v0[0] + B[1] * v0[1] + B[2] * v0[2]);
      for (int k = 0; k < 3; k++) { t.coef[1][k] = A[k]; t.coef[2][k] = B[k]; t.coef[0][k] = -(A[k] + B[k]); }
      t.coef[1][3] = a0; t.coef[2][3] = b0; t.coef[0][3] = 1.0 - a0 - b0;
      const double ktol = (c + dist) * g * 1.0001 + 1e-7;
      const double kt = 2.0 * (c + dist) * std::max(1.0, inv_sin) + 1e-7;
      if (!std::isfinite(ktol) || !std::isfinite(static_cast<float>(a0)) || !std::isfinite(static_cast<float>(b0)) || !std::isfinite(kt)) { always_index.push_back(idx); continue; }
      if (!grp) { groups.push_back(Group{{n[0], n[1], n[2]}, d0, 0.0, 0.0, {}, false, false}); grp = &groups.back(); }
      grp->kt = std::max(grp->kt, kt);
      grp->ktol = std::max(grp->ktol, ktol);
      grp->tris.push_back(t);
    } else if (o.kind == AMBER_PRIM_SPHERE) {
      DevSphereFilter f;
      std::memset(&f, 0, sizeof f);
      for (int k = 0; k < 3; k++) f.c[k] = static_cast<float>(double(o.a[k]) - center[k]);   // centred
      f.r2 = static_cast<float>(double(o.radius) * double(o.radius));
      f.ktol = 1e-5f;
      if (!std::isfinite(f.r2)) { always_index.push_back(idx); continue; }
      fp.spheres.push_back(f);
      sphere_index.push_back(idx);
    } else {
      always_index.push_back(idx);                   // disk, cylinder: always tested exactly
    }
  }
  auto record = [](const Tri& t, int row_a, int row_b) {          // rows = vertex whose coordinate function is evaluated
    DevTriFilter f;
    std::memset(&f, 0, sizeof f);
    for (int k = 0; k < 4; k++) { f.c[k][0] = static_cast<float>(t.coef[row_a][k]); f.c[k][1] = static_cast<float>(t.coef[row_b][k]); }
    return f;
  };
  // Program order.  Phase A prunes by distance ONE-SIDEDLY (ClosestHitTwoPhase): a certain hit only removes candidates
  // that are evaluated after it, so occluders should come first.  Spheres are evaluated before all planes; planes are
  // sorted by the distance of their triangles' centroid from the area-weighted centroid of all triangles -- inner
  // surfaces (the water surface, the light just under the ceiling) before the enclosing walls, the far-away aperture
  // last.  Any order is correct; this one only makes the pruning effective.
  {
    double cw[3] = {0, 0, 0}, wsum = 0;
    std::vector<double> gc(groups.size() * 3, 0.0);
    for (size_t gi = 0; gi < groups.size(); gi++) {
      double gw = 0;
      for (const Tri& t : groups[gi].tris) {
        const double e1[3] = {t.v[1][0] - t.v[0][0], t.v[1][1] - t.v[0][1], t.v[1][2] - t.v[0][2]};
        const double e2[3] = {t.v[2][0] - t.v[0][0], t.v[2][1] - t.v[0][1], t.v[2][2] - t.v[0][2]};
        const double cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double area = 0.5 * len3(cr);
        for (int k = 0; k < 3; k++) gc[gi * 3 + k] += area * (t.v[0][k] + t.v[1][k] + t.v[2][k]) / 3.0;
        gw += area;
      }
      for (int k = 0; k < 3; k++) { cw[k] += gc[gi * 3 + k]; if (gw > 0) gc[gi * 3 + k] /= gw; }
      wsum += gw;
    }
    if (wsum > 0) for (int k = 0; k < 3; k++) cw[k] /= wsum;
    // Parallel planes are kept adjacent (they share n.d, n.o and the reciprocal on the device): planes are classed by
    // normal direction up to sign; classes are ordered by their innermost member, members by distance.  A member whose
    // normal is the negative of its class's first normal is stored with normal and offset negated (the same plane).
    std::vector<int> cls(groups.size(), -1);
    std::vector<size_t> cls_first;
    for (size_t gi = 0; gi < groups.size(); gi++) {
      for (size_t c = 0; c < cls_first.size() && cls[gi] < 0; c++) {
        const Group& f = groups[cls_first[c]];
        const double dot = f.n[0] * groups[gi].n[0] + f.n[1] * groups[gi].n[1] + f.n[2] * groups[gi].n[2];
        const double sgn = dot < 0 ? -1.0 : 1.0;
        // The same direction only if the BINARY32 normals the device will use are identical up to sign: the member then
        // adopts the class's normal without moving its plane (its d0 was computed with its own normal; a merely nearly
        // parallel plane -- 1e-6 rad passes a 1e-12 test on the dot product -- would be tilted by that angle times the scene
        // size, which the rounding-only tolerances kt / ktol do not cover).
        bool same = true;
        for (int k = 0; k < 3; k++) same = same && static_cast<float>(groups[gi].n[k]) == static_cast<float>(sgn * f.n[k]);
        if (same) {
          cls[gi] = static_cast<int>(c);
          for (int k = 0; k < 3; k++) groups[gi].n[k] = sgn * f.n[k];
          groups[gi].flip = dot < 0;
        }
      }
      if (cls[gi] < 0) { cls[gi] = static_cast<int>(cls_first.size()); cls_first.push_back(gi); }
    }
    auto dist2 = [&](size_t gi) { double d2 = 0; for (int k = 0; k < 3; k++) d2 += (gc[gi * 3 + k] - cw[k]) * (gc[gi * 3 + k] - cw[k]); return d2; };
    std::vector<double> cls_min(cls_first.size(), 1e300);
    for (size_t gi = 0; gi < groups.size(); gi++) cls_min[cls[gi]] = std::min(cls_min[cls[gi]], dist2(gi));
    std::vector<size_t> perm(groups.size());
    for (size_t gi = 0; gi < perm.size(); gi++) perm[gi] = gi;
    std::stable_sort(perm.begin(), perm.end(), [&](size_t x, size_t y) {
      if (cls[x] != cls[y]) return cls_min[cls[x]] != cls_min[cls[y]] ? cls_min[cls[x]] < cls_min[cls[y]] : cls[x] < cls[y];
      return dist2(x) < dist2(y);
    });
    std::vector<Group> sorted;
    int prev_cls = -1;
    for (size_t gi : perm) {
      sorted.push_back(groups[gi]);
      sorted.back().same_normal = cls[gi] == prev_cls;
      prev_cls = cls[gi];
    }
    groups.swap(sorted);
  }
  for (const Group& g : groups) {
    // Parallelogram pairs: triangle j shares two corners with triangle i and its third corner is D = A + C - B
    // (B = i's unshared corner), everything up to 1e-6 of the scene diameter; the actual mismatch (corners are
    // rebuilt from binary32 edge vectors) joins the barycentric tolerance through the gradient magnitude.
    const size_t nt = g.tris.size();
    std::vector<int> partner(nt, -1), unshared(nt, -1);
    double pair_slack = 0;
    auto apart = [](const double* p, const double* r) { return std::max(std::fabs(p[0] - r[0]), std::max(std::fabs(p[1] - r[1]), std::fabs(p[2] - r[2]))); };
    for (size_t i = 0; i < nt; i++) {
      if (partner[i] >= 0) continue;
      for (size_t j = i + 1; j < nt && partner[i] < 0; j++) {
        if (partner[j] >= 0) continue;
        int match_i[3] = {-1, -1, -1}, n_shared = 0;               // match_i[k] = corner of j equal to corner k of i
        bool used_j[3] = {false, false, false};
        double err = 0;
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++)
            if (!used_j[l] && match_i[k] < 0 && apart(g.tris[i].v[k], g.tris[j].v[l]) <= 1e-6 * diam) {
              match_i[k] = l; used_j[l] = true; n_shared++;
              err = std::max(err, apart(g.tris[i].v[k], g.tris[j].v[l]));
            }
        if (n_shared != 2) continue;
        int bi = -1, dj = -1;
        for (int k = 0; k < 3; k++) if (match_i[k] < 0) bi = k;
        for (int l = 0; l < 3; l++) if (!used_j[l]) dj = l;
        double expect[3];
        for (int c2 = 0; c2 < 3; c2++) {
          expect[c2] = -g.tris[i].v[bi][c2];
          for (int k = 0; k < 3; k++) if (k != bi) expect[c2] += g.tris[i].v[k][c2];
        }
        err = std::max(err, apart(expect, g.tris[j].v[dj]));
        if (err <= 1e-6 * diam) {
          partner[i] = static_cast<int>(j); partner[j] = static_cast<int>(i); unshared[i] = bi;
          pair_slack = std::max(pair_slack, 4.0 * err * std::max(g.tris[i].g, g.tris[j].g));
        }
      }
    }
    DevPlane p;
    std::memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; k++) p.n[k] = static_cast<float>(g.flip ? -g.n[k] : g.n[k]);   // the class's first normal, bit for bit
    // the pair evaluation adds two roundings of 1 - x (<= 1.2e-7 each) and the corner mismatch measured above
    p.d0 = static_cast<float>(g.flip ? -g.d0 : g.d0); p.kt = static_cast<float>(g.kt * 1.0001); p.ktol = static_cast<float>((g.ktol + 4e-7 + pair_slack) * 1.0001);
    for (size_t i = 0; i < nt; i++) {
      if (partner[i] < 0 || static_cast<size_t>(partner[i]) < i) continue;
      const int bi = unshared[i], ai = (bi + 1) % 3;
      fp.tris.push_back(record(g.tris[i], bi, ai));
      fp.order.push_back(g.tris[i].index); fp.order.push_back(g.tris[static_cast<size_t>(partner[i])].index);
      p.n_pairs++;
    }
    for (size_t i = 0; i < nt; i++) {
      if (partner[i] >= 0) continue;
      fp.tris.push_back(record(g.tris[i], 1, 2));                  // u, v = coordinates of v1, v2
      fp.order.push_back(g.tris[i].index);
      p.n_tris++;
    }
    if (g.same_normal) p.n_tris |= 0x80000000u;
    fp.planes.push_back(p);
  }
  // Slabs first.  A plane that holds exactly one parallelogram pair and nothing else -- a wall, a floor, a rectangular
  // light -- and is followed by another such plane with the SAME stored normal forms a slab with it; the slabs go first, in
  // their relative order, and the device runs them two planes per iteration with neither inner record loops nor the
  // "same normal?" branch (ClosestHitTwoPhase: that code is bound by instruction delivery, and every scalar branch of the
  // nested loops costs about as much as four VALU instructions).  Everything else follows in the general form; "same
  // normal as the previous plane" is re-derived for the new order.
  {
    struct Span { size_t rec, slot; };
    std::vector<Span> span(fp.planes.size());
    { size_t rec = 0, slot = 0;
      for (size_t i = 0; i < fp.planes.size(); i++) {
        span[i] = Span{rec, slot};
        const uint32_t nt = fp.planes[i].n_tris & 0x7fffffffu, np = fp.planes[i].n_pairs;
        rec += np + nt; slot += 2 * np + nt;
      } }
    auto simple = [&](size_t i) { return fp.planes[i].n_pairs == 1 && (fp.planes[i].n_tris & 0x7fffffffu) == 0; };
    std::vector<char> in_slab(fp.planes.size(), 0);
    for (size_t i = 0; i + 1 < fp.planes.size(); i++)
      if (!in_slab[i] && simple(i) && simple(i + 1) && std::memcmp(fp.planes[i].n, fp.planes[i + 1].n, sizeof fp.planes[i].n) == 0) { in_slab[i] = in_slab[i + 1] = 1; i++; }
    std::vector<DevPlane> planes; std::vector<DevTriFilter> tris; std::vector<uint32_t> order;
    for (int pass = 0; pass < 2; pass++)
      for (size_t i = 0; i < fp.planes.size(); i++) {
        if ((in_slab[i] != 0) != (pass == 0)) continue;
        DevPlane q = fp.planes[i];
        const uint32_t nt = q.n_tris & 0x7fffffffu, np = q.n_pairs;
        q.n_tris = nt;
        if (!planes.empty() && std::memcmp(planes.back().n, q.n, sizeof q.n) == 0) q.n_tris |= 0x80000000u;
        planes.push_back(q);
        for (uint32_t k = 0; k < np + nt; k++) tris.push_back(fp.tris[span[i].rec + k]);
        for (uint32_t k = 0; k < 2 * np + nt; k++) order.push_back(fp.order[span[i].slot + k]);
        if (pass == 0) fp.n_simple_planes++;
      }
    fp.planes.swap(planes); fp.tris.swap(tris); fp.order.swap(order);
  }
  fp.n_prog_tris = static_cast<uint32_t>(fp.order.size());
  fp.order.insert(fp.order.end(), sphere_index.begin(), sphere_index.end());
  for (uint32_t idx : always_index) { fp.always_mask |= 1u << fp.order.size(); fp.order.push_back(idx); }
}

}  // namespace amber_filter
