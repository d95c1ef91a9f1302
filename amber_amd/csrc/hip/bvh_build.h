// bvh_build.h -- host-side BVH build and flattening for the device traversal (engine BVH).
//
// Replaces amber::raytracer::BVH (/root/reference/include/amber/raytracer/acceleration_bvh.h:134-312: top-down SAH,
// pointer tree of unique_ptr<Node>, recursive Cast) with a builder designed for the GPU side:
//   * binned SAH (16 bins per axis over the CENTROID bounds, O(N log N)) instead of the reference's
//     3 sorts + 15 candidate planes per node (26.6 s for 1M spheres);
//   * a flat array of 64-byte 2-wide nodes (planes grouped for packed FMAs): a node stores BOTH children's boxes, so one 64-byte fetch decides
//     both subtrees (the reference also tests both children at the parent, acceleration_bvh.h:360-372);
//   * leaves of up to kLeafSize objects referenced through a permutation array;
//   * depth capped (kMaxDepth) so that the per-lane traversal stack fits a fixed LDS allocation.
// The tree only has to be CONSERVATIVE: the closest hit is decided by the exact reference-arithmetic primitive
// tests with the (t, object index) tie rule, i.e. the List semantics (acceleration_list.h:51-68), independent of
// topology.  Boxes are padded (see PadBox) so that rounding in the primitive tests can never place an accepted
// hit outside its leaf's box.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pt_device.h"

namespace amber_bvh {

using amber_dev::DevBvhNode;
using amber_dev::DevBvhNodeQ;
using amber_dev::DevBvhNodeQ4;
using amber_dev::DevObject;

#ifndef AMBER_BVH_LEAF_SIZE
#define AMBER_BVH_LEAF_SIZE 3   // config 3 at 128 spp: 2 -> 151.5 ms, 3 -> 151.1, 4 -> 157.9, 6 -> 173.6 (round 1); 1 -> 120.2, 2 -> 106.3, 3 -> 101.6 (round 3: two-stage sphere leaves, binary16 planes)
#endif
constexpr int kLeafSize = AMBER_BVH_LEAF_SIZE;   // <= 3 (quantised leaf references: 2-bit count + the spheres-only flag)
static_assert(kLeafSize >= 1 && kLeafSize <= 3, "DevBvhNodeQ leaf references hold a 2-bit count");
constexpr int kMaxDepth = 30;     // device stack holds 32 entries
#ifndef AMBER_BVH_BINS
#define AMBER_BVH_BINS 16   // config 3 at 128 spp (round 3): 8 bins 102.4 ms, 16 -> 101.9, 32 -> 104.5
#endif
constexpr int kBins = AMBER_BVH_BINS;

struct Box {
  float mn[3], mx[3];
  void reset() { for (int c = 0; c < 3; c++) { mn[c] = 3.0e38f; mx[c] = -3.0e38f; } }
  void grow(const Box& b) { for (int c = 0; c < 3; c++) { mn[c] = std::min(mn[c], b.mn[c]); mx[c] = std::max(mx[c], b.mx[c]); } }
  void grow(const float p[3]) { for (int c = 0; c < 3; c++) { mn[c] = std::min(mn[c], p[c]); mx[c] = std::max(mx[c], p[c]); } }
  double area() const {
    const double x = double(mx[0]) - mn[0], y = double(mx[1]) - mn[1], z = double(mx[2]) - mn[2];
    return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + y * z + z * x);
  }
};

// Bounds of one object (same shapes as Primitive::BoundingBox, primitive_*.cc), widened outward.
// sphere_slack2: the reference's sphere test (primitive_sphere.cc:75-107, algebra.h:31-52) evaluates the discriminant
// b*b - 4*c in binary32 with b ~ 2D and c ~ D^2 for a ray origin at distance D, so it ACCEPTS rays whose impact
// parameter p satisfies p^2 <= r^2 + ~12 eps D^2 -- for a small distant sphere noticeably outside the geometric
// sphere (r = 0.005, D = 6: 3.5e-3).  Those accepted hits must stay inside the leaf's box, so spheres are bounded
// with r' = sqrt(r^2 + 16 eps D_scene^2) (found when the 1M-sphere scene lost 7 paths in 1.4e9 to tighter boxes).
// tri_reach: the reference's Moeller-Trumbore test (primitive_triangle.cc:97-128) forms u and v with an absolute error of
// about 6 eps |T| |E| / |det| each and tests fl(u) + fl(v) <= 1, so along a NEEDLE (one edge much shorter than the
// others) it accepts rays that cross the plane up to ~ 36 eps |T| e_max / e_min beyond the short edge -- 2.5e-3 scene
// units in the fuzzer's seed 308053 (e_max / e_min = 1e4), far outside the geometric box.  Triangle boxes are widened
// by that amount with |T| <= tri_reach (the scene diameter); for a well-shaped triangle it is 1e-6 of the scene.
// What no static box can cover is a ray within a fraction of a degree of a triangle's plane and nearly parallel to
// one of its edges, where the same error grows without bound: engine LIST keeps such numerical-noise hits, engine
// BVH (like the reference's own BVH, which culls with unpadded boxes) may not -- see DESIGN.md section 5.
inline Box ObjectBox(const DevObject& o, double sphere_slack2 = 0.0, double tri_reach = 0.0) {
  Box b; b.reset();
  const uint32_t kind = o.kind & 0xffu;
  if (kind == 0) {            // triangle: v0, v0+E1, v0+E2
    float p[3];
    b.grow(o.a);
    for (int c = 0; c < 3; c++) p[c] = o.a[c] + o.e1[c];
    b.grow(p);
    for (int c = 0; c < 3; c++) p[c] = o.a[c] + o.e2[c];
    b.grow(p);
    if (tri_reach > 0.0) {
      double l1 = 0, l2 = 0, l3 = 0;
      for (int c = 0; c < 3; c++) { l1 += double(o.e1[c]) * o.e1[c]; l2 += double(o.e2[c]) * o.e2[c]; const double e3 = double(o.e2[c]) - o.e1[c]; l3 += e3 * e3; }
      const double emax = std::sqrt(std::max(l1, std::max(l2, l3))), emin = std::sqrt(std::min(l1, std::min(l2, l3)));
      if (emin > 0.0 && std::isfinite(emax)) {
        const double m = std::min(tri_reach, 36.0 * 5.9604644775390625e-08 * tri_reach * emax / emin);
        for (int c = 0; c < 3; c++) { b.mn[c] = static_cast<float>(b.mn[c] - m); b.mx[c] = static_cast<float>(b.mx[c] + m); }
      }
    }
  } else if (kind == 1) {     // sphere
    const float r = static_cast<float>(std::sqrt(double(o.radius) * o.radius + sphere_slack2) * 1.000001);
    for (int c = 0; c < 3; c++) { b.mn[c] = o.a[c] - r; b.mx[c] = o.a[c] + r; }
  } else if (kind == 2) {     // disk: the test is geometric for any normal length (t from the plane, |p - c|^2 <= r^2, primitive_disk.cc:94-114)
    const double r = std::fabs(double(o.radius)) * 1.000001;
    for (int c = 0; c < 3; c++) { b.mn[c] = static_cast<float>(o.a[c] - r); b.mx[c] = static_cast<float>(o.a[c] + r); }
  } else {                    // cylinder (primitive_cylinder.cc:100-142) with axis vector N of length nu, not guaranteed unit:
    // the accepted points P satisfy |P_perp|^2 + (1 - nu^2)^2 |P_par|^2 = r^2 and 0 <= P.N <= height, i.e. radial distance
    // <= r and axial distance <= height / nu from the base centre: bounding sphere of radius hypot(r, height / nu)
    const double nu = std::sqrt(double(o.e1[0]) * o.e1[0] + double(o.e1[1]) * o.e1[1] + double(o.e1[2]) * o.e1[2]);
    const double axial = nu > 1e-30 ? std::fabs(double(o.height)) / nu : 1e30;
    const double r = std::min(1e30, std::hypot(double(o.radius), axial) * 1.000001);
    for (int c = 0; c < 3; c++) { b.mn[c] = static_cast<float>(o.a[c] - r); b.mx[c] = static_cast<float>(o.a[c] + r); }
  }
  return b;
}

// Padding: an accepted reference hit lies within a few binary32 roundings of the primitive (relative to the
// coordinates involved); 2^-16 of the scene extent plus 2^-16 of the coordinate magnitude is ~250x that.
inline void PadBox(Box& b, float scene_extent) {
  for (int c = 0; c < 3; c++) {
    const float m = std::max(std::fabs(b.mn[c]), std::fabs(b.mx[c]));
    const float pad = 1.52587890625e-05f * (scene_extent + m) + 1e-30f;
    b.mn[c] -= pad; b.mx[c] += pad;
  }
}

// Arrays shared by every builder of one tree: index ranges handed to different builders are disjoint.
struct Shared {
  std::vector<Box> boxes;
  std::vector<float> cent;          // 3 per object
  std::vector<uint32_t> index;      // permutation (leaf order)
  float extent = 0;
};

// A subtree whose construction was deferred to a worker thread: range, depth, and where its root reference goes.
struct Deferred { uint32_t first, last, depth, parent; bool is_left; };

struct Builder {
  Shared& s;
  std::vector<DevBvhNode> nodes;
  uint32_t max_depth_seen = 0;
  uint32_t grain = 0;               // > 0: ranges of at most `grain` objects below depth 0 are deferred, not built
  std::vector<Deferred>* deferred = nullptr;

  explicit Builder(Shared& shared) : s(shared) {}

  static int32_t LeafRef(uint32_t first, uint32_t count) { return -static_cast<int32_t>(first * 8u + count) - 1; }   // count <= 7
  static constexpr int32_t kPending = 0x7ffffffe;       // placeholder of a deferred child reference

  // returns child reference (>= 0 inner node index, < 0 leaf) and the box of the range
  int32_t Build(uint32_t first, uint32_t last, uint32_t depth, Box& out_box, uint32_t parent = 0, bool is_left = false) {
    std::vector<Box>& boxes = s.boxes;
    std::vector<float>& cent = s.cent;
    std::vector<uint32_t>& index = s.index;
    max_depth_seen = std::max(max_depth_seen, depth);
    Box bb; bb.reset();
    Box cb; cb.reset();
    for (uint32_t i = first; i < last; i++) { bb.grow(boxes[index[i]]); cb.grow(&cent[3 * index[i]]); }
    out_box = bb;
    const uint32_t n = last - first;
    if (n <= static_cast<uint32_t>(kLeafSize)) return LeafRef(first, n);
    if (grain && depth > 0 && n <= grain) { deferred->push_back(Deferred{first, last, depth, parent, is_left}); return kPending; }
    // the remaining levels must be able to hold n objects even with median splits
    const bool force_median = (depth + 1 + static_cast<uint32_t>(std::ceil(std::log2(double(n) / kLeafSize))) + 2) >= static_cast<uint32_t>(kMaxDepth);
    uint32_t mid = first + n / 2;
    int axis = 0;
    {
      float best_ext = -1;
      for (int c = 0; c < 3; c++) { const float e = cb.mx[c] - cb.mn[c]; if (e > best_ext) { best_ext = e; axis = c; } }
    }
    bool split_found = false;
    if (!force_median) {
      double best_cost = 1e300; int best_axis = -1, best_bin = -1;
      for (int c = 0; c < 3; c++) {
        const float lo = cb.mn[c], ext = cb.mx[c] - cb.mn[c];
        if (!(ext > 0)) continue;
        Box bin_box[kBins]; uint32_t bin_cnt[kBins];
        for (int b = 0; b < kBins; b++) { bin_box[b].reset(); bin_cnt[b] = 0; }
        const float scale = kBins / ext;
        for (uint32_t i = first; i < last; i++) {
          int b = static_cast<int>((cent[3 * index[i] + c] - lo) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          bin_box[b].grow(boxes[index[i]]); bin_cnt[b]++;
        }
        double right_area[kBins]; uint32_t right_cnt[kBins];
        Box acc; acc.reset(); uint32_t cnt = 0;
        for (int b = kBins - 1; b > 0; b--) { acc.grow(bin_box[b]); cnt += bin_cnt[b]; right_area[b] = acc.area(); right_cnt[b] = cnt; }
        acc.reset(); cnt = 0;
        for (int b = 0; b < kBins - 1; b++) {
          acc.grow(bin_box[b]); cnt += bin_cnt[b];
          if (cnt == 0 || right_cnt[b + 1] == 0) continue;
          const double cost = acc.area() * cnt + right_area[b + 1] * right_cnt[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = c; best_bin = b; }
        }
      }
      if (best_axis >= 0) {
        const float lo = cb.mn[best_axis], scale = kBins / (cb.mx[best_axis] - cb.mn[best_axis]);
        auto it = std::partition(index.begin() + first, index.begin() + last, [&](uint32_t id) {
          int b = static_cast<int>((cent[3 * id + best_axis] - lo) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          return b <= best_bin;
        });
        mid = static_cast<uint32_t>(it - index.begin());
        split_found = mid > first && mid < last;
      }
    }
    if (!split_found) {   // median split along the widest centroid axis (also the depth-cap fallback)
      mid = first + n / 2;
      std::nth_element(index.begin() + first, index.begin() + mid, index.begin() + last,
                       [&](uint32_t a, uint32_t b) { return cent[3 * a + axis] < cent[3 * b + axis]; });
    }
    const uint32_t me = static_cast<uint32_t>(nodes.size());
    nodes.emplace_back();
    Box lb, rb;
    const int32_t l = Build(first, mid, depth + 1, lb, me, true);
    const int32_t r = Build(mid, last, depth + 1, rb, me, false);
    PadBox(lb, s.extent); PadBox(rb, s.extent);
    DevBvhNode& nd = nodes[me];
    nd.lxy[0] = lb.mn[0]; nd.lxy[1] = lb.mn[1]; nd.lxy[2] = lb.mx[0]; nd.lxy[3] = lb.mx[1];
    nd.rxy[0] = rb.mn[0]; nd.rxy[1] = rb.mn[1]; nd.rxy[2] = rb.mx[0]; nd.rxy[3] = rb.mx[1];
    nd.z[0] = lb.mn[2]; nd.z[1] = lb.mx[2]; nd.z[2] = rb.mn[2]; nd.z[3] = rb.mx[2];
    nd.left = l; nd.right = r; nd.pad0 = 0; nd.pad1 = 0;
    return static_cast<int32_t>(me);
  }
};

struct FlatBvh {
  float bounds_min[3] = {0, 0, 0}, bounds_max[3] = {0, 0, 0};   // bounds of all (widened) object boxes
  float min_sphere_radius = 0;       // smallest |radius| over the spheres, 0 if there is none (BvhBegin's per-ray margin)
  bool has_spheres = false;
  std::vector<DevBvhNode> nodes;     // nodes[0] is the root (if root_ref >= 0)
  std::vector<uint32_t> prim_index;  // leaf order -> object index
  int32_t root_ref = -1;             // >= 0: inner node 0 ; < 0: the whole scene is one leaf
  uint32_t depth = 0;
};

// ---- tree quality (round 5) -----------------------------------------------------------------------------
// Surface-area-heuristic figures of a built tree, from the (padded) child boxes its nodes carry: the expected number of inner nodes
// and of leaf objects a random ray that hits the scene box visits (Goldsmith / Salmon), and how much the leaves overlap.
struct BvhQuality {
  double inner_area = 0;       // sum over inner nodes of area(node) / area(root): expected node visits
  double leaf_area = 0;        // sum over leaves of area(leaf) / area(root): expected leaf visits
  double leaf_object_area = 0; // the same weighted by the leaf's object count: expected exact tests
  double leaf_volume = 0;      // sum of leaf volumes / root volume: how many leaf boxes contain a random point (1 = no overlap, no gaps)
  uint32_t leaves = 0, inner = 0, depth = 0;
};
inline Box ChildBox(const DevBvhNode& nd, int side) {
  Box b;
  const float* xy = side ? nd.rxy : nd.lxy;
  b.mn[0] = xy[0]; b.mn[1] = xy[1]; b.mx[0] = xy[2]; b.mx[1] = xy[3];
  b.mn[2] = nd.z[side ? 2 : 0]; b.mx[2] = nd.z[side ? 3 : 1];
  return b;
}
inline void SetChildBox(DevBvhNode& nd, int side, const Box& b) {
  float* xy = side ? nd.rxy : nd.lxy;
  xy[0] = b.mn[0]; xy[1] = b.mn[1]; xy[2] = b.mx[0]; xy[3] = b.mx[1];
  nd.z[side ? 2 : 0] = b.mn[2]; nd.z[side ? 3 : 1] = b.mx[2];
}
inline double BoxVolume(const Box& b) {
  const double x = double(b.mx[0]) - b.mn[0], y = double(b.mx[1]) - b.mn[1], z = double(b.mx[2]) - b.mn[2];
  return (x < 0 || y < 0 || z < 0) ? 0.0 : x * y * z;
}
inline BvhQuality MeasureBvh(const std::vector<DevBvhNode>& nodes, int32_t root_ref) {
  BvhQuality q;
  if (root_ref < 0 || nodes.empty()) return q;
  Box root = ChildBox(nodes[root_ref], 0); root.grow(ChildBox(nodes[root_ref], 1));
  const double ra = root.area(), rv = BoxVolume(root);
  struct Item { int32_t node; uint32_t depth; };
  std::vector<Item> todo{{root_ref, 1u}};
  q.inner_area = 1.0; q.inner = 1;
  while (!todo.empty()) {
    const Item it = todo.back(); todo.pop_back();
    q.depth = std::max(q.depth, it.depth);
    for (int side = 0; side < 2; side++) {
      const int32_t ref = side ? nodes[it.node].right : nodes[it.node].left;
      const Box b = ChildBox(nodes[it.node], side);
      if (ref >= 0) { q.inner_area += b.area() / ra; q.inner++; todo.push_back({ref, it.depth + 1}); }
      else {
        const uint32_t count = static_cast<uint32_t>(-(ref + 1)) & 7u;
        q.leaf_area += b.area() / ra; q.leaf_object_area += count * b.area() / ra; q.leaf_volume += rv > 0 ? BoxVolume(b) / rv : 0.0; q.leaves++;
      }
    }
  }
  return q;
}

// (Tree rotations after the build -- Kensler 2008, exchanging a node's child with a grandchild where that shrinks the inner box -- were
//  built and measured in round 5: 8 524 rotations on the 1M-sphere tree lower the inner-node term from 230.415 to 230.274 and the kernel time
//  not at all; removed.  EXPERIMENTS.md, profiles/r05_config3_treelet_rotations_ab.txt.)

// The tree the device traverses (DevBvhNodeQ): child boxes on a 16-bit grid over the bounds of all node boxes, min planes
// rounded down and max planes up (checked in exact arithmetic: gmin and step are binary32, q * step and the sum are exact
// in binary64), leaf references re-encoded as first*16 + all_triangles*8 + all_spheres*4 + count.
struct QuantizedBvh {
  std::vector<DevBvhNodeQ> nodes;
  float gmin[3] = {0, 0, 0}, step[3] = {1, 1, 1}, reach[3] = {0, 0, 0};   // plane = gmin + value * step ; reach >= |value * step| for every stored value
  int32_t root_ref = -1;
};
// AMBER_BVH_F16 (default): the 16 bits of a plane are a binary16 NUMBER u in [-1, 1], plane = centre + u * half extent -- so that the
// device's slab parameter u * A + B is ONE v_fma_mix_f32 (it converts a binary16 operand on the way in) instead of a
// v_cvt_f32_u32 and an fma: 12 of a node visit's 46 vector instructions less.  The price is resolution: 11 significant bits
// (2^-11 of the coordinate, i.e. 5e-4 of the half extent in the outer half of the scene, finer towards the centre) against the
// uniform 1.5e-5 of the integer grid -- leaf boxes of the 1M-sphere scene grow by about 1 % per axis.  Denormal binary16 values are
// never stored (what a v_fma_mix does with them depends on the wave's denormal mode): a plane inside +-2^-14 goes to 0 or +-2^-14.
#ifndef AMBER_BVH_F16
#define AMBER_BVH_F16 1
#endif
inline double F16Value(uint16_t h) {                          // exact
  const int e = (h >> 10) & 31, m = h & 1023;
  const double v = e == 0 ? std::ldexp(double(m), -24) : std::ldexp(double(1024 + m), e - 25);
  return (h & 0x8000u) ? -v : v;
}
inline uint16_t F16Step(uint16_t h, bool up) {                // the next NORMAL (or zero) binary16 value above / below h; h is normal or zero, finite
  const bool neg = (h & 0x8000u) != 0;
  const uint16_t mag = h & 0x7fffu;
  if (mag == 0) return up ? 0x0400u : 0x8400u;               // 0 -> +-2^-14
  if (neg == up) return mag == 0x0400u ? 0u : static_cast<uint16_t>((neg ? 0x8000u : 0u) | (mag - 1u));   // towards zero
  return mag >= 0x7bffu ? h : static_cast<uint16_t>((neg ? 0x8000u : 0u) | (mag + 1u));                    // away from zero (stops at 65504)
}
inline uint16_t F16Nearest(double u) {                        // some normal-or-zero binary16 value near u (the caller walks to the side it needs)
  if (!(u == u)) return 0u;
  const bool neg = u < 0; double a = std::fabs(u);
  if (a < std::ldexp(1.0, -14)) return 0u;
  if (a >= 65504.0) return static_cast<uint16_t>((neg ? 0x8000u : 0u) | 0x7bffu);
  int e; const double f = std::frexp(a, &e);                  // a = f * 2^e, f in [0.5, 1)
  int m = static_cast<int>(std::floor(f * 2048.0)) - 1024;    // 11 significant bits, truncated
  if (m < 0) m = 0; if (m > 1023) m = 1023;
  return static_cast<uint16_t>((neg ? 0x8000u : 0u) | (uint32_t(e + 14) << 10) | uint32_t(m));
}
// Centre and half extent of the binary16 grid of one axis whose node boxes span [lo, hi]: |plane - mid| <= half for every plane.
// A scene with NO extent on the axis (a planar mesh in x = 1234.5) gets half = 2^-20 |mid|, not a denormal-sized step: with a step
// far below the rounding of mid, PlaneWord's search for a representable value beyond its guard walked all 30 000 binary16
// values per plane (a millisecond per node) and ended at +-65504, outside the |value| <= 1 + one step that `reach` promises.
inline void F16AxisGrid(double lo, double hi, float& mid, float& half) {
  mid = static_cast<float>(0.5 * (lo + hi));
  half = static_cast<float>(std::max(hi - double(mid), double(mid) - lo) * 1.000001);
  const float floor_half = std::max(1e-30f, std::fabs(mid) * 9.5367431640625e-07f);        // 2^-20 |mid|: 16 binary32 ulps of the coordinate
  if (!(half > floor_half)) half = floor_half;
  while (!(double(mid) + double(half) >= hi && double(mid) - double(half) <= lo)) half = std::nextafter(half, 3.0e38f);
}
// The word of one axis of a box: value(min) | value(max) << 16 with  gmin + value(min) * step <= mn  and  gmin + value(max) * step >= mx,
// checked in extended precision with a guard of 2^-50 of the operands' magnitude (one representable value further out when in doubt).
inline uint32_t PlaneWord(float mn, float mx, float gmin, float step) {
#if AMBER_BVH_F16
  const long double g = gmin, st = step;
  const long double guard = (std::fabs((long double)gmin) + std::fabs((long double)step) * 2.0L) * 0x1p-50L;
  auto plane = [&](uint16_t h) { return g + (long double)F16Value(h) * st; };
  uint16_t a = F16Nearest((double(mn) - double(gmin)) / double(step)), b = F16Nearest((double(mx) - double(gmin)) / double(step));
  while (plane(a) > (long double)mn - guard) { const uint16_t n = F16Step(a, false); if (n == a) break; a = n; }
  for (;;) { const uint16_t n = F16Step(a, true); if (n == a || plane(n) > (long double)mn - guard) break; a = n; }        // the tightest such value
  while (plane(b) < (long double)mx + guard) { const uint16_t n = F16Step(b, true); if (n == b) break; b = n; }
  for (;;) { const uint16_t n = F16Step(b, false); if (n == b || plane(n) < (long double)mx + guard) break; b = n; }
  return uint32_t(a) | (uint32_t(b) << 16);
#else
  const double g = gmin, st = step;
  double a = std::floor((double(mn) - g) / st), b = std::ceil((double(mx) - g) / st);
  a = std::min(65535.0, std::max(0.0, a)); b = std::min(65535.0, std::max(0.0, b));
  while (a > 0 && g + a * st > mn) a -= 1;                                                   // exact: conservative in every case
  while (b < 65535 && g + b * st < mx) b += 1;
  return uint32_t(a) | (uint32_t(b) << 16);
#endif
}
inline uint32_t EmptyPlaneWord() {                            // min above max: no ray enters
#if AMBER_BVH_F16
  return 0x3c00u | (0xbc00u << 16);                           // min +1, max -1
#else
  return 0x0000ffffu;                                         // min 65535, max 0
#endif
}
// -DAMBER_BVH_TRI_LEAVES=0 (measurement): leaves of triangles keep round 4's generic per-object test (64-byte record, three divisions each).
#ifndef AMBER_BVH_TRI_LEAVES
#define AMBER_BVH_TRI_LEAVES 1
#endif
// Device leaf reference: -(ref + 1) = first * 16 + all_triangles * 8 + all_spheres * 4 + count (count <= 3; first < 2^26: ValidateScene).
// `kind_of_slot(slot)` = primitive kind of the object in leaf-order slot `slot` (0 triangle, 1 sphere, ...).
template <typename KindOfSlot>
inline int32_t QuantizedLeafRef(int32_t ref, KindOfSlot kind_of_slot) {
  if (ref >= 0) return ref;
  const uint32_t r = static_cast<uint32_t>(-(ref + 1));
  const uint32_t first = r >> 3, count = r & 7u;
  bool spheres = count > 0, tris = count > 0;
  for (uint32_t k = 0; k < count; k++) { const uint32_t kind = kind_of_slot(first + k); spheres = spheres && kind == 1u; tris = tris && kind == 0u; }
#if !AMBER_BVH_TRI_LEAVES
  tris = false;
#endif
  return -static_cast<int32_t>(first * 16u + (tris ? 8u : 0u) + (spheres ? 4u : 0u) + count) - 1;
}
template <typename KindOfSlot>
inline QuantizedBvh QuantizeBvh(const std::vector<DevBvhNode>& bin, int32_t root_ref, KindOfSlot kind_of_slot) {
  QuantizedBvh out;
  out.root_ref = QuantizedLeafRef(root_ref, kind_of_slot);
  if (bin.empty()) return out;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  auto child_box = [](const DevBvhNode& nd, int side, float mn[3], float mx[3]) {
    const float* xy = side ? nd.rxy : nd.lxy;
    mn[0] = xy[0]; mn[1] = xy[1]; mx[0] = xy[2]; mx[1] = xy[3];
    mn[2] = nd.z[side ? 2 : 0]; mx[2] = nd.z[side ? 3 : 1];
  };
  for (const DevBvhNode& nd : bin)
    for (int side = 0; side < 2; side++) {
      float mn[3], mx[3]; child_box(nd, side, mn, mx);
      for (int c = 0; c < 3; c++) { lo[c] = std::min<double>(lo[c], mn[c]); hi[c] = std::max<double>(hi[c], mx[c]); }
    }
  for (int c = 0; c < 3; c++) {
#if AMBER_BVH_F16
    float mid, half;
    F16AxisGrid(lo[c], hi[c], mid, half);
    out.gmin[c] = mid; out.step[c] = half;
    out.reach[c] = static_cast<float>(double(half) * 1.002);                                // |value| <= 1 + one binary16 step
#else
    out.gmin[c] = std::nextafter(static_cast<float>(lo[c]), -3.0e38f);                    // <= lo
    double st = (hi[c] - double(out.gmin[c])) / 65535.0;
    float stf = static_cast<float>(st);
    while (!(double(out.gmin[c]) + 65535.0 * double(stf) >= hi[c])) stf = std::nextafter(stf, 3.0e38f);
    if (!(stf > 0.0f)) stf = 1e-30f;
    out.step[c] = stf;
    out.reach[c] = static_cast<float>(65535.0 * double(stf) * 1.0001);
#endif
  }
  out.nodes.resize(bin.size());
  for (size_t i = 0; i < bin.size(); i++) {
    const DevBvhNode& nd = bin[i];
    DevBvhNodeQ& q = out.nodes[i];
    for (int side = 0; side < 2; side++) {
      float mn[3], mx[3]; child_box(nd, side, mn, mx);
      for (int c = 0; c < 3; c++) q.w[3 * side + c] = PlaneWord(mn[c], mx[c], out.gmin[c], out.step[c]);   // one word per axis: min | max << 16
    }
    q.left = QuantizedLeafRef(nd.left, kind_of_slot);
    q.right = QuantizedLeafRef(nd.right, kind_of_slot);
  }
  return out;
}

// AMBER_BVH_WIDE builds: the 2-wide tree collapsed to 4-wide nodes on the grid of `grid` (QuantizeBvh's result for the same
// tree).  A node takes the two children of its 2-wide node and, while it has room, replaces the inner child with the
// largest surface by that child's two children.  Absent children: reference -1 (a leaf of no objects), inverted box.
struct QuantizedBvh4 {
  std::vector<amber_dev::DevBvhNodeQ4> nodes;
  int32_t root_ref = -1;
  uint32_t depth = 0;
};
template <typename KindOfSlot>
inline QuantizedBvh4 CollapseBvh4(const std::vector<DevBvhNode>& bin, int32_t root_ref, const QuantizedBvh& grid, KindOfSlot kind_of_slot) {
  QuantizedBvh4 out;
  out.root_ref = QuantizedLeafRef(root_ref, kind_of_slot);
  if (bin.empty() || root_ref < 0) return out;
  struct Kid { int32_t ref; float mn[3], mx[3]; };
  auto kids_of = [&](int32_t node, Kid k[2]) {
    const DevBvhNode& nd = bin[node];
    for (int side = 0; side < 2; side++) {
      const float* xy = side ? nd.rxy : nd.lxy;
      k[side].ref = side ? nd.right : nd.left;
      k[side].mn[0] = xy[0]; k[side].mn[1] = xy[1]; k[side].mx[0] = xy[2]; k[side].mx[1] = xy[3];
      k[side].mn[2] = nd.z[side ? 2 : 0]; k[side].mx[2] = nd.z[side ? 3 : 1];
    }
  };
  auto area = [](const Kid& k) { const double x = double(k.mx[0]) - k.mn[0], y = double(k.mx[1]) - k.mn[1], z = double(k.mx[2]) - k.mn[2]; return x * y + y * z + z * x; };
  struct Job { int32_t node2; uint32_t me; uint32_t depth; };
  std::vector<Job> todo;
  out.nodes.emplace_back();
  todo.push_back(Job{root_ref, 0u, 1u});
  out.root_ref = 0;
  while (!todo.empty()) {
    const Job job = todo.back(); todo.pop_back();
    out.depth = std::max(out.depth, job.depth);
    Kid kid[4]; int n = 2;
    kids_of(job.node2, kid);
    while (n < 4) {
      int best = -1; double best_area = -1;
      for (int i = 0; i < n; i++) if (kid[i].ref >= 0 && area(kid[i]) > best_area) { best_area = area(kid[i]); best = i; }
      if (best < 0) break;
      Kid two[2]; kids_of(kid[best].ref, two);
      kid[best] = two[0]; kid[n++] = two[1];
    }
    amber_dev::DevBvhNodeQ4 q;
    for (int i = 0; i < 4; i++) {
      if (i >= n) { for (int c = 0; c < 3; c++) q.w[3 * i + c] = EmptyPlaneWord(); q.child[i] = -1; continue; }
      for (int c = 0; c < 3; c++) q.w[3 * i + c] = PlaneWord(kid[i].mn[c], kid[i].mx[c], grid.gmin[c], grid.step[c]);
      if (kid[i].ref >= 0) {
        const uint32_t idx = static_cast<uint32_t>(out.nodes.size());
        out.nodes.emplace_back();
        todo.push_back(Job{kid[i].ref, idx, job.depth + 1});
        q.child[i] = static_cast<int32_t>(idx);
      } else {
        q.child[i] = QuantizedLeafRef(kid[i].ref, kind_of_slot);
      }
    }
    out.nodes[job.me] = q;
  }
  return out;
}

// Runs fn(k) for k in [0, n) on `threads` threads (work handed out by an atomic counter).
template <typename Fn>
inline void ParallelFor(size_t n, unsigned threads, Fn fn) {
  if (threads <= 1 || n <= 1) { for (size_t k = 0; k < n; k++) fn(k); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> pool;
  const unsigned nt = static_cast<unsigned>(std::min<size_t>(threads, n));
  for (unsigned t = 0; t < nt; t++) pool.emplace_back([&] { for (size_t k; (k = next.fetch_add(1)) < n;) fn(k); });
  for (auto& th : pool) th.join();
}

// The top of the tree is built on the calling thread down to ranges of n / (8 * threads) objects; those subtrees are
// built by worker threads, each into its own node array, and appended in the (deterministic) order in which the
// top-level pass met them.  Topology, boxes and leaf order do not depend on the number of threads; only the node
// numbering differs from a single-threaded build.  AMBER_BVH_THREADS overrides the thread count (1 = serial).
inline FlatBvh BuildBvh(const std::vector<DevObject>& objs) {
  const uint32_t n = static_cast<uint32_t>(objs.size());
  unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* env = std::getenv("AMBER_BVH_THREADS")) threads = static_cast<unsigned>(std::max(1, std::atoi(env)));
  if (n < 65536) threads = 1;
  Shared s;
  s.boxes.resize(n); s.cent.resize(3 * size_t(n)); s.index.resize(n);
  Box all; all.reset();
  for (uint32_t i = 0; i < n; i++) all.grow(ObjectBox(objs[i]));
  const double dx = double(all.mx[0]) - all.mn[0], dy = double(all.mx[1]) - all.mn[1], dz = double(all.mx[2]) - all.mn[2];
  double slack_factor = 16.0;
  if (const char* env = std::getenv("AMBER_BVH_SPHERE_SLACK")) slack_factor = std::atof(env);   // test hook: the image must not depend on it
  const double sphere_slack2 = slack_factor * 5.9604644775390625e-08 * (dx * dx + dy * dy + dz * dz);   // 16 eps D^2
  const double scene_diag = std::sqrt(dx * dx + dy * dy + dz * dz);
  const size_t n_chunks = threads > 1 ? threads * 4 : 1;
  std::vector<Box> chunk_box(n_chunks);
  ParallelFor(n_chunks, threads, [&](size_t c) {
    Box cb; cb.reset();
    for (size_t i = n * c / n_chunks; i < n * (c + 1) / n_chunks; i++) {
      s.boxes[i] = ObjectBox(objs[i], sphere_slack2, scene_diag);
      for (int k = 0; k < 3; k++) s.cent[3 * i + k] = 0.5f * (s.boxes[i].mn[k] + s.boxes[i].mx[k]);
      s.index[i] = static_cast<uint32_t>(i);
      cb.grow(s.boxes[i]);
    }
    chunk_box[c] = cb;
  });
  all.reset();
  for (const Box& cb : chunk_box) all.grow(cb);
  s.extent = std::max(all.mx[0] - all.mn[0], std::max(all.mx[1] - all.mn[1], all.mx[2] - all.mn[2]));

  Builder top(s);
  std::vector<Deferred> tasks;
  if (threads > 1) { top.grain = std::max<uint32_t>(4096u, n / (8u * threads)); top.deferred = &tasks; }
  top.nodes.reserve(threads > 1 ? 4096 : n);
  Box root_box;
  FlatBvh out;
  out.root_ref = top.Build(0, n, 0, root_box);
  out.depth = top.max_depth_seen;
  out.nodes = std::move(top.nodes);
  if (!tasks.empty()) {
    std::vector<std::vector<DevBvhNode>> sub_nodes(tasks.size());
    std::vector<int32_t> sub_root(tasks.size());
    std::vector<uint32_t> sub_depth(tasks.size());
    ParallelFor(tasks.size(), threads, [&](size_t k) {
      Builder b(s);
      b.nodes.reserve((tasks[k].last - tasks[k].first) / 2 + 16);
      Box box;
      sub_root[k] = b.Build(tasks[k].first, tasks[k].last, tasks[k].depth, box);
      sub_depth[k] = b.max_depth_seen;
      sub_nodes[k] = std::move(b.nodes);
    });
    size_t total = out.nodes.size();
    for (const auto& v : sub_nodes) total += v.size();
    out.nodes.reserve(total);
    for (size_t k = 0; k < tasks.size(); k++) {
      const int32_t base = static_cast<int32_t>(out.nodes.size());
      for (DevBvhNode nd : sub_nodes[k]) {
        if (nd.left >= 0) nd.left += base;
        if (nd.right >= 0) nd.right += base;
        out.nodes.push_back(nd);
      }
      const int32_t ref = sub_root[k] >= 0 ? sub_root[k] + base : sub_root[k];
      DevBvhNode& parent = out.nodes[tasks[k].parent];
      (tasks[k].is_left ? parent.left : parent.right) = ref;
      out.depth = std::max(out.depth, sub_depth[k]);
    }
  }
  out.prim_index = std::move(s.index);
  for (int c = 0; c < 3; c++) { out.bounds_min[c] = all.mn[c]; out.bounds_max[c] = all.mx[c]; }
  float rmin = 3.0e38f;
  for (const DevObject& o : objs)
    if ((o.kind & 0xffu) == 1) { out.has_spheres = true; rmin = std::min(rmin, std::fabs(o.radius)); }
  out.min_sphere_radius = out.has_spheres ? rmin : 0.0f;
  return out;
}

}  // namespace amber_bvh
