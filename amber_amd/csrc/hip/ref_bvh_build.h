// ref_bvh_build.h -- the reference's own BVH, node for node (engine REFERENCE_BVH).
//
// The other engines return the closest hit of the reference's List acceleration (acceleration_list.h:51-68).  The reference's
// command line casts through its BVH (cornel_box.cc:198, application.cc:74-87), which is a slightly different function of the
// ray: a distance tie goes to the object the traversal meets first, and a hit is lost when the near child's hit lies in front
// of the far child's box although an object of the far child reaches out of that box by a rounding error
// (acceleration_bvh.h:386-391).  On the 1M-sphere scene that is 117 of 2 073 600 pixels.  A user who needs every ray to get the hit the
// command line's BVH gives it asks for AMBER_ENGINE_REFERENCE_BVH: this file builds the tree acceleration_bvh.h:134-312 builds -- same
// topology, same boxes, same object order inside the leaves -- and ClosestHitReferenceBvh (dev_closest_hit.h) walks it in the order
// acceleration_bvh.h:340-403 does.
//
// What has to be reproduced, and how:
//   * CreateNode (:158-180): FindSplit, leaf iff best cost > N * kIntersectionCost, children over [first, middle) / [middle, last)
//     with the boxes found for the split.
//   * FindSplit (:197-241): FindSplitAxis for x, y, z in that order -- each SORTS the range -- then one more sort on the chosen
//     axis.  The order of objects with equal centres after a std::sort is a property of the sort algorithm, and the leaf scan
//     returns the first of two equally distant hits, so the sequence of sorts is run as the reference runs it, with std::sort,
//     on 16-byte keys {centre, index}: std::sort's moves depend on the comparator's answers only, not on the element type.
//     (Built with the same C++ library as the reference, the order is the reference's; the oracle is built the same way.)
//   * FindSplitAxis (:243-296): n_splits = min(15, floor(log2 N)) planes spaced evenly over the node's BOX extent, lower_bound
//     on "centre < plane", SAH cost 2 * 2 + (nl * A(l) + nr * A(r)) / A(parent); strict < keeps the first best plane.
//   * boxes and centres of the primitives: primitive_triangle.cc:70-95, primitive_sphere.cc:62-73, primitive_disk.cc:75-92,
//     primitive_cylinder.cc:73-90; AABB union, Empty and SurfaceArea: aabb.h:88-171.
// All of it in binary32 without contraction (-ffp-contract=off, Makefile), as the reference's -O2 x86-64 build evaluates it.
//
// Subtrees work on disjoint ranges, so the two recursive calls of a large node (the first six levels) run on two threads; the
// result does not depend on that.  1M objects: 1.5-2 s on the GPU box's 16 cores, of which the root's four sorts of the whole
// scene are a second (the reference: 26.6 s).
#pragma once

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <future>
#include <memory>
#include <vector>

#include "../../../include/amber_hip.h"
#include "pt_device.h"

namespace amber_refbvh {

struct Box { float mn[3], mx[3]; };

inline Box EmptyBox() { return Box{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; }      // aabb.h:88-96
inline Box Union(const Box& a, const Box& b) {                                                          // aabb.h:117-132
  Box r;
  for (int c = 0; c < 3; c++) { r.mn[c] = std::min(a.mn[c], b.mn[c]); r.mx[c] = std::max(a.mx[c], b.mx[c]); }
  return r;
}
inline float SurfaceArea(const Box& b) {                                                                // aabb.h:164-171
  const float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
  return 2 * (sx * sy + sy * sz + sz * sx);
}
inline Box DiskBox(const float* center, const float* normal, float radius) {                            // primitive_disk.cc:81-92
  Box r;
  for (int c = 0; c < 3; c++) {
    const float factor = std::sqrt(1 - normal[c] * normal[c]);
    r.mn[c] = center[c] - radius * factor; r.mx[c] = center[c] + radius * factor;
  }
  return r;
}
inline Box PrimitiveBox(const AmberFlatObject& o) {
  const float* p = o.p;
  Box r;
  switch (o.kind) {
    case AMBER_PRIM_TRIANGLE:                                                                           // primitive_triangle.cc:80-95
      for (int c = 0; c < 3; c++) { r.mn[c] = std::min({p[c], p[3 + c], p[6 + c]}); r.mx[c] = std::max({p[c], p[3 + c], p[6 + c]}); }
      return r;
    case AMBER_PRIM_SPHERE:                                                                             // primitive_sphere.cc:69-73
      for (int c = 0; c < 3; c++) { r.mn[c] = p[c] - p[3]; r.mx[c] = p[c] + p[3]; }
      return r;
    case AMBER_PRIM_DISK:
      return DiskBox(p, p + 3, p[6]);
    default: {                                                                                          // primitive_cylinder.cc:73-84
      float top[3];
      for (int c = 0; c < 3; c++) top[c] = p[c] + p[7] * p[3 + c];
      return Union(Union(EmptyBox(), DiskBox(p, p + 3, p[6])), DiskBox(top, p + 3, p[6]));
    }
  }
}

struct Key { float c[3]; uint32_t index; };      // Primitive::Center() and the object's place in the scene

inline Key PrimitiveKey(const AmberFlatObject& o, uint32_t index) {
  const float* p = o.p;
  Key k; k.index = index;
  switch (o.kind) {
    case AMBER_PRIM_TRIANGLE:                                                                           // primitive_triangle.cc:70-78
      for (int c = 0; c < 3; c++) k.c[c] = (p[c] + p[3 + c] + p[6 + c]) / 3;
      break;
    case AMBER_PRIM_CYLINDER:                                                                           // primitive_cylinder.cc:86-90
      for (int c = 0; c < 3; c++) k.c[c] = p[c] + p[7] / 2 * p[3 + c];
      break;
    default:                                                                                            // sphere, disk: the centre
      for (int c = 0; c < 3; c++) k.c[c] = p[c];
  }
  return k;
}

struct Node {                                    // BVH::Node, acceleration_bvh.h:96-120
  std::unique_ptr<Node> left, right;
  uint32_t first = 0, count = 0;                 // a leaf: objects [first, first + count) of the sorted order
  Box bb;
};

constexpr uint32_t kMaxDepth = 2048;             // levels this build follows the reference's recursion (config 3: 21, the 1M-triangle terrain: 24)

struct Tree {
  bool too_deep = false;                         // the reference's recursion goes beyond kMaxDepth levels on this scene: no tree (root is not the reference's)
  std::unique_ptr<Node> root;
  std::vector<uint32_t> order;                   // objects_ after the build: position -> scene index
  uint32_t n_inner = 0, n_leaves = 0, depth = 0; // depth: edges on the longest root-to-leaf walk
  uint32_t largest_leaf = 0;
};

namespace detail {

using It = std::vector<Key>::iterator;

struct Split { float cost = FLT_MAX; It middle; Box bl = EmptyBox(), br = EmptyBox(); };                // :120-128, :403-408

inline Box RangeBox(const std::vector<Box>& boxes, It first, It last) {                                 // :182-194 (min / max: the order is immaterial)
  Box bb = EmptyBox();
  for (It i = first; i != last; ++i) bb = Union(bb, boxes[i->index]);
  return bb;
}

template <int kAxis>
Split FindSplitAxis(const std::vector<Box>& boxes, It first, It last, const Box& bb) {                  // :243-296
  std::sort(first, last, [](const Key& a, const Key& b) { return a.c[kAxis] < b.c[kAxis]; });
  const std::size_t n_splits = std::min<std::size_t>(15.0f, std::log2(std::distance(first, last)));
  Split split;
  for (std::size_t i = 0; i < n_splits; i++) {
    const float split_point = bb.mn[kAxis] + (bb.mx[kAxis] - bb.mn[kAxis]) * (i + 1) / (n_splits + 1);
    const It middle = std::lower_bound(first, last, split_point, [](const Key& k, const float sp) { return k.c[kAxis] < sp; });
    const Box bl = RangeBox(boxes, first, middle), br = RangeBox(boxes, middle, last);
    const std::size_t nl = std::distance(first, middle), nr = std::distance(middle, last);
    const float cost = 2 * 2.0f + (nl * SurfaceArea(bl) + nr * SurfaceArea(br)) / SurfaceArea(bb) * 1.0f;   // :298-312
    if (cost < split.cost) { split.cost = cost; split.middle = middle; split.bl = bl; split.br = br; }
  }
  return split;
}

inline Split FindSplit(const std::vector<Box>& boxes, It first, It last, const Box& bb) {               // :197-241
  const Split sx = FindSplitAxis<0>(boxes, first, last, bb);
  const Split sy = FindSplitAxis<1>(boxes, first, last, bb);
  const Split sz = FindSplitAxis<2>(boxes, first, last, bb);
  // the iterators of sx / sy point into a range that has been sorted again since: the reference returns them as they are, and
  // after its final sort on the chosen axis they delimit the same two sets (the sort on one axis is a function of the set up to
  // the order of equal keys, and lower_bound's partition point does not depend on that order)
  if (sx.cost < sy.cost && sx.cost < sz.cost) {
    std::sort(first, last, [](const Key& a, const Key& b) { return a.c[0] < b.c[0]; });
    return sx;
  } else if (sy.cost < sx.cost) {
    std::sort(first, last, [](const Key& a, const Key& b) { return a.c[1] < b.c[1]; });
    return sy;
  } else {
    std::sort(first, last, [](const Key& a, const Key& b) { return a.c[2] < b.c[2]; });
    return sz;
  }
}

struct Builder {
  const std::vector<Box>& boxes;
  It begin;
  std::atomic<bool>* too_deep;

  std::unique_ptr<Node> Create(It first, It last, const Box& bb, uint32_t depth) const {                // :158-180
    const Split split = FindSplit(boxes, first, last, bb);
    auto node = std::make_unique<Node>();
    node->bb = bb;
    // The reference recurses as deep as its splits lead it (objects spaced geometrically along a line peel off a few per level) and would
    // overflow its call stack; this build stops at kMaxDepth levels and reports the scene as refused (Tree::too_deep) rather than guess a tree.
    if (depth >= kMaxDepth && !(split.cost > std::distance(first, last) * 1.0f)) too_deep->store(true);
    if (split.cost > std::distance(first, last) * 1.0f || depth >= kMaxDepth) {
      node->first = static_cast<uint32_t>(first - begin);
      node->count = static_cast<uint32_t>(last - first);
    } else if (depth < 6 && std::distance(first, last) > 5000) {
      auto left = std::async(std::launch::async, [&] { return Create(first, split.middle, split.bl, depth + 1); });
      node->right = Create(split.middle, last, split.br, depth + 1);
      node->left = left.get();
    } else {
      node->left = Create(first, split.middle, split.bl, depth + 1);
      node->right = Create(split.middle, last, split.br, depth + 1);
    }
    return node;
  }
};

inline void Measure(const Node* n, uint32_t depth, Tree& t) {
  if (!n->left) {
    t.n_leaves++; t.depth = std::max(t.depth, depth); t.largest_leaf = std::max(t.largest_leaf, n->count);
    return;
  }
  t.n_inner++;
  Measure(n->left.get(), depth + 1, t);
  Measure(n->right.get(), depth + 1, t);
}

}  // namespace detail

// The build orders objects by their centres with std::sort, whose behaviour is undefined when the comparison is not a strict weak order: a NaN
// centre.  (The reference has no such check; amber_hip_pt_create refuses the scene for this engine instead of inheriting the undefined behaviour.)
inline bool CentresAreOrdered(const AmberFlatObject* objects, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    const Key k = PrimitiveKey(objects[i], i);
    for (int c = 0; c < 3; c++) if (std::isnan(k.c[c])) return false;
  }
  return true;
}

inline Tree Build(const AmberFlatObject* objects, uint32_t n) {                                         // BVH::BVH, BuildBVH :134-156
  std::vector<Box> boxes(n);
  std::vector<Key> keys(n);
  for (uint32_t i = 0; i < n; i++) { boxes[i] = PrimitiveBox(objects[i]); keys[i] = PrimitiveKey(objects[i], i); }
  std::atomic<bool> too_deep{false};
  const detail::Builder b{boxes, keys.begin(), &too_deep};
  Tree t;
  t.root = b.Create(keys.begin(), keys.end(), detail::RangeBox(boxes, keys.begin(), keys.end()), 0);
  t.order.resize(n);
  for (uint32_t i = 0; i < n; i++) t.order[i] = keys[i].index;
  detail::Measure(t.root.get(), 0, t);
  t.too_deep = too_deep.load();
  return t;
}

// ---- the device image: 64-byte inner nodes holding BOTH children's boxes (the parent tests them, :357-372), a table of leaves
using FlatNode = amber_dev::DevRefNode;
using FlatLeaf = amber_dev::DevRefLeaf;
struct FlatTree {
  std::vector<FlatNode> nodes;
  std::vector<FlatLeaf> leaves;
  int32_t root = 0;                              // >= 0: nodes[root]; < 0: leaves[-(root + 1)] (a scene the reference keeps in one leaf)
};

// A leaf whose objects are all spheres (all triangles) says so in the two top bits of its count: the device then scans the 16-byte sphere
// (48-byte triangle) records of engine BVH's leaf arrays instead of the 64-byte object records -- the same operands, the same order.
constexpr uint32_t kLeafAllSpheres = 0x80000000u, kLeafAllTriangles = 0x40000000u, kLeafCountMask = 0x3fffffffu;

inline int32_t Flatten(const Node* n, const Tree& t, const AmberFlatObject* objects, FlatTree& out) {
  if (!n->left) {
    bool spheres = true, triangles = true;
    for (uint32_t k = n->first; k < n->first + n->count; k++) {
      spheres = spheres && objects[t.order[k]].kind == AMBER_PRIM_SPHERE;
      triangles = triangles && objects[t.order[k]].kind == AMBER_PRIM_TRIANGLE;
    }
    out.leaves.push_back(FlatLeaf{n->first, n->count | (spheres ? kLeafAllSpheres : 0u) | (triangles ? kLeafAllTriangles : 0u)});
    return -static_cast<int32_t>(out.leaves.size());
  }
  const size_t at = out.nodes.size();
  out.nodes.emplace_back();
  const int32_t l = Flatten(n->left.get(), t, objects, out), r = Flatten(n->right.get(), t, objects, out);
  FlatNode& f = out.nodes[at];
  for (int c = 0; c < 3; c++) {
    f.lmin[c] = n->left->bb.mn[c]; f.lmax[c] = n->left->bb.mx[c];
    f.rmin[c] = n->right->bb.mn[c]; f.rmax[c] = n->right->bb.mx[c];
  }
  f.left = l; f.right = r; f.pad[0] = f.pad[1] = 0;
  return static_cast<int32_t>(at);
}
inline FlatTree Flatten(const Tree& t, const AmberFlatObject* objects) {
  FlatTree out;
  out.nodes.reserve(t.n_inner); out.leaves.reserve(t.n_leaves);
  out.root = Flatten(t.root.get(), t, objects, out);
  return out;
}

}  // namespace amber_refbvh
