// ref_bvh_dump.hip -- host-only driver (tests/test_reference_bvh_build.py): runs the PRODUCT's builder of the reference's BVH
// (amber_amd/csrc/hip/ref_bvh_build.h) on a flattened scene read from a file and prints what the oracle prints about the tree
// it builds from the same objects: node / leaf counts, depth, a digest of the pre-order walk (tag, box, leaf range -- the
// definition of oracle_scene_bvh_digest), and writes the object order.  No GPU call.
//   ref_bvh_dump <objects.bin: AmberFlatObject records> <order_out.bin: uint32 per object>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../amber_amd/csrc/hip/ref_bvh_build.h"

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: ref_bvh_dump objects.bin order_out.bin\n"); return 2; }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<AmberFlatObject> objects(static_cast<size_t>(bytes) / sizeof(AmberFlatObject));
  if (std::fread(objects.data(), sizeof(AmberFlatObject), objects.size(), f) != objects.size()) { std::fprintf(stderr, "short read\n"); return 2; }
  std::fclose(f);

  const amber_refbvh::Tree tree = amber_refbvh::Build(objects.data(), static_cast<uint32_t>(objects.size()));
  uint64_t h = 14695981039346656037ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  std::vector<const amber_refbvh::Node*> stack{tree.root.get()};
  while (!stack.empty()) {
    const amber_refbvh::Node* n = stack.back(); stack.pop_back();
    const bool leaf = !n->left;
    const uint32_t tag = leaf ? 0u : 1u;
    const float bb[6] = {n->bb.mn[0], n->bb.mn[1], n->bb.mn[2], n->bb.mx[0], n->bb.mx[1], n->bb.mx[2]};
    mix(&tag, 4); mix(bb, 24);
    if (leaf) { const uint32_t range[2] = {n->first, n->count}; mix(range, 8); }
    else { stack.push_back(n->right.get()); stack.push_back(n->left.get()); }
  }
  // the flattened image the device walks: every reference in range, every object in exactly one leaf
  const amber_refbvh::FlatTree flat = amber_refbvh::Flatten(tree, objects.data());
  std::vector<int> seen(objects.size(), 0);
  for (const auto& lf : flat.leaves)
    for (uint32_t k = lf.first; k < lf.first + (lf.count & amber_refbvh::kLeafCountMask); k++) { if (k >= objects.size()) { std::fprintf(stderr, "leaf range\n"); return 1; } seen[tree.order[k]]++; }
  for (int s : seen) if (s != 1) { std::fprintf(stderr, "an object sits in %d leaves\n", s); return 1; }
  for (const auto& nd : flat.nodes)
    for (int32_t r : {nd.left, nd.right})
      if (r >= static_cast<int32_t>(flat.nodes.size()) || -(r + 1) >= static_cast<int32_t>(flat.leaves.size())) { std::fprintf(stderr, "child reference\n"); return 1; }
  if (flat.nodes.size() != tree.n_inner || flat.leaves.size() != tree.n_leaves) { std::fprintf(stderr, "flattened counts\n"); return 1; }

  f = std::fopen(argv[2], "wb");
  if (!f) { std::perror(argv[2]); return 2; }
  std::fwrite(tree.order.data(), sizeof(uint32_t), tree.order.size(), f);
  std::fclose(f);
  std::printf("nodes %u leaves %u depth %u largest_leaf %u digest %llu\n", tree.n_inner + tree.n_leaves, tree.n_leaves, tree.depth, tree.largest_leaf, static_cast<unsigned long long>(h));
  return 0;
}
