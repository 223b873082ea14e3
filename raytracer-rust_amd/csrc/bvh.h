// Host-side BVH builder (binned SAH) producing the flattened two-child node
// layout the closest-hit kernel traverses.  Replaces the reference's single
// per-mesh AABB cull + brute-force triangle loop
// (/root/reference/src/raytracer/scene/intersections.rs:142-186) and the linear
// object scan (/root/reference/src/raytracer/scene/scene.rs:35-52).
#pragma once
#include <cstdint>
#include <vector>

namespace rtamd {

struct PrimBox {
    float mn[3], mx[3];
};

// 64-byte node, 4 x float4:
//   n0 = (c0.min.x, c0.max.x, c0.min.y, c0.max.y)
//   n1 = (c1.min.x, c1.max.x, c1.min.y, c1.max.y)
//   n2 = (c0.min.z, c0.max.z, c1.min.z, c1.max.z)
//   n3 = (bits(child0), bits(child1), 0, 0)
// child >= 0: inner node index.  child < 0: leaf, ~child = (first << 3) | (count - 1),
// `first` indexes the leaf-ordered primitive array.  An absent child has an
// inverted box (min = +inf, max = -inf) that no ray can hit.
struct GpuNode {
    float n0[4], n1[4], n2[4];
    int32_t child0, child1, pad0, pad1;
};
static_assert(sizeof(GpuNode) == 64, "node must be 64 bytes");

struct Bvh {
    std::vector<GpuNode> nodes;    // nodes[0] is the root (always an inner node when n_prims > 0)
    std::vector<uint32_t> order;   // order[k] = input index of the k-th primitive in leaf order
    uint32_t max_depth = 0;        // number of inner nodes on the longest root->leaf path
    double build_ms = 0;
};

// Boxes must already be conservative for the f32 data the kernel tests; the
// builder pads every node box by a few ulps on top (see bvh.cpp).
void build_bvh(const std::vector<PrimBox>& boxes, Bvh& out, int threads = 0);

}  // namespace rtamd
