// Host-side BVH builder (binned SAH, collapsed to 4-wide nodes with quantised child boxes)
// producing the flattened layout the closest-hit kernel traverses.  Replaces the reference's single
// per-mesh AABB cull + brute-force triangle loop
// (/root/reference/src/raytracer/scene/intersections.rs:142-186) and the linear
// object scan (/root/reference/src/raytracer/scene/scene.rs:35-52).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rtamd {

struct PrimBox {
    float mn[3], mx[3];
};

// 64-byte 4-wide node with 8-bit quantised child boxes (4 x 16-byte words):
//   w0 = (p.x, p.y, p.z, exps)     p = min corner of the node's box; exps = ex | ey << 8 | ez << 16,
//                                  biased exponent bytes: scale_axis = bits_to_float(e << 23)
//   w1 = (child0..child3)          child >= 0: inner node index.  child < 0 and != CHILD_EMPTY: leaf,
//                                  ~child = (first << 3) | (count - 1), `first` indexes the
//                                  leaf-ordered primitive array.
//   w2 = (qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4])   one byte per child, child i in bits 8i..8i+7
//   w3 = (qhi.y[4], qhi.z[4], scale_y, scale_z)   the y and z scales again, as floats (the kernel reads
//                                  them from here and takes scale_x from exps with one shift)
// child box = [p + qlo * scale, p + qhi * scale] (qlo rounded down, qhi rounded up: conservative).
// The kernel bounds its own rounding error per node, so no further padding is needed.
constexpr int32_t CHILD_EMPTY = (int32_t)0x80000000;
constexpr uint32_t MAX_LEAF_PRIMS = 2;   // primitives per leaf both builders produce (the leaf code allows 8)
struct GpuNode {
    float p[3];
    uint32_t exps;
    int32_t child[4];
    uint32_t qlo[3], qhi_x;
    uint32_t qhi_y, qhi_z;
    float scale_y, scale_z;
};
static_assert(sizeof(GpuNode) == 64, "node must be 64 bytes");

struct Bvh {
    std::vector<GpuNode> nodes;    // nodes[0] is the root (always an inner node when n_prims > 0)
    std::vector<uint32_t> order;   // order[k] = input index of the k-th primitive in leaf order
    uint32_t max_depth = 0;        // inner nodes on the longest root->leaf path
    uint32_t max_stack = 0;        // traversal-stack entries a lane can need (incl. the sentinel)
    double build_ms = 0;
};

// Boxes must already be conservative for the f32 data the kernel tests.
void build_bvh(const std::vector<PrimBox>& boxes, Bvh& out, int threads = 0);

// Throws std::runtime_error unless the tree is one the closest-hit kernel can traverse correctly
// (see bvh.cpp).  Used by rt_host_bvh_validate and by the GPU builder's test entry.
void validate_bvh(const Bvh& bvh, const std::vector<PrimBox>& boxes);

// GPU builder (bvh_gpu.hip): same contract as build_bvh, on the current device; stream is a hipStream_t.
// Returns 0 or fills err.
int build_bvh_gpu(const std::vector<PrimBox>& boxes, Bvh& out, void* stream, std::string& err);

}  // namespace rtamd
