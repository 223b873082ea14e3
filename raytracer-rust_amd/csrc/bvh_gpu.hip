// GPU BVH builder: the same 4-wide quantised nodes as bvh.cpp, built on the device in
// milliseconds for scenes that change every frame.  Linear BVH: 63-bit Morton codes of the primitive
// centroids, one radix sort (rocPRIM), the binary radix tree of Karras 2012 (one thread per internal
// node), boxes bottom-up (one thread per leaf, the second arrival at a node continues), then a
// level-by-level collapse into 4-wide nodes with the greedy largest-area rule of bvh.cpp.
// The tree is of lower quality than the host's binned-SAH tree (more nodes per ray): it is an option
// (rt_context_set_bvh_builder), not the default.  No reference counterpart: the reference has one
// AABB per mesh (/root/reference/src/raytracer/scene/mesh.rs:318-326).
#include <hip/hip_runtime.h>
#include <string.h>   // rocprim's texture_cache_iterator.hpp uses memset without including it
#include <cstring>
using std::memset;
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "bvh.h"

namespace rtamd {
namespace {

#define GPU_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); goto fail; } \
    } while (0)

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {   // 21 bits -> every third bit
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// key = 63-bit Morton code of the centroid (21 bits per axis); value = primitive index.  Equal keys are
// told apart by their sorted position (delta below), so the tree is well defined for any input.
__global__ __launch_bounds__(256) void k_lbvh_keys(const PrimBox* __restrict__ boxes, uint32_t n, float lx, float ly, float lz, float sx,
                                                    float sy, float sz, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PrimBox b = boxes[i];
    const float cx = (0.5f * (b.mn[0] + b.mx[0]) - lx) * sx, cy = (0.5f * (b.mn[1] + b.mx[1]) - ly) * sy, cz = (0.5f * (b.mn[2] + b.mx[2]) - lz) * sz;
    const float top = 2097151.f;
    const uint32_t qx = (uint32_t)fminf(fmaxf(cx, 0.f), top), qy = (uint32_t)fminf(fmaxf(cy, 0.f), top), qz = (uint32_t)fminf(fmaxf(cz, 0.f), top);
    keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
    vals[i] = i;
}

// Binary radix tree over the sorted keys.  Child encoding: >= 0 internal node, < 0: ~(leaf position).
struct Tree {
    int* left;        // [n-1]
    int* right;       // [n-1]
    int* parent;      // [n-1] parent of internal node (root: -1)
    int* leaf_parent; // [n]
    uint32_t* first;  // [n-1] range of sorted positions covered
    uint32_t* last;
    float* box;       // [n-1][6]
    uint32_t* flag;   // [n-1] arrivals
    float* cost;      // [n-1][4] SAH cost of handing the subtree to its parent as at most k+1 children (bvh.cpp)
    uint8_t* part;    // [n-1][4] 0 = the node stays ONE child, i = its left subtree gets i of the k+1 slots
    uint8_t* part_node; // [n-1] the node as a wide node: its left subtree gets i of the 4 slots
};

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = k[i] ^ k[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);     // equal codes: the sorted positions decide
}

__global__ __launch_bounds__(256) void k_lbvh_tree(const unsigned long long* __restrict__ keys, int n, Tree t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int s = lmax >> 1; s >= 1; s >>= 1)
        if (delta(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int tt = (l + 1) >> 1;; tt = (tt + 1) >> 1) {
        if (delta(keys, n, i, i + (s + tt) * d) > dnode) s += tt;
        if (tt <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int lc = (lo == gamma) ? ~gamma : gamma;
    const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    t.left[i] = lc; t.right[i] = rc;
    t.first[i] = (uint32_t)lo; t.last[i] = (uint32_t)hi;
    if (lc >= 0) t.parent[lc] = i; else t.leaf_parent[~lc] = i;
    if (rc >= 0) t.parent[rc] = i; else t.leaf_parent[~rc] = i;
    if (i == 0) t.parent[0] = -1;
}

__device__ __forceinline__ void load_box(const Tree& t, const PrimBox* __restrict__ boxes, const uint32_t* __restrict__ vals, int c, float b[6]) {
    if (c < 0) {
        const PrimBox p = boxes[vals[~c]];
        b[0] = p.mn[0]; b[1] = p.mn[1]; b[2] = p.mn[2]; b[3] = p.mx[0]; b[4] = p.mx[1]; b[5] = p.mx[2];
    } else {
        // written by another thread before its (fenced) arrival at our common parent; this thread has
        // passed an agent-scope fence since (k_lbvh_boxes), or reads in a later kernel (k_lbvh_collapse)
        for (int k = 0; k < 6; ++k) b[k] = t.box[6 * (size_t)c + k];
    }
}

__device__ __forceinline__ uint32_t kid_size(const Tree& t, int c) { return c < 0 ? 1u : t.last[c] - t.first[c] + 1u; }
__device__ __forceinline__ float box_half_area(const float b[6]) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
// cost of child c of a node for 1..4 slots: a leaf costs area * primitives whatever it is given
__device__ __forceinline__ void child_cost(const Tree& t, int c, const float b[6], float out[4]) {
    if (c < 0) {
        const float v = box_half_area(b);
        out[0] = out[1] = out[2] = out[3] = v;
    } else {
        for (int k = 0; k < 4; ++k) out[k] = t.cost[4 * (size_t)c + k];   // written before c's (fenced) arrival here
    }
}

__global__ __launch_bounds__(256) void k_lbvh_boxes(const uint32_t* __restrict__ vals, const PrimBox* __restrict__ boxes, int n, Tree t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = t.leaf_parent[i];
    // at most n-1 steps: every internal node is completed exactly once
    for (int guard = 0; guard < n && p >= 0; ++guard) {
        __threadfence();
        if (atomicAdd(&t.flag[p], 1u) == 0u) return;       // first arrival: the sibling subtree is not done yet
        __threadfence();
        float a[6], b[6];
        load_box(t, boxes, vals, t.left[p], a);
        load_box(t, boxes, vals, t.right[p], b);
        float cl[4], cr[4], m[6];
        child_cost(t, t.left[p], a, cl);
        child_cost(t, t.right[p], b, cr);
        for (int k = 0; k < 3; ++k) {
            m[k] = fminf(a[k], b[k]);
            m[3 + k] = fmaxf(a[3 + k], b[3 + k]);
            t.box[6 * (size_t)p + k] = m[k];
            t.box[6 * (size_t)p + 3 + k] = m[3 + k];
        }
        // the collapse tables of bvh.cpp (dynamic programming on the SAH cost, a node step = 1.5 primitive steps)
        float as_node = INFINITY;
        uint8_t pn = 1;
        for (int i = 1; i <= 3; ++i) {
            const float c = cl[i - 1] + cr[3 - i];
            if (c < as_node) { as_node = c; pn = (uint8_t)i; }
        }
        as_node += 1.5f * box_half_area(m);
        // at most MAX_LEAF_PRIMS primitives below: in ONE slot the subtree is a leaf, never a wide node
        const uint32_t psize = kid_size(t, p);
        if (psize <= MAX_LEAF_PRIMS) as_node = box_half_area(m) * (float)psize;
        t.part_node[p] = pn;
        t.cost[4 * (size_t)p] = as_node; t.part[4 * (size_t)p] = 0;
        for (int k = 2; k <= 4; ++k) {
            float best = as_node; uint8_t choice = 0;
            for (int i = 1; i < k; ++i) {
                const float c = cl[i - 1] + cr[k - i - 1];
                if (c < best) { best = c; choice = (uint8_t)i; }
            }
            t.cost[4 * (size_t)p + k - 1] = best; t.part[4 * (size_t)p + k - 1] = choice;
        }
        p = t.parent[p];
    }
}

struct Item { int node; uint32_t wide; };


// One thread per wide node of the current level.
__global__ __launch_bounds__(128) void k_lbvh_collapse(const uint32_t* __restrict__ vals, const PrimBox* __restrict__ boxes, Tree t,
                                                        const Item* __restrict__ items, uint32_t n_items, Item* __restrict__ next,
                                                        uint32_t* __restrict__ counters /* [0] next count, [1] wide count, [2] overflow */,
                                                        uint32_t max_wide, GpuNode* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const Item it = items[i];
    int kids[4];
    float kb[4][6];
    int nk = 0;
    {   // hand out the four slots as the tables say (bvh.cpp)
        int todo_node[8], todo_k[8], nt = 0;
        const int pn = min(3, max(1, (int)t.part_node[it.node]));
        todo_node[nt] = t.right[it.node]; todo_k[nt++] = 4 - pn;
        todo_node[nt] = t.left[it.node]; todo_k[nt++] = pn;
        while (nt > 0 && nk < 4) {
            const int c = todo_node[--nt], k = todo_k[nt];
            const int li = c < 0 ? 0 : (int)t.part[4 * (size_t)c + k - 1];   // a small subtree may stay one leaf or hand its primitives to free slots
            if (li <= 0 || li >= k || nt > 5) { kids[nk++] = c; continue; }   // one child (leaf or wide node)
            todo_node[nt] = t.right[c]; todo_k[nt++] = k - li;
            todo_node[nt] = t.left[c]; todo_k[nt++] = li;
        }
    }
    for (int k = 0; k < nk; ++k) load_box(t, boxes, vals, kids[k], kb[k]);
    // node box and scales (bvh.cpp encode_node)
    float nb[6];
    for (int k = 0; k < 6; ++k) nb[k] = t.box[6 * (size_t)it.node + k];
    GpuNode g;
    uint32_t e8[3];
    double scale[3];
    for (int k = 0; k < 3; ++k) {
        g.p[k] = nb[k];
        double ext = (double)nb[3 + k] - (double)nb[k];
        if (!(ext > 0.0)) ext = 0.0;
        int e = (ext > 0.0) ? (int)ceil(log2(ext / 255.0)) : -126;
        while (e < 127 && ldexp(255.0, e) < ext) e++;
        e = min(127, max(-126, e));
        e8[k] = (uint32_t)(e + 127);
        scale[k] = ldexp(1.0, e);
    }
    g.exps = e8[0] | (e8[1] << 8) | (e8[2] << 16);
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        int32_t code = CHILD_EMPTY;
        if (c < nk) {
            const int kid = kids[c];
            const uint32_t size = kid_size(t, kid);
            if (kid >= 0 && size > MAX_LEAF_PRIMS) {
                const uint32_t w = atomicAdd(&counters[1], 1u);
                if (w < max_wide) {
                    const uint32_t q = atomicAdd(&counters[0], 1u);
                    next[q] = Item{kid, w};
                    code = (int32_t)w;
                } else {
                    atomicAdd(&counters[2], 1u);      // cannot happen (max_wide = n); keeps the tree consistent if it did
                }
            } else {
                const uint32_t first = kid < 0 ? (uint32_t)~kid : t.first[kid];
                code = ~(int32_t)((first << 3) | (size - 1u));
            }
        }
        g.child[c] = code;
        for (int k = 0; k < 3; ++k) {
            uint32_t lo = 255, hi = 0;
            if (c < nk && code != CHILD_EMPTY) {
                double l = floor(((double)kb[c][k] - (double)g.p[k]) / scale[k]);
                double h = ceil(((double)kb[c][3 + k] - (double)g.p[k]) / scale[k]);
                l = fmin(255.0, fmax(0.0, l));
                h = fmin(255.0, fmax(0.0, h));
                // containment is what matters: check the decoded planes themselves (q * 2^e is exact)
                while (l > 0.0 && (double)g.p[k] + l * scale[k] > (double)kb[c][k]) l -= 1.0;
                while (h < 255.0 && (double)g.p[k] + h * scale[k] < (double)kb[c][3 + k]) h += 1.0;
                lo = (uint32_t)l;
                hi = (uint32_t)h;
            }
            qlo[k] |= lo << (8 * c);
            qhi[k] |= hi << (8 * c);
        }
    }
    g.qlo[0] = qlo[0]; g.qlo[1] = qlo[1]; g.qlo[2] = qlo[2];
    g.qhi_x = qhi[0]; g.qhi_y = qhi[1]; g.qhi_z = qhi[2];
    g.scale_y = (float)scale[1]; g.scale_z = (float)scale[2];
    nodes[it.wide] = g;
}

}  // namespace

// Builds `out` (same contract as build_bvh) on the current device.  Returns 0 or fills err.
// Small inputs go to the host builder.
int build_bvh_gpu(const std::vector<PrimBox>& boxes, Bvh& out, void* stream, std::string& err) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t n = (uint32_t)boxes.size();
    err.clear();
    if (n < 1024u) { build_bvh(boxes, out); return 0; }
    auto t0 = std::chrono::steady_clock::now();
    // centroid bounds on the host (one pass over data that is on the host anyway)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float c = 0.5f * (boxes[i].mn[k] + boxes[i].mx[k]);
            lo[k] = std::min(lo[k], c); hi[k] = std::max(hi[k], c);
        }
    float sc[3];
    for (int k = 0; k < 3; ++k) sc[k] = hi[k] > lo[k] ? 2097152.f / (hi[k] - lo[k]) : 0.f;

    PrimBox* d_boxes = nullptr;
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
    uint32_t *d_vals = nullptr, *d_vsorted = nullptr;
    void* d_temp = nullptr;
    Tree t{};
    Item* d_items[2] = {nullptr, nullptr};
    uint32_t* d_counters = nullptr;
    GpuNode* d_nodes = nullptr;
    size_t temp_bytes = 0;
    const uint32_t max_wide = n;
    uint32_t levels = 0, n_items = 1, wide_count = 1;
    const uint32_t blocks = (n + 255u) / 256u;

    GPU_TRY(hipMalloc((void**)&d_boxes, (size_t)n * sizeof(PrimBox)));
    GPU_TRY(hipMalloc((void**)&d_keys, (size_t)n * 8));
    GPU_TRY(hipMalloc((void**)&d_sorted, (size_t)n * 8));
    GPU_TRY(hipMalloc((void**)&d_vals, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&d_vsorted, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.left, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.right, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.parent, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.leaf_parent, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.first, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.last, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.box, (size_t)n * 24));
    GPU_TRY(hipMalloc((void**)&t.flag, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.cost, (size_t)n * 16));
    GPU_TRY(hipMalloc((void**)&t.part, (size_t)n * 4));
    GPU_TRY(hipMalloc((void**)&t.part_node, (size_t)n));
    GPU_TRY(hipMalloc((void**)&d_items[0], (size_t)n * sizeof(Item)));
    GPU_TRY(hipMalloc((void**)&d_items[1], (size_t)n * sizeof(Item)));
    GPU_TRY(hipMalloc((void**)&d_counters, 16));
    GPU_TRY(hipMalloc((void**)&d_nodes, (size_t)max_wide * sizeof(GpuNode)));
    GPU_TRY(hipMemcpyAsync(d_boxes, boxes.data(), (size_t)n * sizeof(PrimBox), hipMemcpyHostToDevice, st));
    GPU_TRY(hipMemsetAsync(t.flag, 0, (size_t)n * 4, st));
    GPU_TRY(hipMemsetAsync(t.parent, 0xFF, (size_t)n * 4, st));
    GPU_TRY(hipMemsetAsync(t.leaf_parent, 0xFF, (size_t)n * 4, st));

    hipLaunchKernelGGL(k_lbvh_keys, dim3(blocks), dim3(256), 0, st, d_boxes, n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], d_keys, d_vals);
    GPU_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_sorted, d_vals, d_vsorted, (size_t)n, 0u, 63u, st));
    GPU_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)));
    GPU_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_sorted, d_vals, d_vsorted, (size_t)n, 0u, 63u, st));
    hipLaunchKernelGGL(k_lbvh_tree, dim3(blocks), dim3(256), 0, st, d_sorted, (int)n, t);
    hipLaunchKernelGGL(k_lbvh_boxes, dim3(blocks), dim3(256), 0, st, d_vsorted, d_boxes, (int)n, t);
    GPU_TRY(hipGetLastError());

    // collapse, one level per launch (the root, binary node 0, is wide node 0)
    {
        const Item root{0, 0u};
        const uint32_t init[4] = {0u, 1u, 0u, 0u};
        GPU_TRY(hipMemcpyAsync(d_items[0], &root, sizeof(root), hipMemcpyHostToDevice, st));
        GPU_TRY(hipMemcpyAsync(d_counters, init, sizeof(init), hipMemcpyHostToDevice, st));
        GPU_TRY(hipStreamSynchronize(st));
    }
    while (n_items > 0) {
        if (++levels > 4096u) { err = "GPU BVH: collapse does not terminate"; goto fail; }
        hipLaunchKernelGGL(k_lbvh_collapse, dim3((n_items + 127u) / 128u), dim3(128), 0, st, d_vsorted, d_boxes, t, d_items[(levels - 1) & 1],
                           n_items, d_items[levels & 1], d_counters, max_wide, d_nodes);
        uint32_t c[4];
        GPU_TRY(hipMemcpyAsync(c, d_counters, sizeof(c), hipMemcpyDeviceToHost, st));
        GPU_TRY(hipStreamSynchronize(st));
        if (c[2]) { err = "GPU BVH: node pool overflow"; goto fail; }
        n_items = c[0]; wide_count = c[1];
        if (n_items > n) { err = "GPU BVH: queue overflow"; goto fail; }
        GPU_TRY(hipMemsetAsync(d_counters, 0, 4, st));     // next-level count
    }
    out.nodes.resize(wide_count);
    GPU_TRY(hipMemcpyAsync(out.nodes.data(), d_nodes, (size_t)wide_count * sizeof(GpuNode), hipMemcpyDeviceToHost, st));
    out.order.resize(n);
    GPU_TRY(hipMemcpyAsync(out.order.data(), d_vsorted, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    GPU_TRY(hipStreamSynchronize(st));
    out.max_depth = levels;
    out.max_stack = 1u + 3u * levels;     // up to three siblings wait per level, plus the sentinel
    out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    err.clear();
fail:
    (void)hipFree(d_boxes); (void)hipFree(d_keys); (void)hipFree(d_sorted); (void)hipFree(d_vals); (void)hipFree(d_vsorted); (void)hipFree(d_temp);
    (void)hipFree(t.left); (void)hipFree(t.right); (void)hipFree(t.parent); (void)hipFree(t.leaf_parent);
    (void)hipFree(t.first); (void)hipFree(t.last); (void)hipFree(t.box); (void)hipFree(t.flag); (void)hipFree(t.cost); (void)hipFree(t.part); (void)hipFree(t.part_node);
    (void)hipFree(d_items[0]); (void)hipFree(d_items[1]); (void)hipFree(d_counters); (void)hipFree(d_nodes);
    return err.empty() ? 0 : 1;
}

}  // namespace rtamd
