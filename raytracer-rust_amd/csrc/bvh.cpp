#include "bvh.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <limits>
#include <thread>

namespace rtamd {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kMedianDepth = 40;   // beyond this depth fall back to median splits (bounds the depth)

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<float>::infinity(); mx[k] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const float* lo, const float* hi) {
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], lo[k]); mx[k] = std::max(mx[k], hi[k]); }
    }
    void grow(const Box& b) { grow(b.mn, b.mx); }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box box;
    int32_t left = -1, right = -1;   // children (TmpNode indices) or -1 for a leaf
    uint32_t first = 0, count = 0;   // range of `order`
};

struct Builder {
    const std::vector<PrimBox>& prims;
    std::vector<uint32_t>& order;
    std::vector<float> cx, cy, cz;   // centroids
    uint32_t max_leaf;               // primitives per leaf of the binary tree (the leaf encoding allows 8)
    Builder(const std::vector<PrimBox>& p, std::vector<uint32_t>& o, uint32_t max_leaf_) : prims(p), order(o), max_leaf(max_leaf_) {
        size_t n = p.size();
        cx.resize(n); cy.resize(n); cz.resize(n);
        for (size_t i = 0; i < n; ++i) {
            cx[i] = 0.5f * (p[i].mn[0] + p[i].mx[0]);
            cy[i] = 0.5f * (p[i].mn[1] + p[i].mx[1]);
            cz[i] = 0.5f * (p[i].mn[2] + p[i].mx[2]);
        }
    }
    float centroid(uint32_t prim, int axis) const { return axis == 0 ? cx[prim] : (axis == 1 ? cy[prim] : cz[prim]); }

    // Splits [first, first+count) in place; returns the size of the left part (0 = make a leaf).
    uint32_t split(uint32_t first, uint32_t count, uint32_t depth) {
        if (count <= max_leaf) return 0;
        uint32_t* idx = order.data() + first;
        Box cb; cb.reset();
        for (uint32_t i = 0; i < count; ++i) { float c[3] = {cx[idx[i]], cy[idx[i]], cz[idx[i]]}; cb.grow(c, c); }
        int best_axis = -1; int best_bin = -1; float best_cost = std::numeric_limits<float>::infinity();
        if (depth < kMedianDepth) {
            for (int axis = 0; axis < 3; ++axis) {
                float lo = cb.mn[axis], hi = cb.mx[axis];
                if (!(hi > lo)) continue;
                float scale = (float)kBins / (hi - lo);
                Box bb[kBins]; uint32_t bc[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
                for (uint32_t i = 0; i < count; ++i) {
                    int b = (int)((centroid(idx[i], axis) - lo) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bb[b].grow(prims[idx[i]].mn, prims[idx[i]].mx); bc[b]++;
                }
                float la[kBins]; uint32_t lc[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int b = 0; b < kBins - 1; ++b) { acc.grow(bb[b]); c += bc[b]; la[b] = acc.half_area(); lc[b] = c; }
                acc.reset(); c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bb[b]); c += bc[b];
                    if (lc[b - 1] == 0 || c == 0) continue;
                    float cost = la[b - 1] * (float)lc[b - 1] + acc.half_area() * (float)c;
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b - 1; }
                }
            }
        }
        uint32_t nl = 0;
        if (best_axis >= 0) {
            float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
            float scale = (float)kBins / (hi - lo);
            int axis = best_axis, bin = best_bin;
            uint32_t* mid = std::partition(idx, idx + count, [&](uint32_t p) {
                int b = (int)((centroid(p, axis) - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= bin;
            });
            nl = (uint32_t)(mid - idx);
        }
        if (nl == 0 || nl == count) {
            // median split along the widest centroid axis (also the depth-bounding fallback)
            int axis = 0; float w = -1.f;
            for (int k = 0; k < 3; ++k) { float e = cb.mx[k] - cb.mn[k]; if (e > w) { w = e; axis = k; } }
            nl = count / 2;
            std::nth_element(idx, idx + nl, idx + count, [&](uint32_t a, uint32_t b) {
                float ca = centroid(a, axis), cbv = centroid(b, axis);
                return ca < cbv || (ca == cbv && a < b);
            });
        }
        return nl;
    }

    // Builds the subtree for a range into `nodes` (appending); returns its root index.
    int32_t build_range(std::vector<TmpNode>& nodes, uint32_t first, uint32_t count, uint32_t depth) {
        struct Item { int32_t node; uint32_t depth; };
        int32_t root = (int32_t)nodes.size();
        nodes.emplace_back();
        nodes[root].first = first; nodes[root].count = count;
        std::vector<Item> stack; stack.push_back({root, depth});
        while (!stack.empty()) {
            Item it = stack.back(); stack.pop_back();
            uint32_t f = nodes[it.node].first, c = nodes[it.node].count;
            Box b; b.reset();
            for (uint32_t i = 0; i < c; ++i) b.grow(prims[order[f + i]].mn, prims[order[f + i]].mx);
            nodes[it.node].box = b;
            uint32_t nl = split(f, c, it.depth);
            if (nl == 0) continue;
            int32_t l = (int32_t)nodes.size(); nodes.emplace_back();
            int32_t r = (int32_t)nodes.size(); nodes.emplace_back();
            nodes[l].first = f; nodes[l].count = nl;
            nodes[r].first = f + nl; nodes[r].count = c - nl;
            nodes[it.node].left = l; nodes[it.node].right = r;
            stack.push_back({l, it.depth + 1});
            stack.push_back({r, it.depth + 1});
        }
        return root;
    }
};

// Quantises the child boxes of one wide node (see bvh.h).
void encode_node(GpuNode& g, const Box& node_box, const Box* child_boxes, const int32_t* child_codes, int n_children) {
    std::memset(&g, 0, sizeof(g));
    uint32_t e8[3];
    double scale[3];
    for (int k = 0; k < 3; ++k) {
        g.p[k] = node_box.mn[k];
        double ext = (double)node_box.mx[k] - (double)g.p[k];
        if (!(ext > 0.0)) ext = 0.0;
        // smallest power of two s with 255 * s >= ext
        int e = (ext > 0.0 && std::isfinite(ext)) ? (int)std::ceil(std::log2(ext / 255.0)) : (ext > 0.0 ? 127 : -126);
        while (e < 127 && std::ldexp(255.0, e) < ext) e++;
        e = std::min(127, std::max(-126, e));
        e8[k] = (uint32_t)(e + 127);
        scale[k] = std::ldexp(1.0, e);
    }
    g.exps = e8[0] | (e8[1] << 8) | (e8[2] << 16);
    g.scale_y = (float)scale[1]; g.scale_z = (float)scale[2];
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        g.child[i] = i < n_children ? child_codes[i] : CHILD_EMPTY;
        for (int k = 0; k < 3; ++k) {
            uint32_t lo = 255, hi = 0;   // empty child: inverted (and masked by CHILD_EMPTY in the kernel)
            if (i < n_children) {
                double l = std::floor(((double)child_boxes[i].mn[k] - (double)g.p[k]) / scale[k]);
                double h = std::ceil(((double)child_boxes[i].mx[k] - (double)g.p[k]) / scale[k]);
                l = std::min(255.0, std::max(0.0, l));
                h = std::min(255.0, std::max(0.0, h));
                // containment is what matters: check the decoded planes themselves (q * 2^e is exact)
                while (l > 0.0 && (double)g.p[k] + l * scale[k] > (double)child_boxes[i].mn[k]) l -= 1.0;
                while (h < 255.0 && (double)g.p[k] + h * scale[k] < (double)child_boxes[i].mx[k]) h += 1.0;
                lo = (uint32_t)l;
                hi = (uint32_t)h;
            }
            qlo[k] |= lo << (8 * i);
            qhi[k] |= hi << (8 * i);
        }
    }
    g.qlo[0] = qlo[0]; g.qlo[1] = qlo[1]; g.qlo[2] = qlo[2];
    g.qhi_x = qhi[0]; g.qhi_y = qhi[1]; g.qhi_z = qhi[2];
}

}  // namespace

void build_bvh(const std::vector<PrimBox>& boxes, Bvh& out, int threads) {
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)boxes.size();
    out.nodes.clear(); out.order.resize(n); out.max_depth = 0;
    for (uint32_t i = 0; i < n; ++i) out.order[i] = i;
    if (n == 0) { out.build_ms = 0; return; }
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(boxes[i].mn[k]) || !std::isfinite(boxes[i].mx[k]) || boxes[i].mn[k] > boxes[i].mx[k]) {
                char msg[96];
                snprintf(msg, sizeof msg, "primitive box %u is not finite or is inverted", i);
                throw std::invalid_argument(msg);
            }

    // The binary tree is split down to single primitives; the collapse below decides which pairs stay one
    // leaf of two (MAX_LEAF_PRIMS) and which become two leaves in free slots of their parent.
    bool leaf_dp = true;
    uint32_t max_leaf = 1;
    if (const char* e = getenv("RTAMD_MAX_LEAF")) { max_leaf = (uint32_t)std::min(8, std::max(1, atoi(e))); leaf_dp = false; }   // tuning experiments only: fixed leaf size
    Builder B(boxes, out.order, max_leaf);
    std::vector<TmpNode> tmp;
    if (threads <= 0) {
        threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        // RTAMD_BVH_THREADS: cap for hosts whose cores are shared by several ranks (bench.py: cores / world size)
        if (const char* e = getenv("RTAMD_BVH_THREADS")) threads = std::max(1, std::min(threads, atoi(e)));
    }

    if (n < 65536 || threads == 1) {
        tmp.reserve(2 * (size_t)n / 2 + 16);
        B.build_range(tmp, 0, n, 0);
    } else {
        // split the top of the tree serially until there are enough subtrees, then build those in parallel
        struct Job { int32_t node; uint32_t depth; };
        tmp.emplace_back(); tmp[0].first = 0; tmp[0].count = n;
        std::vector<Job> frontier{{0, 0}};
        while ((int)frontier.size() < threads * 4) {
            size_t bi = 0;   // split the largest frontier range next
            for (size_t i = 1; i < frontier.size(); ++i) if (tmp[frontier[i].node].count > tmp[frontier[bi].node].count) bi = i;
            Job j = frontier[bi];
            uint32_t f = tmp[j.node].first, c = tmp[j.node].count;
            if (c <= 4096) break;
            uint32_t nl = B.split(f, c, j.depth);
            if (nl == 0) break;
            int32_t l = (int32_t)tmp.size(); tmp.emplace_back();
            int32_t r = (int32_t)tmp.size(); tmp.emplace_back();
            tmp[l].first = f; tmp[l].count = nl; tmp[r].first = f + nl; tmp[r].count = c - nl;
            tmp[j.node].left = l; tmp[j.node].right = r;
            frontier[bi] = {l, j.depth + 1};
            frontier.push_back({r, j.depth + 1});
        }
        std::vector<std::vector<TmpNode>> sub(frontier.size());
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= frontier.size()) break;
                B.build_range(sub[i], tmp[frontier[i].node].first, tmp[frontier[i].node].count, frontier[i].depth);
            }
        };
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back(work);
        for (auto& t : ts) t.join();
        // splice the subtrees in: sub[i][0] replaces the frontier node, the rest is appended
        for (size_t i = 0; i < frontier.size(); ++i) {
            int32_t base = (int32_t)tmp.size();
            auto fix = [&](int32_t c) { return c < 0 ? c : (c == 0 ? frontier[i].node : base + c - 1); };
            TmpNode rootn = sub[i][0];
            rootn.left = fix(rootn.left); rootn.right = fix(rootn.right);
            tmp[frontier[i].node] = rootn;
            for (size_t k = 1; k < sub[i].size(); ++k) {
                TmpNode t = sub[i][k];
                t.left = fix(t.left); t.right = fix(t.right);
                tmp.push_back(t);
            }
        }
        // boxes of the serially split top nodes (children first: process in reverse creation order)
        std::vector<int32_t> top;
        {
            std::vector<int32_t> st{0};
            std::vector<char> is_frontier(tmp.size(), 0);
            for (auto& j : frontier) is_frontier[j.node] = 1;
            while (!st.empty()) {
                int32_t x = st.back(); st.pop_back();
                if (is_frontier[x] || tmp[x].left < 0) continue;
                top.push_back(x); st.push_back(tmp[x].left); st.push_back(tmp[x].right);
            }
        }
        for (size_t k = top.size(); k-- > 0;) {
            TmpNode& t = tmp[top[k]];
            t.box = tmp[t.left].box; t.box.grow(tmp[t.right].box);
        }
    }

    // collapse the binary tree to 4-wide nodes and flatten depth-first.  Which binary nodes survive as
    // wide nodes is chosen by dynamic programming on the SAH cost (Ylitie et al. 2017, for 4 slots):
    // cost[x][k-1] = cheapest way to hand the subtree of x to its parent as at most k children, where a
    // child is a leaf (area * primitives * c_prim) or a wide node (area * c_node + its own four slots
    // split between its two binary children).  Children are created after their parents in tmp, so a
    // reverse sweep is a post-order.  A binary node of at most MAX_LEAF_PRIMS primitives may also be kept
    // as ONE leaf (area * primitives * c_prim) when that is cheaper than what its children cost in the slot(s)
    // it gets.  (Against the greedy rule "adopt the children of the child with the largest area" and fixed
    // leaves of two: 15 % fewer nodes, 1.5-5 % fewer node visits and 12 % fewer primitive tests per ray,
    // C4 +3 %, C5 +1.4 % rays/s.)
    auto leaf_code = [&](const TmpNode& t) { return ~(int32_t)((t.first << 3) | (t.count - 1)); };
    float c_node = 1.5f;   // cost of a node step in primitive steps (1, 1.5, 2, 3 measured: 1.5)
    const float c_prim = 1.0f;
    if (const char* e = getenv("RTAMD_CNODE")) c_node = (float)atof(e);   // tuning experiments only
    std::vector<uint8_t> as_leaf(tmp.size(), 0);             // an inner binary node that is better off as one leaf
    std::vector<std::array<float, 4>> cost(tmp.size());
    std::vector<std::array<uint8_t, 4>> part(tmp.size());   // [k-1]: 0 = x stays ONE child, i = left subtree gets i of the k slots
    std::vector<uint8_t> part_node(tmp.size(), 0);           // x as a wide node: its left subtree gets i of the 4 slots
    const bool greedy = getenv("RTAMD_COLLAPSE_GREEDY") != nullptr;   // tuning experiments only
    if (!greedy) {
        for (size_t xi = tmp.size(); xi-- > 0;) {
            const TmpNode& t = tmp[xi];
            const float area = t.box.half_area();
            if (t.left < 0) {
                for (int k = 0; k < 4; ++k) { cost[xi][k] = area * (float)t.count * c_prim; part[xi][k] = 0; }
                continue;
            }
            const auto &cl = cost[(size_t)t.left], &cr = cost[(size_t)t.right];
            float as_node = INFINITY;
            for (int i = 1; i <= 3; ++i) {
                const float c = cl[i - 1] + cr[4 - i - 1];
                if (i == 1 || c < as_node) { as_node = c; part_node[xi] = (uint8_t)i; }   // (i == 1 first: costs that overflowed to inf / NaN must still yield a valid split)
            }
            as_node += area * c_node;
            if (leaf_dp && t.count <= MAX_LEAF_PRIMS && area * (float)t.count * c_prim <= as_node) {
                as_node = area * (float)t.count * c_prim;
                as_leaf[xi] = 1;
            }
            cost[xi][0] = as_node; part[xi][0] = 0;
            for (int k = 2; k <= 4; ++k) {
                float best = as_node; uint8_t choice = 0;
                for (int i = 1; i < k; ++i) {
                    const float c = cl[i - 1] + cr[k - i - 1];
                    if (c < best) { best = c; choice = (uint8_t)i; }
                }
                cost[xi][k - 1] = best; part[xi][k - 1] = choice;
            }
        }
    }
    struct Wide { int32_t tmp; int32_t kids[4]; int n; };
    auto gather = [&](int32_t x) {
        Wide w; w.tmp = x; w.n = 0;
        if (tmp[x].left < 0) { w.kids[w.n++] = x; return w; }      // a lone leaf: one child
        if (greedy) {
            w.kids[w.n++] = tmp[x].left; w.kids[w.n++] = tmp[x].right;
            while (w.n < 4) {
                int best = -1; float area = -1.f;
                for (int i = 0; i < w.n; ++i)
                    if (tmp[w.kids[i]].left >= 0 && tmp[w.kids[i]].box.half_area() > area) { area = tmp[w.kids[i]].box.half_area(); best = i; }
                if (best < 0) break;
                int32_t c = w.kids[best];
                w.kids[best] = tmp[c].left; w.kids[w.n++] = tmp[c].right;
            }
            return w;
        }
        // hand out the four slots as the table says (explicit stack: at most 4 entries)
        struct Todo { int32_t node; int k; };
        Todo todo[8]; int nt = 0;
        todo[nt++] = {tmp[x].right, 4 - (int)part_node[(size_t)x]};
        todo[nt++] = {tmp[x].left, (int)part_node[(size_t)x]};
        while (nt > 0) {
            const Todo td = todo[--nt];
            const int i = tmp[td.node].left < 0 ? 0 : (int)part[(size_t)td.node][td.k - 1];
            if (i == 0) { w.kids[w.n++] = td.node; continue; }
            todo[nt++] = {tmp[td.node].right, td.k - i};
            todo[nt++] = {tmp[td.node].left, i};
        }
        return w;
    };
    struct Item { int32_t tmp; int32_t gpu; uint32_t depth; uint32_t pending; };
    std::vector<Item> st;
    out.nodes.clear();
    out.nodes.emplace_back();
    st.push_back({0, 0, 1, 1});   // pending starts at 1: the stack sentinel
    out.max_stack = 1;
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        Wide w = gather(it.tmp);
        out.max_depth = std::max(out.max_depth, it.depth);
        // while a child of this node is traversed, up to n-1 siblings wait on the stack
        const uint32_t pending = it.pending + (uint32_t)(w.n - 1);
        out.max_stack = std::max(out.max_stack, pending);
        Box cb[4]; int32_t codes[4];
        for (int i = 0; i < w.n; ++i) {
            const TmpNode& c = tmp[w.kids[i]];
            cb[i] = c.box;
            if (c.left >= 0 && !as_leaf[(size_t)w.kids[i]]) {
                codes[i] = (int32_t)out.nodes.size();
                out.nodes.emplace_back();
                st.push_back({w.kids[i], codes[i], it.depth + 1, pending});
            } else {
                codes[i] = leaf_code(c);
            }
        }
        GpuNode g;
        encode_node(g, tmp[it.tmp].box, cb, codes, w.n);
        out.nodes[(size_t)it.gpu] = g;
    }
    out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Checks what the closest-hit kernel relies on: every primitive is in exactly one leaf, every
// (decoded, quantised) child box contains all the boxes of its subtree, the stack bound holds.
// Throws std::runtime_error with the first violation.
namespace {
[[noreturn]] void raise(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw std::runtime_error(buf);
}
}  // namespace
void validate_bvh(const Bvh& bvh, const std::vector<PrimBox>& pb) {
    const uint64_t n = pb.size();
    if (n == 0) return;
    if (bvh.nodes.empty()) raise("no nodes for %llu primitives", (unsigned long long)n);
    if (bvh.order.size() != n) raise("order has %zu entries for %llu primitives", bvh.order.size(), (unsigned long long)n);
    std::vector<uint8_t> seen(n, 0);
    struct Item { int32_t node; uint32_t depth; uint32_t pending; };
    // returns the union of the subtree's primitive boxes; checks containment on the way up
    struct Ext { float mn[3], mx[3]; };
    std::vector<Ext> ext(bvh.nodes.size());
    std::vector<char> done(bvh.nodes.size(), 0);
    std::vector<int32_t> st{0};
    uint32_t worst_stack = 1;
    std::vector<uint32_t> pend(bvh.nodes.size(), 0);
    pend[0] = 1;
    while (!st.empty()) {
        int32_t x = st.back();
        const GpuNode& g = bvh.nodes[(size_t)x];
        int nchild = 0;
        for (int c = 0; c < 4; ++c) if (g.child[c] != CHILD_EMPTY) nchild++;
        if (!done[x]) {
            done[x] = 1;
            bool pushed = false;
            worst_stack = std::max(worst_stack, pend[x] + (uint32_t)(nchild - 1));
            for (int c = 0; c < 4; ++c)
                if (g.child[c] != CHILD_EMPTY && g.child[c] >= 0) {
                    if ((size_t)g.child[c] >= bvh.nodes.size()) raise("node %d: child index out of range", x);
                    pend[(size_t)g.child[c]] = pend[x] + (uint32_t)(nchild - 1);
                    st.push_back(g.child[c]); pushed = true;
                }
            if (pushed) continue;
        }
        st.pop_back();
        Ext e;
        for (int k = 0; k < 3; ++k) { e.mn[k] = INFINITY; e.mx[k] = -INFINITY; }
        const uint32_t ex[3] = {g.exps & 0xFFu, (g.exps >> 8) & 0xFFu, (g.exps >> 16) & 0xFFu};
        const uint32_t qlo[3] = {g.qlo[0], g.qlo[1], g.qlo[2]}, qhi[3] = {g.qhi_x, g.qhi_y, g.qhi_z};
        {
            uint32_t by = ex[1] << 23, bz = ex[2] << 23, gy, gz;
            memcpy(&gy, &g.scale_y, 4); memcpy(&gz, &g.scale_z, 4);
            if (gy != by || gz != bz) raise("node %d: float scales differ from the exponent bytes", x);
        }
        for (int c = 0; c < 4; ++c) {
            if (g.child[c] == CHILD_EMPTY) continue;
            Ext sub;
            for (int k = 0; k < 3; ++k) { sub.mn[k] = INFINITY; sub.mx[k] = -INFINITY; }
            if (g.child[c] >= 0) sub = ext[(size_t)g.child[c]];
            else {
                uint32_t code = (uint32_t)~g.child[c], first = code >> 3, count = (code & 7u) + 1u;
                for (uint32_t i = first; i < first + count; ++i) {
                    if (i >= n) raise("node %d: leaf range out of bounds", x);
                    uint32_t prim = bvh.order[i];
                    if (prim >= n || seen[prim]) raise("primitive %u referenced twice or out of range", prim);
                    seen[prim] = 1;
                    for (int k = 0; k < 3; ++k) { sub.mn[k] = std::min(sub.mn[k], pb[prim].mn[k]); sub.mx[k] = std::max(sub.mx[k], pb[prim].mx[k]); }
                }
            }
            for (int k = 0; k < 3; ++k) {
                uint32_t bits = ex[k] << 23;
                float scale;
                memcpy(&scale, &bits, 4);
                // decoded exactly (q * 2^e is exact, the sum is evaluated in double here)
                double lo = (double)g.p[k] + (double)((qlo[k] >> (8 * c)) & 0xFFu) * (double)scale;
                double hi = (double)g.p[k] + (double)((qhi[k] >> (8 * c)) & 0xFFu) * (double)scale;
                if (lo > (double)sub.mn[k] || hi < (double)sub.mx[k])
                    raise("node %d child %d (code %d) axis %d: decoded box [%g,%g] does not contain [%g,%g]", x, c, g.child[c], k, lo, hi, sub.mn[k], sub.mx[k]);
                e.mn[k] = std::min(e.mn[k], sub.mn[k]); e.mx[k] = std::max(e.mx[k], sub.mx[k]);
            }
        }
        ext[(size_t)x] = e;
    }
    for (uint64_t i = 0; i < n; ++i) if (!seen[i]) raise("primitive %llu is in no leaf", (unsigned long long)i);
    if (worst_stack > bvh.max_stack) raise("max_stack %u is below the real bound %u", bvh.max_stack, worst_stack);
}

}  // namespace rtamd
