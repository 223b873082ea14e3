// C-ABI of the render path (include/rtamd.h): context, scene upload (f64 -> device
// layout + BVH), the wavefront loop and the statistics.  No torch types, no exceptions
// across the boundary.
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/rtamd.h"
#include "bvh.h"
#include "rt_kernels.h"

using namespace rtamd;

static_assert(RT_MAX_ITERATIONS == CTL_MAX_IT, "iteration table sizes must agree");

static thread_local std::string g_global_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct rt_context {
    int device = 0;
    LaunchCfg cfg;
    std::string error;
    hipStream_t own_stream = nullptr;

    // scene
    bool have_scene = false;
    DevScene sc{};
    std::vector<void*> scene_allocs;
    int bvh_builder = 0;      // 0: host binned SAH (bvh.cpp), 1: GPU LBVH (bvh_gpu.hip)
    uint32_t bvh_nodes = 0, bvh_depth = 0, bvh_prims = 0;
    double bvh_build_ms = 0;

    // wavefront state (grown on demand, reused across calls)
    uint32_t capacity = 0;      // rays per queue
    uint32_t path_capacity = 0; // radiance slots
    RayQueue q[2]{};
    float2* hits = nullptr;    // (t, code) per ray
    float4* radiance = nullptr;
    uint32_t* ctl = nullptr;
    unsigned long long* stats_dev = nullptr;   // CTL_MAX_IT + 3 words
    uint64_t paths_total = 0;
    uint64_t rays0_total = 0;   // primary rays actually traced (iteration 0)

    // timing of closest-hit launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double hit_ms = 0;
    uint32_t hit_launches = 0;
};

namespace {

int fail(rt_context* ctx, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_global_error = buf;
    return 1;
}

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <class T>
int upload(rt_context* ctx, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    ctx->scene_allocs.push_back(p);
    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T*>(p);
    return 0;
}

void free_scene(rt_context* ctx) {
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->have_scene = false;
}

void free_state(rt_context* ctx) {
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(ctx->q[k].a); (void)hipFree(ctx->q[k].b); (void)hipFree(ctx->q[k].c);
        ctx->q[k] = RayQueue{};
    }
    (void)hipFree(ctx->hits); ctx->hits = nullptr;
    (void)hipFree(ctx->radiance); ctx->radiance = nullptr;
    ctx->capacity = 0; ctx->path_capacity = 0;
}

int ensure_state(rt_context* ctx, uint32_t paths, uint32_t capacity) {
    if (!ctx->ctl) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->ctl, CTL_WORDS * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->stats_dev, (CTL_MAX_IT + 3) * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemset(ctx->stats_dev, 0, (CTL_MAX_IT + 3) * sizeof(unsigned long long)));
    }
    if (capacity > ctx->capacity || paths > ctx->path_capacity) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        free_state(ctx);
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].a, (size_t)capacity * sizeof(float4)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].b, (size_t)capacity * sizeof(float4)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].c, (size_t)capacity * sizeof(float4)));
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->hits, (size_t)capacity * sizeof(float2)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->radiance, (size_t)paths * sizeof(float4)));
        ctx->capacity = capacity;
        ctx->path_capacity = paths;
    }
    return 0;
}

inline float next_down(float v) { return std::nextafter(v, -std::numeric_limits<float>::infinity()); }
inline float next_up(float v) { return std::nextafter(v, std::numeric_limits<float>::infinity()); }

int check_params(rt_context* ctx, const rt_render_params* p, uint32_t* s_begin, uint32_t* s_end, uint32_t* tcount) {
    if (!ctx->have_scene) return fail(ctx, "no scene uploaded");
    if (!p) return fail(ctx, "params is NULL");
    if (p->ssaa_mode != RT_SSAA_UNIFORM && p->ssaa_mode != RT_SSAA_JITTER) return fail(ctx, "invalid ssaa_mode %d", p->ssaa_mode);
    if (p->ssaa_n == 0 || p->ssaa_n > 4096) return fail(ctx, "ssaa_n must be in 1..4096 (got %u)", p->ssaa_n);
    if (p->depth > 255) return fail(ctx, "depth must be in 0..255 (the reference's u8), got %u", p->depth);
    uint32_t nn = p->ssaa_n * p->ssaa_n;
    *s_begin = p->sample_begin;
    *s_end = p->sample_end == 0 ? nn : p->sample_end;
    if (*s_end > nn || *s_begin > *s_end) return fail(ctx, "sample range [%u,%u) outside [0,%u)", *s_begin, *s_end, nn);
    *tcount = p->tile_count == 0 ? 1 : p->tile_count;
    if (p->tile_index >= *tcount) return fail(ctx, "tile_index %u >= tile_count %u", p->tile_index, *tcount);
    return 0;
}

}  // namespace

extern "C" {

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->error.c_str() : g_global_error.c_str(); }

int rt_context_create(int device_id, rt_context** out) {
    if (!out) return fail(nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, "device %d out of range (have %d)", device_id, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    rt_context* ctx = new rt_context();
    ctx->device = device_id;
    ctx->cfg.n_cu = (uint32_t)prop.multiProcessorCount;
    DeviceGuard g(device_id);
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = ctx;
    return 0;
}

void rt_context_destroy(rt_context* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipDeviceSynchronize();
    free_scene(ctx);
    free_state(ctx);
    (void)hipFree(ctx->ctl);
    (void)hipFree(ctx->stats_dev);
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int rt_scene_upload(rt_context* ctx, const rt_scene_desc* d) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!d) return fail(ctx, "scene is NULL");
    if (d->width == 0 || d->height == 0) return fail(ctx, "width and height must be non-zero");
    if ((uint64_t)d->width * d->height > (1ull << 31)) return fail(ctx, "image too large");
    if (d->n_triangles + d->n_spheres >= (1ull << 28)) return fail(ctx, "too many primitives (limit 2^28)");
    if (d->n_planes >= (1u << 20)) return fail(ctx, "too many planes");
    auto bad_mat = [&](int32_t m) { return m < 0 || (uint32_t)m >= d->n_materials; };
    for (uint32_t i = 0; i < d->n_spheres; ++i) if (bad_mat(d->sphere_material[i])) return fail(ctx, "sphere %u: material index out of range", i);
    for (uint32_t i = 0; i < d->n_planes; ++i) if (bad_mat(d->plane_material[i])) return fail(ctx, "plane %u: material index out of range", i);
    for (uint64_t i = 0; i < d->n_triangles; ++i)
        if (bad_mat(d->tri_material[i])) return fail(ctx, "triangle %llu: material index out of range (a face before any usemtl?)", (unsigned long long)i);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->mat_type[i] < 0 || d->mat_type[i] > 4) return fail(ctx, "material %u: unknown type %d", i, d->mat_type[i]);
        if (d->mat_type[i] == RT_MAT_TEXTURE && (d->mat_texture[i] < 0 || (uint32_t)d->mat_texture[i] >= d->n_textures))
            return fail(ctx, "material %u: texture index out of range", i);
    }
    DeviceGuard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    free_scene(ctx);

    // ---- scan rank (gid): objects in YAML order, triangles in file order (scene.rs:38, intersections.rs:148)
    struct Key { int32_t obj, tri; uint32_t kind, idx; };
    std::vector<Key> keys;
    keys.reserve(d->n_spheres + d->n_planes + d->n_triangles);
    for (uint32_t i = 0; i < d->n_spheres; ++i) keys.push_back({d->sphere_object[i], -1, 0u, i});
    for (uint32_t i = 0; i < d->n_planes; ++i) keys.push_back({d->plane_object[i], -1, 1u, i});
    for (uint64_t i = 0; i < d->n_triangles; ++i) keys.push_back({d->tri_object[i], d->tri_index[i], 2u, (uint32_t)i});
    std::vector<uint32_t> perm(keys.size());
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = (uint32_t)i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        return keys[a].obj < keys[b].obj || (keys[a].obj == keys[b].obj && keys[a].tri < keys[b].tri);
    });
    std::vector<uint32_t> gid_sphere(d->n_spheres), gid_plane(d->n_planes), gid_tri(d->n_triangles);
    for (size_t r = 0; r < perm.size(); ++r) {
        const Key& k = keys[perm[r]];
        if (k.kind == 0) gid_sphere[k.idx] = (uint32_t)r;
        else if (k.kind == 1) gid_plane[k.idx] = (uint32_t)r;
        else gid_tri[k.idx] = (uint32_t)r;
    }

    // ---- primitives for the BVH: spheres first, then triangles (input order)
    const uint32_t n_prims = (uint32_t)(d->n_spheres + d->n_triangles);
    std::vector<PrimBox> boxes(n_prims);
    std::vector<float> tri_f32((size_t)d->n_triangles * 9);
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        double r = std::fabs(d->sphere_radius[i]);
        for (int k = 0; k < 3; ++k) {
            boxes[i].mn[k] = next_down((float)(d->sphere_center[3 * i + k] - r));
            boxes[i].mx[k] = next_up((float)(d->sphere_center[3 * i + k] + r));
        }
    }
    for (uint64_t i = 0; i < d->n_triangles; ++i) {
        PrimBox& b = boxes[d->n_spheres + i];
        for (int k = 0; k < 3; ++k) { b.mn[k] = std::numeric_limits<float>::infinity(); b.mx[k] = -b.mn[k]; }
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                float f = (float)d->tri_vertices[9 * i + 3 * v + k];
                tri_f32[9 * i + 3 * v + k] = f;
                b.mn[k] = std::min(b.mn[k], f); b.mx[k] = std::max(b.mx[k], f);
            }
        // the kernel tests A + a*e1 + b*e2 with e1 = fl(B-A), e2 = fl(C-A): one more ulp of slack
        for (int k = 0; k < 3; ++k) { b.mn[k] = next_down(b.mn[k]); b.mx[k] = next_up(b.mx[k]); }
    }
    // Spheres that dwarf the rest of the scene (RTIOW's radius-1000 ground) stay out of the BVH: their
    // boxes enclose everything, and they are tested once per ray in full f64 beside the planes.
    std::vector<uint32_t> bvh_src, big_src;
    {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        auto grow = [&](const double* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } };
        for (uint32_t i = 0; i < d->n_spheres; ++i) grow(&d->sphere_center[3 * i]);
        for (uint64_t i = 0; i < d->n_triangles * 3; ++i) grow(&d->tri_vertices[3 * i]);
        double diag = 0;
        for (int k = 0; k < 3; ++k) if (hi[k] > lo[k]) diag += (hi[k] - lo[k]) * (hi[k] - lo[k]);
        diag = std::sqrt(diag);
        for (uint32_t i = 0; i < d->n_spheres; ++i) {
            const bool big = d->n_spheres + d->n_triangles > 1 && std::fabs(d->sphere_radius[i]) > 0.25 * diag && big_src.size() < 8;
            (big ? big_src : bvh_src).push_back(i);
        }
        for (uint64_t i = 0; i < d->n_triangles; ++i) bvh_src.push_back((uint32_t)(d->n_spheres + i));
    }
    std::vector<PrimBox> bvh_boxes(bvh_src.size());
    for (size_t i = 0; i < bvh_src.size(); ++i) bvh_boxes[i] = boxes[bvh_src[i]];
    Bvh bvh;
    if (ctx->bvh_builder == 1) {
        std::string berr;
        if (build_bvh_gpu(bvh_boxes, bvh, ctx->own_stream, berr)) return fail(ctx, "GPU BVH build failed: %s", berr.c_str());
    } else {
        build_bvh(bvh_boxes, bvh);
    }
    const uint32_t n_leaf_prims = (uint32_t)bvh_src.size();

    // ---- records: BVH primitives in leaf order, then the big spheres
    std::vector<float4> prims((size_t)n_prims * 3);
    std::vector<int4> meta(n_prims);
    std::vector<float4> smeta((size_t)n_prims * 4);   // 64-byte shading records: (normal, material | flags) + a copy of the primitive record
    std::vector<float> uvs;
    for (uint32_t k = 0; k < n_prims; ++k) {
        uint32_t src = k < n_leaf_prims ? bvh_src[bvh.order[k]] : big_src[k - n_leaf_prims];
        float4* r = &prims[(size_t)k * 3];
        if (src < d->n_spheres) {
            double v[4] = {d->sphere_center[3 * src], d->sphere_center[3 * src + 1], d->sphere_center[3 * src + 2], d->sphere_radius[src]};
            std::memcpy(r, v, 32);
            uint32_t w[4] = {0u, gid_sphere[src], PRIM_SPHERE, 0u};
            std::memcpy(&r[2], w, 16);
            meta[k] = make_int4(d->sphere_object[src], -1, d->sphere_material[src], -1);
            uint32_t mw = (uint32_t)d->sphere_material[src] | SMETA_SPHERE;
            float mf; std::memcpy(&mf, &mw, 4);
            smeta[(size_t)k * 4] = make_float4(0.f, 0.f, 0.f, mf);
        } else {
            uint64_t t = src - d->n_spheres;
            const float* v = &tri_f32[9 * t];
            float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
            float e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
            r[0] = make_float4(v[0], v[1], v[2], e1[0]);
            r[1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
            uint32_t gidw = gid_tri[t], tag = PRIM_TRIANGLE;
            float gf, tf;
            std::memcpy(&gf, &gidw, 4); std::memcpy(&tf, &tag, 4);
            r[2] = make_float4(e2[2], gf, tf, 0.f);
            int uv_off = -1;
            bool has_uv = d->tri_uv && (!d->tri_has_uv || d->tri_has_uv[t]);
            if (has_uv) {
                uv_off = (int)(uvs.size() / 6);
                for (int q = 0; q < 6; ++q) uvs.push_back((float)d->tri_uv[6 * t + q]);
            }
            meta[k] = make_int4(d->tri_object[t], d->tri_index[t], d->tri_material[t], uv_off);
            // unit normal in f64 from the f64 vertices (intersections.rs:153-154,169), rounded once
            const double* V = &d->tri_vertices[9 * t];
            const double ab[3] = {V[3] - V[0], V[4] - V[1], V[5] - V[2]}, ac[3] = {V[6] - V[0], V[7] - V[1], V[8] - V[2]};
            double nn[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
            const double nl = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            if (nl > 0.0) { nn[0] /= nl; nn[1] /= nl; nn[2] /= nl; }
            uint32_t mw = (uint32_t)d->tri_material[t] | (has_uv ? SMETA_HAS_UV : 0u);
            float mf; std::memcpy(&mf, &mw, 4);
            smeta[(size_t)k * 4] = make_float4((float)nn[0], (float)nn[1], (float)nn[2], mf);
        }
        for (int q = 0; q < 3; ++q) smeta[(size_t)k * 4 + 1 + q] = r[q];
    }
    std::vector<float4> nodes(bvh.nodes.size() * 4);
    if (!bvh.nodes.empty()) std::memcpy(nodes.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(GpuNode));

    std::vector<float4> planes((size_t)d->n_planes * 2);
    std::vector<int32_t> plane_obj(d->n_planes);
    for (uint32_t i = 0; i < d->n_planes; ++i) {
        float gf, mf;
        uint32_t gw = gid_plane[i];
        int32_t mw = d->plane_material[i];
        std::memcpy(&gf, &gw, 4); std::memcpy(&mf, &mw, 4);
        planes[2 * i] = make_float4((float)d->plane_center[3 * i], (float)d->plane_center[3 * i + 1], (float)d->plane_center[3 * i + 2], gf);
        planes[2 * i + 1] = make_float4((float)d->plane_normal[3 * i], (float)d->plane_normal[3 * i + 1], (float)d->plane_normal[3 * i + 2], mf);
        plane_obj[i] = d->plane_object[i];
    }
    std::vector<float4> mats((size_t)d->n_materials * 2);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        mats[2 * i] = make_float4((float)d->mat_rgb[3 * i], (float)d->mat_rgb[3 * i + 1], (float)d->mat_rgb[3 * i + 2], (float)d->mat_param[i]);
        int32_t w[4] = {d->mat_type[i], d->mat_texture ? d->mat_texture[i] : -1, 0, 0};
        std::memcpy(&mats[2 * i + 1], w, 16);
    }
    std::vector<int4> textures(d->n_textures);
    std::vector<uchar4> texels;
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        size_t n = (size_t)d->tex_width[i] * d->tex_height[i];
        if (n == 0 || n > (1ull << 30)) return fail(ctx, "texture %u has an invalid size", i);
        textures[i] = make_int4((int)texels.size(), (int)d->tex_width[i], (int)d->tex_height[i], 0);
        const uint8_t* px = d->tex_rgb8[i];
        for (size_t k = 0; k < n; ++k) texels.push_back(make_uchar4(px[3 * k], px[3 * k + 1], px[3 * k + 2], 255));
    }
    std::vector<float4> lights(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i)
        lights[i] = make_float4((float)d->light_point[3 * i], (float)d->light_point[3 * i + 1], (float)d->light_point[3 * i + 2], 0.f);

    DevScene sc{};
    if (upload(ctx, nodes, &sc.nodes) || upload(ctx, prims, &sc.prims) || upload(ctx, meta, &sc.meta) || upload(ctx, smeta, &sc.smeta) || upload(ctx, uvs, &sc.uvs) ||
        upload(ctx, planes, &sc.planes) || upload(ctx, plane_obj, &sc.plane_object) || upload(ctx, mats, &sc.materials) ||
        upload(ctx, textures, &sc.textures) || upload(ctx, texels, &sc.texels) || upload(ctx, lights, &sc.lights)) {
        free_scene(ctx);
        return 1;
    }
    sc.n_prims = n_prims;
    sc.n_leaf_prims = n_leaf_prims;
    sc.n_nodes = (uint32_t)bvh.nodes.size();
    sc.n_planes = d->n_planes;
    sc.n_lights = d->n_lights;
    for (int k = 0; k < 3; ++k) {
        sc.bg[k] = (float)d->background[k];
        sc.eye[k] = (float)d->cam_eye[k];
        sc.lle[k] = d->cam_lower_left[k] - d->cam_eye[k];
        sc.hz[k] = d->cam_horizontal[k];
        sc.vt[k] = d->cam_vertical[k];
    }
    sc.width = d->width; sc.height = d->height;
    sc.stack_depth = std::max<uint32_t>(bvh.max_stack, 2);
    sc.spill_levels = sc.stack_depth > LDS_STACK ? sc.stack_depth - LDS_STACK : 0;
    {   // overflow area of the LDS traversal stacks: one slice per persistent closest-hit wave
        const size_t waves = (size_t)ctx->cfg.n_cu * HIT_BLOCKS_PER_CU * 4;
        const size_t bytes = std::max<size_t>(16, waves * 64 * sc.spill_levels * sizeof(uint32_t));
        void* sp = nullptr;
        HIP_TRY(ctx, hipMalloc(&sp, bytes));
        ctx->scene_allocs.push_back(sp);
        sc.spill = reinterpret_cast<uint32_t*>(sp);
    }
    ctx->sc = sc;
    ctx->bvh_nodes = sc.n_nodes; ctx->bvh_depth = bvh.max_depth; ctx->bvh_prims = n_prims; ctx->bvh_build_ms = bvh.build_ms;
    ctx->have_scene = true;
    return 0;
}

// One call = raytrace.rs:126-143 for a sample range and a tile set (see rtamd.h).
static int render_accumulate_impl(rt_context* ctx, const rt_render_params* p, float* accum_dev, hipStream_t st, bool hits_only,
                                  PassParams* pp_out) {
    uint32_t sb, se, tcount;
    if (check_params(ctx, p, &sb, &se, &tcount)) return 1;
    if (!accum_dev && !hits_only) return fail(ctx, "accum_dev is NULL");
    const DevScene& sc = ctx->sc;
    const uint32_t tiles_x = (sc.width + 7) / 8, tiles_y = (sc.height + 7) / 8;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    const uint32_t n_local = p->tile_index < n_tiles ? (uint32_t)((n_tiles - p->tile_index + tcount - 1) / tcount) : 0;
    if (n_local == 0 || sb == se) return 0;
    uint32_t max_paths = p->max_paths_per_pass ? p->max_paths_per_pass : (1u << 27);   // few, large launches: 16 Mi-path passes were 15 % slower
    max_paths = std::max<uint32_t>(64, max_paths & ~63u);
    const uint32_t gens = sc.n_lights ? (p->max_shadow_generations ? p->max_shadow_generations : 16u) : 0u;
    if (p->depth + gens + 1 > (uint32_t)CTL_MAX_IT) return fail(ctx, "depth + shadow generations exceeds %d", CTL_MAX_IT - 1);

    // tiles per pass / samples per pass.  With the default pass size a device that cannot hold the
    // queues (128 B per path) gets smaller passes instead of an error.
    uint32_t tiles_pass, samples_pass;
    for (;;) {
        tiles_pass = (uint32_t)std::min<uint64_t>(n_local, max_paths / 64);
        samples_pass = std::max<uint32_t>(1, std::min<uint32_t>(se - sb, max_paths / (tiles_pass * 64)));
        const uint32_t pass_paths = tiles_pass * 64 * samples_pass;
        // a firing path spawns one shadow ray per light (raytrace.rs:78-85)
        const uint64_t cap64 = (uint64_t)pass_paths * std::max<uint32_t>(1, std::min<uint32_t>(sc.n_lights, 4));
        if (cap64 > 0xFFFFFF00ull) {
            if (!p->max_paths_per_pass && max_paths > (1u << 20)) { max_paths >>= 1; continue; }
            return fail(ctx, "pass too large");
        }
        if (ensure_state(ctx, pass_paths, (uint32_t)cap64) == 0) break;
        free_state(ctx);
        (void)hipGetLastError();
        if (p->max_paths_per_pass || max_paths <= (1u << 20)) return 1;
        max_paths >>= 1;
    }
    const uint32_t capacity = ctx->capacity;
    const bool counters = (p->flags & RT_FLAG_COUNTERS) != 0, timing = (p->flags & RT_FLAG_TIMING) != 0;

    for (uint32_t lt0 = 0; lt0 < n_local; lt0 += tiles_pass) {
        const uint32_t nt = std::min(tiles_pass, n_local - lt0);
        for (uint32_t s0 = sb; s0 < se; s0 += samples_pass) {
            const uint32_t ns = std::min(samples_pass, se - s0);
            PassParams pp{};
            pp.npix_pad = nt * 64; pp.n_paths = pp.npix_pad * ns; pp.ltile_begin = lt0;
            pp.tiles_x = tiles_x; pp.tiles_y = tiles_y; pp.tile_index = p->tile_index; pp.tile_count = tcount;
            pp.s_begin = s0; pp.n_samples = ns; pp.ssaa_n = p->ssaa_n; pp.jitter = p->ssaa_mode == RT_SSAA_JITTER;
            pp.step = 1.0f / (float)p->ssaa_n;                         // anti_aliasing.rs:94
            pp.off_x = (float)p->sample_offset[0]; pp.off_y = (float)p->sample_offset[1];
            pp.seed_lo = (uint32_t)p->seed; pp.seed_hi = (uint32_t)(p->seed >> 32);
            if (pp_out) *pp_out = pp;
            HIP_TRY(ctx, hipMemsetAsync(ctx->ctl, 0, CTL_WORDS * sizeof(uint32_t), st));
            launch_raygen(ctx->cfg, sc, pp, ctx->q[0], ctx->radiance, ctx->ctl, capacity, st);
            {   // primary rays = pixels of the pass's tiles that lie inside the image
                uint64_t px = 0;
                for (uint32_t lt = lt0; lt < lt0 + nt; ++lt) {
                    const uint64_t gt = (uint64_t)lt * tcount + p->tile_index;
                    if (gt >= n_tiles) break;
                    const uint32_t ty = (uint32_t)(gt / tiles_x), tx = (uint32_t)(gt % tiles_x);
                    px += (uint64_t)std::min(8u, sc.width - tx * 8) * std::min(8u, sc.height - ty * 8);
                }
                ctx->paths_total += px * ns;
                if (hits_only || p->depth > 0) ctx->rays0_total += px * ns;
            }
            const uint32_t depth = hits_only ? 1u : p->depth;
            const uint32_t iters = depth ? depth + gens : 0u;          // raytrace.rs:52-54: depth 0 traces nothing
            for (uint32_t it = 0; it < iters; ++it) {
                RayQueue qi = ctx->q[it & 1], qo = ctx->q[(it + 1) & 1];
                const uint32_t max_rays = it < depth ? pp.n_paths : capacity;
                if (timing) {
                    if (ctx->events_used == ctx->events.size()) {
                        hipEvent_t a, b;
                        HIP_TRY(ctx, hipEventCreate(&a));
                        HIP_TRY(ctx, hipEventCreate(&b));
                        ctx->events.push_back({a, b});
                    }
                    HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used].first, st));
                }
                launch_closest_hit(ctx->cfg, sc, qi, ctx->hits, ctx->ctl, (int)it, capacity, max_rays, counters, st);
                if (timing) {
                    HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used].second, st));
                    ctx->events_used++;
                }
                if (hits_only) break;
                const int last = it + 1 >= depth;
                launch_shade(ctx->cfg, sc, qi, ctx->hits, qo, ctx->radiance, ctx->ctl, (int)it, last, capacity, max_rays, st);
            }
            if (!hits_only) launch_accumulate(sc, pp, ctx->radiance, accum_dev, st);
            launch_stats(ctx->ctl, ctx->stats_dev, capacity, st);
            HIP_TRY(ctx, hipGetLastError());
            if (hits_only) return 0;   // single pass by construction
        }
    }
    return 0;
}

int rt_render_accumulate(rt_context* ctx, const rt_render_params* params, float* accum_dev, void* stream) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    DeviceGuard g(ctx->device);
    return render_accumulate_impl(ctx, params, accum_dev, (hipStream_t)stream, false, nullptr);
}

int rt_resolve(rt_context* ctx, const float* accum_dev, uint32_t spp_total, uint8_t* rgb8_dev, float* color_dev, void* stream) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, "no scene uploaded");
    if (!accum_dev) return fail(ctx, "accum_dev is NULL");
    if (spp_total == 0) return fail(ctx, "spp_total must be non-zero");
    DeviceGuard g(ctx->device);
    const uint32_t n = ctx->sc.width * ctx->sc.height * 3u;
    launch_resolve(accum_dev, n, 1.0f / (float)spp_total, rgb8_dev, color_dev, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int rt_render_image(rt_context* ctx, const rt_render_params* params, float* accum_host, uint8_t* rgb8_host) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    uint32_t sb, se, tcount;
    if (check_params(ctx, params, &sb, &se, &tcount)) return 1;
    DeviceGuard g(ctx->device);
    const size_t n = (size_t)ctx->sc.width * ctx->sc.height * 3;
    float* accum = nullptr;
    uint8_t* rgb8 = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&accum, n * sizeof(float)));
    int rc = 0;
    hipStream_t st = ctx->own_stream;
    // rays dropped because a queue was full (only the branching shadow-ray process of scenes with
    // several lights can do that, raytrace.rs:78-85) make the image wrong: this call reports them
    auto dropped = [&]() -> unsigned long long {
        unsigned long long v = 0;
        if (ctx->stats_dev) (void)hipMemcpy(&v, ctx->stats_dev + CTL_MAX_IT, sizeof(v), hipMemcpyDeviceToHost);
        return v;
    };
    const unsigned long long dropped_before = dropped();
    do {
        if (hipMemsetAsync(accum, 0, n * sizeof(float), st) != hipSuccess) { rc = fail(ctx, "hipMemsetAsync failed"); break; }
        if ((rc = render_accumulate_impl(ctx, params, accum, st, false, nullptr))) break;
        if (rgb8_host) {
            if (hipMalloc((void**)&rgb8, n) != hipSuccess) { rc = fail(ctx, "hipMalloc failed"); break; }
            // raytrace.rs:144: divide by the number of samples of the pixel (here: of the range traced)
            if ((rc = rt_resolve(ctx, accum, se - sb ? se - sb : 1, rgb8, nullptr, st))) break;
        }
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { rc = fail(ctx, "render failed: %s", hipGetErrorString(e)); break; }
        if (accum_host && hipMemcpy(accum_host, accum, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, "copy failed"); break; }
        if (rgb8_host && hipMemcpy(rgb8_host, rgb8, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, "copy failed"); break; }
        const unsigned long long lost = dropped() - dropped_before;
        if (lost) rc = fail(ctx, "%llu rays were dropped: the shadow-ray queue overflowed (%u lights; lower max_shadow_generations or max_paths_per_pass)",
                            lost, ctx->sc.n_lights);
    } while (0);
    (void)hipFree(accum);
    (void)hipFree(rgb8);
    return rc;
}

int rt_primary_hits(rt_context* ctx, double dx, double dy, int32_t* object_id, int32_t* triangle_id, float* t, float* normal_xyz) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, "no scene uploaded");
    if (!object_id || !triangle_id || !t || !normal_xyz) return fail(ctx, "output pointer is NULL");
    DeviceGuard g(ctx->device);
    const size_t npx = (size_t)ctx->sc.width * ctx->sc.height;
    const uint32_t tiles = ((ctx->sc.width + 7) / 8) * ((ctx->sc.height + 7) / 8);
    rt_render_params p{};
    p.ssaa_mode = RT_SSAA_UNIFORM; p.ssaa_n = 1; p.depth = 1; p.tile_count = 1;
    p.max_paths_per_pass = tiles * 64;   // one pass over every tile
    p.sample_offset[0] = dx; p.sample_offset[1] = dy;
    int32_t *d_obj = nullptr, *d_tri = nullptr;
    float *d_t = nullptr, *d_n = nullptr;
    int rc = 0;
    hipStream_t st = ctx->own_stream;
    do {
        if (hipMalloc((void**)&d_obj, npx * 4) != hipSuccess || hipMalloc((void**)&d_tri, npx * 4) != hipSuccess ||
            hipMalloc((void**)&d_t, npx * 4) != hipSuccess || hipMalloc((void**)&d_n, npx * 12) != hipSuccess) { rc = fail(ctx, "hipMalloc failed"); break; }
        PassParams pp{};
        if ((rc = render_accumulate_impl(ctx, &p, nullptr, st, true, &pp))) break;
        launch_hit_info(ctx->sc, pp, ctx->q[0], ctx->hits, ctx->ctl, ctx->capacity, d_obj, d_tri, d_t, d_n, st);
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { rc = fail(ctx, "primary hits failed: %s", hipGetErrorString(e)); break; }
        if (hipMemcpy(object_id, d_obj, npx * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(triangle_id, d_tri, npx * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(t, d_t, npx * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(normal_xyz, d_n, npx * 12, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, "copy failed"); break; }
    } while (0);
    (void)hipFree(d_obj); (void)hipFree(d_tri); (void)hipFree(d_t); (void)hipFree(d_n);
    return rc;
}

int rt_context_set_bvh_builder(rt_context* ctx, int builder) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (builder != 0 && builder != 1) return fail(ctx, "unknown BVH builder %d (0: host SAH, 1: GPU LBVH)", builder);
    ctx->bvh_builder = builder;
    return 0;
}

int rt_bvh_build_validate(rt_context* ctx, uint64_t n, const float* boxes, uint32_t* n_nodes, uint32_t* depth, uint32_t* max_stack, double* build_ms) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!boxes && n) return fail(ctx, "boxes is NULL");
    if (n > 0x7FFFFFF0ull) return fail(ctx, "too many primitives");
    DeviceGuard g(ctx->device);
    std::vector<PrimBox> pb(n);
    for (uint64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { pb[i].mn[k] = boxes[6 * i + k]; pb[i].mx[k] = boxes[6 * i + 3 + k]; }
    Bvh bvh;
    if (ctx->bvh_builder == 1) {
        std::string berr;
        if (build_bvh_gpu(pb, bvh, ctx->own_stream, berr)) return fail(ctx, "GPU BVH build failed: %s", berr.c_str());
    } else {
        build_bvh(pb, bvh);
    }
    if (n_nodes) *n_nodes = (uint32_t)bvh.nodes.size();
    if (depth) *depth = bvh.max_depth;
    if (max_stack) *max_stack = bvh.max_stack;
    if (build_ms) *build_ms = bvh.build_ms;
    try {
        if (n && bvh.order.size() != n) return fail(ctx, "order has %zu entries for %llu primitives", bvh.order.size(), (unsigned long long)n);
        validate_bvh(bvh, pb);
    } catch (const std::exception& e) {
        return fail(ctx, "%s", e.what());
    }
    return 0;
}

int rt_get_stats(rt_context* ctx, rt_stats* out) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!out) return fail(ctx, "out is NULL");
    DeviceGuard g(ctx->device);
    std::memset(out, 0, sizeof(*out));
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ctx->stats_dev) {
        std::vector<unsigned long long> h(CTL_MAX_IT + 3);
        HIP_TRY(ctx, hipMemcpy(h.data(), ctx->stats_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        h[0] = ctx->rays0_total;
        for (int i = 0; i < CTL_MAX_IT; ++i) { out->rays_per_iteration[i] = h[i]; out->rays_total += h[i]; }
        out->queue_overflow = h[CTL_MAX_IT];
        out->nodes_visited = h[CTL_MAX_IT + 1];
        out->prims_tested = h[CTL_MAX_IT + 2];
    }
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second) == hipSuccess) { ctx->hit_ms += ms; ctx->hit_launches++; }
    }
    ctx->events_used = 0;
    out->hit_ms = ctx->hit_ms;
    out->hit_launches = ctx->hit_launches;
    out->paths = ctx->paths_total;
    out->bvh_nodes = ctx->bvh_nodes; out->bvh_depth = ctx->bvh_depth; out->bvh_prims = ctx->bvh_prims; out->bvh_build_ms = ctx->bvh_build_ms;
    return 0;
}

int rt_debug_gather(rt_context* ctx, uint64_t table_bytes, uint32_t iters, double* bytes_loaded) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (table_bytes < 4096 || iters == 0 || !bytes_loaded) return fail(ctx, "invalid argument");
    DeviceGuard g(ctx->device);
    const uint64_t n_records = table_bytes / 64;
    if (n_records > 0xFFFFFFFFull) return fail(ctx, "table too large");
    float4* table = nullptr;
    float* sink = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&table, n_records * 64));
    if (hipMalloc((void**)&sink, 16) != hipSuccess) { (void)hipFree(table); return fail(ctx, "hipMalloc failed"); }
    int rc = 0;
    const uint32_t blocks = ctx->cfg.n_cu * 8;
    if (hipMemsetAsync(table, 0, n_records * 64, ctx->own_stream) != hipSuccess) rc = fail(ctx, "memset failed");
    if (!rc) {
        launch_calib_gather(table, (uint32_t)n_records, iters, blocks, sink, ctx->own_stream);
        hipError_t e = hipStreamSynchronize(ctx->own_stream);
        if (e != hipSuccess) rc = fail(ctx, "calibration kernel failed: %s", hipGetErrorString(e));
    }
    *bytes_loaded = (double)blocks * 256.0 * (double)iters * 64.0;
    (void)hipFree(table);
    (void)hipFree(sink);
    return rc;
}

int rt_stats_reset(rt_context* ctx) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    DeviceGuard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ctx->stats_dev) HIP_TRY(ctx, hipMemset(ctx->stats_dev, 0, (CTL_MAX_IT + 3) * sizeof(unsigned long long)));
    ctx->events_used = 0; ctx->hit_ms = 0; ctx->hit_launches = 0; ctx->paths_total = 0; ctx->rays0_total = 0;
    return 0;
}

}  // extern "C"
