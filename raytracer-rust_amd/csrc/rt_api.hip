// C-ABI of the render path (include/rtamd.h): context, scene upload (f64 -> device
// layout + BVH), the wavefront loop and the statistics.  No torch types, no exceptions
// across the boundary.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/rtamd.h"
#include "bvh.h"
#include "rt_kernels.h"

using namespace rtamd;

static_assert(RT_MAX_ITERATIONS == CTL_MAX_IT, "iteration table sizes must agree");

static thread_local std::string g_global_error;

// Scratch of rt_render_accumulate_noalloc (rtamd.h): the wavefront state of ONE pass -- control block,
// two ray queues, hit records, sample radiance -- for callers that must not allocate (HIP graph capture,
// serving loops).  Every GPU the library runs on holds its own copy of this array until the process ends.
// The name is a host-side stand-in, not a device address: rt_context_create asks the runtime for the
// address on its device (hipGetSymbolAddress) and only that pointer is ever handed to kernels or memsets.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[256] = {};      // per device

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
    bool any_hit_last = true; // the paths' last rays stop at their first hit in scenes without emitters; RTAMD_ANY_HIT=0 (read at rt_context_create, applied at scene upload): closest hit and full shading as for every ray (A/B, tests)
    bool gen_primary = false; // RTAMD_GEN_PRIMARY=1: iteration 0 forms the primary rays itself instead of reading what k_raygen wrote.  Measured and left
                              // off: k_raygen's 0.9 ms per 133 M paths go, but k_closest_hit (+0.65 ms) and k_shade (+0.8 ms) pay more for forming the ray
                              // again than they save in reads (C5 24.75 -> 25.3 ms per 64-spp step, C3 14.8 -> 15.2)
    bool shade_split = false; // RTAMD_SHADE_SPLIT=1: k_shade in two launches per iteration (CLS).  Measured and left off: bit-identical renders, C5 24.34 -> 24.46 ms
                              // per 64-spp step, C4 107.7 -> 115.4 -- the triangle code a mixed wave runs for a handful of lanes is not what its rounds wait for
    bool tile_entry = true;   // primary rays enter the BVH at their tile's entry points (k_tile_entry); RTAMD_TILE_ENTRY=0: at the root (A/B, tests)
    uint32_t bvh_nodes = 0, bvh_depth = 0, bvh_prims = 0;
    double bvh_build_ms = 0;

    // wavefront state (grown on demand, reused across calls)
    uint32_t capacity = 0;      // rays per queue
    uint32_t path_capacity = 0; // radiance slots
    RayQueue q[2]{};
    float2* hits = nullptr;    // (t, code) per ray
    uint32_t* heavy = nullptr; // queue slots of the rays k_shade<.., 1> leaves to k_shade<.., 2> (one per ray of a queue at most)
    float4* radiance = nullptr;
    uint32_t* ctl = nullptr;
    unsigned long long* stats_dev = nullptr;   // CTL_MAX_IT + 1 + N_STAT64 words
    uint64_t paths_total = 0;
    uint64_t rays0_total = 0;   // primary rays actually traced (iteration 0)

    // multi-GPU exchange step (rt_accum_reduce, rt_render_image_devices): grown on demand
    float* own_accum = nullptr;    // this context's partial frame
    float* staging = nullptr;      // landing buffer of peer copies
    size_t own_accum_values = 0, staging_values = 0;
    void* scratch_ptr = nullptr;   // device address of `scratch` on this context's GPU
    // RCCL communicator of a group of contexts (rt_accum_reduce_rccl), kept by the group's first context: one rank per GPU
    std::vector<ncclComm_t> comms;
    std::vector<int> comm_devices;
    bool rccl_unavailable = false; // communicator creation failed once: rt_render_image_devices stays with peer copies

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
        (void)hipFree(ctx->q[k].a); (void)hipFree(ctx->q[k].b); (void)hipFree(ctx->q[k].c); (void)hipFree(ctx->q[k].e); (void)hipFree(ctx->q[k].f);
        ctx->q[k] = RayQueue{};
    }
    (void)hipFree(ctx->hits); ctx->hits = nullptr;
    (void)hipFree(ctx->heavy); ctx->heavy = nullptr;
    (void)hipFree(ctx->radiance); ctx->radiance = nullptr;
    ctx->capacity = 0; ctx->path_capacity = 0;
}

int ensure_state(rt_context* ctx, uint32_t paths, uint32_t capacity) {
    if (capacity > ctx->capacity || paths > ctx->path_capacity) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        free_state(ctx);
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].a, (size_t)capacity * sizeof(float4)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].b, (size_t)capacity * sizeof(float4)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].c, (size_t)capacity * sizeof(float4)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].e, (size_t)capacity * sizeof(uint4)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->q[k].f, (size_t)capacity * sizeof(uint4)));
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->hits, (size_t)capacity * sizeof(float2)));
        if (ctx->shade_split) HIP_TRY(ctx, hipMalloc((void**)&ctx->heavy, (size_t)capacity * sizeof(uint32_t)));
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
    if (const char* e = getenv("RTAMD_TILE_ENTRY")) ctx->tile_entry = atoi(e) != 0;
    if (const char* e = getenv("RTAMD_GEN_PRIMARY")) ctx->gen_primary = atoi(e) != 0;
    if (const char* e = getenv("RTAMD_SHADE_SPLIT")) ctx->shade_split = atoi(e) != 0;
    if (const char* e = getenv("RTAMD_ANY_HIT")) ctx->any_hit_last = atoi(e) != 0;
    ctx->device = device_id;
    ctx->cfg.n_cu = (uint32_t)prop.multiProcessorCount;
    DeviceGuard g(device_id);
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, "hipStreamCreate: %s", hipGetErrorString(e)); }
    // control block and statistics live as long as the context (no entry point allocates them later)
    e = hipMalloc((void**)&ctx->ctl, CTL_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->stats_dev, (CTL_MAX_IT + 1 + N_STAT64) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->stats_dev, 0, (CTL_MAX_IT + 1 + N_STAT64) * sizeof(unsigned long long));
    if (e != hipSuccess) { rt_context_destroy(ctx); return fail(nullptr, "hipMalloc: %s", hipGetErrorString(e)); }
    // The kernels' code object is loaded (and whatever the runtime allocates for it, allocated) at the first launch on a
    // device: do that here, so that no later entry point -- rt_render_accumulate_noalloc under graph capture least of all --
    // is the first.  (The control block is all zero: the statistics kernel adds nothing.)
    e = hipMemset(ctx->ctl, 0, CTL_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) { launch_stats(ctx->ctl, ctx->stats_dev, 0u, nullptr); e = hipDeviceSynchronize(); }
    if (e != hipSuccess) { rt_context_destroy(ctx); return fail(nullptr, "first kernel launch failed: %s", hipGetErrorString(e)); }
    {   // device address of the no-alloc scratch array, once per device, with that device current
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (device_id < (int)(sizeof(g_scratch_ptr) / sizeof(g_scratch_ptr[0])) && !g_scratch_ptr[device_id]) {
            void* ptr = nullptr;
            e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) { rt_context_destroy(ctx); return fail(nullptr, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e)); }
            g_scratch_ptr[device_id] = ptr;
        }
        ctx->scratch_ptr = device_id < (int)(sizeof(g_scratch_ptr) / sizeof(g_scratch_ptr[0])) ? g_scratch_ptr[device_id] : nullptr;
    }
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
    (void)hipFree(ctx->own_accum);
    (void)hipFree(ctx->staging);
    for (ncclComm_t c : ctx->comms) (void)ncclCommDestroy(c);
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

}  // extern "C" (reopened below)

// Host-side image of one scene in the device layout (rt_device.h): flattened and its BVH built ONCE,
// then copied to any number of contexts (rt_scene_upload_devices).
struct FlatScene {
    std::vector<float4> nodes, prims, smeta, planes, mats, lights;
    std::vector<int4> meta, textures;
    std::vector<float> uvs;
    std::vector<int32_t> plane_obj;
    std::vector<double> mat_param64; // the material parameters as given (f64)
    std::vector<double> planes64;    // centre, normal of every plane as given (f64)
    std::vector<uchar4> texels;
    uint32_t n_prims = 0, n_leaf_prims = 0, n_nodes = 0, n_planes = 0, n_lights = 0, n_tris = 0;
    float bg[3], eye[3];
    double eye64[3], lle[3], hz[3], vt[3], ll[3];
    uint32_t width = 0, height = 0, stack_depth = 2, bvh_depth = 0;
    bool no_emitters = false;   // no Emissive and no Texture material (DevScene::no_emitters)
    double bvh_build_ms = 0;
};

// what the reference's `unwrap()`s would panic on, and what would make this library read out of bounds
static int validate_desc(const rt_scene_desc* d, std::string& err) {
    char buf[256];
    auto bad = [&](const char* fmt, unsigned long long i) { snprintf(buf, sizeof(buf), fmt, i); err = buf; return 1; };
    if (d->width == 0 || d->height == 0) { err = "width and height must be non-zero"; return 1; }
    if ((uint64_t)d->width * d->height > (1ull << 31)) { err = "image too large"; return 1; }
    if (d->n_triangles + d->n_spheres >= (1ull << 28)) { err = "too many primitives (limit 2^28)"; return 1; }
    if (d->n_planes >= (1u << 20)) { err = "too many planes"; return 1; }
    if (d->n_materials && (!d->mat_type || !d->mat_rgb || !d->mat_param)) { err = "material arrays are NULL"; return 1; }
    if (d->n_textures && (!d->tex_width || !d->tex_height || !d->tex_rgb8)) { err = "texture arrays are NULL"; return 1; }
    if (d->n_spheres && (!d->sphere_center || !d->sphere_radius || !d->sphere_material || !d->sphere_object)) { err = "sphere arrays are NULL"; return 1; }
    if (d->n_planes && (!d->plane_center || !d->plane_normal || !d->plane_material || !d->plane_object)) { err = "plane arrays are NULL"; return 1; }
    if (d->n_triangles && (!d->tri_vertices || !d->tri_material || !d->tri_object || !d->tri_index)) { err = "triangle arrays are NULL"; return 1; }
    if (d->n_lights && !d->light_point) { err = "light_point is NULL"; return 1; }
    auto bad_mat = [&](int32_t m) { return m < 0 || (uint32_t)m >= d->n_materials; };
    for (uint32_t i = 0; i < d->n_spheres; ++i) if (bad_mat(d->sphere_material[i])) return bad("sphere %llu: material index out of range", i);
    for (uint32_t i = 0; i < d->n_planes; ++i) if (bad_mat(d->plane_material[i])) return bad("plane %llu: material index out of range", i);
    for (uint64_t i = 0; i < d->n_triangles; ++i)
        if (bad_mat(d->tri_material[i])) return bad("triangle %llu: material index out of range (a face before any usemtl?)", i);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->mat_type[i] < 0 || d->mat_type[i] > 4) return bad("material %llu: unknown type", i);
        if (d->mat_type[i] == RT_MAT_TEXTURE && (!d->mat_texture || d->mat_texture[i] < 0 || (uint32_t)d->mat_texture[i] >= d->n_textures))
            return bad("material %llu: texture index out of range", i);
    }
    // Every coordinate must be finite as an f32 (the device's type): the reference would carry NaN / inf through to the
    // pixels, here they would reach the BVH builder and the traversal's rounding bounds.
    auto finite_all = [](const double* v, uint64_t n) -> int64_t {
        for (uint64_t i = 0; i < n; ++i) if (!std::isfinite((float)v[i])) return (int64_t)i;
        return -1;
    };
    int64_t at;
    if ((at = finite_all(d->sphere_center, 3ull * d->n_spheres)) >= 0) return bad("sphere %llu: centre is not finite", (unsigned long long)at / 3);
    if ((at = finite_all(d->sphere_radius, d->n_spheres)) >= 0) return bad("sphere %llu: radius is not finite", (unsigned long long)at);
    if ((at = finite_all(d->plane_center, 3ull * d->n_planes)) >= 0) return bad("plane %llu: centre is not finite", (unsigned long long)at / 3);
    if ((at = finite_all(d->plane_normal, 3ull * d->n_planes)) >= 0) return bad("plane %llu: normal is not finite", (unsigned long long)at / 3);
    if ((at = finite_all(d->tri_vertices, 9ull * d->n_triangles)) >= 0) return bad("triangle %llu: a vertex is not finite", (unsigned long long)at / 9);
    if ((at = finite_all(d->light_point, 3ull * d->n_lights)) >= 0) return bad("light %llu: sample point is not finite", (unsigned long long)at / 3);
    if (finite_all(d->cam_eye, 3) >= 0 || finite_all(d->cam_lower_left, 3) >= 0 || finite_all(d->cam_horizontal, 3) >= 0 || finite_all(d->cam_vertical, 3) >= 0) {
        err = "camera vectors are not finite (look_at == eye, up parallel to the view direction, fovy out of range?)";
        return 1;
    }
    if (finite_all(d->background, 3) >= 0) { err = "background is not finite"; return 1; }
    return 0;
}

// f64 scene -> device layout + BVH (built with ctx's builder; the GPU builder runs on ctx's device)
static int flatten_scene(rt_context* ctx, const rt_scene_desc* d, FlatScene& f, std::string& err) {
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
        DeviceGuard g(ctx->device);
        if (build_bvh_gpu(bvh_boxes, bvh, ctx->own_stream, berr)) { err = "GPU BVH build failed: " + berr; return 1; }
    } else {
        build_bvh(bvh_boxes, bvh);
    }
    const uint32_t n_leaf_prims = (uint32_t)bvh_src.size();

    // ---- records: BVH primitives in leaf order, then the big spheres
    std::vector<float4>& prims = f.prims;
    std::vector<int4>& meta = f.meta;
    std::vector<float4>& smeta = f.smeta;   // 96-byte shading records: (normal, material | flags) + the reference's f64 operands (rt_device.h)
    std::vector<float>& uvs = f.uvs;
    prims.assign((size_t)n_prims * 3, make_float4(0.f, 0.f, 0.f, 0.f));
    meta.assign(n_prims, make_int4(0, 0, 0, 0));
    smeta.assign((size_t)n_prims * SMETA_WORDS, make_float4(0.f, 0.f, 0.f, 0.f));
    uvs.clear();
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
            smeta[(size_t)k * SMETA_WORDS] = make_float4(0.f, 0.f, 0.f, mf);
            std::memcpy(&smeta[(size_t)k * SMETA_WORDS + 1], v, 32);
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
            // last word: what the f32 edge tests of k_closest_hit allow for (RT_EDGE_TOL, see the leaf step), per unit of distance between the ray's origin and the triangle
            const float el = std::max(std::fabs(e1[0]) + std::fabs(e1[1]) + std::fabs(e1[2]), std::fabs(e2[0]) + std::fabs(e2[1]) + std::fabs(e2[2]));
            r[2] = make_float4(e2[2], gf, tf, 0x1p-25f * el);
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
            smeta[(size_t)k * SMETA_WORDS] = make_float4((float)nn[0], (float)nn[1], (float)nn[2], mf);
            // the reference's own operands, in its own number type: A, ab = B - A, ac = C - A (intersections.rs:149-154)
            const double ops[10] = {V[0], V[1], V[2], ab[0], ab[1], ab[2], ac[0], ac[1], ac[2], 0.0};
            std::memcpy(&smeta[(size_t)k * SMETA_WORDS + 1], ops, sizeof(ops));
        }
    }
    std::vector<float4>& nodes = f.nodes;
    nodes.assign(bvh.nodes.size() * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    if (!bvh.nodes.empty()) std::memcpy(nodes.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(GpuNode));

    std::vector<float4>& planes = f.planes;
    std::vector<int32_t>& plane_obj = f.plane_obj;
    planes.assign((size_t)d->n_planes * 2, make_float4(0.f, 0.f, 0.f, 0.f));
    plane_obj.assign(d->n_planes, 0);
    for (uint32_t i = 0; i < d->n_planes; ++i) {
        float gf, mf;
        uint32_t gw = gid_plane[i];
        int32_t mw = d->plane_material[i];
        std::memcpy(&gf, &gw, 4); std::memcpy(&mf, &mw, 4);
        planes[2 * i] = make_float4((float)d->plane_center[3 * i], (float)d->plane_center[3 * i + 1], (float)d->plane_center[3 * i + 2], gf);
        planes[2 * i + 1] = make_float4((float)d->plane_normal[3 * i], (float)d->plane_normal[3 * i + 1], (float)d->plane_normal[3 * i + 2], mf);
        plane_obj[i] = d->plane_object[i];
        for (int k = 0; k < 3; ++k) f.planes64.push_back(d->plane_center[3 * i + k]);
        for (int k = 0; k < 3; ++k) f.planes64.push_back(d->plane_normal[3 * i + k]);
    }
    std::vector<float4>& mats = f.mats;
    mats.assign((size_t)d->n_materials * 2, make_float4(0.f, 0.f, 0.f, 0.f));
    f.no_emitters = d->n_lights == 0;
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        mats[2 * i] = make_float4((float)d->mat_rgb[3 * i], (float)d->mat_rgb[3 * i + 1], (float)d->mat_rgb[3 * i + 2], (float)d->mat_param[i]);
        f.mat_param64.push_back(d->mat_param[i]);
        int32_t w[4] = {d->mat_type[i], d->mat_texture ? d->mat_texture[i] : -1, 0, 0};
        std::memcpy(&mats[2 * i + 1], w, 16);
        if (d->mat_type[i] == RT_MAT_EMISSIVE || d->mat_type[i] == RT_MAT_TEXTURE) f.no_emitters = false;
    }
    std::vector<int4>& textures = f.textures;
    std::vector<uchar4>& texels = f.texels;
    textures.assign(d->n_textures, make_int4(0, 0, 0, 0));
    texels.clear();
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        size_t n = (size_t)d->tex_width[i] * d->tex_height[i];
        if (n == 0 || n > (1ull << 30) || !d->tex_rgb8[i]) { err = "texture " + std::to_string(i) + " has an invalid size or no pixels"; return 1; }
        textures[i] = make_int4((int)texels.size(), (int)d->tex_width[i], (int)d->tex_height[i], 0);
        const uint8_t* px = d->tex_rgb8[i];
        for (size_t k = 0; k < n; ++k) texels.push_back(make_uchar4(px[3 * k], px[3 * k + 1], px[3 * k + 2], 255));
    }
    std::vector<float4>& lights = f.lights;
    lights.assign(d->n_lights, make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t i = 0; i < d->n_lights; ++i)
        lights[i] = make_float4((float)d->light_point[3 * i], (float)d->light_point[3 * i + 1], (float)d->light_point[3 * i + 2], 0.f);

    f.n_prims = n_prims;
    f.n_leaf_prims = n_leaf_prims;
    f.n_nodes = (uint32_t)bvh.nodes.size();
    f.n_planes = d->n_planes;
    f.n_tris = (uint32_t)d->n_triangles;
    f.n_lights = d->n_lights;
    for (int k = 0; k < 3; ++k) {
        f.bg[k] = (float)d->background[k];
        f.eye[k] = (float)d->cam_eye[k];
        f.eye64[k] = d->cam_eye[k];
        f.lle[k] = d->cam_lower_left[k] - d->cam_eye[k];
        f.ll[k] = d->cam_lower_left[k];
        f.hz[k] = d->cam_horizontal[k];
        f.vt[k] = d->cam_vertical[k];
    }
    f.width = d->width; f.height = d->height;
    f.stack_depth = std::max<uint32_t>(bvh.max_stack, 2);
    f.bvh_depth = bvh.max_depth;
    f.bvh_build_ms = bvh.build_ms;
    return 0;
}

// copies a flattened scene to the context's GPU
static int upload_flat(rt_context* ctx, const FlatScene& f) {
    DeviceGuard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    free_scene(ctx);
    DevScene sc{};
    if (upload(ctx, f.nodes, &sc.nodes) || upload(ctx, f.prims, &sc.prims) || upload(ctx, f.meta, &sc.meta) || upload(ctx, f.smeta, &sc.smeta) ||
        upload(ctx, f.uvs, &sc.uvs) || upload(ctx, f.planes, &sc.planes) || upload(ctx, f.plane_obj, &sc.plane_object) || upload(ctx, f.planes64, &sc.planes64) || upload(ctx, f.mat_param64, &sc.mat_param64) ||
        upload(ctx, f.mats, &sc.materials) || upload(ctx, f.textures, &sc.textures) || upload(ctx, f.texels, &sc.texels) ||
        upload(ctx, f.lights, &sc.lights)) {
        free_scene(ctx);
        return 1;
    }
    sc.n_prims = f.n_prims;
    sc.n_leaf_prims = f.n_leaf_prims;
    sc.n_nodes = f.n_nodes;
    sc.n_planes = f.n_planes;
    sc.n_tris = f.n_tris;
    sc.n_lights = f.n_lights;
    sc.no_emitters = f.no_emitters && ctx->any_hit_last ? 1u : 0u;
    for (int k = 0; k < 3; ++k) { sc.bg[k] = f.bg[k]; sc.eye[k] = f.eye[k]; sc.eye64[k] = f.eye64[k]; sc.lle[k] = f.lle[k]; sc.ll[k] = f.ll[k]; sc.hz[k] = f.hz[k]; sc.vt[k] = f.vt[k]; }
    sc.width = f.width; sc.height = f.height;
    sc.stack_depth = f.stack_depth;
    sc.spill_levels = sc.stack_depth > LDS_STACK ? sc.stack_depth - LDS_STACK : 0;
    {   // overflow area of the LDS traversal stacks: one slice per persistent closest-hit wave
        const size_t waves = (size_t)ctx->cfg.n_cu * HIT_BLOCKS_PER_CU * 4;
        const size_t bytes = std::max<size_t>(16, waves * 64 * sc.spill_levels * sizeof(uint32_t));
        void* sp = nullptr;
        hipError_t e = hipMalloc(&sp, bytes);
        if (e != hipSuccess) { free_scene(ctx); return fail(ctx, "hipMalloc of the stack overflow area failed: %s", hipGetErrorString(e)); }
        ctx->scene_allocs.push_back(sp);
        sc.spill = reinterpret_cast<uint32_t*>(sp);
    }
    ctx->sc = sc;
    ctx->bvh_nodes = sc.n_nodes; ctx->bvh_depth = f.bvh_depth; ctx->bvh_prims = f.n_prims; ctx->bvh_build_ms = f.bvh_build_ms;
    ctx->have_scene = true;
    return 0;
}

extern "C" int rt_scene_upload_devices(rt_context* const* ctxs, uint32_t n, const rt_scene_desc* d) {
    if (!ctxs || n == 0 || !ctxs[0]) return fail(nullptr, "ctxs is NULL or empty");
    for (uint32_t k = 0; k < n; ++k) if (!ctxs[k]) return fail(ctxs[0], "context %u is NULL", k);
    if (!d) return fail(ctxs[0], "scene is NULL");
    try {
        std::string err;
        if (validate_desc(d, err)) { for (uint32_t k = 1; k < n; ++k) ctxs[k]->error = err; return fail(ctxs[0], "%s", err.c_str()); }
        FlatScene f;
        if (flatten_scene(ctxs[0], d, f, err)) return fail(ctxs[0], "%s", err.c_str());
        for (uint32_t k = 0; k < n; ++k)
            if (upload_flat(ctxs[k], f)) {
                if (k) fail(ctxs[0], "context %u: %s", k, ctxs[k]->error.c_str());
                // all or nothing: no context of the group is left with the new scene while another holds the old one
                for (uint32_t j = 0; j < n; ++j) { DeviceGuard g(ctxs[j]->device); (void)hipDeviceSynchronize(); free_scene(ctxs[j]); }
                return 1;
            }
        return 0;
    } catch (const std::exception& e) {   // bad_alloc / length_error of the host-side staging: no exception crosses the ABI
        return fail(ctxs[0], "scene upload failed: %s", e.what());
    }
}

extern "C" int rt_scene_upload(rt_context* ctx, const rt_scene_desc* d) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    rt_context* one[1] = {ctx};
    return rt_scene_upload_devices(one, 1, d);
}

extern "C" {

// Wavefront state of one pass: the context's grown-on-demand buffers, or carved out of the static scratch array.
struct WaveState {
    RayQueue q[2];
    float2* hits;
    uint32_t* heavy;     // NULL: every iteration is shaded in one launch (the no-alloc entry point: the static scratch has no room for the list)
    float4* radiance;
    uint32_t* ctl;
    uint32_t capacity;   // rays per queue
};

// rays a queue must hold per path of a pass: on the last bounce a firing path spawns one shadow ray per
// light (raytrace.rs:78-85)
static inline uint32_t rays_per_path(const DevScene& sc) { return std::max<uint32_t>(1u, sc.n_lights); }

static inline size_t align256(size_t v) { return (v + 255u) & ~(size_t)255u; }

// bytes of static scratch a pass of `paths` paths needs (rtamd.h: RT_NOALLOC_SCRATCH_BYTES)
static size_t noalloc_bytes(uint32_t paths, uint32_t per_path) {
    const size_t cap = (size_t)paths * per_path;
    return align256(CTL_WORDS * sizeof(uint32_t)) + 6 * align256(cap * sizeof(float4)) + 4 * align256(cap * sizeof(uint4)) +
           align256(cap * sizeof(float2)) + align256((size_t)paths * sizeof(float4));
}

// One call = raytrace.rs:126-143 for a sample range and a tile set (see rtamd.h).
// use_scratch: the no-alloc entry point -- one pass, state in the static scratch array, nothing allocated.
static int render_accumulate_impl(rt_context* ctx, const rt_render_params* p, float* accum_dev, hipStream_t st, bool hits_only,
                                  PassParams* pp_out, WaveState* ws_out, bool use_scratch = false) {
    uint32_t sb, se, tcount;
    if (check_params(ctx, p, &sb, &se, &tcount)) return 1;
    if (!accum_dev && !hits_only) return fail(ctx, "accum_dev is NULL");
    const DevScene& sc = ctx->sc;
    const uint32_t tiles_x = (sc.width + 7) / 8, tiles_y = (sc.height + 7) / 8;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    const uint32_t n_local = p->tile_index < n_tiles ? (uint32_t)((n_tiles - p->tile_index + tcount - 1) / tcount) : 0;
    if (n_local == 0 || sb == se) return 0;
    const uint32_t per_path = rays_per_path(sc);
    const uint32_t gens = sc.n_lights ? (p->max_shadow_generations ? p->max_shadow_generations : 16u) : 0u;
    if (p->depth + gens + 1 > (uint32_t)CTL_MAX_IT) return fail(ctx, "depth + shadow generations exceeds %d", CTL_MAX_IT - 1);
    const bool counters = (p->flags & RT_FLAG_COUNTERS) != 0, timing = (p->flags & RT_FLAG_TIMING) != 0;

    uint32_t tiles_pass, samples_pass;
    WaveState ws{};
    if (use_scratch) {
        // no allocation of any kind on this path: no hipMalloc, no event creation, no growing vectors
        if (timing) return fail(ctx, "RT_FLAG_TIMING is not available without allocation (it creates events)");
        if (!ctx->scratch_ptr) return fail(ctx, "no scratch array on device %d", ctx->device);
        const uint64_t paths64 = (uint64_t)n_local * 64u * (se - sb);
        if (p->max_paths_per_pass && p->max_paths_per_pass < paths64)
            return fail(ctx, "the call has %llu paths but max_paths_per_pass = %u: the no-alloc entry point renders ONE pass", (unsigned long long)paths64,
                        p->max_paths_per_pass);
        if (paths64 > 0xFFFFFFFFull / per_path || noalloc_bytes((uint32_t)paths64, per_path) > (size_t)RT_NOALLOC_SCRATCH_BYTES)
            return fail(ctx, "the call has %llu paths (tiles x 64 x samples) with %u queue slots each: more than the %u MiB static scratch holds "
                             "(%llu bytes needed); split the samples or the tiles over several calls",
                        (unsigned long long)paths64, per_path, (unsigned)(RT_NOALLOC_SCRATCH_BYTES >> 20),
                        (unsigned long long)(paths64 > 0xFFFFFFFFull / per_path ? ~0ull : noalloc_bytes((uint32_t)paths64, per_path)));
        tiles_pass = n_local; samples_pass = se - sb;
        const uint32_t paths = (uint32_t)paths64;
        const size_t cap = (size_t)paths * per_path;
        unsigned char* base = static_cast<unsigned char*>(ctx->scratch_ptr);   // the runtime's address of `scratch` on this GPU, never the symbol
        size_t off = 0;
        auto carve = [&](size_t bytes) { unsigned char* r = base + off; off += align256(bytes); return r; };
        ws.ctl = reinterpret_cast<uint32_t*>(carve(CTL_WORDS * sizeof(uint32_t)));
        for (int k = 0; k < 2; ++k) {
            ws.q[k].a = reinterpret_cast<float4*>(carve(cap * sizeof(float4)));
            ws.q[k].b = reinterpret_cast<float4*>(carve(cap * sizeof(float4)));
            ws.q[k].c = reinterpret_cast<float4*>(carve(cap * sizeof(float4)));
            ws.q[k].e = reinterpret_cast<uint4*>(carve(cap * sizeof(uint4)));
            ws.q[k].f = reinterpret_cast<uint4*>(carve(cap * sizeof(uint4)));
        }
        ws.hits = reinterpret_cast<float2*>(carve(cap * sizeof(float2)));
        ws.radiance = reinterpret_cast<float4*>(carve((size_t)paths * sizeof(float4)));
        ws.capacity = (uint32_t)cap;
    } else {
        // Few, large launches: fill and drain of the persistent waves and the tail of every launch are paid per launch, so a pass is
        // as large as the GPU's memory allows -- same-box, C5's 512-spp step of 1.06 G paths: 192.7 ms in eight passes of 128 Mi paths,
        // 188.2 in four, 184.9 in two, 178.2 in ONE pass (197 GB of wavefront state out of 288 GB); C4's 4K frame 107.7 -> 99.2 ms in one
        // pass instead of four.  The default pass is therefore what fits in 70 % of the device memory that is free now (plus what this
        // context already holds), at most 2^30 paths and 2^31 queue slots (32-bit slot arithmetic); scenes with many lights need one
        // queue slot per light and path.
        uint32_t max_paths = p->max_paths_per_pass;
        if (!max_paths) {
            const uint64_t call_paths = (uint64_t)n_local * 64u * (se - sb);
            const uint64_t hard = std::min<uint64_t>(1ull << 30, (1ull << 31) / per_path);
            if (std::min(call_paths, hard) <= ctx->path_capacity && std::min(call_paths, hard) * per_path <= ctx->capacity) {
                max_paths = (uint32_t)std::min<uint64_t>(hard, ctx->path_capacity);     // what the context holds already is enough: no query
            } else {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = (size_t)64 << 30; (void)hipGetLastError(); }
                const double per_slot = 6.0 * sizeof(float4) + 4.0 * sizeof(uint4) + sizeof(float2) + (ctx->shade_split ? sizeof(uint32_t) : 0u);
                const double held = (double)ctx->capacity * per_slot + (double)ctx->path_capacity * sizeof(float4);
                const double budget = 0.7 * ((double)free_b + held);
                const double fit = budget / (per_slot * per_path + sizeof(float4));
                max_paths = (uint32_t)std::min<double>((double)hard, std::max<double>((double)(1u << 20), fit));
            }
        }
        max_paths = std::max<uint32_t>(64, max_paths & ~63u);
        // tiles per pass / samples per pass.  With the default pass size a device that cannot hold the
        // queues after all (16 + 168 * max(1, lights) bytes per path: 184 B) gets smaller passes instead of an error.
        uint64_t cap64;
        for (;;) {
            tiles_pass = (uint32_t)std::min<uint64_t>(n_local, max_paths / 64);
            samples_pass = std::max<uint32_t>(1, std::min<uint32_t>(se - sb, max_paths / (tiles_pass * 64)));
            const uint32_t pass_paths = tiles_pass * 64 * samples_pass;
            cap64 = (uint64_t)pass_paths * per_path;
            if (cap64 > 0xFFFFFF00ull) {
                if (max_paths > 64u && (!p->max_paths_per_pass || per_path > 1u)) { max_paths = std::max(64u, (max_paths >> 1) & ~63u); continue; }
                return fail(ctx, "pass too large");
            }
            if (ensure_state(ctx, pass_paths, (uint32_t)cap64) == 0) break;
            free_state(ctx);
            (void)hipGetLastError();
            if (p->max_paths_per_pass || max_paths <= (1u << 20)) return 1;
            max_paths >>= 1;
        }
        ws.q[0] = ctx->q[0]; ws.q[1] = ctx->q[1]; ws.hits = ctx->hits; ws.heavy = ctx->heavy; ws.radiance = ctx->radiance; ws.ctl = ctx->ctl;
        // the capacity of THIS call, not of the (possibly larger) buffers an earlier call left behind: whether rays
        // are dropped must not depend on the context's history
        ws.capacity = (uint32_t)cap64;
    }
    const uint32_t capacity = ws.capacity;
    if (ws_out) *ws_out = ws;

    for (uint32_t lt0 = 0; lt0 < n_local; lt0 += tiles_pass) {
        const uint32_t nt = std::min(tiles_pass, n_local - lt0);
        for (uint32_t s0 = sb; s0 < se; s0 += samples_pass) {
            const uint32_t ns = std::min(samples_pass, se - s0);
            PassParams pp{};
            pp.npix_pad = nt * 64; pp.n_paths = pp.npix_pad * ns; pp.ltile_begin = lt0;
            pp.tiles_x = tiles_x; pp.tiles_y = tiles_y; pp.tile_index = p->tile_index; pp.tile_count = tcount;
            pp.s_begin = s0; pp.n_samples = ns; pp.ssaa_n = p->ssaa_n; pp.jitter = p->ssaa_mode == RT_SSAA_JITTER;
            pp.step = 1.0f / (float)p->ssaa_n;                         // anti_aliasing.rs:94
            pp.step64 = 1.0 / (double)p->ssaa_n;
            pp.off_x = (float)p->sample_offset[0]; pp.off_y = (float)p->sample_offset[1];
            pp.seed_lo = (uint32_t)p->seed; pp.seed_hi = (uint32_t)(p->seed >> 32);
            // scenes without lights: every path stores its radiance exactly once (k_shade), no clear needed
            pp.inv_npix_pad = 1.0 / (double)pp.npix_pad; pp.inv_tiles_x = 1.0 / (double)tiles_x;
            pp.clear_radiance = (sc.n_lights > 0 || hits_only || p->depth == 0) ? 1u : 0u;
            if (pp_out) *pp_out = pp;
            HIP_TRY(ctx, hipMemsetAsync(ws.ctl, 0, CTL_WORDS * sizeof(uint32_t), st));
            // iteration 0 forms the primary rays itself (k_closest_hit<.., GEN>, k_shade<.., 2>) unless something needs what k_raygen
            // leaves behind: the queue words (rt_primary_hits / rt_primary_rays) or cleared radiance words (scenes with lights)
            const bool gen = ctx->gen_primary && !hits_only && p->depth > 0 && !pp.clear_radiance;
            // entry points of the tiles' primary rays into the BVH (k_tile_entry): 4 or 8 words per tile of the pass, kept in the first
            // words of queue 1's word f, which nothing touches before k_shade of iteration 0 writes the rays of iteration 1
            TileEntry entry{nullptr, pp.npix_pad, pp.inv_npix_pad};
            const bool with_entry = ctx->tile_entry && sc.n_leaf_prims && (hits_only || p->depth > 0);
            if (with_entry) entry.table = reinterpret_cast<const int4*>(ws.q[1].f);
            if (gen) {
                launch_pass_begin(ws.ctl, pp.n_paths, st);
                if (with_entry) launch_tile_entry(sc, pp, reinterpret_cast<int4*>(ws.q[1].f), st);
            } else if (with_entry) {
                launch_raygen_entry(ctx->cfg, sc, pp, ws.q[0], ws.radiance, ws.ctl, reinterpret_cast<int4*>(ws.q[1].f), st);   // both in one launch
            } else {
                launch_raygen(ctx->cfg, sc, pp, ws.q[0], ws.radiance, ws.ctl, capacity, st);
            }
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
                RayQueue qi = ws.q[it & 1], qo = ws.q[(it + 1) & 1];
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
                // 0: a bounce before the last; 1: last bounce or a shadow generation, shadow rays go to the next
                // generation; 2: the last generation this call runs (max_shadow_generations): nothing is emitted
                const int last = it + 1 < depth ? 0 : (it + 1 < iters ? 1 : 2);
                // the paths' last rays in a scene without emitters only ask WHETHER they hit (see k_closest_hit)
                const bool any_hit = !hits_only && last != 0 && sc.no_emitters;
                launch_closest_hit(ctx->cfg, sc, qi, ws.hits, ws.ctl, (int)it, capacity, max_rays, counters, it == 0 ? entry : TileEntry{nullptr, 0u, 0.0}, any_hit, gen && it == 0 ? &pp : nullptr, st);
                if (timing) {
                    HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used].second, st));
                    ctx->events_used++;
                }
                if (hits_only) break;
                // scenes with triangles, the bounces between the first and the last: the hits on triangles are shaded by a launch of
                // their own (k_shade, CLS)
                if (ctx->shade_split && sc.n_tris && it > 0 && last == 0 && ws.heavy) {
                    launch_shade(ctx->cfg, sc, qi, ws.hits, qo, ws.radiance, ws.ctl, (int)it, last, capacity, max_rays, nullptr, 1, ws.heavy, st);
                    launch_shade(ctx->cfg, sc, qi, ws.hits, qo, ws.radiance, ws.ctl, (int)it, last, capacity, max_rays, nullptr, 2, ws.heavy, st);
                } else {
                    launch_shade(ctx->cfg, sc, qi, ws.hits, qo, ws.radiance, ws.ctl, (int)it, last, capacity, max_rays, gen && it == 0 ? &pp : nullptr, 0, nullptr, st);
                }
            }
            if (!hits_only) launch_accumulate(sc, pp, ws.radiance, accum_dev, st);
            launch_stats(ws.ctl, ctx->stats_dev, capacity, st);
            HIP_TRY(ctx, hipGetLastError());
            if (hits_only) return 0;   // single pass by construction
        }
    }
    return 0;
}

int rt_render_accumulate(rt_context* ctx, const rt_render_params* params, float* accum_dev, void* stream) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    DeviceGuard g(ctx->device);
    try {
        return render_accumulate_impl(ctx, params, accum_dev, (hipStream_t)stream, false, nullptr, nullptr);
    } catch (const std::exception& e) {
        return fail(ctx, "render failed: %s", e.what());
    }
}

int rt_render_accumulate_noalloc(rt_context* ctx, const rt_render_params* params, float* accum_dev, void* stream) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    DeviceGuard g(ctx->device);
    return render_accumulate_impl(ctx, params, accum_dev, (hipStream_t)stream, false, nullptr, nullptr, true);
}

int rt_resolve(rt_context* ctx, const float* accum_dev, uint32_t spp_total, uint8_t* rgb8_dev, float* color_dev, void* stream) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, "no scene uploaded");
    if (!accum_dev) return fail(ctx, "accum_dev is NULL");
    if (spp_total == 0) return fail(ctx, "spp_total must be non-zero");
    DeviceGuard g(ctx->device);
    const size_t n = (size_t)ctx->sc.width * ctx->sc.height * 3u;
    launch_resolve(accum_dev, n, 1.0f / (float)spp_total, rgb8_dev, color_dev, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// rays dropped so far because a queue was full (only the branching shadow-ray process of scenes with four
// or more lights can do that, raytrace.rs:78-85); they make an image wrong, so the synchronous calls report them
static unsigned long long dropped_rays(rt_context* ctx) {
    unsigned long long v = 0;
    if (ctx->stats_dev) (void)hipMemcpy(&v, ctx->stats_dev + CTL_MAX_IT, sizeof(v), hipMemcpyDeviceToHost);
    return v;
}
static int report_dropped(rt_context* ctx, unsigned long long lost) {
    return fail(ctx, "%llu rays were dropped: the shadow-ray queue overflowed.  The scene has %u lights: from 4 lights on every shadow ray spawns "
                     "more than one successor on average (raytrace.rs:78-85) and the reference itself recurses without bound; "
                     "set max_shadow_generations to cut the process off earlier",
                lost, ctx->sc.n_lights);
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
    const unsigned long long dropped_before = dropped_rays(ctx);
    try {
        do {
            if (hipMemsetAsync(accum, 0, n * sizeof(float), st) != hipSuccess) { rc = fail(ctx, "hipMemsetAsync failed"); break; }
            if ((rc = render_accumulate_impl(ctx, params, accum, st, false, nullptr, nullptr))) break;
            if (rgb8_host) {
                if (hipMalloc((void**)&rgb8, n) != hipSuccess) { rc = fail(ctx, "hipMalloc failed"); break; }
                // raytrace.rs:144: divide by the number of samples of the pixel (here: of the range traced)
                if ((rc = rt_resolve(ctx, accum, se - sb ? se - sb : 1, rgb8, nullptr, st))) break;
            }
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { rc = fail(ctx, "render failed: %s", hipGetErrorString(e)); break; }
            if (accum_host && hipMemcpy(accum_host, accum, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, "copy failed"); break; }
            if (rgb8_host && hipMemcpy(rgb8_host, rgb8, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, "copy failed"); break; }
            const unsigned long long lost = dropped_rays(ctx) - dropped_before;
            if (lost) rc = report_dropped(ctx, lost);
        } while (0);
    } catch (const std::exception& e) {
        rc = fail(ctx, "render failed: %s", e.what());
    }
    (void)hipFree(accum);
    (void)hipFree(rgb8);
    return rc;
}

// shared by rt_primary_hits and rt_primary_rays: one uniform-grid sample per pixel, closest hit only
static int primary_pass(rt_context* ctx, double dx, double dy, int32_t* object_id, int32_t* triangle_id, float* t, float* normal_xyz, float* dir_xyz) {
    DeviceGuard g(ctx->device);
    const size_t npx = (size_t)ctx->sc.width * ctx->sc.height;
    const uint32_t tiles = ((ctx->sc.width + 7) / 8) * ((ctx->sc.height + 7) / 8);
    rt_render_params p{};
    p.ssaa_mode = RT_SSAA_UNIFORM; p.ssaa_n = 1; p.depth = 1; p.tile_count = 1;
    p.max_paths_per_pass = tiles * 64;   // one pass over every tile
    p.sample_offset[0] = dx; p.sample_offset[1] = dy;
    int32_t *d_obj = nullptr, *d_tri = nullptr;
    float *d_t = nullptr, *d_n = nullptr, *d_d = nullptr;
    int rc = 0;
    hipStream_t st = ctx->own_stream;
    try {
        do {
            if (object_id && (hipMalloc((void**)&d_obj, npx * 4) != hipSuccess || hipMalloc((void**)&d_tri, npx * 4) != hipSuccess ||
                              hipMalloc((void**)&d_t, npx * 4) != hipSuccess || hipMalloc((void**)&d_n, npx * 12) != hipSuccess)) { rc = fail(ctx, "hipMalloc failed"); break; }
            if (dir_xyz && hipMalloc((void**)&d_d, npx * 12) != hipSuccess) { rc = fail(ctx, "hipMalloc failed"); break; }
            PassParams pp{};
            WaveState ws{};
            if ((rc = render_accumulate_impl(ctx, &p, nullptr, st, true, &pp, &ws))) break;
            launch_hit_info(ctx->sc, pp, ws.q[0], ws.hits, ws.ctl, ws.capacity, d_obj, d_tri, d_t, d_n, d_d, st);
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { rc = fail(ctx, "primary hits failed: %s", hipGetErrorString(e)); break; }
            if (object_id && (hipMemcpy(object_id, d_obj, npx * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(triangle_id, d_tri, npx * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(t, d_t, npx * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(normal_xyz, d_n, npx * 12, hipMemcpyDeviceToHost) != hipSuccess)) { rc = fail(ctx, "copy failed"); break; }
            if (dir_xyz && hipMemcpy(dir_xyz, d_d, npx * 12, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, "copy failed"); break; }
        } while (0);
    } catch (const std::exception& e) {
        rc = fail(ctx, "primary hits failed: %s", e.what());
    }
    (void)hipFree(d_obj); (void)hipFree(d_tri); (void)hipFree(d_t); (void)hipFree(d_n); (void)hipFree(d_d);
    return rc;
}

int rt_primary_hits(rt_context* ctx, double dx, double dy, int32_t* object_id, int32_t* triangle_id, float* t, float* normal_xyz) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, "no scene uploaded");
    if (!object_id || !triangle_id || !t || !normal_xyz) return fail(ctx, "output pointer is NULL");
    return primary_pass(ctx, dx, dy, object_id, triangle_id, t, normal_xyz, nullptr);
}

int rt_primary_rays(rt_context* ctx, double dx, double dy, float origin_xyz[3], float* dir_xyz) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, "no scene uploaded");
    if (!origin_xyz || !dir_xyz) return fail(ctx, "output pointer is NULL");
    for (int k = 0; k < 3; ++k) origin_xyz[k] = ctx->sc.eye[k];
    return primary_pass(ctx, dx, dy, nullptr, nullptr, nullptr, nullptr, dir_xyz);
}

// ---------------------------------------------------------------- multi-GPU exchange step inside the library
static int ensure_values(rt_context* ctx, float** buf, size_t* have, size_t want) {
    if (*have >= want) return 0;
    (void)hipFree(*buf);
    *buf = nullptr; *have = 0;
    HIP_TRY(ctx, hipMalloc((void**)buf, want * sizeof(float)));
    *have = want;
    return 0;
}

int rt_accum_reduce(rt_context* dst, float* dst_accum_dev, rt_context* const* src, const float* const* src_accum_dev, uint32_t n_src, void* stream) {
    if (!dst) return fail(nullptr, "dst is NULL");
    if (!dst->have_scene) return fail(dst, "no scene uploaded");
    if (!dst_accum_dev || (n_src && (!src || !src_accum_dev))) return fail(dst, "NULL argument");
    DeviceGuard g(dst->device);
    const size_t n = (size_t)dst->sc.width * dst->sc.height * 3;
    hipStream_t st = (hipStream_t)stream;
    for (uint32_t k = 0; k < n_src; ++k) {
        if (!src[k] || !src_accum_dev[k]) return fail(dst, "source %u is NULL", k);
        if (src_accum_dev[k] == dst_accum_dev) return fail(dst, "source %u is the destination itself", k);
        if (!src[k]->have_scene || src[k]->sc.width != dst->sc.width || src[k]->sc.height != dst->sc.height)
            return fail(dst, "source %u renders a different frame size", k);
        const float* from = src_accum_dev[k];
        if (src[k]->device != dst->device) {
            // partial frame of another GPU: one peer copy over xGMI into the landing buffer, then the add
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, dst->device, src[k]->device);
            if (can) {
                hipError_t pe = hipDeviceEnablePeerAccess(src[k]->device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();   // hipErrorPeerAccessAlreadyEnabled is fine; without peer access the copy is staged by the runtime
            }
            if (ensure_values(dst, &dst->staging, &dst->staging_values, n)) return 1;
            HIP_TRY(dst, hipMemcpyPeerAsync(dst->staging, dst->device, from, src[k]->device, n * sizeof(float), st));
            from = dst->staging;
        }
        launch_accum_add(dst->cfg, dst_accum_dev, from, n, st);   // stream order also protects the landing buffer's reuse
        HIP_TRY(dst, hipGetLastError());
    }
    return 0;
}

// communicator over the GPUs of ctxs[0..n), one rank per context, kept by ctxs[0] until the set of GPUs changes
static int ensure_comms(rt_context* root, rt_context* const* ctxs, uint32_t n) {
    std::vector<int> devs(n);
    for (uint32_t k = 0; k < n; ++k) devs[k] = ctxs[k]->device;
    if (root->comm_devices == devs && root->comms.size() == n) return 0;
    for (ncclComm_t c : root->comms) (void)ncclCommDestroy(c);
    root->comms.clear(); root->comm_devices.clear();
    std::vector<ncclComm_t> comms(n, nullptr);
    const ncclResult_t r = ncclCommInitAll(comms.data(), (int)n, devs.data());
    if (r != ncclSuccess) return fail(root, "ncclCommInitAll over %u GPUs failed: %s", n, ncclGetErrorString(r));
    root->comms = comms; root->comm_devices = devs;
    return 0;
}

int rt_accum_reduce_rccl(rt_context* const* ctxs, float* const* accum_dev, uint32_t n) {
    if (!ctxs || n == 0 || !ctxs[0]) return fail(nullptr, "ctxs is NULL or empty");
    rt_context* root = ctxs[0];
    if (!accum_dev) return fail(root, "accum_dev is NULL");
    if (!root->have_scene) return fail(root, "no scene uploaded");
    for (uint32_t k = 0; k < n; ++k) {
        if (!ctxs[k] || !accum_dev[k]) return fail(root, "context or frame %u is NULL", k);
        if (!ctxs[k]->have_scene || ctxs[k]->sc.width != root->sc.width || ctxs[k]->sc.height != root->sc.height)
            return fail(root, "context %u renders a different frame size", k);
        for (uint32_t j = 0; j < k; ++j)
            if (ctxs[j]->device == ctxs[k]->device)
                return fail(root, "contexts %u and %u are on the same GPU (%d): an RCCL communicator has one rank per GPU; use rt_accum_reduce", j, k, ctxs[k]->device);
    }
    try {
        if (ensure_comms(root, ctxs, n)) return 1;
        const size_t count = (size_t)root->sc.width * root->sc.height * 3;
        // ONE reduce(sum, f32) of the frame to the first context's GPU; every rank's part is queued on its context's own stream
        ncclResult_t r = ncclGroupStart();
        for (uint32_t k = 0; k < n && r == ncclSuccess; ++k)
            r = ncclReduce(accum_dev[k], accum_dev[k], count, ncclFloat, ncclSum, 0, root->comms[k], ctxs[k]->own_stream);
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return fail(root, "ncclReduce failed: %s", ncclGetErrorString(r));
        for (uint32_t k = 0; k < n; ++k) {
            DeviceGuard g(ctxs[k]->device);
            HIP_TRY(root, hipStreamSynchronize(ctxs[k]->own_stream));
        }
    } catch (const std::exception& ex) {
        return fail(root, "RCCL reduce failed: %s", ex.what());
    }
    return 0;
}

int rt_render_image_devices(rt_context* const* ctxs, uint32_t n, const rt_render_params* params, int split_tiles, float* accum_host, uint8_t* rgb8_host) {
    if (!ctxs || n == 0 || !ctxs[0]) return fail(nullptr, "ctxs is NULL or empty");
    rt_context* root = ctxs[0];
    for (uint32_t k = 0; k < n; ++k) {
        if (!ctxs[k]) return fail(root, "context %u is NULL", k);
        for (uint32_t j = 0; j < k; ++j) if (ctxs[j] == ctxs[k]) return fail(root, "context %u is listed twice", k);
    }
    uint32_t sb, se, tcount;
    if (check_params(root, params, &sb, &se, &tcount)) return 1;
    const size_t nval = (size_t)root->sc.width * root->sc.height * 3;
    for (uint32_t k = 1; k < n; ++k)
        if (!ctxs[k]->have_scene || ctxs[k]->sc.width != root->sc.width || ctxs[k]->sc.height != root->sc.height)
            return fail(root, "context %u holds a different scene (use rt_scene_upload_devices)", k);
    try {
        // one host thread per GPU, as the contexts are single-threaded objects bound to one device each
        std::vector<int> rcs(n, 0);
        std::vector<unsigned long long> before(n, 0);
        std::vector<std::thread> threads;
        threads.reserve(n);
        auto body = [&](uint32_t k) {
            rt_context* c = ctxs[k];
            try {
                DeviceGuard g(c->device);
                rt_render_params p = *params;
                bool work = true;
                if (split_tiles) {
                    // tiles t with t % tcount == tile_index, dealt round-robin once more between the n devices
                    p.tile_count = tcount * n; p.tile_index = params->tile_index + k * tcount;
                } else {
                    // contiguous sample ranges, the first (count % n) devices take one more (same RNG keys as one device)
                    const uint32_t cnt = se - sb, base = cnt / n, extra = cnt % n;
                    p.sample_begin = sb + k * base + std::min(k, extra);
                    p.sample_end = p.sample_begin + base + (k < extra ? 1u : 0u);
                    work = p.sample_end > p.sample_begin;     // more devices than samples: this one only clears its frame
                }
                before[k] = dropped_rays(c);
                if (ensure_values(c, &c->own_accum, &c->own_accum_values, nval)) { rcs[k] = 1; return; }
                if (hipMemsetAsync(c->own_accum, 0, nval * sizeof(float), c->own_stream) != hipSuccess) { rcs[k] = fail(c, "hipMemsetAsync failed"); return; }
                if (work) {
                    try { rcs[k] = render_accumulate_impl(c, &p, c->own_accum, c->own_stream, false, nullptr, nullptr); }
                    catch (const std::exception& e) { rcs[k] = fail(c, "render failed: %s", e.what()); }
                }
                if (!rcs[k]) {
                    hipError_t e = hipStreamSynchronize(c->own_stream);
                    if (e != hipSuccess) rcs[k] = fail(c, "render failed: %s", hipGetErrorString(e));
                }
            } catch (...) {          // (bad_alloc of an error string included: nothing leaves a thread, std::terminate would cross the ABI)
                rcs[k] = 1;
            }
        };
        try {
            for (uint32_t k = 0; k < n; ++k) threads.emplace_back(body, k);
        } catch (const std::exception& e) {
            for (auto& t : threads) t.join();     // the threads that did start finish before the error is reported
            return fail(root, "could not start a host thread per device: %s", e.what());
        }
        for (auto& t : threads) t.join();
        for (uint32_t k = 0; k < n; ++k)
            if (rcs[k]) return k ? fail(root, "device %d: %s", ctxs[k]->device, ctxs[k]->error.c_str()) : 1;
        // exchange: the partial frames are summed on the first context's GPU (peer copies + adds), resolved there
        // (raytrace.rs:144-150, image.rs:37-43) and copied to the host once
        DeviceGuard g(root->device);
        if (n > 1) {
            // contexts on n distinct GPUs: one RCCL reduce over xGMI; contexts that share a GPU (or a host where the communicator
            // cannot be created): peer copies into a landing buffer + adds on the first GPU
            bool distinct = true;
            for (uint32_t k = 0; k < n; ++k) for (uint32_t j = 0; j < k; ++j) distinct = distinct && ctxs[j]->device != ctxs[k]->device;
            bool done = false;
            if (distinct && !root->rccl_unavailable) {
                std::vector<float*> frames(n);
                for (uint32_t k = 0; k < n; ++k) frames[k] = ctxs[k]->own_accum;
                if (rt_accum_reduce_rccl(ctxs, frames.data(), n) == 0) done = true;
                else if (root->comms.empty()) {
                    root->rccl_unavailable = true;
                    fprintf(stderr, "rtamd: %s; the partial frames are summed over peer copies instead\n", root->error.c_str());
                } else return 1;
            }
            if (!done) {
                std::vector<const float*> parts;
                for (uint32_t k = 1; k < n; ++k) parts.push_back(ctxs[k]->own_accum);
                if (rt_accum_reduce(root, root->own_accum, ctxs + 1, parts.data(), n - 1, root->own_stream)) return 1;
            }
        }
        uint8_t* rgb8 = nullptr;
        int rc = 0;
        do {
            if (rgb8_host) {
                if (hipMalloc((void**)&rgb8, nval) != hipSuccess) { rc = fail(root, "hipMalloc failed"); break; }
                if ((rc = rt_resolve(root, root->own_accum, se - sb ? se - sb : 1, rgb8, nullptr, root->own_stream))) break;
            }
            hipError_t e = hipStreamSynchronize(root->own_stream);
            if (e != hipSuccess) { rc = fail(root, "exchange failed: %s", hipGetErrorString(e)); break; }
            if (accum_host && hipMemcpy(accum_host, root->own_accum, nval * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(root, "copy failed"); break; }
            if (rgb8_host && hipMemcpy(rgb8_host, rgb8, nval, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(root, "copy failed"); break; }
            unsigned long long lost = 0;
            for (uint32_t k = 0; k < n; ++k) { DeviceGuard gk(ctxs[k]->device); lost += dropped_rays(ctxs[k]) - before[k]; }
            if (lost) rc = report_dropped(root, lost);
        } while (0);
        (void)hipFree(rgb8);
        return rc;
    } catch (const std::exception& e) {
        return fail(root, "multi-device render failed: %s", e.what());
    }
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
    try {
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
        std::vector<unsigned long long> h(CTL_MAX_IT + 1 + N_STAT64);
        HIP_TRY(ctx, hipMemcpy(h.data(), ctx->stats_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        h[0] = ctx->rays0_total;
        for (int i = 0; i < CTL_MAX_IT; ++i) { out->rays_per_iteration[i] = h[i]; out->rays_total += h[i]; }
        out->queue_overflow = h[CTL_MAX_IT];
        out->nodes_visited = h[CTL_MAX_IT + 1];
        out->prims_tested = h[CTL_MAX_IT + 2];
        out->wave_node_steps = h[CTL_MAX_IT + 3];
        out->wave_leaf_steps = h[CTL_MAX_IT + 4];
        out->wave_refills = h[CTL_MAX_IT + 5];
        out->self_hit_rays = h[CTL_MAX_IT + 6];
        out->step_lanes = h[CTL_MAX_IT + 7];
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
    if (ctx->stats_dev) HIP_TRY(ctx, hipMemset(ctx->stats_dev, 0, (CTL_MAX_IT + 1 + N_STAT64) * sizeof(unsigned long long)));
    ctx->events_used = 0; ctx->hit_ms = 0; ctx->hit_launches = 0; ctx->paths_total = 0; ctx->rays0_total = 0;
    return 0;
}

}  // extern "C"
