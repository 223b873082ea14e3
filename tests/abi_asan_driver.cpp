// The C ABI end to end from a plain C++ host, with the library's HOST code under AddressSanitizer + UBSan
// (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: device code is not instrumented).  Built by __graft_entry__.build()
// into tests/_build/abi_asan, run on the GPU by tests/test_gpu_extra.py::test_c_abi_host_code_under_asan.
//
// It walks every entry point of include/rtamd.h the way a C / Rust host would (no Python, no torch): upload, render to host
// buffers, hit and ray probes, the device-pointer calls, the entry point that allocates nothing, two contexts on one GPU with
// the on-device sum, re-uploads of scenes of different sizes (buffer reuse), and the error paths (NULL arguments, bad
// parameters, non-finite geometry).  Results are cross-checked against each other (host call == device-pointer call ==
// no-alloc call == two contexts), so a sanitizer-clean run is also a functional one.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <unistd.h>

#include "../include/rtamd.h"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); exit(1); } \
    } while (0)
#define OK(ctx, call)                                                                \
    do {                                                                             \
        if ((call) != 0) { fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call, rt_last_error(ctx)); exit(1); } \
    } while (0)
#define FAILS(ctx, call)                                                             \
    do {                                                                             \
        if ((call) == 0) { fprintf(stderr, "%s:%d: %s should have failed\n", __FILE__, __LINE__, #call); exit(1); } \
        CHECK(rt_last_error(ctx) && rt_last_error(ctx)[0]);                          \
    } while (0)
#define HIP(call) CHECK((call) == hipSuccess)

struct Scene {
    std::vector<int32_t> mat_type, mat_texture, sphere_material, sphere_object, plane_material, plane_object, tri_material, tri_object, tri_index;
    std::vector<double> mat_rgb, mat_param, sphere_center, sphere_radius, plane_center, plane_normal, tri_vertices, tri_uv, light_point;
    std::vector<uint8_t> tri_has_uv, texels;
    std::vector<uint32_t> tex_w, tex_h;
    std::vector<const uint8_t*> tex_ptr;
    rt_scene_desc d;
};

static uint64_t lcg = 12345;
static double urand() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) * (1.0 / 9007199254740992.0); }

// camera.rs:26-54
static void camera(rt_scene_desc& d, const double eye[3], const double at[3], double fovy) {
    const double up[3] = {0, 1, 0};
    auto sub = [](const double* a, const double* b, double* o) { for (int k = 0; k < 3; ++k) o[k] = a[k] - b[k]; };
    auto norm = [](double* a) { const double l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); for (int k = 0; k < 3; ++k) a[k] /= l; };
    auto cross = [](const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    double w[3], u[3], v[3];
    sub(eye, at, w); norm(w);
    cross(up, w, u); norm(u);
    cross(w, u, v);
    const double h = std::tan(fovy * M_PI / 180.0 / 2.0), vh = 2.0 * h, vw = vh * (double)d.width / (double)d.height;
    for (int k = 0; k < 3; ++k) {
        d.cam_eye[k] = eye[k];
        d.cam_horizontal[k] = u[k] * vw / d.width;
        d.cam_vertical[k] = v[k] * vh / d.height;
        d.cam_lower_left[k] = eye[k] - u[k] * vw / 2 - v[k] * vh / 2 - w[k];
    }
}

// n_spheres small spheres + one big one, a plane, a strip of n_tris textured / plain triangles, n_lights lights
static void make_scene(Scene& s, uint32_t w, uint32_t h, int n_spheres, int n_tris, int n_lights, bool plane) {
    s = Scene();
    memset(&s.d, 0, sizeof s.d);
    lcg = 12345 + (uint64_t)n_spheres * 7919u + (uint64_t)n_tris;   // the same arguments give the same scene
    auto mat = [&](int type, double r, double g, double b, double p, int tex) {
        s.mat_type.push_back(type); s.mat_rgb.insert(s.mat_rgb.end(), {r, g, b}); s.mat_param.push_back(p); s.mat_texture.push_back(tex);
        return (int32_t)s.mat_type.size() - 1;
    };
    const int32_t lam = mat(RT_MAT_LAMBERTIAN, 0.6, 0.5, 0.4, 0, -1), met = mat(RT_MAT_METAL, 0.8, 0.8, 0.9, 0.2, -1), die = mat(RT_MAT_DIELECTRIC, 1, 1, 1, 1.5, -1),
                  tex = mat(RT_MAT_TEXTURE, 0, 0, 0, 0, 0), emi = mat(RT_MAT_EMISSIVE, 2.0, 1.8, 1.5, 0, -1);
    s.tex_w = {4}; s.tex_h = {3};
    for (int i = 0; i < 4 * 3 * 3; ++i) s.texels.push_back((uint8_t)(40 + 5 * i));
    s.tex_ptr = {s.texels.data()};
    int32_t obj = 0;
    for (int i = 0; i < n_spheres; ++i) {
        s.sphere_center.insert(s.sphere_center.end(), {6.0 * urand() - 3.0, 0.3 + urand(), 6.0 * urand() - 3.0});
        s.sphere_radius.push_back(0.15 + 0.3 * urand());
        s.sphere_material.push_back(i % 7 == 0 ? emi : (i % 3 == 0 ? met : (i % 3 == 1 ? die : lam)));
        s.sphere_object.push_back(obj++);
    }
    if (n_spheres) {   // a big sphere (kept out of the BVH)
        s.sphere_center.insert(s.sphere_center.end(), {0, -1000.0, 0}); s.sphere_radius.push_back(1000.0); s.sphere_material.push_back(lam); s.sphere_object.push_back(obj++);
    }
    if (plane) {
        s.plane_center.insert(s.plane_center.end(), {0, 6.0, 0}); s.plane_normal.insert(s.plane_normal.end(), {0, -1, 0});
        s.plane_material.push_back(emi); s.plane_object.push_back(obj++);
    }
    for (int i = 0; i < n_tris; ++i) {   // one mesh object: a ribbon
        const double x = -3.0 + 6.0 * (i / 2) / std::max(1, n_tris / 2), dx = 6.0 / std::max(1, n_tris / 2), y0 = 0.2, y1 = 1.4, z = -1.5 + 0.3 * std::sin(0.5 * i);
        if (i % 2 == 0) s.tri_vertices.insert(s.tri_vertices.end(), {x, y0, z, x + dx, y0, z, x, y1, z + 0.2});
        else s.tri_vertices.insert(s.tri_vertices.end(), {x + dx, y0, z, x + dx, y1, z + 0.2, x, y1, z + 0.2});
        s.tri_uv.insert(s.tri_uv.end(), {0.1, 0.1, 0.9, 0.1, 0.1, 0.9});
        s.tri_has_uv.push_back((uint8_t)(i % 4 < 2));
        s.tri_material.push_back(i % 4 < 2 ? tex : lam);
        s.tri_object.push_back(obj); s.tri_index.push_back(i);
    }
    for (int i = 0; i < n_lights; ++i) s.light_point.insert(s.light_point.end(), {2.0 * i - 1.0, 5.0, 1.0});
    rt_scene_desc& d = s.d;
    d.width = w; d.height = h;
    d.background[0] = 0.7; d.background[1] = 0.8; d.background[2] = 1.0;
    const double eye[3] = {0.5, 1.5, 6.0}, at[3] = {0, 0.6, 0};
    camera(d, eye, at, 40.0);
    d.n_materials = (uint32_t)s.mat_type.size(); d.mat_type = s.mat_type.data(); d.mat_rgb = s.mat_rgb.data(); d.mat_param = s.mat_param.data(); d.mat_texture = s.mat_texture.data();
    d.n_textures = 1; d.tex_width = s.tex_w.data(); d.tex_height = s.tex_h.data(); d.tex_rgb8 = s.tex_ptr.data();
    d.n_spheres = (uint32_t)s.sphere_radius.size(); d.sphere_center = s.sphere_center.data(); d.sphere_radius = s.sphere_radius.data();
    d.sphere_material = s.sphere_material.data(); d.sphere_object = s.sphere_object.data();
    d.n_planes = (uint32_t)s.plane_material.size(); d.plane_center = s.plane_center.data(); d.plane_normal = s.plane_normal.data();
    d.plane_material = s.plane_material.data(); d.plane_object = s.plane_object.data();
    d.n_triangles = (uint64_t)s.tri_material.size(); d.tri_vertices = s.tri_vertices.data(); d.tri_uv = s.tri_uv.data(); d.tri_has_uv = s.tri_has_uv.data();
    d.tri_material = s.tri_material.data(); d.tri_object = s.tri_object.data(); d.tri_index = s.tri_index.data();
    d.n_lights = (uint32_t)n_lights; d.light_point = s.light_point.data();
}

static rt_render_params params(uint32_t n, uint32_t depth, uint64_t seed) {
    rt_render_params p;
    memset(&p, 0, sizeof p);
    p.ssaa_mode = RT_SSAA_JITTER; p.ssaa_n = n; p.depth = depth; p.seed = seed;
    return p;
}

static double max_rel_diff(const std::vector<float>& a, const std::vector<float>& b) {
    CHECK(a.size() == b.size());
    double m = 0;
    for (size_t i = 0; i < a.size(); ++i) m = std::max(m, std::fabs((double)a[i] - b[i]) / std::max(1.0, std::fabs((double)a[i])));
    return m;
}

int main() {
    rt_context* ctx = nullptr;
    FAILS(nullptr, rt_context_create(0, nullptr));
    FAILS(nullptr, rt_context_create(12345, &ctx));
    OK(nullptr, rt_context_create(0, &ctx));
    rt_stats st;

    // nothing uploaded yet: every call that needs a scene fails with a message
    rt_render_params p = params(2, 5, 7);
    FAILS(ctx, rt_render_image(ctx, &p, nullptr, nullptr));
    FAILS(ctx, rt_scene_upload(ctx, nullptr));
    FAILS(ctx, rt_scene_upload(nullptr, nullptr));

    Scene s;
    // scenes of different sizes one after the other: device buffers are reused / regrown / shrunk
    const struct { uint32_t w, h; int sph, tri, lights; bool plane; } cases[] = {
        {96, 64, 40, 300, 0, true}, {33, 17, 0, 0, 0, false}, {200, 120, 300, 5000, 2, true}, {8, 8, 1, 1, 1, false}, {96, 64, 40, 300, 0, true}};
    std::vector<float> first_frame;
    for (const auto& c : cases) {
        make_scene(s, c.w, c.h, c.sph, c.tri, c.lights, c.plane);
        OK(ctx, rt_scene_upload(ctx, &s.d));
        const size_t npx = (size_t)c.w * c.h;
        std::vector<float> acc(npx * 3), t(npx), nrm(npx * 3), dir(npx * 3);
        std::vector<uint8_t> rgb(npx * 3);
        std::vector<int32_t> oid(npx), tid(npx);
        p = params(3, 5, 7);
        OK(ctx, rt_stats_reset(ctx));
        OK(ctx, rt_render_image(ctx, &p, acc.data(), rgb.data()));
        OK(ctx, rt_render_image(ctx, &p, nullptr, rgb.data()));
        OK(ctx, rt_render_image(ctx, &p, acc.data(), nullptr));
        OK(ctx, rt_get_stats(ctx, &st));
        CHECK(st.paths == 3 * npx * 9 && st.rays_total >= st.paths && st.queue_overflow == 0);
        float org[3];
        OK(ctx, rt_primary_hits(ctx, 0.5, 0.5, oid.data(), tid.data(), t.data(), nrm.data()));
        OK(ctx, rt_primary_rays(ctx, 0.5, 0.5, org, dir.data()));
        for (size_t i = 0; i < npx; ++i) CHECK((oid[i] == -1) == (t[i] == -1.f));
        if (first_frame.empty()) first_frame = acc;
        else if (&c == &cases[4]) CHECK(max_rel_diff(first_frame, acc) == 0.0);   // same scene, same seed, buffers reused in between: same bits (no lights: no float atomics)

        // device-pointer calls against the host call
        float *acc_dev = nullptr, *col_dev = nullptr;
        uint8_t* rgb_dev = nullptr;
        HIP(hipMalloc((void**)&acc_dev, npx * 3 * sizeof(float)));
        HIP(hipMalloc((void**)&col_dev, npx * 3 * sizeof(float)));
        HIP(hipMalloc((void**)&rgb_dev, npx * 3));
        hipStream_t stream;
        HIP(hipStreamCreate(&stream));
        HIP(hipMemsetAsync(acc_dev, 0, npx * 3 * sizeof(float), stream));
        OK(ctx, rt_render_accumulate(ctx, &p, acc_dev, stream));
        OK(ctx, rt_resolve(ctx, acc_dev, 9, rgb_dev, col_dev, stream));
        OK(ctx, rt_resolve(ctx, acc_dev, 9, nullptr, nullptr, stream));
        HIP(hipStreamSynchronize(stream));
        std::vector<float> acc2(npx * 3);
        std::vector<uint8_t> rgb2(npx * 3);
        HIP(hipMemcpy(acc2.data(), acc_dev, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(rgb2.data(), rgb_dev, npx * 3, hipMemcpyDeviceToHost));
        CHECK(max_rel_diff(acc, acc2) <= (c.lights ? 1e-5 : 0.0));
        if (!c.lights) CHECK(rgb == rgb2);

        // the entry point that allocates nothing: sample by sample and in tile sets so that every call fits the static array
        HIP(hipMemsetAsync(acc_dev, 0, npx * 3 * sizeof(float), stream));
        const uint32_t tile_sets = npx > 20000 ? 4 : 1;
        for (uint32_t smp = 0; smp < 9; ++smp)
            for (uint32_t ts = 0; ts < tile_sets; ++ts) {
                rt_render_params q = p;
                q.sample_begin = smp; q.sample_end = smp + 1; q.tile_index = ts; q.tile_count = tile_sets;
                OK(ctx, rt_render_accumulate_noalloc(ctx, &q, acc_dev, stream));
            }
        HIP(hipStreamSynchronize(stream));
        HIP(hipMemcpy(acc2.data(), acc_dev, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
        CHECK(max_rel_diff(acc, acc2) <= 1e-5);
        rt_render_params big = params(64, 5, 7);
        if (npx * 4096 > 400000) FAILS(ctx, rt_render_accumulate_noalloc(ctx, &big, acc_dev, stream));
        big = p; big.flags = RT_FLAG_TIMING;
        FAILS(ctx, rt_render_accumulate_noalloc(ctx, &big, acc_dev, stream));

        // counters and timing builds, small passes, sample and tile ranges
        rt_render_params q = p;
        q.flags = RT_FLAG_COUNTERS | RT_FLAG_TIMING; q.max_paths_per_pass = 64 * 37;
        OK(ctx, rt_render_image(ctx, &q, acc2.data(), nullptr));
        CHECK(max_rel_diff(acc, acc2) <= 1e-5);
        OK(ctx, rt_get_stats(ctx, &st));
        if (c.sph + c.tri > 1) CHECK(st.nodes_visited > 0);
        CHECK(st.hit_launches > 0);

        // error paths
        q = p; q.ssaa_n = 0; FAILS(ctx, rt_render_image(ctx, &q, acc2.data(), nullptr));
        q = p; q.depth = 256; FAILS(ctx, rt_render_image(ctx, &q, acc2.data(), nullptr));
        q = p; q.sample_begin = 5; q.sample_end = 100; FAILS(ctx, rt_render_image(ctx, &q, acc2.data(), nullptr));
        q = p; q.tile_index = 3; q.tile_count = 3; FAILS(ctx, rt_render_image(ctx, &q, acc2.data(), nullptr));
        FAILS(ctx, rt_render_image(ctx, nullptr, acc2.data(), nullptr));
        FAILS(ctx, rt_render_accumulate(ctx, &p, nullptr, stream));
        FAILS(ctx, rt_resolve(ctx, nullptr, 9, rgb_dev, nullptr, stream));
        FAILS(ctx, rt_resolve(ctx, acc_dev, 0, rgb_dev, nullptr, stream));
        FAILS(ctx, rt_get_stats(ctx, nullptr));
        FAILS(ctx, rt_primary_rays(ctx, 0, 0, nullptr, nullptr));

        // two contexts on this GPU: one flatten + BVH build for both, split render, sum on the device
        rt_context* ctx2 = nullptr;
        OK(nullptr, rt_context_create(0, &ctx2));
        rt_context* both[2] = {ctx, ctx2};
        OK(ctx, rt_scene_upload_devices(both, 2, &s.d));
        for (int split = 0; split < 2; ++split) {
            std::vector<uint8_t> rgb3(npx * 3);
            OK(ctx, rt_render_image_devices(both, 2, &p, split, acc2.data(), rgb3.data()));
            CHECK(max_rel_diff(acc, acc2) <= 1e-5);
        }
        float* acc_dev2 = nullptr;
        HIP(hipMalloc((void**)&acc_dev2, npx * 3 * sizeof(float)));
        HIP(hipMemset(acc_dev, 0, npx * 3 * sizeof(float)));
        HIP(hipMemset(acc_dev2, 0, npx * 3 * sizeof(float)));
        rt_render_params h0 = p, h1 = p;
        h0.sample_end = 4; h1.sample_begin = 4;
        OK(ctx, rt_render_accumulate(ctx, &h0, acc_dev, stream));
        OK(ctx2, rt_render_accumulate(ctx2, &h1, acc_dev2, nullptr));
        HIP(hipDeviceSynchronize());
        rt_context* srcs[1] = {ctx2};
        const float* src_ptrs[1] = {acc_dev2};
        OK(ctx, rt_accum_reduce(ctx, acc_dev, srcs, src_ptrs, 1, stream));
        FAILS(ctx, rt_accum_reduce(ctx, acc_dev, srcs, nullptr, 1, stream));
        const float* alias[1] = {acc_dev};
        FAILS(ctx, rt_accum_reduce(ctx, acc_dev, srcs, alias, 1, stream));
        HIP(hipStreamSynchronize(stream));
        HIP(hipMemcpy(acc2.data(), acc_dev, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
        CHECK(max_rel_diff(acc, acc2) <= 1e-5);
        {   // the same exchange as an RCCL collective: a communicator of one rank leaves the frame as it is; two contexts on one
            // GPU are refused (one rank per GPU)
            rt_context* one[1] = {ctx};
            float* frame[1] = {acc_dev};
            OK(ctx, rt_accum_reduce_rccl(one, frame, 1));
            OK(ctx, rt_accum_reduce_rccl(one, frame, 1));   // (the communicator is kept)
            std::vector<float> acc3(npx * 3);
            HIP(hipMemcpy(acc3.data(), acc_dev, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
            CHECK(acc3 == acc2);
            rt_context* same[2] = {ctx, ctx2};
            float* frames[2] = {acc_dev, acc_dev2};
            FAILS(ctx, rt_accum_reduce_rccl(same, frames, 2));
            FAILS(ctx, rt_accum_reduce_rccl(one, nullptr, 1));
        }
        rt_context_destroy(ctx2);
        HIP(hipFree(acc_dev2));
        HIP(hipFree(acc_dev)); HIP(hipFree(col_dev)); HIP(hipFree(rgb_dev));
        HIP(hipStreamDestroy(stream));
        printf("scene %ux%u, %d spheres, %d triangles, %d lights: ok (%llu rays)\n", c.w, c.h, c.sph, c.tri, c.lights, (unsigned long long)st.rays_total);
        fflush(stdout);
    }

    // bad scenes leave the context usable
    make_scene(s, 64, 64, 10, 20, 0, true);
    s.tri_material[3] = 99;
    FAILS(ctx, rt_scene_upload(ctx, &s.d));
    s.tri_material[3] = 0;
    s.sphere_center[4] = NAN;
    FAILS(ctx, rt_scene_upload(ctx, &s.d));
    s.sphere_center[4] = 0.5;
    s.d.width = 0;
    FAILS(ctx, rt_scene_upload(ctx, &s.d));
    s.d.width = 64;
    s.d.tri_vertices = nullptr;
    FAILS(ctx, rt_scene_upload(ctx, &s.d));
    s.d.tri_vertices = s.tri_vertices.data();
    OK(ctx, rt_scene_upload(ctx, &s.d));

    // the GPU BVH builder through the same calls
    OK(ctx, rt_context_set_bvh_builder(ctx, 1));
    FAILS(ctx, rt_context_set_bvh_builder(ctx, 7));
    make_scene(s, 96, 64, 200, 3000, 0, true);
    OK(ctx, rt_scene_upload(ctx, &s.d));
    std::vector<float> a1((size_t)96 * 64 * 3), a2(a1.size());
    p = params(2, 4, 3);
    OK(ctx, rt_render_image(ctx, &p, a1.data(), nullptr));
    OK(ctx, rt_context_set_bvh_builder(ctx, 0));
    OK(ctx, rt_scene_upload(ctx, &s.d));
    OK(ctx, rt_render_image(ctx, &p, a2.data(), nullptr));
    CHECK(max_rel_diff(a1, a2) <= 1e-4);   // the tree decides nothing but the order of equal-distance candidates
    std::vector<float> boxes;
    for (int i = 0; i < 1000; ++i) { const double x = 10 * urand(), y = 10 * urand(), z = 10 * urand(); boxes.insert(boxes.end(), {(float)x, (float)y, (float)z, (float)(x + urand()), (float)(y + urand()), (float)(z + urand())}); }
    uint32_t nn, dd, ms;
    double bms;
    for (int b = 0; b < 2; ++b) {
        OK(ctx, rt_context_set_bvh_builder(ctx, b));
        OK(ctx, rt_bvh_build_validate(ctx, 1000, boxes.data(), &nn, &dd, &ms, &bms));
        OK(ctx, rt_bvh_build_validate(ctx, 0, nullptr, &nn, &dd, &ms, &bms));
        OK(ctx, rt_bvh_build_validate(ctx, 1, boxes.data(), nullptr, nullptr, nullptr, nullptr));
    }
    FAILS(ctx, rt_bvh_build_validate(ctx, 5, nullptr, &nn, &dd, &ms, &bms));
    double loaded = 0;
    OK(ctx, rt_debug_gather(ctx, 1 << 20, 4, &loaded));
    rt_context_destroy(ctx);
    rt_context_destroy(nullptr);
    printf("abi_asan_driver: all checks passed\n");
    fflush(stdout);
    fflush(stderr);
    _exit(0);   // not through the static destructors: ASan's allocator and the HIP runtime disagree about who is unloaded first
}
