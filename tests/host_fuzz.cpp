// Mutation fuzzer for the host side above the C ABI (rt_host.cpp: YAML subset, OBJ import, PNG decoder; bvh.cpp: builder),
// built with AddressSanitizer + UBSan by tests/test_host_fuzz.py (g++, host code only: no GPU, no HIP).
//
//   host_fuzz <golden dir> <work dir> <iterations> <seed>
//
// Every input the loaders get here is a corrupted copy of a fixture the reference ships (scenes/*/*.yaml, cube.obj,
// cube_texture.png): bytes flipped, ranges cut, lines dropped / doubled / swapped, numbers replaced by nan, inf, 1e308,
// -1, 4294967296, empty.  The contract under test is INTEGRATION.md's: every function returns 0 / non-zero with a message,
// nothing throws, aborts, reads out of bounds or overflows across the ABI.  A loaded scene is then walked through its
// rt_scene_desc (every array read to its declared length) so that a short array shows up as an ASan report here rather
// than as a GPU fault later.  The device entry points rt_host.cpp calls are stubbed below: they are not under test.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include <signal.h>
#include <unistd.h>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#endif

#include "../include/rtamd_host.h"

// watchdog: no single call may take longer than this (a hostile input must not hang the loader either)
static void on_alarm(int) {
    const char msg[] = "host_fuzz: a call ran for more than 20 s; stack of the main thread:\n";
    (void)!write(2, msg, sizeof msg - 1);
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_print_stack_trace();
#endif
    _exit(3);
}

extern "C" {
int rt_context_create(int32_t, rt_context**) { return 1; }
void rt_context_destroy(rt_context*) {}
const char* rt_last_error(const rt_context*) { return "stub"; }
int rt_scene_upload(rt_context*, const rt_scene_desc*) { return 1; }
int rt_scene_upload_devices(rt_context* const*, uint32_t, const rt_scene_desc*) { return 1; }
int rt_render_image(rt_context*, const rt_render_params*, float*, uint8_t*) { return 1; }
int rt_render_image_devices(rt_context* const*, uint32_t, const rt_render_params*, int32_t, float*, uint8_t*) { return 1; }
}

static std::string slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static void spit(const std::string& p, const std::string& s) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write(s.data(), (std::streamsize)s.size());
}

static std::mt19937_64 rng;
static size_t rnd(size_t n) { return n ? (size_t)(rng() % n) : 0; }

static std::vector<std::string> lines_of(const std::string& s) {
    std::vector<std::string> v;
    std::string cur;
    for (char c : s) {
        if (c == '\n') { v.push_back(cur); cur.clear(); } else cur.push_back(c);
    }
    if (!cur.empty()) v.push_back(cur);
    return v;
}
static std::string join(const std::vector<std::string>& v) {
    std::string s;
    for (auto& l : v) { s += l; s.push_back('\n'); }
    return s;
}

static const char* const NUMBERS[] = {"nan", "inf", "-inf", "1e308", "-1e308", "1e-320", "-1", "0", "-0.0", "4294967296", "18446744073709551616",
                                      "", "0x10", "1e", ".", "--3", "1.0.0", "99999999999999999999999999999"};

// text mutations (YAML, OBJ)
static std::string mutate_text(const std::string& in) {
    std::vector<std::string> L = lines_of(in);
    const int n_mut = 1 + (int)rnd(3);
    for (int m = 0; m < n_mut && !L.empty(); ++m) {
        const size_t i = rnd(L.size());
        switch (rnd(9)) {
            case 0: L.erase(L.begin() + (long)i); break;
            case 1: L.insert(L.begin() + (long)i, L[rnd(L.size())]); break;
            case 2: std::swap(L[i], L[rnd(L.size())]); break;
            case 3: {   // replace one number of the line
                std::string& s = L[i];
                std::vector<std::pair<size_t, size_t>> nums;
                for (size_t a = 0; a < s.size();) {
                    if (isdigit((unsigned char)s[a]) || ((s[a] == '-' || s[a] == '.') && a + 1 < s.size() && isdigit((unsigned char)s[a + 1]))) {
                        size_t b = a + 1;
                        while (b < s.size() && (isdigit((unsigned char)s[b]) || s[b] == '.' || s[b] == 'e' || s[b] == 'E' || s[b] == '-' || s[b] == '+')) ++b;
                        nums.push_back({a, b});
                        a = b;
                    } else ++a;
                }
                if (!nums.empty()) {
                    auto [a, b] = nums[rnd(nums.size())];
                    s = s.substr(0, a) + NUMBERS[rnd(sizeof(NUMBERS) / sizeof(NUMBERS[0]))] + s.substr(b);
                }
                break;
            }
            case 4: if (!L[i].empty()) L[i][rnd(L[i].size())] = (char)rnd(256); break;
            case 5: L[i] = L[i].substr(0, rnd(L[i].size() + 1)); break;
            case 6: L[i] = std::string(rnd(5), ' ') + L[i]; break;                 // indentation change
            case 7: L.resize(rnd(L.size() + 1)); break;                            // truncation
            case 8: L[i] += std::string(1 + rnd(3000), "0123456789abcdef /-:[],"[rnd(23)]); break;   // very long line
        }
    }
    return join(L);
}
// byte mutations (PNG)
static std::string mutate_bytes(const std::string& in) {
    std::string s = in;
    const int n_mut = 1 + (int)rnd(4);
    for (int m = 0; m < n_mut && !s.empty(); ++m) {
        switch (rnd(5)) {
            case 0: s[rnd(s.size())] = (char)rnd(256); break;
            case 1: s.resize(rnd(s.size() + 1)); break;
            case 2: { const size_t a = rnd(s.size()), len = rnd(64); s.erase(a, len); break; }
            case 3: { const size_t a = rnd(s.size()); s.insert(a, std::string(rnd(64), (char)rnd(256))); break; }
            case 4: {   // overwrite a big-endian u32 (chunk lengths, IHDR width/height) with an extreme value
                if (s.size() >= 40) {
                    const size_t a = 8 + rnd(32);
                    const uint32_t vals[] = {0u, 1u, 0x7FFFFFFFu, 0xFFFFFFFFu, 0x10000u, 65535u};
                    const uint32_t v = vals[rnd(6)];
                    for (int k = 0; k < 4; ++k) s[a + (size_t)k] = (char)(v >> (24 - 8 * k));
                }
                break;
            }
        }
    }
    return s;
}

// reads everything a loaded scene declares; returns a checksum so that nothing is optimised away
static double walk(const rt_scene_desc* d) {
    double acc = d->width + d->height;
    for (uint32_t i = 0; i < d->n_materials; ++i) acc += d->mat_type[i] + d->mat_rgb[3 * i] + d->mat_rgb[3 * i + 2] + d->mat_param[i] + d->mat_texture[i];
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const size_t n = (size_t)d->tex_width[i] * d->tex_height[i] * 3;
        if (n) acc += d->tex_rgb8[i][0] + d->tex_rgb8[i][n - 1];
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i) acc += d->sphere_center[3 * i + 2] + d->sphere_radius[i] + d->sphere_material[i] + d->sphere_object[i];
    for (uint32_t i = 0; i < d->n_planes; ++i) acc += d->plane_center[3 * i + 2] + d->plane_normal[3 * i + 2] + d->plane_material[i] + d->plane_object[i];
    for (uint64_t i = 0; i < d->n_triangles; ++i) {
        acc += d->tri_vertices[9 * i] + d->tri_vertices[9 * i + 8] + d->tri_material[i] + d->tri_object[i] + d->tri_index[i];
        if (d->tri_has_uv && d->tri_has_uv[i]) acc += d->tri_uv[6 * i] + d->tri_uv[6 * i + 5];
        // the indices the upload would use must be in range
        if (d->tri_material[i] < 0 || (uint32_t)d->tri_material[i] >= d->n_materials) { fprintf(stderr, "triangle material out of range\n"); abort(); }
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i)
        if (d->sphere_material[i] < 0 || (uint32_t)d->sphere_material[i] >= d->n_materials) { fprintf(stderr, "sphere material out of range\n"); abort(); }
    for (uint32_t i = 0; i < d->n_planes; ++i)
        if (d->plane_material[i] < 0 || (uint32_t)d->plane_material[i] >= d->n_materials) { fprintf(stderr, "plane material out of range\n"); abort(); }
    for (uint32_t i = 0; i < d->n_materials; ++i)
        if (d->mat_texture[i] >= (int32_t)d->n_textures) { fprintf(stderr, "texture index out of range\n"); abort(); }
    for (uint32_t i = 0; i < d->n_lights; ++i) acc += d->light_point[3 * i + 2];
    return acc;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: host_fuzz <golden dir> <work dir> <iterations> <seed>\n"); return 2; }
    const std::string golden = argv[1], work = argv[2];
    const int iters = atoi(argv[3]);
    rng.seed((uint64_t)atoll(argv[4]));
    signal(SIGALRM, on_alarm);
    const std::string cube_yaml = slurp(golden + "/scenes/cube/cube.yaml"), cube_obj = slurp(golden + "/scenes/cube/cube.obj"),
                      cube_png = slurp(golden + "/scenes/cube/cube_texture.png"), spheres_yaml = slurp(golden + "/scenes/spheres/spheres.yaml");
    if (cube_yaml.empty() || cube_obj.empty() || cube_png.empty() || spheres_yaml.empty()) { fprintf(stderr, "fixtures missing\n"); return 2; }
    // a small texture keeps an iteration cheap: the unmodified PNG is used in the rounds that do not mutate it
    const std::string dir = work + "/scenes/cube";
    (void)system(("mkdir -p '" + dir + "' '" + work + "/scenes/spheres'").c_str());
    char err[512];
    double acc = 0;
    int loaded = 0, rejected = 0;
    for (int it = 0; it < iters; ++it) {
        const int what = (int)rnd(5);
        spit(dir + "/cube.yaml", what == 0 ? mutate_text(cube_yaml) : cube_yaml);
        spit(dir + "/cube.obj", what == 1 ? mutate_text(cube_obj) : cube_obj);
        spit(dir + "/cube_texture.png", what == 2 ? mutate_bytes(cube_png) : cube_png);
        const char* path = (dir + "/cube.yaml").c_str();
        std::string p = dir + "/cube.yaml";
        if (what == 3) { p = work + "/scenes/spheres/spheres.yaml"; spit(p, mutate_text(spheres_yaml)); }
        (void)path;
        err[0] = 0;
        alarm(20);
        if (getenv("HOST_FUZZ_VERBOSE")) { fprintf(stderr, "scene %d: what %d\n", it, what); fflush(stderr); }
        rt_host_scene* s = rt_host_scene_load(p.c_str(), work.c_str(), err, sizeof err);
        if (s) {
            ++loaded;
            acc += walk(rt_host_scene_desc(s));
            for (uint32_t i = 0; i < rt_host_scene_num_objects(s) + 2; ++i) acc += rt_host_scene_object_kind(s, i);
            rt_host_scene_free(s);
        } else {
            ++rejected;
            if (!err[0]) { fprintf(stderr, "iteration %d: load failed without a message\n", it); return 1; }
        }
        if (what == 2 || what == 4) {   // the PNG reader on its own, and the parser of --ssaa
            uint8_t* px = nullptr;
            uint32_t w = 0, h = 0;
            if (rt_host_read_png((dir + "/cube_texture.png").c_str(), &px, &w, &h, err, sizeof err) == 0) {
                if ((size_t)w * h) acc += px[0] + px[(size_t)w * h * 3 - 1];
                rt_host_free(px);
            }
            int32_t mode;
            uint32_t n;
            std::string t = mutate_text(rnd(2) ? "uniform:4\n" : "jitter:16\n");
            if (!t.empty() && t.back() == '\n') t.pop_back();
            acc += rt_host_ssaa_parse(t.c_str(), &mode, &n, err, sizeof err);
        }
    }
    // the BVH builder on hostile boxes
    const float INF = INFINITY, NANF = NAN;
    for (int it = 0; it < iters / 4 + 8; ++it) {
        const uint64_t n = it < 6 ? (uint64_t)it : 1 + rnd(it % 7 == 0 ? 70000 : 600);
        std::vector<float> b(6 * n);
        const int kind = (int)rnd(6);
        for (uint64_t i = 0; i < n; ++i) {
            float c[3], e[3];
            for (int a = 0; a < 3; ++a) {
                c[a] = kind == 1 ? 1.0f : (float)((double)(rng() % 2000001) / 1000.0 - 1000.0);      // kind 1: all boxes identical
                e[a] = kind == 2 ? 0.f : (float)((double)(rng() % 1000) / 100.0);                   // kind 2: points
                if (kind == 3 && rnd(50) == 0) c[a] = rnd(2) ? INF : -INF;
                if (kind == 4 && rnd(50) == 0) c[a] = NANF;
                if (kind == 5) c[a] *= 1e30f;
            }
            for (int a = 0; a < 3; ++a) { b[6 * i + a] = c[a] - e[a]; b[6 * i + 3 + a] = c[a] + e[a]; }
        }
        uint32_t nodes = 0, depth = 0, stack = 0;
        err[0] = 0;
        alarm(20);
        if (getenv("HOST_FUZZ_VERBOSE")) { fprintf(stderr, "bvh %d: n %llu kind %d\n", it, (unsigned long long)n, kind); fflush(stderr); }
        const int rc = rt_host_bvh_validate(n, b.data(), &nodes, &depth, &stack, err, sizeof err);
        if (rc != 0 && !err[0]) { fprintf(stderr, "bvh %d: failure without a message\n", it); return 1; }
        if (rc != 0 && kind != 3 && kind != 4) { fprintf(stderr, "bvh %d: finite boxes (kind %d, n %llu) rejected: %s\n", it, kind, (unsigned long long)n, err); return 1; }
        acc += nodes + depth + stack;
    }
    alarm(0);
    printf("host_fuzz: %d iterations, %d scenes loaded, %d rejected with a message, checksum %.6g\n", iters, loaded, rejected, acc);
    return 0;
}
