// The threaded host BVH builder (bvh.cpp: the top of the tree is split serially, the subtrees are built by worker threads) under
// ThreadSanitizer, on enough boxes to take the parallel path; built and run by tests/test_host_fuzz.py.  Device entry points stubbed.
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>
#include "../include/rtamd_host.h"
extern "C" {
int rt_context_create(int32_t, rt_context**) { return 1; }
void rt_context_destroy(rt_context*) {}
const char* rt_last_error(const rt_context*) { return "stub"; }
int rt_scene_upload(rt_context*, const rt_scene_desc*) { return 1; }
int rt_scene_upload_devices(rt_context* const*, uint32_t, const rt_scene_desc*) { return 1; }
int rt_render_image(rt_context*, const rt_render_params*, float*, uint8_t*) { return 1; }
int rt_render_image_devices(rt_context* const*, uint32_t, const rt_render_params*, int32_t, float*, uint8_t*) { return 1; }
}
int main() {
    std::mt19937_64 rng(7);
    for (uint64_t n : {70000ull, 300000ull}) {
        std::vector<float> b(6 * n);
        for (uint64_t i = 0; i < n; ++i) {
            float c[3], e[3];
            for (int a = 0; a < 3; ++a) { c[a] = (float)(rng() % 100000) / 1000.f; e[a] = (float)(rng() % 100) / 1000.f; }
            for (int a = 0; a < 3; ++a) { b[6 * i + a] = c[a] - e[a]; b[6 * i + 3 + a] = c[a] + e[a]; }
        }
        uint32_t nodes, depth, stack; char err[256] = "";
        int rc = rt_host_bvh_validate(n, b.data(), &nodes, &depth, &stack, err, sizeof err);
        printf("n %llu rc %d nodes %u depth %u stack %u %s\n", (unsigned long long)n, rc, nodes, depth, stack, err);
        if (rc) return 1;
    }
    return 0;
}
