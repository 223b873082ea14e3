/* Compiles the C ABI (include/rtamd.h, include/rtamd_host.h) as C11 -- the nearest compilable proxy for what a
 * Rust `-sys` crate (rust/rtamd-sys/src/lib.rs) assumes -- and pins the size of every struct and the offset and
 * size of every field (LP64).  Built by __graft_entry__.build() with
 *   gcc -std=c11 -Wall -Werror -pedantic -o tests/abi_check tests/abi_check.c
 * so a layout change breaks the build until this table (tools/gen_abi_check.py), raytracer-rust_amd/lib.py (ctypes)
 * and rust/rtamd-sys/abi_layout.json (the table rust/rtamd-sys/src/lib.rs is stated to follow) are updated together.
 * Run, it prints the layout as JSON; tests/test_abi.py compares that with the ctypes structs and the table.
 * Reference seam: /root/reference/src/raytracer/raytrace.rs:108-156, /root/reference/src/main.rs:21-26. */
#include <stddef.h>
#include <stdio.h>

#include "../include/rtamd.h"
#include "../include/rtamd_host.h"

#define SIZE_IS(T, n) _Static_assert(sizeof(T) == (n), "sizeof(" #T ") changed")
#define FIELD_IS(T, f, off, sz)                                                  \
    _Static_assert(offsetof(T, f) == (off), "offsetof(" #T ", " #f ") changed"); \
    _Static_assert(sizeof(((T*)0)->f) == (sz), "sizeof(" #T "." #f ") changed")

SIZE_IS(rt_scene_desc, 352);
FIELD_IS(rt_scene_desc, width, 0, 4);
FIELD_IS(rt_scene_desc, height, 4, 4);
FIELD_IS(rt_scene_desc, background, 8, 24);
FIELD_IS(rt_scene_desc, cam_eye, 32, 24);
FIELD_IS(rt_scene_desc, cam_lower_left, 56, 24);
FIELD_IS(rt_scene_desc, cam_horizontal, 80, 24);
FIELD_IS(rt_scene_desc, cam_vertical, 104, 24);
FIELD_IS(rt_scene_desc, n_materials, 128, 4);
FIELD_IS(rt_scene_desc, mat_type, 136, 8);
FIELD_IS(rt_scene_desc, mat_rgb, 144, 8);
FIELD_IS(rt_scene_desc, mat_param, 152, 8);
FIELD_IS(rt_scene_desc, mat_texture, 160, 8);
FIELD_IS(rt_scene_desc, n_textures, 168, 4);
FIELD_IS(rt_scene_desc, tex_width, 176, 8);
FIELD_IS(rt_scene_desc, tex_height, 184, 8);
FIELD_IS(rt_scene_desc, tex_rgb8, 192, 8);
FIELD_IS(rt_scene_desc, n_spheres, 200, 4);
FIELD_IS(rt_scene_desc, sphere_center, 208, 8);
FIELD_IS(rt_scene_desc, sphere_radius, 216, 8);
FIELD_IS(rt_scene_desc, sphere_material, 224, 8);
FIELD_IS(rt_scene_desc, sphere_object, 232, 8);
FIELD_IS(rt_scene_desc, n_planes, 240, 4);
FIELD_IS(rt_scene_desc, plane_center, 248, 8);
FIELD_IS(rt_scene_desc, plane_normal, 256, 8);
FIELD_IS(rt_scene_desc, plane_material, 264, 8);
FIELD_IS(rt_scene_desc, plane_object, 272, 8);
FIELD_IS(rt_scene_desc, n_triangles, 280, 8);
FIELD_IS(rt_scene_desc, tri_vertices, 288, 8);
FIELD_IS(rt_scene_desc, tri_uv, 296, 8);
FIELD_IS(rt_scene_desc, tri_has_uv, 304, 8);
FIELD_IS(rt_scene_desc, tri_material, 312, 8);
FIELD_IS(rt_scene_desc, tri_object, 320, 8);
FIELD_IS(rt_scene_desc, tri_index, 328, 8);
FIELD_IS(rt_scene_desc, n_lights, 336, 4);
FIELD_IS(rt_scene_desc, light_point, 344, 8);

SIZE_IS(rt_render_params, 72);
FIELD_IS(rt_render_params, ssaa_mode, 0, 4);
FIELD_IS(rt_render_params, ssaa_n, 4, 4);
FIELD_IS(rt_render_params, depth, 8, 4);
FIELD_IS(rt_render_params, seed, 16, 8);
FIELD_IS(rt_render_params, sample_begin, 24, 4);
FIELD_IS(rt_render_params, sample_end, 28, 4);
FIELD_IS(rt_render_params, tile_index, 32, 4);
FIELD_IS(rt_render_params, tile_count, 36, 4);
FIELD_IS(rt_render_params, max_paths_per_pass, 40, 4);
FIELD_IS(rt_render_params, max_shadow_generations, 44, 4);
FIELD_IS(rt_render_params, flags, 48, 4);
FIELD_IS(rt_render_params, sample_offset, 56, 16);

SIZE_IS(rt_stats, 2680);
FIELD_IS(rt_stats, rays_total, 0, 8);
FIELD_IS(rt_stats, rays_per_iteration, 8, 2560);
FIELD_IS(rt_stats, paths, 2568, 8);
FIELD_IS(rt_stats, nodes_visited, 2576, 8);
FIELD_IS(rt_stats, prims_tested, 2584, 8);
FIELD_IS(rt_stats, queue_overflow, 2592, 8);
FIELD_IS(rt_stats, hit_launches, 2600, 4);
FIELD_IS(rt_stats, hit_ms, 2608, 8);
FIELD_IS(rt_stats, bvh_nodes, 2616, 4);
FIELD_IS(rt_stats, bvh_depth, 2620, 4);
FIELD_IS(rt_stats, bvh_prims, 2624, 4);
FIELD_IS(rt_stats, bvh_build_ms, 2632, 8);
FIELD_IS(rt_stats, wave_node_steps, 2640, 8);
FIELD_IS(rt_stats, wave_leaf_steps, 2648, 8);
FIELD_IS(rt_stats, wave_refills, 2656, 8);
FIELD_IS(rt_stats, self_hit_rays, 2664, 8);
FIELD_IS(rt_stats, step_lanes, 2672, 8);

_Static_assert(RT_MAX_ITERATIONS == 320, "RT_MAX_ITERATIONS changed");
_Static_assert(RT_MAT_LAMBERTIAN == 0 && RT_MAT_METAL == 1 && RT_MAT_DIELECTRIC == 2 && RT_MAT_TEXTURE == 3 && RT_MAT_EMISSIVE == 4, "material tags changed");
_Static_assert(RT_SSAA_UNIFORM == 0 && RT_SSAA_JITTER == 1, "ssaa tags changed");
_Static_assert(RT_FLAG_COUNTERS == 1 && RT_FLAG_TIMING == 2, "flag bits changed");
_Static_assert(RT_NOALLOC_SCRATCH_BYTES == (64u << 20), "scratch size changed");
_Static_assert(sizeof(void*) == 8 && sizeof(size_t) == 8, "LP64 expected");

/* every entry point is declared with a C-compatible signature (an unevaluated use: nothing is linked) */
#define DECLARED(f) _Static_assert(sizeof(&f) == sizeof(void (*)(void)), #f " is not declared")
DECLARED(rt_accum_reduce);
DECLARED(rt_accum_reduce_rccl);
DECLARED(rt_bvh_build_validate);
DECLARED(rt_context_create);
DECLARED(rt_context_destroy);
DECLARED(rt_context_set_bvh_builder);
DECLARED(rt_debug_gather);
DECLARED(rt_get_stats);
DECLARED(rt_host_bvh_validate);
DECLARED(rt_host_camera);
DECLARED(rt_host_compute_image);
DECLARED(rt_host_compute_image_devices);
DECLARED(rt_host_free);
DECLARED(rt_host_read_png);
DECLARED(rt_host_scene_desc);
DECLARED(rt_host_scene_free);
DECLARED(rt_host_scene_load);
DECLARED(rt_host_scene_num_objects);
DECLARED(rt_host_scene_object_kind);
DECLARED(rt_host_ssaa_parse);
DECLARED(rt_host_write_png);
DECLARED(rt_host_write_png16);
DECLARED(rt_last_error);
DECLARED(rt_primary_hits);
DECLARED(rt_primary_rays);
DECLARED(rt_render_accumulate);
DECLARED(rt_render_accumulate_noalloc);
DECLARED(rt_render_image);
DECLARED(rt_render_image_devices);
DECLARED(rt_resolve);
DECLARED(rt_scene_upload);
DECLARED(rt_scene_upload_devices);
DECLARED(rt_stats_reset);
#define N_ENTRY_POINTS 33

#define P_STRUCT(T) printf("%s\"" #T "\": {\"size\": %zu, \"fields\": [", first_struct ? "" : ", ", sizeof(T)); first_struct = 0; first_field = 1
#define P_FIELD(T, f) printf("%s[\"" #f "\", %zu, %zu]", first_field ? "" : ", ", offsetof(T, f), sizeof(((T*)0)->f)); first_field = 0
#define P_END() printf("]}")

int main(void) {
    int first_struct = 1, first_field = 1;
    printf("{");
    P_STRUCT(rt_scene_desc);
    P_FIELD(rt_scene_desc, width);
    P_FIELD(rt_scene_desc, height);
    P_FIELD(rt_scene_desc, background);
    P_FIELD(rt_scene_desc, cam_eye);
    P_FIELD(rt_scene_desc, cam_lower_left);
    P_FIELD(rt_scene_desc, cam_horizontal);
    P_FIELD(rt_scene_desc, cam_vertical);
    P_FIELD(rt_scene_desc, n_materials);
    P_FIELD(rt_scene_desc, mat_type);
    P_FIELD(rt_scene_desc, mat_rgb);
    P_FIELD(rt_scene_desc, mat_param);
    P_FIELD(rt_scene_desc, mat_texture);
    P_FIELD(rt_scene_desc, n_textures);
    P_FIELD(rt_scene_desc, tex_width);
    P_FIELD(rt_scene_desc, tex_height);
    P_FIELD(rt_scene_desc, tex_rgb8);
    P_FIELD(rt_scene_desc, n_spheres);
    P_FIELD(rt_scene_desc, sphere_center);
    P_FIELD(rt_scene_desc, sphere_radius);
    P_FIELD(rt_scene_desc, sphere_material);
    P_FIELD(rt_scene_desc, sphere_object);
    P_FIELD(rt_scene_desc, n_planes);
    P_FIELD(rt_scene_desc, plane_center);
    P_FIELD(rt_scene_desc, plane_normal);
    P_FIELD(rt_scene_desc, plane_material);
    P_FIELD(rt_scene_desc, plane_object);
    P_FIELD(rt_scene_desc, n_triangles);
    P_FIELD(rt_scene_desc, tri_vertices);
    P_FIELD(rt_scene_desc, tri_uv);
    P_FIELD(rt_scene_desc, tri_has_uv);
    P_FIELD(rt_scene_desc, tri_material);
    P_FIELD(rt_scene_desc, tri_object);
    P_FIELD(rt_scene_desc, tri_index);
    P_FIELD(rt_scene_desc, n_lights);
    P_FIELD(rt_scene_desc, light_point);
    P_END();
    P_STRUCT(rt_render_params);
    P_FIELD(rt_render_params, ssaa_mode);
    P_FIELD(rt_render_params, ssaa_n);
    P_FIELD(rt_render_params, depth);
    P_FIELD(rt_render_params, seed);
    P_FIELD(rt_render_params, sample_begin);
    P_FIELD(rt_render_params, sample_end);
    P_FIELD(rt_render_params, tile_index);
    P_FIELD(rt_render_params, tile_count);
    P_FIELD(rt_render_params, max_paths_per_pass);
    P_FIELD(rt_render_params, max_shadow_generations);
    P_FIELD(rt_render_params, flags);
    P_FIELD(rt_render_params, sample_offset);
    P_END();
    P_STRUCT(rt_stats);
    P_FIELD(rt_stats, rays_total);
    P_FIELD(rt_stats, rays_per_iteration);
    P_FIELD(rt_stats, paths);
    P_FIELD(rt_stats, nodes_visited);
    P_FIELD(rt_stats, prims_tested);
    P_FIELD(rt_stats, queue_overflow);
    P_FIELD(rt_stats, hit_launches);
    P_FIELD(rt_stats, hit_ms);
    P_FIELD(rt_stats, bvh_nodes);
    P_FIELD(rt_stats, bvh_depth);
    P_FIELD(rt_stats, bvh_prims);
    P_FIELD(rt_stats, bvh_build_ms);
    P_FIELD(rt_stats, wave_node_steps);
    P_FIELD(rt_stats, wave_leaf_steps);
    P_FIELD(rt_stats, wave_refills);
    P_FIELD(rt_stats, self_hit_rays);
    P_FIELD(rt_stats, step_lanes);
    P_END();
    printf(", \"entry_points\": %zu}\n", (size_t)N_ENTRY_POINTS);
    return 0;
}
