/*
 * rtamd_host.h -- host side of the drop-in: the reference's scene, mesh, camera,
 * sampling and image surface restated in C++ behind a C ABI (the reference is
 * Rust, which cannot be built in this environment).
 * The loaders produce the rt_scene_desc that rt_scene_upload() consumes.
 * Errors are returned as codes + messages; the reference panics instead.
 */
#ifndef RTAMD_HOST_H
#define RTAMD_HOST_H

#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

/* serde_yaml::from_reader::<Scene> (raytrace.rs:114-115, scene.rs:18-27,75-89,158-186,
 * materials.rs:18-26) + Camera::new (raytrace.rs:117-124, camera.rs:26-54) +
 * Scene::precompute (raytrace.rs:125, scene.rs:54-72) + flattening.
 * Paths inside the YAML are resolved against base_dir (NULL = the process working
 * directory, which is what the reference does).  Returns NULL on error and writes the
 * message to err. */
rt_host_scene* rt_host_scene_load(const char* yaml_path, const char* base_dir, char* err, size_t err_len);
void rt_host_scene_free(rt_host_scene* scene);
const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* scene);
/* scene.objects.len() and the kind of each object (0 sphere, 1 plane, 2 mesh) */
uint32_t rt_host_scene_num_objects(const rt_host_scene* scene);
int rt_host_scene_object_kind(const rt_host_scene* scene, uint32_t index);

/* SuperSampling::from_str (anti_aliasing.rs:50-83).  NB: "uniform:N" yields Jitter(N)
 * exactly like the reference (anti_aliasing.rs:75); the extra spelling "grid:N" gives the
 * true uniform grid of anti_aliasing.rs:93-102 (reachable only from code in the
 * reference).  On error returns non-zero and err holds the reference's message, e.g.
 * "unknown method"; Display adds the prefix "error parsing ssaa argument: ". */
int rt_host_ssaa_parse(const char* text, int32_t* mode, uint32_t* n, char* err, size_t err_len);

/* Camera::new (camera.rs:26-54) */
void rt_host_camera(const double eye[3], const double look_at[3], const double up[3], double fovy, uint32_t width, uint32_t height,
                    double out_eye[3], double out_horizontal[3], double out_vertical[3], double out_lower_left[3]);

/* image::write_image (image.rs:149-163): RGB8 PNG, creates parent directories. */
int rt_host_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height, char* err, size_t err_len);
/* No reference counterpart (the reference writes 8 bits only, image.rs:149-163): the same
 * file layout with 16-bit samples, for the colour plane of rt_resolve quantised as (65535.999 * c) as u16. */
int rt_host_write_png16(const char* path, const uint16_t* rgb16, uint32_t width, uint32_t height, char* err, size_t err_len);
/* image::read_image (image.rs:166-184): 8-bit RGB PNG -> malloc'ed bytes (free with rt_host_free). */
int rt_host_read_png(const char* path, uint8_t** rgb8, uint32_t* width, uint32_t* height, char* err, size_t err_len);
void rt_host_free(void* p);

/* No reference counterpart (the reference has one AABB per mesh, mesh.rs:318-326): builds the BVH
 * the closest-hit kernel traverses over n boxes (6 floats each: min xyz, max xyz) and verifies it on
 * the CPU -- every primitive in exactly one leaf, every quantised child box contains its subtree,
 * the reported traversal-stack bound holds. */
int rt_host_bvh_validate(uint64_t n, const float* boxes, uint32_t* n_nodes, uint32_t* depth, uint32_t* max_stack, char* err, size_t err_len);

/* raytracer::compute_image (raytrace.rs:108-156) on the GPU `device_id`:
 * load, upload, render all N*N samples at `depth`, resolve, write the PNG. */
int rt_host_compute_image(const char* ssaa, uint32_t depth, const char* scene_path, const char* output_path, const char* base_dir,
                          int device_id, uint64_t seed, char* err, size_t err_len);

/* The same on n_devices GPUs from one process (additive flags of SURVEY 8b: --devices, --split): one
 * context and host thread per device, the scene flattened and its BVH built once
 * (rt_scene_upload_devices); split_tiles = 0 splits the N*N samples (disjoint ranges, the same RNG
 * keys a single device would use), 1 deals the 8x8 tiles round-robin; the partial frames are summed
 * on the first GPU over peer copies and resolved there as raytrace.rs:144-150 / image.rs:37-43 do
 * (rt_render_image_devices); nothing but the final RGB8 frame passes through host memory. */
int rt_host_compute_image_devices(const char* ssaa, uint32_t depth, const char* scene_path, const char* output_path, const char* base_dir,
                                  const int* device_ids, uint32_t n_devices, int split_tiles, uint64_t seed, char* err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_HOST_H */
