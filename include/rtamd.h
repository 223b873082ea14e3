/*
 * rtamd.h -- C ABI of the MI355X (gfx950) render path for bsati/raytracer-rust.
 *
 * The reference has no FFI / plugin interface of its own.  The seam this
 * library replaces is the body of `compute_image` between `scene.precompute()`
 * and `image::write_image` (/root/reference/src/raytracer/raytrace.rs:125-155):
 * input = a loaded Scene + Camera + SuperSampling + depth, output = H x W
 * colours, top row first.  Every entry point below cites the reference code it
 * stands in for.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; the message is
 *    available from rt_last_error(ctx) (or rt_last_error(NULL) for failures of
 *    rt_context_create).  No exception or abort crosses this boundary: the
 *    reference's `unwrap()` panics become error codes.
 *  - a context is bound to one GPU and is NOT thread-safe; calls on one context
 *    must not overlap.  One process per GPU is the intended use.
 *  - images are row-major, row 0 = TOP of the view (raytrace.rs:128 `.rev()`),
 *    3 channels interleaved.
 *  - "host" pointers are ordinary memory, "dev" pointers are device memory on
 *    the context's GPU (e.g. a torch tensor's data_ptr()).
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

/* materials.rs:20-26 `enum Material` (tag spelling "Dieletrics" is the reference's) */
enum rt_material_type {
    RT_MAT_LAMBERTIAN = 0, /* materials.rs:40-66   rgb = albedo                       */
    RT_MAT_METAL = 1,      /* materials.rs:155-183 rgb = albedo, param = fuzziness    */
    RT_MAT_DIELECTRIC = 2, /* materials.rs:90-153  rgb = tint, param = refraction idx */
    RT_MAT_TEXTURE = 3,    /* materials.rs:185-253 texture = index into textures      */
    RT_MAT_EMISSIVE = 4    /* materials.rs:68-88   rgb = color                        */
};

/*
 * Flattened scene, in the reference's own number type (f64).  It is what a
 * Rust host would produce from `Scene` (scene.rs:18-27) after
 * `Camera::new` (camera.rs:26-54) and `Scene::precompute` (scene.rs:54-72).
 * The library copies everything during rt_scene_upload; the caller keeps
 * ownership.  `*_object` is the position of the owning object in
 * `scene.objects` (scene.rs:26); `tri_index` the position of the triangle in
 * `mesh.triangles` (mesh.rs:296).  Ties in hit distance are resolved towards
 * the smaller (object, triangle) pair exactly as the reference's strict `>`
 * comparisons do (scene.rs:43, intersections.rs:170).
 */
typedef struct rt_scene_desc {
    uint32_t width, height;        /* scene.rs:21-22 */
    double background[3];          /* scene.rs:23    */
    /* camera.rs:7-12, computed by Camera::new on the host */
    double cam_eye[3], cam_lower_left[3], cam_horizontal[3], cam_vertical[3];

    uint32_t n_materials;
    const int32_t* mat_type;       /* rt_material_type                      */
    const double* mat_rgb;         /* 3 per material                        */
    const double* mat_param;       /* fuzziness / refraction_index / 0      */
    const int32_t* mat_texture;    /* texture index or -1                   */

    uint32_t n_textures;           /* image.rs:166-184 read_image output    */
    const uint32_t* tex_width;
    const uint32_t* tex_height;
    const uint8_t* const* tex_rgb8; /* width*height*3 bytes each, row-major */

    uint32_t n_spheres;            /* scene.rs:174-179 */
    const double* sphere_center;   /* 3 per sphere */
    const double* sphere_radius;
    const int32_t* sphere_material;
    const int32_t* sphere_object;

    uint32_t n_planes;             /* scene.rs:181-186 */
    const double* plane_center;    /* 3 per plane */
    const double* plane_normal;    /* 3 per plane, as written (not normalised) */
    const int32_t* plane_material;
    const int32_t* plane_object;

    uint64_t n_triangles;          /* mesh.rs:295-302, positions pre-resolved */
    const double* tri_vertices;    /* 9 per triangle: A, B, C by vertex_idx   */
    const double* tri_uv;          /* 6 per triangle (uv of idx 0,1,2) or NULL */
    const uint8_t* tri_has_uv;     /* per triangle flag, or NULL (= tri_uv != NULL) */
    const int32_t* tri_material;
    const int32_t* tri_object;
    const int32_t* tri_index;

    uint32_t n_lights;             /* scene.rs:114-146: sample_points[0] of each light */
    const double* light_point;     /* 3 per light */
} rt_scene_desc;

enum rt_ssaa_mode { RT_SSAA_UNIFORM = 0, RT_SSAA_JITTER = 1 }; /* anti_aliasing.rs:6-9 */

enum rt_render_flags {
    RT_FLAG_COUNTERS = 1, /* count BVH nodes visited / primitives tested (slower kernels) */
    RT_FLAG_TIMING = 2    /* HIP-event timing of every closest-hit launch                  */
};

/*
 * One render call = `compute_image`'s pixel loop (raytrace.rs:126-154) restricted
 * to a sample range and a tile set, so that the work can be split across GPUs.
 */
typedef struct rt_render_params {
    int32_t ssaa_mode;       /* rt_ssaa_mode                                             */
    uint32_t ssaa_n;         /* grid resolution N; N*N samples per pixel                 */
    uint32_t depth;          /* max_depth of Ray::trace (raytrace.rs:51), 0..255         */
    uint64_t seed;           /* RNG seed (the reference seeds from the OS)               */
    uint32_t sample_begin;   /* samples [begin,end) of the x-major N*N list              */
    uint32_t sample_end;     /* (anti_aliasing.rs:96-100); end = 0 means N*N             */
    uint32_t tile_index;     /* 8x8-pixel tiles are dealt round-robin: this call renders */
    uint32_t tile_count;     /* tiles t with t % tile_count == tile_index; 0 means 1     */
    uint32_t max_paths_per_pass;     /* 0 = default: as many paths as fit in 70 % of the    */
                                     /* device memory free at the call, at most 2^30 (the   */
                                     /* wavefront state is 16 + 168 * max(1, n_lights)      */
                                     /* bytes per path: 184 B, 197 GB for 2^30 paths; few,  */
                                     /* large launches are faster: DESIGN.md section 3)     */
    uint32_t max_shadow_generations; /* cap of the shadow-ray branching process          */
                                     /* (raytrace.rs:78-85 recursion); 0 = default 16    */
    uint32_t flags;          /* rt_render_flags                                          */
    double sample_offset[2]; /* added to every sample position (used by the hit-ID gate) */
} rt_render_params;

#define RT_MAX_ITERATIONS 320

typedef struct rt_stats {
    uint64_t rays_total;                       /* closest-hit queries actually traced        */
    uint64_t rays_per_iteration[RT_MAX_ITERATIONS]; /* iteration = bounce, then shadow generations */
    uint64_t paths;                            /* primary rays generated                     */
    uint64_t nodes_visited;                    /* RT_FLAG_COUNTERS only                      */
    uint64_t prims_tested;                     /* RT_FLAG_COUNTERS only                      */
    uint64_t queue_overflow;                   /* rays dropped because a queue was full      */
    uint32_t hit_launches;                     /* RT_FLAG_TIMING: closest-hit launches timed */
    double hit_ms;                             /* RT_FLAG_TIMING: sum of their durations     */
    uint32_t bvh_nodes, bvh_depth, bvh_prims;
    double bvh_build_ms;
    /* RT_FLAG_COUNTERS only: what the closest-hit WAVES did (the roofline model of bench.py is built on these) */
    uint64_t wave_node_steps;                  /* steps in which a wave ran the BVH-node code                */
    uint64_t wave_leaf_steps;                  /* steps in which a wave ran the primitive code               */
    uint64_t wave_refills;                     /* times a wave fetched new rays into idle lanes              */
    uint64_t self_hit_rays;                    /* rays that re-hit the triangle they start on (no traversal) */
    uint64_t step_lanes;                       /* lanes that had work, summed over all steps                 */
} rt_stats;

/* No reference counterpart (the reference is single-process CPU code). */
int rt_context_create(int device_id, rt_context** out);
void rt_context_destroy(rt_context* ctx);
const char* rt_last_error(const rt_context* ctx);

/* Stands in for Scene ownership by compute_image (raytrace.rs:114-125): narrows the
 * f64 scene to the device layout and builds the BVH that replaces the linear scans of
 * scene.rs:35-52 and intersections.rs:148-186. */
int rt_scene_upload(rt_context* ctx, const rt_scene_desc* scene);

/* raytrace.rs:126-143: traces the selected samples of the selected tiles and ADDS the
 * per-pixel sum of sample colours into accum_dev (float, height*width*3, device).
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream). */
int rt_render_accumulate(rt_context* ctx, const rt_render_params* params, float* accum_dev, void* stream);

/* No counterpart in the reference (which allocates freely, e.g. one Vec per pixel, anti_aliasing.rs:95).
 * rt_render_accumulate for callers that CANNOT allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same result; the wavefront state
 * (control block, two ray queues, hit records, sample radiance) lives in a static device array of
 * RT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends, and the call
 * makes no hipMalloc / hipFree / event creation.  Limits, all reported as errors:
 *  - the call is ONE pass: paths = (tiles of the call) * 64 * (samples of the call) must satisfy
 *    16 KiB + paths * (16 + 168 * max(1, n_lights)) bytes <= RT_NOALLOC_SCRATCH_BYTES (the control block and the 256-byte
 *    alignment of the 13 carved arrays included; about 364,000 paths without lights, 190,000 with two); split larger frames
 *    over several calls by sample range or tile set;
 *  - RT_FLAG_TIMING is refused (it creates events).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two contexts, two streams)
 * race; calls in one stream are ordered.  Asynchronous on `stream` like rt_render_accumulate. */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_accumulate_noalloc(rt_context* ctx, const rt_render_params* params, float* accum_dev, void* stream);

/* raytrace.rs:144-150 + image.rs:29-43: mean over spp_total, sqrt, clamp to <= 1,
 * trunc(255.999*c) saturating.  Either output may be NULL.  Device pointers. */
int rt_resolve(rt_context* ctx, const float* accum_dev, uint32_t spp_total, uint8_t* rgb8_dev, float* color_dev, void* stream);

/* Whole pixel loop of compute_image (raytrace.rs:126-154) with host outputs
 * (either may be NULL): accum_host = per-pixel sum of sample colours (float),
 * rgb8_host = the bytes write_image would store (image.rs:128-135). Synchronous.
 * Fails (after filling the outputs) if rays were dropped because a queue was full.  A queue holds
 * n_lights rays per path of the pass, so the shadow rays of the last bounce always fit; only the
 * branching process after it can overflow, in scenes with 4 or more lights where every shadow ray
 * has more than one successor on average (raytrace.rs:78-85 spawns L rays per firing path and each
 * can fire again with probability 0.1 L; the reference itself recurses without bound there).
 * Whether that happens depends on the call's parameters only, not on what the context rendered
 * before; max_shadow_generations cuts the process off.  rt_render_accumulate is asynchronous and
 * reports the same count through rt_get_stats().queue_overflow. */
int rt_render_image(rt_context* ctx, const rt_render_params* params, float* accum_host, uint8_t* rgb8_host);

/* Closest hit of ONE primary ray per pixel at sub-pixel offset (dx,dy)
 * (camera.rs:62-67 spawn_ray + scene.rs:35-52).  Host outputs, height*width each
 * (normal: *3).  object_id = triangle_id = -1 and t = -1 on a miss;
 * triangle_id = -1 for spheres and planes.  Synchronous. */
int rt_primary_hits(rt_context* ctx, double dx, double dy, int32_t* object_id, int32_t* triangle_id, float* t, float* normal_xyz);

/* The device's own primary rays for the same sample position as rt_primary_hits: origin (3 floats, the eye)
 * and one unit direction per pixel (height*width*3 floats, host), exactly the f32 values k_raygen hands to the
 * closest-hit kernel (camera.rs:62-67 in f64, rounded once).  Lets the parity tests feed the oracle the very
 * rays the device traced, so that the 1e-5 gate on t and normal holds for every hit.  Synchronous. */
int rt_primary_rays(rt_context* ctx, double dx, double dy, float origin_xyz[3], float* dir_xyz);

/* Counters since the last rt_stats_reset (synchronises the context's streams). */
int rt_get_stats(rt_context* ctx, rt_stats* out);
int rt_stats_reset(rt_context* ctx);

/* ---- several GPUs from one process.  No reference counterpart: raytrace.rs:126-154 is one process on one
 * machine.  (bench.py's one-process-per-GPU path uses torch.distributed/RCCL for the same exchange.) ---- */

/* rt_scene_upload to n contexts (one per GPU): the scene is flattened and its BVH built ONCE (with ctxs[0]'s
 * builder), then copied to every context's GPU.  On error the message is on ctxs[0]. */
int rt_scene_upload_devices(rt_context* const* ctxs, uint32_t n, const rt_scene_desc* scene);

/* The exchange step of the sample / tile split, on the GPUs: dst_accum_dev (on dst's GPU) += each
 * src_accum_dev[k] (on src[k]'s GPU; height*width*3 floats).  A partial frame on another GPU comes over with one
 * peer copy (hipMemcpyPeerAsync: xGMI when peer access is available) into a landing buffer on dst's GPU,
 * followed by an add kernel; nothing passes through host memory.  Asynchronous on `stream` of dst's GPU; the
 * sources must be complete (synchronise their streams first). */
int rt_accum_reduce(rt_context* dst, float* dst_accum_dev, rt_context* const* src, const float* const* src_accum_dev, uint32_t n_src, void* stream);

/* The same exchange as ONE RCCL collective (ncclReduce(sum, float, root = GPU 0) of the W x H x 3 accumulators):
 * accum_dev[0] (on ctxs[0]'s GPU) += accum_dev[k], k = 1..n-1, every accum_dev[k] height*width*3 floats on ctxs[k]'s GPU.
 * The contexts must be on n DISTINCT GPUs (an RCCL communicator has one rank per GPU; n = 1 is a communicator of one rank);
 * contexts that share a GPU use rt_accum_reduce.  The communicator is created at the first call (ncclCommInitAll over the
 * contexts' GPUs), kept by ctxs[0] and re-created when the set of GPUs changes.  The frames must be complete; every rank's part
 * is queued on its context's own stream and the call returns when the reduce has finished on all GPUs.  The frames of the other
 * contexts are left as they are. */
int rt_accum_reduce_rccl(rt_context* const* ctxs, float* const* accum_dev, uint32_t n);

/* rt_render_image on n contexts that hold the same scene: split_tiles = 0 splits the samples of the call into
 * contiguous ranges (same RNG keys as one device), 1 deals the call's 8x8 tiles round-robin; one host thread
 * per context renders, then the partial frames are summed on ctxs[0]'s GPU -- contexts on distinct GPUs: rt_accum_reduce_rccl
 * (one RCCL reduce over xGMI); contexts that share a GPU, or a host where the communicator cannot be created: rt_accum_reduce
 * (peer copies) -- rt_resolve runs there and the result is copied to the host once.  Outputs as rt_render_image. */
int rt_render_image_devices(rt_context* const* ctxs, uint32_t n, const rt_render_params* params, int split_tiles, float* accum_host,
                            uint8_t* rgb8_host);

/* No reference counterpart (the reference has one AABB per mesh, mesh.rs:318-326).
 * Chooses the builder of the BVH that rt_scene_upload makes: 0 = host, binned SAH, threaded (default;
 * best trees: 0.2 s for 1M triangles), 1 = GPU linear BVH (Morton codes, radix sort, Karras tree,
 * collapse to the same 4-wide quantised nodes; milliseconds, more nodes per ray). */
int rt_context_set_bvh_builder(rt_context* ctx, int builder);

/* Builds a BVH over n boxes (6 floats each: min xyz, max xyz) with the context's builder on its
 * device and verifies it on the CPU like rt_host_bvh_validate: every primitive in exactly one leaf,
 * every quantised child box contains its subtree, the reported traversal-stack bound holds. */
int rt_bvh_build_validate(rt_context* ctx, uint64_t n, const float* boxes, uint32_t* n_nodes, uint32_t* depth, uint32_t* max_stack,
                          double* build_ms);

/* Measurement aid, no reference counterpart: runs a gather of whole 64-byte records (the BVH node
 * access pattern) over a table of table_bytes and reports the bytes it loaded, so that rocprofv3's
 * FETCH_SIZE can be calibrated for this pattern (tools/calib_fetch.py). */
int rt_debug_gather(rt_context* ctx, uint64_t table_bytes, uint32_t iters, double* bytes_loaded);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
