// Device-side data layout shared by the kernels (rt_kernels.hip) and the C-ABI
// host code (rt_api.hip).  Everything is written for gfx950 (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtamd {

constexpr uint32_t HIT_MISS = 0xFFFFFFFFu;    // hit code: nothing hit
constexpr uint32_t HIT_PLANE_BIT = 0x80000000u;  // hit code: plane index in the low bits
constexpr uint32_t SRC_NONE = 0xFFFFFFFEu;    // ray does not start on a primitive
constexpr uint32_t SRC_INVALID = 0xFFFFFFFDu; // padding slot of a border tile: not a ray
constexpr uint32_t HIT_INVALID = 0xFFFFFFFCu; // hit code of such a slot
constexpr uint32_t SRC_SELF_BIT = 0x40000000u;  // src = SRC_SELF_BIT | triangle: the ray re-hits the triangle it starts on at t ~ 0 (decided by k_shade, see there)
constexpr uint32_t SRC_E_BIT = 0x20000000u;     // src | SRC_E_BIT (src below SRC_SPECIAL): queue word e holds the ray's f64 origin (it leaves a triangle or a plane)
constexpr uint32_t SRC_D_BIT = 0x10000000u;     // ... and e.w, f hold its f64 direction (it left a mirror or a dielectric); only with SRC_E_BIT
constexpr uint32_t HIT_SPHERE_BIT = 0x20000000u; // hit code (scenes with triangles): the primitive is a sphere -- k_shade sorts the rays of a block by what they hit
                                                 // (triangle / sphere / plane / nothing) before it shades them, see there; strip_hit() removes it
constexpr uint32_t AUX_NEAR_BIT = 0x40000000u;  // runner-up word of a hit record (scenes with triangles): AUX_NEAR_BIT | triangle = a candidate at t ~ 0
                                                // that the f32 traversal cannot rank against the nearest hit; k_shade's f64 test decides (resolve_tie)
constexpr uint32_t SRC_SPECIAL = 0xFFFFFFF0u;   // SRC_NONE, SRC_INVALID: no flag bits to read
constexpr uint32_t SRC_PRIM_MASK = 0x0FFFFFFFu; // primitive index of such a src (rt_scene_upload limits primitives to 2^28)
constexpr uint32_t PRIM_TRIANGLE = 0u, PRIM_SPHERE = 1u;
constexpr uint32_t SMETA_HAS_UV = 0x80000000u;
constexpr uint32_t SMETA_SPHERE = 0x40000000u;
constexpr uint32_t SMETA_FLAGS = SMETA_HAS_UV | SMETA_SPHERE;
constexpr int32_t CHILD_EMPTY_CODE = (int32_t)0x80000000;   // == bvh.h CHILD_EMPTY
// RT_POP_CULL: a traversal-stack entry carries the entry distance of its box beside the child code (8 bytes), and an entry whose
// distance lies beyond the nearest hit found since it was pushed is dropped when it is popped -- without it every such entry
// costs a whole node step (four loads, ~170 VALU) only to find that no child can be hit any more.
#ifndef RT_POP_CULL
#define RT_POP_CULL 0
#endif
#ifndef RT_LDS_STACK
#define RT_LDS_STACK (RT_POP_CULL ? 9 : 16)
#endif
constexpr uint32_t LDS_STACK = RT_LDS_STACK;    // traversal-stack entries per lane kept in LDS (4 KB per wave without RT_POP_CULL, 4.5 KB with)
constexpr uint32_t STACK_ENTRY_WORDS = RT_POP_CULL ? 2 : 1;
constexpr uint32_t HIT_BLOCKS_PER_CU = 8;     // persistent closest-hit blocks (256 threads) per CU

// control block (uint32 words, zeroed at the start of every pass)
constexpr int CTL_MAX_IT = 320;               // == RT_MAX_ITERATIONS
constexpr int CTL_QCOUNT = 0;                 // [it]   rays in the queue consumed by iteration `it`
constexpr int CTL_FETCH = CTL_MAX_IT + 8;     // [segment][it] dynamic-fetch cursors of the closest-hit kernel: the queue is cut
#ifndef RT_FETCH_SEGS
#define RT_FETCH_SEGS 8
#endif
constexpr int CTL_FETCH_SEGS = RT_FETCH_SEGS;             //   into 8 segments with one cursor each (a single address sustains ~90 atomics/us)
constexpr int CTL_FETCH_STRIDE = CTL_MAX_IT + 8;   // words between the cursors of two segments: different cache lines
constexpr int CTL_OVERFLOW = CTL_FETCH + CTL_FETCH_SEGS * CTL_FETCH_STRIDE;
// 64-bit counters of the counters build (RT_FLAG_COUNTERS), even word index = 8-byte aligned:
//   0 BVH nodes visited (lanes)   1 primitives tested (lanes)   2 node steps (waves)   3 leaf steps (waves)
//   4 refills (waves)             5 rays that re-hit their own triangle (no traversal)  6 lanes with work summed over the steps
constexpr int CTL_STAT64 = CTL_OVERFLOW + 2;
constexpr int N_STAT64 = 8;
constexpr int CTL_HEAVY = CTL_STAT64 + 2 * N_STAT64;   // [it & 1] rays k_shade<.., 1> left to k_shade<.., 2> (see k_shade, CLS)
constexpr int CTL_WORDS = CTL_HEAVY + 2;
static_assert(CTL_STAT64 % 2 == 0, "64-bit counters must be 8-byte aligned");

// Ray queue, structure of arrays of 16-byte words so that every wave-level access
// is one contiguous 1 KiB transaction:
//   a = (o.x, o.y, o.z, d.x)
//   b = (d.y, d.z, bits(src prim code), bits(path id))      -- a,b: all the closest-hit kernel reads
//   c = (throughput.r, .g, .b, bits(rng state))
//   e = one word per slot, touched only by rays with SRC_E_BIT (in scenes with triangles: every ray but the primary ones): the
//       low 29 mantissa bits of the x, y, z of the ray's f64 origin, whose sign, exponent and upper 23 mantissa bits are the
//       f32 origin in `a` (truncated, not rounded, for these rays).  Whether a ray re-hits the triangle it lands on depends
//       on the last bits of its origin (see tri_hit_ref), so in such scenes every hit point is the reference's own f64
//       function of the ray (plane: intersections.rs:84-95, sphere: :42-68, triangle: :153-172) and travels on exactly:
//       an f32 rounding with random low bits gave 0.54 instead of 0.34 as the re-hit rate of a triangle behind a mirror
//       plane at y = -0.6 (a point of that plane has y = -0.6 to an ulp of f64 in the reference), and +4 ... +11 % on
//       axis-aligned boxes seen through a glass sphere
//   f = (low bits of d.y, of d.z, neighbour, 0).  With e.w: the same for the x (e.w), y, z of the DIRECTION of a ray that left a
//       mirror or a dielectric on a triangle or a plane (SRC_D_BIT), which the reference computes from the arriving ray alone
//       (materials.rs:119-183): its last bits are tied to that ray's (reflecting off an axis-aligned face flips one sign and
//       changes nothing else), and that decides what the ray does on the next triangle.  Diffuse and fuzzy bounces add a
//       random vector: their directions stay f32 and get random low bits where they enter f64 arithmetic.  neighbour: for a
//       ray that re-hits a triangle at t ~ 0 (SRC_SELF_BIT) the coplanar partner of that triangle, if there is one (two
//       coincident faces; HIT_MISS otherwise).  Touched only by rays with SRC_D_BIT or SRC_SELF_BIT
struct RayQueue {
    float4* a;
    float4* b;
    float4* c;
    uint4* e;
    uint4* f;
};

// Primitive record, 48 bytes = 3 x float4, stored in BVH leaf order.
//   triangle: r0 = (A.x, A.y, A.z, e1.x) r1 = (e1.y, e1.z, e2.x, e2.y) r2 = (e2.z, bits(gid), bits(PRIM_TRIANGLE), 0)
//             e1 = B - A, e2 = C - A  (the reference's ab / ac, intersections.rs:153-154)
//   sphere:   r0,r1 = 4 doubles (c.x, c.y, c.z, radius)                r2 = (0, bits(gid), bits(PRIM_SPHERE), 0)
// gid = rank of the primitive in the reference's scan order (object index, then triangle
// index): the tie-break key of scene.rs:43 / intersections.rs:170.
//
// meta (int4 per primitive, leaf order): (object index, triangle index or -1, material, uv offset or -1)
// smeta (SMETA_WORDS x float4 = 96 bytes per primitive, leaf order), what shading gathers:
//   [0] = (n.x, n.y, n.z, bits(material | SMETA_HAS_UV | SMETA_SPHERE))
//   triangle: [1..5] = the reference's own f64 operands A, ab = B - A, ac = C - A (intersections.rs:149-154) as
//             nine doubles (A.x A.y | A.z ab.x | ab.y ab.z | ac.x ac.y | ac.z 0): k_shade evaluates the hit point
//             and the self-intersection of the next ray with the reference's f64 operation sequence (see there)
//   sphere:   [1..2] = (c.x c.y | c.z radius) as doubles
//   n = the triangle's unit normal, normalized((B-A) x (C-A)) (intersections.rs:169) evaluated in f64 on
//   the host and rounded once.  The device's edges come from f32 vertices: on a 1M-triangle mesh (edges of
//   0.01 at coordinates of 6) their cross product is off by 1.5e-5, above the 1e-5 bar.  Spheres: n unused.
constexpr uint32_t SMETA_WORDS = 6;
struct DevScene {
    const float4* nodes;      // 4 x 16 bytes per 4-wide quantised node (bvh.h)
    const float4* prims;      // 3 x float4 per primitive
    const int4* meta;
    const float4* smeta;
    const float* uvs;         // 6 floats per primitive that has uv: (u0,v0,u1,v1,u2,v2)
    uint32_t n_prims;         // all primitive records
    uint32_t n_leaf_prims;    // records [0, n_leaf_prims) are referenced by BVH leaves; the rest are big spheres
    uint32_t n_nodes;
    uint32_t n_tris;          // triangles among the primitives (0: k_shade runs its variant without the f64 triangle arithmetic)
    const float4* planes;     // 2 x float4 per plane: (c.xyz, bits(gid)), (n.xyz, bits(material))
    const int32_t* plane_object;
    const double* mat_param64; // the material parameter (fuzziness / refraction index) as given
    const double* planes64;   // 6 doubles per plane: centre, normal as given (the reference's own operands of intersections.rs:84-95)
    uint32_t n_planes;
    const float4* materials;  // 2 x float4 per material: (r,g,b,param), bits(type, texture, 0, 0)
    const int4* textures;     // (texel offset, width, height, 0)
    const uchar4* texels;
    const float4* lights;     // xyz of sample_points[0]
    uint32_t n_lights;
    uint32_t no_emitters;     // no Emissive and no Texture material (whose back face emits): a path's last ray contributes the background or nothing
    float bg[3];
    float eye[3];
    double eye64[3];          // the reference's own f64 eye: origin of the primary rays in k_shade's f64 hit-point arithmetic
    // primary-ray directions are formed in f64 (the reference's type) and rounded once:
    // an f32 camera loses 1e-4 relative in t on grazing rays
    double lle[3], hz[3], vt[3];          // lle = lower_left - eye
    double ll[3];                         // lower_left
    uint32_t width, height;
    uint32_t stack_depth;     // traversal-stack entries a lane can need (bvh.h max_stack)
    uint32_t* spill;          // global overflow of the LDS stacks: [wave][level - LDS_STACK][lane]
    uint32_t spill_levels;    // max(0, stack_depth - LDS_STACK)
};

// Entry points of the primary rays of an 8x8 pixel tile into the BVH (k_tile_entry): up to four child codes (inner node index,
// leaf code, CHILD_EMPTY_CODE = unused), nearest first, such that no ray of the tile can hit anything of the BVH outside their
// subtrees.  table == nullptr: every ray starts at the root.
struct TileEntry {
    const int4* table;        // one entry per local tile of the pass
    uint32_t npix_pad;        // PassParams::npix_pad, inv_npix_pad: path slot -> local tile
    double inv_npix_pad;
};

struct PassParams {
    uint32_t n_paths;         // (number of local tiles in the pass) * 64 * n_samples
    uint32_t npix_pad;        // (number of local tiles in the pass) * 64
    uint32_t ltile_begin;     // first local tile of the pass
    uint32_t tiles_x, tiles_y;
    uint32_t tile_index, tile_count;
    uint32_t s_begin, n_samples;
    uint32_t ssaa_n;
    int32_t jitter;
    float step;
    double step64;            // 1 / ssaa_n in f64 (anti_aliasing.rs:94)
    float off_x, off_y;
    uint32_t seed_lo, seed_hi;
    double inv_npix_pad, inv_tiles_x;   // 1.0 / npix_pad, 1.0 / tiles_x (div_by; 1.0 / ssaa_n is step64)
    uint32_t clear_radiance;  // k_raygen zeroes the paths' radiance words: scenes with lights (shadow rays add to them), passes that shade nothing
};

}  // namespace rtamd
