// Wavefront path-tracing kernels for gfx950 (MI355X).  Hand-written HIP, wave = 64.
//
//   k_raygen       anti_aliasing.rs:93-124 + camera.rs:62-67 + raytrace.rs:32-37
//   k_closest_hit  scene.rs:35-52 + intersections.rs:40-190 (BVH instead of the linear scans)
//   k_shade        materials.rs:28-253 + raytrace.rs:51-96 (iterative form)
//   k_accumulate   raytrace.rs:134-143 (sum of the samples of a pixel)
//   k_resolve      raytrace.rs:144-150 + image.rs:29-43
//
// (file:line are relative to /root/reference/src/raytracer/.)
#include "rt_kernels.h"

#include <cstdlib>

namespace rtamd {
namespace {

// ---------------------------------------------------------------- small helpers
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       // vec3.rs:81-83
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }  // vec3.rs:104-110
// vec3.rs:115-121: a zero vector is returned unchanged
__device__ __forceinline__ f3 normalized(f3 a) {
    float l2 = dot(a, a);
    if (l2 == 0.f) return a;
    float inv = 1.0f / sqrtf(l2);
    return a * inv;
}
__device__ __forceinline__ f3 reflect(f3 v, f3 n) { return v - n * (2.f * dot(v, n)); }   // vec3.rs:129-131
__device__ __forceinline__ bool near_zero(f3 v) { return fabsf(v.x) < 1e-8f && fabsf(v.y) < 1e-8f && fabsf(v.z) < 1e-8f; }  // vec3.rs:55-58

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Queue words, hit records and sample radiance are written once and read once by a later launch: streamed with the non-temporal
// cache policy (RT_NT), so that they do not displace the BVH and the shading records from L2 (same-box A/B: C5 27.75 -> 26.96 ms
// per 64-spp step, C4 120.2 -> 115.8; what IS re-read -- leaf primitives, shading records -- loses with the same policy).
#ifndef RT_NT
#define RT_NT 1
#endif

typedef float v4f_ __attribute__((ext_vector_type(4)));
typedef float v2f_ __attribute__((ext_vector_type(2)));
typedef uint32_t v4u_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if RT_NT
    const v4f_ v = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
#if RT_NT
    const v4u_ v = __builtin_nontemporal_load(reinterpret_cast<const v4u_*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ float2 ld_stream(const float2* p) {
#if RT_NT
    const v2f_ v = __builtin_nontemporal_load(reinterpret_cast<const v2f_*>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if RT_NT
    __builtin_nontemporal_store(v4f_{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
#if RT_NT
    __builtin_nontemporal_store(v4u_{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u_*>(p));
#else
    *p = v;
#endif
}
#ifndef RT_NT_HITS
#define RT_NT_HITS 0   // hit records are written a few lanes at a time as rays finish: L2 combines them (cached: closest-hit -0.5 %)
#endif
#ifndef RT_NT_RAD
#define RT_NT_RAD 1
#endif
__device__ __forceinline__ void st_stream(float2* p, float2 v);
__device__ __forceinline__ void st_stream(float4* p, float4 v);
__device__ __forceinline__ void st_hits(float2* p, float2 v) { if (RT_NT_HITS) st_stream(p, v); else *p = v; }
__device__ __forceinline__ void st_rad(float4* p, float4 v) { if (RT_NT_RAD) st_stream(p, v); else *p = v; }
__device__ __forceinline__ void st_stream(float2* p, float2 v) {
#if RT_NT
    __builtin_nontemporal_store(v2f_{v.x, v.y}, reinterpret_cast<v2f_*>(p));
#else
    *p = v;
#endif
}

// PCG (rxs-m-xs 32): one 32-bit state per path.  The reference's thread_rng stream is not a
// parity target.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ uint32_t rng_next(uint32_t& s) {
    s = s * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
// uniform in [0,1) with 24 bits (rng.gen::<f64>() stand-in)
__device__ __forceinline__ float rng_unit(uint32_t& s) { return (float)(rng_next(s) >> 8) * 0x1p-24f; }

// Uniform direction on the unit sphere.  Same distribution as the reference's
// normalised rejection sample (vec3.rs:40-52) without the divergent loop.
__device__ __forceinline__ f3 random_unit_vector(uint32_t& s) {
    float z = 1.f - 2.f * rng_unit(s);
    float r = sqrtf(fmaxf(0.f, 1.f - z * z));
    float u = rng_unit(s);
    // v_sin_f32 / v_cos_f32 take revolutions
    return mk(r * __builtin_amdgcn_cosf(u), r * __builtin_amdgcn_sinf(u), z);
}
// Uniform point in the unit ball (vec3.rs:40-47): direction * cbrt(u).
__device__ __forceinline__ f3 random_in_unit_sphere(uint32_t& s) {
    f3 d = random_unit_vector(s);
    float u = rng_unit(s);
    float rad = (u > 0.f) ? __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u) * (1.0f / 3.0f)) : 0.f;
    return d * rad;
}

// Wave-level compaction: every lane of the wave must call this (converged).  Lanes with
// `want` get a unique slot; one atomic per wave.
__device__ __forceinline__ uint32_t wave_alloc(bool want, uint32_t per_lane, uint32_t* counter) {
    const uint64_t m = __ballot(want);
    const uint32_t cnt = (uint32_t)__popcll(m);
    uint32_t base = 0;
    if (lane_id() == 0 && cnt) base = atomicAdd(counter, cnt * per_lane);
    base = __builtin_amdgcn_readfirstlane(base);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    return base + rank * per_lane;
}

// ---------------------------------------------------------------- raygen
// One thread per path slot.  Path slot pid = s_local * npix_pad + pl, pl = ltile*64 + lane-in-tile:
// a wave covers one 8x8 pixel tile so that primary rays of a wave are coherent.
// n / d for a divisor the host knows (inv = 1.0 / d): floor((n + 0.5) * inv) is exact for n < 2^51 (the fractional part of
// (n + 0.5) / d keeps 0.5 / d away from the integers, the product's rounding error n 2^-52 / d stays below that) -- four
// instructions instead of the ~35 of a 32-bit division: k_raygen was bound by its arithmetic, not by its stores
// (16 bytes per path at 2.6 TB/s in scenes without triangles)
__device__ __forceinline__ uint32_t div_by(uint32_t n, double inv) { return (uint32_t)(((double)n + 0.5) * inv); }
__device__ __forceinline__ bool slot_to_pixel(const DevScene& sc, const PassParams& pp, uint32_t pl, uint32_t& x, uint32_t& row) {
    const uint32_t lt = pp.ltile_begin + (pl >> 6);
    const uint32_t l = pl & 63u;
    const uint64_t gt = (uint64_t)lt * pp.tile_count + pp.tile_index;
    if (gt >= (uint64_t)pp.tiles_x * pp.tiles_y) return false;
    const uint32_t ty = div_by((uint32_t)gt, pp.inv_tiles_x), tx = (uint32_t)gt - ty * pp.tiles_x;
    x = tx * 8u + (l & 7u);
    row = ty * 8u + (l >> 3);
    return x < sc.width && row < sc.height;
}

#ifndef RT_SHADE_EARLY_MAT
#define RT_SHADE_EARLY_MAT 1
#endif
#ifndef RT_SHADE_PREFETCH_S
#define RT_SHADE_PREFETCH_S 1   // the first word of the next round's shading record is gathered behind the slot claim (see k_shade)
#endif
#ifndef RT_PRIMARY_REFORDER
#define RT_PRIMARY_REFORDER 1
#endif
#ifndef RT_EDGE_TOL
#define RT_EDGE_TOL 1   // k_closest_hit's triangle test accepts a ray within f32 rounding outside an edge (see the leaf step); 0: the bare tests
#endif
#ifndef RT_SPHERE_E
#define RT_SPHERE_E 1   // a hit on a sphere reads the ray's carried f64 origin like a hit on a triangle or a plane (see k_shade, ray64)
#endif
#ifndef RT_SPHERE_RANK32
#define RT_SPHERE_RANK32 2   // scenes with triangles: spheres are ranked by an f32 test, k_shade forms the distance in f64 (1: those in BVH leaves only; 0: f64 discriminants)
#endif
#ifndef RT_NEAR_F64
#define RT_NEAR_F64 1
#endif
#ifndef RT_TIE_SHADE
#define RT_TIE_SHADE 1    // 0: k_shade ignores the runner-up (measurements only)
#endif
#ifndef RT_PRIMARY_REF64
#define RT_PRIMARY_REF64 1   // scenes with triangles: the f64 direction of a primary ray travels exactly (queue word e)
#endif
// The primary ray of path `pid` of a pass: sample position (anti_aliasing.rs:93-124), camera.rs:62-67 in f64 (lower_left +
// horizontal*x + vertical*y - eye) and Ray::new's normalisation (raytrace.rs:32-37); raytrace.rs:128: output row 0 is
// j = height-1.  false: a padding slot of a border tile.  rng: the path's generator after the two jitter draws.
__device__ __forceinline__ void split64(double v, float& hi, uint32_t& lo);
__device__ __forceinline__ bool primary_ray(const DevScene& sc, const PassParams& pp, uint32_t pid, uint32_t& rng, double& dx, double& dy, double& dz, bool exact = true) {
    const uint32_t s_local = div_by(pid, pp.inv_npix_pad);
    const uint32_t pl = pid - s_local * pp.npix_pad;
    uint32_t x = 0, row = 0;
    if (!slot_to_pixel(sc, pp, pl, x, row)) return false;
    const uint32_t s = pp.s_begin + s_local;
    const uint32_t si = div_by(s, pp.step64), sj = s - si * pp.ssaa_n;   // x-major sample order, anti_aliasing.rs:96-100
    rng = pcg_hash(pcg_hash(pcg_hash((row * sc.width + x) ^ pp.seed_lo) + s) ^ pp.seed_hi);
#if RT_PRIMARY_REFORDER
    // the reference's own f64 sequence: base + i * step + gen_range(0.0..step) (anti_aliasing.rs:93-124, 53 random bits),
    // ((lower_left + horizontal * x) + vertical * y) - eye (camera.rs:62-67), x / norm (vec3.rs:115-121)
    {
#pragma clang fp contract(off)
    double fx = (double)si * pp.step64, fy = (double)sj * pp.step64;
    const double bx = (double)x + (double)pp.off_x, by = (double)(sc.height - 1u - row) + (double)pp.off_y;
    double px = bx + fx, py = by + fy;
    if (pp.jitter) {
        const uint32_t a0 = rng_next(rng), a1 = rng_next(rng), b0 = rng_next(rng), b1 = rng_next(rng);
        px += ((double)(a0 >> 5) * 67108864.0 + (double)(a1 >> 6)) * 0x1p-53 * pp.step64;
        py += ((double)(b0 >> 5) * 67108864.0 + (double)(b1 >> 6)) * 0x1p-53 * pp.step64;
    }
    const double ddx = ((sc.ll[0] + sc.hz[0] * px) + sc.vt[0] * py) - sc.eye64[0];
    const double ddy = ((sc.ll[1] + sc.hz[1] * px) + sc.vt[1] * py) - sc.eye64[1];
    const double ddz = ((sc.ll[2] + sc.hz[2] * px) + sc.vt[2] * py) - sc.eye64[2];
    const double l2 = ddx * ddx + ddy * ddy + ddz * ddz;
    if (!exact && l2 > 1e-30 && l2 < 1e30) {
        // scenes without triangles consume the direction rounded to f32 only: 1 / norm from the f32 hardware estimate and one
        // Newton step in f64 (relative error ~1e-14; the rounding to f32 then differs from that of the exact quotient for one
        // component in 10^6), six f64 instructions where the exact form has a square root and a division
        const double y = (double)__builtin_amdgcn_rsqf((float)l2);
        const double r = y * (1.5 - 0.5 * l2 * y * y);
        dx = ddx * r; dy = ddy * r; dz = ddz * r;
        return true;
    }
    const double dl = sqrt(l2);
    if (dl != 0.0) {
        // x / norm for three x: one division, then q = x * r corrected by its own remainder -- the last step of the division
        // algorithm itself, the correctly rounded quotient but for rare double-rounding cases (two divisions less per ray,
        // here and in k_shade's copy: 0.2 ms of a C5 step)
        const double r = 1.0 / dl;
        double q;
        q = ddx * r; dx = __builtin_fma(__builtin_fma(-q, dl, ddx), r, q);
        q = ddy * r; dy = __builtin_fma(__builtin_fma(-q, dl, ddy), r, q);
        q = ddz * r; dz = __builtin_fma(__builtin_fma(-q, dl, ddz), r, q);
    } else { dx = ddx; dy = ddy; dz = ddz; }
    }
    return true;
#else
    float fx = (float)si * pp.step + pp.off_x, fy = (float)sj * pp.step + pp.off_y;
    if (pp.jitter) {                                              // anti_aliasing.rs:117-120, gen_range(0.0..step)
        fx += rng_unit(rng) * pp.step;
        fy += rng_unit(rng) * pp.step;
    }
    const double px = (double)x + (double)fx, py = (double)(sc.height - 1u - row) + (double)fy;
    const double ddx = sc.lle[0] + sc.hz[0] * px + sc.vt[0] * py;
    const double ddy = sc.lle[1] + sc.hz[1] * px + sc.vt[1] * py;
    const double ddz = sc.lle[2] + sc.hz[2] * px + sc.vt[2] * py;
    const double dl = sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
    const double inv = dl != 0.0 ? 1.0 / dl : 1.0;
    dx = ddx * inv; dy = ddy * inv; dz = ddz * inv;
    return true;
#endif
}

// (block0: the first block of the launch that does this work -- the blocks before it find the tiles' entry points, k_raygen_entry)
__device__ __forceinline__ void raygen_body(const DevScene& sc, const PassParams& pp, const RayQueue& q, float4* __restrict__ radiance,
                                            uint32_t* __restrict__ ctl, uint32_t block0) {
    const uint32_t stride = (gridDim.x - block0) * blockDim.x;
    for (uint32_t pid = (blockIdx.x - block0) * blockDim.x + threadIdx.x; pid < pp.n_paths; pid += stride) {   // n_paths % 64 == 0: wave-uniform
        if (pp.clear_radiance) radiance[pid] = make_float4(0.f, 0.f, 0.f, 0.f);   // (scenes without lights: every path stores its radiance exactly once, k_shade)
        // slot == path id: no compaction (one atomic per wave on a single counter costs more than the
        // whole kernel).  Padding slots of border tiles are flagged and skipped downstream.
        // A primary ray is 16 bytes, (d.x, d.y, d.z, bits(rng)): its origin is the eye, its throughput 1,
        // its path id the slot and it starts on nothing; iteration 0 of k_closest_hit and k_shade fill
        // that in.  d == 0 marks a padding slot.  (48-byte primaries cost 2 ms more per 133 M paths.)
        float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t rng;
        double dx, dy, dz;
        if (primary_ray(sc, pp, pid, rng, dx, dy, dz, RT_PRIMARY_REF64 && sc.n_tris != 0)) {
            A = make_float4((float)dx, (float)dy, (float)dz, __uint_as_float(rng));
            if (RT_PRIMARY_REF64 && sc.n_tris) {
                // scenes with triangles: the f64 direction travels exactly, f32 truncation in a, the bits below in e (see
                // RayQueue::e; forming it again in k_shade costs more than these 16 + 16 bytes: registers)
                uint4 lo = make_uint4(0u, 0u, 0u, 0u);
                split64(dx, A.x, lo.x); split64(dy, A.y, lo.y); split64(dz, A.z, lo.z);
                st_stream(&q.e[pid], lo);
            }
        }
        st_stream(&q.a[pid], A);
    }
    if (blockIdx.x == block0 && threadIdx.x == 0) ctl[CTL_QCOUNT + 0] = pp.n_paths;
}
__global__ __launch_bounds__(256) void k_raygen(DevScene sc, PassParams pp, RayQueue q, float4* __restrict__ radiance,
                                                uint32_t* __restrict__ ctl, uint32_t capacity) {
    raygen_body(sc, pp, q, radiance, ctl, 0u);
}

// ---------------------------------------------------------------- closest hit
typedef float v4f __attribute__((ext_vector_type(4)));
struct Best {
    float t;         // scenes with triangles: times 1 + 2 RT_TIE_EPS (see consider)
    uint32_t code;   // HIT_MISS, prim index (leaf order) or HIT_PLANE_BIT | plane
    uint32_t aux;    // scenes with triangles: a candidate whose f32 t is within RT_TIE_EPS of the best one (HIT_MISS: none);
                     // scenes without: the scan rank (gid) of the best one, the tie-break key
};

// scene.rs:43 / intersections.rs:170: strict '>' while scanning in object/triangle order
// == smallest t, then smallest scan rank (gid).
// Scenes with triangles (tris: k_closest_hit's template parameter): two candidates whose f32 distances agree to RT_TIE_EPS are BOTH reported, the
// nearer as the hit, the other as the runner-up -- a box that stands exactly on the ground plane, a box on a box.  In the
// reference the f64 roundings of the two distances decide which surface such a ray meets, ray by ray; the f32 order here
// says nothing about that, so k_shade evaluates both with the reference's f64 sequences (see there).  The distances are
// kept times tie_k = 1 + 2 RT_TIE_EPS, so that the node test, which culls against best.t, cannot drop the subtree of a
// runner-up (one multiplication per accepted candidate instead of one per node step; the hit record carries no distance
// in such scenes, k_shade forms every one in f64).  Scenes without triangles: tie_k = 1, best.t is the distance itself.
#ifndef RT_TIE_EPS
#define RT_TIE_EPS 0x1p-13f
#endif
// hit code without its class bit (HIT_SPHERE_BIT): the plain primitive index / plane code / HIT_MISS / HIT_INVALID
__device__ __forceinline__ uint32_t strip_hit(uint32_t code) {
    return (code >= SRC_SPECIAL || (code & HIT_PLANE_BIT)) ? code : (code & SRC_PRIM_MASK);
}
__device__ __forceinline__ uint32_t gid_of(const DevScene& sc, uint32_t code) {
    return __float_as_uint((code & HIT_PLANE_BIT) ? sc.planes[2 * (size_t)(code & ~HIT_PLANE_BIT)].w : sc.prims[3 * (size_t)code + 2].y);
}
__device__ __forceinline__ void consider(bool tris, float tie_k, Best& best, float t, uint32_t code, uint32_t gid) {
    t *= tie_k;
    const float gap = t - best.t;                                // (-inf while there is no hit)
    if (tris) {
        const bool win = gap < 0.f;
        const uint32_t other = win ? best.code : code;
        uint32_t aux = best.aux;
        if (fabsf(gap) <= RT_TIE_EPS * t) aux = other;
        else if (win) aux = HIT_MISS;
        if ((best.aux >> 30) != 1u) best.aux = aux;                // (a candidate at t ~ 0 keeps the slot: AUX_NEAR_BIT, see the leaf step)
        if (win) { best.t = t; best.code = code; }
    } else if (gap < 0.f || (gap == 0.f && gid < best.aux)) {
        best.t = t; best.code = code; best.aux = gid;
    }
}

// intersections.rs:40-80.  The f32 form of c = |oc|^2 - r^2 cancels catastrophically for the large
// spheres the scenes use (radius 1000 ground, radius 100 light at distance 866), so oc, a, b/2, c and
// the discriminant are formed in f64 (the reference's own number type; a dozen half-rate FMAs).
// from_self: the ray starts on this very sphere, so one root is 0 (rejected by the reference's
// t > 1e-5) and the other is -b/a exactly.
struct SphereQ {
    double a, hb, disc;   // a = d.d, hb = d.oc (= b/2), disc = hb^2 - a*c
};
__device__ __forceinline__ SphereQ sphere_quadratic(const float4 r0, const float4 r1, f3 o, f3 d) {
    const double cx = __hiloint2double(__float_as_int(r0.y), __float_as_int(r0.x));
    const double cy = __hiloint2double(__float_as_int(r0.w), __float_as_int(r0.z));
    const double cz = __hiloint2double(__float_as_int(r1.y), __float_as_int(r1.x));
    const double rad = __hiloint2double(__float_as_int(r1.w), __float_as_int(r1.z));
    const double ocx = (double)o.x - cx, ocy = (double)o.y - cy, ocz = (double)o.z - cz;
    const double dx = d.x, dy = d.y, dz = d.z;
    SphereQ q;
    q.a = dx * dx + dy * dy + dz * dz;
    q.hb = dx * ocx + dy * ocy + dz * ocz;
    q.disc = q.hb * q.hb - q.a * (ocx * ocx + ocy * ocy + ocz * ocz - rad * rad);
    return q;
}
// f64 roots for the big spheres (tested once per ray at refill, all lanes together).  The square root and
// the reciprocal come from the f32 hardware estimates and one Newton step in f64 each (relative error
// ~1e-14, far below the f32 result's half ulp even after the cancellation in -hb -+ sq); the library's
// f64 sqrt and two f64 divisions were ~150 half-rate instructions per refill.
__device__ __forceinline__ double rcp_newton(double a) {
    const double r = (double)__builtin_amdgcn_rcpf((float)a);
    return r * (2.0 - a * r);
}
__device__ __forceinline__ float sphere_hit64(const float4 r0, const float4 r1, f3 o, f3 d, bool from_self) {
    const SphereQ q = sphere_quadratic(r0, r1, o, d);
    const double ia = rcp_newton(q.a);     // a = d.d ~ 1
    if (from_self) {
        const double t = -2.0 * q.hb * ia;
        return t > 1e-5 ? (float)t : -1.f;
    }
    if (!(q.disc >= 0.0)) return -1.f;
    double sq;
    if (q.disc > 1e-30 && q.disc < 1e30) {
        const double y = (double)__builtin_amdgcn_rsqf((float)q.disc);
        sq = q.disc * (y * (1.5 - 0.5 * q.disc * y * y));
    } else {
        sq = sqrt(q.disc);
    }
    const double t1 = (-q.hb - sq) * ia, t2 = (-q.hb + sq) * ia;
    if (t1 > 1e-5) return (float)t1;
    if (t2 > 1e-5) return (float)t2;
    return -1.f;
}
// Spheres inside BVH leaves: f64 discriminant, then f32 square root and division (the cancellation
// is in the discriminant; sqrt/div in f64 cost ~100 instructions for the whole wave).
__device__ __forceinline__ float sphere_hit(const float4 r0, const float4 r1, f3 o, f3 d, bool from_self) {
    const SphereQ q = sphere_quadratic(r0, r1, o, d);
    const float ia = 1.0f / (float)q.a;
    if (from_self) {
        const float t = (float)(-2.0 * q.hb) * ia;
        return t > 1e-5f ? t : -1.f;
    }
    if (!(q.disc >= 0.0)) return -1.f;
    const double sq = (double)sqrtf((float)q.disc);
    const float t1 = (float)(-q.hb - sq) * ia, t2 = (float)(-q.hb + sq) * ia;
    if (t1 > 1e-5f) return t1;
    if (t2 > 1e-5f) return t2;
    return -1.f;
}

// The same test where the distance only RANKS candidates (scenes with triangles: k_shade forms every distance again with the
// reference's f64 sequence, and two candidates within RT_TIE_EPS of each other are both handed to it): all in f32, with the
// discriminant in the form that does not cancel, r^2 - |oc - (oc.d) d|^2 (the squared distance of the centre from the ray,
// d a unit vector to f32 rounding) -- accurate to a few ulp of r^2 for the spheres of BVH leaves (radius below a quarter of the
// scene diagonal; the big ones are tested in f64 at the refill).  22 VALU instructions instead of 112 with 31 in f64.
__device__ __forceinline__ float sphere_rank32(const float4 r0, const float4 r1, f3 o, f3 d, bool from_self) {
    const float cx = (float)__hiloint2double(__float_as_int(r0.y), __float_as_int(r0.x));
    const float cy = (float)__hiloint2double(__float_as_int(r0.w), __float_as_int(r0.z));
    const float cz = (float)__hiloint2double(__float_as_int(r1.y), __float_as_int(r1.x));
    const float rad = (float)__hiloint2double(__float_as_int(r1.w), __float_as_int(r1.z));
    const f3 oc = mk(o.x - cx, o.y - cy, o.z - cz);
    const float hb = dot(oc, d);
    if (from_self) {
        const float t = -2.f * hb;
        return t > 1e-5f ? t : -1.f;
    }
    const f3 l = mk(fmaf(-hb, d.x, oc.x), fmaf(-hb, d.y, oc.y), fmaf(-hb, d.z, oc.z));
    const float disc = fmaf(rad, rad, -dot(l, l));
    if (!(disc >= 0.f)) return -1.f;
    const float sq = sqrtf(disc);
    const float t1 = -hb - sq, t2 = -hb + sq;
    if (t1 > 1e-5f) return t1;
    if (t2 > 1e-5f) return t2;
    return -1.f;
}

// Normal of a sphere hit, intersections.rs:68-69: (o + t d - c) / r.  The f32 t is good to 6e-8 relative, which
// moves the hit point by 6e-8 t: on a sphere of radius 0.01 five units away that is 3e-5 of the normal, above
// the 1e-5 bar.  One Newton step of the quadratic a t^2 + 2 hb t + c in f64 makes t exact to ~1e-14 for this
// purpose; the normal is then formed in f64 and rounded once.
__device__ __forceinline__ f3 sphere_normal(const float4 r0, const float4 r1, f3 o, f3 d, float t) {
    const double cx = __hiloint2double(__float_as_int(r0.y), __float_as_int(r0.x));
    const double cy = __hiloint2double(__float_as_int(r0.w), __float_as_int(r0.z));
    const double cz = __hiloint2double(__float_as_int(r1.y), __float_as_int(r1.x));
    const double rad = __hiloint2double(__float_as_int(r1.w), __float_as_int(r1.z));
    const double ocx = (double)o.x - cx, ocy = (double)o.y - cy, ocz = (double)o.z - cz;
    const double dx = d.x, dy = d.y, dz = d.z;
    const double a = dx * dx + dy * dy + dz * dz, hb = dx * ocx + dy * ocy + dz * ocz;
    const double cc = ocx * ocx + ocy * ocy + ocz * ocz - rad * rad;
    double t64 = (double)t;
    const double f = (a * t64 + 2.0 * hb) * t64 + cc, fp = 2.0 * (a * t64 + hb);
    if (fabs(fp) > 1e-12) t64 -= f * (double)__builtin_amdgcn_rcpf((float)fp);    // |f / fp| ~ 1e-7 t: an f32 reciprocal is plenty
    const double ir = 1.0 / rad;
    return mk((float)((ocx + dx * t64) * ir), (float)((ocy + dy * t64) * ir), (float)((ocz + dz * t64) * ir));
}

// ---- The triangle arithmetic of the reference, in its own number type (f64) and its own operation sequence
// (intersections.rs:123-126,153-172), without FMA contraction.
//
// The reference accepts a triangle hit at t >= 0 with NO epsilon (intersections.rs:166), so a ray that STARTS on a
// triangle re-hits it at t ~ 1e-16 whenever rounding puts its origin on the near side.  That is not a coin toss: the
// origin p = o + d * t0 was computed from the same determinants, and the correlation between the roundings of t0 and
// of det(p - A, ab, ac) gives every triangle and viewpoint its own re-hit probability (measured on the reference's
// algorithm: 0.14 ... 1.0; after a first re-hit 0.6 ... 0.97 for the next one).  A re-hit shades the surface again
// (a Lambertian bounce costs another factor of albedo, a Texture front face becomes an emitting back face), so a
// converged image of a mesh depends on these probabilities.  They can only be reproduced by doing what the reference
// does: k_shade therefore evaluates t0, the hit point and the self-intersection of the ray it emits with the
// reference's f64 sequence on the reference's f64 operands A, ab, ac (smeta), and hands the verdict to the next
// iteration's closest-hit kernel in the ray's src word (SRC_SELF_BIT), together with the f64 hit point (queue word e),
// whose last bits decide the re-hits that follow.  In scenes with triangles the ORIGIN of every ray is exact: each hit
// point (plane, sphere, triangle) is computed with the reference's f64 sequence and carried on as the f32 origin plus its
// 29 lower mantissa bits (RayQueue::e; an f32 origin with random low bits changed the re-hit rate of a triangle behind a
// mirror plane from 0.34 to 0.54).  DIRECTIONS are exact after a mirror or dielectric bounce off a triangle or a plane
// (RayQueue::f, see k_shade) and f32 otherwise; where an f32 direction enters the f64 sequence its 29 missing mantissa bits
// are filled with random bits (dither64): operands with trailing zeros make half of the sequence's products exact and
// shift the probabilities (0.13 instead of 0.36 on one test triangle), while random low bits are what the reference's
// own f64 directions carry.  (The same sequence in f32 -- round 1 -- gave
// 0.05 ... 0.95 where the reference has 0.3; it only happened to agree on scenes/cube, whose coordinates are exact.)
#ifndef RT_FAST_DIV
// 0: the reference's quotients are IEEE divisions.  1 (measured, off): three quotients by one divisor as one division and
// x / m = q + (x - q m) r, q = x r, r = 1 / m (Markstein) -- the correctly rounded quotient for ordinary operands, no faster in
// k_shade (which waits on memory), and WRONG where r overflows: a direction with an exact zero component enters the f64 sequences
// with dither bits as a denormal ~1e-315, the determinants it scales are denormal too, their quotient is an ordinary number and
// r = 1 / m is inf.  Axis-aligned scenes at exactly representable coordinates hit that case: 5 findings in 500 scenes of
// tools/soak_objects.py (objects 7 ... 40 % too bright) until the divisions came back.
#define RT_FAST_DIV 0
#endif
__device__ __forceinline__ double quot_ref(double x, double m, double r) {   // x / m, r = 1 / m (correctly rounded)
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-q, m, x), r, q);
}
struct d3 {
    double x, y, z;
};
__device__ __forceinline__ d3 mkd(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ double det3_ref(d3 v1, d3 v2, d3 v3) {
#pragma clang fp contract(off)
    return (v1.x * v2.y * v3.z + v2.x * v3.y * v1.z + v3.x * v1.y * v2.z) - (v3.x * v2.y * v1.z + v2.x * v1.y * v3.z + v1.x * v3.y * v2.z);
}
__device__ __forceinline__ d3 sub_ref(d3 a, d3 b) {
#pragma clang fp contract(off)
    return mkd(a.x - b.x, a.y - b.y, a.z - b.z);
}
// raytrace.rs:40-42: origin + direction * t, two roundings per component
__device__ __forceinline__ d3 at_ref(d3 o, d3 d, double t) {
#pragma clang fp contract(off)
    return mkd(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
}
__device__ __forceinline__ f3 at_ref(f3 o, f3 d, float t) {   // the same in f32, for hits on spheres and planes
#pragma clang fp contract(off)
    return mk(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
}
// vec3.rs in f64, operation by operation (no contraction): what the specular scatter of k_shade is made of
__device__ __forceinline__ double dot_ref(d3 a, d3 b) {          // vec3.rs:81-83
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ d3 neg_ref(d3 a) { return mkd(-a.x, -a.y, -a.z); }
__device__ __forceinline__ d3 cross_ref(d3 a, d3 b) {            // vec3.rs:104-110
#pragma clang fp contract(off)
    return mkd(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ d3 normalized_ref(d3 v) {             // vec3.rs:115-121
#pragma clang fp contract(off)
    const double n = sqrt(dot_ref(v, v));
#if RT_FAST_DIV
    if (n != 0.0) { const double r = 1.0 / n; return mkd(quot_ref(v.x, n, r), quot_ref(v.y, n, r), quot_ref(v.z, n, r)); }
#else
    if (n != 0.0) return mkd(v.x / n, v.y / n, v.z / n);
#endif
    return v;
}
__device__ __forceinline__ d3 reflect_ref(d3 v, d3 n) {          // vec3.rs:129-131
#pragma clang fp contract(off)
    const double k = 2.0 * dot_ref(v, n);
    return mkd(v.x - n.x * k, v.y - n.y * k, v.z - n.z * k);
}
__device__ __forceinline__ d3 refract_ref(d3 dir, d3 n, double ratio) {   // materials.rs:105-110
#pragma clang fp contract(off)
    const double c = fmin(dot_ref(n, neg_ref(dir)), 1.0);
    const d3 perp = mkd((dir.x + n.x * c) * ratio, (dir.y + n.y * c) * ratio, (dir.z + n.z * c) * ratio);
    const double k = -sqrt(fabs(1.0 - dot_ref(perp, perp)));
    return mkd(perp.x + n.x * k, perp.y + n.y * k, perp.z + n.z * k);
}
// one iteration of the loop of intersections.rs:148-186: a = Ma / M, b = Mb / M, t = -(Mt / M); true = the candidate
// passes the reject test of line 166 (a NaN candidate, M == 0, is not taken: the reference takes it only while it
// has no other hit, and a ray parallel to the triangle it starts on leaves it)
__device__ __forceinline__ bool tri_hit_ref(d3 o, d3 d, d3 A, d3 ab, d3 ac, double& t, double& a, double& b) {
#pragma clang fp contract(off)
    const d3 res = sub_ref(o, A);
    const double m = det3_ref(d, ab, ac), mt = det3_ref(res, ab, ac), ma = det3_ref(d, res, ac), mb = det3_ref(d, ab, res);
#if RT_FAST_DIV
    // three quotients by one divisor: one division, then x / m = q + (x - q m) r with q = x r, r = 1 / m -- the last step of the
    // division algorithm itself (Markstein), the correctly rounded quotient but for divisors whose mantissa is all ones.  M == 0
    // (a ray in the triangle's plane): r = inf and every quotient is NaN where the reference has +-inf or NaN; both fail the
    // comparisons below
    const double r = 1.0 / m;
    a = quot_ref(ma, m, r);
    b = quot_ref(mb, m, r);
    t = -quot_ref(mt, m, r);
#else
    a = ma / m;
    b = mb / m;
    t = -(mt / m);
#endif
    return a >= 0.0 && b >= 0.0 && a + b <= 1.0 && t >= 0.0;
}
// f32 value -> f64 with random bits in the 28 low mantissa bits the f32 does not have (|change| < half an f32 ulp)
__device__ __forceinline__ double dither64(float v, uint32_t h) {
    const double x = (double)v;
    return __hiloint2double(__double2hiint(x), __double2loint(x) ^ (int)(h & 0x0FFFFFFFu));
}
__device__ __forceinline__ d3 dither64(f3 v, uint32_t& h) {
    d3 r;
    h = pcg_hash(h); r.x = dither64(v.x, h);
    h = pcg_hash(h); r.y = dither64(v.y, h);
    h = pcg_hash(h); r.z = dither64(v.z, h);
    return r;
}

// An f64 coordinate as (f32 with the same sign, exponent and upper 23 mantissa bits, the 29 mantissa bits below): the form in
// which a ray's f64 origin travels in the queue (RayQueue::e).  Below the range of normal f32 numbers (|v| < 2^-126) the pair
// holds the f64 itself, less its 8 lowest mantissa bits, behind an f32 exponent field of 0: (sign | payload bits 54..32,
// payload bits 31..0), payload = f64 exponent << 44 | mantissa >> 8 -- to an f32 reader a denormal, i.e. 0.  Such values
// matter: a face in the plane y = 0 is left at y = 1e-15, re-hit at t = 3e-31, again at 7e-47, and the SIGN of that y
// decides when the path gets out (intersections.rs:166 has no epsilon); with these flushed to 0 a glass box whose top face
// lay in y = 0 kept 0.4 % of its paths for ever (tools/soak_objects.py seed 2444).  Above the f32 range: the f32 rounding.
__device__ __forceinline__ void split64(double v, float& hi, uint32_t& lo) {
    const uint32_t hw = (uint32_t)__double2hiint(v), lw = (uint32_t)__double2loint(v);
    const uint32_t e = (hw >> 20) & 0x7FFu;
    if (e >= 1151u) { hi = (float)v; lo = 0u; return; }          // (inf, NaN, |v| >= 2^128: f32 exponent field 255)
    if (e <= 896u) {
        // payload bits: 54..44 = e, 43..24 = hw mantissa (20 bits), 23..0 = lw >> 8
        hi = __uint_as_float((hw & 0x80000000u) | (e << 12) | ((hw & 0xFFFFFu) >> 8));
        lo = ((hw & 0xFFu) << 24) | (lw >> 8);
        return;
    }
    hi = __uint_as_float((hw & 0x80000000u) | ((e - 896u) << 23) | ((hw & 0xFFFFFu) << 3) | (lw >> 29));
    lo = lw & 0x1FFFFFFFu;
}
__device__ __forceinline__ double join64(float hi, uint32_t lo) {
    const uint32_t fb = __float_as_uint(hi);
    const uint32_t fe = (fb >> 23) & 0xFFu;
    if (fe == 255u) return (double)hi;
    if (fe == 0u) {
        const uint32_t e = (fb >> 12) & 0x7FFu;
        const uint32_t hw = (fb & 0x80000000u) | (e << 20) | ((fb & 0xFFFu) << 8) | (lo >> 24);
        return __hiloint2double((int)hw, (int)(lo << 8));
    }
    const uint32_t hw = (fb & 0x80000000u) | ((fe + 896u) << 20) | ((fb >> 3) & 0xFFFFFu);
    return __hiloint2double((int)hw, (int)((fb << 29) | lo));
}

#ifndef RT_DIR_REF64
#define RT_DIR_REF64 1   // scenes with triangles: mirror and dielectric bounces off triangles and planes in f64, their directions carried exactly (RayQueue::f); 2: off spheres too
#endif
constexpr int32_t STACK_SENTINEL = 0x7FFFFFFF;   // bottom of every traversal stack: "ray finished"
#ifndef RT_ENTRY_K
#define RT_ENTRY_K 4     // entry points per tile (k_tile_entry), 4 or 8, in scenes without triangles ...
#endif
#ifndef RT_ENTRY_K_TRIS
#define RT_ENTRY_K_TRIS 8   // ... and with: a pyramid through a mesh meets a near and a far side (same-box A/B, 4 -> 8: C4 -2 %, C5 -0.4 %; C3 +2 %)
#endif
#ifndef RT_ENTRY_SUB
#define RT_ENTRY_SUB 1   // entry points per SUB x SUB part of a tile (1: the 8x8 tile, 2: its 4x4 quarters, 4: 2x2 pixels)
#endif
#ifndef RT_ENTRY_PRIMS
#define RT_ENTRY_PRIMS 1
#endif
#ifndef RT_ENTRY_NEAREST
#define RT_ENTRY_NEAREST 1
#endif
#ifndef RT_ENTRY_MAXH
#define RT_ENTRY_MAXH 4  // a frontier node gives way to its children only if the tile's pyramid reaches at most this many of them
#endif
#ifndef RT_REFILL
#define RT_REFILL 32
#endif
constexpr uint32_t REFILL_IDLE_LANES = RT_REFILL;   // refill the wave once this many lanes have no ray

// Persistent waves with per-LANE ray replacement.  A wave owns a private range of ray indices
// (taken from the iteration's cursor with one atomic per range); a lane whose ray is finished
// takes the next index of the range, so the 64 lanes stay busy although rays need very different
// numbers of traversal steps (with one ray per lane per round, 25 % of the VALU lanes were active
// on incoherent rays: rocprofv3 SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU).
#ifndef RT_CHUNK_MAX
#define RT_CHUNK_MAX 256u
#endif
#ifndef RT_CHUNK_DIV
#define RT_CHUNK_DIV 8u
#endif
#ifndef RT_VOTE_WN
#define RT_VOTE_WN 2    // a node step is chosen when (lanes with a node) * WL >= (lanes with a parked leaf) * WN
#define RT_VOTE_WL 1    // (13:9, 1:1, 2:1, 3:1, 9:13 measured)
#endif
#ifndef RT_HIT_MIN_WAVES
#define RT_HIT_MIN_WAVES 8   // waves per SIMD the register allocator must allow: 64 VGPRs, no spills (6 waves/80 VGPRs: 4.57 ms per launch, 7: 4.38, 8: 4.30)
#endif

// any_hit: the launch serves the LAST rays of the paths of a scene without emitters (no Emissive or Texture material, hence no
// lights): such a ray contributes the background if it escapes and nothing otherwise (raytrace.rs:52-54,95: whatever it hits
// scatters into a ray that returns black), so WHICH surface it hits first does not matter and its traversal ends at the first
// hit it finds.  A candidate at t ~ 0 that f32 cannot decide (AUX_NEAR_BIT) is no such hit: it still goes to k_shade's f64 test.
// GEN: iteration 0 without k_raygen -- a lane forms the primary ray of the path slot it takes (primary_ray; the direction from the
// fast normalisation, which is all a traversal that only ranks candidates needs) instead of reading it from the queue, and
// k_shade<.., 2> forms it again (exactly, in scenes with triangles): 32 bytes per path less to write and 32 to read, one launch
// less, and a refill of primary rays waits for no memory at all.
template <bool COUNT, bool TRIS, bool GEN>   // TRIS: the scene has triangles (what Best::aux holds, see consider)
__global__ __launch_bounds__(256, RT_HIT_MIN_WAVES) void k_closest_hit(DevScene sc, RayQueue q, float2* __restrict__ hits, uint32_t* __restrict__ ctl,
                                                    int it, uint32_t capacity, TileEntry entry, int any_hit, PassParams pp) {
    // per-lane traversal stacks: the first LDS_STACK levels in LDS, laid out [wave][level][lane]
    // (bank = lane: conflict-free), deeper levels (rare) in a global overflow area
    const uint32_t lane = lane_id();
    Best best;
#if RT_POP_CULL
    // an entry = (child code, sort key of the node step that pushed it: the bits of the box's entry distance, slot in the low two);
    // pop drops entries whose box begins beyond the nearest hit so far -- the very test the node step applies to a child
    // (tn <= min(tf, best.t)), repeated with the best.t of now
    typedef uint32_t ent_t __attribute__((ext_vector_type(2)));
    __shared__ ent_t lds_stack[4 * LDS_STACK * 64];
    typedef __attribute__((address_space(3))) ent_t lds_ent;
    lds_ent* const stk = (lds_ent*)lds_stack + (threadIdx.x >> 6) * (LDS_STACK * 64u) + lane;
    ent_t* const spill = reinterpret_cast<ent_t*>(sc.spill) + ((size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * sc.spill_levels) * 64u + lane;
    auto push = [&](uint32_t& sp_, uint32_t v, uint32_t key) {
        if (sp_ < LDS_STACK) stk[sp_ * 64u] = ent_t{v, key};
        else spill[(size_t)(sp_ - LDS_STACK) * 64u] = ent_t{v, key};
        sp_++;
    };
    auto pop = [&](uint32_t& sp_) -> int32_t {
        for (;;) {
            sp_--;
            ent_t v = stk[min(sp_, LDS_STACK - 1u) * 64u];
            if (sp_ >= LDS_STACK) v = spill[(size_t)(sp_ - LDS_STACK) * 64u];
            // (distances are non-negative floats: their bits order as integers; the sentinel's key is 0)
            if ((v.y & ~3u) <= __float_as_uint(best.t)) return (int32_t)v.x;
        }
    };
#else
    __shared__ uint32_t lds_stack[4 * LDS_STACK * 64];
    // address-space-qualified so that stack accesses stay ds_read/ds_write (never flat)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const stk = (lds_u32*)lds_stack + (threadIdx.x >> 6) * (LDS_STACK * 64u) + lane;
    uint32_t* const spill = sc.spill + ((size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * sc.spill_levels) * 64u + lane;
    auto push = [&](uint32_t& sp_, uint32_t v, uint32_t) {
        if (sp_ < LDS_STACK) stk[sp_ * 64u] = v;
        else spill[(size_t)(sp_ - LDS_STACK) * 64u] = v;
        sp_++;
    };
    auto pop = [&](uint32_t& sp_) -> int32_t {
        sp_--;
        // the LDS read is unconditional (clamped index) and the rare deep entry overrides it: a select
        // between an LDS and a global address would turn into a flat load
        uint32_t v = stk[min(sp_, LDS_STACK - 1u) * 64u];
        if (sp_ >= LDS_STACK) v = spill[(size_t)(sp_ - LDS_STACK) * 64u];
        return (int32_t)v;
    };
#endif
    // A launch is sized for the most rays the iteration can have (the host does not wait for the count).  Blocks the
    // actual queue has no ray for leave at once: every wave that stays probes all fetch segments before it gives up, and
    // with thousands of idle waves those atomics -- not the few rays -- set the time of a small launch (the late
    // iterations of deep paths, the shadow-ray generations).
    if ((uint64_t)blockIdx.x * blockDim.x >= (uint64_t)min(ctl[CTL_QCOUNT + it], capacity) + 64u * CTL_FETCH_SEGS) return;
    // planes and big spheres (what every ray tests outside the BVH) staged in LDS
    constexpr uint32_t FIXED_PLANES = 4, FIXED_SPHERES = 4;
    __shared__ v4f fixed[2 * FIXED_PLANES + 3 * FIXED_SPHERES];
    {
        const uint32_t np = min(sc.n_planes, FIXED_PLANES), ns = min(sc.n_prims - sc.n_leaf_prims, FIXED_SPHERES);
        if (threadIdx.x < 2u * np) { const float4 v = sc.planes[threadIdx.x]; fixed[threadIdx.x] = v4f{v.x, v.y, v.z, v.w}; }
        if (threadIdx.x >= 64u && threadIdx.x - 64u < 3u * ns) {
            const float4 v = sc.prims[3 * (size_t)sc.n_leaf_prims + threadIdx.x - 64u];
            fixed[2 * FIXED_PLANES + threadIdx.x - 64u] = v4f{v.x, v.y, v.z, v.w};
        }
        __syncthreads();
    }
    const uint32_t n = min(ctl[CTL_QCOUNT + it], capacity);
    const float4* __restrict__ nodes = sc.nodes;
    const float4* __restrict__ prims = sc.prims;
    uint32_t cnt_nodes = 0, cnt_prims = 0, cnt_wnode = 0, cnt_wleaf = 0, cnt_wrefill = 0, cnt_self = 0, cnt_lanes = 0;   // counters build only
    // The queue is cut into CTL_FETCH_SEGS segments, each with its own cursor.  A wave starts in the
    // segment of its block (consecutive blocks go to different XCDs) and moves on to the next one
    // when a segment is used up, so the cursors see an eighth of the atomics each and the ranges can
    // be small: with one cursor and ranges of 256 the waves finished up to a range apart (4.4 of 6
    // waves per SIMD resident on average over a launch).
    const uint32_t total_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t chunk = min(n >= (1u << 28) ? 2u * RT_CHUNK_MAX : RT_CHUNK_MAX, max(64u, (n / (total_waves * RT_CHUNK_DIV)) & ~63u));   // 128 at 9 M rays, 256 from 17 M on (64..1088 measured), 512 from 268 M on (1 G rays: -0.8 %)
    const uint32_t seg_len = ((n + CTL_FETCH_SEGS - 1) / CTL_FETCH_SEGS + 63u) & ~63u;
    uint32_t seg = blockIdx.x % CTL_FETCH_SEGS, segs_left = CTL_FETCH_SEGS;
    // wave-uniform range state
    uint32_t cur = 0, cur_end = 0;
    bool exhausted = false;
    // per-lane ray state
    bool active = false;
    uint32_t ray = 0, src = SRC_NONE, sp = 0;
    uint32_t pcur = 0, pend = 0;   // primitives of the leaf the lane is in (pcur < pend: leaf mode)
    int32_t node = STACK_SENTINEL;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    float idx = 0.f, idy = 0.f, idz = 0.f;
    constexpr bool tris = TRIS;
    constexpr float tie_k = TRIS ? 1.f + 2.f * RT_TIE_EPS : 1.f;
    best.t = 0.f; best.code = HIT_MISS; best.aux = HIT_MISS;

    for (;;) {
        // ---------------- refill: idle lanes take the next rays of the wave's range
        {
            const uint64_t idle = __ballot(!active);
            if (idle != 0ull) {
                asm volatile("; RTMARK refill");      // (region markers for tools/isa_histogram.py: comments, no code)
                if (COUNT) cnt_wrefill++;
                while (cur >= cur_end && !exhausted) {
                    const uint32_t seg_begin = seg * seg_len, seg_end = min(seg_begin + seg_len, n);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&ctl[CTL_FETCH + seg * CTL_FETCH_STRIDE + it], chunk);
                    base = seg_begin + __builtin_amdgcn_readfirstlane(base);
                    if (base < seg_end) { cur = base; cur_end = min(base + chunk, seg_end); }
                    else {
                        seg = (seg + 1u) % CTL_FETCH_SEGS;
                        if (--segs_left == 0u) exhausted = true;
                    }
                }
                if (cur < cur_end) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    const uint32_t cand = cur + rank;
                    const bool take = !active && cand < cur_end;
                    cur = min(cur + (uint32_t)__popcll(idle), cur_end);
                    if (take) {
                        asm volatile("; RTMARK refill_take");
                        ray = cand;
                        // A refill is a chain of dependent memory round trips during which the wave's other
                        // lanes wait, so it is kept to one: the ray words a and b.  Planes and big spheres come
                        // from LDS; whether the ray re-hits the triangle it starts on was decided by k_shade.
                        float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B;
                        if (!GEN) A = ld_stream(&q.a[ray]);
                        // where the ray enters the BVH: the root, or for a primary ray the entry points of its 8x8 tile (k_tile_entry)
                        int4 F = make_int4(0, CHILD_EMPTY_CODE, CHILD_EMPTY_CODE, CHILD_EMPTY_CODE);
                        auto entry_index = [&]() -> size_t {
                            const uint32_t pl = ray - div_by(ray, entry.inv_npix_pad) * entry.npix_pad;   // tile * 64 + (row in tile) * 8 + column
                            constexpr uint32_t S = RT_ENTRY_SUB, W = 8u / S;
                            return (size_t)(pl >> 6) * (S * S) + (((pl >> 3) & 7u) / W) * S + (pl & 7u) / W;
                        };
                        constexpr int EK = TRIS ? RT_ENTRY_K_TRIS : RT_ENTRY_K;
                        int4 Fx[EK / 4];
#pragma unroll
                        for (int k = 1; k < EK / 4; ++k) Fx[k] = make_int4(CHILD_EMPTY_CODE, CHILD_EMPTY_CODE, CHILD_EMPTY_CODE, CHILD_EMPTY_CODE);
                        if (it == 0 && entry.table) {
                            const int4* __restrict__ fe = entry.table + (EK / 4) * entry_index();
                            F = fe[0];
#pragma unroll
                            for (int k = 1; k < EK / 4; ++k) Fx[k] = fe[k];
                        }
                        if (GEN) {
                            uint32_t rng_;
                            double gx = 0.0, gy = 0.0, gz = 0.0;
                            const bool ok = primary_ray(sc, pp, ray, rng_, gx, gy, gz, false);
                            B = make_float4((float)gy, (float)gz, __uint_as_float(ok ? SRC_NONE : SRC_INVALID), 0.f);
                            A = make_float4(sc.eye[0], sc.eye[1], sc.eye[2], (float)gx);
                        } else if (it == 0) {   // primary ray: (d, rng) only, see k_raygen
                            B = make_float4(A.y, A.z, __uint_as_float(A.x == 0.f && A.y == 0.f && A.z == 0.f ? SRC_INVALID : SRC_NONE), 0.f);
                            A = make_float4(sc.eye[0], sc.eye[1], sc.eye[2], A.x);
                        } else {
                            B = ld_stream(&q.b[ray]);
                        }
                        src = __float_as_uint(B.z);
                        if (src < SRC_SPECIAL) src &= ~(SRC_E_BIT | SRC_D_BIT);   // (k_shade's business: the ray carries its f64 origin / direction)
                        if (src == SRC_INVALID) {
                            st_hits(&hits[ray], make_float2(-1.f, __uint_as_float(HIT_INVALID)));   // padding slot
                        } else if ((src >> 30) == 1u) {
                            // the ray re-hits the triangle it starts on at t ~ 1e-16 (k_shade's verdict, see tri_hit_ref):
                            // nothing can be closer, no traversal
                            st_hits(&hits[ray], make_float2(__uint_as_float(HIT_MISS), __uint_as_float(src & SRC_PRIM_MASK)));   // (self-hits exist in scenes with triangles only: first word = runner-up, none)
                            if (COUNT) cnt_self++;
                        } else {
                            o = mk(A.x, A.y, A.z); d = mk(A.w, B.x, B.y);
                            best.t = __builtin_inff(); best.code = HIT_MISS; best.aux = HIT_MISS;   // (no triangles: gid 0xFFFFFFFF, above every rank)
                            // (a ray that starts on a triangle does not hit it again -- k_shade has decided that -- and the
                            // traversal below skips it)
                            // (the pointer is laundered so that the reads stay here: hoisted out of the loop
                            // they would occupy 80 registers for the whole kernel)
                            typedef __attribute__((address_space(3))) v4f lds_f4;
                            auto ld = [](lds_f4* q_) { const v4f v = *q_; return make_float4(v.x, v.y, v.z, v.w); };
                            lds_f4* fx = (lds_f4*)fixed;
                            asm volatile("" : "+v"(fx));
                            // planes: short linear list (they are unbounded).  intersections.rs:82-104
                            for (uint32_t p = 0; p < sc.n_planes; ++p) {
                                float4 P0, P1;
                                if (p < FIXED_PLANES) { P0 = ld(fx + 2 * p); P1 = ld(fx + 2 * p + 1); }
                                else { P0 = sc.planes[2 * p]; P1 = sc.planes[2 * p + 1]; }
                                const f3 nrm = mk(P1.x, P1.y, P1.z);
                                const float dn = dot(nrm, d);
                                if (fabsf(dn) < 1e-6f) continue;
                                if (src == (HIT_PLANE_BIT | p)) continue;   // t would be ~0 < 1e-5 in the reference's f64
                                const float t = dot(mk(P0.x, P0.y, P0.z) - o, nrm) / dn;
                                if (!(t >= 1e-5f)) continue;
                                consider(tris, tie_k, best, t, HIT_PLANE_BIT | p, __float_as_uint(P0.w));
                            }
                            // big spheres (kept out of the BVH), full f64
                            for (uint32_t pi = sc.n_leaf_prims; pi < sc.n_prims; ++pi) {
                                const uint32_t k = pi - sc.n_leaf_prims;
                                float4 r0, r1, r2;
                                if (k < FIXED_SPHERES) { r0 = ld(fx + 2 * FIXED_PLANES + 3 * k); r1 = ld(fx + 2 * FIXED_PLANES + 3 * k + 1); r2 = ld(fx + 2 * FIXED_PLANES + 3 * k + 2); }
                                else { r0 = prims[3 * pi]; r1 = prims[3 * pi + 1]; r2 = prims[3 * pi + 2]; }
                                if (COUNT) cnt_prims++;
                                // (scenes with triangles: the distance only ranks, see sphere_rank32; for a radius-1000 sphere its f32
                                // discriminant is good to ~3e-4 of a distance of 12, inside the band RT_TIE_EPS hands to k_shade)
                                const float t = (TRIS && RT_SPHERE_RANK32 > 1) ? sphere_rank32(r0, r1, o, d, src == pi) : sphere_hit64(r0, r1, o, d, src == pi);
                                if (t > 0.f) consider(tris, tie_k, best, t, tris ? pi | HIT_SPHERE_BIT : pi, __float_as_uint(r2.y));
                            }
                            if (sc.n_leaf_prims && F.x != CHILD_EMPTY_CODE && !(any_hit && best.code != HIT_MISS)) {
                                const float ooeps = 0x1p-80f;
                                // (v_rcp_f32, 1 ulp: the node test's rounding bound has room for it, see there;
                                // three IEEE divisions were 30 instructions per refill)
                                idx = __builtin_amdgcn_rcpf(fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
                                idy = __builtin_amdgcn_rcpf(fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
                                idz = __builtin_amdgcn_rcpf(fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
#if RT_POP_CULL
                                stk[0] = ent_t{(uint32_t)STACK_SENTINEL, 0u};
#else
                                stk[0] = (uint32_t)STACK_SENTINEL;
#endif
                                sp = 1;
#pragma unroll
                                for (int k = EK / 4 - 1; k >= 1; --k) {
                                    if (Fx[k].w != CHILD_EMPTY_CODE) push(sp, (uint32_t)Fx[k].w, 0u);
                                    if (Fx[k].z != CHILD_EMPTY_CODE) push(sp, (uint32_t)Fx[k].z, 0u);
                                    if (Fx[k].y != CHILD_EMPTY_CODE) push(sp, (uint32_t)Fx[k].y, 0u);
                                    if (Fx[k].x != CHILD_EMPTY_CODE) push(sp, (uint32_t)Fx[k].x, 0u);
                                }
                                if (F.w != CHILD_EMPTY_CODE) push(sp, (uint32_t)F.w, 0u);
                                if (F.z != CHILD_EMPTY_CODE) push(sp, (uint32_t)F.z, 0u);
                                if (F.y != CHILD_EMPTY_CODE) push(sp, (uint32_t)F.y, 0u);
                                node = F.x;          // (the root is always an inner node; a tile's entry point may be a leaf)
                                pcur = pend = 0;
                                if (node < 0) {
                                    const uint32_t code = (uint32_t)~node;
                                    pcur = code >> 3;
                                    pend = pcur + (code & 7u) + 1u;
                                    node = pop(sp);
                                }
                                active = true;
                            } else {
                                st_hits(&hits[ray], make_float2(tris ? __uint_as_float(best.aux) : best.t, __uint_as_float(best.code)));
                            }
                        }
                    }
                }
            }
        }
        asm volatile("; RTMARK control");
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---------------- traversal steps until enough lanes ran out of work.
        // Every step is ONE memory round trip: a lane is either at an inner node (four 16-byte words)
        // or at one primitive of a leaf (three words).
        for (;;) {
            // A lane holds up to two pieces of work: an inner node to test (node) and a leaf whose
            // primitives are pending (pcur < pend).  A ray that reaches a leaf parks it and goes on with
            // the next node from its stack, so it takes part in node steps AND in leaf steps; the order
            // in which a ray's nodes and leaves are done does not change the closest hit.
            //   node: inner node index | STACK_SENTINEL (stack used up) | < 0: a second leaf, waiting
            //         for the parked one to be finished (the lane then sits node steps out)
            const bool leafm = active && pcur < pend;
            const bool nodem = active && (uint32_t)node < (uint32_t)STACK_SENTINEL;
            // The wave runs EITHER the node code or the leaf code in a step, whichever has more lanes
            // waiting (weighted by what the two cost).  Running both in every step costs the sum of the
            // two for the work of one.  (The choice is wave-uniform, so each side is straight-line code
            // behind a scalar branch; masking the two sides of a merged step by the same vote measured
            // 12 % slower.)
            const uint32_t nn = (uint32_t)__popcll(__ballot(nodem)), nl = (uint32_t)__popcll(__ballot(leafm));
            const bool do_node = nn * RT_VOTE_WL >= nl * RT_VOTE_WN;
            if (COUNT) { if (do_node) cnt_wnode++; else cnt_wleaf++; cnt_lanes += (do_node ? nodem : leafm) ? 1u : 0u; }
            if (do_node && nodem) {
                asm volatile("; RTMARK node");
                int32_t nxt;
                const float4* __restrict__ base = nodes + 4 * (size_t)node;
                const float4 L0 = base[0], L1 = base[1], L2 = base[2], L3 = base[3];
                // 4-wide node, 8-bit child boxes relative to (p, 2^e) -- bvh.h
                if (COUNT) cnt_nodes++;
                const uint32_t ex = __float_as_uint(L0.w);
                // scale_x: the shift moves the low bit of the next exponent byte into the sign, |.| drops it
                // (a free source modifier); scale_y and scale_z are stored as floats
                const float sx = fabsf(__uint_as_float(ex << 23)), sy = L3.z, sz = L3.w;
                const float pox = L0.x - o.x, poy = L0.y - o.y, poz = L0.z - o.z;
                // plane distance t = q * (s * idir) + (p - o) * idir: s is a power of two, so s * idir and
                // q * (s * idir) are exact and one FMA per plane carries two roundings in all
                const float sxi = sx * idx, syi = sy * idy, szi = sz * idz;
                const float bxi = pox * idx, byi = poy * idy, bzi = poz * idz;
                // Bound of the rounding error of those distances, PER AXIS: 2^-21 (|p-o| + 256 s) |idir|,
                // moved into the constant term of the near (-) and far (+) planes.  One bound for all three
                // axes (their maximum) is just as safe but useless: an axis the ray is almost parallel to
                // has a huge |idir|, and its bound would then loosen the other two slabs so far that such
                // a ray walks most of the tree (those few rays were the multi-ms tail of a launch).
                // ((|p-o| + 256 s) |idir| from the products that exist anyway: |b| + 256 |s idir|)
                const float wx = fmaf(256.f, fabsf(sxi), fabsf(bxi)), wy = fmaf(256.f, fabsf(syi), fabsf(byi)), wz = fmaf(256.f, fabsf(szi), fabsf(bzi));
                const float bnx = fmaf(wx, -0x1p-21f, bxi), bny = fmaf(wy, -0x1p-21f, byi), bnz = fmaf(wz, -0x1p-21f, bzi);
                const float bfx = fmaf(wx, 0x1p-21f, bxi), bfy = fmaf(wy, 0x1p-21f, byi), bfz = fmaf(wz, 0x1p-21f, bzi);
                // entry / exit planes by the sign of the ray direction
                const uint32_t qlx = __float_as_uint(L2.x), qly = __float_as_uint(L2.y), qlz = __float_as_uint(L2.z);
                const uint32_t qhx = __float_as_uint(L2.w), qhy = __float_as_uint(L3.x), qhz = __float_as_uint(L3.y);
                const uint32_t enx = idx >= 0.f ? qlx : qhx, exx = idx >= 0.f ? qhx : qlx;
                const uint32_t eny = idy >= 0.f ? qly : qhy, exy = idy >= 0.f ? qhy : qly;
                const uint32_t enz = idz >= 0.f ? qlz : qhz, exz = idz >= 0.f ? qhz : qlz;
                uint32_t key[4];
                int32_t cc[4] = {__float_as_int(L1.x), __float_as_int(L1.y), __float_as_int(L1.z), __float_as_int(L1.w)};
                uint32_t nhit = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float tnx = fmaf((float)((enx >> (8 * c)) & 0xFFu), sxi, bnx);
                    const float tny = fmaf((float)((eny >> (8 * c)) & 0xFFu), syi, bny);
                    const float tnz = fmaf((float)((enz >> (8 * c)) & 0xFFu), szi, bnz);
                    const float tfx = fmaf((float)((exx >> (8 * c)) & 0xFFu), sxi, bfx);
                    const float tfy = fmaf((float)((exy >> (8 * c)) & 0xFFu), syi, bfy);
                    const float tfz = fmaf((float)((exz >> (8 * c)) & 0xFFu), szi, bfz);
                    const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), 0.f);
                    const float tf = fminf(fminf(fminf(tfx, tfy), tfz), best.t);
                    const bool h = tn <= tf && cc[c] != CHILD_EMPTY_CODE;
                    nhit += h ? 1u : 0u;
                    // sortable key: distance bits (non-negative float == monotonic uint) with the slot in the low bits
                    key[c] = h ? ((__float_as_uint(tn) & 0x7FFFFFFCu) | (uint32_t)c) : 0xFFFFFFFFu;
                }
                // sort the four (key, child) pairs by distance: 5-comparator network
#define RT_CSWAP(a, b)                                                   \
    {                                                                    \
        const bool sw = key[b] < key[a];                                 \
        const uint32_t ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b]; \
        const int32_t ca = sw ? cc[b] : cc[a], cb = sw ? cc[a] : cc[b];   \
        key[a] = ka; key[b] = kb; cc[a] = ca; cc[b] = cb;                \
    }
                RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
                if (nhit == 0u) {
                    nxt = pop(sp);
                } else {
                    // nearest child next; the others wait on the stack, nearest of them on top
                    if (nhit > 3u) push(sp, (uint32_t)cc[3], key[3]);
                    if (nhit > 2u) push(sp, (uint32_t)cc[2], key[2]);
                    if (nhit > 1u) push(sp, (uint32_t)cc[1], key[1]);
                    nxt = cc[0];
                }
                if (nxt < 0 && !leafm) {
                    // a leaf (~nxt = (first << 3) | (count - 1)) and nothing parked: park it, go on with the stack
                    const uint32_t code = (uint32_t)~nxt;
                    pcur = code >> 3;
                    pend = pcur + (code & 7u) + 1u;
                    nxt = pop(sp);
                }
                node = nxt;
                asm volatile("; RTMARK control");
            }
            if (!do_node && leafm) {
                asm volatile("; RTMARK leaf");
                const float4* __restrict__ base = prims + 3 * (size_t)pcur;
                const float4 L0 = base[0], L1 = base[1], L2 = base[2];   // (cached: the same policy as the queues here costs 9 % (C5) / 23 % (C4) of the kernel)
                if (COUNT) cnt_prims++;
                const uint32_t pi = pcur;
                const uint32_t gid = __float_as_uint(L2.y);
                if (__float_as_uint(L2.z) == PRIM_TRIANGLE) {
                    if (pi != src) {   // a ray does not hit the triangle it leaves (k_shade's verdict, SRC_SELF_BIT otherwise)
                        asm volatile("; RTMARK leaf_tri");
                        // intersections.rs:148-168 (Cramer's rule == Moeller-Trumbore):
                        // a = Ma/M, b = Mb/M, t = -Mt/M; reject a<0 || b<0 || a+b>1 || t<0 (no epsilon)
                        const f3 v0 = mk(L0.x, L0.y, L0.z), e1 = mk(L0.w, L1.x, L1.y), e2 = mk(L1.z, L1.w, L2.x);
                        const f3 pv = cross(d, e2);
                        const float det = dot(e1, pv);
                        const f3 tv = o - v0;
                        const f3 qv = cross(tv, e1);
                        // numerators of a, b, t with the sign of det folded in: the four reject tests
                        // need no division (a < 0 <=> ua*sgn(det) < 0, a + b > 1 <=> ua + ub > |det|)
                        const float sg = det < 0.f ? -1.f : 1.f;
                        const float ua = dot(tv, pv) * sg, ub = dot(d, qv) * sg, ut = dot(e2, qv) * sg, ad = fabsf(det);
                        // The three edge tests with a margin.  The triangle tested here is not the reference's: its vertices are rounded to
                        // f32 (6e-7 at coordinates of 10, 3e-5 of a 0.02-wide triangle), and so are the ray and the products above.  With the
                        // bare tests a ray within that distance of a shared edge could be accepted by the wrong neighbour -- or by NEITHER, and
                        // go on through a closed mesh to whatever lies behind (C4: 42 and 74 of 2 M primary rays, tools/id_mismatch_probe.py).
                        // The margin, 2^-25 |o - v0| max(|e1|, |e2|) in 1-norms (the last two factors from the host, L2.w), is the largest
                        // error seen on C4 (6e-5 of a triangle's width); it makes the triangles overlap instead: near a shared edge both are
                        // candidates at one distance, hit and runner-up, and k_shade keeps the one the reference's f64 test accepts
                        // (resolve_tie).  It is capped at 1/512 of the triangle's width: for a grazing ray |det| is small, the uncapped margin
                        // many per cent of the triangle, and a triangle the ray misses then hides the hit behind it.  C4's middle 1080p: 116 -> 0
                        // id mismatches, C5: 6 -> 4 (a ray through a vertex, where more than two triangles tie, and the horizon of the ground
                        // sphere); uncapped 1, four times the margin 10.  A term in |o|, for the rounding of vertices far from the origin, was taken out again:
                        // at coordinates of 1000 the f32 distances are no better than RT_TIE_EPS, a candidate beyond a crease then outranks
                        // the true hit instead of tying with it, and the room of test_a_closed_mesh_room_far_from_the_origin leaked (+3 %).
                        // PRIMARY RAYS ONLY.  A candidate that only the margin lets in and that has no partner at its distance is taken as it
                        // is, and among the rays that start on surfaces there are whole families for which that is wrong: the shadow rays from
                        // a box face to a light that lies in the plane of that face run along the neighbouring faces' edges, the reference
                        // itself decides each of them by the rounding of a 1e-16, the bare f32 test by the rounding of a 1e-7 -- a fair coin
                        // either way -- and the margin by "hit" (tools/soak_objects.py, SOAK_TILT=1, seed 21018: a glass box 1.2 % too dark).
                        // From the eye no such family exists once the samples are jittered.  The later rays keep the bare tests and with
                        // them the cracks (4e-5 of the rays that reach a mesh of 0.02-wide triangles at a distance of 13).
                                                const float tol = (RT_EDGE_TOL && it == 0) ? fminf(L2.w * (fabsf(tv.x) + fabsf(tv.y) + fabsf(tv.z)), ad * (1.f / 512.f)) : 0.f;
                        if (ad > 0.f && ua >= -tol && ub >= -tol && ua + ub <= ad + tol && ut >= 0.f) {
                            const float inv = 1.0f / ad;
                            const float t = ut * inv;
                            // A triangle at t ~ 0: from a point ON a surface f32 cannot tell the reference's t = +1e-16 (a
                            // hit, intersections.rs:166 has no epsilon) from -1e-16.  k_shade has decided those in f64 for the
                            // triangle the ray starts on and for a coplanar neighbour (SRC_SELF_BIT); what is left here are
                            // the neighbour of a ray that k_shade found to LEAVE both, third faces in the same plane -- and
                            // true hits this close: the adjacent face at a concave crease, a mesh touching the surface the
                            // ray starts on.  The bound grows with the coordinates (0.02 at 1000), so such a candidate is not
                            // dropped (round 2 did: paths leaked through creases of meshes far from the origin) but handed
                            // to k_shade beside the nearest hit beyond the bound, in the runner-up word with AUX_NEAR_BIT;
                            // k_shade evaluates both with the reference's f64 sequence and keeps what the reference keeps
                            // (resolve_tie).  One slot: of several such candidates the last one tested stays.
                            if (t > 0x1p-17f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)))
                                consider(tris, tie_k, best, t, pi, gid);   // the barycentrics are recomputed by k_shade where a texture needs them
                            else if (tris && RT_NEAR_F64)
                                best.aux = pi | AUX_NEAR_BIT;              // f32 cannot rank it: k_shade's f64 test decides (resolve_tie)
                        }
                    }
                } else {
                    // (o and d are laundered so that their f64 copies are made here: hoisted out of the loop
                    // they occupy 14 registers for the whole kernel, the difference between 6 and 8 waves/SIMD)
                    f3 o_ = o, d_ = d;
                    asm volatile("; RTMARK leaf_sphere" : "+v"(o_.x), "+v"(o_.y), "+v"(o_.z), "+v"(d_.x), "+v"(d_.y), "+v"(d_.z));
                    const float t = (TRIS && RT_SPHERE_RANK32) ? sphere_rank32(L0, L1, o_, d_, src == pi) : sphere_hit(L0, L1, o_, d_, src == pi);
                    if (t > 0.f) consider(tris, tie_k, best, t, tris ? pi | HIT_SPHERE_BIT : pi, gid);
                }
                asm volatile("; RTMARK leaf_end");
                if (any_hit && best.code != HIT_MISS) { node = STACK_SENTINEL; pend = pcur; }   // a hit: that is all the path's last ray asks
                pcur++;
                if (pcur == pend && node < 0) {
                    // the leaf that was waiting moves in, the stack gives the next piece of work
                    const uint32_t code = (uint32_t)~node;
                    pcur = code >> 3;
                    pend = pcur + (code & 7u) + 1u;
                    node = pop(sp);
                }
                asm volatile("; RTMARK control");
            }
            if (active && node == STACK_SENTINEL && pcur >= pend) {
                st_hits(&hits[ray], make_float2(tris ? __uint_as_float(best.aux) : best.t, __uint_as_float(best.code)));
                active = false;
            }
            const uint32_t n_active = (uint32_t)__popcll(__ballot(active));
            if (n_active == 0u) break;
            if (!(exhausted && cur >= cur_end) && n_active <= 64u - REFILL_IDLE_LANES) break;   // go and refill
        }
    }
    if (COUNT) {
        // wave reduction of the per-lane counters, one 64-bit add per wave and counter; the per-wave counters are
        // wave-uniform, lane 0 holds them
        uint64_t cn = cnt_nodes, cp = cnt_prims, cs = cnt_self, cl = cnt_lanes;
        for (int off = 32; off > 0; off >>= 1) {
            cn += __shfl_down(cn, off);
            cp += __shfl_down(cp, off);
            cs += __shfl_down(cs, off);
            cl += __shfl_down(cl, off);
        }
        if (lane == 0) {
            unsigned long long* st64 = reinterpret_cast<unsigned long long*>(&ctl[CTL_STAT64]);
            atomicAdd(st64 + 0, (unsigned long long)cn);
            atomicAdd(st64 + 1, (unsigned long long)cp);
            atomicAdd(st64 + 2, (unsigned long long)cnt_wnode);
            atomicAdd(st64 + 3, (unsigned long long)cnt_wleaf);
            atomicAdd(st64 + 4, (unsigned long long)cnt_wrefill);
            atomicAdd(st64 + 5, (unsigned long long)cs);
            atomicAdd(st64 + 6, (unsigned long long)cl);
        }
    }
}

// ---------------------------------------------------------------- entry points of the primary rays
// The 64 primary rays of a wave are the samples of one 8x8 pixel tile: a thin pyramid from the eye (2.6 mrad wide at 1080p,
// fovy 20).  Walking the top of the BVH once per TILE instead of once per ray: one thread per tile descends from the root with a
// conservative pyramid / box test and keeps a frontier of at most four children (inner nodes or leaves) outside whose subtrees no
// ray of the tile can hit anything; a node of the frontier is replaced by those of its children the pyramid reaches while the
// frontier still fits.  The rays of the tile then start their own traversal at the frontier (k_closest_hit, iteration 0) instead
// of at the root, and a tile whose pyramid reaches nothing (sky, ground beyond the last sphere) needs no traversal at all.
// The test: a box is cut off if it lies wholly on the outer side of one of the pyramid's four side planes (all through the eye),
// or farther from the eye than the farthest point at which any ray of the tile can still find a hit -- known when all four
// corner rays hit the same plane or the same big sphere (the things every ray tests outside the BVH; both are convex, so every
// ray between the corners hits them too, nearer than the farthest corner hit).  All in f64 on the exact child boxes (p + q 2^e),
// with the tile widened by RT_ENTRY_PAD pixels on every side: far more than the f32 rounding of a ray direction (6e-8) or the
// slack of the f32 primitive tests, so the frontier never loses a hit the per-ray traversal from the root would find.
#ifndef RT_ENTRY_PAD
#define RT_ENTRY_PAD 0.05
#endif
__device__ __forceinline__ void tile_entry_body(const DevScene& sc, const PassParams& pp, int4* __restrict__ table) {
    constexpr uint32_t S = RT_ENTRY_SUB, SW = 8u / S;
    const uint32_t te = blockIdx.x * 64u + threadIdx.x;   // (local tile of the pass, part of the tile): 64 per block
    if (te >= (pp.npix_pad >> 6) * (S * S)) return;
    const uint32_t tl = te / (S * S), sub = te % (S * S);
    const int32_t EMPTY = CHILD_EMPTY_CODE;
    const uint64_t gt = (uint64_t)(pp.ltile_begin + tl) * pp.tile_count + pp.tile_index;
    constexpr int KMAX = RT_ENTRY_K > RT_ENTRY_K_TRIS ? RT_ENTRY_K : RT_ENTRY_K_TRIS;
    const int K = sc.n_tris ? RT_ENTRY_K_TRIS : RT_ENTRY_K;
    int4* const out = table + (size_t)te * (K / 4);
    if (gt >= (uint64_t)pp.tiles_x * pp.tiles_y || sc.n_leaf_prims == 0) {
        for (int k = 0; k < K / 4; ++k) out[k] = make_int4(EMPTY, EMPTY, EMPTY, EMPTY);
        return;
    }
    const uint32_t ty = (uint32_t)(gt / pp.tiles_x), tx = (uint32_t)(gt - (uint64_t)ty * pp.tiles_x);
    const uint32_t x_lo = min(tx * 8u + (sub % S) * SW, sc.width), x_hi = min(x_lo + SW, sc.width);
    const uint32_t r_lo = min(ty * 8u + (sub / S) * SW, sc.height), r_hi = min(r_lo + SW, sc.height);
    // sample positions of the tile (primary_ray): px = x + off_x + [0, 1], py = (height - 1 - row) + off_y + [0, 1]
    const double pxl = (double)x_lo + (double)pp.off_x - RT_ENTRY_PAD, pxh = (double)x_hi + (double)pp.off_x + RT_ENTRY_PAD;
    const double pyl = (double)(sc.height - r_hi) + (double)pp.off_y - RT_ENTRY_PAD, pyh = (double)(sc.height - r_lo) + (double)pp.off_y + RT_ENTRY_PAD;
    const d3 eye = mkd(sc.eye64[0], sc.eye64[1], sc.eye64[2]);
    auto dir = [&](double px, double py) {
        return mkd(sc.ll[0] + sc.hz[0] * px + sc.vt[0] * py - eye.x, sc.ll[1] + sc.hz[1] * px + sc.vt[1] * py - eye.y, sc.ll[2] + sc.hz[2] * px + sc.vt[2] * py - eye.z);
    };
    auto dotd = [](d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto crossd = [](d3 a, d3 b) { return mkd(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); };
    d3 c[4] = {dir(pxl, pyl), dir(pxh, pyl), dir(pxh, pyh), dir(pxl, pyh)};
    // inward normals of the four side planes (edge k, k+1; the corner opposite decides the orientation)
    d3 pn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d3 n = crossd(c[k], c[(k + 1) & 3]);
        if (dotd(n, c[(k + 2) & 3]) < 0.0) n = mkd(-n.x, -n.y, -n.z);
        pn[k] = n;
    }
    // the farthest distance at which a ray of the tile can still find its closest hit
    double tmax = __builtin_inf();
    {
        d3 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double l = sqrt(dotd(c[k], c[k]));
            u[k] = l > 0.0 ? mkd(c[k].x / l, c[k].y / l, c[k].z / l) : c[k];
        }
        for (uint32_t p = 0; p < sc.n_planes; ++p) {                      // intersections.rs:82-104
            const double* __restrict__ P = sc.planes64 + 6 * (size_t)p;
            const d3 pc = mkd(P[0] - eye.x, P[1] - eye.y, P[2] - eye.z), n = mkd(P[3], P[4], P[5]);
            const double num = dotd(pc, n);
            double far = 0.0;
            bool all = true;
            const double dn0 = dotd(n, u[0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double dn = dotd(n, u[k]);
                const double t = num / dn;
                all = all && fabs(dn) > 1e-4 && (dn > 0.0) == (dn0 > 0.0) && t > 1e-3;
                far = fmax(far, t);
            }
            if (all) tmax = fmin(tmax, far);
        }
        for (uint32_t pi = sc.n_leaf_prims; pi < sc.n_prims; ++pi) {      // big spheres, intersections.rs:40-80
            const double2* __restrict__ S = reinterpret_cast<const double2*>(sc.prims + 3 * (size_t)pi);
            const double2 s0 = S[0], s1 = S[1];
            const d3 oc = mkd(eye.x - s0.x, eye.y - s0.y, eye.z - s1.x);
            const double r2 = s1.y * s1.y, cc = dotd(oc, oc) - r2;
            if (!(cc > 1e-6 * r2)) continue;                              // the eye is inside or on the sphere: no bound from it
            double far = 0.0;
            bool all = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double hb = dotd(oc, u[k]), disc = hb * hb - cc;
                const double t = -hb - sqrt(fmax(disc, 0.0));
                all = all && disc > 0.0 && t > 1e-3;
                far = fmax(far, t);
            }
            if (all) tmax = fmin(tmax, far);
        }
    }
    const double tmax2 = tmax < 1e150 ? tmax * tmax * (1.0 + 4e-3) : __builtin_inf();
    // frontier: codes, squared distance of the box from the eye (the order in which the rays will try them), "not to be expanded"
    // (the frontier lives in LDS, [entry][thread]: indexed by run-time values, registers would turn into scratch memory)
    __shared__ int32_t s_code[KMAX][64];
    __shared__ double s_key[KMAX][64];
    const uint32_t tx_ = threadIdx.x;
#define code(k) s_code[k][tx_]
#define key(k) s_key[k][tx_]
    for (int k = 0; k < KMAX; ++k) { code(k) = EMPTY; key(k) = 0.0; }
    code(0) = 0;
    uint32_t fin = 0u;   // bit k: entry k is not to be expanded
    int n = 1;
    for (int iter = 0; iter < 96; ++iter) {
        // nearest first: most rays end in the nearest part of what the pyramid reaches, so that part gets the entry points
        int i = -1;
        double ki = 0.0;
        for (int k = 0; k < n; ++k) {
            const double kk = key(k);
            if (code(k) >= 0 && !((fin >> k) & 1u) && (i < 0 || (RT_ENTRY_NEAREST && kk < ki))) { i = k; ki = kk; }
        }
        if (i < 0) break;
        const float4* __restrict__ base = sc.nodes + 4 * (size_t)code(i);
        const float4 L0 = base[0], L1 = base[1], L2 = base[2], L3 = base[3];
        const double sx = (double)fabsf(__uint_as_float(__float_as_uint(L0.w) << 23)), sy = (double)L3.z, sz = (double)L3.w;
        const uint32_t ql[3] = {__float_as_uint(L2.x), __float_as_uint(L2.y), __float_as_uint(L2.z)};
        const uint32_t qh[3] = {__float_as_uint(L2.w), __float_as_uint(L3.x), __float_as_uint(L3.y)};
        const int32_t ch[4] = {__float_as_int(L1.x), __float_as_int(L1.y), __float_as_int(L1.z), __float_as_int(L1.w)};
        bool hit[4];
        double hk[4];
        int nh = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const d3 lo = mkd((double)L0.x + (double)((ql[0] >> (8 * k)) & 0xFFu) * sx, (double)L0.y + (double)((ql[1] >> (8 * k)) & 0xFFu) * sy,
                              (double)L0.z + (double)((ql[2] >> (8 * k)) & 0xFFu) * sz);
            const d3 hi = mkd((double)L0.x + (double)((qh[0] >> (8 * k)) & 0xFFu) * sx, (double)L0.y + (double)((qh[1] >> (8 * k)) & 0xFFu) * sy,
                              (double)L0.z + (double)((qh[2] >> (8 * k)) & 0xFFu) * sz);
            bool out = ch[k] == EMPTY;
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
                const d3 nn = pn[pl];
                const d3 v = mkd((nn.x >= 0.0 ? hi.x : lo.x) - eye.x, (nn.y >= 0.0 ? hi.y : lo.y) - eye.y, (nn.z >= 0.0 ? hi.z : lo.z) - eye.z);
                out = out || dotd(nn, v) < 0.0;
            }
            const double ddx = fmax(fmax(lo.x - eye.x, eye.x - hi.x), 0.0), ddy = fmax(fmax(lo.y - eye.y, eye.y - hi.y), 0.0),
                         ddz = fmax(fmax(lo.z - eye.z, eye.z - hi.z), 0.0);
            const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            out = out || d2 > tmax2;
            hit[k] = !out; hk[k] = d2;
            nh += out ? 0 : 1;
        }
        if (n - 1 + nh <= K && nh <= RT_ENTRY_MAXH) {
            // the node gives way to the children the pyramid reaches (the last entry takes its place: the order is made at the end)
            n--;
            code(i) = code(n); key(i) = key(n);
            fin = (fin & ~(1u << i)) | (((fin >> n) & 1u) << i);
            fin &= ~(1u << n);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (hit[k]) { code(n) = ch[k]; key(n) = hk[k]; n++; }
        } else {
            fin |= 1u << i;
        }
    }
#if RT_ENTRY_PRIMS
    // The leaves of the frontier, primitive by primitive: a ray that starts at a leaf tests every primitive of it without a box test of
    // its own, so a primitive the pyramid cannot reach -- a sphere wholly beyond one of the side planes or farther than the farthest
    // possible hit, a triangle with all three vertices beyond one side plane -- is taken out here (a leaf of two becomes a leaf of one:
    // the leaf code (first << 3) | (count - 1) names any run of the leaf-ordered primitives).
    for (int k = 0; k < n;) {
        const int32_t c = code(k);
        if (c >= 0) { ++k; continue; }
        const uint32_t lc = (uint32_t)~c, first = lc >> 3, cnt = (lc & 7u) + 1u;
        uint32_t keep = 0u;
        for (uint32_t j = 0; j < cnt && j < 8u; ++j) {
            const float4* __restrict__ R = sc.prims + 3 * (size_t)(first + j);
            const float4 r0 = R[0], r1 = R[1], r2 = R[2];
            bool out = false;
            if (__float_as_uint(r2.z) == PRIM_TRIANGLE) {
                const d3 A = mkd((double)r0.x - eye.x, (double)r0.y - eye.y, (double)r0.z - eye.z);
                const d3 B = mkd(A.x + (double)r0.w, A.y + (double)r1.x, A.z + (double)r1.y), C = mkd(A.x + (double)r1.z, A.y + (double)r1.w, A.z + (double)r2.x);
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) out = out || (dotd(pn[pl], A) < 0.0 && dotd(pn[pl], B) < 0.0 && dotd(pn[pl], C) < 0.0);
            } else {
                const double2* __restrict__ S = reinterpret_cast<const double2*>(R);
                const double2 s0 = S[0], s1 = S[1];
                const d3 cc = mkd(s0.x - eye.x, s0.y - eye.y, s1.x - eye.z);
                const double rad = fabs(s1.y) * (1.0 + 1e-6);
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) out = out || dotd(pn[pl], cc) < -rad * sqrt(dotd(pn[pl], pn[pl]));
                const double dc = sqrt(dotd(cc, cc)) - rad;
                out = out || (dc > 0.0 && dc * dc > tmax2);
            }
            if (!out) keep |= 1u << j;
        }
        if (keep == 0u) {                    // nothing of the leaf can be hit: the entry goes (the last one takes its place)
            n--;
            code(k) = code(n); key(k) = key(n);
            continue;
        }
        if (cnt == 2u && keep != 3u) code(k) = (int32_t)~(((first + (keep >> 1)) << 3) | 0u);   // one of two stays
        ++k;
    }
#endif
    // nearest first
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) {
            const double ka = key(a), kb = key(b);
            if (kb < ka) { key(a) = kb; key(b) = ka; const int32_t tc = code(a); code(a) = code(b); code(b) = tc; }
        }
    for (int k = 0; k < K / 4; ++k)
        out[k] = make_int4(n > 4 * k ? code(4 * k) : EMPTY, n > 4 * k + 1 ? code(4 * k + 1) : EMPTY, n > 4 * k + 2 ? code(4 * k + 2) : EMPTY, n > 4 * k + 3 ? code(4 * k + 3) : EMPTY);
#undef code
#undef key
}
__global__ __launch_bounds__(64) void k_tile_entry(DevScene sc, PassParams pp, int4* __restrict__ table) { tile_entry_body(sc, pp, table); }
// k_raygen and k_tile_entry as ONE launch: the first entry_blocks blocks find the entry points (one wave each: a chain of dependent
// node reads per tile, 0.2 ms on its own with most of the chip idle), the others write the primary rays (bound by their stores)
__global__ __launch_bounds__(256) void k_raygen_entry(DevScene sc, PassParams pp, RayQueue q, float4* __restrict__ radiance, uint32_t* __restrict__ ctl,
                                                      int4* __restrict__ table, uint32_t entry_blocks) {
    if (blockIdx.x < entry_blocks) {
        if (threadIdx.x < 64u) tile_entry_body(sc, pp, table);
        return;
    }
    raygen_body(sc, pp, q, radiance, ctl, entry_blocks);
}

// ---------------------------------------------------------------- shading
// materials.rs:197-230 bilinear texture lookup; texel = byte/255.999 (image.rs:45-51).
// With integral coordinates floor == ceil and the weights collapse to the exact texel, so
// the reference's two branches are one formula.  Out-of-range indices (u == 1 or v == 0,
// where the reference would panic) are clamped.
__device__ __forceinline__ f3 texture_albedo(const DevScene& sc, int tex, float u, float v) {
    const int4 T = sc.textures[tex];
    const float W = (float)T.y, H = (float)T.z;
    const float x = W * u, y = H - H * v;
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf, x1 = (int)ceilf(x), y1 = (int)ceilf(y);
    const float fx = x - xf, fy = y - yf;
    const int last = T.y * T.z - 1;
    auto px = [&](int xx, int yy) {
        int idx = yy * T.y + xx;
        idx = idx < 0 ? 0 : (idx > last ? last : idx);
        const uchar4 c = sc.texels[T.x + idx];
        const float k = (float)(1.0 / 255.999);
        return mk((float)c.x * k, (float)c.y * k, (float)c.z * k);
    };
    const f3 c00 = px(x0, y0), c01 = px(x0, y1), c10 = px(x1, y0), c11 = px(x1, y1);
    return c00 * ((1.f - fx) * (1.f - fy)) + c10 * (fx * (1.f - fy)) + c01 * ((1.f - fx) * fy) + c11 * (fx * fy);
}

// materials.rs:105-110
__device__ __forceinline__ f3 refract(f3 dir, f3 n, float ratio) {
    const float cos_theta = fminf(dot(n, -dir), 1.f);
    const f3 perp = (dir + n * cos_theta) * ratio;
    const f3 par = n * -sqrtf(fabsf(1.f - dot(perp, perp)));
    return perp + par;
}
// materials.rs:113-116
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {
    float r0 = (1.f - ref_idx) / (1.f + ref_idx);
    r0 = r0 * r0;
    const float x = 1.f - cosine;
    return r0 + (1.f - r0) * (x * x * x * x * x);
}

// Block-level compaction: every thread of the block must call this (converged).  Threads with
// `want` get `per_lane` consecutive slots; ONE atomic per 1024-thread block.
#ifndef RT_SHADE_THREADS
#define RT_SHADE_THREADS 1024
#endif
#ifndef RT_SHADE_MINWAVES
#define RT_SHADE_MINWAVES 8   // scenes without triangles: <= 64 VGPRs, two 1024-thread blocks per CU, one computes while the other waits at its barriers
#endif
#ifndef RT_SHADE_MINWAVES_TRIS
#define RT_SHADE_MINWAVES_TRIS 4   // with triangles: the f64 hit-point / self-intersection arithmetic needs ~90 VGPRs (64 with spills: 30 % slower)
#endif
#ifndef RT_SHADE_PREFETCH
#define RT_SHADE_PREFETCH 1
#endif
#ifndef RT_SHADE_GRID_MUL
#define RT_SHADE_GRID_MUL 2u
#endif
constexpr int SHADE_THREADS = RT_SHADE_THREADS;
__device__ __forceinline__ uint32_t block_alloc(bool want, uint32_t per_lane, uint32_t* counter, uint32_t* lds /* [17] */) {
    const uint64_t m = __ballot(want);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) lds[wave] = (uint32_t)__popcll(m) * per_lane;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < SHADE_THREADS / 64; ++w) { const uint32_t c = lds[w]; lds[w] = total; total += c; }
        lds[16] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = lds[16] + lds[wave] + rank * per_lane;
    __syncthreads();   // lds is reused by the next call
    return slot;
}

// The same for two kinds of slots at once (k_shade, CLS = 1: a slot of the next queue for a ray that goes on, a slot of the list
// of rays left to the second launch): one pass through the two barriers, two atomics by one thread.  lds: [34]
__device__ __forceinline__ uint32_t block_alloc2(bool want, uint32_t* counter, bool want2, uint32_t* counter2, uint32_t& slot2, uint32_t* lds) {
    const uint64_t m = __ballot(want), m2 = __ballot(want2);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t rank2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
    if (lane == 0) { lds[wave] = (uint32_t)__popcll(m); lds[17 + wave] = (uint32_t)__popcll(m2); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0, total2 = 0;
        for (int w = 0; w < SHADE_THREADS / 64; ++w) {
            const uint32_t c = lds[w], c2 = lds[17 + w];
            lds[w] = total; total += c;
            lds[17 + w] = total2; total2 += c2;
        }
        lds[16] = total ? atomicAdd(counter, total) : 0u;
        lds[33] = total2 ? atomicAdd(counter2, total2) : 0u;
    }
    __syncthreads();
    const uint32_t slot = lds[16] + lds[wave] + rank;
    slot2 = lds[33] + lds[17 + wave] + rank2;
    __syncthreads();   // lds is reused by the next call
    return slot;
}

// The reference's distance to primitive / plane `c` along the f64 ray, or +inf where it reports no hit
// (intersections.rs:42-67, 84-93, 153-168), operation by operation.
__device__ __forceinline__ double ref_t64(const DevScene& sc, uint32_t c, d3 o64, d3 d64) {
#pragma clang fp contract(off)
    const double none = __builtin_inf();
    if (c & HIT_PLANE_BIT) {
        const double* __restrict__ P64 = sc.planes64 + 6 * (size_t)(c & ~HIT_PLANE_BIT);
        const d3 c64 = mkd(P64[0], P64[1], P64[2]), n64 = mkd(P64[3], P64[4], P64[5]);
        const double dot_nd = dot_ref(n64, d64);
        if (fabs(dot_nd) < 1e-6) return none;
        const double tp = dot_ref(sub_ref(c64, o64), n64) / dot_nd;
        return tp < 1e-5 ? none : tp;
    }
    const float4* __restrict__ rec = sc.smeta + SMETA_WORDS * (size_t)c;
    const double2* __restrict__ r64 = reinterpret_cast<const double2*>(rec + 1);
    if (!(__float_as_uint(rec[0].w) & SMETA_SPHERE)) {
        const double2 w0 = r64[0], w1 = r64[1], w2 = r64[2], w3 = r64[3], w4 = r64[4];
        double t0, a, b;
        return tri_hit_ref(o64, d64, mkd(w0.x, w0.y, w1.x), mkd(w1.y, w2.x, w2.y), mkd(w3.x, w3.y, w4.x), t0, a, b) ? t0 : none;
    }
    const double2 c0 = r64[0], c1 = r64[1];
    const d3 oc = sub_ref(o64, mkd(c0.x, c0.y, c1.x));
    const double qa = dot_ref(d64, d64), qb = 2.0 * dot_ref(d64, oc), qc = dot_ref(oc, oc) - c1.y * c1.y;
    double disc = qb * qb - 4.0 * qa * qc;
    if (!(disc >= 0.0)) return none;
    disc = sqrt(disc);
    const double t1 = (-qb - disc) / (2.0 * qa), t2 = (-qb + disc) / (2.0 * qa);
    return t1 > 1e-5 ? t1 : (t2 > 1e-5 ? t2 : none);
}
// Hit and runner-up of k_closest_hit (two candidates at the same distance as far as f32 can tell): the one the reference
// keeps -- the smaller f64 t, the first in scan order on equality (scene.rs:35-50, intersections.rs:148-186).  near_tri: the
// loser, if it is a triangle the f64 test accepts (a coplanar neighbour the next ray may meet at t ~ 1e-16, see leaves_src).
__device__ __forceinline__ uint32_t resolve_tie(const DevScene& sc, uint32_t code, uint32_t aux, d3 o64, d3 d64, uint32_t& near_tri) {
    // aux: the runner-up (a hit code), or AUX_NEAR_BIT | triangle: a candidate at t ~ 0 that the f32 traversal could not rank
    // (k_closest_hit's leaf step); code may then be HIT_MISS.  A candidate the f64 test rejects never wins.
    const uint32_t code2 = strip_hit((aux >> 30) == 1u ? (aux & SRC_PRIM_MASK) : aux);
    const double none = __builtin_inf();
    const double ta = code == HIT_MISS ? none : ref_t64(sc, code, o64, d64), tb = ref_t64(sc, code2, o64, d64);
    if (code == HIT_MISS) return tb < none ? code2 : HIT_MISS;
    const bool second = tb < ta || (tb == ta && tb < none && gid_of(sc, code2) < gid_of(sc, code));
    const uint32_t lose = second ? code : code2;
    if (!(lose & HIT_PLANE_BIT) && (second ? ta : tb) < none && !(__float_as_uint(sc.smeta[SMETA_WORDS * (size_t)lose].w) & SMETA_SPHERE)) near_tri = lose;
    return second ? code2 : code;
}

// One thread per ray of iteration `it`.  last != 0: this is the last allowed bounce
// (current_depth == max_depth-1) or a shadow-ray generation (trace(scene, 0, 1)): the
// scattered ray is worthless (it would return black at depth == max_depth) and only the
// light-sampling branch of raytrace.rs:74-87 can continue the path.  last == 2: the caller runs no
// further generation (rt_render_params.max_shadow_generations), so nothing is emitted.
// CLS (scenes with triangles, the bounces between the first and the last): the rays of an iteration shaded in TWO launches.  A hit
// on a triangle costs two evaluations of the reference's f64 determinants (hit point, re-hit test), a hit on a sphere the f64
// quadratic, a miss next to nothing, and a wave pays for every kind one of its 64 lanes holds: after the first bounce one ray in
// eight of C5 hits a triangle, so nearly every wave ran the triangle code for a handful of lanes.  CLS = 1 shades everything but the
// hits on triangles and the hits with a runner-up to decide (resolve_tie), whose queue slots it appends to a list (one more count
// in the block's slot claim, no further barrier); CLS = 2 runs over that list with waves that hold nothing else.  The rays a path
// emits, its random numbers and its radiance word do not depend on the launch that shades it: renders are bit-identical to CLS = 0.
template <bool TRIS, int PRIMARY, int CLS>   // PRIMARY: iteration 0; 1: the queue holds 16-byte primary rays (k_raygen), 2: no queue, the ray is formed here (see k_closest_hit, GEN)
__global__ __launch_bounds__(SHADE_THREADS, TRIS ? RT_SHADE_MINWAVES_TRIS : RT_SHADE_MINWAVES) void k_shade(DevScene sc, RayQueue qi, const float2* __restrict__ hits, RayQueue qo,
                                               float4* __restrict__ radiance, uint32_t* __restrict__ ctl, int it, int last,
                                               uint32_t capacity, PassParams pp, uint32_t* __restrict__ heavy) {
    constexpr bool GEN = PRIMARY == 2;
    static_assert(CLS == 0 || (TRIS && PRIMARY == 0), "the two-launch form exists for the later bounces of scenes with triangles");
    __shared__ uint32_t lds_alloc[34];
    uint32_t parity = 0;
    // CLS = 2: the rays are those of the list CLS = 1 made (two counters taking turns: this launch clears the one the next iteration's list will count in)
    const uint32_t n = CLS == 2 ? min(ctl[CTL_HEAVY + (it & 1)], capacity) : min(ctl[CTL_QCOUNT + it], capacity);
    if (CLS == 2 && blockIdx.x == 0 && threadIdx.x == 0) ctl[CTL_HEAVY + ((it + 1) & 1)] = 0u;
    const uint32_t n_pad = (n + (SHADE_THREADS - 1)) & ~(uint32_t)(SHADE_THREADS - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    (void)parity;
    // The ray and hit words of a round are loaded one round ahead (RT_SHADE_PREFETCH), right before the round's slot claim:
    // a round is a chain of dependent waits (queue words, shading record, the claim's atomic between two barriers) that all
    // 16 waves of the block walk in step, and this takes the first link off it.
    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA, nC = nA;
    float2 nH = make_float2(0.f, 0.f);
    uint32_t nMy = 0;   // CLS = 2: the queue slot of the next round's ray (its words are gathered at the top of the round)
    auto load_round = [&](uint32_t j) {
        if (CLS == 2) { nMy = heavy[j]; return; }
        if (!GEN) nA = ld_stream(&qi.a[j]);
        nH = ld_stream(&hits[j]);
        if (!PRIMARY) { nB = ld_stream(&qi.b[j]); nC = ld_stream(&qi.c[j]); }
    };
    // ... and the first word of the shading record of the next round's hit (normal / material / kind) right behind the claim, when
    // the hit word has arrived: the material gather, which needs it, then starts at the top of the round beside the record's other
    // words instead of one memory round trip later.
    float4 nS = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t nS_code = HIT_MISS;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (RT_SHADE_PREFETCH && i < n) load_round(i);
    for (; i < n_pad; i += stride) {   // block-uniform trip count
        bool emit = false;       // continue with a scattered ray
        bool fire = false;       // spawn shadow rays
        bool on_tri = false;     // the hit is on a triangle
        bool has_p64 = false;    // ... or on a plane (scenes with triangles): p64 is the f64 hit point, the rays that leave carry it
        f3 p = mk(0, 0, 0), ndir = mk(0, 0, 0), thr = mk(0, 0, 0);
        d3 p64 = mkd(0, 0, 0);
        uint32_t rng = 0, pid = 0, code = HIT_INVALID, dh = 0;
        uint32_t near_tri = HIT_MISS;   // a triangle coplanar with the hit that lost the f64 comparison (see the tie resolution below)
        f3 rad = mk(0, 0, 0);    // what the path's end contributes (miss: background, emitter: its colour; absorbed or out of bounces: black)
        bool has_rad = false;
        bool dir_exact = false;  // the scattered direction is the reference's f64 one: ndir holds its upper bits, d_lo the lower ones (RayQueue::f)
        uint32_t d_lo_x = 0, d_lo_y = 0, d_lo_z = 0;
        if (!RT_SHADE_PREFETCH && i < n) load_round(i);
        float4 A = nA, B = nB, C = nC;
        float2 H = nH;
        uint32_t my = i;                                  // the queue slot of the ray this thread shades
        if (CLS == 2 && i < n) {
            my = nMy;
            A = ld_stream(&qi.a[my]); B = ld_stream(&qi.b[my]); C = ld_stream(&qi.c[my]); H = ld_stream(&hits[my]);
        }
        if (!(i < n)) H = make_float2(__uint_as_float(HIT_MISS), __uint_as_float(HIT_INVALID));
        bool defer = false;      // CLS = 1: this ray is left to the second launch
        if (__float_as_uint(H.y) != HIT_INVALID) {
            f3 o, d;
            uint32_t src = SRC_NONE;
            uint4 E0g = make_uint4(0u, 0u, 0u, 0u);   // GEN: the low bits of the f64 direction (what k_raygen writes to queue word e)
            if (GEN) {
                // the ray k_raygen would have written for this slot (the same function, the same bits)
                double gx = 0.0, gy = 0.0, gz = 0.0;
                (void)primary_ray(sc, pp, my, rng, gx, gy, gz, RT_PRIMARY_REF64 && TRIS);
                o = mk(sc.eye[0], sc.eye[1], sc.eye[2]);
                d = mk((float)gx, (float)gy, (float)gz);
                if (RT_PRIMARY_REF64 && TRIS) { split64(gx, d.x, E0g.x); split64(gy, d.y, E0g.y); split64(gz, d.z, E0g.z); }
                pid = my;
                thr = mk(1.f, 1.f, 1.f);
            } else if (PRIMARY) {   // primary ray: (d, rng) only, see k_raygen
                o = mk(sc.eye[0], sc.eye[1], sc.eye[2]); d = mk(A.x, A.y, A.z);
                pid = my;
                thr = mk(1.f, 1.f, 1.f);
                rng = __float_as_uint(A.w);
            } else {
                o = mk(A.x, A.y, A.z); d = mk(A.w, B.x, B.y);
                src = __float_as_uint(B.z);
                pid = __float_as_uint(B.w);
                thr = mk(C.x, C.y, C.z);
                rng = __float_as_uint(C.w);
            }
            code = strip_hit(__float_as_uint(H.y));
            if (CLS == 1) {
                const uint32_t raw = __float_as_uint(H.y), aux = __float_as_uint(H.x);
                defer = (aux != HIT_MISS && strip_hit(aux) != code) || (code < SRC_SPECIAL && !(code & HIT_PLANE_BIT) && !(raw & HIT_SPHERE_BIT));
                if (defer) code = HIT_INVALID;   // (nothing below touches it: no radiance word, no ray)
            }
            float u = 0.f, v = 0.f;
            // The ray as the f64 arithmetic below sees it: the exact eye for a primary ray, the carried f64 origin for a ray
            // that left a triangle, a plane or a sphere (SRC_E_BIT), a dithered f32 origin otherwise; the carried f64
            // direction after a specular bounce (SRC_D_BIT), a dithered f32 direction otherwise.
            d3 d64 = mkd(0, 0, 0);
            uint32_t sw = 0;
            auto ray64 = [&](d3& o64, bool flat) {   // flat: the hit is on a triangle or a plane
                dh = pcg_hash(rng ^ 0x68E31DA4u);          // dither bits: a stream of their own, rng is not advanced
                // (A hit on a sphere reads the carried words too.  Leaving them out there -- RT_SPHERE_E = 0: its hit point and normal, a square
                // root and a division, mix every bit of the ray anew, 16 bytes less to read for nine rays in ten of C5, -1.2 % -- passed 3,000
                // soak scenes and failed two of the next 1,400 by 5 %: a ray INSIDE a glass ball starts on the sphere it hits, where
                // |o - c|^2 - r^2 is zero to an ulp of f64 in the reference (tools/soak_objects.py seed 7052), and a ray that comes off a flat
                // glass face onto a mirror ball and back carries the face's exact coordinates with it (seed 8309).)
                const bool carried = !PRIMARY && src < SRC_SPECIAL && (src & SRC_E_BIT) && (flat || RT_SPHERE_E);
                uint4 E = make_uint4(0u, 0u, 0u, 0u);
                if (carried) E = ld_stream(&qi.e[my]);
                if (PRIMARY) o64 = mkd(sc.eye64[0], sc.eye64[1], sc.eye64[2]);
                else if (carried) o64 = mkd(join64(o.x, E.x), join64(o.y, E.y), join64(o.z, E.z));
                else o64 = dither64(o, dh);
                if (RT_PRIMARY_REF64 && PRIMARY && (flat || RT_PRIMARY_REF64 > 1)) {
                    // in the reference a primary direction is a unit vector to 1e-16 with the bits of its own f64
                    // sequence (primary_ray), and at exactly representable coordinates what the ray does on the triangle it
                    // reaches depends on them.  (Not read for a hit on a sphere: its hit point and normal mix every bit
                    // anew, as for the exact specular directions; RT_PRIMARY_REF64 = 2 reads it there too.)
                    const uint4 E0 = GEN ? E0g : ld_stream(&qi.e[my]);
                    d64 = mkd(join64(d.x, E0.x), join64(d.y, E0.y), join64(d.z, E0.z));
                } else {
                    const bool exact_d = RT_DIR_REF64 && carried && (src & SRC_D_BIT), self_ray = carried && (src >> 30) == 1u;
                    uint4 F = make_uint4(0u, 0u, HIT_MISS, 0u);
                    if (exact_d || self_ray) F = ld_stream(&qi.f[my]);
                    if (exact_d) d64 = mkd(join64(d.x, E.w), join64(d.y, F.x), join64(d.z, F.y));
                    else d64 = dither64(d, dh);
                    if (self_ray) near_tri = F.z;   // the coplanar partner of the triangle this ray re-hits (HIT_MISS: none), see leaves_src
                }
            };
            if (TRIS) {
                // Two candidates at the same distance as far as f32 can tell (k_closest_hit's runner-up: a box face in the
                // ground plane, a box on a box): the reference scans every object in f64 and keeps the smallest t, the
                // first in scan order on equality (scene.rs:35-50, intersections.rs:148-186), so which of two coplanar
                // surfaces a ray meets is decided by the roundings of two f64 distances, ray by ray.  Both distances
                // are formed here with the reference's sequences; the loser is remembered: if it is a triangle, the ray
                // that leaves may meet it at t ~ 1e-16 (leaves_src).  (A rare branch of its own, ahead of the gathers of
                // the hit's records, so that they do not wait for the queue word of the f64 ray.)
                // The same for a candidate at t ~ 0 (AUX_NEAR_BIT): the hit may then have been a miss so far.
                const uint32_t code2 = __float_as_uint(H.x);
                if (RT_TIE_SHADE && CLS != 1 && code != HIT_INVALID && code2 != HIT_MISS && strip_hit(code2) != code) {   // (CLS = 1 has left these to the second launch)
                    d3 o64t;
                    ray64(o64t, true);
                    code = resolve_tie(sc, code, code2, o64t, d64, near_tri);
                }
            }
            if (code == HIT_INVALID) {
                // padding slot: not a ray
            } else if (code == HIT_MISS) {
                rad = thr * mk(sc.bg[0], sc.bg[1], sc.bg[2]); has_rad = true;   // raytrace.rs:95
            } else if (last && sc.no_emitters) {
                // the path's last ray has hit something in a scene without emitters: whatever the material does, the scattered ray
                // returns black (raytrace.rs:52-54) and there is no light to sample -- nothing to gather, nothing to compute
                // (k_closest_hit stopped at the first hit it found for this ray: `code` need not be the closest one)
            } else {
                float t = H.x;
                f3 nrm;
                int mat;
                // (the material words are asked for as soon as the material is known, ahead of the f64 hit-point arithmetic: behind it,
                // where they are used, their round trip was one more link of the round's chain of waits)
                float4 M0, M1;
                if (code & HIT_PLANE_BIT) {
                    const uint32_t pl = code & ~HIT_PLANE_BIT;
                    const float4 P1 = sc.planes[2 * pl + 1];
                    nrm = mk(P1.x, P1.y, P1.z);                                            // intersections.rs:96 (as given)
                    mat = __float_as_int(P1.w);
                    if (RT_SHADE_EARLY_MAT) { M0 = sc.materials[2 * mat]; M1 = sc.materials[2 * mat + 1]; }
                    if (TRIS) {
                        // intersections.rs:84-95 in f64 on the reference's operands: the point a ray leaves a plane from is
                        // ON it to an ulp of f64 (e.g. y = -0.6 exactly for 98.7 % of the rays, one ulp off for the rest), and
                        // the re-hit rate of the triangle such a ray lands on depends on that (see RayQueue::e)
#pragma clang fp contract(off)
                        const double* __restrict__ P64 = sc.planes64 + 6 * (size_t)pl;
                        const d3 c64 = mkd(P64[0], P64[1], P64[2]), n64 = mkd(P64[3], P64[4], P64[5]);
                        d3 o64;
                        ray64(o64, true);
                        const double dot_nd = n64.x * d64.x + n64.y * d64.y + n64.z * d64.z;
                        const d3 co = sub_ref(c64, o64);
                        const double t64 = (co.x * n64.x + co.y * n64.y + co.z * n64.z) / dot_nd;
                        p64 = at_ref(o64, d64, t64);
                        p = mk((float)p64.x, (float)p64.y, (float)p64.z);
                        has_p64 = true;
                    } else {
                        p = at_ref(o, d, t);                                               // raytrace.rs:40-42
                    }
                } else {
                    // one 96-byte shading record per primitive
                    const float4* __restrict__ rec = sc.smeta + SMETA_WORDS * (size_t)code;
                    const float4 S = (RT_SHADE_PREFETCH_S && nS_code == code) ? nS : rec[0];
                    sw = __float_as_uint(S.w);
                    mat = (int)(sw & ~SMETA_FLAGS);
                    if (RT_SHADE_EARLY_MAT) { M0 = sc.materials[2 * mat]; M1 = sc.materials[2 * mat + 1]; }
                    if (TRIS && CLS != 1 && !(sw & SMETA_SPHERE)) {   // (CLS = 1: every primitive it shades is a sphere)
                        nrm = mk(S.x, S.y, S.z);                                           // intersections.rs:169, see smeta
                        // t, the hit point and the barycentrics with the reference's own f64 sequence on its own f64
                        // operands (see tri_hit_ref).  The ray enters it as f64: the exact eye for a primary ray, the
                        // carried f64 origin for a ray that re-hits its triangle, dithered f32 values otherwise.
                        const double2* __restrict__ r64 = reinterpret_cast<const double2*>(rec + 1);
                        const double2 w0 = r64[0], w1 = r64[1], w2 = r64[2], w3 = r64[3], w4 = r64[4];   // (cached: streamed past the caches like the queues, C5 +3.5 %, C4 +9 % step time -- the re-hit test reads the record again)
                        d3 o64;
                        ray64(o64, true);
                        double t0, a, b;
                        tri_hit_ref(o64, d64, mkd(w0.x, w0.y, w1.x), mkd(w1.y, w2.x, w2.y), mkd(w3.x, w3.y, w4.x), t0, a, b);
                        p64 = at_ref(o64, d64, t0);                                        // intersections.rs:172
                        p = mk((float)p64.x, (float)p64.y, (float)p64.z);
                        on_tri = true;
                        has_p64 = true;
                        if (sw & SMETA_HAS_UV) {                                           // intersections.rs:177-183
                            const float* uv = sc.uvs + 6 * (size_t)sc.meta[code].w;
                            const double c = 1.0 - a - b;
                            u = (float)(a * (double)uv[2] + b * (double)uv[4] + c * (double)uv[0]);
                            v = (float)(a * (double)uv[3] + b * (double)uv[5] + c * (double)uv[1]);
                        }
                    } else {
                        const float4 r0 = rec[1], r1 = rec[2];
                        if (TRIS) {
                            // intersections.rs:42-69 in f64, operation by operation: in scenes with triangles every hit point
                            // is the reference's own f64 function of the ray, because the rays that leave it carry it as
                            // their origin and the last bits of an origin decide what happens on the triangle it reaches
#pragma clang fp contract(off)
                            const double cx = __hiloint2double(__float_as_int(r0.y), __float_as_int(r0.x));
                            const double cy = __hiloint2double(__float_as_int(r0.w), __float_as_int(r0.z));
                            const double cz = __hiloint2double(__float_as_int(r1.y), __float_as_int(r1.x));
                            const double rad = __hiloint2double(__float_as_int(r1.w), __float_as_int(r1.z));
                            d3 o64;
                            ray64(o64, false);
                            const d3 oc = sub_ref(o64, mkd(cx, cy, cz));
                            const double qa = d64.x * d64.x + d64.y * d64.y + d64.z * d64.z;
                            const double qb = 2.0 * (d64.x * oc.x + d64.y * oc.y + d64.z * oc.z);
                            const double qc = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - rad * rad;
                            double disc = qb * qb - 4.0 * qa * qc;
                            disc = sqrt(disc > 0.0 ? disc : 0.0);
                            // intersections.rs:54-63: the nearer root if it is above 1e-5, else the farther one (the second
                            // division only then)
                            double t64 = (-qb - disc) / (2.0 * qa);
                            if (!(t64 > 1e-5)) t64 = (-qb + disc) / (2.0 * qa);
                            p64 = at_ref(o64, d64, t64);
                            p = mk((float)p64.x, (float)p64.y, (float)p64.z);
                            const double ir = 1.0 / rad;    // (the normal is rounded to f32: one reciprocal instead of the reference's three divisions)
                            nrm = mk((float)((p64.x - cx) * ir), (float)((p64.y - cy) * ir), (float)((p64.z - cz) * ir));
                            has_p64 = true;
                        } else {
                            p = at_ref(o, d, t);                                           // intersections.rs:68
                            nrm = sphere_normal(r0, r1, o, d, t);                          // intersections.rs:69
                        }
                    }
                }
                if (!RT_SHADE_EARLY_MAT) { M0 = sc.materials[2 * mat]; M1 = sc.materials[2 * mat + 1]; }
                const int mtype = __float_as_int(M1.x);
                // intersection_info.normal as the reference has it, in f64 (only the specular materials ask for it; the
                // records are read again rather than kept in registers)
                auto normal64 = [&]() -> d3 {
#pragma clang fp contract(off)
                    if (code & HIT_PLANE_BIT) {                                            // intersections.rs:96 (as given)
                        const double* __restrict__ P64 = sc.planes64 + 6 * (size_t)(code & ~HIT_PLANE_BIT);
                        return mkd(P64[3], P64[4], P64[5]);
                    }
                    const double2* __restrict__ r64 = reinterpret_cast<const double2*>(sc.smeta + SMETA_WORDS * (size_t)code + 1);
                    if (CLS != 1 && !(sw & SMETA_SPHERE)) {                                // intersections.rs:169
                        const double2 w1 = r64[1], w2 = r64[2], w3 = r64[3], w4 = r64[4];
                        return normalized_ref(cross_ref(mkd(w1.y, w2.x, w2.y), mkd(w3.x, w3.y, w4.x)));
                    }
                    const double2 c0 = r64[0], c1 = r64[1];                                // intersections.rs:69
                    return mkd((p64.x - c0.x) / c1.y, (p64.y - c0.y) / c1.y, (p64.z - c1.x) / c1.y);
                };
                // Flat surfaces only: there a specular bounce keeps the arriving ray's bits (the normal is a constant, often
                // with exact zeros), and the triangles it goes on to are met with directions tied to the one before.  A sphere's
                // normal (p - c) / r is rounded anew at every hit and mixes into every component, so nothing is carried over;
                // with RT_DIR_REF64 = 2 spheres take the f64 path too (C5: -8.4 %).
                const bool exact_bounce = TRIS && RT_DIR_REF64 && (RT_DIR_REF64 > 1 || (code & HIT_PLANE_BIT) || !(sw & SMETA_SPHERE));
                auto set_exact = [&](d3 nd) {   // Ray::new (raytrace.rs:32-37) has normalised nd
                    split64(nd.x, ndir.x, d_lo_x); split64(nd.y, ndir.y, d_lo_y); split64(nd.z, ndir.z, d_lo_z);
                    dir_exact = true;
                };
                f3 albedo = mk(M0.x, M0.y, M0.z);
                bool scattered = false;
                if (mtype == 4) {                                   // Emissive, materials.rs:80-88
                    rad = thr * albedo; has_rad = true;             // raytrace.rs:90
                } else if (mtype == 0) {                            // Lambertian, materials.rs:52-66
                    f3 dir = nrm + random_unit_vector(rng);
                    if (near_zero(dir)) dir = nrm;
                    ndir = normalized(dir);
                    scattered = true;
                } else if (mtype == 1) {                            // Metal, materials.rs:168-183
                    if (exact_bounce && M0.w == 0.f) {
                        // a mirror: the scattered direction is a function of the ray alone, so it is the reference's own f64
                        // sequence on the f64 ray and travels on exactly (see RayQueue::f); the fuzz term is +-0
                        const d3 n64 = normal64();
                        const d3 nd = normalized_ref(reflect_ref(normalized_ref(d64), n64));
                        scattered = dot_ref(nd, n64) > 0.0;
                        set_exact(nd);
                    } else {
                        const f3 refl = reflect(normalized(d), nrm);
                        ndir = normalized(refl + random_in_unit_sphere(rng) * M0.w);
                        scattered = dot(ndir, nrm) > 0.f;           // else None: the path ends black (raytrace.rs:92)
                    }
                } else if (mtype == 2) {                            // Dieletrics, materials.rs:119-153
                    if (exact_bounce) {
                        // the same in f64 on the f64 ray, normal and refraction index: the direction that leaves is a function
                        // of the ray alone, and in the reference its last bits are tied to those of the ray that arrived (a
                        // reflection off an axis-aligned face only flips a sign).  What the ray then does on the triangle it
                        // reaches depends on that: with f32 directions + random low bits 124 of 600,000 paths through an
                        // axis-aligned glass box left it at the third interface where the reference has none.
#pragma clang fp contract(off)
                        const d3 n64 = normal64();
                        const double ior = sc.mat_param64[mat];
                        const bool front = dot_ref(n64, d64) <= 0.0;
                        const double ratio = front ? 1.0 / ior : ior;
                        const d3 nn = front ? n64 : neg_ref(n64);
                        const double cos_theta = fmin(dot_ref(neg_ref(d64), nn), 1.0);
                        const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                        bool refl = ratio * sin_theta > 1.0;                  // cannot_refract
                        if (!refl) {                                          // the RNG draw is skipped otherwise (materials.rs:143-144)
                            double r0 = (1.0 - ratio) / (1.0 + ratio);
                            r0 = r0 * r0;
                            const double x = 1.0 - cos_theta, x2 = x * x;
                            refl = r0 + (1.0 - r0) * (x2 * x2 * x) > (double)rng_unit(rng);
                        }
                        set_exact(normalized_ref(refl ? reflect_ref(d64, nn) : refract_ref(d64, nn, ratio)));
                    } else {
                        const bool front = dot(nrm, d) <= 0.f;
                        const float ratio = front ? 1.0f / M0.w : M0.w;
                        const f3 nn = front ? nrm : -nrm;
                        const float cos_theta = fminf(dot(-d, nn), 1.f);
                        const float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
                        const bool cannot_refract = ratio * sin_theta > 1.f;
                        f3 dir = refract(d, nn, ratio);
                        // the RNG draw is skipped when cannot_refract (short circuit, materials.rs:143-144)
                        if (cannot_refract || reflectance(cos_theta, ratio) > rng_unit(rng)) dir = reflect(d, nn);
                        ndir = normalized(dir);
                    }
                    scattered = true;
                } else {                                            // Texture, materials.rs:233-253
                    albedo = texture_albedo(sc, __float_as_int(M1.y), u, v);
                    if (dot(d, nrm) > 0.f) {
                        rad = thr * albedo; has_rad = true;              // back face emits the texel
                    } else {
                        f3 dir = nrm + random_unit_vector(rng);
                        if (near_zero(dir) || dot(nrm, dir) <= 0.f) dir = nrm;
                        ndir = normalized(dir);
                        scattered = true;
                    }
                }
                if (scattered) {
                    if (!last) {
                        thr = thr * albedo;                         // raytrace.rs:88, light_color is black before the last bounce
                        emit = true;
                    } else if (sc.n_lights) {
                        // raytrace.rs:64-87
                        const float prob = (mtype == 2) ? 0.05f : 0.1f;
                        const float L = (float)sc.n_lights;
                        if (rng_unit(rng) > 1.f - L * prob) {
                            fire = true;
                            thr = thr * albedo * (1.0f / L);        // albedo * target / lights_len
                        }
                    }
                }
            }
        }
        if (code != HIT_INVALID) {
            if (sc.n_lights == 0) {
                // without lights a path contributes exactly once, when it ends: ONE plain 16-byte store per path, black included
                // (k_raygen then need not clear the word: 16 bytes per path less)
                if (!emit) st_rad(&radiance[pid], make_float4(rad.x, rad.y, rad.z, 0.f));
            } else if (has_rad) {
                // shadow rays of one path end in different lanes / launches
                float* r = reinterpret_cast<float*>(radiance + pid);
                atomicAdd(r + 0, rad.x);
                atomicAdd(r + 1, rad.y);
                atomicAdd(r + 2, rad.z);
            }
        }
        // A ray that leaves a triangle: does it re-hit that triangle at t ~ 0?  (intersections.rs:166 has no epsilon;
        // see tri_hit_ref.)  The reference's f64 test on the f64 hit point; the verdict travels in the src word.
        // rng_out: the rng word the emitted ray carries.  The verdict is formed on the very f64 ray the next k_shade will form
        // from the queue words (same origin bits, same dither bits of the direction: the next launch derives them from
        // that word), as in the reference, where the verdict IS the next intersection.
        // exact: dir_out is ndir after a specular bounce, whose f64 value travels with the ray (SRC_D_BIT).
        // Besides the triangle the ray starts on there may be a coplanar neighbour (near_tri: the face of another box, or
        // a box face in the plane that was hit): the reference meets whichever its f64 test accepts at the smaller t, the
        // first in scan order on equality.  A ray that leaves a plane names the neighbour as the primitive to skip (the
        // plane itself needs no name: t ~ 0 fails its t >= 1e-5).
        auto leaves_src = [&](f3 dir_out, bool exact, uint32_t rng_out, uint32_t& src_out, uint32_t& other_out) -> bool {
            other_out = HIT_MISS;
            const uint32_t flags = has_p64 ? (SRC_E_BIT | (exact ? SRC_D_BIT : 0u)) : 0u;
            src_out = code | flags;
            if (!TRIS || CLS == 1 || (!on_tri && near_tri == HIT_MISS)) return false;   // (CLS = 1 shades no hit on a triangle and no hit with a coplanar partner)
            if (!on_tri) src_out = near_tri | flags;
            uint32_t dh = pcg_hash(rng_out ^ 0x68E31DA4u);
            const d3 nd64 = exact ? mkd(join64(dir_out.x, d_lo_x), join64(dir_out.y, d_lo_y), join64(dir_out.z, d_lo_z)) : dither64(dir_out, dh);
            auto test = [&](uint32_t c, double& t1) -> bool {
                const double2* __restrict__ r64 = reinterpret_cast<const double2*>(sc.smeta + SMETA_WORDS * (size_t)c + 1);
                const double2 w0 = r64[0], w1 = r64[1], w2 = r64[2], w3 = r64[3], w4 = r64[4];   // (re-read: cheaper than 18 live registers across the scatter code; kept in LDS instead: C4 -0.5 %, C5 unchanged)
                double a1, b1;
                return tri_hit_ref(p64, nd64, mkd(w0.x, w0.y, w1.x), mkd(w1.y, w2.x, w2.y), mkd(w3.x, w3.y, w4.x), t1, a1, b1);
            };
            uint32_t who = HIT_MISS;
            double t_self = 0.0, t_near = 0.0;
            if (on_tri && test(code, t_self)) who = code;
            if (near_tri != HIT_MISS && test(near_tri, t_near)) {       // rare
                if (who == HIT_MISS || t_near < t_self || (t_near == t_self && gid_of(sc, near_tri) < gid_of(sc, code))) who = near_tri;
            }
            if (who == HIT_MISS) return false;
            src_out = who | SRC_SELF_BIT | flags;
            // the ray that leaves `who` may in turn meet the other one of the pair: it travels on as that ray's neighbour
            // (queue word f), so that two coincident faces hand a path back and forth as they do in the reference
            other_out = who == code ? near_tri : (on_tri ? code : HIT_MISS);
            return true;
        };
        uint4 p_lo = make_uint4(0u, 0u, 0u, 0u);
        if (TRIS && has_p64) {   // the f64 hit point as (f32 origin, low bits): see RayQueue::e
            split64(p64.x, p.x, p_lo.x); split64(p64.y, p.y, p_lo.y); split64(p64.z, p.z, p_lo.z);
        }
        dir_exact = dir_exact && has_p64 && emit;
        uint32_t src_next = code, other_next = HIT_MISS;
        if (!last && emit) (void)leaves_src(ndir, dir_exact, rng, src_next, other_next);   // before the claim: its loads and f64 work overlap nothing behind the barriers
        if (RT_SHADE_PREFETCH) {
            const uint64_t j = (uint64_t)i + stride;
            if (j < n) load_round((uint32_t)j);
        }
        if (!last) {
            // (ordering the block's slots by the octant of the new direction was measured: no gain)
            uint32_t slot;
            if (CLS == 1) {
                uint32_t hslot = 0;
                slot = block_alloc2(emit, &ctl[CTL_QCOUNT + it + 1], defer, &ctl[CTL_HEAVY + (it & 1)], hslot, lds_alloc);
                if (defer) heavy[hslot] = my;
            } else {
                slot = block_alloc(emit, 1u, &ctl[CTL_QCOUNT + it + 1], lds_alloc);
            }
            if (RT_SHADE_PREFETCH && RT_SHADE_PREFETCH_S && CLS != 2) {
                const uint32_t c = strip_hit(__float_as_uint(nH.y));
                nS_code = ((uint64_t)i + stride < n && c < SRC_SPECIAL && !(c & HIT_PLANE_BIT)) ? c : HIT_MISS;
                if (nS_code != HIT_MISS) nS = sc.smeta[SMETA_WORDS * (size_t)c];
            }
            if (emit) {
                if (slot < capacity) {
                    const uint32_t src_out = src_next;
                    st_stream(&qo.a[slot], make_float4(p.x, p.y, p.z, ndir.x));
                    st_stream(&qo.b[slot], make_float4(ndir.y, ndir.z, __uint_as_float(src_out), __uint_as_float(pid)));
                    st_stream(&qo.c[slot], make_float4(thr.x, thr.y, thr.z, __uint_as_float(rng)));
                    if (TRIS && has_p64) st_stream(&qo.e[slot], make_uint4(p_lo.x, p_lo.y, p_lo.z, d_lo_x));
                    if (TRIS && (dir_exact || (src_out >> 30) == 1u)) st_stream(&qo.f[slot], make_uint4(d_lo_y, d_lo_z, other_next, 0u));
                } else {
                    atomicAdd(&ctl[CTL_OVERFLOW], 1u);
                }
            }
        } else if (sc.n_lights && last == 1) {
            const uint32_t L = sc.n_lights;
            const uint32_t slot0 = block_alloc(fire, L, &ctl[CTL_QCOUNT + it + 1], lds_alloc);
            if (fire) {
                for (uint32_t l = 0; l < L; ++l) {
                    const uint32_t slot = slot0 + l;
                    if (slot >= capacity) { atomicAdd(&ctl[CTL_OVERFLOW], 1u); continue; }
                    const float4 Lp = sc.lights[l];
                    const f3 sd = normalized(mk(Lp.x, Lp.y, Lp.z) - p);      // raytrace.rs:79-82
                    uint32_t src_out, other_out;
                    const uint32_t rng_l = pcg_hash(rng + 0x9E3779B9u * (l + 1u));
                    (void)leaves_src(sd, false, rng_l, src_out, other_out);
                    qo.a[slot] = make_float4(p.x, p.y, p.z, sd.x);
                    qo.b[slot] = make_float4(sd.y, sd.z, __uint_as_float(src_out), __uint_as_float(pid));
                    qo.c[slot] = make_float4(thr.x, thr.y, thr.z, __uint_as_float(rng_l));
                    if (TRIS && has_p64) qo.e[slot] = p_lo;
                    if (TRIS && (src_out >> 30) == 1u) qo.f[slot] = make_uint4(0u, 0u, other_out, 0u);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- accumulate / resolve
// raytrace.rs:134-143: sum of the sample colours of a pixel, in a fixed order.
__global__ __launch_bounds__(256) void k_accumulate(DevScene sc, PassParams pp, const float4* __restrict__ radiance,
                                                    float* __restrict__ accum) {
    const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
    if (pl >= pp.npix_pad) return;
    uint32_t x, row;
    if (!slot_to_pixel(sc, pp, pl, x, row)) return;
    float r = 0.f, g = 0.f, b = 0.f;
    for (uint32_t s = 0; s < pp.n_samples; ++s) {
        const float4 c = ld_stream(&radiance[(size_t)s * pp.npix_pad + pl]);
        r += c.x; g += c.y; b += c.z;
    }
    float* dst = accum + ((size_t)row * sc.width + x) * 3u;
    dst[0] += r; dst[1] += g; dst[2] += b;
}

// raytrace.rs:144-150: mean, sqrt gamma, clamp to <= 1 (no lower clamp);
// image.rs:37-43: (255.999 * c) as u8, saturating, NaN -> 0.
__device__ __forceinline__ uint8_t to_output(float c) {
    const float v = 255.999f * c;
    if (!(v == v)) return 0;
    if (v <= 0.f) return 0;
    if (v >= 255.f) return 255;
    return (uint8_t)v;
}
__global__ __launch_bounds__(256) void k_resolve(const float* __restrict__ accum, size_t n_values, float inv_spp,
                                                 uint8_t* __restrict__ rgb8, float* __restrict__ color) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    // f64::min(NaN, 1.0) == 1.0 in Rust, as fminf does here
    const float c = fminf(sqrtf(accum[i] * inv_spp), 1.0f);
    if (color) color[i] = c;
    if (rgb8) rgb8[i] = to_output(c);
}

// dst += src: the add of the multi-GPU exchange step (rt_accum_reduce), 16 bytes per lane
__global__ __launch_bounds__(256) void k_accum_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(dst)[i], b = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] += src[n4 * 4 + threadIdx.x];
}

// per-pixel hit record for rt_primary_hits
__global__ __launch_bounds__(256) void k_hit_info(DevScene sc, PassParams pp, RayQueue q, const float2* __restrict__ hits,
                                                  const uint32_t* __restrict__ ctl, uint32_t capacity, int32_t* __restrict__ obj,
                                                  int32_t* __restrict__ tri, float* __restrict__ tout, float* __restrict__ nout,
                                                  float* __restrict__ dout) {
    const uint32_t n = min(ctl[CTL_QCOUNT + 0], capacity);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 D = q.a[i];                   // iteration 0: (d, rng), slot == path id
    const float2 H = hits[i];
    const float4 A = make_float4(sc.eye[0], sc.eye[1], sc.eye[2], D.x), B = make_float4(D.y, D.z, 0.f, 0.f);
    const uint32_t pid = i;
    uint32_t x, row;
    if (__float_as_uint(H.y) == HIT_INVALID || !slot_to_pixel(sc, pp, pid % pp.npix_pad, x, row)) return;
    const size_t px = (size_t)row * sc.width + x;
    if (dout) { dout[px * 3 + 0] = D.x; dout[px * 3 + 1] = D.y; dout[px * 3 + 2] = D.z; }   // the ray k_raygen made (rt_primary_rays)
    if (!obj) return;
    uint32_t code = strip_hit(__float_as_uint(H.y));
    if (sc.n_tris && __float_as_uint(H.x) != HIT_MISS && strip_hit(__float_as_uint(H.x)) != code) {   // runner-up or candidate at t ~ 0: as in k_shade
        const d3 d64 = mkd((double)A.w, (double)B.x, (double)B.y);   // (the f32 ray rt_primary_rays reports, as for the triangle's t below)
        uint32_t near_tri = HIT_MISS;
        code = resolve_tie(sc, code, __float_as_uint(H.x), mkd(sc.eye64[0], sc.eye64[1], sc.eye64[2]), d64, near_tri);
    }
    int32_t o = -1, t = -1;
    float tt = -1.f;
    f3 nrm = mk(0, 0, 0);
    if (code != HIT_MISS) {
        tt = H.x;   // (scenes with triangles: the runner-up, see Best; the distance is formed again below)
        if (code & HIT_PLANE_BIT) {
            const uint32_t p = code & ~HIT_PLANE_BIT;
            const float4 P0 = sc.planes[2 * p], P1 = sc.planes[2 * p + 1];
            nrm = mk(P1.x, P1.y, P1.z);
            if (sc.n_tris) tt = dot(mk(P0.x, P0.y, P0.z) - mk(A.x, A.y, A.z), nrm) / dot(nrm, mk(A.w, B.x, B.y));
            o = sc.plane_object[p];
        } else {
            const float4 r0 = sc.prims[3 * code], r1 = sc.prims[3 * code + 1], r2 = sc.prims[3 * code + 2];
            const int4 M = sc.meta[code];
            o = M.x; t = M.y;
            if (__float_as_uint(r2.z) == PRIM_TRIANGLE) {
                const float4* __restrict__ rec = sc.smeta + SMETA_WORDS * (size_t)code;
                const float4 S = rec[0];
                nrm = mk(S.x, S.y, S.z);
                // the distance k_shade works with: the reference's f64 sequence on this very ray (the traversal's f32 t only
                // ranks the candidates; on a grazing hit it is good to 1e-4, not 1e-5)
                const double2* __restrict__ r64 = reinterpret_cast<const double2*>(rec + 1);
                const double2 w0 = r64[0], w1 = r64[1], w2 = r64[2], w3 = r64[3], w4 = r64[4];
                double t0, a, b;
                tri_hit_ref(mkd(sc.eye64[0], sc.eye64[1], sc.eye64[2]), mkd((double)A.w, (double)B.x, (double)B.y), mkd(w0.x, w0.y, w1.x),
                            mkd(w1.y, w2.x, w2.y), mkd(w3.x, w3.y, w4.x), t0, a, b);
                tt = (float)t0;
            } else {
                if (sc.n_tris) tt = code >= sc.n_leaf_prims ? sphere_hit64(r0, r1, mk(A.x, A.y, A.z), mk(A.w, B.x, B.y), false) : sphere_hit(r0, r1, mk(A.x, A.y, A.z), mk(A.w, B.x, B.y), false);
                nrm = sphere_normal(r0, r1, mk(A.x, A.y, A.z), mk(A.w, B.x, B.y), tt);
            }
        }
    }
    obj[px] = o; tri[px] = t; tout[px] = tt;
    nout[px * 3 + 0] = nrm.x; nout[px * 3 + 1] = nrm.y; nout[px * 3 + 2] = nrm.z;
}

// Calibration of the rocprofv3 FETCH_SIZE counter for this path's access pattern: every lane reads
// whole 64-byte records (four 16-byte loads, like a BVH node) at pseudo-random places of a table far
// larger than the Infinity Cache, so the bytes that must come from HBM are known exactly.
__global__ __launch_bounds__(256) void k_calib_gather(const float4* __restrict__ table, uint32_t n_records, uint32_t iters,
                                                      float* __restrict__ sink) {
    uint32_t h = pcg_hash(blockIdx.x * blockDim.x + threadIdx.x);
    float acc = 0.f;
    for (uint32_t k = 0; k < iters; ++k) {
        h = pcg_hash(h + k);
        const float4* r = table + 4 * (size_t)(h % n_records);
        const float4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) sink[0] = acc;   // keeps the loads alive
}

// the pass's number of path slots as the ray count of iteration 0 (what k_raygen does when it runs)
__global__ void k_pass_begin(uint32_t* __restrict__ ctl, uint32_t n_paths) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl[CTL_QCOUNT + 0] = n_paths;
}

// adds the pass's counters into the persistent 64-bit statistics
__global__ void k_stats(const uint32_t* __restrict__ ctl, unsigned long long* __restrict__ stats, uint32_t capacity) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 0 && i < CTL_MAX_IT) stats[i] += min(ctl[CTL_QCOUNT + i], capacity);   // iteration 0: host-side count of valid pixels
    if (i == CTL_MAX_IT) stats[i] += ctl[CTL_OVERFLOW];
    if (i > CTL_MAX_IT && i <= CTL_MAX_IT + N_STAT64) stats[i] += reinterpret_cast<const unsigned long long*>(&ctl[CTL_STAT64])[i - CTL_MAX_IT - 1];
}

}  // namespace

// ---------------------------------------------------------------- launch wrappers
static inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_raygen(const LaunchCfg& cfg, const DevScene& sc, const PassParams& pp, RayQueue q, float4* radiance, uint32_t* ctl,
                   uint32_t capacity, hipStream_t st) {
    uint32_t blocks = std::min<uint32_t>(cdiv(pp.n_paths, 256), cfg.n_cu * 16u);
    hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(256), 0, st, sc, pp, q, radiance, ctl, capacity);
}
void launch_raygen_entry(const LaunchCfg& cfg, const DevScene& sc, const PassParams& pp, RayQueue q, float4* radiance, uint32_t* ctl, int4* table,
                         hipStream_t st) {
    const uint32_t entry_blocks = cdiv((uint64_t)(pp.npix_pad >> 6) * (RT_ENTRY_SUB * RT_ENTRY_SUB), 64);
    const uint32_t blocks = std::min<uint32_t>(cdiv(pp.n_paths, 256), cfg.n_cu * 16u);
    hipLaunchKernelGGL(k_raygen_entry, dim3(entry_blocks + blocks), dim3(256), 0, st, sc, pp, q, radiance, ctl, table, entry_blocks);
}
void launch_pass_begin(uint32_t* ctl, uint32_t n_paths, hipStream_t st) {
    hipLaunchKernelGGL(k_pass_begin, dim3(1), dim3(64), 0, st, ctl, n_paths);
}
void launch_tile_entry(const DevScene& sc, const PassParams& pp, int4* table, hipStream_t st) {
    hipLaunchKernelGGL(k_tile_entry, dim3(cdiv((uint64_t)(pp.npix_pad >> 6) * (RT_ENTRY_SUB * RT_ENTRY_SUB), 64)), dim3(64), 0, st, sc, pp, table);
}
void launch_closest_hit(const LaunchCfg& cfg, const DevScene& sc, RayQueue q, float2* hits, uint32_t* ctl, int it, uint32_t capacity,
                        uint32_t max_rays, bool counters, TileEntry entry, bool any_hit, const PassParams* gen, hipStream_t st) {
    uint32_t per_cu = HIT_BLOCKS_PER_CU;
    if (const char* e = getenv("RTAMD_HIT_BLOCKS_PER_CU")) per_cu = std::min<uint32_t>(HIT_BLOCKS_PER_CU, std::max(1, atoi(e)));   // tuning experiments only
    uint32_t blocks = std::min<uint32_t>(cfg.n_cu * per_cu, std::max<uint32_t>(1, cdiv(max_rays, 256)));
    const size_t lds = 0;
    const dim3 g(blocks), b(256);
    const int ah = any_hit ? 1 : 0;
    PassParams pp0{};
    const PassParams& pp = gen ? *gen : pp0;
#define RT_HIT_LAUNCH(C, T, G) hipLaunchKernelGGL((k_closest_hit<C, T, G>), g, b, lds, st, sc, q, hits, ctl, it, capacity, entry, ah, pp)
    if (sc.n_tris) {
        if (gen) { if (counters) RT_HIT_LAUNCH(true, true, true); else RT_HIT_LAUNCH(false, true, true); }
        else { if (counters) RT_HIT_LAUNCH(true, true, false); else RT_HIT_LAUNCH(false, true, false); }
    } else {
        if (gen) { if (counters) RT_HIT_LAUNCH(true, false, true); else RT_HIT_LAUNCH(false, false, true); }
        else { if (counters) RT_HIT_LAUNCH(true, false, false); else RT_HIT_LAUNCH(false, false, false); }
    }
#undef RT_HIT_LAUNCH
}
void launch_shade(const LaunchCfg& cfg, const DevScene& sc, RayQueue qi, const float2* hits, RayQueue qo, float4* radiance,
                  uint32_t* ctl, int it, int last, uint32_t capacity, uint32_t max_rays, const PassParams* gen, int cls, uint32_t* heavy, hipStream_t st) {
    uint32_t blocks = std::min<uint32_t>(cfg.n_cu * RT_SHADE_GRID_MUL, std::max<uint32_t>(1, cdiv(max_rays, SHADE_THREADS)));
    const dim3 g(blocks), b(SHADE_THREADS);
    PassParams pp0{};
    const PassParams& pp = gen ? *gen : pp0;
#define RT_SHADE_LAUNCH(T, P, C) hipLaunchKernelGGL((k_shade<T, P, C>), g, b, 0, st, sc, qi, hits, qo, radiance, ctl, it, last, capacity, pp, heavy)
    if (sc.n_tris) {
        if (it == 0) { if (gen) RT_SHADE_LAUNCH(true, 2, 0); else RT_SHADE_LAUNCH(true, 1, 0); }
        else if (cls == 1) RT_SHADE_LAUNCH(true, 0, 1);
        else if (cls == 2) RT_SHADE_LAUNCH(true, 0, 2);
        else RT_SHADE_LAUNCH(true, 0, 0);
    } else {
        if (it == 0) { if (gen) RT_SHADE_LAUNCH(false, 2, 0); else RT_SHADE_LAUNCH(false, 1, 0); }
        else RT_SHADE_LAUNCH(false, 0, 0);
    }
#undef RT_SHADE_LAUNCH
}
void launch_accumulate(const DevScene& sc, const PassParams& pp, const float4* radiance, float* accum, hipStream_t st) {
    hipLaunchKernelGGL(k_accumulate, dim3(cdiv(pp.npix_pad, 256)), dim3(256), 0, st, sc, pp, radiance, accum);
}
void launch_resolve(const float* accum, size_t n_values, float inv_spp, uint8_t* rgb8, float* color, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, dim3(cdiv(n_values, 256)), dim3(256), 0, st, accum, n_values, inv_spp, rgb8, color);
}
void launch_hit_info(const DevScene& sc, const PassParams& pp, RayQueue q, const float2* hits, const uint32_t* ctl, uint32_t capacity,
                     int32_t* obj, int32_t* tri, float* t, float* nrm, float* dir, hipStream_t st) {
    hipLaunchKernelGGL(k_hit_info, dim3(cdiv(pp.n_paths, 256)), dim3(256), 0, st, sc, pp, q, hits, ctl, capacity, obj, tri, t, nrm, dir);
}
void launch_accum_add(const LaunchCfg& cfg, float* dst, const float* src, size_t n, hipStream_t st) {
    const uint32_t blocks = std::min<uint32_t>(cfg.n_cu * 8u, std::max<uint32_t>(1, cdiv(n / 4 + 1, 256)));
    hipLaunchKernelGGL(k_accum_add, dim3(blocks), dim3(256), 0, st, dst, src, n);
}
void launch_calib_gather(const float4* table, uint32_t n_records, uint32_t iters, uint32_t blocks, float* sink, hipStream_t st) {
    hipLaunchKernelGGL(k_calib_gather, dim3(blocks), dim3(256), 0, st, table, n_records, iters, sink);
}
void launch_stats(const uint32_t* ctl, unsigned long long* stats, uint32_t capacity, hipStream_t st) {
    hipLaunchKernelGGL(k_stats, dim3(cdiv(CTL_MAX_IT + 1 + N_STAT64, 64)), dim3(64), 0, st, ctl, stats, capacity);
}

}  // namespace rtamd
