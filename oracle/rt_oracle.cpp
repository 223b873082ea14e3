// ============================================================================
// rt_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain C++ restatement (templated on the real type, f64 by default) of the
// algorithm of bsati/raytracer-rust's render path.  It is NOT part of the
// product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load it, and only as the checker / the reported CPU baseline.
//
// The reference (Rust) cannot be compiled in this environment (no rustc/cargo),
// so this restatement is pinned by the reference's own inline known-answer
// tests (see tests/test_oracle_kat.py) and by the reference's two committed renders
// (outputs of the real reference, settings unknown): besides their deterministic
// plateaus, a converged oracle render agrees with them to 46.9 dB (spheres) and
// 52.4 dB (cube) at 1024 spp, mean difference 0.004 of 255 levels -- that pins
// Ray::trace / compute_image / Camera / texture lookup / OBJ import, which have no
// reference unit test (the two scenes use all five materials, spheres, planes, a
// textured OBJ mesh and two lights).  What neither scene has (v//vn faces, several
// `o` objects, mesh lights) stays "parity unpinned" (SURVEY.md section 8c).
//
// Optional conservative cull (orc_enable_accel, OFF by default, never used by the timed cpu_baseline): the
// reference tests EVERY triangle of a mesh for every ray (intersections.rs:148-186), 33 ms per ray and core on
// the 1,000,000-triangle mesh, which puts converged-image parity at full size out of reach of a test.  With the
// cull on, mesh_intersect runs the SAME per-triangle arithmetic, but only on the triangles of the BVH leaves whose
// (padded) boxes the ray's line passes, and picks the winner by (t, triangle index), which is what the reference's
// strict '>' in file order yields.  Culled triangles are ones the brute-force loop would have rejected, so results
// are bit-identical; tests/test_oracle_kat.py checks exactly that (hits and whole renders, cull on against off).
//
// Every function cites the reference file:line (relative to /root/reference/)
// that it follows.  The real type R is `double` for the oracle proper; the
// `float` instantiation exists only as a diagnostic (how much does f32 change?).
// ============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------
// RNG.  The reference uses rand::thread_rng() (ChaCha12, OS seeded): its stream
// is not a parity target.  We use xoshiro256++ seeded by
// splitmix64 so that renders are reproducible per (seed, pixel).
// ---------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        uint64_t x = seed;
        for (int i = 0; i < 4; ++i) s[i] = splitmix64(x);
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return result;
    }
    // rng.gen::<f64>(): uniform in [0,1), 53 bits
    double gen() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    // gen_range(lo..hi) half open; gen_range(lo..=hi) differs only on a null set
    double gen_range(double lo, double hi) { return lo + (hi - lo) * gen(); }
};

// ---------------------------------------------------------------------------
// Vector3 -- src/math/vec3.rs
// ---------------------------------------------------------------------------
template <class R>
struct V3 {
    R d[3];
    V3() : d{0, 0, 0} {}
    V3(R x, R y, R z) : d{x, y, z} {}
    R operator[](int i) const { return d[i]; }
    R& operator[](int i) { return d[i]; }
    V3 operator+(const V3& o) const { return V3(d[0] + o.d[0], d[1] + o.d[1], d[2] + o.d[2]); }  // vec3.rs:200-206
    V3 operator-(const V3& o) const { return V3(d[0] - o.d[0], d[1] - o.d[1], d[2] - o.d[2]); }  // vec3.rs:209-215
    V3 operator*(R s) const { return V3(d[0] * s, d[1] * s, d[2] * s); }                         // vec3.rs:227-233
    V3 operator/(R s) const { return V3(d[0] / s, d[1] / s, d[2] / s); }                         // vec3.rs:236-242
    V3 operator-() const { return V3(-d[0], -d[1], -d[2]); }                                     // vec3.rs:244-250
    R dot(const V3& o) const { return d[0] * o.d[0] + d[1] * o.d[1] + d[2] * o.d[2]; }           // vec3.rs:81-83
    R sqr_len() const { return dot(*this); }                                                     // vec3.rs:94-96
    R len() const { return std::sqrt(sqr_len()); }                                               // vec3.rs:87-89
    V3 cross(const V3& o) const {                                                                // vec3.rs:104-110
        return V3(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]);
    }
    V3 normalized() const {                                                                      // vec3.rs:115-121
        R n = len();
        if (n != (R)0) return V3(d[0] / n, d[1] / n, d[2] / n);
        return *this;
    }
    V3 reflect(const V3& n) const { return *this - n * ((R)2 * dot(n)); }                        // vec3.rs:129-131
    V3 vmin(const V3& o) const { return V3(std::fmin(d[0], o.d[0]), std::fmin(d[1], o.d[1]), std::fmin(d[2], o.d[2])); }  // vec3.rs:140-146
    V3 vmax(const V3& o) const { return V3(std::fmax(d[0], o.d[0]), std::fmax(d[1], o.d[1]), std::fmax(d[2], o.d[2])); }  // vec3.rs:155-161
    bool near_zero() const {                                                                     // vec3.rs:55-58
        const R eps = (R)1e-8;
        return std::fabs(d[0]) < eps && std::fabs(d[1]) < eps && std::fabs(d[2]) < eps;
    }
    static V3 random(Rng& g, R lo, R hi) {                                                       // vec3.rs:30-37
        R x = (R)g.gen_range(lo, hi), y = (R)g.gen_range(lo, hi), z = (R)g.gen_range(lo, hi);
        return V3(x, y, z);
    }
    static V3 random_in_unit_sphere(Rng& g) {                                                    // vec3.rs:40-47
        for (;;) {
            V3 v = random(g, (R)-1, (R)1);
            if (v.sqr_len() < (R)1) return v;
        }
    }
    static V3 random_unit_vector(Rng& g) { return random_in_unit_sphere(g).normalized(); }      // vec3.rs:50-52
};

// Color -- src/raytracer/image.rs:10-115 (same arithmetic as a 3-vector)
template <class R>
struct Col {
    R r, g, b;
    Col() : r(0), g(0), b(0) {}
    Col(R r_, R g_, R b_) : r(r_), g(g_), b(b_) {}
    Col operator+(const Col& o) const { return Col(r + o.r, g + o.g, b + o.b); }   // image.rs:77-83
    Col operator*(const Col& o) const { return Col(r * o.r, g * o.g, b * o.b); }   // image.rs:101-107
    Col operator*(R s) const { return Col(r * s, g * s, b * s); }                  // image.rs:109-115
    void operator+=(const Col& o) { r += o.r; g += o.g; b += o.b; }                // image.rs:85-91
    void operator/=(R s) { r /= s; g /= s; b /= s; }                               // image.rs:93-99
    void clamp() { r = std::fmin(r, (R)1); g = std::fmin(g, (R)1); b = std::fmin(b, (R)1); }  // image.rs:29-33
};

// Rust `as u8`: saturating, NaN -> 0.   image.rs:37-43
static inline uint8_t sat_u8(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

// ---------------------------------------------------------------------------
// Ray -- src/raytracer/raytrace.rs:20-43
// ---------------------------------------------------------------------------
template <class R>
struct Ray {
    V3<R> origin, direction;
    Ray() {}
    Ray(const V3<R>& o, const V3<R>& d) : origin(o), direction(d.normalized()) {}   // raytrace.rs:32-37
    V3<R> at(R t) const { return origin + direction * t; }                          // raytrace.rs:40-42
};

// ---------------------------------------------------------------------------
// Materials -- src/raytracer/scene/materials.rs
// ---------------------------------------------------------------------------
enum MatType { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2, TEXTURE = 3, EMISSIVE = 4 };

template <class R>
struct Material {
    int type = LAMBERTIAN;
    Col<R> color;          // albedo / tint / emissive colour
    R param = 0;           // fuzziness (Metal) or refraction_index (Dieletrics)
    // Texture (materials.rs:185-194): pixel_colors = byte/255.999 (image.rs:45-51)
    std::vector<Col<R>> pixels;
    R tex_w = 0, tex_h = 0;
};

template <class R>
struct Hit {                       // IntersectionInfo, intersections.rs:18-25  (+ ids)
    V3<R> point, normal;
    const Material<R>* material = nullptr;
    R t = 0;
    bool has_uv = false;
    R u = 0, v = 0;
    int32_t object_index = -1, triangle_index = -1;
};

template <class R>
struct Scatter {                   // Option<(Option<Ray>, Color)>, materials.rs:14-16
    bool some = false;             // outer Option
    bool has_ray = false;          // inner Option
    Ray<R> ray;
    Col<R> color;
};

static std::atomic<uint64_t> g_tex_oob{0};

// materials.rs:222-230 (index y*W + x).  The reference would panic on an
// out-of-range index; the oracle clamps and counts it (never happens on cube).
template <class R>
static Col<R> tex_pixel(const Material<R>& m, size_t x, size_t y) {
    size_t idx = y * (size_t)m.tex_w + x;
    if (idx >= m.pixels.size()) { g_tex_oob++; idx = m.pixels.empty() ? 0 : m.pixels.size() - 1; }
    if (m.pixels.empty()) return Col<R>();
    return m.pixels[idx];
}

// materials.rs:197-220
template <class R>
static Col<R> get_albedo(const Material<R>& m, R u, R v) {
    R x = m.tex_w * u;
    R y = m.tex_h - (m.tex_h * v);
    if (std::fmod(x, (R)1) != (R)0 || std::fmod(y, (R)1) != (R)0) {
        size_t x_top = (size_t)std::ceil(x), y_top = (size_t)std::ceil(y);
        size_t x_bot = (size_t)std::floor(x), y_bot = (size_t)std::floor(y);
        Col<R> c00 = tex_pixel(m, x_bot, y_bot);
        Col<R> c01 = tex_pixel(m, x_bot, y_top);
        Col<R> c10 = tex_pixel(m, x_top, y_bot);
        Col<R> c11 = tex_pixel(m, x_top, y_top);
        R fx = x - std::floor(x), fy = y - std::floor(y);
        return c00 * ((R)1 - fx) * ((R)1 - fy) + c10 * fx * ((R)1 - fy) + c01 * ((R)1 - fx) * fy + c11 * fx * fy;
    }
    return tex_pixel(m, (size_t)x, (size_t)y);
}

// materials.rs:105-110
template <class R>
static V3<R> refract(const V3<R>& direction, const V3<R>& normal, R ratio) {
    R cos_theta = std::fmin(normal.dot(-direction), (R)1);
    V3<R> r_out_perp = (direction + normal * cos_theta) * ratio;
    V3<R> r_out_parallel = normal * -std::sqrt(std::fabs((R)1 - r_out_perp.sqr_len()));
    return r_out_perp + r_out_parallel;
}
// materials.rs:113-116
template <class R>
static R reflectance(R cosine, R ref_idx) {
    R r0 = ((R)1 - ref_idx) / ((R)1 + ref_idx);
    r0 = r0 * r0;
    R x = (R)1 - cosine;
    return r0 + ((R)1 - r0) * (x * x * x * x * x);
}

// materials.rs:28-38 dispatch + the five impls
template <class R>
static Scatter<R> scatter(const Material<R>& m, const Ray<R>& ray, const Hit<R>& hit, Rng& g) {
    Scatter<R> s;
    switch (m.type) {
    case LAMBERTIAN: {                                           // materials.rs:52-66
        V3<R> dir = hit.normal + V3<R>::random_unit_vector(g);
        if (dir.near_zero()) dir = hit.normal;
        s.some = true; s.has_ray = true; s.ray = Ray<R>(hit.point, dir); s.color = m.color;
        return s;
    }
    case EMISSIVE:                                               // materials.rs:80-88
        s.some = true; s.has_ray = false; s.color = m.color;
        return s;
    case DIELECTRIC: {                                           // materials.rs:119-153
        bool front_face = hit.normal.dot(ray.direction) <= (R)0;
        R ratio = front_face ? (R)1 / m.param : m.param;
        V3<R> normal = front_face ? hit.normal : -hit.normal;
        V3<R> unit_direction = ray.direction;
        R cos_theta = std::fmin((-unit_direction).dot(normal), (R)1);
        R sin_theta = std::sqrt((R)1 - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > (R)1;
        V3<R> direction = refract(unit_direction, normal, ratio);
        // short-circuit: the RNG draw is skipped when cannot_refract (materials.rs:143-144)
        if (cannot_refract || reflectance(cos_theta, ratio) > (R)g.gen()) direction = unit_direction.reflect(normal);
        s.some = true; s.has_ray = true; s.ray = Ray<R>(hit.point, direction); s.color = m.color;
        return s;
    }
    case METAL: {                                                // materials.rs:168-183
        V3<R> reflected = ray.direction.normalized().reflect(hit.normal);
        Ray<R> scattered(hit.point, reflected + V3<R>::random_in_unit_sphere(g) * m.param);
        if (scattered.direction.dot(hit.normal) > (R)0) {
            s.some = true; s.has_ray = true; s.ray = scattered; s.color = m.color;
            return s;
        }
        s.some = false;
        return s;
    }
    case TEXTURE: {                                              // materials.rs:233-253
        // u/v are unwrap()ed first: a Texture on a mesh without vt panics in the
        // reference; the oracle uses (0,0) then (never happens on the fixtures).
        Col<R> att = get_albedo(m, hit.has_uv ? hit.u : (R)0, hit.has_uv ? hit.v : (R)0);
        if (ray.direction.dot(hit.normal) > (R)0) { s.some = true; s.has_ray = false; s.color = att; return s; }
        V3<R> dir = hit.normal + V3<R>::random_unit_vector(g);
        if (dir.near_zero() || hit.normal.dot(dir) <= (R)0) dir = hit.normal;
        s.some = true; s.has_ray = true; s.ray = Ray<R>(hit.point, dir); s.color = att;
        return s;
    }
    }
    return s;
}

// ---------------------------------------------------------------------------
// Objects and intersections -- scene.rs:174-186, mesh.rs:295-373, intersections.rs
// ---------------------------------------------------------------------------
template <class R>
struct AABB { V3<R> mn, mx; };

// intersections.rs:192-251 (Woo)
template <class R>
static bool aabb_intersect(const AABB<R>& bb, const Ray<R>& ray) {
    enum { LEFT = 0, RIGHT = 1, MIDDLE = 2, NONE = 3 };
    int quadrant[3] = {NONE, NONE, NONE};
    R candidate[3] = {-1, -1, -1};
    bool inside = true;
    for (int i = 0; i < 3; ++i) {
        if (ray.origin[i] < bb.mn[i]) { quadrant[i] = LEFT; candidate[i] = bb.mn[i]; inside = false; }
        else if (ray.origin[i] > bb.mx[i]) { quadrant[i] = RIGHT; candidate[i] = bb.mx[i]; inside = false; }
        else quadrant[i] = MIDDLE;
    }
    if (inside) return true;
    R max_t[3] = {-1, -1, -1};
    for (int i = 0; i < 3; ++i)
        if (quadrant[i] != MIDDLE && ray.direction[i] != (R)0) max_t[i] = (candidate[i] - ray.origin[i]) / ray.direction[i];
    int which = 0;
    for (int i = 0; i < 3; ++i)
        if (max_t[which] < max_t[i]) which = i;
    for (int i = 0; i < 3; ++i) {
        if (which != i) {
            R c = ray.origin[i] + max_t[which] * ray.direction[i];
            if (c < bb.mn[i] || c > bb.mx[i]) return false;
        }
    }
    return true;
}

template <class R>
struct Sphere { V3<R> center; R radius; int material; };
template <class R>
struct Plane { V3<R> center, normal; int material; };
struct Triangle {                       // mesh.rs:351-356
    size_t vertex_idx[3];
    bool has_uv = false;
    size_t uv_idx[3] = {0, 0, 0};
    size_t material_idx = 0;
};
template <class R>
struct Mesh {                           // mesh.rs:295-302
    std::vector<Triangle> triangles;
    std::vector<int> materials;         // indices into Scene::materials
    std::vector<V3<R>> vertex_positions;
    std::vector<std::pair<R, R>> uvs;
    bool has_aabb = false;
    AABB<R> aabb;
    // conservative cull (test infrastructure, see the file header): binary BVH over padded triangle boxes
    struct AccelNode { double mn[3], mx[3]; int32_t left, right; uint32_t first, count; };
    std::vector<AccelNode> accel_nodes;
    std::vector<uint32_t> accel_order;      // triangle indices in leaf order
    std::vector<uint32_t> accel_always;     // triangles with a zero or non-finite normal: M == 0 for every ray, never culled
    void compute_aabb() {               // mesh.rs:318-326
        const R big = std::numeric_limits<R>::max();
        V3<R> mn(big, big, big), mx(-big, -big, -big);
        for (auto& vp : vertex_positions) { mn = mn.vmin(vp); mx = mx.vmax(vp); }
        aabb.mn = mn; aabb.mx = mx; has_aabb = true;
    }
};

enum ObjKind { OBJ_SPHERE = 0, OBJ_PLANE = 1, OBJ_MESH = 2 };
struct ObjRef { int kind; int index; };

template <class R>
struct Scene {
    size_t width = 0, height = 0;
    Col<R> background;
    V3<R> eye, look_at, up;
    R fovy = 0;
    std::vector<Material<R>> materials;
    std::vector<Sphere<R>> spheres;
    std::vector<Plane<R>> planes;
    std::vector<Mesh<R>> meshes;
    std::vector<ObjRef> objects;              // YAML order == object index
    std::vector<std::vector<V3<R>>> lights;   // Light::sample_points per light
    // camera.rs:7-12
    V3<R> cam_eye, cam_horizontal, cam_vertical, cam_lower_left;
    std::atomic<uint64_t> rays_traced{0};
};

// intersections.rs:40-80
template <class R>
static bool sphere_intersect(const Scene<R>& sc, const Sphere<R>& s, const Ray<R>& ray, Hit<R>& out) {
    V3<R> dir = ray.direction;
    V3<R> oc = ray.origin - s.center;
    R a = dir.sqr_len();
    R b = (R)2 * dir.dot(oc);
    R c = oc.sqr_len() - s.radius * s.radius;
    R d = b * b - (R)4 * a * c;
    if (d >= (R)0) {
        d = std::sqrt(d);
        R t1 = (-b - d) / ((R)2 * a);
        R t2 = (-b + d) / ((R)2 * a);
        const R MAXV = std::numeric_limits<R>::max();
        R t = MAXV;
        if (t1 > (R)1e-5 && t1 < t) t = t1;
        if (t2 > (R)1e-5 && t2 < t) t = t2;
        if (t == MAXV) return false;
        out = Hit<R>();
        out.point = ray.at(t);
        out.normal = (out.point - s.center) / s.radius;
        out.material = &sc.materials[s.material];
        out.t = t;
        return true;
    }
    return false;
}

// intersections.rs:82-104
template <class R>
static bool plane_intersect(const Scene<R>& sc, const Plane<R>& p, const Ray<R>& ray, Hit<R>& out) {
    R dot_nd = p.normal.dot(ray.direction);
    if (std::fabs(dot_nd) < (R)1e-6) return false;
    R t = (p.center - ray.origin).dot(p.normal) / dot_nd;
    if (t < (R)1e-5) return false;
    out = Hit<R>();
    out.point = ray.at(t);
    out.normal = p.normal;
    out.material = &sc.materials[p.material];
    out.t = t;
    return true;
}

// intersections.rs:123-126
template <class R>
static R determinant(const V3<R>& v1, const V3<R>& v2, const V3<R>& v3) {
    return (v1[0] * v2[1] * v3[2] + v2[0] * v3[1] * v1[2] + v3[0] * v1[1] * v2[2]) -
           (v3[0] * v2[1] * v1[2] + v2[0] * v1[1] * v3[2] + v1[0] * v3[1] * v2[2]);
}

// One iteration of the triangle loop of intersections.rs:148-186: true when the candidate passes the reject test
// (NaN passes, as in the reference); fills t and, when asked, the full record.
template <class R>
static inline bool triangle_candidate(const Scene<R>& sc, const Mesh<R>& m, size_t ti, const Ray<R>& ray, R& t_out, Hit<R>* info) {
    const Triangle& tr = m.triangles[ti];
    V3<R> A = m.vertex_positions[tr.vertex_idx[0]];
    V3<R> B = m.vertex_positions[tr.vertex_idx[1]];
    V3<R> C = m.vertex_positions[tr.vertex_idx[2]];
    V3<R> ab = B - A, ac = C - A;
    V3<R> res = ray.origin - A;
    R det_m = determinant(ray.direction, ab, ac);
    R det_m_t = determinant(res, ab, ac);
    R det_m_a = determinant(ray.direction, res, ac);
    R det_m_b = determinant(ray.direction, ab, res);
    R a = det_m_a / det_m;
    R b = det_m_b / det_m;
    R t = -(det_m_t / det_m);
    if (a < (R)0 || b < (R)0 || a + b > (R)1 || t < (R)0) return false;   // NaN passes, as in the reference
    t_out = t;
    if (info) {
        *info = Hit<R>();
        info->point = ray.at(t);
        info->normal = ab.cross(ac).normalized();
        info->material = tr.material_idx < m.materials.size() ? &sc.materials[m.materials[tr.material_idx]] : nullptr;
        info->t = t;
        info->triangle_index = (int32_t)ti;
        if (tr.has_uv) {
            auto uv1 = m.uvs[tr.uv_idx[1]];
            auto uv2 = m.uvs[tr.uv_idx[2]];
            auto uv3 = m.uvs[tr.uv_idx[0]];
            info->has_uv = true;
            info->u = a * uv1.first + b * uv2.first + ((R)1 - a - b) * uv3.first;
            info->v = a * uv1.second + b * uv2.second + ((R)1 - a - b) * uv3.second;
        }
    }
    return true;
}

// The same result as the loop of mesh_intersect below, from the triangles of the BVH leaves whose padded boxes the
// ray's LINE passes (no cut by distance: nothing depends on rounding of t).  The loop keeps the first accepted
// candidate and replaces it only when best.t > t, in file order: the winner is the accepted candidate with the
// smallest (t, index) -- unless the FIRST accepted candidate has t = NaN, which then sticks.
template <class R>
static bool mesh_intersect_culled(const Scene<R>& sc, const Mesh<R>& m, const Ray<R>& ray, Hit<R>& out) {
    const double o[3] = {(double)ray.origin[0], (double)ray.origin[1], (double)ray.origin[2]};
    const double d[3] = {(double)ray.direction[0], (double)ray.direction[1], (double)ray.direction[2]};
    bool have = false, have_nan = false;
    R best_t = 0;
    size_t best_i = 0, nan_i = 0, min_i = (size_t)-1;
    auto consider = [&](size_t ti) {
        R t;
        if (!triangle_candidate<R>(sc, m, ti, ray, t, nullptr)) return;
        if (ti < min_i) min_i = ti;
        if (t != t) { if (!have_nan || ti < nan_i) { nan_i = ti; have_nan = true; } return; }
        if (!have || t < best_t || (t == best_t && ti < best_i)) { best_t = t; best_i = ti; have = true; }
    };
    for (uint32_t ti : m.accel_always) consider(ti);
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const typename Mesh<R>::AccelNode& n = m.accel_nodes[(size_t)stack[--sp]];
        // slab test of the whole line against the padded box, in f64; axis-parallel components test containment
        double t0 = -std::numeric_limits<double>::infinity(), t1 = std::numeric_limits<double>::infinity();
        bool miss = false;
        for (int k = 0; k < 3 && !miss; ++k) {
            if (d[k] == 0.0) { if (o[k] < n.mn[k] || o[k] > n.mx[k]) miss = true; continue; }
            double a = (n.mn[k] - o[k]) / d[k], b = (n.mx[k] - o[k]) / d[k];
            if (a > b) std::swap(a, b);
            // one part in 1e9 of slack on the parametric interval, on top of the padded box
            a -= 1e-9 * (std::fabs(a) + 1.0); b += 1e-9 * (std::fabs(b) + 1.0);
            if (a > t0) t0 = a;
            if (b < t1) t1 = b;
        }
        if (miss || t0 > t1) continue;
        if (n.left < 0) { for (uint32_t k = 0; k < n.count; ++k) consider(m.accel_order[n.first + k]); }
        else if (sp + 2 <= 128) { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
    size_t win;
    if (have_nan && nan_i == min_i) win = nan_i;      // the first accepted candidate is NaN: it is never replaced
    else if (have) win = best_i;
    else return false;
    R t;
    triangle_candidate<R>(sc, m, win, ray, t, &out);
    return true;
}

// intersections.rs:139-189
template <class R>
static bool mesh_intersect(const Scene<R>& sc, const Mesh<R>& m, const Ray<R>& ray, Hit<R>& out) {
    bool found = false;
    if (m.has_aabb && !aabb_intersect(m.aabb, ray)) return false;
    if (!m.accel_nodes.empty()) return mesh_intersect_culled(sc, m, ray, out);
    for (size_t ti = 0; ti < m.triangles.size(); ++ti) {
        R t;
        Hit<R> info;
        if (!triangle_candidate<R>(sc, m, ti, ray, t, nullptr)) continue;
        if (!found || out.t > t) {
            triangle_candidate<R>(sc, m, ti, ray, t, &info);
            out = info;
            found = true;
        }
    }
    return found;
}

// Builds the cull structure of one mesh: median splits on the widest centroid axis, leaves of <= 4 triangles,
// boxes padded by 1e-7 of their size + 1e-12 (the triangle arithmetic itself is exact to ~1e-16 relative).
template <class R>
static void build_accel(Mesh<R>& m) {
    const size_t n = m.triangles.size();
    m.accel_nodes.clear(); m.accel_order.clear(); m.accel_always.clear();
    std::vector<double> lo(3 * n), hi(3 * n), cen(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const Triangle& tr = m.triangles[i];
        const V3<R> A = m.vertex_positions[tr.vertex_idx[0]], B = m.vertex_positions[tr.vertex_idx[1]], C = m.vertex_positions[tr.vertex_idx[2]];
        const V3<R> nn = (B - A).cross(C - A);
        const bool finite = std::isfinite((double)nn[0]) && std::isfinite((double)nn[1]) && std::isfinite((double)nn[2]);
        if (!finite || (nn[0] == (R)0 && nn[1] == (R)0 && nn[2] == (R)0)) { m.accel_always.push_back((uint32_t)i); continue; }
        for (int k = 0; k < 3; ++k) {
            lo[3 * i + k] = std::min({(double)A[k], (double)B[k], (double)C[k]});
            hi[3 * i + k] = std::max({(double)A[k], (double)B[k], (double)C[k]});
            cen[3 * i + k] = 0.5 * (lo[3 * i + k] + hi[3 * i + k]);
        }
        m.accel_order.push_back((uint32_t)i);
    }
    if (m.accel_order.empty()) { m.accel_nodes.push_back({{0, 0, 0}, {-1, -1, -1}, -1, -1, 0, 0}); return; }
    struct Job { int32_t node; uint32_t first, count; };
    std::vector<Job> jobs;
    m.accel_nodes.push_back({});
    jobs.push_back({0, 0, (uint32_t)m.accel_order.size()});
    while (!jobs.empty()) {
        const Job j = jobs.back(); jobs.pop_back();
        typename Mesh<R>::AccelNode node{};
        double cmn[3], cmx[3];
        for (int k = 0; k < 3; ++k) { node.mn[k] = cmn[k] = 1e300; node.mx[k] = cmx[k] = -1e300; }
        for (uint32_t q = 0; q < j.count; ++q) {
            const uint32_t ti = m.accel_order[j.first + q];
            for (int k = 0; k < 3; ++k) {
                node.mn[k] = std::min(node.mn[k], lo[3 * ti + k]); node.mx[k] = std::max(node.mx[k], hi[3 * ti + k]);
                cmn[k] = std::min(cmn[k], cen[3 * ti + k]); cmx[k] = std::max(cmx[k], cen[3 * ti + k]);
            }
        }
        for (int k = 0; k < 3; ++k) {
            const double pad = 1e-7 * (node.mx[k] - node.mn[k]) + 1e-7 * std::max(std::fabs(node.mn[k]), std::fabs(node.mx[k])) + 1e-12;
            node.mn[k] -= pad; node.mx[k] += pad;
        }
        node.left = node.right = -1; node.first = j.first; node.count = j.count;
        if (j.count > 4) {
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
            const uint32_t half = j.count / 2;
            uint32_t* b = m.accel_order.data() + j.first;
            std::nth_element(b, b + half, b + j.count, [&](uint32_t x, uint32_t y) {
                return cen[3 * x + axis] < cen[3 * y + axis] || (cen[3 * x + axis] == cen[3 * y + axis] && x < y);
            });
            node.left = (int32_t)m.accel_nodes.size(); m.accel_nodes.push_back({});
            node.right = (int32_t)m.accel_nodes.size(); m.accel_nodes.push_back({});
            jobs.push_back({node.left, j.first, half});
            jobs.push_back({node.right, j.first + half, j.count - half});
        }
        m.accel_nodes[(size_t)j.node] = node;
    }
}

// scene.rs:35-52 (+ object ids; first object wins ties because of strict '>')
template <class R>
static bool closest_intersection(const Scene<R>& sc, const Ray<R>& ray, Hit<R>& info) {
    bool have = false;
    for (size_t oi = 0; oi < sc.objects.size(); ++oi) {
        Hit<R> h;
        bool hit = false;
        const ObjRef& o = sc.objects[oi];
        if (o.kind == OBJ_SPHERE) hit = sphere_intersect(sc, sc.spheres[o.index], ray, h);
        else if (o.kind == OBJ_PLANE) hit = plane_intersect(sc, sc.planes[o.index], ray, h);
        else hit = mesh_intersect(sc, sc.meshes[o.index], ray, h);
        if (hit) {
            h.object_index = (int32_t)oi;
            if (have) { if (info.t > h.t) info = h; }
            else { info = h; have = true; }
        }
    }
    return have;
}

// scene.rs:91-111 is_light, scene.rs:124-146 Light::from, scene.rs:54-72 precompute
// (texture pixels are supplied already decoded by the caller).
template <class R>
static void precompute(Scene<R>& sc) {
    sc.lights.clear();
    for (auto& o : sc.objects) {
        if (o.kind == OBJ_MESH) {
            Mesh<R>& m = sc.meshes[o.index];
            m.compute_aabb();
            bool is_light = false;
            for (int mi : m.materials) if (sc.materials[mi].type == EMISSIVE) { is_light = true; break; }
            if (is_light) {
                std::vector<V3<R>> pts;
                for (auto& tr : m.triangles) {
                    if (tr.material_idx < m.materials.size() && sc.materials[m.materials[tr.material_idx]].type == EMISSIVE) {
                        V3<R> p = (m.vertex_positions[tr.vertex_idx[0]] + m.vertex_positions[tr.vertex_idx[1]] +
                                   m.vertex_positions[tr.vertex_idx[2]]) / (R)3;
                        pts.push_back(p);
                    }
                }
                sc.lights.push_back(pts);
            }
        } else if (o.kind == OBJ_SPHERE) {
            if (sc.materials[sc.spheres[o.index].material].type == EMISSIVE) sc.lights.push_back({sc.spheres[o.index].center});
        } else {
            if (sc.materials[sc.planes[o.index].material].type == EMISSIVE) sc.lights.push_back({sc.planes[o.index].center});
        }
    }
}

// camera.rs:26-54
template <class R>
static void camera_new(Scene<R>& sc) {
    V3<R> view_vec = sc.look_at - sc.eye;
    R distance = view_vec.len();
    V3<R> view = view_vec / distance;
    R aspect = (R)sc.width / (R)sc.height;
    R image_height = (R)2 * distance * std::tan((R)0.5 * (sc.fovy * (R)(M_PI / 180.0)));
    R image_width = image_height * aspect;
    V3<R> horizontal = view.cross(sc.up).normalized() * image_width / (R)sc.width;
    V3<R> vertical = horizontal.cross(view).normalized() * image_height / (R)sc.height;
    V3<R> lower_left = sc.look_at - horizontal * ((R)0.5 * (R)sc.width) - vertical * ((R)0.5 * (R)sc.height);
    sc.cam_eye = sc.eye; sc.cam_horizontal = horizontal; sc.cam_vertical = vertical; sc.cam_lower_left = lower_left;
}
// camera.rs:62-67
template <class R>
static Ray<R> spawn_ray(const Scene<R>& sc, R x, R y) {
    return Ray<R>(sc.cam_eye, sc.cam_lower_left + sc.cam_horizontal * x + sc.cam_vertical * y - sc.cam_eye);
}

// raytrace.rs:51-96.  `nest` only guards the C++ stack: the reference recursion
// through shadow rays is unbounded in principle; it ends almost
// surely when L*prob*L < 1.  nest counts shadow-ray generations.
static const int kMaxShadowNest = 64;
template <class R>
static Col<R> trace(const Ray<R>& ray, Scene<R>& sc, int current_depth, int max_depth, Rng& g, int nest, uint64_t& nrays) {
    if (current_depth == max_depth) return Col<R>(0, 0, 0);
    Hit<R> hit;
    nrays++;
    if (closest_intersection(sc, ray, hit)) {
        if (!hit.material) return Col<R>(0, 0, 0);
        Scatter<R> s = scatter(*hit.material, ray, hit, g);
        if (s.some) {
            if (s.has_ray) {
                Col<R> scattered_color = trace(s.ray, sc, current_depth + 1, max_depth, g, nest, nrays);
                R prob = (R)0.1;
                if (hit.material->type == DIELECTRIC) prob = (R)0.05;
                Col<R> light_color(0, 0, 0);
                R lights_len = (R)sc.lights.size();
                if (lights_len > (R)0 && (R)g.gen() > ((R)1 - lights_len * prob) && current_depth == (max_depth - 1)) {
                    for (auto& l : sc.lights) {
                        if (l.empty()) continue;   // reference would panic
                        Ray<R> shadow_ray(hit.point, l[0] - hit.point);
                        Col<R> target = (nest < kMaxShadowNest) ? trace(shadow_ray, sc, 0, 1, g, nest + 1, nrays) : Col<R>(0, 0, 0);
                        light_color += s.color * target;
                    }
                    light_color /= lights_len;
                }
                return light_color + (s.color * scattered_color);
            }
            return s.color;
        }
        return Col<R>(0, 0, 0);
    }
    return sc.background;
}

// anti_aliasing.rs:93-102
static void uniform_grid_sampling(size_t res, double bx, double by, std::vector<std::pair<double, double>>& out) {
    double step = 1.0 / (double)res;
    out.clear();
    for (size_t i = 0; i < res; ++i)
        for (size_t j = 0; j < res; ++j) out.push_back({bx + (double)i * step, by + (double)j * step});
}
// anti_aliasing.rs:111-124
static void jitter_sampling(size_t res, double bx, double by, Rng& g, std::vector<std::pair<double, double>>& out) {
    double step = 1.0 / (double)res;
    out.clear();
    for (size_t i = 0; i < res; ++i)
        for (size_t j = 0; j < res; ++j) {
            double jx = g.gen_range(0.0, step);
            double jy = g.gen_range(0.0, step);
            out.push_back({bx + ((double)i * step) + jx, by + ((double)j * step) + jy});
        }
}

static inline uint64_t pixel_seed(uint64_t seed, uint64_t i, uint64_t j) {
    uint64_t x = seed ^ (i * 0x9E3779B97F4A7C15ull) ^ (j * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull);
    return splitmix64(x);
}

// raytrace.rs:126-154.  Rows are produced for j = height-1 .. 0, so output row 0
// is j = height-1.  Samples [s_begin, s_end) of the N*N list are traced (the
// reference always traces all of them; the sub-range exists for the multi-GPU
// sample-split tests).  sum_out = sum of sample colours (before the mean),
// color_out = mean -> sqrt -> clamp, rgb8_out = to_output.
template <class R>
static void render(Scene<R>& sc, int mode, size_t N, int depth, uint64_t seed, int threads, long s_begin, long s_end,
                   double* sum_out, double* color_out, uint8_t* rgb8_out, long cx0 = 0, long cx1 = -1, long crow0 = 0, long crow1 = -1, long cstep = 1) {
    const size_t W = sc.width, H = sc.height;
    // optional crop [cx0,cx1) x [crow0,crow1) in output coordinates, every cstep-th pixel and row
    // (bounded cpu_baseline samples); outputs keep the full-frame layout
    if (cx1 < 0) { cx0 = 0; cx1 = (long)W; crow0 = 0; crow1 = (long)H; }
    if (s_begin < 0) s_begin = 0;
    if (s_end < 0 || s_end > (long)(N * N)) s_end = (long)(N * N);
    std::atomic<size_t> next_row{(size_t)crow0};
    std::atomic<uint64_t> total_rays{0};
    auto worker = [&]() {
        std::vector<std::pair<double, double>> samples;
        uint64_t nrays = 0;
        for (;;) {
            size_t row = next_row.fetch_add((size_t)cstep);
            if (row >= (size_t)crow1) break;
            size_t j = H - 1 - row;
            for (size_t i = (size_t)cx0; i < (size_t)cx1; i += (size_t)cstep) {
                Rng g(pixel_seed(seed, i, j));
                if (mode == 0) uniform_grid_sampling(N, (double)i, (double)j, samples);
                else jitter_sampling(N, (double)i, (double)j, g, samples);
                Col<R> sum(0, 0, 0);
                size_t count = samples.size();
                for (long s = s_begin; s < s_end; ++s) {
                    // one RNG stream per (pixel, sample) so that a sample sub-range
                    // reproduces exactly what the full range computes for it
                    Rng gs(pixel_seed(seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(s + 1)), i, j));
                    Ray<R> ray = spawn_ray(sc, (R)samples[s].first, (R)samples[s].second);
                    sum += trace(ray, sc, 0, depth, gs, 0, nrays);
                }
                size_t o = (row * W + i) * 3;
                if (sum_out) { sum_out[o] = (double)sum.r; sum_out[o + 1] = (double)sum.g; sum_out[o + 2] = (double)sum.b; }
                Col<R> px = sum;
                px /= (R)count;
                px.r = std::sqrt(px.r); px.g = std::sqrt(px.g); px.b = std::sqrt(px.b);
                px.clamp();
                if (color_out) { color_out[o] = (double)px.r; color_out[o + 1] = (double)px.g; color_out[o + 2] = (double)px.b; }
                if (rgb8_out) {
                    rgb8_out[o] = sat_u8(255.999 * (double)px.r);
                    rgb8_out[o + 1] = sat_u8(255.999 * (double)px.g);
                    rgb8_out[o + 2] = sat_u8(255.999 * (double)px.b);
                }
            }
        }
        total_rays += nrays;
    };
    if (threads <= 1) worker();
    else {
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
    sc.rays_traced += total_rays.load();
}

// One primary ray per pixel at sub-pixel offset (dx,dy): the hit-ID gate.
// Optional crop [x0,x1) x [row0,row1) in output coordinates (row 0 = top).
template <class R>
static void primary_hits(Scene<R>& sc, double dx, double dy, long x0, long x1, long row0, long row1, int32_t* obj, int32_t* tri,
                         double* t, double* normal, double* point) {
    const size_t W = sc.width, H = sc.height;
    if (x1 < 0) { x0 = 0; x1 = (long)W; row0 = 0; row1 = (long)H; }
    size_t cw = (size_t)(x1 - x0);
    for (long row = row0; row < row1; ++row) {
        size_t j = H - 1 - (size_t)row;
        for (long i = x0; i < x1; ++i) {
            size_t o = (size_t)(row - row0) * cw + (size_t)(i - x0);
            Ray<R> ray = spawn_ray(sc, (R)((double)i + dx), (R)((double)j + dy));
            Hit<R> h;
            if (closest_intersection(sc, ray, h)) {
                obj[o] = h.object_index; tri[o] = h.triangle_index; t[o] = (double)h.t;
                for (int k = 0; k < 3; ++k) { normal[o * 3 + k] = (double)h.normal[k]; if (point) point[o * 3 + k] = (double)h.point[k]; }
            } else {
                obj[o] = -1; tri[o] = -1; t[o] = -1.0;
                for (int k = 0; k < 3; ++k) { normal[o * 3 + k] = 0.0; if (point) point[o * 3 + k] = 0.0; }
            }
        }
    }
}

}  // namespace orc

// ============================================================================
// C interface (ctypes).  Handles are opaque; `prec` 0 = f64 (the oracle),
// 1 = f32 (diagnostic only).
// ============================================================================
using namespace orc;

struct OrcScene {
    int prec;
    Scene<double> d;
    Scene<float> f;
};

template <class R>
static int add_material_t(Scene<R>& sc, int type, const double* rgb, double param, int tw, int th, const uint8_t* tex) {
    Material<R> m;
    m.type = type;
    m.color = Col<R>((R)rgb[0], (R)rgb[1], (R)rgb[2]);
    m.param = (R)param;
    if (type == TEXTURE && tex && tw > 0 && th > 0) {
        m.tex_w = (R)tw; m.tex_h = (R)th;
        m.pixels.resize((size_t)tw * th);
        // image.rs:45-51 from_output: byte / 255.999
        for (size_t i = 0; i < (size_t)tw * th; ++i)
            m.pixels[i] = Col<R>((R)(tex[i * 3] / 255.999), (R)(tex[i * 3 + 1] / 255.999), (R)(tex[i * 3 + 2] / 255.999));
    }
    sc.materials.push_back(m);
    return (int)sc.materials.size() - 1;
}

template <class R>
static void add_mesh_t(Scene<R>& sc, long nverts, const double* verts, long ntris, const int64_t* vidx, const int64_t* uvidx,
                       long nuv, const double* uvs, const int64_t* tri_mat, long nmats, const int32_t* mats) {
    Mesh<R> m;
    for (long i = 0; i < nverts; ++i) m.vertex_positions.push_back(V3<R>((R)verts[i * 3], (R)verts[i * 3 + 1], (R)verts[i * 3 + 2]));
    for (long i = 0; i < nuv; ++i) m.uvs.push_back({(R)uvs[i * 2], (R)uvs[i * 2 + 1]});
    for (long i = 0; i < nmats; ++i) m.materials.push_back(mats[i]);
    for (long i = 0; i < ntris; ++i) {
        Triangle t;
        for (int k = 0; k < 3; ++k) t.vertex_idx[k] = (size_t)vidx[i * 3 + k];
        if (uvidx) { t.has_uv = true; for (int k = 0; k < 3; ++k) t.uv_idx[k] = (size_t)uvidx[i * 3 + k]; }
        t.material_idx = (size_t)tri_mat[i];
        m.triangles.push_back(t);
    }
    sc.meshes.push_back(std::move(m));
    sc.objects.push_back({OBJ_MESH, (int)sc.meshes.size() - 1});
}

#define BOTH(expr_d, expr_f) do { if (s->prec == 0) { expr_d; } else { expr_f; } } while (0)

extern "C" {

OrcScene* orc_scene_new(int prec, long width, long height, const double* bg, const double* eye, const double* look_at,
                        const double* up, double fovy) {
    OrcScene* s = new OrcScene();
    s->prec = prec;
    s->d.width = width; s->d.height = height; s->f.width = width; s->f.height = height;
    s->d.background = Col<double>(bg[0], bg[1], bg[2]);
    s->f.background = Col<float>((float)bg[0], (float)bg[1], (float)bg[2]);
    s->d.eye = V3<double>(eye[0], eye[1], eye[2]); s->f.eye = V3<float>((float)eye[0], (float)eye[1], (float)eye[2]);
    s->d.look_at = V3<double>(look_at[0], look_at[1], look_at[2]);
    s->f.look_at = V3<float>((float)look_at[0], (float)look_at[1], (float)look_at[2]);
    s->d.up = V3<double>(up[0], up[1], up[2]); s->f.up = V3<float>((float)up[0], (float)up[1], (float)up[2]);
    s->d.fovy = fovy; s->f.fovy = (float)fovy;
    return s;
}
void orc_scene_free(OrcScene* s) { delete s; }

int orc_add_material(OrcScene* s, int type, const double* rgb, double param, int tw, int th, const uint8_t* tex) {
    if (s->prec == 0) return add_material_t(s->d, type, rgb, param, tw, th, tex);
    return add_material_t(s->f, type, rgb, param, tw, th, tex);
}
void orc_add_sphere(OrcScene* s, const double* c, double r, int mat) {
    if (s->prec == 0) { s->d.spheres.push_back({V3<double>(c[0], c[1], c[2]), r, mat}); s->d.objects.push_back({OBJ_SPHERE, (int)s->d.spheres.size() - 1}); }
    else { s->f.spheres.push_back({V3<float>((float)c[0], (float)c[1], (float)c[2]), (float)r, mat}); s->f.objects.push_back({OBJ_SPHERE, (int)s->f.spheres.size() - 1}); }
}
void orc_add_plane(OrcScene* s, const double* c, const double* n, int mat) {
    if (s->prec == 0) { s->d.planes.push_back({V3<double>(c[0], c[1], c[2]), V3<double>(n[0], n[1], n[2]), mat}); s->d.objects.push_back({OBJ_PLANE, (int)s->d.planes.size() - 1}); }
    else { s->f.planes.push_back({V3<float>((float)c[0], (float)c[1], (float)c[2]), V3<float>((float)n[0], (float)n[1], (float)n[2]), mat}); s->f.objects.push_back({OBJ_PLANE, (int)s->f.planes.size() - 1}); }
}
void orc_add_mesh(OrcScene* s, long nverts, const double* verts, long ntris, const int64_t* vidx, const int64_t* uvidx, long nuv,
                  const double* uvs, const int64_t* tri_mat, long nmats, const int32_t* mats) {
    BOTH(add_mesh_t(s->d, nverts, verts, ntris, vidx, uvidx, nuv, uvs, tri_mat, nmats, mats),
         add_mesh_t(s->f, nverts, verts, ntris, vidx, uvidx, nuv, uvs, tri_mat, nmats, mats));
}
// Test infrastructure (file header): conservative cull in front of the triangle loop of every mesh, same results.
void orc_enable_accel(OrcScene* s, int on) {
    if (s->prec == 0) { for (auto& m : s->d.meshes) { if (on) build_accel(m); else { m.accel_nodes.clear(); m.accel_order.clear(); m.accel_always.clear(); } } }
    else { for (auto& m : s->f.meshes) { if (on) build_accel(m); else { m.accel_nodes.clear(); m.accel_order.clear(); m.accel_always.clear(); } } }
}
// Camera::new + Scene::precompute (raytrace.rs:117-125)
void orc_prepare(OrcScene* s) { BOTH((camera_new(s->d), precompute(s->d)), (camera_new(s->f), precompute(s->f))); }

int orc_num_lights(OrcScene* s) { return s->prec == 0 ? (int)s->d.lights.size() : (int)s->f.lights.size(); }
int orc_light_points(OrcScene* s, int li, double* out, int max_points) {
    int n = 0;
    if (s->prec == 0) { for (auto& p : s->d.lights[li]) { if (n < max_points) { out[n * 3] = p[0]; out[n * 3 + 1] = p[1]; out[n * 3 + 2] = p[2]; } n++; } }
    else { for (auto& p : s->f.lights[li]) { if (n < max_points) { out[n * 3] = p[0]; out[n * 3 + 1] = p[1]; out[n * 3 + 2] = p[2]; } n++; } }
    return n;
}
void orc_camera_get(OrcScene* s, double* eye, double* horizontal, double* vertical, double* lower_left) {
    for (int k = 0; k < 3; ++k) {
        if (s->prec == 0) { eye[k] = s->d.cam_eye[k]; horizontal[k] = s->d.cam_horizontal[k]; vertical[k] = s->d.cam_vertical[k]; lower_left[k] = s->d.cam_lower_left[k]; }
        else { eye[k] = s->f.cam_eye[k]; horizontal[k] = s->f.cam_horizontal[k]; vertical[k] = s->f.cam_vertical[k]; lower_left[k] = s->f.cam_lower_left[k]; }
    }
}
void orc_spawn_ray(OrcScene* s, double x, double y, double* o, double* d) {
    if (s->prec == 0) { auto r = spawn_ray(s->d, x, y); for (int k = 0; k < 3; ++k) { o[k] = r.origin[k]; d[k] = r.direction[k]; } }
    else { auto r = spawn_ray(s->f, (float)x, (float)y); for (int k = 0; k < 3; ++k) { o[k] = r.origin[k]; d[k] = r.direction[k]; } }
}
void orc_primary_hits(OrcScene* s, double dx, double dy, long x0, long x1, long row0, long row1, int32_t* obj, int32_t* tri, double* t,
                      double* normal, double* point) {
    BOTH(primary_hits(s->d, dx, dy, x0, x1, row0, row1, obj, tri, t, normal, point),
         primary_hits(s->f, dx, dy, x0, x1, row0, row1, obj, tri, t, normal, point));
}
// Closest hit of an arbitrary ray (direction is normalised by Ray::new).
int orc_closest_hit(OrcScene* s, const double* o, const double* d, int32_t* obj, int32_t* tri, double* t, double* point, double* normal,
                    double* uv) {
    if (s->prec == 0) {
        Ray<double> r(V3<double>(o[0], o[1], o[2]), V3<double>(d[0], d[1], d[2]));
        Hit<double> h;
        if (!closest_intersection(s->d, r, h)) return 0;
        *obj = h.object_index; *tri = h.triangle_index; *t = h.t;
        for (int k = 0; k < 3; ++k) { point[k] = h.point[k]; normal[k] = h.normal[k]; }
        uv[0] = h.has_uv ? h.u : -1; uv[1] = h.has_uv ? h.v : -1;
        return 1;
    }
    Ray<float> r(V3<float>((float)o[0], (float)o[1], (float)o[2]), V3<float>((float)d[0], (float)d[1], (float)d[2]));
    Hit<float> h;
    if (!closest_intersection(s->f, r, h)) return 0;
    *obj = h.object_index; *tri = h.triangle_index; *t = h.t;
    for (int k = 0; k < 3; ++k) { point[k] = h.point[k]; normal[k] = h.normal[k]; }
    uv[0] = h.has_uv ? h.u : -1; uv[1] = h.has_uv ? h.v : -1;
    return 1;
}
// Closest hits of n arbitrary rays (f64 oracle only; directions are normalised by Ray::new, raytrace.rs:32-37),
// threaded over the rays.  Misses: obj = tri = -1, t = -1.  The GPU parity tests feed it the device's own
// f32 rays (rt_primary_rays), so that hit distance and normal can be held to 1e-5 on every hit.
void orc_closest_hits(OrcScene* s, long n, const double* origins, const double* dirs, int threads, int32_t* obj, int32_t* tri, double* t,
                      double* normal, double* point) {
    std::atomic<long> next{0};
    auto worker = [&]() {
        for (;;) {
            const long b = next.fetch_add(256);
            if (b >= n) break;
            for (long i = b; i < std::min(n, b + 256); ++i) {
                Ray<double> r(V3<double>(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), V3<double>(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
                Hit<double> h;
                if (closest_intersection(s->d, r, h)) {
                    obj[i] = h.object_index; tri[i] = h.triangle_index; t[i] = h.t;
                    for (int k = 0; k < 3; ++k) { normal[3 * i + k] = h.normal[k]; point[3 * i + k] = h.point[k]; }
                } else {
                    obj[i] = -1; tri[i] = -1; t[i] = -1.0;
                    for (int k = 0; k < 3; ++k) { normal[3 * i + k] = 0.0; point[3 * i + k] = 0.0; }
                }
            }
        }
    };
    if (threads <= 1) worker();
    else {
        std::vector<std::thread> ts;
        for (int k = 0; k < threads; ++k) ts.emplace_back(worker);
        for (auto& th : ts) th.join();
    }
}
// mode 0 = Uniform(N), 1 = Jitter(N).  Returns rays traced (closest-hit queries).
uint64_t orc_render(OrcScene* s, int mode, long N, int depth, uint64_t seed, int threads, long s_begin, long s_end, double* sum_out,
                    double* color_out, uint8_t* rgb8_out) {
    if (s->prec == 0) { s->d.rays_traced = 0; render(s->d, mode, (size_t)N, depth, seed, threads, s_begin, s_end, sum_out, color_out, rgb8_out); return s->d.rays_traced.load(); }
    s->f.rays_traced = 0; render(s->f, mode, (size_t)N, depth, seed, threads, s_begin, s_end, sum_out, color_out, rgb8_out); return s->f.rays_traced.load();
}
uint64_t orc_render_crop(OrcScene* s, int mode, long N, int depth, uint64_t seed, int threads, long x0, long x1, long row0, long row1,
                         long step, double* sum_out) {
    if (step < 1) step = 1;
    if (s->prec == 0) { s->d.rays_traced = 0; render(s->d, mode, (size_t)N, depth, seed, threads, 0, -1, sum_out, nullptr, nullptr, x0, x1, row0, row1, step); return s->d.rays_traced.load(); }
    s->f.rays_traced = 0; render(s->f, mode, (size_t)N, depth, seed, threads, 0, -1, sum_out, nullptr, nullptr, x0, x1, row0, row1, step); return s->f.rays_traced.load();
}
uint64_t orc_texture_oob_count() { return g_tex_oob.load(); }

// ---- unit-level entry points for the known-answer tests --------------------
void orc_vec_dot_cross(const double* a, const double* b, double* dot, double* cross) {
    V3<double> A(a[0], a[1], a[2]), B(b[0], b[1], b[2]);
    *dot = A.dot(B);
    V3<double> c = A.cross(B);
    for (int k = 0; k < 3; ++k) cross[k] = c[k];
}
void orc_vec_normalized(const double* a, double* out) { V3<double> n = V3<double>(a[0], a[1], a[2]).normalized(); for (int k = 0; k < 3; ++k) out[k] = n[k]; }
void orc_vec_reflect(const double* a, const double* n, double* out) { V3<double> r = V3<double>(a[0], a[1], a[2]).reflect(V3<double>(n[0], n[1], n[2])); for (int k = 0; k < 3; ++k) out[k] = r[k]; }
int orc_vec_near_zero(const double* a) { return V3<double>(a[0], a[1], a[2]).near_zero() ? 1 : 0; }
void orc_random_unit_vector(uint64_t seed, double* out) { Rng g(seed); V3<double> v = V3<double>::random_unit_vector(g); for (int k = 0; k < 3; ++k) out[k] = v[k]; }
void orc_random_in_unit_sphere(uint64_t seed, double* out) { Rng g(seed); V3<double> v = V3<double>::random_in_unit_sphere(g); for (int k = 0; k < 3; ++k) out[k] = v[k]; }

// SuperSampling::from_str, anti_aliasing.rs:50-83.  kind: 0 Uniform, 1 Jitter.
// NB "uniform:N" yields Jitter(N) (anti_aliasing.rs:75).
int orc_ssaa_parse(const char* str, int* kind, uint64_t* res, char* err, int errlen) {
    std::string s(str);
    std::vector<std::string> parts;
    size_t pos = 0;
    for (;;) { size_t q = s.find(':', pos); if (q == std::string::npos) { parts.push_back(s.substr(pos)); break; } parts.push_back(s.substr(pos, q - pos)); pos = q + 1; }
    auto fail = [&](const char* m) { snprintf(err, errlen, "%s", m); return 1; };
    auto parse_usize = [](const std::string& t, uint64_t* out) {
        // Rust usize::from_str: optional '+', then 1+ ASCII digits, no whitespace
        size_t i = 0;
        if (i < t.size() && t[i] == '+') i++;
        if (i >= t.size()) return false;
        uint64_t v = 0;
        for (; i < t.size(); ++i) { if (t[i] < '0' || t[i] > '9') return false; uint64_t nv = v * 10 + (uint64_t)(t[i] - '0'); if (nv < v) return false; v = nv; }
        *out = v;
        return true;
    };
    if (parts[0] == "jitter") {
        if (parts.size() < 2) return fail("no arguments supplied, need resolution for jitter");
        if (!parse_usize(parts[1], res)) return fail("resolution has to be an integer");
        *kind = 1; return 0;
    }
    if (parts[0] == "uniform") {
        if (parts.size() < 2) return fail("no arguments supplied, need resolution for uniform");
        if (!parse_usize(parts[1], res)) return fail("resolution has to be an integer");
        *kind = 1; return 0;
    }
    return fail("unknown method");
}
// SuperSampling::sample, anti_aliasing.rs:12-19.  out_xy holds 2*N*N doubles.
void orc_ssaa_sample(int kind, long N, long x, long y, uint64_t seed, double* out_xy) {
    std::vector<std::pair<double, double>> v;
    if (kind == 0) uniform_grid_sampling((size_t)N, (double)x, (double)y, v);
    else { Rng g(seed); jitter_sampling((size_t)N, (double)x, (double)y, g, v); }
    for (size_t i = 0; i < v.size(); ++i) { out_xy[i * 2] = v[i].first; out_xy[i * 2 + 1] = v[i].second; }
}
int orc_aabb_intersect(const double* mn, const double* mx, const double* o, const double* d) {
    AABB<double> bb{V3<double>(mn[0], mn[1], mn[2]), V3<double>(mx[0], mx[1], mx[2])};
    Ray<double> r(V3<double>(o[0], o[1], o[2]), V3<double>(d[0], d[1], d[2]));
    return aabb_intersect(bb, r) ? 1 : 0;
}
void orc_refract(const double* d, const double* n, double ratio, double* out) {
    V3<double> r = refract(V3<double>(d[0], d[1], d[2]), V3<double>(n[0], n[1], n[2]), ratio);
    for (int k = 0; k < 3; ++k) out[k] = r[k];
}
double orc_reflectance(double cosine, double ref_idx) { return reflectance(cosine, ref_idx); }
// Material::scatter on a hand-made IntersectionInfo (materials.rs tests :290-378).
// returns: 0 = None, 1 = Some((None, c)), 2 = Some((Some(ray), c))
int orc_scatter(int type, const double* rgb, double param, const double* ro, const double* rd, const double* point, const double* normal,
                uint64_t seed, double* out_dir, double* out_color) {
    Material<double> m; m.type = type; m.color = Col<double>(rgb[0], rgb[1], rgb[2]); m.param = param;
    Ray<double> ray(V3<double>(ro[0], ro[1], ro[2]), V3<double>(rd[0], rd[1], rd[2]));
    Hit<double> h; h.point = V3<double>(point[0], point[1], point[2]); h.normal = V3<double>(normal[0], normal[1], normal[2]); h.material = &m;
    Rng g(seed);
    Scatter<double> s = scatter(m, ray, h, g);
    if (!s.some) return 0;
    out_color[0] = s.color.r; out_color[1] = s.color.g; out_color[2] = s.color.b;
    if (!s.has_ray) return 1;
    for (int k = 0; k < 3; ++k) out_dir[k] = s.ray.direction[k];
    return 2;
}
void orc_to_output(const double* c, uint8_t* out) { for (int k = 0; k < 3; ++k) out[k] = sat_u8(255.999 * c[k]); }
void orc_texture_albedo(int tw, int th, const uint8_t* tex, double u, double v, double* out) {
    Scene<double> sc; double rgb[3] = {0, 0, 0};
    add_material_t(sc, TEXTURE, rgb, 0, tw, th, tex);
    Col<double> c = get_albedo(sc.materials[0], u, v);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}

}  // extern "C"
