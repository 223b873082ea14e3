#!/usr/bin/env python3
"""Why a lone axis-aligned glass box was 1.1 % too bright on the device (numerics): the reference's
dielectric path over the box's 12 triangles restated in numpy f64 (intersections.rs:148-186, materials.rs:105-153,
raytrace.rs:32-42; no BVH, no device), N paths with common random numbers, in five forms that differ only in how the
scattered DIRECTION is produced:
  ref    f64 scatter on the f64 ray                                  (the reference)
  dev    f32 scatter, random bits below the f32 where it enters f64  (the device before the fix)
  dev64  f64 scatter; only the primary directions are f32 + random   (the device now, for flat surfaces)
  qK     f64 scatter, then the K lowest mantissa bits re-randomised  (how many bits must travel? all of them)
Prints the rays alive per iteration; "ended at it" 3 is the third interface.  CPU only: python tools/glass_box_sim.py [N]"""
import numpy as np, sys
f32 = np.float32
lo, hi = np.array([-0.83, -0.21, -1.37]), np.array([0.71, 1.13, 0.29])
c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
vs = []
for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
    vs += [c[q[0]], c[q[1]], c[q[2]], c[q[0]], c[q[2]], c[q[3]]]
V = np.array(vs).reshape(12, 3, 3)
TA, TAB, TAC = V[:, 0], V[:, 1] - V[:, 0], V[:, 2] - V[:, 0]
def normalize(v): return v / np.sqrt((v * v).sum(-1, keepdims=True))
TN = np.array([np.cross(TAB[k], TAC[k]) / np.linalg.norm(np.cross(TAB[k], TAC[k])) for k in range(12)])
IOR = 1.886
def det3(v1, v2, v3):
    return (v1[..., 0]*v2[..., 1]*v3[..., 2] + v2[..., 0]*v3[..., 1]*v1[..., 2] + v3[..., 0]*v1[..., 1]*v2[..., 2]) - \
           (v3[..., 0]*v2[..., 1]*v1[..., 2] + v2[..., 0]*v1[..., 1]*v3[..., 2] + v1[..., 0]*v3[..., 1]*v2[..., 2])
def intersect(o, d):
    """reference mesh loop: first-wins min t over the 12 triangles (no AABB: rays that matter are inside it)"""
    best_t = np.full(len(o), np.inf); best_k = np.full(len(o), -1)
    with np.errstate(all="ignore"):
        for k in range(12):
            res = o - TA[k]
            M = det3(d, TAB[k][None], TAC[k][None]); Mt = det3(res, TAB[k][None], TAC[k][None]); Ma = det3(d, res, TAC[k][None]); Mb = det3(d, TAB[k][None], res)
            a = Ma / M; b = Mb / M; t = -(Mt / M)
            ok = ~((a < 0) | (b < 0) | (a + b > 1) | (t < 0)) & ~np.isnan(t)
            take = ok & ((best_k < 0) | (best_t > t))
            best_t[take] = t[take]; best_k[take] = k
    return best_k, best_t
def dither(x32, rng):
    x = x32.astype(np.float64); bits = x.view(np.uint64)
    return (bits ^ rng.integers(0, 1 << 28, size=x.shape, dtype=np.uint64)).view(np.float64)
def scatter64(d, n, u):
    front = (n * d).sum(-1) <= 0
    ratio = np.where(front, 1.0 / IOR, IOR)
    nn = np.where(front[:, None], n, -n)
    cos = np.minimum((-d * nn).sum(-1), 1.0)
    sin = np.sqrt(1.0 - cos * cos)
    cannot = ratio * sin > 1.0
    cos2 = np.minimum((nn * -d).sum(-1), 1.0)
    perp = (d + nn * cos2[:, None]) * ratio[:, None]
    par = nn * -np.sqrt(np.abs(1.0 - (perp * perp).sum(-1)))[:, None]
    refr = perp + par
    r0 = ((1.0 - ratio) / (1.0 + ratio)) ** 2
    refl_p = r0 + (1.0 - r0) * (1.0 - cos) ** 5
    use_refl = cannot | (refl_p > u)
    refl = d - nn * (2.0 * (d * nn).sum(-1))[:, None]
    out = np.where(use_refl[:, None], refl, refr)
    return normalize(out)
def scatter32(d, n, u):
    d = d.astype(f32); n = n.astype(f32); ior = f32(IOR)
    front = (n * d).sum(-1, dtype=f32) <= 0
    ratio = np.where(front, f32(1.0) / ior, ior).astype(f32)
    nn = np.where(front[:, None], n, -n)
    cos = np.minimum((-d * nn).sum(-1, dtype=f32), f32(1.0))
    sin = np.sqrt(f32(1.0) - cos * cos)
    cannot = ratio * sin > f32(1.0)
    perp = (d + nn * cos[:, None]) * ratio[:, None]
    par = nn * -np.sqrt(np.abs(f32(1.0) - (perp * perp).sum(-1, dtype=f32)))[:, None]
    refr = perp + par
    r0 = ((f32(1.0) - ratio) / (f32(1.0) + ratio)) ** 2
    x = f32(1.0) - cos
    refl_p = r0 + (f32(1.0) - r0) * (x * x * x * x * x)
    use_refl = cannot | (refl_p > u.astype(f32))
    refl = d - nn * (f32(2.0) * (d * nn).sum(-1, dtype=f32))[:, None]
    out = np.where(use_refl[:, None], refl, refr).astype(f32)
    return (out / np.sqrt((out * out).sum(-1, dtype=f32, keepdims=True))).astype(f32)
# camera (camera.rs:26-54)
W, H = 48, 36
eye = np.array([0.0, 1.0, 6.0]); at = np.array([0.0, 0.3, 0.0]); up = np.array([0.0, 1.0, 0.0])
w = normalize(eye - at); u_ = normalize(np.cross(up, w)); v_ = np.cross(w, u_)
hh = np.tan(np.radians(40.0) / 2); vh = 2 * hh; vw = vh * W / H
horiz = u_ * vw / W; vert = v_ * vh / H; ll = eye - u_ * vw / 2 - v_ * vh / 2 - w
R = np.random.default_rng(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
px = R.uniform(0, W, N); py = R.uniform(0, H, N)
d0 = normalize(ll + horiz * px[:, None] + vert * py[:, None] - eye)
U = R.random((8, N))
def run(mode):
    o = np.broadcast_to(eye, (N, 3)).copy(); d = d0.copy()
    d32 = d0.astype(f32)
    if mode != "ref":
        d = dither(d32, np.random.default_rng(100))
    alive = np.ones(N, bool); counts = []; hist = []
    for it in range(7):
        idx = np.nonzero(alive)[0]
        k, t = intersect(o[idx], d[idx])
        hitm = k >= 0
        counts.append(len(idx))
        alive[idx[~hitm]] = False
        idx = idx[hitm]; k = k[hitm]; t = t[hitm]
        p = o[idx] + d[idx] * t[:, None]
        if mode == "ref":
            nd = scatter64(d[idx], TN[k], U[it][idx])
        elif mode == "dev":
            nd32 = scatter32(d32[idx], TN[k], U[it][idx]); d32[idx] = nd32
            nd = dither(nd32, np.random.default_rng(200 + it))
        elif mode == "dev64":      # f64 scatter on the dithered f64 direction, result kept in f64 (what carrying f64 directions would give)
            nd = scatter64(d[idx], TN[k], U[it][idx])
        elif mode.startswith("q"):   # f64 scatter, then keep only (53 - drop) mantissa bits and randomise the rest
            drop = int(mode[1:])
            nd = scatter64(d[idx], TN[k], U[it][idx])
            b = nd.view(np.uint64); mask = np.uint64((1 << drop) - 1)
            nd = ((b & ~mask) | np.random.default_rng(300 + it).integers(0, 1 << drop, size=b.shape, dtype=np.uint64)).view(np.float64)
        o[idx] = p; d[idx] = nd
        hist.append((idx, k, t))
    return counts, hist
for mode in ("ref", "dev", "dev64", "q1", "q12"):
    counts, hist = run(mode)
    print("%-6s rays per iteration %s   ended at it: %s" % (mode, counts, [counts[i] - counts[i + 1] for i in range(1, len(counts) - 1)]))