#!/usr/bin/env python3
"""Self-intersection rate of rays that leave a triangle (intersections.rs:166 accepts t >= 0), device against oracle, for
PRIMARY hits (eye -> triangle) and for SECONDARY hits (eye -> perfect mirror plane -> triangle).  One Lambertian triangle of
albedo 0.5 under a white background, depth = the bounce after the triangle: a pixel is 0.5 * (1 - p_self) where the path
reaches the triangle.  python tools/selfhit_probe.py <first seed> <last seed>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
rt = importlib.import_module("raytracer-rust_amd")
W, H = 32, 24
CAM = dict(eye=[0.0, 1.0, 6.0], look_at=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)
ctx = rt.Context(0)


def run(objs, depth, seed):
    bg = [1.0, 1.0, 1.0]
    ctx.upload(rt.Scene.from_objects(W, H, bg, CAM, objs))
    osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=CAM, objects=[
        (dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(len(o["tris"]), np.int64)))
         if o["type"] == "Mesh" else o) for o in objs]))
    A = sum(osc.render(mode="jitter", N=16, depth=depth, seed=10 * seed + k)["sum"] for k in range(4)) / 1024.0
    acc, _ = ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=seed))
    D = acc.astype(np.float64) / 4096.0
    m = (A[..., 0] > 1e-6) & (A[..., 0] < 0.999)          # pixels whose paths reach the triangle
    return D[m, 0].mean(), A[m, 0].mean(), int(m.sum())


for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(seed)
    c = np.array([rng.uniform(-1.0, 1.0), rng.uniform(0.6, 1.6), rng.uniform(-1.0, 1.0)])
    verts = c + rng.uniform(0.4, 1.5) * rng.normal(size=(3, 3)) * np.array([1.0, 0.4, 1.0])
    tri = dict(type="Mesh", verts=verts, tris=np.array([[0, 1, 2]]), materials=[dict(type="Lambertian", albedo=[0.5, 0.5, 0.5])])
    tri_flipped = dict(tri, tris=np.array([[0, 2, 1]]))
    mirror = dict(type="Plane", center=[0.0, -0.6, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Metal", albedo=[1.0, 1.0, 1.0], fuzziness=0.0))
    out = []
    for name, objs, depth in (("primary", [tri], 2), ("primary flipped", [tri_flipped], 2), ("via mirror", [mirror, tri], 3), ("via mirror flipped", [mirror, tri_flipped], 3)):
        d, a, n = run(objs, depth, seed)
        out.append("%s: dev %.4f orc %.4f (p_self %.3f / %.3f, %d px)" % (name, d, a, 1 - d / 0.5, 1 - a / 0.5, n))
    print("seed %d | " % seed + " | ".join(out), flush=True)
