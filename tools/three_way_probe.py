#!/usr/bin/env python3
"""Three surfaces at one distance: two boxes that overlap and both stand exactly on the ground plane (their bottom faces and the
plane coincide where they overlap), seen from above; device against oracle per object and depth.  python tools/three_way_probe.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
rt = importlib.import_module("raytracer-rust_amd")
W, H = 48, 36
QUADS = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
def box(lo, hi):
    c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
    return np.array([c[i] for q in QUADS for i in (q[0], q[1], q[2], q[0], q[2], q[3])])
gy = -0.4371
cam = dict(eye=[0.3, 3.1, 4.9], look_at=[0.0, -0.2, -0.3], up=[0.0, 1.0, 0.0], fovy=35.0)
glass = lambda ior, tint: dict(type="Dieletrics", tint=tint, refraction_index=ior)
objs = [dict(type="Mesh", verts=box([-1.31, gy, -1.07], [0.42, 0.38, 0.51]), tris=np.arange(36).reshape(12, 3), materials=[glass(1.31, [0.9, 0.8, 0.7])]),
        dict(type="Mesh", verts=box([-0.37, gy, -0.63], [1.23, 0.71, 0.93]), tris=np.arange(36).reshape(12, 3), materials=[glass(1.52, [0.7, 0.9, 0.8])]),
        dict(type="Plane", center=[0.0, gy, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Lambertian", albedo=[0.6, 0.6, 0.6]))]
bg = [0.7, 0.8, 0.9]
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_objects(W, H, bg, cam, objs))
osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=cam, objects=[(dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(12, np.int64))) if o["type"] == "Mesh" else o) for o in objs]))
ids = osc.primary_hits(0.5, 0.5)["object_id"]
for depth in (3, 5, 8):
    A = [osc.render(mode="jitter", N=16, depth=depth, seed=100 + k)["sum"] / 256.0 for k in range(64)]
    D = [ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=5 + k))[0].astype(np.float64) / 4096.0 for k in range(4)]
    for oi in range(3):
        m = ids == oi
        a = np.array([x[m].mean() for x in A]); d = np.array([x[m].mean() for x in D])
        se = a.std(ddof=1) * np.sqrt(1 / 64 + 1 / 64)
        print("depth %d object %d: %4d px  device %.5f  oracle %.5f  ratio %.4f  z %+.1f" % (depth, oi, m.sum(), d.mean(), a.mean(), d.mean() / a.mean(), (d.mean() - a.mean()) / se), flush=True)
