#!/usr/bin/env python3
"""Three coincident faces: the front faces of three boxes lie in one plane (z = 0.25) and overlap (the scene of
tests/test_gpu_parity.py::test_three_coincident_faces); device against oracle per object and depth, with error bars
(4 device renders of 4096 spp, 64 oracle renders of 256 spp).  python tools/three_faces_probe.py [two]   (two: boxes a and b only)"""
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
flip = lambda v: v.reshape(-1, 3, 3)[:, [0, 2, 1]].reshape(-1, 3)
mesh = lambda v, m: dict(type="Mesh", verts=v, tris=np.arange(36).reshape(12, 3), materials=[m])
cam = dict(eye=[1.875, 1.875, 5.625], look_at=[0.375, 0.375, -0.375], up=[0.0, 1.0, 0.0], fovy=40.0)
objs = [mesh(box([-1.625, 0.5, -0.25], [-0.625, 1.375, 0.25]), dict(type="Metal", albedo=[0.8, 0.7, 0.6], fuzziness=0.3)),
        mesh(flip(box([-1.875, 0.375, -1.25], [-0.75, 1.5, 0.25])), dict(type="Lambertian", albedo=[0.5, 0.6, 0.7])),
        mesh(box([-1.25, 0.125, -0.75], [0.25, 1.125, 0.25]), dict(type="Lambertian", albedo=[0.7, 0.4, 0.3]))]
if len(sys.argv) > 1 and sys.argv[1] == "two":
    objs = objs[:2]
bg = [0.7, 0.8, 0.9]
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_objects(W, H, bg, cam, objs))
osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=cam, objects=[dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(12, np.int64))) for o in objs]))
ids = osc.primary_hits(0.5, 0.5)["object_id"]
for depth in (1, 2, 4):
    A = [osc.render(mode="jitter", N=16, depth=depth, seed=100 + k)["sum"] / 256.0 for k in range(64)]
    D = [ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=5 + k))[0].astype(np.float64) / 4096.0 for k in range(4)]
    for oi in range(len(objs)):
        m = ids == oi
        a = np.array([x[m].mean() for x in A]); d = np.array([x[m].mean() for x in D])
        se = a.std(ddof=1) * np.sqrt(1 / 64 + 1 / 64)
        print("depth %d object %d: %4d px  device %.5f  oracle %.5f  ratio %.4f +- %.4f  z %+.1f" % (depth, oi, m.sum(), d.mean(), a.mean(), d.mean() / a.mean(), se / a.mean(), (d.mean() - a.mean()) / se), flush=True)
