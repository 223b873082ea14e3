#!/usr/bin/env python3
"""One axis-aligned box (12 triangles, coordinates that are no round numbers) under a bright background: device against oracle
on the pixels that see the box, per material and depth.  python tools/box_probe.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
rt = importlib.import_module("raytracer-rust_amd")
W, H = 48, 36
CAM = dict(eye=[0.0, 1.0, 6.0], look_at=[0.0, 0.3, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)
ctx = rt.Context(0)
lo, hi = np.array([-0.83, -0.21, -1.37]), np.array([0.71, 1.13, 0.29])
c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
vs = []
for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
    vs += [c[q[0]], c[q[1]], c[q[2]], c[q[0]], c[q[2]], c[q[3]]]
verts = np.array(vs)
if os.environ.get("BOX_ROTATE"):      # the same box turned out of the axes
    a = 0.37
    Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.array([[1, 0, 0], [0, np.cos(0.21), -np.sin(0.21)], [0, np.sin(0.21), np.cos(0.21)]])
    verts = verts @ Rm.T
bg = [0.9, 0.65, 0.95]
mats = dict(lambertian=dict(type="Lambertian", albedo=[0.7, 0.6, 0.5]), mirror=dict(type="Metal", albedo=[0.9, 0.9, 0.9], fuzziness=0.0),
            metal=dict(type="Metal", albedo=[0.9, 0.9, 0.9], fuzziness=0.3), glass=dict(type="Dieletrics", tint=[0.78, 0.77, 0.8], refraction_index=1.886),
            glass15=dict(type="Dieletrics", tint=[1.0, 1.0, 1.0], refraction_index=1.5))
for name in sys.argv[1:] or list(mats):
    for depth in (2, 3, 4, 6):
        objs = [dict(type="Mesh", verts=verts, tris=np.arange(36).reshape(12, 3), materials=[mats[name]])]
        ctx.upload(rt.Scene.from_objects(W, H, bg, CAM, objs))
        osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=CAM, objects=[dict(type="Mesh", materials=[mats[name]], mesh=dict(verts_np=verts, tris_np=objs[0]["tris"], tri_mat_np=np.zeros(12, np.int64)))]))
        A = sum(osc.render(mode="jitter", N=16, depth=depth, seed=100 + k)["sum"] for k in range(16)) / 4096.0
        acc, _ = ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=5))
        D = acc.astype(np.float64) / 4096.0
        m = osc.primary_hits(0.5, 0.5)["object_id"] == 0
        ctx.stats_reset(); ctx.render_image(rt.RenderParams("jitter:8", depth=depth, seed=5, counters=True)); st = ctx.stats()
        print("%-10s depth %d: %4d px  device %.5f  oracle %.5f  ratio %.4f   rays per iteration %s  self-hit rays %d" %
              (name, depth, m.sum(), D[m, 0].mean(), A[m, 0].mean(), D[m, 0].mean() / A[m, 0].mean(), st["rays_per_iteration"][:depth], st.get("self_hit_rays", -1)), flush=True)
