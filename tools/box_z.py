#!/usr/bin/env python3
"""tools/box_probe.py with error bars: the mean radiance of the pixels that see the box, device (K renders of 4096 spp, seeds
differ) against oracle (K x 16 renders of 256 spp), standard errors from the spread of the renders.
python tools/box_z.py <material> <depth> [K]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
rt = importlib.import_module("raytracer-rust_amd")
W, H = 48, 36
CAM = dict(eye=[0.0, 1.0, 6.0], look_at=[0.0, 0.3, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)
lo, hi = np.array([-0.83, -0.21, -1.37]), np.array([0.71, 1.13, 0.29])
c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
verts = np.array([c[i] for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)] for i in (q[0], q[1], q[2], q[0], q[2], q[3])])
if os.environ.get("BOX_ROTATE"):
    a = 0.37
    verts = verts @ (np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.array([[1, 0, 0], [0, np.cos(0.21), -np.sin(0.21)], [0, np.sin(0.21), np.cos(0.21)]])).T
bg = [0.9, 0.65, 0.95]
mats = dict(lambertian=dict(type="Lambertian", albedo=[0.7, 0.6, 0.5]), mirror=dict(type="Metal", albedo=[0.9, 0.9, 0.9], fuzziness=0.0),
            metal=dict(type="Metal", albedo=[0.9, 0.9, 0.9], fuzziness=0.3), glass=dict(type="Dieletrics", tint=[0.78, 0.77, 0.8], refraction_index=1.886),
            glass15=dict(type="Dieletrics", tint=[1.0, 1.0, 1.0], refraction_index=1.5))
name, depth = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = rt.Context(0)
tris = np.arange(36).reshape(12, 3)
ctx.upload(rt.Scene.from_objects(W, H, bg, CAM, [dict(type="Mesh", verts=verts, tris=tris, materials=[mats[name]])]))
osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=CAM, objects=[dict(type="Mesh", materials=[mats[name]], mesh=dict(verts_np=verts, tris_np=tris, tri_mat_np=np.zeros(12, np.int64)))]))
m = osc.primary_hits(0.5, 0.5)["object_id"] == 0
dev = [ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=1000 + k))[0].astype(np.float64)[m, 0].mean() / 4096.0 for k in range(K)]
ora = [osc.render(mode="jitter", N=16, depth=depth, seed=7000 + k)["sum"][m, 0].mean() / 256.0 for k in range(16 * K)]
d, o = np.mean(dev), np.mean(ora)
sd, so = np.std(dev, ddof=1) / np.sqrt(K), np.std(ora, ddof=1) / np.sqrt(16 * K)
print("%-10s depth %d: device %.5f +- %.5f  oracle %.5f +- %.5f  ratio %.4f +- %.4f  z %+.2f" % (name, depth, d, sd, o, so, d / o, np.hypot(sd, so) / o, (d - o) / np.hypot(sd, so)), flush=True)
