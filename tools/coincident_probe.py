#!/usr/bin/env python3
"""Two boxes whose front faces coincide (z = 0.25, the second box wound inwards): device against oracle on the pixels of the
first, per depth, with the device's ray counters.  python tools/coincident_probe.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
rt = importlib.import_module("raytracer-rust_amd")
W, H = 48, 36
QUADS = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
def box(lo, hi, flip):
    c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
    v = np.array([c[i] for q in QUADS for i in (q[0], q[1], q[2], q[0], q[2], q[3])])
    return v.reshape(-1, 3, 3)[:, [0, 2, 1]].reshape(-1, 3) if flip else v
cam = dict(eye=[1.875, 1.875, 5.625], look_at=[0.375, 0.375, -0.375], up=[0.0, 1.0, 0.0], fovy=40.0)
variant = sys.argv[1] if len(sys.argv) > 1 else "both"
objs = [dict(type="Mesh", verts=box([-1.625, 0.5, -0.25], [-0.625, 1.375, 0.25], False), tris=np.arange(36).reshape(12, 3), materials=[dict(type="Metal", albedo=[0.8, 0.7, 0.6], fuzziness=0.3)])]
if variant != "one":
    objs.append(dict(type="Mesh", verts=box([-1.875, 0.375, -1.25], [-0.75, 1.5, 0.25], variant != "noflip"), tris=np.arange(36).reshape(12, 3), materials=[dict(type="Lambertian", albedo=[0.5, 0.6, 0.7])]))
bg = [0.7, 0.8, 0.9]
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_objects(W, H, bg, cam, objs))
osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=cam, objects=[dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(12, np.int64))) for o in objs]))
ids = osc.primary_hits(0.5, 0.5)["object_id"]
for oi in range(len(objs)):
    m = ids == oi
    for depth in (2, 3, 4):
        A = np.mean([osc.render(mode="jitter", N=16, depth=depth, seed=100 + k)["sum"][m].mean() / 256.0 for k in range(32)])
        D = np.mean([ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=5 + k))[0].astype(np.float64)[m].mean() / 4096.0 for k in range(2)])
        ctx.stats_reset(); ctx.render_image(rt.RenderParams("jitter:8", depth=depth, seed=5, counters=True)); st = ctx.stats()
        print("%s object %d depth %d: %4d px  device %.5f  oracle %.5f  ratio %.4f   rays per iteration %s  self-hit rays %d" % (variant, oi, depth, m.sum(), D, A, D / max(A, 1e-12), st["rays_per_iteration"][:depth], st.get("self_hit_rays", -1)), flush=True)
