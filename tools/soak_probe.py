#!/usr/bin/env python3
"""Where does a scene of tools/soak_images.py differ?  python tools/soak_probe.py <seed> [depth [i,j,... objects to drop]]: mean radiance of device and
oracle (4096 spp) per object seen by the primary ray."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "tools", "soak_images.py")).read().replace("\nmain()\n", "\n")
ns = {"__file__": os.path.join(ROOT, "tools", "soak_images.py")}
exec(compile(src, "soak_images", "exec"), ns)
rt, O, W, H, CAM = ns["rt"], ns["O"], ns["W"], ns["H"], ns["CAM"]
seed = int(sys.argv[1])
SEED0 = int(os.environ.get("PROBE_SEED0", "100"))     # other oracle / device random streams
objs, bg, depth = (ns["room"] if os.environ.get("SOAK_ROOM") else ns["scene"])(seed)
if len(sys.argv) > 2:
    depth = int(sys.argv[2])
drop = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []      # object indices to leave out
objs = [o for i, o in enumerate(objs) if i not in drop]
if os.environ.get("PROBE_DUMMY_TRI"):      # a tiny triangle nobody sees: switches the kernels to their with-triangles variants
    objs.append(dict(type="Mesh", verts=np.array([[50.0, 50.0, 50.0], [50.01, 50.0, 50.0], [50.0, 50.01, 50.0]]), tris=np.array([[0, 1, 2]]),
                     materials=[dict(type="Lambertian", albedo=[0.5, 0.5, 0.5])]))
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_objects(W, H, bg, CAM, objs))
osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=CAM, objects=[
    (dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(len(o["tris"]), np.int64)))
     if o["type"] == "Mesh" else o) for o in objs]))
A = sum(osc.render(mode="jitter", N=16, depth=depth, seed=SEED0 + k)["sum"] for k in range(16)) / 4096.0
acc, _ = ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=seed + SEED0))
D = acc.astype(np.float64) / 4096.0
ids = osc.primary_hits(0.5, 0.5)["object_id"]
print("seed %d depth %d: frame mean device %.5f oracle %.5f" % (seed, depth, D.mean(), A.mean()))
for i in np.unique(ids):
    m = ids == i
    name = "miss" if i < 0 else "%d %s %s" % (i, objs[i]["type"], (objs[i].get("material") or objs[i]["materials"][0])["type"])
    print("  %-32s %5d px   device %.5f   oracle %.5f   ratio %.4f" % (name, m.sum(), D[m].mean(), A[m].mean(), D[m].mean() / max(A[m].mean(), 1e-12)))
