#!/usr/bin/env python3
"""Noise of the device's own frames on a soak room: PSNR between two device renders (4096 spp, different seeds) beside the
PSNR between two oracle renders of the same sample count.  python tools/room_noise.py <seed>..."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "tools", "soak_images.py")).read().replace("\nmain()\n", "\n")
ns = {"__file__": os.path.join(ROOT, "tools", "soak_images.py")}
exec(compile(src, "soak_images", "exec"), ns)
rt, O, W, H, CAM = ns["rt"], ns["O"], ns["W"], ns["H"], ns["CAM"]
ctx = rt.Context(0)
g = lambda x: np.sqrt(np.minimum(np.maximum(x, 0.0), 1.0))
psnr = lambda a, b: 10 * np.log10(1.0 / max(np.mean((g(a) - g(b)) ** 2), 1e-12))
for seed in map(int, sys.argv[1:]):
    objs, bg, depth = ns["room"](seed)
    ctx.upload(rt.Scene.from_objects(W, H, bg, CAM, objs))
    osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=CAM, objects=[
        (dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(len(o["tris"]), np.int64))) if o["type"] == "Mesh" else o) for o in objs]))
    D = [ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=seed + 17 * k))[0].astype(np.float64) / 4096.0 for k in range(2)]
    A = [np.mean([osc.render(mode="jitter", N=16, depth=depth, seed=5000 * seed + 16 * h + k)["sum"] / 256.0 for k in range(16)], axis=0) for h in range(2)]
    print("seed %d depth %d: device-device %.1f dB, oracle-oracle %.1f dB, device-oracle %.1f / %.1f dB; means device %.5f %.5f oracle %.5f %.5f" %
          (seed, depth, psnr(D[0], D[1]), psnr(A[0], A[1]), psnr(D[0], A[0]), psnr(D[1], A[1]), D[0].mean(), D[1].mean(), A[0].mean(), A[1].mean()), flush=True)
