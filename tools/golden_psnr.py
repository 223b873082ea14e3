#!/usr/bin/env python3
"""PSNR of converged device renders of the reference's two scenes against the reference's own PNGs (tests/golden/scenes/*/*.png,
render settings unknown), over depth: which settings are the goldens consistent with, and how far does their own noise allow a pin?"""
import importlib, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-rust_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
ctx = rt.Context(0)
for name in ("spheres", "cube"):
    scene = rt.Scene.load(os.path.join(GOLDEN, "scenes", name, name + ".yaml"), base_dir=GOLDEN)
    ctx.upload(scene)
    gold = np.asarray(Image.open(os.path.join(GOLDEN, "scenes", name, name + ".png")).convert("RGB")).astype(float)
    for depth, n in [(d, 32) for d in (1, 2, 3, 4, 5, 6, 8, 12, 20, 50)] + [(12, 64), (12, 128), (12, 200)]:
        _, rgb = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=depth, seed=11))
        d = rgb.astype(float) - gold
        mse = np.mean(d ** 2)
        # the same after a 4x4 box filter of both (takes the goldens' own noise out)
        f = lambda a: a.reshape(125, 4, 125, 4, 3).mean(axis=(1, 3))
        mse_f = np.mean((f(rgb.astype(float)) - f(gold)) ** 2)
        print("%-8s jitter:%-3d depth %2d  PSNR %.2f dB   4x4-filtered %.2f dB   mean diff %+.3f" % (name, n, depth, 10 * np.log10(255 ** 2 / mse), 10 * np.log10(255 ** 2 / mse_f), d.mean()), flush=True)
