#!/usr/bin/env python3
"""Statistical soak of the Texture path through the real loaders: per seed a temporary scene in the reference's own formats
(YAML + OBJ with v/vt faces + PNG): one to three textured axis-aligned boxes and a textured tilted quad at coordinates that
are no round numbers, over a ground plane, with spheres and one emissive sphere (a light); both sides of the faces are seen
(front: scatters with the texel as albedo, back: emits the texel, materials.rs:233-253 -- and a ray that re-hits the face it
leaves turns a front face into a back face, so the re-hit rates show directly).  Device (rt.Scene.load) at 4096 spp against
the oracle (its own loader) at 2 x 16 x 256 spp.  python tools/soak_textures.py <first seed> <last seed>"""
import importlib, os, sys, tempfile
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
rt = importlib.import_module("raytracer-rust_amd")
W, H = 48, 36


def write_scene(d, seed):
    rng = np.random.default_rng(seed)
    tw, th = int(rng.integers(2, 9)), int(rng.integers(2, 9))
    Image.fromarray(rng.integers(30, 250, size=(th, tw, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "tex.png"))
    v, vt, f = [], [], []
    ground_y = float(rng.choice([-0.6, -0.37, 0.0, -1.0]))
    def quad(c):      # c: 4 corners; two triangles with uvs over a random sub-rectangle of the texture
        u0, u1 = sorted(rng.uniform(0.02, 0.98, size=2)); v0, v1 = sorted(rng.uniform(0.02, 0.98, size=2))
        b, t = len(v), len(vt)
        v.extend(c); vt.extend([(u0, v0), (u1, v0), (u1, v1), (u0, v1)])
        idx = [(0, 1, 2), (0, 2, 3)] if rng.random() < 0.5 else [(0, 2, 1), (0, 3, 2)]
        for tri in idx:
            f.append(" ".join("%d/%d" % (b + k + 1, t + k + 1) for k in tri))
    for _ in range(int(rng.integers(1, 4))):
        lo = np.array([rng.uniform(-2.4, 1.2), ground_y + rng.uniform(0.0, 0.6), rng.uniform(-2.5, 0.3)])
        hi = lo + rng.uniform(0.4, 1.4, size=3)
        c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
        for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
            quad([c[i] for i in q])
    p0 = np.array([rng.uniform(-2.5, 0.5), ground_y + rng.uniform(0.2, 1.0), rng.uniform(-1.0, 1.5)])
    e1, e2 = rng.normal(size=3) * np.array([1.0, 0.3, 0.6]), rng.normal(size=3) * np.array([0.3, 1.0, 0.4])
    quad([p0, p0 + e1, p0 + e1 + e2, p0 + e2])
    with open(os.path.join(d, "mesh.obj"), "w") as fh:
        fh.write("o Mesh\n" + "".join("v %r %r %r\n" % tuple(float(x) for x in p) for p in v) + "".join("vt %r %r\n" % (float(a), float(b)) for a, b in vt) +
                 "usemtl Tex\n" + "".join("f %s\n" % x for x in f))
    col = lambda lo_, hi_: "[%s]" % ", ".join(repr(float(x)) for x in rng.uniform(lo_, hi_, size=3))
    objs = ["  - type: Mesh\n    path: ./mesh.obj\n    materials:\n      Tex:\n        type: Texture\n        texture_path: ./tex.png\n",
            "  - type: Plane\n    center: [0.0, %r, 0.0]\n    normal: [0.0, 1.0, 0.0]\n    material:\n      type: %s\n" % (ground_y, "Lambertian\n      albedo: " + col(0.2, 0.9) if rng.random() < 0.7 else "Metal\n      albedo: " + col(0.5, 0.95) + "\n      fuzziness: 0.1")]
    for k in range(int(rng.integers(1, 5))):
        kind = rng.integers(0, 3)
        m = ("Lambertian\n      albedo: " + col(0.2, 0.9), "Metal\n      albedo: " + col(0.5, 0.95) + "\n      fuzziness: %r" % float(rng.choice([0.0, 0.2])),
             "Dieletrics\n      tint: [1.0, 1.0, 1.0]\n      refraction_index: %r" % float(rng.uniform(1.2, 1.8)))[kind]
        objs.append("  - type: Sphere\n    center: [%r, %r, %r]\n    radius: %r\n    material:\n      type: %s\n" %
                    (float(rng.uniform(-2.5, 2.5)), ground_y + float(rng.uniform(0.3, 1.6)), float(rng.uniform(-2.5, 2.0)), float(rng.uniform(0.2, 0.7)), m))
    if rng.random() < 0.7:
        objs.append("  - type: Sphere\n    center: [%r, %r, %r]\n    radius: %r\n    material:\n      type: Emissive\n      color: %s\n" %
                    (float(rng.uniform(-2.0, 2.0)), ground_y + float(rng.uniform(2.0, 3.0)), float(rng.uniform(-2.0, 1.0)), float(rng.uniform(0.3, 0.8)), col(1.0, 4.0)))
    bg = col(0.0, 1.0) if rng.random() < 0.8 else "[0.0, 0.0, 0.0]"
    with open(os.path.join(d, "scene.yaml"), "w") as fh:
        fh.write("width: %d\nheight: %d\nbackground: %s\ncamera:\n  eye: [%r, %r, 6.0]\n  look_at: [0.0, 0.3, 0.0]\n  up: [0.0, 1.0, 0.0]\n  fovy: 40.0\nobjects:\n%s" %
                 (W, H, bg, float(rng.uniform(-2.0, 2.0)), float(rng.uniform(0.6, 3.0)), "".join(objs)))
    return int(rng.integers(2, 8))


def main():
    ctx = rt.Context(0)
    bad = 0
    zs = []
    for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
        with tempfile.TemporaryDirectory() as d:
            depth = write_scene(d, seed)
            ctx.upload(rt.Scene.load(os.path.join(d, "scene.yaml"), base_dir=d))
            osc = O.OracleScene.from_yaml(os.path.join(d, "scene.yaml"), base_dir=d)
        try:
            acc, _ = ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=seed))
        except rt.RtError as e:
            print("seed %d: %s" % (seed, e), flush=True)
            continue
        D = acc.astype(np.float64) / 4096.0
        more = [osc.render(mode="jitter", N=16, depth=depth, seed=1000 * seed + k)["sum"] / 256.0 for k in range(32)]
        A, B = np.mean(more[:16], axis=0), np.mean(more[16:], axis=0)
        g = lambda x: np.sqrt(np.minimum(np.maximum(x, 0.0), 1.0))
        psnr = lambda a, b: 10 * np.log10(1.0 / max(np.mean((g(a) - g(b)) ** 2), 1e-12))
        gain = psnr(D, 0.5 * (A + B)) - psnr(A, B)
        means = np.array([m.mean() for m in more])
        z = (D.mean() - means.mean()) / max(means.std(ddof=1) * np.sqrt(1.0 / 32.0 + 1.0 / 16.0), 1e-12)
        ids = osc.primary_hits(0.5, 0.5)["object_id"]
        mesh = ids == 0
        ratio = D[mesh].mean() / max(0.5 * (A + B)[mesh].mean(), 1e-12) if mesh.sum() else float("nan")
        flag = "" if (gain >= -0.5 and abs(z) <= 5.0) else "   <-- CHECK"
        bad += bool(flag)
        zs.append(z)
        print("seed %4d: depth %d, oracle A-B %.1f dB, device gain %+.1f dB, z %+.2f, mesh pixels %4d device/oracle %.4f%s" % (seed, depth, psnr(A, B), gain, z, mesh.sum(), ratio, flag), flush=True)
    zs = np.array(zs)
    print("%d scenes, %d flagged, mean z %+.3f +- %.3f" % (len(zs), bad, zs.mean(), zs.std() / np.sqrt(len(zs))))
    sys.exit(1 if bad else 0)


main()
