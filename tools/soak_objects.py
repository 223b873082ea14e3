#!/usr/bin/env python3
"""Per-object soak of the specular paths on flat geometry.  tools/soak_images.py compares whole frames, which a 1 % error on
the few pixels of one glass object does not move; here every object is compared on the pixels whose primary hit it is
(tools/box_z.py's statistic), in scenes made of what the bit-level effects of DESIGN.md section 2 live on: axis-aligned boxes
(sometimes at exactly representable coordinates, sometimes standing exactly on the ground plane or on each other), mirror /
glass / diffuse ground planes, an optional glass or mirror sphere, an optional light, cameras that look along an axis or not.
Device: 4 renders of 4096 spp; oracle: 64 renders of 256 spp; standard errors from the spread of the oracle renders (a
device render counts as 16 of them).  An object with |z| > 4 is looked at again with 256 oracle renders and 16 device
renders; it is a finding if |z| > 4.5 stays and the means differ by more than 0.2 %.
python tools/soak_objects.py <first seed> <last seed>     (SOAK_TILT=1: every scene turned out of the axes as a whole; SOAK_DEPTH=n)"""
import importlib, os, sys, time
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


def spec_material(rng):
    k = rng.random()
    if k < 0.45:
        return dict(type="Dieletrics", tint=[float(x) for x in rng.uniform(0.6, 1.0, size=3)], refraction_index=float(rng.choice([1.3, 1.5, 1.886, rng.uniform(1.05, 2.4)])))
    if k < 0.7:
        return dict(type="Metal", albedo=[float(x) for x in rng.uniform(0.5, 0.95, size=3)], fuzziness=0.0)
    if k < 0.8:
        return dict(type="Metal", albedo=[float(x) for x in rng.uniform(0.5, 0.95, size=3)], fuzziness=float(rng.choice([0.05, 0.3])))
    return dict(type="Lambertian", albedo=[float(x) for x in rng.uniform(0.3, 0.9, size=3)])


def scene(seed):
    rng = np.random.default_rng(seed)
    round_numbers = rng.random() < 0.3
    coord = (lambda a, b: float(np.round(rng.uniform(a, b) * 8) / 8)) if round_numbers else (lambda a, b: float(rng.uniform(a, b)))
    ground_y = coord(-1.0, 0.0)
    objs = []
    tops = []
    for b in range(int(rng.integers(1, 4))):
        if tops and rng.random() < 0.4:          # stands exactly on another box
            base = tops[int(rng.integers(len(tops)))]
            lo = np.array([base[0] + coord(0.0, 0.3), base[1], base[2] + coord(0.0, 0.3)])
            hi = lo + np.array([coord(0.3, 0.8), coord(0.3, 0.9), coord(0.3, 0.8)])
        else:
            y0 = ground_y if rng.random() < 0.5 else ground_y + coord(0.01, 0.8)   # on the ground exactly, or floating
            lo = np.array([coord(-2.2, 1.0), y0, coord(-2.5, 0.5)])
            hi = lo + np.array([coord(0.4, 1.6), coord(0.4, 1.5), coord(0.4, 1.6)])
        tops.append((lo[0], hi[1], lo[2]))
        v = box(lo, hi, rng.random() < 0.3)
        objs.append(dict(type="Mesh", verts=v, tris=np.arange(36).reshape(12, 3), materials=[spec_material(rng)]))
    if rng.random() < 0.85:
        gm = spec_material(rng) if rng.random() < 0.6 else dict(type="Lambertian", albedo=[0.6, 0.6, 0.6])
        objs.append(dict(type="Plane", center=[0.0, ground_y, 0.0], normal=[0.0, 1.0, 0.0], material=gm))
    if rng.random() < 0.4:
        objs.append(dict(type="Sphere", center=[coord(-2.0, 2.0), ground_y + coord(0.4, 1.6), coord(-2.0, 1.5)], radius=coord(0.3, 0.7), material=spec_material(rng)))
    if rng.random() < 0.4:
        objs.append(dict(type="Sphere", center=[coord(-3.0, 3.0), coord(2.5, 4.0), coord(-2.0, 2.0)], radius=coord(0.3, 0.8),
                         material=dict(type="Emissive", color=[float(x) for x in rng.uniform(1.0, 6.0, size=3)])))
    if rng.random() < 0.3:     # along an axis: the central column of pixels looks down -z exactly
        ey = coord(0.5, 1.5)
        cam = dict(eye=[0.0, ey, 6.0], look_at=[0.0, ey, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)
    else:
        cam = dict(eye=[coord(-2.0, 2.0), coord(0.3, 2.5), coord(5.0, 7.0)], look_at=[coord(-0.5, 0.5), coord(0.0, 0.6), coord(-1.0, 0.0)], up=[0.0, 1.0, 0.0], fovy=float(rng.uniform(30, 50)))
    bg = [float(x) for x in rng.uniform(0.3, 1.0, size=3)]
    if os.environ.get("SOAK_TILT"):
        # the same scene turned out of the axes as a whole: faces that coincided now coincide to a rounding error only, and no
        # normal has an exact zero
        a, b = rng.uniform(0.1, 0.6, size=2)
        Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        for o in objs:
            if o["type"] == "Mesh":
                o["verts"] = o["verts"] @ Rm.T
            else:
                o["center"] = [float(x) for x in Rm @ np.array(o["center"])]
                if o["type"] == "Plane":
                    o["normal"] = [float(x) for x in Rm @ np.array(o["normal"])]
        cam = dict(cam, eye=[float(x) for x in Rm @ np.array(cam["eye"])], look_at=[float(x) for x in Rm @ np.array(cam["look_at"])], up=[float(x) for x in Rm @ np.array(cam["up"])])
    return objs, bg, cam, int(rng.integers(2, 9)), round_numbers


def main():
    ctx = rt.Context(0)
    n_obj, findings, cleared, zs = 0, 0, 0, []
    for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
        objs, bg, cam, depth, rn = scene(seed)
        depth = int(os.environ.get("SOAK_DEPTH", depth))
        try:
            ctx.upload(rt.Scene.from_objects(W, H, bg, cam, objs))
        except rt.RtError as e:
            print("seed %d: upload: %s" % (seed, e), flush=True)
            continue
        osc = O.OracleScene.from_dict(dict(width=W, height=H, background=bg, camera=cam, objects=[
            (dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(len(o["tris"]), np.int64)))
             if o["type"] == "Mesh" else o) for o in objs]))
        ids = osc.primary_hits(0.5, 0.5)["object_id"]
        dev = lambda k: [ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=977 * seed + j))[0].astype(np.float64) / 4096.0 for j in range(k)]
        ora = lambda k, s0: [osc.render(mode="jitter", N=16, depth=depth, seed=100000 * seed + s0 + j)["sum"] / 256.0 for j in range(k)]
        t = time.time()
        try:
            D, A = dev(4), ora(64, 0)
        except rt.RtError as e:
            print("seed %d: %s" % (seed, e), flush=True)
            continue
        line = []
        for oi, o in enumerate(objs):
            m = ids == oi
            if m.sum() < 12:
                continue
            def stat(D, A):
                d = np.array([x[m].mean() for x in D]); a = np.array([x[m].mean() for x in A])
                se = a.std(ddof=1) * np.sqrt(1.0 / len(a) + 1.0 / (16.0 * len(d)))
                return d.mean(), a.mean(), (d.mean() - a.mean()) / max(se, 1e-15)
            dm, am, z = stat(D, A)
            if am < 1e-9 and dm < 1e-9:
                continue
            n_obj += 1
            mt = (o.get("material") or o["materials"][0])
            tag = "%s%d:%s%s z%+.1f" % (o["type"][0], oi, mt["type"][:3], ("%.2f" % mt.get("refraction_index", mt.get("fuzziness", 0))) if mt["type"] in ("Dieletrics", "Metal") else "", z)
            if abs(z) > 4.0:
                D2, A2 = D + dev(12), A + ora(192, 1000)
                dm, am, z = stat(D2, A2)
                if abs(z) > 4.5 and abs(dm / max(am, 1e-12) - 1.0) > 2e-3:
                    findings += 1
                    tag += " <-- FINDING: %d px, device %.5f oracle %.5f ratio %.4f z %+.1f at 256 oracle renders" % (m.sum(), dm, am, dm / am, z)
                else:
                    cleared += 1
                    tag += " (cleared: ratio %.4f z %+.1f)" % (dm / max(am, 1e-12), z)
            zs.append(z)
            line.append(tag)
        print("seed %5d depth %d%s %4.1f s: %s" % (seed, depth, " round" if rn else "", time.time() - t, "  ".join(line)), flush=True)
    zs = np.array(zs)
    print("%d objects compared, %d findings, %d flagged and cleared, mean z %+.3f +- %.3f, rms z %.2f" % (n_obj, findings, cleared, zs.mean(), zs.std() / np.sqrt(len(zs)), np.sqrt((zs ** 2).mean())))
    sys.exit(1 if findings else 0)


main()
