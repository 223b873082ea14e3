"""Synthetic scenes of BASELINE.json configs C3-C5, expressed in the
reference's own scene vocabulary (YAML objects + OBJ meshes).

Every generator returns a plain dict {width, height, background, camera, objects}; Mesh
objects carry numpy arrays.  `write_files` emits scene.yaml + mesh.obj that the
reference (and the C++ loaders here) can read; `Scene.from_objects` consumes the dict
directly."""
import os

import numpy as np

MASK = (1 << 64) - 1


class SplitMix64:
    """u = (x >> 11) * 2^-53, seed 42 by default."""

    def __init__(self, seed=42):
        self.x = seed & MASK

    def next(self):
        self.x = (self.x + 0x9E3779B97F4A7C15) & MASK
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def unit(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


def rtiow_spheres(seed=42):
    """RTIOW final scene as Sphere objects (C3): ground, 22x22 small spheres, three big ones."""
    g = SplitMix64(seed)
    objs = [dict(type="Sphere", center=[0.0, -1000.0, 0.0], radius=1000.0, material=dict(type="Lambertian", albedo=[0.5, 0.5, 0.5]))]
    for a in range(-11, 11):
        for b in range(-11, 11):
            u1, u2, u3 = g.unit(), g.unit(), g.unit()
            c = [a + 0.9 * u1, 0.2, b + 0.9 * u2]
            if ((c[0] - 4.0) ** 2 + (c[1] - 0.2) ** 2 + c[2] ** 2) ** 0.5 <= 0.9:
                continue
            if u3 < 0.8:
                alb = [g.unit() * g.unit() for _ in range(3)]
                m = dict(type="Lambertian", albedo=alb)
            elif u3 < 0.95:
                alb = [0.5 + 0.5 * g.unit() for _ in range(3)]
                m = dict(type="Metal", albedo=alb, fuzziness=0.5 * g.unit())
            else:
                m = dict(type="Dieletrics", tint=[1.0, 1.0, 1.0], refraction_index=1.5)
            objs.append(dict(type="Sphere", center=c, radius=0.2, material=m))
    objs.append(dict(type="Sphere", center=[0.0, 1.0, 0.0], radius=1.0, material=dict(type="Dieletrics", tint=[1.0, 1.0, 1.0], refraction_index=1.5)))
    objs.append(dict(type="Sphere", center=[-4.0, 1.0, 0.0], radius=1.0, material=dict(type="Lambertian", albedo=[0.4, 0.2, 0.1])))
    objs.append(dict(type="Sphere", center=[4.0, 1.0, 0.0], radius=1.0, material=dict(type="Metal", albedo=[0.7, 0.6, 0.5], fuzziness=0.0)))
    return objs


RTIOW_CAMERA = dict(eye=[13.0, 2.0, 3.0], look_at=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fovy=20.0)
SKY = [0.7, 0.8, 1.0]


def scene_c3(width=1920, height=1080, seed=42):
    return dict(width=width, height=height, background=SKY, camera=RTIOW_CAMERA, objects=rtiow_spheres(seed))


def torus_mesh(nu=1000, nv=500, center=(0.0, 1.6, 0.0), R=3.0, r=1.0, bump=0.05, outward=True):
    """Bumpy torus around the y axis: nu x nv quads -> 2*nu*nv triangles, nu*nv vertices.

    outward: winding such that the geometric normal normalize((B-A) x (C-A)) (intersections.rs:169) points out of
    the tube.  The reference never flips a normal towards the ray, so the winding decides where a Lambertian
    surface scatters to: with outward=False (the round-1 generator) every path that reaches the torus is scattered
    INTO the tube, stays trapped there for all `depth` bounces and returns black -- a valid scene of the reference,
    but a black torus hides any bias of the image gates.  The default is the lit torus."""
    th = np.arange(nu, dtype=np.float64) * (2.0 * np.pi / nu)
    ph = np.arange(nv, dtype=np.float64) * (2.0 * np.pi / nv)
    T, P = np.meshgrid(th, ph, indexing="ij")
    rr = r + bump * np.sin(8.0 * T) * np.sin(12.0 * P)
    x = (R + rr * np.cos(P)) * np.cos(T) + center[0]
    z = (R + rr * np.cos(P)) * np.sin(T) + center[2]
    y = rr * np.sin(P) + center[1]
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    i1, j1 = (i + 1) % nu, (j + 1) % nv
    v00, v10, v01, v11 = i * nv + j, i1 * nv + j, i * nv + j1, i1 * nv + j1
    if outward:
        t1 = np.stack([v00, v11, v10], axis=-1).reshape(-1, 3)
        t2 = np.stack([v00, v01, v11], axis=-1).reshape(-1, 3)
    else:
        t1 = np.stack([v00, v10, v11], axis=-1).reshape(-1, 3)
        t2 = np.stack([v00, v11, v01], axis=-1).reshape(-1, 3)
    tris = np.empty((2 * nu * nv, 3), dtype=np.int64)
    tris[0::2], tris[1::2] = t1, t2
    return verts, tris


def scene_c4(width=3840, height=2160, nu=1000, nv=500, outward=True):
    verts, tris = torus_mesh(nu, nv, outward=outward)
    objs = [dict(type="Mesh", name="Torus", verts=verts, tris=tris, materials=[dict(type="Lambertian", albedo=[0.6, 0.6, 0.6])], mat_names=["M"]),
            dict(type="Plane", center=[0.0, 0.0, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Lambertian", albedo=[0.2, 0.2, 0.2]))]
    cam = dict(eye=[6.0, 5.0, 8.0], look_at=[0.0, 1.2, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)
    return dict(width=width, height=height, background=SKY, camera=cam, objects=objs)


def scene_c5(width=1920, height=1080, nu=1000, nv=500, seed=42, outward=True):
    """C3 spheres + the C4 torus translated to (0, 1.6, -6); no ground plane."""
    verts, tris = torus_mesh(nu, nv, center=(0.0, 1.6, -6.0), outward=outward)
    objs = rtiow_spheres(seed)
    objs.append(dict(type="Mesh", name="Torus", verts=verts, tris=tris, materials=[dict(type="Lambertian", albedo=[0.6, 0.6, 0.6])], mat_names=["M"]))
    return dict(width=width, height=height, background=SKY, camera=RTIOW_CAMERA, objects=objs)


def to_oracle_dict(sc):
    """Same scene in the shape oracle_py.OracleScene.from_dict expects (tests / cpu_baseline)."""
    objs = []
    for o in sc["objects"]:
        if o["type"] == "Mesh":
            n = len(o["tris"])
            objs.append(dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"],
                                                                              tri_mat_np=np.zeros(n, np.int64))))
        else:
            objs.append(o)
    return dict(width=sc["width"], height=sc["height"], background=sc["background"], camera=sc["camera"], objects=objs)


def _fmt(x):
    return repr(float(x))


def _mat_yaml(m, ind):
    out = [ind + "type: " + m["type"]]
    for k in ("albedo", "tint", "color"):
        if k in m:
            out.append(ind + "%s: [%s]" % (k, ", ".join(_fmt(x) for x in m[k])))
    for k in ("fuzziness", "refraction_index"):
        if k in m:
            out.append(ind + "%s: %s" % (k, _fmt(m[k])))
    return out


def write_files(sc, directory, name="scene", path_prefix=None):
    """Writes <directory>/<name>.yaml (+ <name>_<k>.obj per mesh).  Mesh paths inside the YAML are
    written as `path_prefix/<file>` (default: the directory itself), because the reference
    resolves them against the working directory."""
    os.makedirs(directory, exist_ok=True)
    prefix = directory if path_prefix is None else path_prefix
    L = ["width: %d" % sc["width"], "height: %d" % sc["height"], "background: [%s]" % ", ".join(_fmt(x) for x in sc["background"]),
         "camera:"]
    for k in ("eye", "look_at", "up"):
        L.append("  %s: [%s]" % (k, ", ".join(_fmt(x) for x in sc["camera"][k])))
    L.append("  fovy: %s" % _fmt(sc["camera"]["fovy"]))
    L.append("objects:")
    nmesh = 0
    for o in sc["objects"]:
        L.append("  - type: " + o["type"])
        if o["type"] == "Sphere":
            L.append("    center: [%s]" % ", ".join(_fmt(x) for x in o["center"]))
            L.append("    radius: %s" % _fmt(o["radius"]))
            L.append("    material:")
            L += _mat_yaml(o["material"], "      ")
        elif o["type"] == "Plane":
            L.append("    center: [%s]" % ", ".join(_fmt(x) for x in o["center"]))
            L.append("    normal: [%s]" % ", ".join(_fmt(x) for x in o["normal"]))
            L.append("    material:")
            L += _mat_yaml(o["material"], "      ")
        else:
            fn = "%s_%d.obj" % (name, nmesh)
            nmesh += 1
            L.append("    path: %s" % os.path.join(prefix, fn))
            L.append("    materials:")
            names = o.get("mat_names", ["M%d" % i for i in range(len(o["materials"]))])
            for nm, m in zip(names, o["materials"]):
                L.append("      %s:" % nm)
                L += _mat_yaml(m, "        ")
            with open(os.path.join(directory, fn), "w") as f:
                f.write("o %s\n" % o.get("name", "Mesh"))
                f.write("".join("v %r %r %r\n" % (float(a), float(b), float(c)) for a, b, c in o["verts"]))
                f.write("usemtl %s\n" % names[0])
                f.write("".join("f %d %d %d\n" % (a + 1, b + 1, c + 1) for a, b, c in o["tris"]))
    path = os.path.join(directory, name + ".yaml")
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path
