"""ctypes front end of the CPU oracle (oracle/rt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product package.

Scene loading here is deliberately independent of the product's C++ loaders: it
uses PyYAML (serde_yaml stand-in), Pillow (png crate stand-in) and a small OBJ
reader that restates /root/reference/src/raytracer/scene/mesh.rs:22-290, so a
product-vs-oracle comparison also cross-checks the product's YAML/OBJ/PNG code.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAT_TYPES = {"Lambertian": 0, "Metal": 1, "Dieletrics": 2, "Texture": 3, "Emissive": 4}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rt_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_new.argtypes = [C.c_int, C.c_long, C.c_long, dp, dp, dp, dp, C.c_double]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_add_material.restype = C.c_int
        L.orc_add_material.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.orc_add_sphere.argtypes = [C.c_void_p, dp, C.c_double, C.c_int]
        L.orc_add_plane.argtypes = [C.c_void_p, dp, dp, C.c_int]
        L.orc_add_mesh.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p,
                                   C.c_void_p, C.c_long, C.c_void_p]
        L.orc_prepare.argtypes = [C.c_void_p]
        L.orc_enable_accel.argtypes = [C.c_void_p, C.c_int]
        L.orc_num_lights.restype = C.c_int
        L.orc_num_lights.argtypes = [C.c_void_p]
        L.orc_light_points.restype = C.c_int
        L.orc_light_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_camera_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_spawn_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_primary_hits.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_closest_hit.restype = C.c_int
        L.orc_closest_hit.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.orc_closest_hits.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_render.restype = C.c_uint64
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_uint64, C.c_int, C.c_long, C.c_long, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
        L.orc_render_crop.restype = C.c_uint64
        L.orc_render_crop.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_uint64, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long,
                                      C.c_long, C.c_void_p]
        L.orc_texture_oob_count.restype = C.c_uint64
        L.orc_vec_dot_cross.argtypes = [C.c_void_p] * 4
        L.orc_vec_normalized.argtypes = [C.c_void_p] * 2
        L.orc_vec_reflect.argtypes = [C.c_void_p] * 3
        L.orc_vec_near_zero.restype = C.c_int
        L.orc_vec_near_zero.argtypes = [C.c_void_p]
        L.orc_random_unit_vector.argtypes = [C.c_uint64, C.c_void_p]
        L.orc_random_in_unit_sphere.argtypes = [C.c_uint64, C.c_void_p]
        L.orc_ssaa_parse.restype = C.c_int
        L.orc_ssaa_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.c_char_p, C.c_int]
        L.orc_ssaa_sample.argtypes = [C.c_int, C.c_long, C.c_long, C.c_long, C.c_uint64, C.c_void_p]
        L.orc_aabb_intersect.restype = C.c_int
        L.orc_aabb_intersect.argtypes = [C.c_void_p] * 4
        L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_reflectance.restype = C.c_double
        L.orc_reflectance.argtypes = [C.c_double, C.c_double]
        L.orc_scatter.restype = C.c_int
        L.orc_scatter.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_void_p]
        L.orc_to_output.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_texture_albedo.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def _d3(v):
    a = (C.c_double * 3)(*[float(x) for x in v])
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# --------------------------------------------------------------------------
# OBJ reader restating mesh.rs:22-290 (first `o` object only is kept by the
# scene deserializer, scene.rs:169-170).
# --------------------------------------------------------------------------
def load_obj(path):
    """Returns list of mesh dicts (one per `o`), each with verts, uvs, tris
    (vertex idx), uv_idx or None, tri_mat (index into `usemtl` list), mat_names."""
    meshes = []
    vcount = uvcount = ncount = 0  # IndexHelper mesh.rs:237-290
    cur = None
    mat_index = -1
    with open(path, "r") as f:
        for line in f:
            vals = line.split()
            if not vals:
                continue
            head = vals[0]
            if head == "o":
                if cur is not None:
                    vcount += len(cur["verts"])
                    uvcount += len(cur["uvs"])
                    ncount += len(cur["normals"])
                cur = dict(verts=[], uvs=[], normals=[], tris=[], uv_idx=[], tri_mat=[], mat_names=[])
                meshes.append(cur)
                mat_index = -1
            elif head == "v":
                cur["verts"].append((float(vals[1]), float(vals[2]), float(vals[3])))
            elif head == "vn":
                cur["normals"].append((float(vals[1]), float(vals[2]), float(vals[3])))
            elif head == "vt":
                cur["uvs"].append((float(vals[1]), float(vals[2])))
            elif head == "f":
                first = vals[1]
                if "//" in first:
                    raise ValueError("v//vn faces are not parseable by the reference (mesh.rs:206-210)")
                if "/" not in first:  # VPos: exactly three tokens read, extras dropped (mesh.rs:141-146)
                    cur["tris"].append(tuple(int(t) - vcount - 1 for t in vals[1:4]))
                    cur["uv_idx"].append(None)
                elif first.find("/") == first.rfind("/"):  # VPosUv: all tokens, 4th panics (mesh.rs:198-213)
                    if len(vals) > 4:
                        raise ValueError("v/vt face with more than three vertices panics in the reference")
                    vi, ui = [], []
                    for t in vals[1:]:
                        a, b = t.split("/")
                        vi.append(int(a) - vcount - 1)
                        ui.append(int(b) - uvcount - 1)
                    cur["tris"].append(tuple(vi))
                    cur["uv_idx"].append(tuple(ui))
                else:  # VPosUvN: exactly three tokens (mesh.rs:165-181)
                    vi, ui = [], []
                    for t in vals[1:4]:
                        a, b, _c = t.split("/")
                        vi.append(int(a) - vcount - 1)
                        ui.append(int(b) - uvcount - 1)
                    cur["tris"].append(tuple(vi))
                    cur["uv_idx"].append(tuple(ui))
                cur["tri_mat"].append(mat_index)
            elif head == "usemtl":
                cur["mat_names"].append(vals[1])
                mat_index = len(cur["mat_names"]) - 1
    return meshes


def read_png_rgb8(path):
    """image.rs:166-184 read_image: raw decoded bytes grouped in threes."""
    from PIL import Image
    im = Image.open(path)
    if im.mode != "RGB":
        raise ValueError("only RGB8 textures are supported (mode %s)" % im.mode)
    a = np.asarray(im, dtype=np.uint8)
    return np.ascontiguousarray(a), im.width, im.height


def load_scene_dict(yaml_path, base_dir=None):
    """YAML -> plain dict following the serde model (scene.rs:18-27).
    Paths inside the YAML are resolved against base_dir (the reference uses the
    process working directory)."""
    import yaml
    with open(yaml_path, "r") as f:
        y = yaml.safe_load(f)
    base = base_dir if base_dir is not None else os.getcwd()
    objs = []
    for o in y["objects"]:
        t = o["type"]
        if t == "Mesh":
            p = os.path.normpath(os.path.join(base, o["path"]))
            meshes = load_obj(p)
            m = meshes[0]  # scene.rs:170
            if o.get("all_objects") is True:   # additive key: the other `o` objects are appended to the first
                for extra in meshes[1:]:
                    v0, u0, m0 = len(m["verts"]), len(m["uvs"]), len(m["mat_names"])
                    m["tris"] += [tuple(i + v0 for i in t) for t in extra["tris"]]
                    m["uv_idx"] += [None if u is None else tuple(i + u0 for i in u) for u in extra["uv_idx"]]
                    m["tri_mat"] += [k + m0 if k >= 0 else k for k in extra["tri_mat"]]
                    m["verts"] += extra["verts"]
                    m["uvs"] += extra["uvs"]
                    m["mat_names"] += extra["mat_names"]
            mats = []
            for name in m["mat_names"]:
                md = dict(o["materials"][name])
                if md["type"] == "Texture":
                    md["_texture_file"] = os.path.normpath(os.path.join(base, md["texture_path"]))
                mats.append(md)
            objs.append(dict(type="Mesh", mesh=m, materials=mats))
        else:
            objs.append(o)
    return dict(width=int(y["width"]), height=int(y["height"]), background=y["background"], camera=y["camera"], objects=objs)


class OracleScene:
    """Owns an OrcScene handle.  prec=0: f64 oracle; prec=1: f32 diagnostic."""

    def __init__(self, width, height, background, camera, prec=0):
        L = lib()
        self.L = L
        self.width, self.height = int(width), int(height)
        self.h = L.orc_scene_new(prec, self.width, self.height, _d3(background), _d3(camera["eye"]), _d3(camera["look_at"]),
                                 _d3(camera["up"]), float(camera["fovy"]))
        self._keep = []
        self.prec = prec
        self.prepared = False

    def __del__(self):
        try:
            if self.h:
                self.L.orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def add_material(self, md):
        t = MAT_TYPES[md["type"]]
        rgb, param, tw, th, tex = [0.0, 0.0, 0.0], 0.0, 0, 0, None
        if t == 0:
            rgb = md["albedo"]
        elif t == 1:
            rgb, param = md["albedo"], md["fuzziness"]
        elif t == 2:
            rgb, param = md["tint"], md["refraction_index"]
        elif t == 4:
            rgb = md["color"]
        elif t == 3:
            if "_pixels" in md:
                tex_arr, tw, th = md["_pixels"], md["_w"], md["_h"]
            else:
                tex_arr, tw, th = read_png_rgb8(md.get("_texture_file", md["texture_path"]))
            tex_arr = np.ascontiguousarray(tex_arr, dtype=np.uint8)
            self._keep.append(tex_arr)
            tex = _p(tex_arr)
        return self.L.orc_add_material(self.h, t, _d3(rgb), float(param), int(tw), int(th), tex)

    def add_sphere(self, center, radius, material):
        self.L.orc_add_sphere(self.h, _d3(center), float(radius), self.add_material(material))

    def add_plane(self, center, normal, material):
        self.L.orc_add_plane(self.h, _d3(center), _d3(normal), self.add_material(material))

    def add_mesh(self, verts, tris, tri_mat, materials, uvs=None, uv_idx=None):
        verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, dtype=np.int64).reshape(-1, 3)
        tri_mat = np.ascontiguousarray(tri_mat, dtype=np.int64).reshape(-1)
        mats = np.array([self.add_material(m) for m in materials], dtype=np.int32)
        uvs_a = np.ascontiguousarray(uvs, dtype=np.float64).reshape(-1, 2) if uvs is not None and len(uvs) else None
        uvi_a = np.ascontiguousarray(uv_idx, dtype=np.int64).reshape(-1, 3) if uv_idx is not None else None
        self.L.orc_add_mesh(self.h, len(verts), _p(verts), len(tris), _p(tris), _p(uvi_a), 0 if uvs_a is None else len(uvs_a),
                            _p(uvs_a), _p(tri_mat), len(mats), _p(mats))

    def prepare(self):
        self.L.orc_prepare(self.h)
        self.prepared = True
        return self

    def enable_accel(self, on=True):
        """Conservative cull in front of every mesh's triangle loop (rt_oracle.cpp header): the same results, bit for bit,
        without testing all triangles.  Test infrastructure for full-size scenes; never on for the timed cpu_baseline."""
        self.L.orc_enable_accel(self.h, 1 if on else 0)
        return self

    @classmethod
    def from_dict(cls, d, prec=0, accel=False):
        sc = cls(d["width"], d["height"], d["background"], d["camera"], prec=prec)
        for o in d["objects"]:
            if o["type"] == "Sphere":
                sc.add_sphere(o["center"], o["radius"], o["material"])
            elif o["type"] == "Plane":
                sc.add_plane(o["center"], o["normal"], o["material"])
            elif o["type"] == "Mesh":
                m = o["mesh"]
                if "verts_np" in m:
                    sc.add_mesh(m["verts_np"], m["tris_np"], m["tri_mat_np"], o["materials"], m.get("uvs_np"), m.get("uv_idx_np"))
                else:
                    has_uv = len(m["uv_idx"]) > 0 and all(u is not None for u in m["uv_idx"])
                    sc.add_mesh(m["verts"], m["tris"], m["tri_mat"], o["materials"], m["uvs"] if has_uv else None,
                                m["uv_idx"] if has_uv else None)
            else:
                raise ValueError(o["type"])
        sc.prepare()
        return sc.enable_accel() if accel else sc

    @classmethod
    def from_yaml(cls, path, base_dir=None, prec=0):
        return cls.from_dict(load_scene_dict(path, base_dir), prec=prec)

    # -- queries ----------------------------------------------------------
    def lights(self):
        out = []
        for i in range(self.L.orc_num_lights(self.h)):
            buf = np.zeros((64, 3))
            n = self.L.orc_light_points(self.h, i, _p(buf), 64)
            out.append(buf[:min(n, 64)].copy())
        return out

    def camera(self):
        e, h, v, l = (np.zeros(3) for _ in range(4))
        self.L.orc_camera_get(self.h, _p(e), _p(h), _p(v), _p(l))
        return dict(eye=e, horizontal=h, vertical=v, lower_left=l)

    def spawn_ray(self, x, y):
        o, d = np.zeros(3), np.zeros(3)
        self.L.orc_spawn_ray(self.h, float(x), float(y), _p(o), _p(d))
        return o, d

    def primary_hits(self, dx=0.0, dy=0.0, crop=None):
        if crop is None:
            x0, x1, r0, r1 = 0, self.width, 0, self.height
        else:
            x0, x1, r0, r1 = crop
        n = (x1 - x0) * (r1 - r0)
        obj = np.zeros(n, np.int32)
        tri = np.zeros(n, np.int32)
        t = np.zeros(n)
        nrm = np.zeros((n, 3))
        pt = np.zeros((n, 3))
        self.L.orc_primary_hits(self.h, float(dx), float(dy), x0, x1, r0, r1, _p(obj), _p(tri), _p(t), _p(nrm), _p(pt))
        shp = (r1 - r0, x1 - x0)
        return dict(object_id=obj.reshape(shp), triangle_id=tri.reshape(shp), t=t.reshape(shp), normal=nrm.reshape(shp + (3,)),
                    point=pt.reshape(shp + (3,)))

    def closest_hit(self, o, d):
        obj, tri = C.c_int32(-1), C.c_int32(-1)
        t = C.c_double(0)
        pt, nrm, uv = np.zeros(3), np.zeros(3), np.zeros(2)
        oo, dd = np.asarray(o, dtype=np.float64), np.asarray(d, dtype=np.float64)
        ok = self.L.orc_closest_hit(self.h, _p(oo), _p(dd), C.byref(obj), C.byref(tri), C.byref(t), _p(pt), _p(nrm), _p(uv))
        if not ok:
            return None
        return dict(object_id=obj.value, triangle_id=tri.value, t=t.value, point=pt, normal=nrm, uv=uv)

    def closest_hits(self, origins, dirs, threads=None):
        """Closest hits of arbitrary rays (f64 oracle): origins (3,) or (..., 3), dirs (..., 3); returns arrays shaped like dirs[..., 0]."""
        assert self.prec == 0
        dd = np.ascontiguousarray(dirs, dtype=np.float64)
        shp = dd.shape[:-1]
        dd = dd.reshape(-1, 3)
        oo = np.ascontiguousarray(np.broadcast_to(np.asarray(origins, dtype=np.float64).reshape(-1, 3) if np.ndim(origins) > 1 else
                                                  np.asarray(origins, dtype=np.float64), dd.shape))
        n = len(dd)
        obj, tri = np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, nrm, pt = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3))
        if threads is None:
            threads = min(os.cpu_count() or 1, 16)
        self.L.orc_closest_hits(self.h, n, _p(oo), _p(dd), int(threads), _p(obj), _p(tri), _p(t), _p(nrm), _p(pt))
        return dict(object_id=obj.reshape(shp), triangle_id=tri.reshape(shp), t=t.reshape(shp), normal=nrm.reshape(shp + (3,)),
                    point=pt.reshape(shp + (3,)))

    def render(self, mode="jitter", N=2, depth=5, seed=1, threads=None, s_begin=0, s_end=-1):
        """Returns dict(sum, color, rgb8, rays). sum = per-pixel sum of sample colours."""
        if threads is None:
            threads = min(os.cpu_count() or 1, 64)
        H, W = self.height, self.width
        s = np.zeros((H, W, 3))
        c = np.zeros((H, W, 3))
        b = np.zeros((H, W, 3), np.uint8)
        rays = self.L.orc_render(self.h, 0 if mode == "uniform" else 1, int(N), int(depth), int(seed), int(threads), int(s_begin),
                                 int(s_end), _p(s), _p(c), _p(b))
        return dict(sum=s, color=c, rgb8=b, rays=int(rays))

    def render_crop(self, crop, mode="jitter", N=1, depth=5, seed=1, threads=None, step=1):
        """Renders only every step-th pixel/row of crop = (x0, x1, row0, row1); returns dict(sum (full frame, zeros outside), rays)."""
        if threads is None:
            threads = min(os.cpu_count() or 1, 64)
        s = np.zeros((self.height, self.width, 3))
        x0, x1, r0, r1 = crop
        rays = self.L.orc_render_crop(self.h, 0 if mode == "uniform" else 1, int(N), int(depth), int(seed), int(threads), x0, x1, r0, r1, int(step), _p(s))
        return dict(sum=s, rays=int(rays))
