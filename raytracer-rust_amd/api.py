"""Host-side mirror of the reference's render interface (same names, argument meaning
and error behaviour) on top of the C ABI.  Reference: /root/reference/src/ ...
  SuperSampling            raytracer/anti_aliasing.rs:6-84
  Scene                    raytracer/scene/scene.rs:18-27 (+ loaders in rt_host.cpp)
  compute_image            raytracer/raytrace.rs:108-156
"""
import ctypes as C
import os

import numpy as np

from . import lib as _lib
from .lib import RtError

MAT = {"Lambertian": 0, "Metal": 1, "Dieletrics": 2, "Texture": 3, "Emissive": 4}


class SSAADecodeError(ValueError):
    """anti_aliasing.rs:23-45: Display = 'error parsing ssaa argument: {error}'."""

    def __init__(self, error):
        self.error = error
        super().__init__("error parsing ssaa argument: %s" % error)


class SuperSampling:
    """anti_aliasing.rs:6-9.  kind is 'Uniform' or 'Jitter'."""

    def __init__(self, kind, resolution):
        if kind not in ("Uniform", "Jitter"):
            raise ValueError(kind)
        self.kind, self.resolution = kind, int(resolution)

    @classmethod
    def from_str(cls, s):
        """anti_aliasing.rs:50-83; 'uniform:N' yields Jitter(N) like the reference (line 75).
        'grid:N' is the additive spelling of the true uniform grid."""
        L = _lib.load_library()
        mode, n = C.c_int32(0), C.c_uint32(0)
        err = C.create_string_buffer(256)
        if L.rt_host_ssaa_parse(s.encode(), C.byref(mode), C.byref(n), err, 256):
            raise SSAADecodeError(err.value.decode())
        return cls("Jitter" if mode.value == 1 else "Uniform", n.value)

    @property
    def mode(self):
        return 1 if self.kind == "Jitter" else 0

    @property
    def samples_per_pixel(self):
        return self.resolution * self.resolution

    def __repr__(self):
        return "%s(%d)" % (self.kind, self.resolution)


class RenderParams:
    def __init__(self, ssaa, depth=5, seed=0, sample_begin=0, sample_end=0, tile_index=0, tile_count=1, max_paths_per_pass=0,
                 max_shadow_generations=0, counters=False, timing=False, sample_offset=(0.0, 0.0)):
        if isinstance(ssaa, str):
            ssaa = SuperSampling.from_str(ssaa)
        self.ssaa, self.depth, self.seed = ssaa, int(depth), int(seed)
        self.sample_begin, self.sample_end = int(sample_begin), int(sample_end)
        self.tile_index, self.tile_count = int(tile_index), int(tile_count)
        self.max_paths_per_pass, self.max_shadow_generations = int(max_paths_per_pass), int(max_shadow_generations)
        self.counters, self.timing = bool(counters), bool(timing)
        self.sample_offset = (float(sample_offset[0]), float(sample_offset[1]))

    def num_samples(self):
        end = self.sample_end if self.sample_end else self.ssaa.samples_per_pixel
        return end - self.sample_begin

    def to_c(self):
        p = _lib.RenderParamsC()
        p.ssaa_mode, p.ssaa_n, p.depth, p.seed = self.ssaa.mode, self.ssaa.resolution, self.depth, self.seed & (2**64 - 1)
        p.sample_begin, p.sample_end, p.tile_index, p.tile_count = self.sample_begin, self.sample_end, self.tile_index, self.tile_count
        p.max_paths_per_pass, p.max_shadow_generations = self.max_paths_per_pass, self.max_shadow_generations
        p.flags = (1 if self.counters else 0) | (2 if self.timing else 0)
        p.sample_offset[0], p.sample_offset[1] = self.sample_offset
        return p


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Scene:
    """A loaded + flattened scene (rt_scene_desc).  Either from the reference's YAML/OBJ/PNG
    files through the C++ host loaders (Scene.load) or from arrays (Scene.from_objects)."""

    def __init__(self):
        self._L = _lib.load_library()
        self._handle = None      # rt_host_scene*
        self._desc = None        # SceneDesc (owned by python when built from arrays)
        self._keep = []

    def __del__(self):
        try:
            if self._handle:
                self._L.rt_host_scene_free(self._handle)
                self._handle = None
        except Exception:
            pass

    @classmethod
    def load(cls, yaml_path, base_dir=None):
        """serde_yaml::from_reader + Camera::new + Scene::precompute (raytrace.rs:114-125)."""
        s = cls()
        err = C.create_string_buffer(1024)
        h = s._L.rt_host_scene_load(os.fsencode(yaml_path), os.fsencode(base_dir) if base_dir is not None else None, err, 1024)
        if not h:
            raise RtError(err.value.decode())
        s._handle = h
        return s

    @property
    def desc(self):
        if self._handle:
            return self._L.rt_host_scene_desc(self._handle).contents
        return self._desc

    @property
    def width(self):
        return self.desc.width

    @property
    def height(self):
        return self.desc.height

    def object_kinds(self):
        if not self._handle:
            return list(self._kinds)
        n = self._L.rt_host_scene_num_objects(self._handle)
        return [self._L.rt_host_scene_object_kind(self._handle, i) for i in range(n)]

    def arrays(self):
        """numpy views of the flattened scene (for tests)."""
        d = self.desc

        def view(ptr, n, ctype, dtype):
            if not ptr or n == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).copy().astype(dtype)

        nt = int(d.n_triangles)
        return dict(
            background=np.array(list(d.background)), cam_eye=np.array(list(d.cam_eye)), cam_lower_left=np.array(list(d.cam_lower_left)),
            cam_horizontal=np.array(list(d.cam_horizontal)), cam_vertical=np.array(list(d.cam_vertical)),
            mat_type=view(d.mat_type, d.n_materials, C.c_int32, np.int32), mat_rgb=view(d.mat_rgb, 3 * d.n_materials, C.c_double, np.float64).reshape(-1, 3),
            mat_param=view(d.mat_param, d.n_materials, C.c_double, np.float64),
            sphere_center=view(d.sphere_center, 3 * d.n_spheres, C.c_double, np.float64).reshape(-1, 3),
            sphere_radius=view(d.sphere_radius, d.n_spheres, C.c_double, np.float64),
            sphere_object=view(d.sphere_object, d.n_spheres, C.c_int32, np.int32),
            plane_center=view(d.plane_center, 3 * d.n_planes, C.c_double, np.float64).reshape(-1, 3),
            plane_normal=view(d.plane_normal, 3 * d.n_planes, C.c_double, np.float64).reshape(-1, 3),
            plane_object=view(d.plane_object, d.n_planes, C.c_int32, np.int32),
            tri_vertices=view(d.tri_vertices, 9 * nt, C.c_double, np.float64).reshape(-1, 3, 3),
            tri_uv=view(d.tri_uv, 6 * nt, C.c_double, np.float64).reshape(-1, 3, 2),
            tri_material=view(d.tri_material, nt, C.c_int32, np.int32), tri_object=view(d.tri_object, nt, C.c_int32, np.int32),
            tri_index=view(d.tri_index, nt, C.c_int32, np.int32),
            light_point=view(d.light_point, 3 * d.n_lights, C.c_double, np.float64).reshape(-1, 3),
            tex_width=view(d.tex_width, d.n_textures, C.c_uint32, np.int64), tex_height=view(d.tex_height, d.n_textures, C.c_uint32, np.int64),
        )

    @classmethod
    def from_objects(cls, width, height, background, camera, objects):
        """Builds the flattened scene from python objects shaped like the YAML.
        A Mesh object carries numpy arrays: {'type':'Mesh','verts':(V,3),'tris':(T,3),
        'materials':[...], 'tri_mat':(T,) optional}.  Follows rt_host.cpp's flattening
        (Camera::new, lights of Scene::precompute)."""
        s = cls()
        L = s._L
        mats_t, mats_rgb, mats_p = [], [], []

        def add_mat(m):
            t = MAT[m["type"]]
            if t == 3:
                raise RtError("Texture materials need the YAML loader")
            rgb = m.get("albedo") or m.get("tint") or m.get("color")
            mats_t.append(t); mats_rgb.append([float(x) for x in rgb])
            mats_p.append(float(m.get("fuzziness", m.get("refraction_index", 0.0))))
            return len(mats_t) - 1

        sph_c, sph_r, sph_m, sph_o = [], [], [], []
        pl_c, pl_n, pl_m, pl_o = [], [], [], []
        tri_v, tri_m, tri_o, tri_i = [], [], [], []
        lights, kinds = [], []
        for oi, o in enumerate(objects):
            if o["type"] == "Sphere":
                sph_c.append(o["center"]); sph_r.append(o["radius"]); sph_m.append(add_mat(o["material"])); sph_o.append(oi)
                if o["material"]["type"] == "Emissive":
                    lights.append(o["center"])
                kinds.append(0)
            elif o["type"] == "Plane":
                pl_c.append(o["center"]); pl_n.append(o["normal"]); pl_m.append(add_mat(o["material"])); pl_o.append(oi)
                if o["material"]["type"] == "Emissive":
                    lights.append(o["center"])
                kinds.append(1)
            elif o["type"] == "Mesh":
                verts, tris = _arr(o["verts"], np.float64).reshape(-1, 3), _arr(o["tris"], np.int64).reshape(-1, 3)
                gm = np.array([add_mat(m) for m in o["materials"]], dtype=np.int32)
                tm = _arr(o.get("tri_mat", np.zeros(len(tris), np.int64)), np.int64)
                tv = verts[tris]  # (T,3,3)
                tri_v.append(tv); tri_m.append(gm[tm]); tri_o.append(np.full(len(tris), oi, np.int32)); tri_i.append(np.arange(len(tris), dtype=np.int32))
                em = [i for i, m in enumerate(o["materials"]) if m["type"] == "Emissive"]
                if em:
                    first = np.nonzero(np.isin(tm, em))[0]
                    if len(first) == 0:
                        raise RtError("mesh lists an Emissive material no triangle uses")
                    lights.append(list(tv[first[0]].sum(axis=0) / 3.0))
                kinds.append(2)
            else:
                raise RtError("unknown object type %r" % o["type"])
        d = _lib.SceneDesc()
        d.width, d.height = int(width), int(height)
        for k in range(3):
            d.background[k] = float(background[k])
        eye, la, up = (_arr(camera[k], np.float64) for k in ("eye", "look_at", "up"))
        oe, oh, ov, oll = (np.zeros(3) for _ in range(4))
        L.rt_host_camera(eye.ctypes.data, la.ctypes.data, up.ctypes.data, float(camera["fovy"]), d.width, d.height, oe.ctypes.data,
                         oh.ctypes.data, ov.ctypes.data, oll.ctypes.data)
        for k in range(3):
            d.cam_eye[k], d.cam_horizontal[k], d.cam_vertical[k], d.cam_lower_left[k] = oe[k], oh[k], ov[k], oll[k]

        def keep(a, dtype):
            a = _arr(a, dtype)
            s._keep.append(a)
            return a.ctypes.data if a.size else None

        d.n_materials = len(mats_t)
        d.mat_type, d.mat_rgb, d.mat_param = keep(mats_t, np.int32), keep(mats_rgb, np.float64), keep(mats_p, np.float64)
        d.mat_texture = keep(np.full(len(mats_t), -1), np.int32)
        d.n_textures = 0
        d.n_spheres = len(sph_r)
        d.sphere_center, d.sphere_radius = keep(sph_c, np.float64), keep(sph_r, np.float64)
        d.sphere_material, d.sphere_object = keep(sph_m, np.int32), keep(sph_o, np.int32)
        d.n_planes = len(pl_m)
        d.plane_center, d.plane_normal = keep(pl_c, np.float64), keep(pl_n, np.float64)
        d.plane_material, d.plane_object = keep(pl_m, np.int32), keep(pl_o, np.int32)
        nt = sum(len(t) for t in tri_m)
        d.n_triangles = nt
        if nt:
            d.tri_vertices = keep(np.concatenate(tri_v), np.float64)
            d.tri_material, d.tri_object, d.tri_index = keep(np.concatenate(tri_m), np.int32), keep(np.concatenate(tri_o), np.int32), keep(np.concatenate(tri_i), np.int32)
        d.tri_uv, d.tri_has_uv = None, None
        d.n_lights = len(lights)
        d.light_point = keep(lights, np.float64)
        s._desc, s._kinds = d, kinds
        return s


class Context:
    """One GPU.  Not thread-safe (rtamd.h)."""

    def __init__(self, device=0):
        self._L = _lib.load_library()
        h = C.c_void_p()
        if self._L.rt_context_create(int(device), C.byref(h)):
            raise RtError(self._L.rt_last_error(None).decode())
        self._h, self.device, self.scene = h, int(device), None

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_context_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise RtError(self._L.rt_last_error(self._h).decode())

    def upload(self, scene):
        d = scene.desc
        self._check(self._L.rt_scene_upload(self._h, C.byref(d)))
        self.scene = scene
        self.width, self.height = d.width, d.height
        return self

    # ---- device-pointer API (torch tensors are only the memory) -----------------
    def render_accumulate(self, params, accum, stream=None):
        """accum: torch float32 CUDA tensor (H, W, 3) on this device; adds sample sums."""
        import torch
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.numel() == self.width * self.height * 3
        assert accum.device.index == self.device
        st = stream if stream is not None else torch.cuda.current_stream(accum.device).cuda_stream
        p = params.to_c()
        self._check(self._L.rt_render_accumulate(self._h, C.byref(p), C.c_void_p(accum.data_ptr()), C.c_void_p(st)))

    def render_accumulate_noalloc(self, params, accum, stream=None):
        """rt_render_accumulate_noalloc: one pass in the library's static scratch array, no device allocation."""
        import torch
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.numel() == self.width * self.height * 3
        assert accum.device.index == self.device
        st = stream if stream is not None else torch.cuda.current_stream(accum.device).cuda_stream
        p = params.to_c()
        self._check(self._L.rt_render_accumulate_noalloc(self._h, C.byref(p), C.c_void_p(accum.data_ptr()), C.c_void_p(st)))

    def resolve(self, accum, spp_total, want_color=False, stream=None):
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(accum.device).cuda_stream
        rgb8 = torch.empty((self.height, self.width, 3), dtype=torch.uint8, device=accum.device)
        color = torch.empty((self.height, self.width, 3), dtype=torch.float32, device=accum.device) if want_color else None
        self._check(self._L.rt_resolve(self._h, C.c_void_p(accum.data_ptr()), int(spp_total), C.c_void_p(rgb8.data_ptr()),
                                       C.c_void_p(color.data_ptr()) if want_color else None, C.c_void_p(st)))
        return (rgb8, color) if want_color else rgb8

    # ---- host API ---------------------------------------------------------------
    def render_image(self, params, want_accum=True):
        """The pixel loop of compute_image: returns (accum float32 (H,W,3) sums, rgb8 (H,W,3))."""
        acc = np.zeros((self.height, self.width, 3), np.float32) if want_accum else None
        rgb = np.zeros((self.height, self.width, 3), np.uint8)
        p = params.to_c()
        self._check(self._L.rt_render_image(self._h, C.byref(p), acc.ctypes.data if want_accum else None, rgb.ctypes.data))
        return acc, rgb

    def primary_hits(self, dx=0.0, dy=0.0):
        n = self.width * self.height
        obj, tri = np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, nrm = np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
        self._check(self._L.rt_primary_hits(self._h, float(dx), float(dy), obj.ctypes.data, tri.ctypes.data, t.ctypes.data, nrm.ctypes.data))
        shp = (self.height, self.width)
        return dict(object_id=obj.reshape(shp), triangle_id=tri.reshape(shp), t=t.reshape(shp), normal=nrm.reshape(shp + (3,)))

    def primary_rays(self, dx=0.0, dy=0.0):
        """The device's own f32 primary rays at the sample position of primary_hits: (origin (3,), directions (H, W, 3))."""
        o = np.zeros(3, np.float32)
        d = np.zeros((self.height, self.width, 3), np.float32)
        self._check(self._L.rt_primary_rays(self._h, float(dx), float(dy), o.ctypes.data, d.ctypes.data))
        return o, d

    def stats(self):
        s = _lib.StatsC()
        self._check(self._L.rt_get_stats(self._h, C.byref(s)))
        per = [int(x) for x in s.rays_per_iteration]
        while per and per[-1] == 0:
            per.pop()
        return dict(rays_total=int(s.rays_total), rays_per_iteration=per, paths=int(s.paths), nodes_visited=int(s.nodes_visited),
                    prims_tested=int(s.prims_tested), queue_overflow=int(s.queue_overflow), hit_launches=int(s.hit_launches),
                    hit_ms=float(s.hit_ms), bvh_nodes=int(s.bvh_nodes), bvh_depth=int(s.bvh_depth), bvh_prims=int(s.bvh_prims),
                    bvh_build_ms=float(s.bvh_build_ms), wave_node_steps=int(s.wave_node_steps), wave_leaf_steps=int(s.wave_leaf_steps),
                    wave_refills=int(s.wave_refills), self_hit_rays=int(s.self_hit_rays), step_lanes=int(s.step_lanes))

    def stats_reset(self):
        self._check(self._L.rt_stats_reset(self._h))

    def set_bvh_builder(self, builder):
        """'host' (binned SAH, the default) or 'gpu' (linear BVH built on the device) for the next upload."""
        code = {"host": 0, "gpu": 1}.get(builder, builder)
        self._check(self._L.rt_context_set_bvh_builder(self._h, int(code)))

    def bvh_build_validate(self, boxes):
        """Builds a BVH over boxes (n x 6 float32: min xyz, max xyz) with the context's builder and
        checks it on the CPU; returns dict(nodes, depth, max_stack, build_ms)."""
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
        nn, dd, ms = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        t = C.c_double(0.0)
        self._check(self._L.rt_bvh_build_validate(self._h, b.shape[0], b.ctypes.data, C.byref(nn), C.byref(dd), C.byref(ms), C.byref(t)))
        return dict(nodes=nn.value, depth=dd.value, max_stack=ms.value, build_ms=t.value)


def _ctx_array(contexts):
    return (C.c_void_p * len(contexts))(*[c._h for c in contexts])


def upload_devices(contexts, scene):
    """rt_scene_upload_devices: one flatten + BVH build, one copy per context."""
    L = _lib.load_library()
    d = scene.desc
    if L.rt_scene_upload_devices(_ctx_array(contexts), len(contexts), C.byref(d)):
        raise RtError(L.rt_last_error(contexts[0]._h).decode())
    for c in contexts:
        c.scene, c.width, c.height = scene, d.width, d.height


def render_image_devices(contexts, params, split="spp"):
    """rt_render_image_devices: the call's samples (split="spp") or tiles (split="tiles") shared by the contexts,
    partial frames summed on the first context's GPU.  Returns (accum float32 (H,W,3), rgb8 (H,W,3))."""
    L = _lib.load_library()
    c0 = contexts[0]
    acc = np.zeros((c0.height, c0.width, 3), np.float32)
    rgb = np.zeros((c0.height, c0.width, 3), np.uint8)
    p = params.to_c()
    if L.rt_render_image_devices(_ctx_array(contexts), len(contexts), C.byref(p), 1 if split == "tiles" else 0, acc.ctypes.data, rgb.ctypes.data):
        raise RtError(L.rt_last_error(c0._h).decode())
    return acc, rgb


def accum_reduce(dst_ctx, dst_accum, src_contexts, src_accums, stream=None):
    """rt_accum_reduce on torch tensors: dst_accum += each src_accums[k] (each on its context's GPU)."""
    import torch
    L = _lib.load_library()
    st = stream if stream is not None else torch.cuda.current_stream(dst_accum.device).cuda_stream
    ptrs = (C.c_void_p * len(src_accums))(*[a.data_ptr() for a in src_accums])
    if L.rt_accum_reduce(dst_ctx._h, C.c_void_p(dst_accum.data_ptr()), _ctx_array(src_contexts), ptrs, len(src_accums), C.c_void_p(st)):
        raise RtError(L.rt_last_error(dst_ctx._h).decode())


def accum_reduce_rccl(contexts, accums):
    """rt_accum_reduce_rccl on torch tensors: accums[0] += accums[1:], ONE RCCL reduce over a communicator of the contexts' GPUs
    (one rank per GPU: the contexts must be on distinct GPUs).  The tensors must be complete (synchronise first); synchronous."""
    L = _lib.load_library()
    ptrs = (C.c_void_p * len(accums))(*[a.data_ptr() for a in accums])
    if L.rt_accum_reduce_rccl(_ctx_array(contexts), ptrs, len(accums)):
        raise RtError(L.rt_last_error(contexts[0]._h).decode())


def write_png(path, rgb8):
    L = _lib.load_library()
    a = _arr(rgb8, np.uint8)
    err = C.create_string_buffer(512)
    if L.rt_host_write_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], err, 512):
        raise RtError(err.value.decode())


def write_png16(path, color):
    """16-bit RGB PNG of a colour plane in [0, 1] (`Context.resolve(...)[1]`, or gamma'd sample means):
    (65535.999 * c) as u16, the 16-bit analogue of image.rs:37-43.  No reference counterpart."""
    L = _lib.load_library()
    c = np.asarray(color, dtype=np.float64)
    if c.ndim != 3 or c.shape[2] != 3:
        raise ValueError("expected an H x W x 3 array")
    q = np.nan_to_num(65535.999 * c, nan=0.0, posinf=65535.0, neginf=0.0)
    a = np.ascontiguousarray(np.clip(np.trunc(q), 0.0, 65535.0).astype(np.uint16))
    err = C.create_string_buffer(512)
    if L.rt_host_write_png16(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], err, 512):
        raise RtError(err.value.decode())


def read_png(path):
    L = _lib.load_library()
    p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    err = C.create_string_buffer(512)
    if L.rt_host_read_png(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h), err, 512):
        raise RtError(err.value.decode())
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
    finally:
        L.rt_host_free(p)


def compute_image(ssaa, depth, scene_path, output_path, base_dir=None, device=0, seed=0, devices=None, split="spp"):
    """raytracer::compute_image (raytrace.rs:108-156): same four arguments; `ssaa` may be a
    SuperSampling or the CLI string.  devices=[0, 1, ...] renders on several GPUs from this process
    (one context and host thread each; split = "spp" or "tiles")."""
    L = _lib.load_library()
    if isinstance(ssaa, SuperSampling):
        ssaa = ("jitter:%d" if ssaa.kind == "Jitter" else "grid:%d") % ssaa.resolution
    err = C.create_string_buffer(1024)
    if devices is not None:
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = L.rt_host_compute_image_devices(ssaa.encode(), int(depth), os.fsencode(scene_path), os.fsencode(output_path),
                                             os.fsencode(base_dir) if base_dir is not None else None, ids, len(devices),
                                             1 if split == "tiles" else 0, int(seed), err, 1024)
    else:
        rc = L.rt_host_compute_image(ssaa.encode(), int(depth), os.fsencode(scene_path), os.fsencode(output_path),
                                     os.fsencode(base_dir) if base_dir is not None else None, int(device), int(seed), err, 1024)
    if rc:
        msg = err.value.decode()
        if msg.startswith("error parsing ssaa argument: "):
            raise SSAADecodeError(msg[len("error parsing ssaa argument: "):])
        raise RtError(msg)
