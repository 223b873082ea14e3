"""ctypes binding of librtamd.so (include/rtamd.h + include/rtamd_host.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RT_MAX_ITERATIONS = 320
RT_NOALLOC_SCRATCH_BYTES = 64 << 20


class RtError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("background", C.c_double * 3),
        ("cam_eye", C.c_double * 3), ("cam_lower_left", C.c_double * 3), ("cam_horizontal", C.c_double * 3),
        ("cam_vertical", C.c_double * 3),
        ("n_materials", C.c_uint32), ("mat_type", C.c_void_p), ("mat_rgb", C.c_void_p), ("mat_param", C.c_void_p),
        ("mat_texture", C.c_void_p),
        ("n_textures", C.c_uint32), ("tex_width", C.c_void_p), ("tex_height", C.c_void_p), ("tex_rgb8", C.c_void_p),
        ("n_spheres", C.c_uint32), ("sphere_center", C.c_void_p), ("sphere_radius", C.c_void_p), ("sphere_material", C.c_void_p),
        ("sphere_object", C.c_void_p),
        ("n_planes", C.c_uint32), ("plane_center", C.c_void_p), ("plane_normal", C.c_void_p), ("plane_material", C.c_void_p),
        ("plane_object", C.c_void_p),
        ("n_triangles", C.c_uint64), ("tri_vertices", C.c_void_p), ("tri_uv", C.c_void_p), ("tri_has_uv", C.c_void_p),
        ("tri_material", C.c_void_p), ("tri_object", C.c_void_p), ("tri_index", C.c_void_p),
        ("n_lights", C.c_uint32), ("light_point", C.c_void_p),
    ]


class RenderParamsC(C.Structure):
    _fields_ = [
        ("ssaa_mode", C.c_int32), ("ssaa_n", C.c_uint32), ("depth", C.c_uint32), ("seed", C.c_uint64),
        ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32), ("tile_index", C.c_uint32), ("tile_count", C.c_uint32),
        ("max_paths_per_pass", C.c_uint32), ("max_shadow_generations", C.c_uint32), ("flags", C.c_uint32),
        ("sample_offset", C.c_double * 2),
    ]


class StatsC(C.Structure):
    _fields_ = [
        ("rays_total", C.c_uint64), ("rays_per_iteration", C.c_uint64 * RT_MAX_ITERATIONS), ("paths", C.c_uint64),
        ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64), ("queue_overflow", C.c_uint64),
        ("hit_launches", C.c_uint32), ("hit_ms", C.c_double),
        ("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("bvh_prims", C.c_uint32), ("bvh_build_ms", C.c_double),
        ("wave_node_steps", C.c_uint64), ("wave_leaf_steps", C.c_uint64), ("wave_refills", C.c_uint64), ("self_hit_rays", C.c_uint64),
        ("step_lanes", C.c_uint64),
    ]


# every symbol the headers declare (tests/test_host.py::test_library_exports_every_declared_symbol and tests/test_abi.py)
ABI_SYMBOLS = [
    "rt_context_create", "rt_context_destroy", "rt_last_error", "rt_scene_upload", "rt_render_accumulate", "rt_resolve",
    "rt_render_image", "rt_primary_hits", "rt_primary_rays", "rt_render_accumulate_noalloc", "rt_scene_upload_devices", "rt_accum_reduce", "rt_accum_reduce_rccl",
    "rt_render_image_devices", "rt_get_stats", "rt_stats_reset", "rt_debug_gather", "rt_context_set_bvh_builder", "rt_bvh_build_validate",
    "rt_host_scene_load", "rt_host_scene_free", "rt_host_scene_desc", "rt_host_scene_num_objects", "rt_host_scene_object_kind",
    "rt_host_ssaa_parse", "rt_host_camera", "rt_host_write_png", "rt_host_write_png16", "rt_host_read_png", "rt_host_free", "rt_host_compute_image", "rt_host_compute_image_devices", "rt_host_bvh_validate",
]


def library_path():
    # RTAMD_LIB: alternative build of the library (tuning experiments)
    return os.environ.get("RTAMD_LIB") or os.path.join(_HERE, "librtamd.so")


def load_library():
    """Loads the HIP library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RtError("%s is missing: build it with `make -C %s` (python __graft_entry__.py build); there is no CPU fallback"
                      % (path, os.path.join(_HERE, "csrc")))
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as the
    # system one this library links to).  Importing torch first makes the loader resolve our
    # dependency to torch's copy; loading the two runtimes side by side leaves the second
    # without a GPU ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, u32, i32, u64, dbl = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64, C.c_double
    L.rt_context_create.restype = C.c_int
    L.rt_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_context_destroy.restype = None
    L.rt_context_destroy.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = [vp]
    L.rt_scene_upload.restype = C.c_int
    L.rt_scene_upload.argtypes = [vp, C.POINTER(SceneDesc)]
    L.rt_render_accumulate.restype = C.c_int
    L.rt_render_accumulate.argtypes = [vp, C.POINTER(RenderParamsC), vp, vp]
    L.rt_resolve.restype = C.c_int
    L.rt_resolve.argtypes = [vp, vp, u32, vp, vp, vp]
    L.rt_render_image.restype = C.c_int
    L.rt_render_image.argtypes = [vp, C.POINTER(RenderParamsC), vp, vp]
    L.rt_primary_hits.restype = C.c_int
    L.rt_primary_hits.argtypes = [vp, dbl, dbl, vp, vp, vp, vp]
    L.rt_primary_rays.restype = C.c_int
    L.rt_primary_rays.argtypes = [vp, dbl, dbl, vp, vp]
    L.rt_render_accumulate_noalloc.restype = C.c_int
    L.rt_render_accumulate_noalloc.argtypes = [vp, C.POINTER(RenderParamsC), vp, vp]
    L.rt_scene_upload_devices.restype = C.c_int
    L.rt_scene_upload_devices.argtypes = [C.POINTER(vp), u32, C.POINTER(SceneDesc)]
    L.rt_accum_reduce.restype = C.c_int
    L.rt_accum_reduce.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), u32, vp]
    if hasattr(L, "rt_accum_reduce_rccl") or not os.environ.get("RTAMD_LIB"):      # (an RTAMD_LIB build of an older revision may lack it: A/B runs only)
        L.rt_accum_reduce_rccl.restype = C.c_int
        L.rt_accum_reduce_rccl.argtypes = [C.POINTER(vp), C.POINTER(vp), u32]
    L.rt_render_image_devices.restype = C.c_int
    L.rt_render_image_devices.argtypes = [C.POINTER(vp), u32, C.POINTER(RenderParamsC), C.c_int, vp, vp]
    L.rt_get_stats.restype = C.c_int
    L.rt_get_stats.argtypes = [vp, C.POINTER(StatsC)]
    L.rt_debug_gather.restype = C.c_int
    L.rt_context_set_bvh_builder.restype = C.c_int
    L.rt_context_set_bvh_builder.argtypes = [vp, C.c_int]
    L.rt_bvh_build_validate.restype = C.c_int
    L.rt_bvh_build_validate.argtypes = [vp, u64, C.c_void_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(dbl)]
    L.rt_debug_gather.argtypes = [vp, u64, u32, C.POINTER(dbl)]
    L.rt_stats_reset.restype = C.c_int
    L.rt_stats_reset.argtypes = [vp]
    L.rt_host_scene_load.restype = vp
    L.rt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.rt_host_scene_free.restype = None
    L.rt_host_scene_free.argtypes = [vp]
    L.rt_host_scene_desc.restype = C.POINTER(SceneDesc)
    L.rt_host_scene_desc.argtypes = [vp]
    L.rt_host_scene_num_objects.restype = u32
    L.rt_host_scene_num_objects.argtypes = [vp]
    L.rt_host_scene_object_kind.restype = C.c_int
    L.rt_host_scene_object_kind.argtypes = [vp, u32]
    L.rt_host_ssaa_parse.restype = C.c_int
    L.rt_host_ssaa_parse.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(u32), C.c_char_p, C.c_size_t]
    L.rt_host_camera.restype = None
    L.rt_host_camera.argtypes = [vp, vp, vp, dbl, u32, u32, vp, vp, vp, vp]
    L.rt_host_write_png.restype = C.c_int
    L.rt_host_write_png.argtypes = [C.c_char_p, vp, u32, u32, C.c_char_p, C.c_size_t]
    L.rt_host_write_png16.restype = C.c_int
    L.rt_host_write_png16.argtypes = [C.c_char_p, vp, u32, u32, C.c_char_p, C.c_size_t]
    L.rt_host_read_png.restype = C.c_int
    L.rt_host_read_png.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32), C.c_char_p, C.c_size_t]
    L.rt_host_free.restype = None
    L.rt_host_free.argtypes = [vp]
    L.rt_host_bvh_validate.restype = C.c_int
    L.rt_host_bvh_validate.argtypes = [u64, vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.c_char_p, C.c_size_t]
    L.rt_host_compute_image.restype = C.c_int
    L.rt_host_compute_image_devices.restype = C.c_int
    L.rt_host_compute_image_devices.argtypes = [C.c_char_p, u32, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), u32, C.c_int, u64,
                                                C.c_char_p, C.c_size_t]
    L.rt_host_compute_image.argtypes = [C.c_char_p, u32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, u64, C.c_char_p, C.c_size_t]
    _LIB = L
    return L
