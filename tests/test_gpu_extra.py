"""GPU tests (-m gpu) of the widened surface through the real loaders and the C ABI:
`v/vt` faces + a Texture material on a mesh that is not the cube + UVs that reach u = 1 / v = 0, `all_objects: true`,
an emissive OBJ mesh as a light, the float colour plane of rt_resolve, the CLI binary end to end, and the in-library
multi-GPU path (several contexts, partial frames summed on the GPU).  Fixtures: tests/golden/extra/ (make_extra.py)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O
from test_gpu_parity import check_hits, gamma, hits_pair, psnr
from util import GOLDEN, ROOT, pkg, sub

pytestmark = pytest.mark.gpu

rt = pkg()
scenegen = sub("scenegen")


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def load_extra(ctx, name):
    path = os.path.join(GOLDEN, "extra", name + ".yaml")
    ctx.upload(rt.Scene.load(path, base_dir=GOLDEN))
    return O.OracleScene.from_yaml(path, base_dir=GOLDEN)


def test_vt_faces_texture_and_edge_clamp(ctx):
    """mesh.rs:147-157,186-215 (`v/vt` faces), materials.rs:197-253: the quad is seen from its back, where the texture
    emits its texel, so the image is deterministic: hits against the oracle, and the uniform-grid image byte by byte.
    Its UVs span [0,1]^2: near v = 0 the bilinear lookup reads the row below the last one and near u = 1 the column
    right of the last one (materials.rs:207-230); the reference would index past the edge (panic, or the next row),
    the oracle and the device clamp the index to the last texel and otherwise wrap like the reference."""
    osc = load_extra(ctx, "textured_quad")
    got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=0)
    assert (want["triangle_id"] == 0).sum() > 500 and (want["triangle_id"] == 1).sum() > 500
    check_hits(got, want)
    oob_before = O.lib().orc_texture_oob_count()
    ref = osc.render(mode="uniform", N=3, depth=2)
    assert O.lib().orc_texture_oob_count() > oob_before + 100          # the edge case is exercised
    acc, rgb = ctx.render_image(rt.RenderParams("grid:3", depth=2))
    diff = np.abs(rgb.astype(int) - ref["rgb8"].astype(int))
    assert len(np.unique(rgb.reshape(-1, 3), axis=0)) > 500             # a real gradient, not a flat colour
    # f32 bilinear weights against f64: one 8-bit level, and a sample may flip at the quad's silhouette / diagonal
    print("textured quad: %.3f %% of values differ by more than one level, max %d" % (100.0 * (diff > 1).mean(), diff.max()))
    assert (diff > 1).mean() < 2e-3
    assert np.abs(acc.astype(np.float64) - ref["sum"]).mean() < 2e-4 * 9


def test_texture_front_face_scatters_with_the_texel_as_albedo(ctx):
    """The same quad from the other side (front face: Lambertian-like scatter with albedo = texel,
    materials.rs:240-252) under a white background: statistical image parity."""
    path = os.path.join(GOLDEN, "extra", "textured_quad.yaml")
    d = O.load_scene_dict(path, base_dir=GOLDEN)
    d["camera"] = dict(d["camera"], eye=[0.3, 0.2, -5.0])
    d["background"] = [1.0, 1.0, 1.0]
    osc = O.OracleScene.from_dict(d)
    import tempfile
    text = open(path).read().replace("eye: [0.3, 0.2, 5.0]", "eye: [0.3, 0.2, -5.0]").replace("background: [0.1, 0.2, 0.3]", "background: [1.0, 1.0, 1.0]")
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(text)
    try:
        ctx.upload(rt.Scene.load(f.name, base_dir=GOLDEN))
    finally:
        os.unlink(f.name)
    check_hits(*hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=0))
    n = 32
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=3, seed=2))
    ref = osc.render(mode="jitter", N=n, depth=3, seed=9)
    p = psnr(gamma(acc.astype(np.float64), n * n), ref["color"])
    bias = abs(acc.mean() - ref["sum"].mean()) / (n * n)
    print("textured quad, front: PSNR %.2f dB, |bias| %.2e" % (p, bias))
    assert p >= 40.0 and bias <= 1e-3


def test_all_objects_key_and_reference_default(ctx):
    """scene.rs:169-170 keeps only the first `o` of a file; `all_objects: true` (additive) keeps every one.
    Emissive materials: the uniform-grid image is deterministic and must agree with the oracle to the byte."""
    for name, tris in (("two_objects", {0, 1, 2}), ("two_objects_first", {0, 1})):
        osc = load_extra(ctx, name)
        got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=0)
        check_hits(got, want)
        assert set(np.unique(got["triangle_id"])) - {-1} == tris, name
        _, rgb = ctx.render_image(rt.RenderParams("grid:2", depth=1))
        ref = osc.render(mode="uniform", N=2, depth=1)["rgb8"]
        assert np.abs(rgb.astype(int) - ref.astype(int)).max() <= 1, name
    # the second object is green: present only with the key
    load_extra(ctx, "two_objects")
    _, rgb = ctx.render_image(rt.RenderParams("grid:1", depth=1, sample_offset=(0.5, 0.5)))
    assert (rgb[..., 1] > 200).sum() > 100
    load_extra(ctx, "two_objects_first")
    _, rgb = ctx.render_image(rt.RenderParams("grid:1", depth=1, sample_offset=(0.5, 0.5)))
    assert (rgb[..., 1] > 200).sum() == 0


def test_emissive_obj_mesh_is_a_light(ctx):
    """scene.rs:129-143: a mesh with an Emissive material is a light whose sample point is the centroid of its first
    emissive triangle; raytrace.rs:64-87 sends the last bounce's shadow rays there.  Converged image against the oracle."""
    osc = load_extra(ctx, "mesh_light")
    lights = osc.lights()
    assert len(lights) == 1 and np.allclose(lights[0][0], [1.0 / 6.0, 2.4, -1.0 / 6.0])
    assert np.allclose(ctx.scene.arrays()["light_point"], [lights[0][0]])
    check_hits(*hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=1), allow_id_mismatch=1)
    n = 48
    ctx.stats_reset()
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=2, seed=4))
    st = ctx.stats()
    assert st["rays_per_iteration"][2] > 0 and st["queue_overflow"] == 0        # shadow rays were traced
    ref = osc.render(mode="jitter", N=n, depth=2, seed=31)
    ref2 = osc.render(mode="jitter", N=n, depth=2, seed=32)
    p, floor = psnr(gamma(acc.astype(np.float64), n * n), ref["color"]), psnr(ref2["color"], ref["color"])
    bias = abs(acc.mean() - ref["sum"].mean()) / (n * n)
    print("mesh light: PSNR %.2f dB (oracle seed-to-seed %.2f), |bias| %.2e, mean radiance %.3f" % (p, floor, bias, ref["sum"].mean() / (n * n)))
    assert p >= 40.0 and bias <= 1e-3


def test_resolve_colour_plane(ctx):
    """the float colour plane of rt_resolve is min(sqrt(sum / spp), 1) (raytrace.rs:144-150) and the byte
    plane its (255.999 * c) as u8 (image.rs:37-43)."""
    import torch
    load_extra(ctx, "mesh_light")
    p = rt.RenderParams("jitter:3", depth=3, seed=6)
    acc = torch.zeros((ctx.height, ctx.width, 3), dtype=torch.float32, device="cuda:0")
    ctx.render_accumulate(p, acc)
    rgb8, color = ctx.resolve(acc, 9, want_color=True)
    torch.cuda.synchronize()
    a = acc.cpu().numpy().astype(np.float64)
    want = np.minimum(np.sqrt(a / 9.0), 1.0)
    got = color.cpu().numpy()
    assert got.dtype == np.float32 and np.abs(got - want).max() < 2e-7
    assert (want == 1.0).any() and (want < 0.5).any()                         # both the clamp and the gradient are exercised
    bytes_want = np.trunc(255.999 * got.astype(np.float64)).clip(0, 255).astype(np.uint8)
    assert np.abs(rgb8.cpu().numpy().astype(int) - bytes_want.astype(int)).max() <= 1
    assert (rgb8.cpu().numpy() != bytes_want).mean() < 1e-3                    # f32 product at an integer boundary
    # NaN and negative sums: f64::min(NaN, 1.0) == 1.0 and `as u8` saturates (image.rs:37-43)
    acc.zero_()
    # (sqrt of a negative sum is NaN as well, so it ends at 1.0 too: raytrace.rs:146-149 has no lower clamp)
    acc[0, 0, 0], acc[0, 0, 1], acc[0, 0, 2] = float("nan"), -1.0, 1e30
    acc[0, 1, 0], acc[0, 1, 1], acc[0, 1, 2] = 0.0, 0.25, 1.0
    rgb8, color = ctx.resolve(acc, 1, want_color=True)
    assert rgb8[0, 0].tolist() == [255, 255, 255] and color[0, 0].tolist() == [1.0, 1.0, 1.0]
    assert rgb8[0, 1].tolist() == [0, 127, 255] and color[0, 1].tolist() == [0.0, 0.5, 1.0]


def test_cli_binary_end_to_end(ctx, tmp_path):
    """main.rs:21-26: the rtamd-render ELF with the reference's four flags renders scenes/spheres on the GPU."""
    exe = os.path.join(ROOT, "raytracer-rust_amd", "rtamd-render")
    assert os.path.exists(exe), "rtamd-render is missing: run `python __graft_entry__.py build`"
    out = tmp_path / "deep" / "dir" / "spheres.png"
    r = subprocess.run([exe, "-s", "./scenes/spheres/spheres.yaml", "-o", str(out), "-d", "5", "--ssaa", "uniform:3", "--seed", "5"],
                       cwd=GOLDEN, capture_output=True, text=True, timeout=180)   # paths resolve against the working directory, as in the reference
    assert r.returncode == 0, r.stderr
    img = rt.read_png(str(out)).astype(int)
    assert img.shape == (500, 500, 3)
    assert np.all(img == 221, axis=2).sum() > 15000              # trunc(255.999 * sqrt(0.75)): the emissive plane
    ref = tmp_path / "api.png"
    rt.compute_image("uniform:3", 5, os.path.join(GOLDEN, "scenes/spheres/spheres.yaml"), str(ref), base_dir=GOLDEN, seed=5)
    api = rt.read_png(str(ref)).astype(int)
    assert np.abs(img - api).max() <= 1 and (img != api).mean() < 1e-3     # same seed: only the order of the shadow rays' float atomics differs
    r = subprocess.run([exe, "-s", "./scenes/nope.yaml", "-o", str(out)], cwd=GOLDEN, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "nope.yaml" in r.stderr


def test_render_image_devices_sums_on_the_gpu(ctx):
    """rt_scene_upload_devices + rt_render_image_devices on three contexts (here on one GPU: the box has one): the
    frame is split between the contexts, the partial frames are summed on the first context's GPU (rt_accum_reduce)
    and resolved there.  Tile split of a scene without lights: bit-identical to one context; sample split: equal
    up to the order of the float sums."""
    sc = scenegen.scene_c5(width=240, height=136, nu=80, nv=40)
    scene = rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])
    ctx.upload(scene)
    p = rt.RenderParams("jitter:4", depth=5, seed=13)
    one, one_rgb = ctx.render_image(p)
    group = [rt.Context(0) for _ in range(3)]
    try:
        rt.upload_devices(group, scene)
        assert all(c.stats()["bvh_nodes"] == ctx.stats()["bvh_nodes"] for c in group)
        tiles, tiles_rgb = rt.render_image_devices(group, p, split="tiles")
        assert np.array_equal(tiles, one) and np.array_equal(tiles_rgb, one_rgb)
        spp, spp_rgb = rt.render_image_devices(group, p, split="spp")
        assert np.allclose(spp, one, rtol=1e-5, atol=1e-6) and np.abs(spp_rgb.astype(int) - one_rgb.astype(int)).max() <= 1
        # a sub-range and a sub-set of tiles compose: the group renders samples [3, 11) of tiles 1 mod 2
        sub_p = rt.RenderParams("jitter:4", depth=5, seed=13, sample_begin=3, sample_end=11, tile_index=1, tile_count=2)
        want, _ = ctx.render_image(sub_p)
        for split in ("tiles", "spp"):
            got, _ = rt.render_image_devices(group, sub_p, split=split)
            assert np.allclose(got, want, rtol=1e-5, atol=1e-6), split
        # more contexts than samples: the idle one contributes zeros
        few = rt.RenderParams("jitter:4", depth=5, seed=13, sample_begin=0, sample_end=2)
        want, _ = ctx.render_image(few)
        got, _ = rt.render_image_devices(group, few, split="spp")
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
        with pytest.raises(rt.RtError):
            rt.render_image_devices([group[0], group[0]], p)
    finally:
        for c in group:
            c.close()


def test_accum_reduce_on_device_tensors(ctx):
    """rt_accum_reduce: dst += sum of the sources, on the GPU, for frames that are not a multiple of 4 values."""
    import torch
    load_extra(ctx, "two_objects")                 # 80 x 48
    other = rt.Context(0)
    try:
        other.upload(ctx.scene)
        g = torch.Generator(device="cpu").manual_seed(1)
        parts = [torch.rand((ctx.height, ctx.width, 3), generator=g).cuda() for _ in range(3)]
        dst = parts[0].clone()
        rt.accum_reduce(ctx, dst, [other, other], parts[1:])
        torch.cuda.synchronize()
        assert torch.equal(dst, parts[0] + parts[1] + parts[2])
    finally:
        other.close()
    small = rt.Scene.from_objects(3, 3, [0, 0, 0], dict(eye=[0, 0, 5], look_at=[0, 0, 0], up=[0, 1, 0], fovy=40.0), [])   # 27 values
    ctx.upload(small)
    a, b = torch.arange(27, dtype=torch.float32).reshape(3, 3, 3).cuda(), torch.ones((3, 3, 3)).cuda()
    rt.accum_reduce(ctx, a, [ctx], [b])
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), torch.arange(27, dtype=torch.float32).reshape(3, 3, 3) + 1)


def test_accum_reduce_rccl_communicator(ctx):
    """rt_accum_reduce_rccl: the exchange step as ONE RCCL reduce on a communicator the library creates over the contexts' GPUs
    (librtamd.so links librccl.so).  This box has one GPU, so the communicator has one rank: creation, reuse, the collective call
    on the context's stream and the refusal of two contexts on one GPU are what runs here; with two or more GPUs the test also
    renders a frame on every GPU and compares the RCCL sum (and rt_render_image_devices, which then takes this path) with one GPU."""
    import torch
    load_extra(ctx, "two_objects")
    a = torch.rand((ctx.height, ctx.width, 3), generator=torch.Generator().manual_seed(3)).cuda()
    want = a.clone()
    torch.cuda.synchronize()
    rt.accum_reduce_rccl([ctx], [a])
    rt.accum_reduce_rccl([ctx], [a])             # the communicator is kept by the context
    assert torch.equal(a, want)
    other = rt.Context(0)
    try:
        other.upload(ctx.scene)
        with pytest.raises(rt.RtError, match="same GPU"):
            rt.accum_reduce_rccl([ctx, other], [a, want])
    finally:
        other.close()
    n = torch.cuda.device_count()
    if n < 2:
        return
    group = [ctx] + [rt.Context(k) for k in range(1, n)]
    try:
        rt.upload_devices(group, ctx.scene)
        p = rt.RenderParams("jitter:4", depth=5, seed=13)
        one, one_rgb = ctx.render_image(p)
        for split in ("spp", "tiles"):
            got, rgb = rt.render_image_devices(group, p, split=split)          # distinct GPUs: rt_accum_reduce_rccl inside
            assert np.allclose(got, one, rtol=1e-5, atol=1e-6) and np.abs(rgb.astype(int) - one_rgb.astype(int)).max() <= 1, split
        parts = [torch.full((ctx.height, ctx.width, 3), float(k + 1), device="cuda:%d" % k) for k in range(n)]
        for k in range(n):
            torch.cuda.synchronize(k)
        rt.accum_reduce_rccl(group, parts)
        assert torch.all(parts[0] == n * (n + 1) / 2) and all(torch.all(parts[k] == k + 1) for k in range(1, n))
    finally:
        for c in group[1:]:
            c.close()


@pytest.mark.parametrize("ranks,workload,scaling", [(2, "c5", "weak"), (3, "c4", "strong"), (8, "c5", "weak")])
def test_bench_multi_rank_control_flow(ranks, workload, scaling, tmp_path):
    """bench.py's N > 1 branch (process group, per-rank partition, reduce to rank 0, max-over-ranks timing, summed ray
    counts) executed end to end with several ranks sharing this box's one GPU: RTAMD_BENCH_REHEARSAL=gloo swaps RCCL for
    gloo on host copies, everything else is the code the driver's multi-GPU runs execute."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RTAMD_BENCH_REHEARSAL="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--small",
                        "--workload", workload, "--spp-per-step", "64"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1                                     # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == ranks and j["scaling"] == scaling and j["value"] > 0 and "REHEARSAL" in j["config"]["workload"]
    w, h = (320, 180) if workload == "c5" else (384, 216)
    per_step = w * h * 64 * (ranks if scaling == "weak" else 1)          # weak: every rank renders the frame; strong: the ranks share it
    assert j["config"]["paths_per_step"] == per_step
    assert "cpu_baseline" not in j and "other_workloads" not in j         # N = 1 only
    # what a scaling loss would be attributed with: the ranks' own times per step (before the barrier) and the exchange step
    assert 0 < j["ranks"]["ms_per_step_min"] <= j["ranks"]["ms_per_step_max"] <= j["ms_per_step"] * 1.05
    assert j["ranks"]["reduce_bytes"] == w * h * 12


def test_c_abi_host_code_under_asan():
    """tests/abi_asan_driver.cpp: every entry point of include/rtamd.h from a plain C++ host, the library's host code
    instrumented with AddressSanitizer + UBSan (built by __graft_entry__.build() / `make -C raytracer-rust_amd/csrc asan`).
    The driver cross-checks host call == device-pointer call == no-alloc call == two contexts, re-uploads scenes of different
    sizes and walks the error paths."""
    exe = os.path.join(ROOT, "tests", "_build", "abi_asan")
    assert os.path.exists(exe), "tests/_build/abi_asan is missing: run __graft_entry__.build()"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")   # (leaks: the HIP runtime keeps its own at exit)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "all checks passed" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]


def test_tile_entry_points_change_nothing():
    """k_tile_entry: the primary rays of an 8x8 tile start their traversal at the tile's entry points into the BVH instead of
    at the root.  The entry points must lose no hit: ids, t and normal of every primary ray, and whole renders of scenes without
    lights (deterministic per seed), are identical with the entry points switched off (RTAMD_TILE_ENTRY=0: every ray starts at the
    root) -- on the reference's cube, spheres behind and beside each other (C3), the torus seen through its hole with the plane
    behind it (C4) and both together (C5), at sub-pixel offsets at the edges of the pixels, odd image sizes and a tile split."""
    def contexts():
        os.environ["RTAMD_TILE_ENTRY"] = "0"
        try:
            off = rt.Context(0)
        finally:
            del os.environ["RTAMD_TILE_ENTRY"]
        return off, rt.Context(0)

    off, on = contexts()
    try:
        cube = rt.Scene.load(os.path.join(GOLDEN, "scenes", "cube", "cube.yaml"), base_dir=GOLDEN)
        scenes = [("cube", cube)]
        for name, sc in (("c3", scenegen.scene_c3(323, 181)), ("c4", scenegen.scene_c4(387, 213, 120, 60)), ("c5", scenegen.scene_c5(645, 363, 200, 100))):
            scenes.append((name, rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])))
        for name, scene in scenes:
            off.upload(scene)
            on.upload(scene)
            for dx, dy in ((0.0, 0.0), (0.5, 0.5), (0.999, 0.001), (0.001, 0.999)):
                a, b = off.primary_hits(dx, dy), on.primary_hits(dx, dy)
                for k in ("object_id", "triangle_id", "t", "normal"):
                    assert np.array_equal(a[k], b[k]), "%s: %s differs at offset (%g, %g) in %d pixels" % (name, k, dx, dy, (a[k] != b[k]).sum())
            if name == "cube":
                continue          # (two lights: shadow rays of one path add up in an order that is not fixed)
            for kw in (dict(), dict(tile_index=1, tile_count=3)):
                p = rt.RenderParams("jitter:3", depth=4, seed=77, **kw)
                acc_a, _ = off.render_image(p)
                acc_b, _ = on.render_image(p)
                assert np.array_equal(acc_a, acc_b), "%s: renders differ in %d values" % (name, (acc_a != acc_b).sum())
        # corner cases of the pyramid / box test and of the distance bound: the eye inside a big sphere (no bound from it), a plane
        # behind the eye and one in front, a very wide field of view (tiles that span tens of degrees), a one-tile image, sample
        # offsets far outside the pixel (the hit-id gate allows any offset)
        from test_gpu_parity import _box, _mesh
        lam = lambda c: dict(type="Lambertian", albedo=c)
        inside = [dict(type="Sphere", center=[0.2, 0.1, 0.3], radius=60.0, material=lam([0.5, 0.5, 0.5])), _mesh(_box([-1.0, -1.0, -6.0], [1.0, 0.5, -4.0]), lam([0.6, 0.2, 0.2]))] + \
                 [dict(type="Sphere", center=[1.7 * np.cos(k), 0.8 * np.sin(2.0 * k), -5.0 + 1.5 * np.sin(k)], radius=0.35, material=lam([0.2, 0.6, 0.3])) for k in range(12)]
        planes = [dict(type="Plane", center=[0.0, 0.0, 3.0], normal=[0.0, 0.0, -1.0], material=lam([0.3, 0.3, 0.3])),
                  dict(type="Plane", center=[0.0, -1.2, 0.0], normal=[0.0, 1.0, 0.0], material=lam([0.4, 0.4, 0.4])),
                  _mesh(_box([-0.8, -1.2, -5.0], [0.6, 0.3, -3.5]), lam([0.6, 0.6, 0.2])), _mesh(_box([0.9, -1.2, -9.0], [2.1, 1.0, -7.5]), lam([0.2, 0.2, 0.7])),
                  dict(type="Sphere", center=[-1.5, -0.7, -4.0], radius=0.5, material=lam([0.7, 0.3, 0.3]))]
        cases = [("eye inside a big sphere", inside, dict(eye=[0.0, 0.0, 0.0], look_at=[0.0, 0.0, -5.0], up=[0.0, 1.0, 0.0], fovy=50.0), 67, 45),
                 ("planes behind and below", planes, dict(eye=[0.0, 0.0, 0.0], look_at=[0.0, -0.2, -5.0], up=[0.0, 1.0, 0.0], fovy=60.0), 61, 37),
                 ("wide field of view", planes, dict(eye=[0.0, 0.0, 0.0], look_at=[0.0, -0.2, -5.0], up=[0.0, 1.0, 0.0], fovy=170.0), 17, 9),
                 ("one tile", inside, dict(eye=[0.0, 0.0, 0.0], look_at=[0.3, 0.1, -5.0], up=[0.0, 1.0, 0.0], fovy=35.0), 5, 3)]
        for name, objs, cam, w, h in cases:
            scene = rt.Scene.from_objects(w, h, [0.5, 0.6, 0.7], cam, objs)
            off.upload(scene); on.upload(scene)
            for dx, dy in ((0.5, 0.5), (0.0, 0.999), (-3.5, 7.25)):
                a, b = off.primary_hits(dx, dy), on.primary_hits(dx, dy)
                for k in ("object_id", "triangle_id", "t", "normal"):
                    assert np.array_equal(a[k], b[k]), "%s: %s differs at offset (%g, %g) in %d pixels" % (name, k, dx, dy, (a[k] != b[k]).sum())
                assert (dx, dy) != (0.5, 0.5) or (a["object_id"] >= 0).any(), name
            p = rt.RenderParams("jitter:3", depth=3, seed=5)
            acc_a, _ = off.render_image(p)
            acc_b, _ = on.render_image(p)
            assert np.array_equal(acc_a, acc_b), "%s: renders differ in %d values" % (name, (acc_a != acc_b).sum())
        scene = scenes[-1][1]
        off.upload(scene); on.upload(scene)
        # what the entry points save: node visits of the primary rays
        on.stats_reset(); off.stats_reset()
        p = rt.RenderParams("jitter:2", depth=1, seed=5, counters=True)
        off.render_image(p); on.render_image(p)
        n_off, n_on = off.stats()["nodes_visited"], on.stats()["nodes_visited"]
        print("nodes visited by the primary rays of the last scene: %d at the root, %d at the tiles' entry points" % (n_off, n_on))
        assert n_on < n_off
    finally:
        off.close(); on.close()


def test_last_rays_that_stop_at_their_first_hit_change_nothing():
    """In a scene without emitters (no Emissive / Texture material, hence no lights) the last ray of a path contributes the
    background if it escapes and nothing otherwise (raytrace.rs:52-54,95), so k_closest_hit ends its traversal at the first hit
    and k_shade gathers nothing for it.  Renders are identical, value by value, with that switched off (RTAMD_ANY_HIT=0), at
    depth 1 (the primary rays are the last ones), 2 and 5, on spheres (C3), the mesh over a plane (C4) and both (C5); a scene
    WITH emitters (scenes/spheres: Emissive plane and sphere) does not take the path at all."""
    os.environ["RTAMD_ANY_HIT"] = "0"
    try:
        off = rt.Context(0)
    finally:
        del os.environ["RTAMD_ANY_HIT"]
    on = rt.Context(0)
    try:
        for name, sc in (("c3", scenegen.scene_c3(323, 181)), ("c4", scenegen.scene_c4(387, 213, 120, 60)), ("c5", scenegen.scene_c5(645, 363, 200, 100))):
            scene = rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])
            off.upload(scene); on.upload(scene)
            for depth in (1, 2, 5):
                p = rt.RenderParams("jitter:3", depth=depth, seed=31 + depth, counters=True)
                off.stats_reset(); on.stats_reset()
                a, _ = off.render_image(p)
                b, _ = on.render_image(p)
                assert np.array_equal(a, b), "%s depth %d: renders differ in %d values" % (name, depth, (a != b).sum())
                so, sn = off.stats(), on.stats()
                assert so["rays_total"] == sn["rays_total"] and so["rays_per_iteration"] == sn["rays_per_iteration"]
                assert sn["nodes_visited"] < so["nodes_visited"], (name, depth)
        golden = os.path.join(GOLDEN, "scenes", "spheres", "spheres.yaml")
        off.upload(rt.Scene.load(golden, base_dir=GOLDEN)); on.upload(rt.Scene.load(golden, base_dir=GOLDEN))
        p = rt.RenderParams("uniform:2", depth=3, seed=3, counters=True, max_shadow_generations=1)
        off.stats_reset(); on.stats_reset()
        off.render_image(p); on.render_image(p)
        assert off.stats()["nodes_visited"] == on.stats()["nodes_visited"]
    finally:
        off.close(); on.close()


def test_primary_rays_formed_in_iteration_0_change_nothing():
    """RTAMD_GEN_PRIMARY=1 (off by default: measured slower): without k_raygen, k_closest_hit and k_shade of iteration 0 form
    the primary ray of a path slot themselves (primary_ray, the function k_raygen calls).  Scenes without triangles: the two renders are identical value by value (both kernels see the very f32 direction
    k_raygen would have written).  Scenes with triangles: k_shade works on the same exact f64 direction either way, the traversal
    ranks its candidates along the fast-normalised direction (one f32 ulp from the truncated exact one at most), so a ray within an
    ulp of an edge may land on the neighbour: all but a few pixels in 100,000 are identical, and the means agree to 1e-5."""
    os.environ["RTAMD_GEN_PRIMARY"] = "1"
    try:
        on = rt.Context(0)
    finally:
        del os.environ["RTAMD_GEN_PRIMARY"]
    off = rt.Context(0)
    try:
        for name, sc in (("c3", scenegen.scene_c3(323, 181)), ("c4", scenegen.scene_c4(387, 213, 120, 60)), ("c5", scenegen.scene_c5(645, 363, 200, 100))):
            scene = rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])
            off.upload(scene); on.upload(scene)
            for kw in (dict(ssaa="jitter:3", depth=4), dict(ssaa="uniform:2", depth=1), dict(ssaa="jitter:2", depth=3, tile_index=2, tile_count=5),
                       dict(ssaa="jitter:4", depth=2, sample_begin=3, sample_end=11, max_paths_per_pass=64 * 700)):
                p = rt.RenderParams(kw.pop("ssaa"), seed=9, **kw)
                off.stats_reset(); on.stats_reset()
                a, _ = off.render_image(p)
                b, _ = on.render_image(p)
                assert off.stats()["paths"] == on.stats()["paths"] and off.stats()["rays_per_iteration"][0] == on.stats()["rays_per_iteration"][0]
                differ = (a != b).any(axis=2).mean()
                print("%s %s: %.2e of the pixels differ" % (name, kw, differ))
                if name == "c3":
                    assert differ == 0.0
                else:
                    assert differ < 2e-4 and abs(a.mean(dtype=np.float64) - b.mean(dtype=np.float64)) <= 1e-5 * a.mean(dtype=np.float64)
    finally:
        off.close(); on.close()


def test_shading_in_two_launches_changes_nothing():
    """k_shade, CLS (RTAMD_SHADE_SPLIT=1; off by default: measured slower): in scenes with triangles the bounces between the first and the last are shaded in two launches -- everything
    but the hits on triangles and the hits with a runner-up to decide, then those.  The rays a path emits, its random numbers and
    its radiance word do not depend on the launch that shades it: renders without lights are identical value by value with one
    launch per iteration (the default) -- C4, C5, several passes, a tile split, and a scene of coplanar surfaces (two glass
    boxes standing exactly on the ground plane: runner-ups, re-hits with a coplanar partner, exact directions)."""
    from test_gpu_parity import _box, _mesh
    os.environ["RTAMD_SHADE_SPLIT"] = "1"
    try:
        on = rt.Context(0)
    finally:
        del os.environ["RTAMD_SHADE_SPLIT"]
    off = rt.Context(0)
    try:
        gy = -0.4371
        glass = lambda ior, tint: dict(type="Dieletrics", tint=tint, refraction_index=ior)
        boxes = dict(width=96, height=72, background=[0.7, 0.8, 0.9], camera=dict(eye=[0.3, 3.1, 4.9], look_at=[0.0, -0.2, -0.3], up=[0.0, 1.0, 0.0], fovy=35.0),
                     objects=[_mesh(_box([-1.31, gy, -1.07], [0.42, 0.38, 0.51]), glass(1.31, [0.9, 0.8, 0.7])),
                              _mesh(_box([-0.37, gy, -0.63], [1.23, 0.71, 0.93]), glass(1.52, [0.7, 0.9, 0.8])),
                              _mesh(_box([0.5, 0.71, -0.5], [1.0, 1.25, 0.25]), dict(type="Metal", albedo=[0.9, 0.9, 0.9], fuzziness=0.0)),
                              dict(type="Sphere", center=[-0.4, 0.9, 0.1], radius=0.5, material=dict(type="Lambertian", albedo=[0.5, 0.3, 0.2])),
                              dict(type="Plane", center=[0.0, gy, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Lambertian", albedo=[0.6, 0.6, 0.6]))])
        for name, sc in (("boxes", boxes), ("c4", scenegen.scene_c4(387, 213, 120, 60)), ("c5", scenegen.scene_c5(645, 363, 200, 100))):
            scene = rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])
            off.upload(scene); on.upload(scene)
            for kw in (dict(ssaa="jitter:4", depth=6), dict(ssaa="jitter:3", depth=3, max_paths_per_pass=64 * 500), dict(ssaa="jitter:2", depth=8, tile_index=1, tile_count=3),
                       dict(ssaa="jitter:2", depth=2)):
                p = rt.RenderParams(kw.pop("ssaa"), seed=17, **kw)
                off.stats_reset(); on.stats_reset()
                a, _ = off.render_image(p)
                b, _ = on.render_image(p)
                assert np.array_equal(a, b), "%s %s: renders differ in %d values" % (name, kw, (a != b).sum())
                assert off.stats()["rays_per_iteration"] == on.stats()["rays_per_iteration"]
    finally:
        off.close(); on.close()
