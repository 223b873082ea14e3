"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same scenes.

Bars (BASELINE.json north_star):
  * primary-ray closest-hit object / triangle ids: bit-exact on the reference's fixtures
  * hit distance and normal: within 1e-5 relative
  * converged images: PSNR >= 40 dB against the oracle at matched high spp (RNG streams differ)
plus size-independent properties (tile/sample-split linearity, determinism, ray accounting)
and the edge cases (depth 0, empty scene, odd image sizes, single primitives)."""
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from util import GOLDEN, pkg, sub

pytestmark = pytest.mark.gpu

rt = pkg()
scenegen = sub("scenegen")

REL_TOL = 1e-5   # north_star: hit distance and normal within 1e-5 relative
from util import PARITY   # measured parity figures, written to gpurun_out/parity_record.json at the end of the session (conftest.py)


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def load_pair(name, ctx, width=None, height=None):
    path = os.path.join(GOLDEN, "scenes", name, name + ".yaml")
    d = O.load_scene_dict(path, base_dir=GOLDEN)
    if width is None:
        ctx.upload(rt.Scene.load(path, base_dir=GOLDEN))
        return O.OracleScene.from_dict(d)
    # resized variant: rewrite the YAML header only
    import tempfile
    text = open(path).read().replace("width: 500", "width: %d" % width).replace("height: 500", "height: %d" % height)
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(text)
    try:
        ctx.upload(rt.Scene.load(f.name, base_dir=GOLDEN))
    finally:
        os.unlink(f.name)
    d["width"], d["height"] = width, height
    return O.OracleScene.from_dict(d)


def check_hits(got, want, allow_id_mismatch=0):
    """ids: exact (up to allow_id_mismatch); t and normal: within REL_TOL relative on EVERY hit, no allowance."""
    id_bad = (got["object_id"] != want["object_id"]) | (got["triangle_id"] != want["triangle_id"])
    assert id_bad.sum() <= allow_id_mismatch, "%d hit ids differ from the oracle" % id_bad.sum()
    ok = (want["object_id"] >= 0) & ~id_bad
    assert np.all(got["t"][(want["object_id"] < 0) & ~id_bad] == -1.0)
    if not ok.any():
        return int(id_bad.sum()), 0.0, 0.0
    rel_t = np.abs(got["t"][ok].astype(np.float64) - want["t"][ok]) / np.abs(want["t"][ok])
    nerr = np.linalg.norm(got["normal"][ok].astype(np.float64) - want["normal"][ok], axis=-1) / np.linalg.norm(want["normal"][ok], axis=-1)
    assert rel_t.max() < REL_TOL, "max relative t error %.3e" % rel_t.max()
    assert nerr.max() < REL_TOL, "max relative normal error %.3e" % nerr.max()
    return int(id_bad.sum()), float(rel_t.max()), float(nerr.max())


def rays_through_a_mesh(got, want):
    """Rays that hit a triangle in both worlds but at distances more than 1e-3 apart: the device's ray has passed between two
    triangles of a closed mesh (or has been caught by one it misses) -- the cracks that f32 edge tests leave (k_closest_hit's
    margin, RT_EDGE_TOL, closes them)."""
    both = (got["triangle_id"] >= 0) & (want["triangle_id"] >= 0) & (got["triangle_id"] != want["triangle_id"])
    return int((both & (np.abs(got["t"].astype(np.float64) - want["t"]) > 1e-3 * np.abs(want["t"]))).sum())


def oracle_camera_dirs(osc, dx, dy, crop=None):
    """camera.rs:62-67 + raytrace.rs:32-37 in numpy f64 from the oracle's camera: unit direction per pixel."""
    cam = osc.camera()
    x0, x1, r0, r1 = crop if crop is not None else (0, osc.width, 0, osc.height)
    xs = np.arange(x0, x1, dtype=np.float64)[None, :, None] + dx
    ys = (osc.height - 1 - np.arange(r0, r1, dtype=np.float64))[:, None, None] + dy          # raytrace.rs:128: row 0 is j = height-1
    d = cam["lower_left"] + cam["horizontal"] * xs + cam["vertical"] * ys - cam["eye"]
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def hits_pair(ctx, osc, dx=0.0, dy=0.0, crop=None, own_camera_id_slack=None):
    """(device hits, oracle hits) for one primary ray per pixel at sub-pixel offset (dx, dy).

    The device works in f32 and the reference in f64, so the comparison is made in two links that are each strict:
      1. the device's rays (rt_primary_rays) equal the oracle camera's rays (camera.rs:62-67) to f32 rounding
         (|delta| <= 1e-7 per component) -- this pins Camera::spawn_ray / Ray::new;
      2. the oracle is fed those very f32 rays (orc_closest_hits), so t and the normal of every hit can be held to
         1e-5 with no allowance: an f32-rounded direction (4e-8 rad) is otherwise amplified by 1/cos(incidence) at
         sphere silhouettes, which is a property of the ray, not of the intersection code.
    own_camera_id_slack: also compare the ids with the oracle traced along its OWN f64 rays (at most that many
    silhouette pixels may differ)."""
    got = ctx.primary_hits(dx, dy)
    o, d = ctx.primary_rays(dx, dy)
    assert np.array_equal(o.astype(np.float64), osc.camera()["eye"].astype(np.float32).astype(np.float64))
    if crop is not None:
        x0, x1, r0, r1 = crop
        got = {k: v[r0:r1, x0:x1] for k, v in got.items()}
        d = d[r0:r1, x0:x1]
    ref_d = oracle_camera_dirs(osc, dx, dy, crop)
    assert np.abs(d.astype(np.float64) - ref_d).max() <= 1e-7, "device rays differ from camera.rs:62-67 by %.3e" % np.abs(d - ref_d).max()
    # origin: the reference's f64 eye, which the device's f64 hit-point arithmetic uses too (its f32 rounding, the `o`
    # above, only steers the traversal)
    want = osc.closest_hits(osc.camera()["eye"], d.astype(np.float64))
    if own_camera_id_slack is not None:
        own = osc.closest_hits(osc.camera()["eye"], ref_d)      # (threaded; osc.primary_hits is the same thing on one core)
        bad = ((got["object_id"] != own["object_id"]) | (got["triangle_id"] != own["triangle_id"])).sum()
        assert bad <= own_camera_id_slack, "%d ids differ from the oracle traced along its own f64 rays" % bad
    return got, want


# ------------------------------------------------------------------ C2: the hit-id gate
def test_c2_cube_primary_hits_bit_exact(ctx):
    osc = load_pair("cube", ctx)
    got = ctx.primary_hits(0.0, 0.0)
    want = osc.primary_hits(0.0, 0.0)                  # the oracle along its own f64 rays: the gate as BASELINE.json states it
    assert (want["triangle_id"] >= 0).sum() > 10000    # the cube is in view
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=0)
    got2, want2 = hits_pair(ctx, osc, 0.0, 0.0, own_camera_id_slack=0)      # and along the device's f32 rays
    check_hits(got2, want2, allow_id_mismatch=0)
    print("C2: ids exact, max rel t %.2e, normal %.2e" % (rt_, rn))
    PARITY["primary_hits_c2_cube"] = dict(rays=got["t"].size, id_mismatches=bad, max_rel_t=rt_, max_rel_normal=rn)


@pytest.mark.parametrize("off", [(0.5, 0.5), (0.25, 0.75)])
def test_cube_primary_hits_other_offsets(ctx, off):
    osc = load_pair("cube", ctx)
    check_hits(ctx.primary_hits(*off), osc.primary_hits(*off), allow_id_mismatch=0)
    check_hits(*hits_pair(ctx, osc, *off, own_camera_id_slack=0), allow_id_mismatch=0)


def test_c1_spheres_primary_hits(ctx):
    osc = load_pair("spheres", ctx)
    check_hits(*hits_pair(ctx, osc, own_camera_id_slack=0), allow_id_mismatch=0)


def test_c3_rtiow_primary_hits(ctx):
    sc = scenegen.scene_c3(width=480, height=270)
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    # silhouette rays may flip between the f32 rays and the oracle's own f64 rays: allow a handful there
    got, want = hits_pair(ctx, osc, own_camera_id_slack=8)
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=2)
    print("C3 480x270: %d id mismatches of %d, max rel t %.2e, normal %.2e" % (bad, got["t"].size, rt_, rn))


def test_c4_torus_primary_hits(ctx):
    sc = scenegen.scene_c4(width=192, height=108, nu=100, nv=50)    # 10,000 triangles: the oracle is brute force
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    got, want = hits_pair(ctx, osc, own_camera_id_slack=4)
    assert (want["triangle_id"] >= 0).sum() > 2000
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=2)
    print("C4 (10k tris) 192x108: %d id mismatches, max rel t %.2e, normal %.2e" % (bad, rt_, rn))


def test_c5_combined_primary_hits(ctx):
    sc = scenegen.scene_c5(width=240, height=135, nu=80, nv=40)     # 485 spheres + 6,400 triangles
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    got, want = hits_pair(ctx, osc, own_camera_id_slack=6)
    assert (want["triangle_id"] >= 0).sum() > 500 and (want["object_id"] == 0).sum() > 1000    # torus and ground in view
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=2)
    print("C5 (6.4k tris + spheres) 240x135: %d id mismatches, max rel t %.2e, normal %.2e" % (bad, rt_, rn))


def test_deep_traversal_stack_spills_correctly(ctx):
    """4,096 stacked parallel triangles pierced by every ray: all four children of every node are hit,
    so the per-lane stack outgrows its LDS part (16 entries) and uses the global overflow area."""
    n = 4096
    z = -np.arange(n, dtype=np.float64) * 0.01
    base = np.array([[-3.0, -3.0, 0.0], [3.0, -3.0, 0.0], [0.0, 4.0, 0.0]])
    verts = (base[None, :, :] + np.stack([np.zeros(n), np.zeros(n), z], axis=1)[:, None, :]).reshape(-1, 3)
    tris = np.arange(3 * n).reshape(n, 3)
    order = np.random.default_rng(1).permutation(n)           # file order != depth order
    objs = [dict(type="Mesh", verts=verts, tris=tris[order], materials=[dict(type="Emissive", color=[1, 1, 1])])]
    osc = small_scene(ctx, objs, 48, 48)
    got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=0)
    assert (want["triangle_id"] >= 0).sum() > 500
    check_hits(got, want)
    assert np.all(want["triangle_id"][want["triangle_id"] >= 0] == int(np.nonzero(order == 0)[0][0]))


def test_axis_parallel_rays_do_not_walk_the_tree(ctx):
    """Rays with a zero or tiny direction component (here: the centre column and row of an on-axis
    camera, d = (0, 0, -1) in the middle) have a huge 1/d on that axis.  The rounding bound of the
    node test is per axis, so they are culled like any other ray: same hits as the oracle, and the
    nodes visited per ray stay what a BVH walk costs (one bound shared by the three axes made such
    rays visit most of the tree)."""
    n = 96
    # grid lines off the pixel centres and off the cell diagonals (no rays through shared edges)
    X, Y = np.meshgrid(np.linspace(-2.013, 1.987, n + 1), np.linspace(-2.0071, 1.9929, n + 1), indexing="ij")
    Z = 0.05 * np.sin(3.0 * X) * np.cos(2.0 * Y)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    idx = lambda i, j: i * (n + 1) + j
    tris = np.array([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(n) for j in range(n)] +
                    [[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(n) for j in range(n)])
    objs = [dict(type="Mesh", verts=verts, tris=tris, materials=[dict(type="Emissive", color=[1, 1, 1])])]
    osc = small_scene(ctx, objs, 65, 65)
    got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=2)
    assert want["triangle_id"][32, 32] >= 0
    check_hits(got, want, allow_id_mismatch=2)    # rays through shared edges / vertices of the grid
    # a one-pixel-wide image: EVERY ray has d.x == 0 (measured with the shared bound: 98 nodes and
    # 189 triangles per ray; per axis: 9 and 2; with the tiles' entry points, whose leaves a ray tests without
    # a box test of its own: 5.6 and 7.1 -- a pixel of this image spans several triangles)
    for (w, h) in ((1, 65), (65, 1)):
        small_scene(ctx, objs, w, h)
        ctx.stats_reset()
        ctx.render_image(rt.RenderParams("grid:1", depth=1, counters=True, sample_offset=(0.5, 0.5)))
        st = ctx.stats()
        assert st["rays_total"] == 65
        assert st["nodes_visited"] / st["rays_total"] < 20.0 and st["prims_tested"] / st["rays_total"] < 10.0, \
            (st["nodes_visited"] / st["rays_total"], st["prims_tested"] / st["rays_total"])


# ------------------------------------------------------------------ images
def psnr(a, b):
    mse = np.mean((a - b) ** 2)
    return 10.0 * np.log10(1.0 / mse)


def gamma(sum_img, spp):
    return np.minimum(np.sqrt(sum_img / spp), 1.0)


@pytest.mark.parametrize("name,n", [("spheres", 40), ("cube", 40)])
def test_converged_image_parity(ctx, name, n):
    """PSNR >= 40 dB between the GPU render and the f64 oracle at matched spp (n*n) on a
    128x128 version of the reference scene; also no systematic bias."""
    osc = load_pair(name, ctx, 128, 128)
    spp = n * n
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=5, seed=3))
    ref = osc.render(mode="jitter", N=n, depth=5, seed=17)
    g, o = gamma(acc.astype(np.float64), spp), ref["color"]
    p = psnr(g, o)
    bias = abs(np.mean(acc / spp) - np.mean(ref["sum"] / spp))
    print("%s: PSNR %.2f dB at %d spp, |mean radiance bias| %.2e" % (name, p, spp, bias))
    PARITY.setdefault("converged_images", {})["%s 128x128 %d spp" % (name, spp)] = dict(psnr_db=p, abs_bias=bias)
    assert p >= 40.0
    assert bias <= 1e-3


def test_c3_image_parity_small(ctx):
    sc = scenegen.scene_c3(width=160, height=90)
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    n = 32
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=5, seed=5))
    ref = osc.render(mode="jitter", N=n, depth=5, seed=23)
    p = psnr(gamma(acc.astype(np.float64), n * n), ref["color"])
    bias = abs(np.mean(acc) - np.mean(ref["sum"])) / (n * n)
    assert bias <= 1e-3, bias
    print("C3 160x90: PSNR %.2f dB at %d spp" % (p, n * n))
    PARITY.setdefault("converged_images", {})["c3 160x90 %d spp" % (n * n)] = dict(psnr_db=p, abs_bias=bias)
    assert p >= 40.0


def test_c3_image_parity_depth_50(ctx):
    """SURVEY 8d reports C3 at the reference's default depth 5 and at RTIOW's 50: long glass / metal paths."""
    sc = scenegen.scene_c3(width=128, height=72)
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    n = 32
    ctx.stats_reset()
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=50, seed=5))
    st = ctx.stats()
    assert len(st["rays_per_iteration"]) > 30                     # paths of more than 30 segments exist
    ref = osc.render(mode="jitter", N=n, depth=50, seed=23)
    p = psnr(gamma(acc.astype(np.float64), n * n), ref["color"])
    bias = abs(np.mean(acc) - np.mean(ref["sum"])) / (n * n)
    print("C3 128x72 depth 50: PSNR %.2f dB at %d spp, |bias| %.2e, %.2f rays per path (oracle %.2f)"
          % (p, n * n, bias, st["rays_total"] / st["paths"], ref["rays"] / (128 * 72 * n * n)))
    PARITY.setdefault("converged_images", {})["c3 128x72 depth 50 %d spp" % (n * n)] = dict(psnr_db=p, abs_bias=bias, rays_per_path=st["rays_total"] / st["paths"], oracle_rays_per_path=ref["rays"] / (128 * 72 * n * n))
    assert p >= 40.0 and bias <= 1e-3
    assert abs(st["rays_total"] / st["paths"] - ref["rays"] / (128 * 72 * n * n)) < 0.01


def test_c5_image_parity_small(ctx):
    """Lambertian mesh + spheres: exercises the triangle self-hit quirk on a curved mesh."""
    sc = scenegen.scene_c5(width=128, height=72, nu=60, nv=30)
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    n = 20
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=5, seed=5))
    ref = osc.render(mode="jitter", N=n, depth=5, seed=23)
    p = psnr(gamma(acc.astype(np.float64), n * n), ref["color"])
    ref2 = osc.render(mode="jitter", N=n, depth=5, seed=24)
    print("C5 128x72 (3.6k tris): PSNR %.2f dB at %d spp (oracle seed-to-seed floor %.2f dB)" % (p, n * n, psnr(ref2["color"], ref["color"])))
    PARITY.setdefault("converged_images", {})["c5 128x72 3.6k tris %d spp" % (n * n)] = dict(psnr_db=p, oracle_seed_to_seed_db=psnr(ref2["color"], ref["color"]))
    assert p >= 40.0


def test_golden_plateaus_of_the_reference_renders(ctx):
    """Deterministic facts of the reference's own PNGs."""
    from PIL import Image
    for name, value, minimum in (("spheres", 221, 0.9), ("cube", 114, 0.5)):
        load_pair(name, ctx)
        _, rgb = ctx.render_image(rt.RenderParams("uniform:2", depth=5, seed=9))
        gold = np.asarray(Image.open(os.path.join(GOLDEN, "scenes", name, name + ".png")).convert("RGB"))
        gm, mm = np.all(gold == value, axis=2), np.all(rgb == value, axis=2)
        assert (gm & mm).sum() >= minimum * gm.sum(), (name, (gm & mm).sum(), gm.sum())


def test_converged_renders_match_the_reference_pngs(ctx):
    """The reference's own renders (scenes/*/*.png, settings unknown) against converged device renders: the only outputs of
    the real reference this repository can hold, and so the pin of everything its unit tests leave open -- Camera::new /
    spawn_ray, Ray::trace with the light-sampling branch, the materials, texture lookup, OBJ import, compute_image's
    mean / sqrt / clamp / quantise.  Measured (tools/golden_psnr.py, depth 12):

        spp (jitter:n)      spheres.png            cube.png             (4x4 = both images box-filtered 4x4 first)
        1024  (32)      46.9 dB, 4x4 59.0 dB   52.4 dB, 4x4 61.4 dB
        4096  (64)      50.5 dB, 4x4 62.5 dB   54.7 dB, 4x4 62.8 dB
        40000 (200)     52.5 dB, 4x4 64.2 dB   55.8 dB, 4x4 62.3 dB    mean difference +0.002 / +0.003 of 255 levels

    Two independently quantised 8-bit images of the same radiance differ by 55.9 dB, so at 40,000 spp the device's frame and
    the reference's are apart by the rounding to 8 bits and the goldens' own residual noise.  The same figures hold for every
    depth >= 8 (spheres) / >= 3 (cube); at the CLI's default depth 5 spheres.png is 5 dB worse after filtering (53.6 dB at
    1024 spp, mean -0.08): it was rendered with more than 5 bounces.  The oracle gives the same numbers (46.91 / 52.41 dB at
    1024 spp; tests/test_oracle_kat.py runs it at 256 spp)."""
    from PIL import Image
    box = lambda a: a.reshape(125, 4, 125, 4, 3).mean(axis=(1, 3))
    psnr = lambda a, b: 10 * np.log10(255.0 ** 2 / np.mean((a - b) ** 2))
    for name, raw_db, filtered_db in (("spheres", 49.5, 60.5), ("cube", 53.5, 61.0)):
        load_pair(name, ctx)
        gold = np.asarray(Image.open(os.path.join(GOLDEN, "scenes", name, name + ".png")).convert("RGB")).astype(float)
        _, rgb = ctx.render_image(rt.RenderParams("jitter:64", depth=12, seed=11))
        img = rgb.astype(float)
        PARITY.setdefault("against_the_reference_pngs_4096spp_depth12", {})[name] = dict(psnr_db=psnr(img, gold), psnr_db_box4x4=psnr(box(img), box(gold)), mean_difference_levels=(img - gold).mean())
        assert psnr(img, gold) >= raw_db, (name, psnr(img, gold))
        assert psnr(box(img), box(gold)) >= filtered_db, (name, psnr(box(img), box(gold)))
        assert abs((img - gold).mean()) <= 0.02, (name, (img - gold).mean())
        # the gate sees a missing bounce: short of convergence in depth the same render fails it by far
        _, rgb = ctx.render_image(rt.RenderParams("jitter:32", depth=3 if name == "spheres" else 1, seed=11))
        assert psnr(box(rgb.astype(float)), box(gold)) < 30.0


# ------------------------------------------------------------------ origins in f64 (found by tools/soak_images.py)
def _box(lo, hi):
    lo, hi = np.array(lo, float), np.array(hi, float)
    c = np.array([[lo[0] if (k >> 2) & 1 == 0 else hi[0], lo[1] if (k >> 1) & 1 == 0 else hi[1], lo[2] if k & 1 == 0 else hi[2]] for k in range(8)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    return np.array([c[i] for q in quads for i in (q[0], q[1], q[2], q[0], q[2], q[3])])


def _mean_where(ctx, objs, bg, cam, w, h, depth, select, seed, chan=0):
    """(device mean, oracle mean) of the red channel (chan: another one, or slice(None) for all three) over the pixels `select(oracle primary ids, oracle image)` picks;
    device 4096 spp, oracle 16 x 256 spp."""
    ctx.upload(rt.Scene.from_objects(w, h, bg, cam, objs))
    osc = O.OracleScene.from_dict(dict(width=w, height=h, background=bg, camera=cam, objects=[
        (dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(len(o["tris"]), np.int64)))
         if o["type"] == "Mesh" else o) for o in objs]))
    want = sum(osc.render(mode="jitter", N=16, depth=depth, seed=50 * seed + k)["sum"] for k in range(16)) / 4096.0
    acc, _ = ctx.render_image(rt.RenderParams("jitter:64", depth=depth, seed=seed))
    got = acc.astype(np.float64) / 4096.0
    m = select(osc.primary_hits(0.5, 0.5)["object_id"], want)
    assert m.sum() >= 10
    import inspect
    PARITY.setdefault("device_over_oracle_mean_radiance", {}).setdefault(inspect.stack()[1].function, []).append(
        dict(pixels=int(m.sum()), depth=depth, seed=seed, ratio=got[m][:, chan].mean() / want[m][:, chan].mean() if want[m][:, chan].mean() else None))
    return got[m][:, chan].mean(), want[m][:, chan].mean()


CAM2 = dict(eye=[0.0, 1.0, 6.0], look_at=[0.0, 0.3, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)


@pytest.mark.parametrize("seed,flipped", [(3, False), (8, False), (9, False), (10, False)])
def test_rehit_rate_of_a_triangle_behind_a_mirror_plane(ctx, seed, flipped):
    """One Lambertian triangle (albedo 0.5) over a perfect mirror at y = -0.6 under a white sky, depth 3: a pixel whose path
    goes eye -> mirror -> triangle (or eye -> triangle -> mirror -> triangle) is 0.5 * (1 - re-hit rate of the ray that leaves the triangle).  That rate depends on the
    LAST BITS of the origin of the ray that arrives: a point of the mirror has y = -0.6 to an ulp of f64 in the reference, and
    when the device handed that ray on as an f32 origin with random low bits the rate was 0.54 instead of 0.34 on the
    triangle of seed 9 (-6 % on these pixels; seeds 3, 8, 10: -3, -1, -1.5 %).  Plane hit points are now computed with the
    reference's f64 sequence and every ray carries its f64 origin (RayQueue::e)."""
    rng = np.random.default_rng(seed)
    c = np.array([rng.uniform(-1.0, 1.0), rng.uniform(0.6, 1.6), rng.uniform(-1.0, 1.0)])
    verts = c + rng.uniform(0.4, 1.5) * rng.normal(size=(3, 3)) * np.array([1.0, 0.4, 1.0])
    tri = dict(type="Mesh", verts=verts, tris=np.array([[0, 2, 1]] if flipped else [[0, 1, 2]]), materials=[dict(type="Lambertian", albedo=[0.5, 0.5, 0.5])])
    mirror = dict(type="Plane", center=[0.0, -0.6, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Metal", albedo=[1.0, 1.0, 1.0], fuzziness=0.0))
    cam = dict(CAM2, look_at=[0.0, 0.0, 0.0])
    got, want = _mean_where(ctx, [mirror, tri], [1.0, 1.0, 1.0], cam, 32, 24, 3, lambda ids, img: (img[..., 0] > 1e-6) & (img[..., 0] < 0.999), seed)   # every pixel a path of which reaches the triangle
    assert abs(got / want - 1.0) < 0.01, (got, want)


def test_axis_aligned_boxes_seen_through_a_glass_sphere(ctx):
    """Rays that leave a sphere land on axis-aligned box faces (x, y or z constant): the re-hit rate there depends on the last
    bits of the point on the sphere, which is the reference's f64 function of the ray (intersections.rs:42-68) and not an
    f32 point with random low bits -- with those the glass pixels of this scene were 3.6 % too bright (11 % in another of the
    2,000 random scenes of tools/soak_images.py).  In scenes with triangles sphere hit points are computed in f64 as well."""
    em = dict(type="Emissive", color=[2.27, 1.07, 0.68])
    glass = dict(type="Sphere", center=[-0.285, 1.179, 1.019], radius=0.746, material=dict(type="Dieletrics", tint=[0.46, 0.37, 0.56], refraction_index=1.58))
    ground = dict(type="Plane", center=[0.0, 0.0, 0.0], normal=[-0.16250880705736634, 1.0, -0.12477666477219285], material=em)
    verts = np.concatenate([_box([-1.267, 0.013, -0.687], [-0.856, 0.539, -0.247]), _box([-1.164, 0.245, -1.59], [-0.071, 1.479, -0.747]),
                            _box([-2.287, 0.503, -1.794], [-1.506, 0.888, -0.796])])
    boxes = dict(type="Mesh", verts=verts, tris=np.arange(len(verts)).reshape(-1, 3), materials=[dict(type="Lambertian", albedo=[0.42, 0.71, 0.55])])
    got, want = _mean_where(ctx, [glass, ground, boxes], [0.626, 0.84, 0.439], CAM2, 48, 36, 4, lambda ids, img: ids == 0, 7)
    assert abs(got / want - 1.0) < 0.012, (got, want)


def test_paths_inside_a_glass_sphere_in_a_scene_with_triangles(ctx):
    """A glass ball (ior 1.3) beside two boxes, depth 7.  A ray inside the ball starts ON the sphere it hits next: |o - c|^2 - r^2 is
    zero to an ulp of f64 in the reference, and where the grazing rays that circle inside the ball come down again depends on it.
    When hits on spheres did not read the ray's carried f64 origin (an experiment of round 3: an f32 origin with random low bits
    was taken to be statistically the same there) the ball of this scene (tools/soak_objects.py seed 7052) was 5.5 % too bright."""
    metal = dict(type="Mesh", verts=_box([0.6788559479263121, -0.14106760500931015, -2.3187110072372614], [1.9416799363717638, 0.6151505817655675, -1.055602460113846]),
                 tris=np.arange(36).reshape(12, 3), materials=[dict(type="Metal", albedo=[0.8652623759134979, 0.6648663076338095, 0.6661495853462316], fuzziness=0.3)])
    gbox = dict(type="Mesh", verts=_box([0.7608002587679906, 0.6151505817655675, -2.0279059522689877], [1.2034704890123549, 1.0167221883679338, -1.372098259896267]),
                tris=np.arange(36).reshape(12, 3), materials=[dict(type="Dieletrics", tint=[0.8323012310339676, 0.9122932201948575, 0.8047355850700093], refraction_index=1.3)])
    ball = dict(type="Sphere", center=[1.2583646024669162, 0.3064544837089029, 0.024018765506871986], radius=0.32036603777811695,
                material=dict(type="Dieletrics", tint=[0.962846918752405, 0.7041745198063561, 0.7784885398438626], refraction_index=1.3))
    cam = dict(eye=[-0.5203155940400985, 0.4871152545134185, 6.742745453529028], look_at=[0.17299150450016387, 0.29057064485837225, -0.36332810553280315],
               up=[0.0, 1.0, 0.0], fovy=35.108566505316226)
    got, want = _mean_where(ctx, [metal, gbox, ball], [0.31526761861579167, 0.6347078316194672, 0.4709119900656106], cam, 48, 36, 7, lambda ids, img: ids == 2, 3)
    assert abs(got / want - 1.0) < 0.02, (got, want)


def test_a_mirror_ball_that_reaches_into_a_glass_box(ctx):
    """Three glass boxes (ior 1.5) on a glass ground plane and a mirror ball that reaches into one of them, depth 7: a ray inside that box
    comes off an axis-aligned face, is reflected by the ball and returns to a face -- it carries the face's exact coordinates with it,
    and the re-hit rate where it lands depends on them.  When hits on spheres did not read the ray's carried f64 origin and direction
    (an experiment of round 3, see k_shade's ray64) this box was 4.8 % too bright (tools/soak_objects.py seed 8309)."""
    def gbox(lo, hi, tint, flip):
        v = _box(lo, hi)
        if flip:
            v = v.reshape(-1, 3, 3)[:, [0, 2, 1]].reshape(-1, 3)
        return dict(type="Mesh", verts=v, tris=np.arange(36).reshape(12, 3), materials=[dict(type="Dieletrics", tint=tint, refraction_index=1.5)])
    b0 = gbox([-1.1807942616506237, 0.03656563987947631, -1.6089911818014622], [0.09908144467888236, 0.78253971956042, -0.43992606706680437],
              [0.7772670320788712, 0.9499073495821635, 0.7042505343539354], False)
    b1 = gbox([0.39572719569778814, -0.27400775218062334, -0.7860813082890725], [0.9766963067103964, 1.1751641139144837, 0.4883751050874159],
              [0.8837588538655262, 0.672486581354505, 0.8057833194848955], True)
    b2 = gbox([0.41318971339585986, 1.1751641139144837, -0.7015813376904707], [1.123353413655269, 2.0289362995270226, -0.05622752517936713],
              [0.7024539616448543, 0.728577511043323, 0.9863635670895208], True)
    ground = dict(type="Plane", center=[0.0, -0.27400775218062334, 0.0], normal=[0.0, 1.0, 0.0],
                  material=dict(type="Dieletrics", tint=[0.6127751462449792, 0.7079720775600078, 0.732610772178722], refraction_index=1.5))
    ball = dict(type="Sphere", center=[0.35184461165613357, 0.26769024633634375, -0.5538524036247869], radius=0.46806194401869505,
                material=dict(type="Metal", albedo=[0.8469419459140477, 0.9423428724419294, 0.9015713086988704], fuzziness=0.0))
    cam = dict(eye=[1.1306143416305008, 0.3197011598215237, 5.079201644725585], look_at=[-0.39461354378169045, 0.13433166125148915, -0.1500790943510547],
               up=[0.0, 1.0, 0.0], fovy=39.360277059158925)
    got, want = _mean_where(ctx, [b0, b1, b2, ground, ball], [0.39905980020419407, 0.8595669272282482, 0.8887123406214668], cam, 48, 36, 7, lambda ids, img: ids == 1, 5)
    assert abs(got / want - 1.0) < 0.02, (got, want)


def test_a_glass_box_alone_traps_what_the_reference_traps(ctx):
    """One axis-aligned glass box (ior 1.886, above sqrt 2: what refracts in through one face is totally reflected by its
    neighbours) under a bright sky.  In the reference a ray that is reflected by the back face returns to the front face with
    a direction whose bits are those of the way in, one sign flipped; there it re-hits the triangle it leaves (no epsilon,
    intersections.rs:166) and stays: of 600,000 paths 0 get out at the third interface (3 of 200,000 in another run of
    tools/glass_box_sim.py).  With f32 directions that get random low bits where they enter f64 arithmetic 124 did (36), and these pixels were 1.1 % too bright from depth 4 on.  Mirror
    and dielectric bounces off triangles and planes are now the reference's f64 sequence and their directions travel exactly
    (RayQueue::f)."""
    verts = _box([-0.83, -0.21, -1.37], [0.71, 1.13, 0.29])
    box = dict(type="Mesh", verts=verts, tris=np.arange(36).reshape(12, 3), materials=[dict(type="Dieletrics", tint=[0.78, 0.77, 0.8], refraction_index=1.886)])
    got, want = _mean_where(ctx, [box], [0.9, 0.65, 0.95], CAM2, 48, 36, 6, lambda ids, img: ids == 0, 2)
    assert abs(got / want - 1.0) < 0.005, (got, want)
    ctx.stats_reset()
    ctx.render_image(rt.RenderParams("jitter:16", depth=6, seed=5, counters=True))
    n = ctx.stats()["rays_per_iteration"]
    # rays alive after the third interface / after the fourth: the reference loses about 1 in 10,000 there, f32 directions lost 25
    assert n[3] > 30000 and n[3] - n[4] <= n[3] // 2500, n


def test_a_glass_box_standing_exactly_on_the_ground_plane(ctx):
    """The bottom face of the box lies IN the ground plane: a ray that comes down inside the glass meets the plane and the
    face at the same distance, and in the reference the roundings of the two f64 distances decide, ray by ray, whether it
    scatters off the ground or is reflected / refracted by the glass (scene.rs:35-50 keeps the smaller t).  The f32
    traversal cannot tell the two apart and used to hand the same one to the shading every time: such boxes were up to
    3.6 times too bright (tools/soak_objects.py, 19 of 300 scenes).  k_closest_hit now reports both and k_shade decides with
    the reference's f64 sequences; the loser, if a triangle, is the neighbour the next ray may meet at t ~ 1e-16."""
    glass = dict(type="Dieletrics", tint=[0.7118, 0.7375, 0.6101], refraction_index=1.321)
    verts = _box([-1.5082, -0.9403, -0.0104], [0.0199, -0.3774, 1.211])
    box = dict(type="Mesh", verts=verts, tris=np.arange(36).reshape(12, 3), materials=[glass])
    ground = dict(type="Plane", center=[0.0, -0.9403, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Lambertian", albedo=[0.6, 0.6, 0.6]))
    cam = dict(eye=[-0.1348, 1.1902, 5.891], look_at=[0.0116, 0.2894, -0.3302], up=[0.0, 1.0, 0.0], fovy=37.97)   # looks down on the top face
    for depth, seed in ((3, 4), (6, 5)):
        got, want = _mean_where(ctx, [box, ground], [0.6574, 0.6599, 0.6919], cam, 48, 36, depth, lambda ids, img: ids == 0, seed)
        assert abs(got / want - 1.0) < 0.015, (depth, got, want)


def test_boxes_at_exactly_representable_coordinates(ctx):
    """Eye, box and ground at multiples of 1/8: distances and hit points are exact far more often than at generic coordinates,
    and what a ray does on the triangle it reaches then depends on the last bits of the PRIMARY direction too (the reference's
    is a unit vector to 1e-16, an f32 direction with random low bits is one to 6e-8).  In scenes with triangles k_raygen
    hands the f64 direction on exactly (RayQueue::e); with f32 primaries this box was 2 % too dark (tools/soak_objects.py
    seed 11; two more of 300 scenes)."""
    cam = dict(eye=[-0.75, 0.5, 5.375], look_at=[-0.5, 0.5, -0.5], up=[0.0, 1.0, 0.0], fovy=32.54405832117061)
    box = dict(type="Mesh", verts=_box([-1.75, -0.5, 0.25], [-1.25, 0.0, 1.75]), tris=np.arange(36).reshape(12, 3),
               materials=[dict(type="Dieletrics", tint=[0.8045560087213051, 0.8651371810067197, 0.7101235263044517], refraction_index=1.886)])
    box2 = dict(type="Mesh", verts=_box([0.375, -0.125, -0.875], [2.0, 0.5, 0.25]), tris=np.arange(36).reshape(12, 3),
                materials=[dict(type="Dieletrics", tint=[0.8366381215796174, 0.6941204926670322, 0.9208810735113934], refraction_index=2.22090029498726)])
    ground = dict(type="Plane", center=[0.0, -0.5, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Dieletrics", tint=[0.958377723300147, 0.7719794768191341, 0.6590765199848376], refraction_index=1.3))
    ball = dict(type="Sphere", center=[1.625, 0.125, -1.875], radius=0.375, material=dict(type="Dieletrics", tint=[0.7875632653689939, 0.9624537355582242, 0.8789444354945191], refraction_index=1.508082892002301))
    light = dict(type="Sphere", center=[3.0, 3.25, 0.75], radius=0.375, material=dict(type="Emissive", color=[1.170251394735188, 5.229450532225288, 3.9394097033343023]))
    got, want = _mean_where(ctx, [box, box2, ground, ball, light], [0.8174728118235843, 0.43695698096172464, 0.3433441646039168], cam, 48, 36, 2, lambda ids, img: ids == 0, 6, chan=slice(None))   # (the light is green and blue)
    assert abs(got / want - 1.0) < 0.008, (got, want)


def test_two_boxes_with_coincident_faces(ctx):
    """The front faces of two boxes lie in one plane (z = 0.25) and overlap; the second box is wound inwards.  In the reference
    a ray that reaches them meets the one with the smaller f64 t (the first in scan order on equality), and the ray that leaves
    re-hits whichever of the two accepts it at t ~ 1e-16 -- with the other material -- so a path is handed back and forth until
    a hit point lands on the far side of the plane.  The device decides the arrival with the reference's two f64 distances, hands
    the partner of a re-hit triangle on with the ray, and its f32 traversal takes no triangle at t ~ 0 (before: 0.32 of the
    reference's radiance on the first box, tools/coincident_probe.py)."""
    cam = dict(eye=[1.875, 1.875, 5.625], look_at=[0.375, 0.375, -0.375], up=[0.0, 1.0, 0.0], fovy=40.0)
    a = dict(type="Mesh", verts=_box([-1.625, 0.5, -0.25], [-0.625, 1.375, 0.25]), tris=np.arange(36).reshape(12, 3), materials=[dict(type="Metal", albedo=[0.8, 0.7, 0.6], fuzziness=0.3)])
    vb = _box([-1.875, 0.375, -1.25], [-0.75, 1.5, 0.25]).reshape(-1, 3, 3)[:, [0, 2, 1]].reshape(-1, 3)
    b = dict(type="Mesh", verts=vb, tris=np.arange(36).reshape(12, 3), materials=[dict(type="Lambertian", albedo=[0.5, 0.6, 0.7])])
    for obj in (0, 1):
        got, want = _mean_where(ctx, [a, b], [0.7, 0.8, 0.9], cam, 48, 36, 4, lambda ids, img: ids == obj, 3 + obj, chan=slice(None))
        assert abs(got / want - 1.0) < 0.03, (obj, got, want)


def test_a_box_face_in_the_plane_y_equals_zero(ctx):
    """The top face of this glass box lies in y = 0.  A ray that leaves it starts at y = 1e-15, not 0: the rounding of
    o + d * t lands next to the plane, and next to 0 an f64 has room for it -- the ray re-hits the face at t = 1e-15, the one
    after it at 3e-31, the next at 7e-47, each time refracted, until the sign of that y lets it go (0.4 % of the paths through
    this box get out with their fifth segment).  The queue carried such a coordinate as an f32 plus 29 bits, which has no
    place for values below 1e-38: they became 0, a hit point exactly ON the face is re-hit for ever, and the box was 2 % too
    dark from depth 5 on (tools/soak_objects.py seed 2444).  split64 / join64 now keep the f64 itself there."""
    cam = dict(eye=[-0.875, 2.0, 6.0], look_at=[0.375, 0.5, -0.25], up=[0.0, 1.0, 0.0], fovy=38.20359598923325)
    glass = dict(type="Dieletrics", tint=[0.9, 0.8, 0.85], refraction_index=1.886)
    vb = _box([-2.0, -0.875, -0.5], [-0.875, 0.0, 0.875]).reshape(-1, 3, 3)[:, [0, 2, 1]].reshape(-1, 3)   # wound inwards, as in the soak scene
    box = dict(type="Mesh", verts=vb, tris=np.arange(36).reshape(12, 3), materials=[glass])
    ground = dict(type="Plane", center=[0.0, -1.0, 0.0], normal=[0.0, 1.0, 0.0], material=glass)
    ball = dict(type="Sphere", center=[0.125, 0.125, 0.0], radius=0.375, material=glass)
    got, want = _mean_where(ctx, [box, ground, ball], [0.7, 0.8, 0.9], cam, 48, 36, 8, lambda ids, img: ids == 0, 9, chan=slice(None))
    assert abs(got / want - 1.0) < 0.01, (got, want)


def test_boxes_stacked_at_multiples_of_an_eighth(ctx):
    """Scene 10033 of tools/soak_objects.py: a Lambertian box on a box (the faces in y = 1.25 coincide), a glass box, ground plane and
    an emitting sphere, every coordinate a multiple of 1/8, the camera looking exactly along -z.  The per-object soak flagged its
    first box at 1.40 of the oracle's radiance (z = +312) on the kernels that formed the reference's three quotients a = Ma / M,
    b = Mb / M, t = -Mt / M with ONE division and two FMAs each (x / m = q + (x - q m) r, r = 1 / m): exact for ordinary operands,
    but a direction with an exact zero component enters the f64 sequence with dither bits as a denormal, M is then denormal, r = inf
    and the quotients NaN where the division gives ordinary numbers.  The quotients are IEEE divisions (RT_FAST_DIV = 0)."""
    lam = lambda c: dict(type="Lambertian", albedo=c)
    objs = [_mesh(_box([-1.125, 0.375, 0.375], [0.0, 1.25, 1.75]), lam([0.7552559979696203, 0.31609736893749824, 0.4499331690700699])),
            _mesh(_box([-1.0, 1.25, 0.375], [-0.25, 1.625, 0.875]), lam([0.32203322900493814, 0.6620052466890284, 0.36084090994144863])),
            _mesh(_box([-1.375, -0.25, -2.0], [-0.625, 1.0, -1.375]), dict(type="Dieletrics", tint=[0.9277299569434767, 0.9985869686072995, 0.9247613444281153], refraction_index=1.886)),
            dict(type="Plane", center=[0.0, -0.25, 0.0], normal=[0.0, 1.0, 0.0], material=lam([0.6, 0.6, 0.6])),
            dict(type="Sphere", center=[2.875, 3.625, 1.75], radius=0.625, material=dict(type="Emissive", color=[3.9736012557661393, 4.8643268841417795, 5.768922539878439]))]
    cam = dict(eye=[0.0, 1.375, 6.0], look_at=[0.0, 1.375, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)
    got, want = _mean_where(ctx, objs, [0.8368196684545302, 0.8460556376207935, 0.5736568618355395], cam, 48, 36, 3, lambda ids, img: ids == 0, 13, chan=slice(None))
    assert abs(got / want - 1.0) < 0.02, (got, want)


def _flip(verts):
    return verts.reshape(-1, 3, 3)[:, [0, 2, 1]].reshape(-1, 3)


def _mesh(verts, material):
    return dict(type="Mesh", verts=verts, tris=np.arange(len(verts)).reshape(-1, 3), materials=[material])


def test_three_surfaces_at_one_distance(ctx):
    """Two overlapping glass boxes that both stand exactly on the ground plane, seen from above: where they overlap, a ray
    that comes down inside the glass meets the plane and TWO bottom faces at one distance.  k_closest_hit reports the nearest
    hit and one runner-up, so one of the three is not looked at ("not covered by construction"); measured
    with error bars by tools/three_way_probe.py: device / oracle 1.002, 0.994 +- 0.003, 1.000 on the three objects.  The gate
    is the one of the two-surface case."""
    gy = -0.4371
    glass = lambda ior, tint: dict(type="Dieletrics", tint=tint, refraction_index=ior)
    objs = [_mesh(_box([-1.31, gy, -1.07], [0.42, 0.38, 0.51]), glass(1.31, [0.9, 0.8, 0.7])),
            _mesh(_box([-0.37, gy, -0.63], [1.23, 0.71, 0.93]), glass(1.52, [0.7, 0.9, 0.8])),
            dict(type="Plane", center=[0.0, gy, 0.0], normal=[0.0, 1.0, 0.0], material=dict(type="Lambertian", albedo=[0.6, 0.6, 0.6]))]
    cam = dict(eye=[0.3, 3.1, 4.9], look_at=[0.0, -0.2, -0.3], up=[0.0, 1.0, 0.0], fovy=35.0)
    for obj in (0, 1, 2):
        got, want = _mean_where(ctx, objs, [0.7, 0.8, 0.9], cam, 48, 36, 5, lambda ids, img: ids == obj, 21 + obj, chan=slice(None))
        assert abs(got / want - 1.0) < 0.015, (obj, got, want)


def test_three_coincident_faces(ctx):
    """The front faces of THREE boxes lie in one plane (z = 0.25) and overlap; the second is wound inwards.  A ray that arrives
    meets the face with the smallest f64 t (first in scan order on equality), the ray that leaves may re-hit any of the three at
    t ~ 1e-16.  The device carries ONE partner with a re-hit ray (queue word f) and one runner-up in a hit record, so of three
    coincident faces one is only seen by the f32 traversal's t ~ 0 candidate slot (AUX_NEAR_BIT); a partner of a partner is
    the same construction.  Gate as for two coincident faces."""
    cam = dict(eye=[1.875, 1.875, 5.625], look_at=[0.375, 0.375, -0.375], up=[0.0, 1.0, 0.0], fovy=40.0)
    a = _mesh(_box([-1.625, 0.5, -0.25], [-0.625, 1.375, 0.25]), dict(type="Metal", albedo=[0.8, 0.7, 0.6], fuzziness=0.3))
    b = _mesh(_flip(_box([-1.875, 0.375, -1.25], [-0.75, 1.5, 0.25])), dict(type="Lambertian", albedo=[0.5, 0.6, 0.7]))
    c = _mesh(_box([-1.25, 0.125, -0.75], [0.25, 1.125, 0.25]), dict(type="Lambertian", albedo=[0.7, 0.4, 0.3]))
    for obj in (0, 1, 2):
        got, want = _mean_where(ctx, [a, b, c], [0.7, 0.8, 0.9], cam, 48, 36, 4, lambda ids, img: ids == obj, 31 + obj, chan=slice(None))
        assert abs(got / want - 1.0) < 0.10, (obj, got, want)     # stated deviation, see the docstring


@pytest.mark.parametrize("offset", [0.0, 1000.0])
def test_a_closed_mesh_room_far_from_the_origin(ctx, offset):
    """A closed room of mesh walls (wound inwards) with a ceiling light, an L of two boxes on its floor (concave creases, one box
    exactly on the floor, the other against it) and the camera inside, under a background 40 times brighter than the light: any
    path that leaks through a crease or a wall shows at once.  The whole scene is translated by (offset, offset, offset).  The
    f32 traversal cannot rank a triangle closer than 2^-17 (|o.x| + |o.y| + |o.z|) -- 0.02 units at 1000 -- against the surface a
    ray starts on; round 2 dropped such candidates, which lost the adjacent wall for every ray that left a surface within that
    distance of a crease (intersections.rs:166 accepts any t >= 0).  They now go to k_shade's f64 test (AUX_NEAR_BIT)."""
    # (turned out of the axes: a hit point on an axis-aligned face rounds ONTO the face more often than not, the ray that leaves
    # re-hits it at t = 0 for ever -- intersections.rs:166 -- and the reference's own axis-aligned room is black)
    ca, sa, cb, sb = np.cos(0.31), np.sin(0.31), np.cos(0.17), np.sin(0.17)
    R = np.array([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]]) @ np.array([[1.0, 0.0, 0.0], [0.0, cb, -sb], [0.0, sb, cb]])
    off = np.array([offset * 1.0000137, offset * 0.9999871, offset * 1.0000213])
    place = lambda v: np.asarray(v, float) @ R.T + off
    wall = dict(type="Lambertian", albedo=[0.7, 0.65, 0.6])
    room = _mesh(place(_flip(_box([-2.013, 0.017, -1.987], [1.991, 3.021, 2.009]))), wall)
    lamp = _mesh(place(_flip(_box([-0.613, 2.917, -0.587], [0.591, 3.221, 0.609]))), dict(type="Emissive", color=[6.0, 6.0, 6.0]))      # pokes through the ceiling: only its lower part is inside
    b1 = _mesh(place(_box([-1.213, 0.017, -1.287], [-0.209, 0.921, -0.491])), dict(type="Lambertian", albedo=[0.3, 0.6, 0.4]))    # stands on the floor
    b2 = _mesh(place(_box([-0.209, 0.017, -1.287], [0.491, 0.521, 0.109])), dict(type="Metal", albedo=[0.8, 0.8, 0.9], fuzziness=0.2))   # ... and against the first box
    cam = dict(eye=list(place([0.7, 1.6, 1.7])), look_at=list(place([-0.4, 0.6, -1.0])), up=list(np.array([0.0, 1.0, 0.0]) @ R.T), fovy=55.0)
    got, want = _mean_where(ctx, [room, lamp, b1, b2], [40.0, 40.0, 40.0], cam, 48, 36, 6, lambda ids, img: ids >= 0, 41, chan=slice(None))
    assert 0.05 < want < 3.0, want          # the oracle's room is lit and closed: nothing of the background gets in
    assert abs(got / want - 1.0) < 0.015, (offset, got, want)


# ------------------------------------------------------------------ properties
def test_tile_split_union_is_bit_identical(ctx):
    load_pair("spheres", ctx)
    full, _ = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=4))
    parts = np.zeros_like(full)
    for k in range(3):
        a, _ = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=4, tile_index=k, tile_count=3))
        assert np.all((a == 0) | (parts == 0))     # disjoint
        parts += a
    assert np.array_equal(parts, full)


def test_kernel_tile_set_equals_partition_helper(ctx):
    D = sub("distributed")
    load_pair("spheres", ctx)
    for k in range(3):
        a, _ = ctx.render_image(rt.RenderParams("grid:1", depth=1, seed=4, tile_index=k, tile_count=3))
        # depth 1: every pixel of the spheres scene sees an emitter, a surface (black) or nothing;
        # use the primary-ray count instead of the colour: rays of this rank = masked pixels
        ctx.stats_reset()
        ctx.render_image(rt.RenderParams("grid:1", depth=1, seed=4, tile_index=k, tile_count=3))
        assert ctx.stats()["rays_per_iteration"][0] == int(D.tile_mask(500, 500, k, 3).sum())
        assert not a[~D.tile_mask(500, 500, k, 3)].any()


def test_sample_split_is_linear(ctx):
    load_pair("spheres", ctx)
    full, _ = ctx.render_image(rt.RenderParams("jitter:4", depth=5, seed=8))
    a, _ = ctx.render_image(rt.RenderParams("jitter:4", depth=5, seed=8, sample_begin=0, sample_end=5))
    b, _ = ctx.render_image(rt.RenderParams("jitter:4", depth=5, seed=8, sample_begin=5, sample_end=16))
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-6)


def test_small_passes_equal_one_big_pass(ctx):
    load_pair("cube", ctx)
    one, _ = ctx.render_image(rt.RenderParams("jitter:3", depth=4, seed=2))
    many, _ = ctx.render_image(rt.RenderParams("jitter:3", depth=4, seed=2, max_paths_per_pass=64 * 100))
    assert np.allclose(one, many, rtol=1e-5, atol=1e-6)


def test_ray_accounting_and_determinism(ctx):
    load_pair("cube", ctx)
    ctx.stats_reset()
    a, rgb_a = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=1, counters=True, timing=True))
    st = ctx.stats()
    assert st["rays_per_iteration"][0] == 500 * 500 * 4 == st["paths"]
    assert all(x >= y for x, y in zip(st["rays_per_iteration"][:5], st["rays_per_iteration"][1:5]))
    assert st["queue_overflow"] == 0 and st["nodes_visited"] > 0 and st["prims_tested"] > 0 and st["hit_ms"] > 0
    b, rgb_b = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=1))
    # contributions of shadow rays are float atomics: order may change the last bits only
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6) and np.abs(rgb_a.astype(int) - rgb_b.astype(int)).max() <= 1
    c, _ = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=2))
    assert not np.array_equal(a, c)


def window_image_parity(ctx, osc, n, windows, depth=5, what=""):
    """Converged-image gate on windows of a full-size frame: the GPU renders the WHOLE frame at n*n spp, the oracle
    (the reference's algorithm behind its conservative cull, bit-identical to the brute force) the windows, twice with
    different seeds.  Per window: PSNR >= 40 dB, mean-abs error of the final colours <= the oracle's own seed-to-seed
    mean-abs error + 1e-3, |bias of the mean radiance| <= 1e-3."""
    spp = n * n
    acc, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=depth, seed=11))
    for (x0, x1, r0, r1) in windows:
        a = osc.render_crop((x0, x1, r0, r1), mode="jitter", N=n, depth=depth, seed=101)["sum"][r0:r1, x0:x1]
        b = osc.render_crop((x0, x1, r0, r1), mode="jitter", N=n, depth=depth, seed=202)["sum"][r0:r1, x0:x1]
        g = acc[r0:r1, x0:x1].astype(np.float64)
        ca, cb, cg = gamma(a, spp), gamma(b, spp), gamma(g, spp)
        p, floor = psnr(cg, ca), psnr(cb, ca)
        mae, mae_floor = np.abs(cg - ca).mean(), np.abs(cb - ca).mean()
        bias = abs(g.mean() - 0.5 * (a.mean() + b.mean())) / spp
        print("%s window x %d..%d rows %d..%d at %d spp: PSNR %.2f dB (oracle seed-to-seed %.2f), mean-abs %.2e (floor %.2e), |bias| %.2e, mean radiance %.3f"
              % (what, x0, x1, r0, r1, spp, p, floor, mae, mae_floor, bias, a.mean() / spp))
        PARITY.setdefault("full_size_windows", {})["%s x%d row%d %d spp" % (what, x0, r0, spp)] = dict(psnr_db=p, oracle_seed_to_seed_db=floor, mean_abs=mae, mean_abs_floor=mae_floor, abs_bias=bias)
        assert a.mean() / spp > 0.02                 # the window is lit: a black window would pass any gate
        assert p >= 40.0
        assert mae <= mae_floor + 1e-3
        assert bias <= 1e-3


def test_full_size_c5_hits_image_and_properties(ctx):
    """BASELINE.json's metric configuration at its full size: 485 spheres + 1,000,000 triangles at 1920x1080.
    Every primary ray of the frame against the oracle (ids, t, normal); converged-image parity of two 32x32 windows
    on the torus at 4096 spp -- the rays after the first bounce at the real tessellation (edges of 0.01 at
    coordinates of 6), where the triangle self-hit quirk (t >= 0 with no epsilon, intersections.rs:166) is most
    sensitive to rounding; then size-independent properties of the whole frame: ray accounting, no queue overflow,
    bit-identical repeat, tile-split union == full frame, sample-split linearity."""
    sc = scenegen.scene_c5()
    assert sc["width"] == 1920 and sc["height"] == 1080
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc), accel=True)
    got, want = hits_pair(ctx, osc, own_camera_id_slack=64)
    on_mesh = want["triangle_id"] >= 0
    assert on_mesh.sum() > 20000 and (want["object_id"] >= 0).mean() > 0.5
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=8)     # see test_full_size_c4_hits_and_image (measured: 4)
    assert rays_through_a_mesh(got, want) == 0
    print("C5 1920x1080, all %d primary rays: %d id mismatches (%d rays on the mesh), max rel t %.2e, normal %.2e"
          % (got["t"].size, bad, on_mesh.sum(), rt_, rn))
    PARITY["primary_hits_c5_1080p"] = dict(rays=got["t"].size, rays_on_mesh=int(on_mesh.sum()), id_mismatches=bad, rays_through_the_mesh=0, max_rel_t=rt_, max_rel_normal=rn)
    rows, cols = np.nonzero(on_mesh)
    r0, c0 = int(np.median(rows)), int(np.median(cols))
    edge_r = int(rows.max()) - 8                                        # lower silhouette of the torus against the ground
    edge_c = int(np.median(cols[rows >= edge_r]))
    windows = [(c0 - 16, c0 + 16, r0 - 16, r0 + 16), (edge_c - 16, edge_c + 16, edge_r - 16, edge_r + 16)]
    window_image_parity(ctx, osc, 64, windows, what="C5 1080p")
    one = dict(depth=5, seed=3, sample_begin=0, sample_end=1)          # 1 spp of a jitter:2 grid
    ctx.stats_reset()
    full, _ = ctx.render_image(rt.RenderParams("jitter:2", **one))
    st = ctx.stats()
    assert st["paths"] == 1920 * 1080 == st["rays_per_iteration"][0] and st["queue_overflow"] == 0
    assert 2.0 < st["rays_total"] / st["paths"] < 4.0
    again, _ = ctx.render_image(rt.RenderParams("jitter:2", **one))
    assert np.array_equal(full, again)                                 # no lights in C5: no float atomics
    parts = np.zeros_like(full)
    for k in range(2):
        a, _ = ctx.render_image(rt.RenderParams("jitter:2", tile_index=k, tile_count=2, **one))
        parts += a
    assert np.array_equal(parts, full)
    two, _ = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=3, sample_begin=0, sample_end=2))
    b, _ = ctx.render_image(rt.RenderParams("jitter:2", depth=5, seed=3, sample_begin=1, sample_end=2))
    assert np.allclose(full + b, two, rtol=1e-5, atol=1e-6)


def test_full_size_c3_hits(ctx):
    """C3 (RTIOW spheres) at 1920x1080: every primary ray against the oracle."""
    sc = scenegen.scene_c3()
    assert (sc["width"], sc["height"]) == (1920, 1080)
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
    got, want = hits_pair(ctx, osc, own_camera_id_slack=64)
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=8)
    print("C3 1920x1080, all %d primary rays: %d id mismatches, max rel t %.2e, normal %.2e" % (got["t"].size, bad, rt_, rn))
    PARITY["primary_hits_c3_1080p"] = dict(rays=got["t"].size, id_mismatches=bad, max_rel_t=rt_, max_rel_normal=rn)


def test_full_size_c4_hits_and_image(ctx):
    """C4 (1,000,000-triangle torus + plane) at 3840x2160: the primary rays of the middle 1920x1080 of the frame and
    of its left border against the oracle, and converged-image parity of two windows at 4096 spp."""
    sc = scenegen.scene_c4()
    assert (sc["width"], sc["height"]) == (3840, 2160)
    ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc), accel=True)
    # BASELINE.json asks for bit-exact ids on scenes/cube.  On the 1M-triangle mesh a ray that passes within f32 rounding of a
    # shared edge (f32 vertices and Moeller-Trumbore on the device, f64 Cramer in the reference) used to land on the neighbouring
    # triangle (42 of these 2 M rays) or between the two, on the far side of the tube (74: tools/id_mismatch_probe.py); since the
    # edge tests of the primary rays have a margin and k_shade / rt_primary_hits decide such a pair in f64 (RT_EDGE_TOL) the ids are
    # the oracle's (measured: 0, and 1 at the second offset).  A ray through a vertex, where more than two triangles tie, can still differ: bounded at 8.
    first = None
    for dx, dy in ((0.0, 0.0), (0.37, 0.71)):            # 2 x 2M rays over ~1M triangle edges
        got, want = hits_pair(ctx, osc, dx, dy, crop=(960, 2880, 540, 1620), own_camera_id_slack=512 if first is None else None)
        on_mesh = want["triangle_id"] >= 0
        assert on_mesh.mean() > 0.5                                      # the torus fills the middle of the frame
        bad, rt_, rn = check_hits(got, want, allow_id_mismatch=8)
        assert rays_through_a_mesh(got, want) == 0
        print("C4 3840x2160, middle 1920x1080 at sub-pixel offset (%.2f, %.2f): %d id mismatches (%d rays on the mesh), max rel t %.2e, normal %.2e"
              % (dx, dy, bad, on_mesh.sum(), rt_, rn))
        if first is None:
            first = (got, want)
            PARITY["primary_hits_c4_4k_middle_1080p"] = dict(rays=got["t"].size, rays_on_mesh=int(on_mesh.sum()), id_mismatches=bad,
                                                             rays_through_the_mesh=0, max_rel_t=rt_, max_rel_normal=rn)
        else:
            PARITY["primary_hits_c4_4k_middle_1080p"]["id_mismatches_at_offset_0.37_0.71"] = bad
    got, want = first
    on_mesh = want["triangle_id"] >= 0
    check_hits(*hits_pair(ctx, osc, crop=(0, 400, 0, 2160), own_camera_id_slack=16), allow_id_mismatch=4)
    rows, cols = np.nonzero(on_mesh)
    r0, c0 = 540 + int(np.median(rows)), 960 + int(np.median(cols))
    # second window: the contact region of torus and plane below the middle of the frame (interreflection)
    hit_plane = (want["object_id"] == 1)
    pr, pc = np.nonzero(hit_plane)
    w2 = (960 + int(np.median(pc)), 540 + int(np.median(pr))) if len(pr) else (c0, r0 + 200)
    windows = [(c0 - 16, c0 + 16, r0 - 16, r0 + 16), (w2[0] - 16, w2[0] + 16, w2[1] - 16, w2[1] + 16)]
    window_image_parity(ctx, osc, 64, windows, what="C4 4K")


def test_device_pointer_api_matches_host_api(ctx):
    import torch
    load_pair("spheres", ctx)
    p = rt.RenderParams("jitter:2", depth=3, seed=6)
    acc_h, rgb_h = ctx.render_image(p)
    acc = torch.zeros((ctx.height, ctx.width, 3), dtype=torch.float32, device="cuda:0")
    ctx.render_accumulate(p, acc)
    rgb = ctx.resolve(acc, 4)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), acc_h, rtol=1e-5, atol=1e-6)
    assert np.abs(rgb.cpu().numpy().astype(int) - rgb_h.astype(int)).max() <= 1


def test_compute_image_on_several_contexts(ctx, tmp_path):
    """--devices / --split (SURVEY 8b additive flags): two contexts (here on the same GPU) share the
    frame; tile split reproduces the single-device image exactly, sample split up to the order of the
    float sums."""
    scene = os.path.join(GOLDEN, "scenes", "spheres", "spheres.yaml")
    one, tiles, spp = str(tmp_path / "one.png"), str(tmp_path / "tiles.png"), str(tmp_path / "spp.png")
    rt.compute_image("jitter:3", 5, scene, one, base_dir=GOLDEN, seed=21)
    rt.compute_image("jitter:3", 5, scene, tiles, base_dir=GOLDEN, seed=21, devices=[0, 0, 0], split="tiles")
    rt.compute_image("jitter:3", 5, scene, spp, base_dir=GOLDEN, seed=21, devices=[0, 0], split="spp")
    a, b, c = rt.read_png(one).astype(int), rt.read_png(tiles).astype(int), rt.read_png(spp).astype(int)
    assert a.shape == (500, 500, 3)
    assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3     # shadow-ray float atomics: last bits only
    assert np.abs(a - c).max() <= 1 and (a != c).mean() < 1e-3
    with pytest.raises(Exception):
        rt.compute_image("jitter:3", 5, scene, one, base_dir=GOLDEN, devices=[99])


# ------------------------------------------------------------------ GPU BVH builder
def _boxes_of_triangles(verts, tris):
    v = np.asarray(verts, np.float32)[np.asarray(tris)]
    return np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1)


def test_gpu_bvh_builder_makes_valid_trees(ctx):
    """The device-built linear BVH passes the same CPU validation as the host tree: every primitive in
    exactly one leaf, every quantised child box contains its subtree, the stack bound holds."""
    rng = np.random.default_rng(5)
    c = rng.uniform(-10, 10, (200000, 3)).astype(np.float32)
    h = rng.uniform(0.001, 0.3, (200000, 3)).astype(np.float32)
    random_boxes = np.concatenate([c - h, c + h], axis=1)
    same = np.tile(np.array([[0.5, 0.5, 0.5, 0.75, 0.75, 0.75]], np.float32), (5000, 1))      # identical keys
    line = np.zeros((3000, 6), np.float32); line[:, 0] = np.arange(3000); line[:, 3] = line[:, 0] + 0.5; line[:, 4:] = 1.0
    tv, tt = scenegen.torus_mesh(1000, 500)
    torus = _boxes_of_triangles(tv, tt)
    fresh = rt.Context(0)
    try:
        fresh.set_bvh_builder("gpu")
        for name, boxes in (("random", random_boxes), ("identical", same), ("line", line), ("small", random_boxes[:300]), ("torus 1M", torus)):
            r = fresh.bvh_build_validate(boxes)
            assert r["nodes"] >= 1 and r["max_stack"] >= 2, (name, r)
            print("GPU BVH %-10s %8d boxes: %7d nodes, depth %2d, %.1f ms" % (name, len(boxes), r["nodes"], r["depth"], r["build_ms"]))
        fresh.set_bvh_builder("host")
        r = fresh.bvh_build_validate(random_boxes)
        assert r["nodes"] > 1000
        with pytest.raises(Exception):
            fresh.set_bvh_builder(7)
    finally:
        fresh.close()


def test_gpu_built_bvh_gives_the_same_hits_and_image(ctx):
    """Same closest hits as the oracle through the device-built tree (only the traversal order differs)."""
    fresh = rt.Context(0)
    try:
        fresh.set_bvh_builder("gpu")
        sc = scenegen.scene_c5(width=240, height=135, nu=80, nv=40)     # 485 spheres + 6,400 triangles
        fresh.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
        osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc))
        check_hits(*hits_pair(fresh, osc, own_camera_id_slack=6), allow_id_mismatch=2)
        gpu_img, _ = fresh.render_image(rt.RenderParams("jitter:2", depth=5, seed=11))
        fresh.set_bvh_builder("host")
        fresh.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
        host_img, _ = fresh.render_image(rt.RenderParams("jitter:2", depth=5, seed=11))
        # same rays, same closest hits: the images agree except where a hit is an exact tie or a ray grazes
        # a box edge (none expected); allow a handful of paths
        diff = np.abs(gpu_img - host_img).max(axis=-1) > 1e-5
        assert diff.mean() < 1e-3, diff.mean()
    finally:
        fresh.close()


# ------------------------------------------------------------------ edge cases
CAM = dict(eye=[0.0, 0.0, 5.0], look_at=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)


def small_scene(ctx, objects, w=37, h=21, bg=(0.25, 0.5, 1.0)):
    ctx.upload(rt.Scene.from_objects(w, h, list(bg), CAM, objects))
    return O.OracleScene.from_dict(dict(width=w, height=h, background=list(bg), camera=CAM, objects=[
        (dict(type="Mesh", materials=o["materials"], mesh=dict(verts_np=o["verts"], tris_np=o["tris"], tri_mat_np=np.zeros(len(o["tris"]), np.int64)))
         if o["type"] == "Mesh" else o) for o in objects]))


def test_depth_zero_is_black(ctx):
    small_scene(ctx, [dict(type="Sphere", center=[0, 0, 0], radius=1.0, material=dict(type="Emissive", color=[1, 1, 1]))])
    acc, rgb = ctx.render_image(rt.RenderParams("jitter:2", depth=0))
    assert not acc.any() and not rgb.any()     # raytrace.rs:52-54


def test_empty_scene_is_background(ctx):
    osc = small_scene(ctx, [])
    acc, rgb = ctx.render_image(rt.RenderParams("jitter:2", depth=5))
    assert np.allclose(acc / 4, [0.25, 0.5, 1.0])
    assert np.array_equal(rgb, osc.render(N=2, depth=5)["rgb8"])
    h = ctx.primary_hits()
    assert np.all(h["object_id"] == -1) and np.all(h["t"] == -1)


def test_single_primitives_and_odd_sizes(ctx):
    tri = dict(type="Mesh", verts=np.array([[-1.0, -1, 0], [1, -1, 0], [0, 1.5, 0.5]]), tris=np.array([[0, 1, 2]]),
               materials=[dict(type="Emissive", color=[0.5, 0.25, 1.0])])
    for objs in ([tri], [dict(type="Sphere", center=[0.2, 0.1, 0], radius=1.3, material=dict(type="Emissive", color=[1, 0.5, 0.25]))],
                 [dict(type="Plane", center=[0, -1, 0], normal=[0, 1, 0], material=dict(type="Emissive", color=[0.5, 0.5, 0.5]))],
                 [tri, dict(type="Sphere", center=[0, 0, -3], radius=2.0, material=dict(type="Emissive", color=[1, 1, 1])),
                  dict(type="Plane", center=[0, -1, 0], normal=[0, 1, 0], material=dict(type="Emissive", color=[0.5, 0.5, 0.5]))]):
        for (w, h) in ((1, 1), (37, 21), (8, 8)):
            osc = small_scene(ctx, objs, w, h)
            check_hits(*hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=0))
            # purely emissive scenes are deterministic at the uniform grid: images must agree to the byte
            _, rgb = ctx.render_image(rt.RenderParams("grid:2", depth=3))
            want = osc.render(mode="uniform", N=2, depth=3)["rgb8"]
            assert np.abs(rgb.astype(int) - want.astype(int)).max() <= 1


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_soups_match_the_oracle(ctx, seed):
    """Random mixtures of meshes (triangle sizes over three decades, slivers, interpenetrating, some
    far away), spheres of very different radii and a plane: ids, t and normals of every primary ray
    against the brute-force oracle.  Exercises leaves of one and two primitives, parked leaves, big
    spheres outside the BVH and rays that miss everything."""
    rng = np.random.default_rng(seed)
    objs = []
    for m in range(int(rng.integers(1, 4))):
        nt = int(rng.integers(50, 1500))
        centre = rng.uniform(-2.0, 2.0, size=(nt, 1, 3)) * np.array([1.0, 0.7, 1.5]) + np.array([0.0, 0.0, -1.0])
        size = 10.0 ** rng.uniform(-2.5, 0.0, size=(nt, 1, 1))
        verts = (centre + size * rng.normal(size=(nt, 3, 3))).reshape(-1, 3)
        if m == 0:
            verts[: 3 * (nt // 10)] *= np.array([1.0, 1.0, 1e-3])      # slivers squeezed into a plane
        objs.append(dict(type="Mesh", verts=verts, tris=np.arange(3 * nt).reshape(nt, 3),
                         materials=[dict(type="Lambertian", albedo=[0.5, 0.5, 0.5])]))
    for k in range(int(rng.integers(0, 30))):
        objs.append(dict(type="Sphere", center=list(rng.uniform(-2.5, 2.5, size=3)), radius=float(10.0 ** rng.uniform(-2.0, 0.3)),
                         material=dict(type="Metal", albedo=[0.8, 0.8, 0.8], fuzziness=0.1)))
    if seed % 2:
        objs.append(dict(type="Sphere", center=[0.0, -500.5, 0.0], radius=498.0, material=dict(type="Lambertian", albedo=[0.3, 0.3, 0.3])))
        objs.append(dict(type="Plane", center=[0.0, 0.0, -6.0], normal=[0.0, 0.2, 1.0], material=dict(type="Lambertian", albedo=[0.2, 0.2, 0.2])))
    osc = small_scene(ctx, objs, 160, 120)
    got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=6)
    assert (want["object_id"] >= 0).mean() > 0.2
    bad, rt_, rn = check_hits(got, want, allow_id_mismatch=3)
    print("soup %d: %d objects, %d id mismatches of %d, max rel t %.2e" % (seed, len(objs), bad, got["t"].size, rt_))
    # the same scene through the GPU-built BVH gives the same hits
    ctx.set_bvh_builder("gpu")
    try:
        small_scene(ctx, objs, 160, 120)
        got2 = ctx.primary_hits(0.5, 0.5)
    finally:
        ctx.set_bvh_builder("host")
    for k in ("object_id", "triangle_id", "t"):
        assert np.array_equal(got[k], got2[k]), k


def test_more_planes_and_big_spheres_than_the_kernel_stages_in_lds(ctx):
    """k_closest_hit keeps the first 4 planes and the first 4 spheres that are too big for the BVH in LDS and reads the rest
    from global memory; at most 8 spheres stay out of the BVH, further big ones go in.  Seven planes (a room and a slanted
    mirror), ten spheres of radius 30...200 around a handful of small ones: primary hits against the oracle, and the mirror
    paths (deterministic) as an image, so that the rays of later iterations cross the same code."""
    em = lambda r, g, b: dict(type="Emissive", color=[r, g, b])
    mirror = dict(type="Metal", albedo=[0.9, 0.9, 0.9], fuzziness=0.0)
    objs = [dict(type="Plane", center=[0, -2.0, 0], normal=[0, 1, 0], material=em(0.2, 0.3, 0.4)),
            dict(type="Plane", center=[0, 6.0, 0], normal=[0, -1, 0], material=em(0.9, 0.9, 0.9)),
            dict(type="Plane", center=[-7.0, 0, 0], normal=[1, 0, 0], material=em(0.8, 0.2, 0.2)),
            dict(type="Plane", center=[7.0, 0, 0], normal=[-1, 0, 0], material=em(0.2, 0.8, 0.2)),
            dict(type="Plane", center=[0, 0, -12.0], normal=[0, 0, 1], material=em(0.3, 0.3, 0.9)),
            dict(type="Plane", center=[0, 0, 9.0], normal=[0, 0, -1], material=em(0.5, 0.5, 0.1)),
            dict(type="Plane", center=[0, -1.0, -8.0], normal=[0.3, 1.0, 0.5], material=mirror)]
    rng = np.random.default_rng(5)
    eye = np.array(CAM["eye"])
    for k in range(10):    # big spheres beyond a point of the room, as seen from the eye: a cap of each is visible
        r = float(rng.uniform(30.0, 200.0))
        p = np.array([rng.uniform(-4, 4), rng.uniform(-1.5, 3.0), rng.uniform(-9.0, -3.0)])
        direction = p - eye + rng.normal(size=3) * 2.0
        direction /= np.linalg.norm(direction)
        objs.append(dict(type="Sphere", center=list(p + direction * r), radius=r, material=mirror if k % 3 == 0 else em(0.1 * k, 0.5, 1.0 - 0.1 * k)))
    for k in range(12):
        objs.append(dict(type="Sphere", center=list(rng.uniform(-2.5, 2.5, size=3) * np.array([1, 0.6, 1])), radius=float(rng.uniform(0.1, 0.6)),
                         material=mirror if k % 2 else em(0.7, (0.1 * k) % 1.0, 0.3)))
    osc = small_scene(ctx, objs, 160, 120)
    got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=6)
    seen = set(np.unique(want["object_id"]).tolist())
    assert {6, 13, 15, 16} <= seen and len(seen) >= 14       # the 7th plane, big spheres beyond the 4th (LDS) and beyond the 8th (in the BVH)
    check_hits(got, want)
    # every path is deterministic (mirrors and emitters only; lights are sampled on the last bounce only, depth is ample)
    _, rgb = ctx.render_image(rt.RenderParams("grid:2", depth=40, seed=1))
    want_rgb = osc.render(mode="uniform", N=2, depth=40, seed=1)["rgb8"]
    diff = np.abs(rgb.astype(int) - want_rgb.astype(int)).max(axis=2)
    assert len(np.unique(want_rgb.reshape(-1, 3), axis=0)) > 20
    # +-1 level is f32 against f64 rounding; a reflection off a small sphere amplifies the f32 rounding of a direction, so a few
    # pixels on the silhouettes seen in a mirror land on the neighbouring emitter
    assert (diff > 1).mean() < 0.01, (diff > 1).mean()


def test_maximum_depth_and_sample_grid(ctx):
    """The ends of the parameter ranges: depth 255 (u8::MAX, main.rs:16) with the default 16 shadow generations on top,
    inside a closed mirror box so that paths really live that long; the largest sample grid (4096 x 4096 cells) over a
    sub-range of its samples; sample ranges at the very end of the grid."""
    mirror = dict(type="Metal", albedo=[1.0, 1.0, 1.0], fuzziness=0.0)
    box = [dict(type="Plane", center=c, normal=n, material=mirror) for c, n in (([0, -3, 0], [0, 1, 0]), ([0, 3, 0], [0, -1, 0]), ([-3, 0, 0], [1, 0, 0]),
                                                                               ([3, 0, 0], [-1, 0, 0]), ([0, 0, -3], [0, 0, 1]), ([0, 0, 8], [0, 0, -1]))]
    glow = dict(type="Sphere", center=[0.5, -0.4, 0.3], radius=0.25, material=dict(type="Emissive", color=[1.0, 0.5, 0.25]))
    osc = small_scene(ctx, box + [glow], 48, 32)
    ctx.stats_reset()
    acc, rgb = ctx.render_image(rt.RenderParams("grid:2", depth=255, seed=3))
    st = ctx.stats()
    assert st["rays_per_iteration"][254] > 0 and st["queue_overflow"] == 0          # some paths are still bouncing at the last allowed bounce
    assert all(a >= b for a, b in zip(st["rays_per_iteration"][:255], st["rays_per_iteration"][1:255]))
    # Albedo 1, fuzziness 0: throughput stays exactly 1 over 255 reflections, so every sample is the emitter's colour, the
    # background's (a ray can slip through a corner: both sides reject hits closer than 1e-5) or black -- every pixel sum
    # must be a * E + b * BG with a + b <= 4.  Which of the three a long path ends in is chaotic: the f64 oracle and its
    # own f32 instantiation differ by 25 % in the lit fraction at this depth (and the f64 oracle by 15 % between two
    # resolutions: a regular grid in a symmetric mirror box is not a random sample), so only short paths are compared.
    E, BG = np.array([1.0, 0.5, 0.25]), np.array([0.25, 0.5, 1.0])
    combos = np.array([a * E + b * BG for a in range(5) for b in range(5 - a)])
    err = np.abs(acc[:, :, None, :] - combos[None, None]).max(axis=3).min(axis=2)
    assert err.max() < 1e-4, err.max()
    want = osc.render(mode="uniform", N=2, depth=255, seed=3)["rgb8"]
    direct = osc.primary_hits(0.25, 0.25)["object_id"] == 6
    assert direct.sum() > 5 and np.abs(rgb[direct].astype(int) - want[direct].astype(int)).max() <= 1
    # After 20 bounces 97 % of the paths are still under way and the lit fraction agrees (f64 oracle 0.6146, its f32 form
    # 0.6328 per pixel sum; pixel-by-pixel comparisons of mirror paths are test_mirror_paths_are_deterministic_and_match_the_oracle's job:
    # here the emitter is a light, which every live path samples at random on its last bounce)
    acc20, _ = ctx.render_image(rt.RenderParams("grid:2", depth=20, seed=3))
    want20 = osc.render(mode="uniform", N=2, depth=20, seed=3)["sum"]
    assert abs(acc20[..., 0].mean() / want20[..., 0].mean() - 1.0) < 0.1
    # the largest grid: 16.7 M cells per pixel, of which the call renders 32 -- first, middle and last ones
    small_scene(ctx, [glow], 16, 8)
    n = 4096
    for s0 in (0, n * n // 2, n * n - 32):
        a, _ = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=3, seed=5, sample_begin=s0, sample_end=s0 + 32))
        assert np.isfinite(a).all() and a.max() <= 32 * 1.0 + 1e-3 and a.min() >= 0.0
    with pytest.raises(rt.RtError):
        ctx.render_image(rt.RenderParams("jitter:%d" % (n + 1), depth=3, sample_begin=0, sample_end=4))


def test_mirror_paths_are_deterministic_and_match_the_oracle(ctx):
    """Perfect mirrors (Metal, fuzziness 0: spheres and a plane, whose intersection tests have an
    epsilon and so never re-hit themselves) in front of emissive meshes and spheres: every path is
    deterministic, so at the uniform grid the image must agree with the f64 oracle pixel by pixel.
    This checks the rays of iterations > 0 (origin on a sphere / plane, BVH traversal of the
    meshes, the miss colour after several reflections) directly rather than statistically."""
    n = 24
    X, Y = np.meshgrid(np.linspace(-4.0, 4.0, n + 1), np.linspace(-2.5, 3.5, n + 1), indexing="ij")
    Z = -3.0 + 0.3 * np.sin(2.0 * X) * np.cos(1.5 * Y)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    idx = lambda i, j: i * (n + 1) + j
    lo = np.array([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(n) for j in range(n) if (i + j) % 2 == 0])
    hi = np.array([[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(n) for j in range(n) if (i // 2 + j) % 3 != 0])
    mirror = lambda a: dict(type="Metal", albedo=a, fuzziness=0.0)
    # (every emissive OBJECT is a light that the last bounce samples at random, raytrace.rs:64-87: few of
    # them, and a depth that no path reaches alive, keep the image deterministic)
    objs = [dict(type="Mesh", verts=np.concatenate([verts, verts + np.array([0.0, 0.0, 0.05])]), tris=np.concatenate([lo, hi + len(verts)]),
                 materials=[dict(type="Emissive", color=[1.0, 0.4, 0.1])]),
            dict(type="Sphere", center=[-1.2, 0.0, 0.0], radius=1.0, material=mirror([0.9, 0.9, 0.9])),
            dict(type="Sphere", center=[1.3, -0.3, 0.5], radius=0.7, material=mirror([0.8, 0.6, 0.9])),
            dict(type="Sphere", center=[0.1, 1.4, -0.6], radius=0.5, material=mirror([0.95, 0.95, 0.5])),
            dict(type="Plane", center=[0.0, -1.2, 0.0], normal=[0.0, 1.0, 0.0], material=mirror([0.7, 0.7, 0.7]))]
    rng = np.random.default_rng(7)
    for k in range(3):
        objs.append(dict(type="Sphere", center=list(rng.uniform(-2.5, 2.5, size=3) * np.array([1.0, 0.5, 0.6]) + np.array([0.0, 0.3, 1.0])),
                         radius=float(rng.uniform(0.15, 0.3)), material=dict(type="Emissive", color=list(rng.uniform(0.2, 1.0, size=3)))))
    osc = small_scene(ctx, objs, 200, 150)
    ctx.stats_reset()
    _, rgb = ctx.render_image(rt.RenderParams("grid:2", depth=40))
    st = ctx.stats()
    assert st["rays_per_iteration"][3] > 1000            # paths of four and more segments exist
    per_it = list(st["rays_per_iteration"]) + [0] * 64
    assert per_it[39] == 0                               # and none lives to the last bounce
    want = osc.render(mode="uniform", N=2, depth=40)["rgb8"]
    diff = np.abs(rgb.astype(int) - want.astype(int)).max(axis=-1)
    # a sample flips where a ray passes within rounding distance of a silhouette or a triangle edge
    print("mirror scene: %.3f %% of pixels differ by more than 1 level, mean |diff| %.4f" % (100.0 * (diff > 1).mean(), diff.mean()))
    assert (diff > 1).mean() < 0.005 and diff.mean() < 0.3     # +-1 level is f32 against f64 rounding


def test_tie_break_prefers_lower_object_index(ctx):
    """scene.rs:43 strict '>': two coincident emissive triangles / spheres -> the first object wins."""
    v = np.array([[-1.0, -1, 0], [1, -1, 0], [0, 1, 0]])
    objs = [dict(type="Mesh", verts=v, tris=np.array([[0, 1, 2]]), materials=[dict(type="Emissive", color=[1, 0, 0])]),
            dict(type="Mesh", verts=v, tris=np.array([[0, 1, 2]]), materials=[dict(type="Emissive", color=[0, 1, 0])]),
            dict(type="Sphere", center=[0, 0, -4], radius=1.0, material=dict(type="Emissive", color=[0, 0, 1])),
            dict(type="Sphere", center=[0, 0, -4], radius=1.0, material=dict(type="Emissive", color=[1, 1, 1]))]
    osc = small_scene(ctx, objs, 64, 64)
    got, want = hits_pair(ctx, osc, 0.5, 0.5, own_camera_id_slack=0)
    check_hits(got, want)
    assert set(np.unique(got["object_id"])) <= {-1, 0, 2}


def test_dropped_shadow_rays_are_an_error(ctx):
    """raytrace.rs:64-87: on the last bounce a path fires with probability 0.1 * L and then spawns L shadow rays, each
    of which can fire again: 0.1 * L * L successors per shadow ray that lands on a scatterer.  The queues hold L rays
    per path, so the first generation always fits; with 8 lights HIDDEN behind a wall every shadow ray lands on a
    scatterer and the process grows 6.4-fold per generation (the reference would recurse without bound): the render
    must say so, not return a darker image -- and it must say so whatever the context rendered before."""
    grey = dict(type="Lambertian", albedo=[0.5, 0.5, 0.5])
    # a closed room of six planes around the camera: every primary ray scatters
    room = [dict(type="Plane", center=c, normal=n, material=grey) for c, n in
            (([0, -1, 0], [0, 1, 0]), ([0, 3, 0], [0, -1, 0]), ([-5, 0, 0], [1, 0, 0]), ([5, 0, 0], [-1, 0, 0]),
             ([0, 0, -4], [0, 0, 1]), ([0, 0, 6], [0, 0, -1]))]
    inside = [dict(type="Sphere", center=[-3.5 + k, 2.0, -1.0], radius=0.3, material=dict(type="Emissive", color=[1, 1, 1])) for k in range(8)]
    outside = [dict(type="Sphere", center=[-3.5 + k, 9.0, -1.0], radius=0.3, material=dict(type="Emissive", color=[1, 1, 1])) for k in range(8)]
    p = rt.RenderParams("jitter:4", depth=1, seed=1)
    for warm in (False, True):
        fresh = rt.Context(0)
        try:
            if warm:     # a context that has rendered a much bigger pass before behaves the same
                small_scene(fresh, room + inside[:2], 256, 256)
                fresh.render_image(rt.RenderParams("jitter:8", depth=2, seed=1))
            small_scene(fresh, room + outside, 64, 64)
            with pytest.raises(Exception, match="dropped"):
                fresh.render_image(p)
            # cutting the process off after the first generation is the documented remedy: nothing is dropped
            fresh.stats_reset()
            fresh.render_image(rt.RenderParams("jitter:4", depth=1, seed=1, max_shadow_generations=1))
            assert fresh.stats()["queue_overflow"] == 0
            # 8 visible lights: shadow rays end on the emitters, the process dies out, nothing is dropped
            small_scene(fresh, room + inside, 64, 64)
            fresh.stats_reset()
            fresh.render_image(p)
            st = fresh.stats()
            assert st["queue_overflow"] == 0 and st["rays_per_iteration"][1] > 64 * 64 * 16 * 4     # ~0.8 * 8 shadow rays per path
            # two lights (the reference's scenes): fine
            small_scene(fresh, room + inside[:2], 64, 64)
            fresh.stats_reset()
            fresh.render_image(p)
            assert fresh.stats()["queue_overflow"] == 0
        finally:
            fresh.close()


def test_invalid_arguments_are_errors(ctx):
    load_pair("cube", ctx)
    with pytest.raises(rt.RtError):
        ctx.render_image(rt.RenderParams(rt.SuperSampling("Jitter", 0)))
    with pytest.raises(rt.RtError):
        ctx.render_image(rt.RenderParams("jitter:2", depth=256))
    with pytest.raises(rt.RtError):
        ctx.render_image(rt.RenderParams("jitter:2", sample_begin=3, sample_end=99))
    with pytest.raises(rt.RtError):
        ctx.render_image(rt.RenderParams("jitter:2", tile_index=2, tile_count=2))


def test_non_finite_geometry_is_an_upload_error(ctx):
    """NaN / inf / beyond-f32 coordinates: the reference would carry them into the pixels; here they are refused before they
    reach the BVH builder (found by tests/host_fuzz.cpp: an infinite box made the node encoder spin)."""
    em = dict(type="Emissive", color=[1, 1, 1])
    tri = lambda v: dict(type="Mesh", verts=np.array([[-1.0, -1, 0], [1, -1, 0], [0, v, 0.5]]), tris=np.array([[0, 1, 2]]), materials=[em])
    for objs in ([dict(type="Sphere", center=[0, float("nan"), 0], radius=1.0, material=em)],
                 [dict(type="Sphere", center=[0, 0, 0], radius=float("inf"), material=em)],
                 [dict(type="Sphere", center=[0, 0, 0], radius=1.0, material=em), dict(type="Plane", center=[0, 1e39, 0], normal=[0, 1, 0], material=em)],
                 [tri(float("-inf"))], [tri(1e300)]):
        with pytest.raises(rt.RtError, match="not finite"):
            ctx.upload(rt.Scene.from_objects(37, 21, [0.25, 0.5, 1.0], CAM, objs))
    with pytest.raises(rt.RtError, match="not finite"):
        ctx.upload(rt.Scene.from_objects(37, 21, [0.25, float("nan"), 1.0], CAM, []))
    # the context is still usable
    small_scene(ctx, [tri(1.5)])
    assert ctx.render_image(rt.RenderParams("jitter:2", depth=2))[0].any()


def test_compute_image_drop_in(ctx, tmp_path):
    """raytracer::compute_image with the reference's four arguments writes the PNG."""
    from PIL import Image
    out = tmp_path / "sub" / "spheres.png"
    rt.compute_image("uniform:4", 5, os.path.join(GOLDEN, "scenes/spheres/spheres.yaml"), str(out), base_dir=GOLDEN)
    img = np.asarray(Image.open(out))
    assert img.shape == (500, 500, 3)
    gold = np.asarray(Image.open(os.path.join(GOLDEN, "scenes/spheres/spheres.png")).convert("RGB"))
    mse = np.mean((img.astype(float) - gold.astype(float)) ** 2)
    assert 10 * np.log10(255.0 ** 2 / mse) > 22.0   # loose: the golden's spp/depth are unknown
    with pytest.raises(rt.SSAADecodeError):
        rt.compute_image("bogus", 5, "x.yaml", str(out))
