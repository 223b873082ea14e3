"""BASELINE.json's target artefact and the gate on it: the combined scene (485 spheres + the 1,000,000-triangle torus)
at 1920x1080 and 1024 spp, rendered by bench.render_north_star (the very function `bench.py --render` runs), checked
on windows of that same frame against the f64 oracle."""
import importlib
import json
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from test_gpu_parity import gamma, psnr
from util import PARITY, ROOT, pkg, sub

pytestmark = pytest.mark.gpu

rt = pkg()
scenegen = sub("scenegen")


def test_north_star_frame_passes_the_window_gate(tmp_path):
    bench = importlib.import_module("bench")
    out = str(tmp_path / "c5_1080p_1024spp.png")
    acc, rgb, sc, info = bench.render_north_star(rt, scenegen, out)
    print("north star: " + json.dumps(info))
    assert acc.shape == (1080, 1920, 3) and info["paths"] == 1920 * 1080 * 1024
    assert np.array_equal(rt.read_png(out), rgb)
    spp = 1024
    # the bytes are the reference's quantisation of the sums (raytrace.rs:144-150, image.rs:37-43)
    want = np.trunc(255.999 * np.minimum(np.sqrt(acc.astype(np.float64) / spp), 1.0)).astype(np.uint8)
    assert np.abs(rgb.astype(int) - want.astype(int)).max() <= 1 and (rgb != want).mean() < 1e-3
    # the gate: windows on the torus, on the big glass / metal spheres and on the ground with small spheres.  The oracle
    # renders each window at 4096 spp (better converged than the frame, so that the comparison measures the frame),
    # and once more with another seed for its own noise floor.
    osc = O.OracleScene.from_dict(scenegen.to_oracle_dict(sc), accel=True)
    hits = osc.closest_hits(osc.camera()["eye"], _dirs(osc, step=8))
    on_mesh = hits["triangle_id"] >= 0
    rows, cols = np.nonzero(on_mesh)
    tr, tc = 8 * int(np.median(rows)), 8 * int(np.median(cols))
    windows = [(tc - 16, tc + 16, tr - 16, tr + 16), (944, 976, 560, 592), (420, 452, 700, 732)]
    for (x0, x1, r0, r1) in windows:
        a = osc.render_crop((x0, x1, r0, r1), mode="jitter", N=64, depth=5, seed=101)["sum"][r0:r1, x0:x1] / 4096.0
        b = osc.render_crop((x0, x1, r0, r1), mode="jitter", N=64, depth=5, seed=202)["sum"][r0:r1, x0:x1] / 4096.0
        g = acc[r0:r1, x0:x1].astype(np.float64) / spp
        ca, cb, cg = gamma(a, 1), gamma(b, 1), gamma(g, 1)
        p, floor = psnr(cg, 0.5 * (ca + cb)), psnr(ca, cb)
        bias = abs(g.mean() - 0.5 * (a.mean() + b.mean()))
        print("window x %d..%d rows %d..%d: PSNR %.2f dB of the 1024-spp frame against the oracle at 2 x 4096 spp (oracle 4096 vs 4096: %.2f dB), "
              "|bias| %.2e, mean radiance %.3f" % (x0, x1, r0, r1, p, floor, bias, a.mean()))
        PARITY.setdefault("north_star_frame_1080p_1024spp", {})["window_x%d_row%d" % (x0, r0)] = dict(psnr_db=p, oracle_seed_to_seed_db=floor, abs_bias=bias, mean_radiance=a.mean())
        assert a.mean() > 0.02 and p >= 40.0 and bias <= 1e-3
    # the WHOLE frame, sparsely: every 16th pixel and row (8,100 pixels) of the 1024-spp frame against the oracle at 4096 spp
    lat = osc.render_crop((0, osc.width, 0, osc.height), mode="jitter", N=64, depth=5, seed=303, step=16)["sum"][::16, ::16] / 4096.0
    g = acc[::16, ::16].astype(np.float64) / spp
    assert lat.shape == g.shape == (68, 120, 3)
    p = psnr(gamma(g, 1), gamma(lat, 1))
    bias = abs(g.mean() - lat.mean())
    print("lattice of %d pixels over the whole frame: PSNR %.2f dB of the 1024-spp frame against the oracle at 4096 spp, |bias| %.2e, mean radiance %.3f"
          % (g.shape[0] * g.shape[1], p, bias, lat.mean()))
    PARITY["north_star_frame_1080p_1024spp"]["lattice_every_16th_pixel_and_row"] = dict(pixels=g.shape[0] * g.shape[1], psnr_db=p, abs_bias=bias, mean_radiance=lat.mean())
    PARITY["north_star_frame_1080p_1024spp"]["info"] = info
    assert p >= 40.0 and bias <= 1e-3


def _dirs(osc, step):
    cam = osc.camera()
    xs = np.arange(0, osc.width, step, dtype=np.float64)[None, :, None]
    ys = (osc.height - 1 - np.arange(0, osc.height, step, dtype=np.float64))[:, None, None]
    return cam["lower_left"] + cam["horizontal"] * xs + cam["vertical"] * ys - cam["eye"]
