"""The C ABI compiled from the C side (tests/abi_check.c, C11, -Wall -Werror -pedantic): every struct size and
field offset is pinned by _Static_asserts there; here the same numbers are compared with the ctypes mirror
(raytracer-rust_amd/lib.py) and with the table the Rust -sys crate is stated to follow
(rust/rtamd-sys/abi_layout.json).  Reference seam: /root/reference/src/raytracer/raytrace.rs:108-156."""
import ctypes as C
import json
import os
import re
import subprocess

from util import ROOT, sub

libmod = sub("lib")

CTYPES = {"rt_scene_desc": libmod.SceneDesc, "rt_render_params": libmod.RenderParamsC, "rt_stats": libmod.StatsC}


def _compiled_layout(tmp_path):
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-o", exe, os.path.join(ROOT, "tests", "abi_check.c")])
    return json.loads(subprocess.check_output([exe]).decode())


def test_header_compiles_as_c11_and_layout_matches_ctypes_and_rust_table(tmp_path):
    layout = _compiled_layout(tmp_path)
    table = json.load(open(os.path.join(ROOT, "rust", "rtamd-sys", "abi_layout.json")))
    for name, ct in CTYPES.items():
        got = layout[name]
        assert got == {"size": table[name]["size"], "fields": [list(f) for f in table[name]["fields"]]}, name
        assert C.sizeof(ct) == got["size"], name
        fields = [f[0] for f in ct._fields_]
        assert fields == [f[0] for f in got["fields"]], name          # same fields, same order
        for fname, off, size in got["fields"]:
            d = getattr(ct, fname)
            assert (d.offset, d.size) == (off, size), (name, fname, d.offset, d.size, off, size)


def test_rust_sys_crate_declares_the_same_fields_and_entry_points():
    """The Rust crate cannot be compiled here; its text is checked against the header instead: same struct
    fields in the same order, every rt_* entry point of rtamd.h declared."""
    rs = open(os.path.join(ROOT, "rust", "rtamd-sys", "src", "lib.rs")).read()
    table = json.load(open(os.path.join(ROOT, "rust", "rtamd-sys", "abi_layout.json")))
    for name in CTYPES:
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, re.S).group(1)
        assert re.findall(r"pub (\w+):", body) == [f[0] for f in table[name]["fields"]], name
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rtamd.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(re.findall(r"pub fn (rt_\w+)\(", rs))


def test_entry_point_count_matches_the_binding(tmp_path):
    assert _compiled_layout(tmp_path)["entry_points"] == len(libmod.ABI_SYMBOLS)
