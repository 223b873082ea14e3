#!/usr/bin/env python3
"""Regenerates tests/abi_check.c and rust/rtamd-sys/abi_layout.json from include/rtamd.h -- run it DELIBERATELY after
a change of the ABI structs (and update raytracer-rust_amd/lib.py and rust/rtamd-sys/src/lib.rs with it): the numbers
are pins, tests/test_abi.py and the _Static_asserts exist to notice changes nobody meant to make."""
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(ROOT, "include", "rtamd.h")).read()


def fields(struct):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        star = decl.rfind("*")
        names = decl[star + 1:] if star >= 0 else decl.split(None, 1)[1]
        out += [re.sub(r"\[.*\]", "", n.strip()) for n in names.split(",")]
    return out


structs = {s: fields(s) for s in ("rt_scene_desc", "rt_render_params", "rt_stats")}
probe = ['#include <stddef.h>', '#include <stdio.h>', '#include "%s"' % os.path.join(ROOT, "include", "rtamd.h"), 'int main(void){']
for s, fs in structs.items():
    probe.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
    probe += ['printf("%s.%s %%zu %%zu\\n", offsetof(%s,%s), sizeof(((%s*)0)->%s));' % (s, f, s, f, s, f) for f in fs]
probe.append('return 0;}')
with tempfile.TemporaryDirectory() as td:
    open(os.path.join(td, "p.c"), "w").write("\n".join(probe))
    subprocess.check_call(["gcc", "-std=c11", os.path.join(td, "p.c"), "-o", os.path.join(td, "p")])
    out = subprocess.check_output([os.path.join(td, "p")]).decode()
layout = {}
for line in out.strip().split("\n"):
    p = line.split()
    if "." in p[0]:
        s, f = p[0].split(".")
        layout[s]["fields"].append([f, int(p[1]), int(p[2])])
    else:
        layout[p[0]] = {"size": int(p[1]), "fields": []}
json.dump(layout, open(os.path.join(ROOT, "rust", "rtamd-sys", "abi_layout.json"), "w"), indent=1)

names = set()
for h in ("rtamd.h", "rtamd_host.h"):
    t = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
    names |= set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", t))
L = ['/* Compiles the C ABI (include/rtamd.h, include/rtamd_host.h) as C11 -- the nearest compilable proxy for what a',
     ' * Rust `-sys` crate (rust/rtamd-sys/src/lib.rs) assumes -- and pins the size of every struct and the offset and',
     ' * size of every field (LP64).  Built by __graft_entry__.build() with',
     ' *   gcc -std=c11 -Wall -Werror -pedantic -o tests/abi_check tests/abi_check.c',
     ' * so a layout change breaks the build until this table (tools/gen_abi_check.py), raytracer-rust_amd/lib.py (ctypes)',
     ' * and rust/rtamd-sys/abi_layout.json (the table rust/rtamd-sys/src/lib.rs is stated to follow) are updated together.',
     ' * Run, it prints the layout as JSON; tests/test_abi.py compares that with the ctypes structs and the table.',
     ' * Reference seam: /root/reference/src/raytracer/raytrace.rs:108-156, /root/reference/src/main.rs:21-26. */',
     '#include <stddef.h>', '#include <stdio.h>', '', '#include "../include/rtamd.h"', '#include "../include/rtamd_host.h"', '',
     '#define SIZE_IS(T, n) _Static_assert(sizeof(T) == (n), "sizeof(" #T ") changed")',
     '#define FIELD_IS(T, f, off, sz)                                                  \\',
     '    _Static_assert(offsetof(T, f) == (off), "offsetof(" #T ", " #f ") changed"); \\',
     '    _Static_assert(sizeof(((T*)0)->f) == (sz), "sizeof(" #T "." #f ") changed")', '']
for s, info in layout.items():
    L.append('SIZE_IS(%s, %d);' % (s, info["size"]))
    L += ['FIELD_IS(%s, %s, %d, %d);' % (s, f, off, sz) for f, off, sz in info["fields"]]
    L.append('')
L += ['_Static_assert(RT_MAX_ITERATIONS == 320, "RT_MAX_ITERATIONS changed");',
      '_Static_assert(RT_MAT_LAMBERTIAN == 0 && RT_MAT_METAL == 1 && RT_MAT_DIELECTRIC == 2 && RT_MAT_TEXTURE == 3 && RT_MAT_EMISSIVE == 4, "material tags changed");',
      '_Static_assert(RT_SSAA_UNIFORM == 0 && RT_SSAA_JITTER == 1, "ssaa tags changed");',
      '_Static_assert(RT_FLAG_COUNTERS == 1 && RT_FLAG_TIMING == 2, "flag bits changed");',
      '_Static_assert(RT_NOALLOC_SCRATCH_BYTES == (64u << 20), "scratch size changed");',
      '_Static_assert(sizeof(void*) == 8 && sizeof(size_t) == 8, "LP64 expected");', '',
      '/* every entry point is declared with a C-compatible signature (an unevaluated use: nothing is linked) */',
      '#define DECLARED(f) _Static_assert(sizeof(&f) == sizeof(void (*)(void)), #f " is not declared")']
L += ['DECLARED(%s);' % n for n in sorted(names)]
L += ['#define N_ENTRY_POINTS %d' % len(names), '',
      '#define P_STRUCT(T) printf("%s\\"" #T "\\": {\\"size\\": %zu, \\"fields\\": [", first_struct ? "" : ", ", sizeof(T)); first_struct = 0; first_field = 1',
      '#define P_FIELD(T, f) printf("%s[\\"" #f "\\", %zu, %zu]", first_field ? "" : ", ", offsetof(T, f), sizeof(((T*)0)->f)); first_field = 0',
      '#define P_END() printf("]}")', '', 'int main(void) {', '    int first_struct = 1, first_field = 1;', '    printf("{");']
for s, info in layout.items():
    L.append('    P_STRUCT(%s);' % s)
    L += ['    P_FIELD(%s, %s);' % (s, f) for f, off, sz in info["fields"]]
    L.append('    P_END();')
L += ['    printf(", \\"entry_points\\": %zu}\\n", (size_t)N_ENTRY_POINTS);', '    return 0;', '}', '']
open(os.path.join(ROOT, "tests", "abi_check.c"), "w").write("\n".join(L))
print("wrote tests/abi_check.c and rust/rtamd-sys/abi_layout.json (%d entry points)" % len(names))
