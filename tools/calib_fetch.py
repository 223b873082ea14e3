#!/usr/bin/env python3
"""Runs the 64-byte-record gather with a known byte count (run it under
`rocprofv3 --pmc FETCH_SIZE` and compare): prints the bytes the kernel loaded."""
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-rust_amd")
ctx = rt.Context(0)
out = C.c_double(0)
table = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 30      # 4 GiB >> 256 MiB Infinity Cache
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = rt.load_library()
if L.rt_debug_gather(ctx._h, table, iters, C.byref(out)):
    raise SystemExit(L.rt_last_error(ctx._h).decode())
print(json.dumps({"table_bytes": table, "iters": iters, "bytes_loaded": out.value}))
