import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pkg():
    """The product package (directory name has a hyphen, so import it by name)."""
    return importlib.import_module("raytracer-rust_amd")


def sub(name):
    return importlib.import_module("raytracer-rust_amd." + name)


# Measured parity figures (PSNR, bias, id mismatches, device / oracle ratios) that the GPU tests put here; conftest.py writes
# them to gpurun_out/parity_record.json when the session ends, and a copy of a full run is kept under profiles/.
PARITY = {}
