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
