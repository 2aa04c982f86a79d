"""Import helper: the package directory is named ``voxel-raytracer_amd`` (hyphen), so it is loaded by path name."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def vrt():
    return importlib.import_module("voxel-raytracer_amd")
