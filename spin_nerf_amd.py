"""Import alias: ``import spin_nerf_amd`` -> the package in ``spin-nerf_amd/`` (hyphenated directory)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("spin-nerf_amd")
