#!/usr/bin/env python3
"""channel bench with an alternative library: X3D_LIB=<path> python scratch/chan_ab.py [bench.py args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x3d2_amd import _lib
if os.environ.get("X3D_LIB"):
    _lib.LIB_PATH = os.environ["X3D_LIB"]
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
