#!/usr/bin/env python
"""OSK_ALT_LIB=<alternative libosk_hip.so> python tools/run_with_lib.py <script.py> [args...]: run a repo script (bench.py, ...) against an
A/B build of the library (tools/_altlib.py) -- same process layout as the plain run."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _altlib

print("library:", _altlib.install() or "shipped", file=sys.stderr)
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name="__main__")
