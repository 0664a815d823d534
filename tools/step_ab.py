#!/usr/bin/env python
"""bench.py's denoise step on an experiment library: OSK_ALT_LIB=<lib> python tools/step_ab.py [bench.py arguments].
(Stand-alone kernel loops run power-throttled on this part -- every attention schedule variant lands within 1 % there -- so
schedule A/B runs are made inside the step, where the kernel mix lets the attention kernel boost.)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _altlib

print("library:", _altlib.install() or "shipped", file=sys.stderr)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
