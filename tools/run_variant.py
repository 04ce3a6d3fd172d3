#!/usr/bin/env python3
"""Diagnostic: run bench.py against an alternative build of the library (A/B experiments).
Usage: python tools/run_variant.py <path/to/libcaenv_variant.so> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collision_avoidance_amd import build as b

b.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
exec(compile(open(sys.argv[0]).read(), sys.argv[0], "exec"))
