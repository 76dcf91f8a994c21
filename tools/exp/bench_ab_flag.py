#!/usr/bin/env python3
"""bench.py with a boolean test hook of spgan.ops / spgan.nets switched off: `bench_ab_flag.py ops.SIDE_BY_SIDE [bench args...]`."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
mod, name = sys.argv[1].split(".")
import importlib
getattr(importlib.import_module("spgan." + mod), name)[0] = False
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
