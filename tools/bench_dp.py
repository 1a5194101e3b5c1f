#!/usr/bin/env python3
"""BASELINE configs[3] data-parallel benchmark: `python bench.py --mode dp --gpus N` (kept as an alias)."""
import os
import runpy
import sys

sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--mode", "dp"] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
