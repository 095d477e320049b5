#!/usr/bin/env python3
"""Development aid: bench.py against a variant build (tools/build_variant.sh): python tools/bench_with_lib.py libagpl_x.so [bench.py args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agpl_amd import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
