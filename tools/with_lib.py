#!/usr/bin/env python3
"""Runs another tool against a different build of libmi355q.so (tools/build_variant.sh), for A / B timing:

  python tools/with_lib.py tools/kbench/_variants/libmi355q_<name>.so tools/gptq_apply_bench.py [args...]
The binding's library path is replaced before anything loads it; the tool itself is unchanged."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)
from mi355q import _ffi  # noqa: E402

_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
print(f"# libmi355q: {_ffi.LIB_PATH}", flush=True)
runpy.run_path(script, run_name="__main__")
