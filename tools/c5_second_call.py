#!/usr/bin/env python3
"""tools/c5_model.py's run, twice in one process: the timeline (MI355Q_TIMELINE=1) and idle accounting of the SECOND call.

  MI355Q_TIMELINE=1 MI355Q_C5_GAPS=1 python tools/c5_second_call.py [--variant mixed]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--variant", default="mixed")
  ap.add_argument("--layers", type=int, default=18)
  ap.add_argument("--calls", type=int, default=2)
  a = ap.parse_args()
  import __graft_entry__ as g
  g.build()
  import c5_model
  from mi355q import distributed as Dm
  Dm.init()
  workdir = c5_model.scratch_dir(a.layers * 1000 * (1 << 20))
  src = c5_model.prepare(a.layers, workdir=workdir)
  for i in range(a.calls):
    res = c5_model.run(a.layers, 128, 512, a.variant, workdir=workdir, src=src)
    res["call"] = i + 1
    print(json.dumps(res), flush=True)
  os.remove(src)


if __name__ == "__main__":
  main()
