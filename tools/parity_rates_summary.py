"""gpurun_out/parity_rates.jsonl (written by tests/parity_rates.py on the GPU box) -> the table
committed as profiles/rNN_parity_rates.txt: observed mismatch fraction / error of every
tolerance-class (T2) comparison of the `-m gpu` suite, next to the bound the test asserts.

usage: python tools/parity_rates_summary.py [jsonl] > profiles/r02_parity_rates.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_rates.jsonl")
  rows = {}
  with open(path) as f:
    for line in f:
      r = json.loads(line)
      rows[r["test"]] = r          # the last run of a test wins
  print("# T2 comparisons of `pytest -m gpu` on MI355X: observed vs asserted bound")
  print("# (int_mismatch_fraction: share of integers that differ from the oracle / reference fixture,")
  print("#  never by more than one step; bounds are <= 10 x observed, floor 1e-5 = SURVEY section 7)")
  print("# floor-based comparisons (tests/parity_rates.py check_with_floor) carry three rates: GPU vs oracle (observed), GPU vs the")
  print("#  oracle with one float32 sum re-ordered (vs_reord), oracle vs re-ordered oracle (floor); bound = min(cap, max(1e-5, k x floor))")
  print(f"{'observed':>11} {'bound':>8} {'step':>4} {'elements':>9} {'vs_reord':>10} {'floor':>10} {'cap':>8}  kind / comparison")
  for name, r in rows.items():
    three = (f"{r['vs_reordered_oracle']:10.3e} {r['oracle_vs_reordered_oracle']:10.3e} {r['cap']:8.0e}" if "cap" in r
             else f"{'-':>10} {'-':>10} {'-':>8}")
    print(f"{r['observed']:11.3e} {r['bound']:8.0e} {str(r.get('max_step', '-')):>4} {str(r.get('elements', '-')):>9} {three}"
          f"  {r['kind']}: {name}")


if __name__ == "__main__":
  main()
