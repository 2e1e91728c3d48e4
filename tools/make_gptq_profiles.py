"""Summaries of the GPTQ dense kernels from rocprofv3 output (kept under profiles/).

  # on the GPU box (counters in their own pass, kernel-trace only):
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 \
            SQ_INSTS_VALU_MFMA_MOPS_F64 -d gpurun_out/gptq_pmc -o p -- python tools/path_bench.py --big
  python tools/make_gptq_profiles.py gpurun_out/gptq_pmc > gpurun_out/r01_gptq_mfma_util.txt
  rocprofv3 --kernel-trace -d gpurun_out/hinv_trace -o p -- python tools/hinv_profile.py 16384
  python tools/make_gptq_profiles.py --phases gpurun_out/hinv_trace > gpurun_out/r01_hinv_phases.txt
"""
import glob
import sqlite3
import sys


def db_of(path):
  return sqlite3.connect(sorted(glob.glob(path + "/**/*.db", recursive=True))[-1])


def mfma(path):
  c = db_of(path)
  rows = c.execute(
      "select k.dispatch_id, k.name, k.duration, k.grid_x*k.grid_y*k.grid_z, p.counter_name, p.value "
      "from kernels k join counters_collection p on p.dispatch_id = k.dispatch_id").fetchall()
  per = {}
  for did, name, dur, threads, cname, val in rows:
    d = per.setdefault(did, dict(name=name, dur=dur, threads=threads))
    d[cname] = d.get(cname, 0.0) + val
  print("# MFMA utilisation of the GPTQ GEMMs from PMC counters (tools/make_gptq_profiles.py)")
  print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE/8) * 1024 SIMDs)")
  print("# flops = MOPS * 512; effective clock = (GRBM_GUI_ACTIVE/8) / duration; the longest dispatches of each kernel:")
  print(f"{'dur_us':>10} {'TFLOP/s':>8} {'MfmaUtil%':>9} {'GHz':>5}  kernel")
  groups = {}
  for d in per.values():
    if "gemm" not in d["name"] and "Cijk" not in d["name"] and "xtx_bf16x3_kernel" not in d["name"]:
      continue
    groups.setdefault(d["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64], []).append(d)
  for name, ds in sorted(groups.items()):
    for d in sorted(ds, key=lambda x: -x["dur"])[:6]:
      gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
      flops = (d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) +
               d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)) * 512
      util = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024) if gui else 0.0
      print(f"{d['dur'] / 1e3:10.1f} {flops / d['dur'] / 1e3:8.2f} {util:9.1f} {gui / d['dur']:5.2f}  {name}")


def phases(path):
  c = db_of(path)
  rows = c.execute("select name, start, end, duration from kernels order by start").fetchall()
  first = max(i for i, r in enumerate(rows) if "copy_damped_lower" in r[0])
  rows = rows[first:]
  last_potf2 = max(i for i, r in enumerate(rows) if "potf2_" in r[0] or "chol_step" in r[0])
  put = max(i for i, r in enumerate(rows) if "diag_inverse" in r[0])
  sym = max(i for i, r in enumerate(rows) if "mirror_lower_f32" in r[0])
  gemms = [i for i, r in enumerate(rows) if "gemm" in r[0]]
  prod = max(i for i in gemms if i < sym)
  prod_end = prod + 1
  # d >= 4096: L^-T L^-1 runs on the bf16 split (ltl_bf16x3: split kernels + xtx_bf16x3_kernel with the triangular k range),
  # not as a kernel with "gemm" in its name -- everything between the last merge product and the mirror is the product
  split_ltl = [i for i, r in enumerate(rows) if put < i < sym and "xtx_bf16x3" in r[0]]
  if split_ltl and split_ltl[-1] > prod:
    prod, prod_end = prod + 1, sym
  def span(a, b):
    sel = rows[a:b]
    return sum(r[3] for r in sel), (sel[-1][2] - sel[0][1]) if sel else 0
  print("# phases of one mi355q_gptq_hinv_f64 call (kernel time / wall span, microseconds)")
  for label, a, b in (("cholesky (two-level blocked, potf2 + GEMMs)", 0, put),
                      ("  of which chol_step / potf2 + trsm_panel (the serial chain)", None, None),
                      ("triangular inverse (pairwise merge GEMMs)", put, prod),
                      ("L^-T L^-1 product", prod, prod_end), ("mirror of the float32 lower triangle", sym, sym + 1)):
    if a is None:
      t = sum(r[3] for r in rows[:last_potf2 + 1] if "potf2_" in r[0] or "trsm_panel" in r[0] or "chol_step" in r[0])
      print(f"{t / 1e3:12.1f} {'':>12}  {label}")
      continue
    k, w = span(a, b)
    print(f"{k / 1e3:12.1f} {w / 1e3:12.1f}  {label}")
  print(f"{'':>12} {(rows[sym][2] - rows[0][1]) / 1e3:12.1f}  whole call")


if __name__ == "__main__":
  if sys.argv[1] == "--phases":
    phases(sys.argv[2])
  else:
    mfma(sys.argv[1])
