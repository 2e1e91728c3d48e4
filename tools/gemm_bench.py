"""FP64 / FP32 GEMM kernel timing for tuning gemm.hip variants on the GPU box.

  python tools/gemm_bench.py [--defs "-DFOO=1;-DBAR"]      # ';'-separated variant flag sets
  python tools/gemm_bench.py --libs "a.so;b.so"              # libraries built beforehand (tools/build_variant.sh)
Each variant compiles csrc/*.{hip,cpp} into /tmp/libmi355q_<n>.so with the extra flags and is
timed (HIP events, interleaved rounds, medians) on three FP64 shapes of the Hessian inverse:
full 8192^3, the rank-64 lower trailing update at n = 16384 and the L^-T L^-1 product.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402

CSRC = os.path.join(ROOT, "ai-edge-quantizer_amd", "csrc")


def build(idx, flags, prebuilt=None):
  out = prebuilt or f"/tmp/libmi355q_{idx}.so"
  if prebuilt is None:
    srcs = [os.path.join(CSRC, s) for s in g.SOURCES]
    cmd = ["/opt/rocm/bin/hipcc", *g.HIPCC_FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *flags, *srcs, "-o", out]
    subprocess.run(cmd, check=True)
  lib = ctypes.CDLL(out)
  lib.mi355q_gemm_f64.restype = ctypes.c_int32
  lib.mi355q_gemm_f64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                  ctypes.c_void_p]
  return lib


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--defs", default="")
  ap.add_argument("--libs", default="")
  ap.add_argument("--rounds", type=int, default=5)
  a = ap.parse_args()
  if a.libs:
    paths = [os.path.abspath(v) for v in a.libs.split(";") if v]
    variants = [[os.path.basename(v)] for v in paths]
    libs = [build(i, [], v) for i, v in enumerate(paths)]
  else:
    variants = [v.split() for v in a.defs.split(";")] if a.defs else [[]]
    libs = [build(i, v) for i, v in enumerate(variants)]
  n = 16384
  A = torch.randn((n, n), dtype=torch.float64, device="cuda")
  B = torch.randn((n, n), dtype=torch.float64, device="cuda")
  C = torch.zeros((n, n), dtype=torch.float64, device="cuda")
  st = torch.cuda.current_stream().cuda_stream
  shapes = {
      "full_8192": dict(M=8192, N=8192, K=8192, a=(n, 1), b=(n, 1), beta=0.0, lower=0, flops=2 * 8192**3),
      "rank64_lower_16384": dict(M=n, N=n, K=64, a=(n, 1), b=(1, n), beta=1.0, lower=1, flops=n * n * 64),
      "rank512_lower_16384": dict(M=n, N=n, K=512, a=(n, 1), b=(1, n), beta=1.0, lower=1, flops=n * n * 512),
      # the trailing update of one 64-column Cholesky step at d = 2048 (first step of an outer block)
      "chol_step_1984x448_k64": dict(M=1984, N=448, K=64, a=(n, 1), b=(1, n), beta=1.0, lower=1, flops=2 * 1984 * 448 * 64),
      # one pair of a low triangular-inverse level and the d = 2048 product
      "merge_1024_kmode0": dict(M=1024, N=1024, K=1024, a=(n, 1), b=(n, 1), beta=0.0, lower=0, flops=2 * 1024**3),
      "AtA_lower_2048": dict(M=2048, N=2048, K=2048, a=(1, n), b=(n, 1), beta=0.0, lower=1, flops=2048**3),
      # the same rank-512 update with B read from a transposed copy of the panel (j contiguous)
      "rank512_lower_16384_Bt": dict(M=n, N=n, K=512, a=(n, 1), b=(n, 1), beta=1.0, lower=1, flops=n * n * 512),
      "rank512_lower_16384_AtBt": dict(M=n, N=n, K=512, a=(1, n), b=(n, 1), beta=1.0, lower=1, flops=n * n * 512),
      "AtA_lower_16384_k8192": dict(M=n, N=n, K=8192, a=(1, n), b=(n, 1), beta=0.0, lower=1, flops=n * n * 8192),
  }
  res = {name: [[] for _ in libs] for name in shapes}
  ref = {}
  for rnd in range(a.rounds + 1):
    for name, s in shapes.items():
      for li, lib in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if s["beta"] != 0.0:
          C.zero_()
        e0.record()
        rc = lib.mi355q_gemm_f64(A.data_ptr(), s["a"][0], s["a"][1], B.data_ptr(), s["b"][0], s["b"][1], C.data_ptr(),
                                 n, 1, s["M"], s["N"], s["K"], 1.0, s["beta"], s["lower"], st)
        e1.record()
        e1.synchronize()
        assert rc == 0
        if rnd:
          res[name][li].append(e0.elapsed_time(e1))
        chk = float(C[: s["M"], : s["N"]].tril().double().sum().item()) if rnd == 0 else None
        if rnd == 0:
          ref.setdefault(name, []).append(chk)
  for name, s in shapes.items():
    for li in range(len(libs)):
      ms = sorted(res[name][li])[len(res[name][li]) // 2]
      print(json.dumps(dict(shape=name, variant=" ".join(variants[li]) or "default", ms=round(ms, 3),
                            tflops=round(s["flops"] / ms / 1e9, 1), checksum=ref[name][li])))


if __name__ == "__main__":
  main()
