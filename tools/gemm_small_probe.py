"""Kernel durations (rocprofv3 timeline) of the Cholesky step's trailing update, for gemm.hip variants.
    rocprofv3 --kernel-trace -d DIR -o p -- python tools/gemm_small_probe.py ; python tools/gemm_small_probe.py --report DIR"""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
VARIANTS = [[], ["-DMI355Q_F64_STAGES=4"]]
CASES = [("beta1 lower", 1.0, 1), ("beta0 lower", 0.0, 1), ("beta1 full", 1.0, 0)]
REPS = 40
if len(sys.argv) > 2 and sys.argv[1] == "--report":
  db = sqlite3.connect(sorted(glob.glob(sys.argv[2] + "/**/*.db", recursive=True))[-1])
  rows = [r for r in db.execute("select name, duration, start from kernels order by start") if "gemm_fast" in r[0] or "gemm_kernel" in r[0]]
  i = 0
  for v in VARIANTS:
    for name, beta, lower in CASES:
      d = sorted(r[1] for r in rows[i:i + REPS])
      i += REPS
      print(f"{' '.join(v) or 'default':28s} {name:12s} median {d[len(d)//2]/1e3:6.2f} us  min {d[0]/1e3:6.2f}")
  sys.exit(0)
import torch
import gemm_bench
libs = [gemm_bench.build(i, v) for i, v in enumerate(VARIANTS)]
n = 2048
A = torch.randn((n, n), dtype=torch.float64, device="cuda")
C = torch.zeros((n, n), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for lib in libs:
  for name, beta, lower in CASES:
    for _ in range(REPS):
      # C[64:, 64:512] -= A[64:, 0:64] A[64:512, 0:64]^T
      rc = lib.mi355q_gemm_f64(A.data_ptr() + 64 * n * 8, n, 1, A.data_ptr() + 64 * n * 8, 1, n, C.data_ptr() + (64 * n + 64) * 8,
                               n, 1, 1984, 448, 64, -1.0, beta, lower, st)
      assert rc == 0
    torch.cuda.synchronize()
