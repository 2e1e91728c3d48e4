"""rocprofv3 durations of the GPTQ far update's GEMM shape: C[2048, N] -= E[2048, 256] x H[256, N] (FP32).
    rocprofv3 --kernel-trace -d DIR -o p -- python tools/gemm_apply_probe.py ; python tools/gemm_apply_probe.py --report DIR"""
import glob, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
CASES = [(1792, 1.0), (1792, 0.0), (256, 1.0), (256, 0.0), (1024, 1.0)]
REPS = 30
if len(sys.argv) > 2 and sys.argv[1] == "--report":
  db = sqlite3.connect(sorted(glob.glob(sys.argv[2] + "/**/*.db", recursive=True))[-1])
  rows = [r for r in db.execute("select name, duration, start from kernels order by start") if "gemm_" in r[0]]
  i = 0
  for n, beta in CASES:
    d = sorted(r[1] for r in rows[i:i + REPS]); name = rows[i][0]; i += REPS
    print(f"N {n:5d} beta {beta}  median {d[len(d)//2]/1e3:6.2f} us  min {d[0]/1e3:6.2f}   {name[40:110]}")
  sys.exit(0)
import torch, __graft_entry__ as g
g.build()
from mi355q import _ffi
lib = _ffi.lib()
d = 2048
E = torch.randn((2048, 256), device="cuda"); H = torch.randn((d, d), device="cuda"); C = torch.zeros((2048, d), device="cuda")
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for n, beta in CASES:
  for _ in range(REPS):
    rc = lib.mi355q_gemm_f32(E.data_ptr(), 256, 1, H.data_ptr() + (d - n) * 4, d, 1, C.data_ptr() + (d - n) * 4, d, 1,
                             2048, n, 256, -1.0, beta, 0, st)
    assert rc == 0
  torch.cuda.synchronize()
