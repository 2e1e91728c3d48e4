import cProfile, pstats, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, __graft_entry__ as g
g.build()
import file_bench
from mi355q import quantizer, recipe
src, dst = "/tmp/fb_src.tflite", "/tmp/fb_dst.tflite"
file_bench.build_model(src, 8, 4096, 11008)
rcp = recipe.dynamic_wi4b128_afp32()
def run():
  qz = quantizer.Quantizer(src, rcp)
  qz.quantize(serialize_to_path=dst)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("wall", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

# ---- where do the uploads' 60 ms go? time every to_device call of one more run
from mi355q import runtime as rt
_orig = rt.to_device
def timed_to_device(a, dtype=None):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  out = _orig(a, dtype)
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  nb = getattr(a, "nbytes", 0)
  if nb > (1 << 20):
    flags = getattr(a, "flags", None)
    print(f"to_device {nb / 1e6:8.1f} MB  {dt * 1e3:6.2f} ms  {nb / dt / 1e9:5.1f} GB/s  dtype {getattr(a, 'dtype', None)}  contiguous {flags.c_contiguous if flags is not None else None}  aligned {a.ctypes.data % 4096 if hasattr(a, 'ctypes') else None}")
  return out
rt.to_device = timed_to_device
import mi355q.requant_queue as rq
if hasattr(rq, "rt"): rq.rt.to_device = timed_to_device
run()
