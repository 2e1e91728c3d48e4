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
