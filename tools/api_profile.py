import cProfile, pstats, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, __graft_entry__ as g
g.build()
from mi355q import qtyping as q, runtime as rt, requant_queue
from mi355q.algorithms.uniform_quantize import naive_min_max_quantize as mm
w = np.random.default_rng(0).standard_normal((4096, 4096), dtype=np.float32) * np.float32(0.02)
cfg = q.TensorQuantizationConfig(num_bits=8, symmetric=True, granularity=q.QuantGranularity.CHANNELWISE)
info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
pool = [rt.HbmArray(torch.from_numpy(w).cuda() + float(i) * 1e-5) for i in range(64)]
def run():
  with requant_queue.batching():
    ps = [mm.get_tensor_quant_params(info, cfg, p) for p in pool]
  return ps
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("us per tensor", (time.perf_counter() - t0) / 64 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
