import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ai-edge-quantizer_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g; g.build()
import torch
from mi355q import qtyping as q, requant_queue, runtime as rt
from mi355q.algorithms.uniform_quantize import naive_min_max_quantize as mm
xs = [torch.randn((4096, 4096), device="cuda") for _ in range(16)]
cfg = q.TensorQuantizationConfig(num_bits=8, symmetric=True, granularity=q.QuantGranularity.CHANNELWISE)
info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0, op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
res = [rt.HbmArray(t) for t in xs] * 16
def run():
  with requant_queue.batching() as queue:
    for r in res:
      mm.get_tensor_quant_params(info, cfg, r)
  torch.cuda.synchronize()
for _ in range(3): run()
t0=time.perf_counter(); run(); print("us/tensor", (time.perf_counter()-t0)/len(res)*1e6)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
