"""Where the host time of OSCAR's public call goes (cProfile around get_tensor_quant_params on a resident 4096 x 4096 weight)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, __graft_entry__ as g
g.build()
from mi355q import qtyping, runtime as rt
from mi355q.algorithms.uniform_quantize import oscar
gran = sys.argv[1] if len(sys.argv) > 1 else "CHANNELWISE"
w = torch.randn((4096, 4096), device="cuda") * 0.02
cfg = qtyping.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=qtyping.QuantGranularity[gran])
info = qtyping.OpInfo(op=qtyping.OperatorT(), op_name=qtyping.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                      op_quant_config=qtyping.OpQuantizationConfig(weight_tensor_config=cfg))
mu2 = np.exp(np.random.default_rng(1).normal(size=4096) * 1.5)
res = rt.HbmArray(w)
for _ in range(3):
  oscar.get_tensor_quant_params(info, cfg, res, {"mu2": mu2})
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
  oscar.get_tensor_quant_params(info, cfg, res, {"mu2": mu2})
torch.cuda.synchronize()
print("per call ms", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
  oscar.get_tensor_quant_params(info, cfg, res, {"mu2": mu2})
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
