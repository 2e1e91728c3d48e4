import os, sys, time
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pool = [rt.HbmArray(torch.from_numpy(w).cuda() + float(i) * 1e-5) for i in range(N)]
for rep in range(6):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  with requant_queue.batching() as qu:
    ps = [mm.get_tensor_quant_params(info, cfg, p) for p in pool]
    t1 = time.perf_counter()
  t2 = time.perf_counter()
  torch.cuda.synchronize(); t3 = time.perf_counter()
  print(f"N={N} enqueue {1e6*(t1-t0)/N:6.2f} us/tensor   exit (flush + wait + scales) {1e6*(t2-t1):7.1f} us   total {1e6*(t3-t0)/N:6.2f} us/tensor")
