"""A few calls of mi355q_gptq_xtx_f32 for rocprofv3.   usage: python tools/xtx_profile.py [d=2048] [tokens=65536]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x = torch.randn((n, d), device="cuda")
for _ in range(4):
  h = ops.gptq_xtx(x, 2.0 / n)
torch.cuda.synchronize()
