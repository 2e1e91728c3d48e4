"""A few mi355q_gptq_apply_f32 calls for rocprofv3.   usage: python tools/apply_profile.py [rows=2048] [d=2048]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
x = torch.randn((4096, d), device="cuda")
hinv, info = ops.gptq_hinv(ops.gptq_xtx(x, 2.0 / 4096))
w = torch.randn((rows, d), device="cuda") * 0.02
scale = (w.abs().amax(dim=1) / 7.0).contiguous()
for _ in range(3):
  ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8)
torch.cuda.synchronize()
