"""Two calls of mi355q_gptq_hinv_f64 on a synthetic SPD matrix, for rocprofv3 (see
tools/make_gptq_profiles.py --phases).   usage: python tools/hinv_profile.py [d=16384]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
d=int(sys.argv[1]) if len(sys.argv)>1 else 16384
x=torch.randn((4096,d),device="cuda")
H=(x.double().T@x.double())/4096+torch.eye(d,device="cuda",dtype=torch.float64)
H64=H.double() if H.dtype!=torch.float64 else H
for _ in range(2):
    r=ops.gptq_hinv(H64)
torch.cuda.synchronize()
