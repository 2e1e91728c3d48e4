import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops, runtime as rt
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = torch.randn((4096, d), device="cuda")
H = (x.double().T @ x.double()) / 4096 + torch.eye(d, device="cuda", dtype=torch.float64)
del x
real = rt.empty
rt.empty = lambda shape, dtype: torch.zeros(shape, dtype=dtype, device=rt.device())
a, ia = ops.gptq_hinv(H)
a = a.clone()
def poisoned(shape, dtype):
  t = real(shape, dtype); t.view(torch.uint8).fill_(0xFF); return t
rt.empty = poisoned
b, ib = ops.gptq_hinv(H)
print("d", d, "equal", bool(torch.equal(a, b)), "finite", bool(torch.isfinite(b).all()), int(ia.item()), int(ib.item()))
