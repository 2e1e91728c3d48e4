"""BASELINE config 5 twice in one process: what the first call of a process pays (allocator growth, page-locking the io
ring, code-object loads) against the second. python tools/c5_cold_warm.py [variant] [layers] [hessian]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g; g.build()
import c5_model
variant = sys.argv[1] if len(sys.argv) > 1 else "gptq"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 18
hessian = sys.argv[3] if len(sys.argv) > 3 else "exact"
workdir = c5_model.scratch_dir(layers * 1000 * (1 << 20))
src = c5_model.prepare(layers, workdir=workdir)
for i in range(3):
  out = c5_model.run(layers=layers, variant=variant, workdir=workdir, src=src, hessian=hessian)
  print(json.dumps({"call": i, **{k: out[k] for k in ("seconds", "calibrate_s", "quantize_and_write_s", "gpu_busy_total_s", "gpu_busy_frac")},
                    "device_allocations": out["hbm"]["device_allocations"]}), flush=True)
os.remove(src)
