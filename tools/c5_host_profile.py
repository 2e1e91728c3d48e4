"""cProfile of one tools/c5_model.py run (host side of BASELINE config 5): python tools/c5_host_profile.py [variant] [layers]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g; g.build()
import c5_model
variant = sys.argv[1] if len(sys.argv) > 1 else "mixed"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 18
c5_model.run(layers=layers, variant=variant)          # warm-up (allocator, pinned ring, kernels)
pr = cProfile.Profile(); pr.enable()
out = c5_model.run(layers=layers, variant=variant)
pr.disable()
print({k: out[k] for k in ("seconds", "calibrate_s", "quantize_and_write_s")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
