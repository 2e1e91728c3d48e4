"""Copies what tools/refresh_profiles.sh left under gpurun_out/<round>/ into profiles/<round>_*, keeping the
hand-written header lines of the files that have them.   usage: python tools/install_profiles.py [r02]

hinv_phases.txt is not installed automatically: profiles/<round>_hinv_phases.txt carries the step-by-step
history above the generated sections; replace its '## d = ...' sections by hand when the chain changed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
O, P = os.path.join(ROOT, "gpurun_out", rnd) + "/", os.path.join(ROOT, "profiles", rnd + "_")


def lines(path):
  with open(path) as fh:
    return fh.read().splitlines(True)


def put(path, text):
  with open(path, "w") as fh:
    fh.write(text)


# kernel trace of the default bench run: first line = the command
old = lines(P + "bench_default_kernel_trace.txt")
put(P + "bench_default_kernel_trace.txt", old[0] + "".join(lines(O + "bench_kernel_trace.txt")))
# kernel trace of path_bench --big + its per-op lines + the sections kept below them
old = lines(P + "paths_kernel_trace.txt")
idx = [i for i, l in enumerate(old) if l.startswith("# tools/path_bench.py --big output")][0]
tail = [i for i, l in enumerate(old) if l.startswith("# tools/gptq_apply_bench.py:")][0]
per_op = [l for l in lines(O + "path_bench.txt") if l.startswith("{")]
put(P + "paths_kernel_trace.txt", old[0] + "".join(lines(O + "paths_kernel_trace.txt")) + old[idx] +
    "".join("# " + l for l in per_op) + "".join(old[tail:]))
# MFMA utilisation: generated table + the clock / power section appended by hand
cur = "".join(lines(P + "gptq_mfma_util.txt"))
k = cur.index("## Shader clock and socket power")
put(P + "gptq_mfma_util.txt", "".join(lines(O + "gptq_mfma_util.txt")).rstrip("\n") + "\n\n" + cur[k:])
put(P + "octav_iterations_and_api_resident.txt", "".join(lines(O + "octav_iterations_and_api_resident.txt")))
put(P + "bench.json", "".join(lines(O + "bench.json")))
put(P + "public_paths.txt", "".join(lines(P + "public_paths.txt")[:2]) + "".join(lines(O + "c4_c5_public.txt")))
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_rates_summary.py"), O + "parity_rates.jsonl"],
                     capture_output=True, text=True, check=True).stdout
put(P + "parity_rates.txt", out)
print(open(O + "gpu_tests_tail.txt").read().strip().splitlines()[-1])
