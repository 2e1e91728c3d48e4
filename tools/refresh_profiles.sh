#!/bin/bash
# Round-end refresh of the measured artefacts, run ON THE GPU BOX from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r02'
# Everything lands under gpurun_out/<round>/; copy what is to be judged into profiles/.
set -u
ROUND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > "$OUT/gpu_tests_tail.txt"
cp gpurun_out/parity_rates.jsonl "$OUT/" 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 400 python tools/path_bench.py --big > "$OUT/path_bench.txt" 2>&1
trace() {   # name, then the command
  local name=$1; shift
  rm -rf "/tmp/prof_$name"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "/tmp/prof_$name" -o p -- "$@" > "$OUT/$name.log" 2>&1)
}
trace bench python "$R/bench.py" --steps 20 --warmup 5
python tools/rocprof_summary.py /tmp/prof_bench > "$OUT/bench_kernel_trace.txt" 2>&1
trace paths python "$R/tools/path_bench.py" --big
python tools/rocprof_summary.py /tmp/prof_paths > "$OUT/paths_kernel_trace.txt" 2>&1
trace hinv2048 python "$R/tools/hinv_profile.py" 2048
trace hinv16384 python "$R/tools/hinv_profile.py" 16384
{
  echo "## d = 2048"; python tools/make_gptq_profiles.py --phases /tmp/prof_hinv2048
  echo "## d = 2048, the first kernels of the chain (start / duration / gap to the previous kernel, us)"
  python tools/kernel_timeline.py /tmp/prof_hinv2048 | head -14
  echo "## d = 16384"; python tools/make_gptq_profiles.py --phases /tmp/prof_hinv16384
  echo "## tools/kbench/potf2_bench (one 64 x 64 diagonal block + the 1984 rows below it; cycles of workgroup 0)"
  timeout 60 tools/kbench/potf2_bench
  echo "## tools/kbench/lat_bench (one wave: cycles per instruction)"
  timeout 60 tools/kbench/lat_bench
} > "$OUT/hinv_phases.txt" 2>&1
rm -rf /tmp/prof_pmc
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 \
    -d /tmp/prof_pmc -o p -- python "$R/tools/path_bench.py" --big > "$OUT/pmc.log" 2>&1)
python tools/make_gptq_profiles.py /tmp/prof_pmc > "$OUT/gptq_mfma_util.txt" 2>&1
rm -rf /tmp/prof_fetch
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch -o p -- python "$R/tools/hinv_profile.py" 16384 > "$OUT/fetch.log" 2>&1)
python tools/pmc_big_kernels.py /tmp/prof_fetch > "$OUT/hinv16384_fetch.txt" 2>&1
ls -la "$OUT"
cat "$OUT/gpu_tests_tail.txt"
