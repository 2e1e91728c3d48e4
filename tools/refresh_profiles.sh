#!/bin/bash
# Round-end refresh of the measured artefacts, run ON THE GPU BOX from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r03'
# Everything lands under gpurun_out/<round>/; copy what is to be judged into profiles/.
set -u
ROUND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 600 bash tools/build_kbench.sh > "$OUT/build_kbench.log" 2>&1    # (r05: the probes below were run without having been built)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > "$OUT/gpu_tests_tail.txt"
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
  echo "## d = 16384, per outer block of the Cholesky (tools/hinv_chain_timeline.py): span against the look-ahead GEMM"
  python tools/hinv_chain_timeline.py /tmp/prof_hinv16384
  echo "## tools/kbench/potf2_bench (one 64 x 64 diagonal block + the 1984 rows below it; cycles of workgroup 0)"
  timeout 60 tools/kbench/potf2_bench
  echo "## tools/kbench/lat_bench (one wave: cycles per instruction)"
  timeout 60 tools/kbench/lat_bench
} > "$OUT/hinv_phases.txt" 2>&1
rm -rf /tmp/prof_pmc
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    -d /tmp/prof_pmc -o p -- python "$R/tools/path_bench.py" --big > "$OUT/pmc.log" 2>&1)
python tools/make_gptq_profiles.py /tmp/prof_pmc > "$OUT/gptq_mfma_util.txt" 2>&1
rm -rf /tmp/prof_fetch
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch -o p -- python "$R/tools/hinv_profile.py" 16384 > "$OUT/fetch.log" 2>&1)
python tools/pmc_big_kernels.py /tmp/prof_fetch > "$OUT/hinv16384_fetch.txt" 2>&1
{
  echo "# tools/octav_iter_bench.py on one MI355X: OCTAV clip search (int4, channelwise), total time (two kernels + memset, wall clock"
  echo "# over 20 calls) vs number of Newton iterations; all 10 run in production (the reference's early stop is global)."
  echo "# octav_rows_kernel<SLOTS, THREADS>: workgroup per row, 16-element pieces in registers (one per thread; 512 / 1024 threads for rows"
  echo "# beyond 4096 elements), run sums listed in LDS (fixed 16-step pass, or a run loop for sparsely selected pieces), the two masks'"
  echo "# serial chains on the lanes of ONE wave; unchanged masks reuse the sums; a guess above the row's largest |x| selects nothing"
  echo "# without touching the masks; once few elements are selected and the guess grows, the candidates are listed and a one-wave"
  echo "# tail kernel (octav_tail_kernel) finishes the row."
  echo "# --- 4096 x 4096, sigma = 0.02 (typical weights: the first guess 1.0 selects nothing, the second (0.0) half of every mask)"
  timeout 200 python tools/octav_iter_bench.py 4096 4096 0.02 2>&1 | grep max_iter
  echo "# --- 4096 x 4096, sigma = 1.0"
  timeout 200 python tools/octav_iter_bench.py 4096 4096 1.0 2>&1 | grep max_iter
  echo "# --- other shapes, sigma = 0.02, all ten iterations"
  for s in "2048 2048" "16384 2048" "2048 8192" "4096 8192" "4096 11008" "2048 16384"; do
    echo "# rows cols = $s"; timeout 200 python tools/octav_iter_bench.py $s 0.02 2>&1 | grep "max_iter=10"
  done
  echo "# tools/octav_block_bench.py (blockwise units: 32 .. 256 octav_unit_lanes_kernel, 512 octav_groups_kernel, 1024+ the rows kernel)"
  timeout 200 python tools/octav_unit_iter_bench.py 128 0.02 2>&1 | grep max_iter
  echo "# ... and with MI355Q_OCTAV_UNIT_LANES=0 (octav_groups_kernel for every blockwise unit length: round 5)"
  MI355Q_OCTAV_UNIT_LANES=0 timeout 200 python tools/octav_block_bench.py 2>&1 | grep -v amdgpu.ids | head -4
  timeout 200 python tools/octav_block_bench.py 2>&1 | grep -v amdgpu.ids
  echo "# tools/api_resident_bench.py (get_tensor_quant_params on HBM-resident weights; batched = inside requant_queue.batching())"
  timeout 300 python tools/api_resident_bench.py 2>&1 | grep workload
} > "$OUT/octav_iterations_and_api_resident.txt" 2>&1
{
  echo "# tools/c5_model.py: BASELINE config 5 as one quantize_litertlm(calibration_data=...) call, 18 Gemma-2B-shaped layers, one GPU"
  for v in gptq mixed hadamard; do timeout 600 python tools/c5_model.py --layers 18 --variant $v 2>&1 | tail -1; done
  echo "# the same GPTQ run with the opt-in two-way float16 Hessian product (--hessian fast)"
  timeout 600 python tools/c5_model.py --layers 18 --variant gptq --hessian fast 2>&1 | tail -1
  echo "# tools/hinv_batched_bench.py: d = 2048 inverses, one call each vs mi355q_gptq_hinv_f64_batched"
  timeout 200 python tools/hinv_batched_bench.py 2048 54 2>&1 | tail -1
  echo "# tools/hinv_accuracy.py: time and error vs the exact FP64 inverse (bf16 split for the float32 steps, then FP64 throughout)"
  timeout 300 python tools/hinv_accuracy.py 4096 8192 16384 2>&1 | grep mode
  MI355Q_HINV_FP64=1 timeout 300 python tools/hinv_accuracy.py 4096 16384 2>&1 | grep mode
  echo "# tools/gptq_apply_bench.py"
  timeout 200 python tools/gptq_apply_bench.py 2>&1 | grep op
  echo "# tools/xtx_bench.py: Hessian product, two-way float16 split (xtx_f16x2.hip) against the three-way bfloat16 split"
  timeout 300 python tools/xtx_bench.py 2>&1 | grep tokens
  echo "# tools/io_ring_bench.py: 1 GiB file -> HBM and back through the io ring (csrc/file_io.hip)"
  timeout 200 python tools/io_ring_bench.py 2>&1 | grep GB/s
  echo "# the 18-layer run with its idle accounting (MI355Q_C5_GAPS=1) and per-call inverse / apply times (MI355Q_C5_TRACE)"
  MI355Q_C5_GAPS=1 MI355Q_C5_TRACE=hinv,apply timeout 600 python tools/c5_model.py --layers 18 --variant gptq 2>&1 | tail -1
} > "$OUT/c5_model.txt" 2>&1
{
  echo "# Where the wall clock of the whole-model calls goes: marks at the phase boundaries (MI355Q_TIMELINE=1, runtime.mark)."
  echo "# tools/c5_model.py --layers 18 --variant gptq (BASELINE config 5, exact Hessian product), two processes:"
  for i in 1 2; do MI355Q_TIMELINE=1 MI355Q_C5_GAPS=1 timeout 600 python tools/c5_model.py --layers 18 --variant gptq 2>/dev/null | tail -1 | python tools/timeline_print.py; done
  echo "# tools/file_bench.py (1.44 GB .tflite -> int4 blockwise-128 / int8 per-channel; every repeat, the last line is the best):"
  MI355Q_TIMELINE=1 timeout 300 python tools/file_bench.py --repeat 3 2>/dev/null | python tools/timeline_print.py
  MI355Q_TIMELINE=1 timeout 300 python tools/file_bench.py --repeat 3 --recipe wi8 2>/dev/null | python tools/timeline_print.py
} > "$OUT/c5_timeline.txt" 2>&1
{
  echo "# tools/octav_fast_bench.py: OCTAV clip search, exact (NumPy-order) kernels against the opt-in one-read kernel"
  timeout 400 python tools/octav_fast_bench.py 2>&1 | grep "^{"
  echo "# tools/hinv_pairs_bench.py (MI355Q_HINV_PAIRS=1: two d >= 4096 inverses in flight; default: one after the other)"
  MI355Q_HINV_PAIRS=1 timeout 300 python tools/hinv_pairs_bench.py 16384 4 2>&1 | tail -1
  timeout 300 python tools/hinv_pairs_bench.py 16384 4 2>&1 | tail -1
  echo "# tools/xtx_bound.py: the Hessian product with every clock / power reading (deep kernel, then round 4's wide kernel)"
  timeout 200 python tools/xtx_bound.py 3 2>/dev/null
  MI355Q_XTX_DEEP=0 timeout 200 python tools/xtx_bound.py 3 2>/dev/null
  echo "# tools/api_resident_timeline.py: 64 resident 4096 x 4096 weights through get_tensor_quant_params inside batching()"
  timeout 200 python tools/api_resident_timeline.py 2>&1 | tail -1
  echo "# tools/c4_bench.py through distributed.calibrate_sharded is in bench.json (sharded.c4_*); tools/c5_second_call.py --variant mixed:"
  MI355Q_TIMELINE=1 MI355Q_C5_GAPS=1 timeout 400 python tools/c5_second_call.py --variant mixed 2>/dev/null | cut -c1-1500
  echo "# tools/oscar_bench.py --api: OSCAR stage by stage (clip_bounds: the prefix route, round 6) and the whole public call"
  timeout 400 python tools/oscar_bench.py --api 2>&1 | grep "^{"
  echo "# the same clip search with the prefix route off (MI355Q_OSCAR_PREFIX=0: full sort + scan of every row, round 5's kernels)"
  MI355Q_OSCAR_PREFIX=0 timeout 400 python tools/oscar_bench.py 2>&1 | grep clip_bounds
} > "$OUT/round6_tools.txt" 2>&1
bash tools/pmc_traffic_refresh.sh "$ROUND" > "$OUT/pmc_traffic.log" 2>&1
bash tools/c2_pmc_refresh.sh "$ROUND" > "$OUT/c2_pmc.log" 2>&1
cp gpurun_out/${ROUND}_c2_pmc_traffic.txt gpurun_out/${ROUND}_c2_rowwise_int8_kernel_trace.txt gpurun_out/pmc_latest.json "$OUT/" 2>/dev/null
{
  echo "# tools/gptq_parity_instances.py: the d = 16384 full chain on several instances, GPU (exact / fast Hessian product) vs the oracle's own chain"
  timeout 1500 python tools/gptq_parity_instances.py 3 32 2>&1 | grep "^{"
} > "$OUT/gptq_parity_instances.txt" 2>&1
for a in "" "--resident"; do timeout 300 python tools/c4_bench.py --samples 128 $a 2>&1 | tail -1; done > "$OUT/c4_c5_public.txt"
for a in "" "--resident"; do timeout 600 python tools/c5_bench.py --samples 4 $a 2>&1 | tail -1; done >> "$OUT/c4_c5_public.txt"
timeout 300 python tools/file_bench.py --repeat 3 2>&1 | tail -1 >> "$OUT/c4_c5_public.txt"
timeout 300 python tools/file_bench.py --repeat 3 --recipe wi8 2>&1 | tail -1 >> "$OUT/c4_c5_public.txt"
ls -la "$OUT"
cat "$OUT/gpu_tests_tail.txt"
