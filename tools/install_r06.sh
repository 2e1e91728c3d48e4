#!/bin/bash
# Copies what tools/refresh_profiles.sh r06 left under gpurun_out/r06/ into profiles/r06_* (a header line each says where it came from),
# then regenerates the documents' generated blocks.   bash tools/install_r06.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r06
P=$R/profiles/r06_
put() {  # source, target, header
  { echo "# $3"; cat "$O/$1"; } > "${P}$2"
}
cp "$O/bench.json" "${P}bench.json"
put bench_kernel_trace.txt bench_default_kernel_trace.txt "rocprofv3 --kernel-trace of: python bench.py --steps 20 --warmup 5 (tools/refresh_profiles.sh r06, tools/rocprof_summary.py)"
put paths_kernel_trace.txt paths_kernel_trace.txt "rocprofv3 --kernel-trace of: python tools/path_bench.py --big (tools/refresh_profiles.sh r06)"
put path_bench.txt public_paths.txt "python tools/path_bench.py --big: every kernel of the path through mi355q.ops, one JSON line per op"
cat "$O/c4_c5_public.txt" >> "${P}public_paths.txt"
put gptq_mfma_util.txt gptq_mfma_util.txt "NOTE: a PMC pass serialises and replays every dispatch -- clocks derived here are NOT those of the un-profiled kernels (profiles/r06_xtx_bound.txt)"
put hinv_phases.txt hinv_phases.txt "phases of the damped inverse (tools/refresh_profiles.sh r06)"
put hinv16384_fetch.txt hinv16384_fetch.txt "FETCH_SIZE of the d = 16384 inverse's large kernels"
put octav_iterations_and_api_resident.txt octav_iterations.txt "OCTAV per iteration (exact kernels), per unit length, and the public call on resident weights"
put c5_model.txt c5_model.txt "BASELINE config 5 and its parts (tools/refresh_profiles.sh r06)"
put c5_timeline.txt c5_timeline.txt "where the wall clock of the whole-model calls goes"
put round6_tools.txt round6_tools.txt "the rounds' own tools: OCTAV one-read kernel, two inverses in flight, Hessian product bound, public-call timeline, C5 mixed second call, OSCAR stage by stage with and without the prefix route"
put gptq_parity_instances.txt gptq_parity_instances.txt "the d = 16384 full chain on several instances"
{ echo "# HBM bytes per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections) against the algorithmic bytes of SURVEY 8d (tools/pmc_traffic_refresh.sh r06)"; cat "$O/pmc_traffic_alg.txt" "$O/pmc_traffic_raw.txt"; } > "${P}pmc_traffic.txt"
for f in r06_c2_pmc_traffic.txt r06_c2_rowwise_int8_kernel_trace.txt; do [ -f "$O/$f" ] && cp "$O/$f" "$R/profiles/$f"; done
[ -f "$O/pmc_latest.json" ] && cp "$O/pmc_latest.json" "$R/profiles/pmc_latest.json"
cp "$O/parity_rates.jsonl" "${P}parity_rates.jsonl"
python "$R/tools/parity_rates_summary.py" "$O/parity_rates.jsonl" > "${P}parity_rates.txt"
cp "$O/gpu_tests_tail.txt" "${P}gpu_tests_tail.txt"
python "$R/tools/parity_docs.py" "$R/profiles/r06_parity_rates.jsonl"
python "$R/tools/kernel_table.py" r06
ls -la "$R/profiles" | grep r06_
