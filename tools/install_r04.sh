#!/bin/bash
# What tools/refresh_profiles.sh r04 (and tools/c2_pmc_refresh.sh r04) left under gpurun_out/ -> profiles/r04_* (run here, after the
# GPU call has merged its gpurun_out/). The parity paragraphs of DESIGN.md / README.md are regenerated from the copied record.
set -eu
cd "$(dirname "$0")/.."
R=gpurun_out/r04
cp $R/bench.json profiles/r04_bench.json
cp $R/bench_kernel_trace.txt profiles/r04_bench_default_kernel_trace.txt
cp $R/paths_kernel_trace.txt profiles/r04_paths_kernel_trace.txt
cp $R/hinv_phases.txt profiles/r04_hinv_phases.txt
cp $R/c5_model.txt profiles/r04_c5_model.txt
cp $R/c5_timeline.txt profiles/r04_c5_timeline.txt
cp $R/octav_iterations_and_api_resident.txt profiles/r04_octav_iterations.txt
cp $R/gptq_mfma_util.txt profiles/r04_gptq_mfma_util.txt
cp $R/gptq_parity_instances.txt profiles/r04_gptq_parity_instances.txt
cat $R/pmc_traffic_alg.txt $R/pmc_traffic_raw.txt > profiles/r04_pmc_traffic.txt
cat $R/path_bench.txt $R/c4_c5_public.txt > profiles/r04_public_paths.txt
cp $R/parity_rates.jsonl profiles/r04_parity_rates.jsonl
python tools/parity_rates_summary.py profiles/r04_parity_rates.jsonl > profiles/r04_parity_rates.txt
cp gpurun_out/r04_c2_pmc_traffic.txt gpurun_out/r04_c2_rowwise_int8_kernel_trace.txt gpurun_out/pmc_latest.json profiles/
python tools/parity_docs.py profiles/r04_parity_rates.jsonl
tail -1 $R/gpu_tests_tail.txt
