#!/bin/bash
# HBM traffic per launch of the HBM-bound kernels against their algorithmic bytes, on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/pmc_traffic_refresh.sh r03'
set -u
ROUND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o p -- python "$R/tools/traffic_driver.py" > "$OUT/traffic_driver.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o p -- python "$R/tools/traffic_driver.py" > /dev/null 2>&1
cd "$R"
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w requant_groups_kernel act_minmax_kernel fwht_tile_kernel octav_rows_kernel octav_tail_kernel octav_unit_lanes_kernel octav_fast_kernel gptq_rows_kernel mse_scale_balanced_kernel clip_prefix_kernel minmax_runs_kernel quantize_rows_vec4_kernel dequantize_rows_vec4_kernel > "$OUT/pmc_traffic_raw.txt" 2>&1
grep "^ALG" "$OUT/traffic_driver.log" > "$OUT/pmc_traffic_alg.txt"
cat "$OUT/pmc_traffic_raw.txt" "$OUT/pmc_traffic_alg.txt"
