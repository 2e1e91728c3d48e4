#!/bin/bash
# second A / B on the GPU box: this commit's library against the rounds 1-4 GEMM (libmi355q_serial.so: per-element epilogue, HIP float4 / double2 staging)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R"; export TMPDIR=/tmp
O=$R/gpurun_out/gemm_ab2; mkdir -p "$O"
V=$R/tools/kbench/_variants
D=$R/ai-edge-quantizer_amd/lib/libmi355q.so
timeout 600 python tools/gemm_bench.py --libs "$V/libmi355q_serial.so;$D" --rounds 5 > "$O/gemm_bench.txt" 2>&1
for rep in 1 2; do
  for lib in "$D" "$V/libmi355q_serial.so"; do
    timeout 300 python tools/with_lib.py "$lib" tools/gptq_apply_bench.py 2>&1 | grep "libmi355q\|gptq_apply" >> "$O/apply.txt"
    timeout 300 python tools/with_lib.py "$lib" tools/hinv_batched_bench.py 2048 54 2>&1 | tail -1 >> "$O/apply.txt"
  done
done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > "$O/gpu_tests.txt"
cat "$O/gemm_bench.txt" "$O/apply.txt" "$O/gpu_tests.txt"
