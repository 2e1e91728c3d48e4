#!/bin/bash
# A / B of the FP64 GEMM epilogue on the GPU box (libraries built beforehand: tools/build_variant.sh serial|earlyc gemm.hip -D...):
#   gpurun -- 'bash tools/gemm_epilogue_ab.sh'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R"; export TMPDIR=/tmp
O=$R/gpurun_out/gemm_ab; mkdir -p "$O"
V=$R/tools/kbench/_variants
D=$R/ai-edge-quantizer_amd/lib/libmi355q.so
timeout 600 python tools/gemm_bench.py --libs "$D;$V/libmi355q_serial.so" --rounds 5 > "$O/gemm_bench.txt" 2>&1
for rep in 1 2; do
  for lib in default "$V/libmi355q_serial.so"; do
    timeout 300 python tools/hinv_variant_bench.py "$lib" 16384 3 2>&1 | grep ms_per_inverse >> "$O/hinv.txt"
  done
done
for lib in default "$V/libmi355q_serial.so"; do
  for d in 2048 4096 8192; do timeout 300 python tools/hinv_variant_bench.py "$lib" $d 4 2>&1 | grep ms_per_inverse >> "$O/hinv.txt"; done
done
timeout 900 python -m pytest tests/test_gpu_gptq.py tests/test_gpu_gptq_c5.py -q -x 2>&1 | tail -3 > "$O/gptq_tests.txt"
cat "$O/gemm_bench.txt" "$O/hinv.txt" "$O/gptq_tests.txt"
