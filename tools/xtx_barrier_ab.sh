#!/bin/bash
# A / B on the GPU box: the ring-buffered Hessian product kernels with the unfenced barrier (this commit) against __syncthreads()
# (libmi355q_fenced.so = tools/build_variant.sh fenced xtx_bf16x3.hip -DMI355Q_FENCED_BARRIER=1)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R"; export TMPDIR=/tmp
O=$R/gpurun_out/xtx_ab; mkdir -p "$O"
V=$R/tools/kbench/_variants
D=$R/ai-edge-quantizer_amd/lib/libmi355q.so
for rep in 1 2; do
  for lib in "$D" "$V/libmi355q_fenced.so"; do
    timeout 300 python tools/with_lib.py "$lib" tools/xtx_bench.py 16384 16384 2048 65536 2>&1 | grep "libmi355q\|tokens" >> "$O/xtx.txt"
  done
done
for depth in 2 3 4; do
  echo "# MI355Q_XTX_DEPTH=$depth (two-way float16 kernel's ring)" >> "$O/xtx.txt"
  MI355Q_XTX_DEPTH=$depth timeout 300 python tools/xtx_bench.py 16384 16384 2>&1 | grep tokens >> "$O/xtx.txt"
done
timeout 900 python -m pytest tests/test_gpu_gptq.py tests/test_gpu_gptq_c5.py tests/test_gpu_c5_model.py -q -x 2>&1 | tail -3 > "$O/gptq_tests.txt"
cat "$O/xtx.txt" "$O/gptq_tests.txt"
