#!/bin/bash
# Builds the two stand-alone probes tools/refresh_profiles.sh runs (git-ignored binaries; they travel with gpurun when they
# were built in the build container, and are built here -- hipcc is on the GPU box too -- when they are missing or older than
# their sources).   bash tools/build_kbench.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
C=ai-edge-quantizer_amd/csrc
newer() { [ ! -x "$1" ] || [ -n "$(find "${@:2}" -newer "$1" 2>/dev/null | head -1)" ]; }
if newer tools/kbench/potf2_bench tools/kbench/potf2_bench.hip $C/gptq.hip $C/gemm.hip $C/common.h $C/gemm.h; then
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DMI355Q_POTF2_PROF -I include -I $C \
    tools/kbench/potf2_bench.hip $C/api.cpp $C/gemm.hip $C/xtx_bf16x3.hip $C/xtx_f16x2.hip $C/file_io.hip -lpthread -ldl \
    -o tools/kbench/potf2_bench
fi
if newer tools/kbench/lat_bench tools/kbench/lat_bench.hip; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/lat_bench.hip -o tools/kbench/lat_bench
fi
ls -la tools/kbench/potf2_bench tools/kbench/lat_bench
