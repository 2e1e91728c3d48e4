#!/bin/bash
# A variant of libmi355q.so for A / B timing on the GPU box, built HERE (hipcc cross-compiles; the box's minutes are for measuring):
#   tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]
# compiles that ONE source with the extra flags, links it with the in-tree objects of the others (run __graft_entry__.py first)
# and leaves tools/kbench/_variants/libmi355q_<name>.so (git-ignored like every .so; it travels with gpurun's snapshot).
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
out=$R/tools/kbench/_variants; mkdir -p "$out"
obj=$out/$(basename "${src%.*}")_$name.o
flags=$(python - <<PY
import sys; sys.path.insert(0, "$R")
import __graft_entry__ as g
print(" ".join(g.COMPILE_FLAGS))
PY
)
/opt/rocm/bin/hipcc $flags -c -I"$R/include" -I"$R/ai-edge-quantizer_amd/csrc" "$@" "$R/ai-edge-quantizer_amd/csrc/$src" -o "$obj"
others=$(ls "$R"/ai-edge-quantizer_amd/lib/obj/*.o | grep -v "/$(basename "${src%.*}").o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others "$obj" -ldl -o "$out/libmi355q_$name.so"
rm -f "$obj"
echo "$out/libmi355q_$name.so"
