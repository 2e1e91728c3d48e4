#!/bin/bash
# d = 16384 damped inverse under the Cholesky's switches: look-ahead on / off, outer block widths. Output: gpurun_out/r05_hinv/.
export TMPDIR=/tmp
out=gpurun_out/r05_hinv
mkdir -p $out
: > $out/variants.txt
for rep in 1 2; do
for v in "" "MI355Q_NO_LOOKAHEAD=1" "MI355Q_CHOL_OB=384" "MI355Q_CHOL_OB=640" "MI355Q_CHOL_OB=768" "MI355Q_CHOL_OB=1024" "MI355Q_FUSED_MAX_M=4096" "MI355Q_FUSED_MAX_M=1024" "MI355Q_CHOL_ASU=0"; do
  echo "## $v" >> $out/variants.txt
  env $v timeout 300 python tools/hinv_variant_bench.py default 16384 3 2>/dev/null | tail -1 >> $out/variants.txt
done
done
cat $out/variants.txt
