#!/bin/bash
# L2 hit rate and HBM-side fetch of the Hessian product kernel, two workgroups per CU (ring of 2) against one (ring of 3); GPU box
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp
for dpt in 2 3; do
  echo "== MI355Q_XTX_DEPTH=$dpt, d = 16384, 16384 tokens per dispatch"
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    rm -rf /tmp/xp; MI355Q_XTX_DEPTH=$dpt timeout 150 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/xp -o p -- python $R/tools/xtx_profile.py 16384 16384 > /tmp/xp.log 2>&1
    python $R/tools/pmc_kernel.py /tmp/xp xtx_f16x2 | sed 's/  */ /g'
  done
done
