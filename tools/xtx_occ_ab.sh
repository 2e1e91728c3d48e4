#!/bin/bash
# L2 hit rate and HBM-side fetch of the Hessian product kernels (d = 16384, 16384 tokens per dispatch); GPU box
#   default: 128 x 256 tiles (xtx_f16x2_wide_kernel); MI355Q_XTX_NARROW=1: 128 x 128 tiles, two workgroups per CU;
#   ... with MI355Q_XTX_DEPTH=3: one workgroup per CU
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp
run() {
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    rm -rf /tmp/xp; env "$@" timeout 150 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/xp -o p -- python $R/tools/xtx_profile.py 16384 16384 > /tmp/xp.log 2>&1
    python $R/tools/pmc_kernel.py /tmp/xp xtx_f16x2 | sed 's/  */ /g'
  done
}
echo "== 128 x 256 tiles"; run A=1
echo "== 128 x 128 tiles, two workgroups per CU"; run MI355Q_XTX_NARROW=1
echo "== 128 x 128 tiles, one workgroup per CU"; run MI355Q_XTX_NARROW=1 MI355Q_XTX_DEPTH=3
