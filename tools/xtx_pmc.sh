#!/bin/bash
# PMC passes over the Hessian product kernel (run on the GPU box):  bash tools/xtx_pmc.sh [d] [tokens]
R=${GRAFT_REPO_ROOT:-$PWD}; D=${1:-16384}; N=${2:-16384}
export TMPDIR=/tmp; cd /tmp
pass() {
  rm -rf /tmp/xp; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/xp -o p -- python "$R/tools/xtx_profile.py" $D $N > /tmp/xp.log 2>&1
  python "$R/tools/pmc_kernel.py" /tmp/xp xtx_f16x2 | sed 's/  */ /g'
}
pass SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16
pass TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
python "$R/tools/rocprof_summary.py" /tmp/xp 2>&1 | head -8
