#!/bin/bash
# Stall accounting of xtx_bf16x3_wide_kernel from SQ counters (one rocprofv3 pass per group), ON THE GPU BOX:  bash tools/xtx_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05d
mkdir -p "$O"
export TMPDIR=/tmp
cat > /tmp/xtx_once.py <<PY
import os, sys
sys.path.insert(0, os.path.join("$R", "ai-edge-quantizer_amd")); sys.path.insert(0, "$R")
import __graft_entry__ as g; g.build()
import torch
from mi355q import ops
x = torch.randn((16384, 16384), device="cuda")
for _ in range(3):
  ops.gptq_xtx_accum(x, None)
torch.cuda.synchronize()
PY
cd /tmp
n=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  n=$((n+1))
  rm -rf /tmp/prof_x$n
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_x$n -o p -- python /tmp/xtx_once.py > /tmp/x$n.log 2>&1
  python - /tmp/prof_x$n "$grp" <<'PY'
import glob, sqlite3, sys
path, grp = sys.argv[1], sys.argv[2]
dbs = sorted(glob.glob(path + "/**/*.db", recursive=True))
if not dbs:
  print("#", grp, ": no database"); sys.exit(0)
c = sqlite3.connect(dbs[-1])
try:
  rows = c.execute("select k.name, p.counter_name, count(*), avg(p.value), avg(k.duration) from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id where k.name like '%xtx_bf16x3%' group by k.name, p.counter_name").fetchall()
except Exception as e:
  print("#", grp, ":", e); sys.exit(0)
for name, counter, n, avg, dur in rows:
  print(f"{counter:34s} {avg:18.1f}   (mean of {n} dispatches, {dur/1e3:9.1f} us each)  {name.split('(')[0][-40:]}")
PY
done > "$O/xtx_pmc_stalls.txt" 2>&1
cat "$O/xtx_pmc_stalls.txt"
