#!/bin/bash
# The d = 16384 Hessian inverse inside a 3-layer C5 run against the same inverse alone: per-kernel averages and the first kernels of the chain
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/p1 /tmp/p2
timeout 300 rocprofv3 --kernel-trace -d /tmp/p1 -o p -- python $R/tools/c5_model.py --layers 3 --variant gptq > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/p2 -o p -- python $R/tools/hinv_profile.py 16384 > /tmp/p2.log 2>&1
echo "== in situ (3-layer C5)"
python $R/tools/chain_window.py /tmp/p1 -1 0 40
echo "== standalone hinv 16384"
python $R/tools/chain_window.py /tmp/p2 -1 0 40
