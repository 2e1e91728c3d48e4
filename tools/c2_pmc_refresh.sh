#!/bin/bash
# The headline (C2) kernel's evidence for a round, on the GPU box:  gpurun -- 'bash tools/c2_pmc_refresh.sh r03'
#   un-profiled bench line, rocprofv3 kernel trace, FETCH_SIZE and WRITE_SIZE passes (separate, as MI355X_MICROARCH.md prescribes)
#   -> gpurun_out/<tag>_c2_pmc_traffic.txt, <tag>_c2_rowwise_int8_kernel_trace.txt, pmc_latest.json (copy them into profiles/)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 30 --warmup 5 --cpu-seconds 0 --extras 0"
python $R/bench.py $ARGS > $R/gpurun_out/bench_c2.log 2> /tmp/b.err
for n in prof_c2 pmc_fetch pmc_write; do rm -rf $R/gpurun_out/$n; done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c2 -o p -- python $R/bench.py $ARGS > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py $ARGS > /tmp/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py $ARGS > /tmp/p3.log 2>&1
cd $R; python tools/make_c2_profiles.py $TAG gpurun_out/bench_c2.log | head -30
cp profiles/${TAG}_c2_pmc_traffic.txt profiles/${TAG}_c2_rowwise_int8_kernel_trace.txt profiles/pmc_latest.json gpurun_out/
rm -rf gpurun_out/prof_c2 gpurun_out/pmc_fetch gpurun_out/pmc_write
