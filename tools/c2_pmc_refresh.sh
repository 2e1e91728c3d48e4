R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 30 --warmup 5 --cpu-seconds 0 --extras 0"
python $R/bench.py $ARGS > $R/gpurun_out/bench_c2.log 2> /tmp/b.err; tail -c 200 /tmp/b.err
for n in prof_c2 pmc_fetch pmc_write; do rm -rf $R/gpurun_out/$n; done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c2 -o p -- python $R/bench.py $ARGS > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py $ARGS > /tmp/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py $ARGS > /tmp/p3.log 2>&1
cd $R; python tools/make_c2_profiles.py r03x gpurun_out/bench_c2.log | head -30
cp profiles/r03x_c2_pmc_traffic.txt profiles/r03x_c2_rowwise_int8_kernel_trace.txt profiles/pmc_latest.json gpurun_out/ 2>/dev/null
ls -la gpurun_out/ | grep -E "r03x|pmc_latest"
du -sh gpurun_out/prof_c2 gpurun_out/pmc_fetch gpurun_out/pmc_write
rm -rf gpurun_out/prof_c2 gpurun_out/pmc_fetch gpurun_out/pmc_write
