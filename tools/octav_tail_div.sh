#!/bin/bash
# Exact OCTAV rows kernel: when to hand a row over to the tail kernel (candidate capacity = len / DIV) against time and HBM writes.
export TMPDIR=/tmp
out=gpurun_out/r05_octav_tail
mkdir -p $out
: > $out/times.txt
for div in 8 16 32 64 128; do
  echo "## MI355Q_OCTAV_TAIL_DIV=$div" >> $out/times.txt
  MI355Q_OCTAV_TAIL_DIV=$div timeout 300 python tools/octav_fast_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  r = json.loads(l)
  if r['unit_len'] in (2048, 4096, 11008, 16384): print(r['units'], r['unit_len'], r['sigma'], 'exact us', r['exact']['us'], 'fast us', r['fast']['us'])
" >> $out/times.txt
done
echo "## MI355Q_OCTAV_TAIL=0" >> $out/times.txt
MI355Q_OCTAV_TAIL=0 timeout 300 python tools/octav_fast_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  r = json.loads(l)
  if r['unit_len'] in (2048, 4096, 11008, 16384): print(r['units'], r['unit_len'], r['sigma'], 'exact us', r['exact']['us'])
" >> $out/times.txt
cat $out/times.txt
# HBM bytes of the rows kernel per divisor (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/traffic_driver.py)
R=$PWD
for div in 8 16 32; do
  cd /tmp; rm -rf /tmp/pmc_f /tmp/pmc_w
  MI355Q_OCTAV_TAIL_DIV=$div timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o p -- python "$R/tools/traffic_driver.py" > /dev/null 2>&1
  MI355Q_OCTAV_TAIL_DIV=$div timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o p -- python "$R/tools/traffic_driver.py" > /dev/null 2>&1
  cd "$R"
  echo "## HBM traffic, MI355Q_OCTAV_TAIL_DIV=$div" >> $out/times.txt
  python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w octav_rows_kernel octav_tail_kernel >> $out/times.txt 2>&1
done
tail -30 $out/times.txt
