#!/usr/bin/env bash
# A/B of the split panel chain (GPRC_CHAIN_SPLIT) on one box: the bit-identity tests under the split form, then factor_bench both ways.
set -uo pipefail
out=gpurun_out/chain_split; mkdir -p $out
GPRC_CHAIN_SPLIT=1 timeout -k 10 400 python -m pytest tests/test_gpu_device_level.py -x -q -m gpu > $out/tests_split.log 2>&1; echo "tests rc=$?" | tee $out/rc.txt
tail -5 $out/tests_split.log
grep -q "rc=0" $out/rc.txt || exit 1
for rep in 1 2; do
  for sp in 0 1; do
    echo "== GPRC_CHAIN_SPLIT=$sp (rep $rep)" | tee -a $out/ab.txt
    GPRC_CHAIN_SPLIT=$sp GPRC_BENCH_INV=1 timeout -k 10 200 python tools/factor_bench.py 8192 10240 12288 16384 20480 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
  done
done
GPRC_CHAIN_SPLIT=1 GPRC_LIB_SUFFIX=_cprof timeout -k 10 200 python tools/chain_prof.py 8192 2>&1 | grep -v amdgpu.ids > $out/prof.txt
tail -40 $out/prof.txt
