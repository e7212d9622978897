#!/usr/bin/env bash
# A/B of the shared service (GPRC_SERVICE_SHARE) on one box: a bit-for-bit soak with it on, then factor_bench both ways.
set -uo pipefail
out=gpurun_out/share; mkdir -p $out
GPRC_SERVICE_SHARE=1 timeout -k 10 300 python tools/chain_soak.py 12 16384 12288 20480 2>&1 | grep -v amdgpu.ids | tail -4 | tee $out/soak.txt
grep -q "done: .* 0 differing" $out/soak.txt || exit 1
for rep in 1 2; do
  for sh in 0 1; do
    echo "== GPRC_SERVICE_SHARE=$sh (rep $rep)" | tee -a $out/ab.txt
    GPRC_SERVICE_SHARE=$sh GPRC_BENCH_INV=1 timeout -k 10 300 python tools/factor_bench.py 12288 16384 20480 32768 65536 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
  done
done
