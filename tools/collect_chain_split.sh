#!/usr/bin/env bash
# Runs on the GPU box: the record behind profiles/r03_chain_split.txt -- factor_bench both chain forms (same box, alternating), the
# service timeline and the chain profile at n = 8192, and the mid-size configurations (C2 bench line, C5 GPC, fit gradient) at the default.
set -uo pipefail
out=gpurun_out/chain_final; mkdir -p $out
for rep in 1 2; do
  for sp in 0 1; do
    echo "== GPRC_CHAIN_SPLIT=$sp (rep $rep)" | tee -a $out/ab.txt
    GPRC_CHAIN_SPLIT=$sp GPRC_BENCH_INV=1 timeout -k 10 200 python tools/factor_bench.py 4096 6144 8192 10240 12288 14336 16384 18432 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
  done
done
for sp in 0 1; do echo "== GPRC_CHAIN_SPLIT=$sp"; GPRC_CHAIN_SPLIT=$sp GPRC_SERVICE_TRACE=1 timeout -k 10 200 python tools/service_trace.py 8192 2>&1 | grep -v amdgpu.ids; done > $out/trace.txt 2>&1
tail -1 $out/trace.txt
GPRC_LIB_SUFFIX=_cprof timeout -k 10 200 python tools/chain_prof.py 8192 2>&1 | grep -v amdgpu.ids > $out/prof.txt
for sp in 0 1; do
  GPRC_CHAIN_SPLIT=$sp timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline 2>&1 | grep "^{" > $out/c2_split$sp.json
  python -c "import json;d=json.load(open('$out/c2_split$sp.json'));print('C2 split=$sp step ms',d['ms_per_step'],'fit',d['phases_ms']['fit_F1_F3'],'TFLOP/s',d['value'])" | tee -a $out/configs.txt
  GPRC_CHAIN_SPLIT=$sp timeout -k 10 300 python tools/bench_gpc.py 2>&1 | grep "^{" > $out/c5_split$sp.json
  python -c "import json;d=json.load(open('$out/c5_split$sp.json'));print('C5 split=$sp ms per IRLS iteration',d['ms_per_irls_iteration'],'fit',d['fit_ms'],'potrf TFLOP/s',d['potrf_tflops_per_iteration'])" | tee -a $out/configs.txt
done
timeout -k 10 300 python tools/bench_fit.py 16384 8 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $out/configs.txt
