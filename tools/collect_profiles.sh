#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + PMC traffic passes for the default
# bench command, written under gpurun_out/final/ (copy the summaries to profiles/ afterwards).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final; mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench_c4.log 2>&1; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 bench.py --no-cpu-baseline > $out/kstats.log 2>&1; echo "kstats rc=$?"
# counter collection serialises kernel dispatches: the factor service (a persistent launch beside the caller's kernels) cannot run
# under it, so the PMC passes use one fused launch per panel instead -- the dominant kernel (solve_left_kernel, predict) is the same
export GPRC_SERVICE=0
W="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-gate"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $W > $out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $W > $out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python3 tools/pmc_summary.py $out/pmc_fetch solve_left_kernel trailing_range_kernel "gemm_nt_kernel<5>" trailing_kernel fill_kernel panel_fused_kernel > $out/pmc_fetch_summary.txt
python3 tools/pmc_summary.py $out/pmc_write solve_left_kernel trailing_range_kernel "gemm_nt_kernel<5>" trailing_kernel fill_kernel panel_fused_kernel > $out/pmc_write_summary.txt
cat $out/pmc_fetch_summary.txt $out/pmc_write_summary.txt
python3 tools/make_pmc_traffic.py $out solve_left_kernel 0 > /dev/null; cp profiles/r03_c4_pmc_traffic.json $out/pmc_traffic.json
rm -f $out/*/*/*kernel_trace.csv $out/pmc_*/*/*counter_collection.csv
grep "^{" $out/bench_c4.log | cut -c1-400
