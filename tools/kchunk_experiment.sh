#!/usr/bin/env bash
# does cutting the long-K left-looking passes into K-chunks (re-aligning an XCD's tiles at every launch
# boundary) reduce the L2<->fabric over-fetch, and does that buy clock or TFLOP/s?  Runs on the GPU box (via gpurun).
# For GPRC_KCHUNK in {whole pass, 16, 8, 4 panels = K 8192, 4096, 2048}: the C4 bench (TFLOP/s of solve_left_kernel from
# the library's HIP events), one --pmc FETCH_SIZE pass and one --pmc GRBM_GUI_ACTIVE pass (own runs, --kernel-trace only).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kchunk; mkdir -p $out
W="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-gate"
for kc in 0 16 8 4; do
  export GPRC_KCHUNK=$kc
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_k$kc.json 2> $out/bench_k$kc.err || { echo "bench k=$kc failed"; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_k$kc -- $W > $out/fetch_k$kc.log 2>&1 || { echo "pmc fetch k=$kc failed"; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/gui_k$kc -- $W > $out/gui_k$kc.log 2>&1 || { echo "pmc gui k=$kc failed"; exit 1; }
done
unset GPRC_KCHUNK
python3 tools/kchunk_summary.py $out > $out/summary.txt
cat $out/summary.txt
rm -rf $out/fetch_k* $out/gui_k*
