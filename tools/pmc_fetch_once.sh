#!/usr/bin/env bash
# one --pmc FETCH_SIZE pass of the C4 bench step; prints per-kernel totals (tools/pmc_summary.py).  Runs on the GPU box.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_once_$1; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-gate > $out.log 2>&1 || { echo "pmc pass failed"; tail -5 $out.log; exit 1; }
python3 tools/pmc_summary.py $out solve_left_kernel trailing_range_kernel
rm -rf $out
