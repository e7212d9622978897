#!/usr/bin/env bash
# Runs on the GPU box: the supporting benches of round 2 (C2, C3, C5, fit, rank replay, TRSV, factor schedules), written under
# gpurun_out/final2/ (copy what is judged to profiles/).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final2; mkdir -p $out
timeout -k 10 200 python3 bench.py --workload c2 --no-cpu-baseline > $out/c2.json 2> $out/c2.err; echo "c2 rc=$?"
timeout -k 10 300 python3 bench.py --workload c3 --no-cpu-baseline > $out/c3.json 2> $out/c3.err; echo "c3 rc=$?"
timeout -k 10 200 python3 tools/bench_gpc.py > $out/c5.json 2> $out/c5.err; echo "c5 rc=$?"
timeout -k 10 200 python3 tools/bench_fit.py > $out/fit.json 2> $out/fit.err; echo "fit rc=$?"
timeout -k 10 300 python3 tools/trsv_bench.py 8192 16384 32768 65536 > $out/trsv.txt 2>&1; echo "trsv rc=$?"
timeout -k 10 300 python3 tools/factor_bench.py 8192 12288 16384 20480 24576 32768 40960 > $out/factor.txt 2>&1; echo "factor rc=$?"
GPRC_SERVICE=0 timeout -k 10 300 python3 tools/factor_bench.py 8192 12288 16384 20480 24576 32768 40960 > $out/factor_noservice.txt 2>&1; echo "factor2 rc=$?"
timeout -k 10 200 python3 tools/service_trace.py 8192 > $out/svc_trace_8192.txt 2>&1; echo "trace rc=$?"
timeout -k 10 200 python3 tools/service_trace.py 16384 > $out/svc_trace_16384.txt 2>&1; echo "trace rc=$?"
timeout -k 10 200 python3 tools/potf2_trace.py 8192 > $out/potf2_trace.txt 2>&1; echo "potf2 rc=$?"
timeout -k 10 600 python3 tools/rank_replay.py --worlds 1,2,4,8 --ranks 0 > $out/rank_replay.json 2> $out/rank_replay.err; echo "replay rc=$?"
