#!/usr/bin/env bash
# SQ counters of the two long-K GEMM kernels that dominate the C4 step (solve_left_kernel 72 %, trailing_range_kernel 22 %):
# MFMA-pipe busy fraction, effective clock, and where the wave cycles go (issue stalls / parked / active by class).
# Counter passes serialise dispatches, so the factor service is off (GPRC_SERVICE=0: one fused launch per panel; neither kernel is
# part of the service).  The program itself follows `--` (no env/bash hop).  Output: gpurun_out/core/summary.txt.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/core; mkdir -p $out
export GPRC_SERVICE=0
W="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-gate --no-abi-host-path --no-vendor-parity ${CORE_ARGS:-}"
rocprofv3 -L > $out/avail.txt 2>&1 || true
pass() {  # name counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- $W > $out/$name.log 2>&1
  echo "$name rc=$?"
}
pass a SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64
pass c SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES
python3 tools/core_counter_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
rm -f $out/*/*/*kernel_trace.csv $out/*/*/*counter_collection.csv
