#!/usr/bin/env bash
# MFMA utilisation of the two GEMM kernels and effective clock, from PMC counters (own pass, --kernel-trace only):
#   busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/mfma; mkdir -p $out
W="python3 bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline --no-parity-gate"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -- $W > $out/pmc.log 2>&1; echo "pmc rc=$?"
python3 - <<'PY' > $out/mfma_util_summary.txt
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob("gpurun_out/mfma/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((s for s in ("gemm_nt_kernel<5>", "trailing_kernel") if s in r["Kernel_Name"]), None)
        if k: rows[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for f in glob.glob("gpurun_out/mfma/pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((s for s in ("gemm_nt_kernel<5>", "trailing_kernel") if s in r["Kernel_Name"]), None)
        if k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
print("workload: bench.py --workload c3 (n = 32768, rationalquadratic), one step, rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
for k in rows:
    busy, gui = rows[k]["SQ_VALU_MFMA_BUSY_CYCLES"], rows[k]["GRBM_GUI_ACTIVE"]
    cyc = gui / 8.0
    print(f"{k}: dispatches {len(n[k])}, MFMA busy cycles {busy:.4g}, GRBM_GUI_ACTIVE {gui:.4g} (/8 XCDs = {cyc:.4g} cycles), "
          f"busy / (1024 SIMDs x cycles) = {busy / (1024 * cyc):.3f}, effective clock = {cyc / dur[k] * 1e-9:.2f} GHz over {dur[k]*1e3:.1f} ms")
PY
cat $out/mfma_util_summary.txt
rm -f $out/pmc/*/*kernel_trace.csv $out/pmc/*/*counter_collection.csv
