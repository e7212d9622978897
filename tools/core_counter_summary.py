"""Summarises tools/collect_core_counters.sh: per kernel, the sum of every collected SQ / GRBM counter over its dispatches, the
kernel's wall time from the kernel trace of the same pass, and the derived figures DESIGN.md quotes."""
import collections, csv, glob, sys
root = sys.argv[1]
KERNELS = ("solve_left_kernel", "trailing_range_kernel", "gemm_nt_kernel<5>", "trailing_kernel", "solve_panel_fused_kernel")
val = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(lambda: collections.defaultdict(float))
for pas in sorted(glob.glob(root + "/[a-z]")):
    name = pas.rsplit("/", 1)[1]
    for f in glob.glob(pas + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((s for s in KERNELS if s in r["Kernel_Name"]), None)
            if k:
                val[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    for f in glob.glob(pas + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((s for s in KERNELS if s in r["Kernel_Name"]), None)
            if k:
                dur[k][name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
print("workload: bench.py (C4: n = 65536, n* = 65536), one step, GPRC_SERVICE=0, rocprofv3 --pmc in three passes (a, b, c)")
for k in KERNELS:
    if k not in val:
        continue
    v = val[k]
    print(f"\n== {k}: {len(disp[k].get('GRBM_GUI_ACTIVE', disp[k][next(iter(disp[k]))]))} dispatches; wall per pass " +
          ", ".join(f"{p} {t * 1e3:.1f} ms" for p, t in sorted(dur[k].items())))
    for c in sorted(v):
        print(f"   {c:34s} {v[c]:.6g}")
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        t = dur[k].get("a", 0.0)
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        clk = cyc / t * 1e-9 if t else float("nan")
        print(f"   -> kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs) {cyc:.4g}; effective clock {clk:.3f} GHz; MFMA busy / (1024 SIMDs x cycles) = {busy:.4f}")
        print(f"   -> busy x clock / 2.4 GHz = {busy * clk / 2.4:.4f} of the 78.6 TFLOP/s peak (quoted at 2.4 GHz)")
    if "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in v:
                print(f"   -> {c} / SQ_WAVE_CYCLES = {v[c] / wc:.4f}")
    if "SQ_WAIT_INST_LDS" in v and "SQ_WAVE_CYCLES" in val[k]:
        pass
