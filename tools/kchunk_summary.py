"""Digest of tools/kchunk_experiment.sh: per GPRC_KCHUNK variant, for solve_left_kernel and trailing_range_kernel:
launches, TFLOP/s (HIP events, bench JSON), FETCH_SIZE per step (x2: gfx950 counts 128-B requests at 64 B,
MI355X_MICROARCH.md), and the effective clock GRBM_GUI_ACTIVE / 8 XCDs / kernel time."""
import csv, glob, json, sys, collections
d = sys.argv[1]
KERNELS = ("solve_left_kernel", "trailing_range_kernel")
print("GPRC_KCHUNK (panels; 0 = whole pass) | kernel | launches/step | TFLOP/s (events) | step ms | FETCH_SIZE x2 GB per step | GB per launch | eff. clock GHz")
for kc in (0, 16, 8, 4):
    try:
        j = json.loads([l for l in open(f"{d}/bench_k{kc}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(kc, "bench missing", e); continue
    fetch, nl = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(f"{d}/fetch_k{kc}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((s for s in KERNELS if s in r["Kernel_Name"]), None)
            if k and r["Counter_Name"] == "FETCH_SIZE":
                fetch[k] += float(r["Counter_Value"]); nl[k].add(r["Dispatch_Id"])
    gui, dur = collections.defaultdict(float), collections.defaultdict(float)
    for f in glob.glob(f"{d}/gui_k{kc}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((s for s in KERNELS if s in r["Kernel_Name"]), None)
            if k and r["Counter_Name"] == "GRBM_GUI_ACTIVE": gui[k] += float(r["Counter_Value"])
    for f in glob.glob(f"{d}/gui_k{kc}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((s for s in KERNELS if s in r["Kernel_Name"]), None)
            if k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    for k, name in zip(KERNELS, ("solve_left", "trailing_left")):
        kk = j["kernels"].get(name, {})
        gb = 2.0 * fetch[k] * 1024.0 / 1e9
        clk = gui[k] / 8.0 / dur[k] * 1e-9 if dur[k] else float("nan")
        print(f"{kc:>2} | {k} | {kk.get('launches', 0) // j['steps']} | {kk.get('tflops')} | {j['ms_per_step']} | {gb:.1f} | {gb / max(len(nl[k]), 1):.2f} | {clk:.2f}")
