"""Summarise rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py DIR [name-substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
subs = sys.argv[2:] or ["gemm_nt_kernel<5>", "trailing_kernel", "fill_kernel", "gemm_nt_kernel<3>", "trsm_panel"]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = next((s for s in subs if s in name), None)
        if key is None:
            continue
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(cnt[k]))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {v:.6g}   per dispatch {v/len(cnt[k]):.6g}")
