"""Builds profiles/r03_c4_pmc_traffic.json (read by bench.py for roofline.traffic) from the two rocprofv3 --pmc passes of
tools/collect_profiles.sh:  python tools/make_pmc_traffic.py gpurun_out/final <kernel symbol> <algorithmic bytes/launch>
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (the counter ticks once per 64 B of a 128-B request);
both counters are reported in kilobytes (1024 B).  Must run BEFORE collect_profiles.sh deletes the per-dispatch CSVs."""
import csv, glob, json, sys
d, sym, algo = sys.argv[1], sys.argv[2], float(sys.argv[3])


def per_launch(sub, counter):
    tot, ids = 0.0, set()
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sym in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
    return tot / max(len(ids), 1), len(ids)


fetch, n = per_launch("pmc_fetch", "FETCH_SIZE")
write, _ = per_launch("pmc_write", "WRITE_SIZE")
json.dump({"kernel": sym, "workload": "c4", "n_gpus": 1, "launches_profiled": n,
           "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B; MI355X_MICROARCH.md, HBM section), WRITE_SIZE as read",
           "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 1 --warmup 0; tools/collect_profiles.sh"},
          open("profiles/r03_c4_pmc_traffic.json", "w"), indent=1)
print(open("profiles/r03_c4_pmc_traffic.json").read())
