"""effective shader clock of the chain-kernel launches of a tests/probes/r05_c2_pmc.sh run: GRBM_GUI_ACTIVE (summed over the 8
XCDs) / 8 / the dispatch's duration (rocprofv3 counter_collection CSV: both in one row)."""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_c2_pmc") + "/c/**/*counter_collection.csv", recursive=True) + \
         glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_c2_pmc") + "/c/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "snr::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("void ", "").replace("snr::", "").split("(")[0]
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            acc[k].append((float(r["Counter_Value"]) / 8.0 / dur, dur))
for k, v in sorted(acc.items()):
    v = v[len(v) // 4:]     # skip the warm-up launches
    print(f"{k:34s} launches {len(v):3d}  duration {sum(d for _, d in v) / len(v) / 1e3:7.1f} us  effective clock {sum(c for c, _ in v) / len(v):.3f} GHz")
