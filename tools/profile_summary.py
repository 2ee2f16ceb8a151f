"""Condense the rocprofv3 CSV outputs of tools/profile.sh into one markdown summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
command = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-hashgrid"
write_json = len(sys.argv) <= 4 or sys.argv[4] != "nojson"


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "").replace("snr::", "")
    return name.split("(")[0][:60]


print(f"# rocprofv3 summary `{tag}` — `{command}`\n")
stats = find(f"{tag}_trace/**/*kernel_stats.csv")
if stats:
    print("## kernel trace (--kernel-trace --stats), every launch of the command (warm-up and profiled repeats included)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for i, r in enumerate(csv.DictReader(open(stats))):
        if i >= 16:
            break
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
pmc_json = {}
for kind, title in (("pmc_sq", "SQ counters (avg per launch)"), ("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{tag}_{kind}/**/*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "snr::" in r["Kernel_Name"]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n## {title}\n")
    for k in sorted(acc):
        vals = ", ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items()))
        print(f"- `{k}` (n={len(next(iter(acc[k].values())))}): {vals}")
        if kind != "pmc_sq":
            for c, v in acc[k].items():
                e = pmc_json.setdefault(k.split("<")[0], {})
                # (template instances of one kernel — composite_train_reg_kernel<1> / <3> — are merged: weighted by launches)
                tot = e.get(c + "_KiB_per_launch", 0.0) * e.get(c + "_launches", 0) + sum(v)
                e[c + "_launches"] = e.get(c + "_launches", 0) + len(v)
                e[c + "_KiB_per_launch"] = tot / e[c + "_launches"]
print("\nFETCH_SIZE / WRITE_SIZE are in KiB per launch as reported; per MI355X_MICROARCH.md the read side of a wide "
      "coalesced stream is under-reported 2x on gfx950 (double FETCH_SIZE before comparing with byte counts).")

# machine-readable copy for bench.py's roofline.traffic / hbm_bytes_per_step (per-kernel averages over all launches of
# the default bench command; launches per step come from bench.py itself)
import json
if pmc_json and write_json:
    with open(os.path.join(out, f"{tag}_pmc.json"), "w") as fh:
        json.dump({"command": command,
                   "workload": "bf16, N_rand=1024, 64c+128f", "kernels": pmc_json}, fh, indent=1)
