#!/bin/bash
# Instruction-cache counters of the MLP kernels inside the real training step (`bench.py --steps 10`): the chain kernels are
# 72-90 KB and the pair kernel 198 KB of code against a 64 KB instruction cache shared by two CUs.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_icache; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-hashgrid --blocks 1"
timeout -k 5 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE \
  --kernel-trace --output-format csv -d $OUT/a -o pmc -- $B > $OUT/a.out 2> $OUT/a.err
timeout -k 5 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB \
  --kernel-trace --output-format csv -d $OUT/b -o pmc -- $B > $OUT/b.out 2> $OUT/b.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r05_icache/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "snr::" in k:
            k = k.replace("void ", "").replace("snr::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k, "launches", len(acc[k]["_dur_us"]) // max(1, len(acc[k]) - 1))
    for n in sorted(c): print(f"   {n:34s} {c[n]:.4g}")
    if "SQC_ICACHE_REQ" in c and c["SQC_ICACHE_REQ"] > 0:
        print(f"   -> hit rate {c.get('SQC_ICACHE_HITS', 0) / c['SQC_ICACHE_REQ']:.4f}  misses/req {c.get('SQC_ICACHE_MISSES', 0) / c['SQC_ICACHE_REQ']:.4f}  duplicate misses/req {c.get('SQC_ICACHE_MISSES_DUPLICATE', 0) / c['SQC_ICACHE_REQ']:.4f}")
    if "SQ_IFETCH" in c and c["SQ_IFETCH"] > 0 and "SQ_IFETCH_LEVEL" in c:
        print(f"   -> mean fetch latency (LEVEL / IFETCH) {c['SQ_IFETCH_LEVEL'] / c['SQ_IFETCH']:.1f} (counter units)")
PY
