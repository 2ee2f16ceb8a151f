#!/bin/bash
# SQ counters of the shipped chain kernels and the chain2 kernels, same process (tests/probes/r05_chain2_check.py --quick [--train])
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_c2_pmc; rm -rf $OUT; mkdir -p $OUT
ARGS="--quick $*"
timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC \
  --kernel-trace --output-format csv -d $OUT/a -o pmc -- python3 tests/probes/r05_chain2_check.py $ARGS > $OUT/a.out 2> $OUT/a.err
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/b -o pmc -- python3 tests/probes/r05_chain2_check.py $ARGS > $OUT/b.out 2> $OUT/b.err
timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/c -o pmc -- python3 tests/probes/r05_chain2_check.py $ARGS > $OUT/c.out 2> $OUT/c.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r05_c2_pmc/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlp_fwd" in k or "mlp_dgrad" in k:
            k = k.replace("void ", "").replace("snr::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, "launches", len(next(iter(acc[k].values()))))
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    for n in sorted(c): print(f"   {n:28s} {c[n]:.4g}")
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        print(f"   -> of wave cycles: parked {c.get('SQ_WAIT_ANY',0)/w:.3f}  issue-stalled {c.get('SQ_WAIT_INST_ANY',0)/w:.3f}  issuing {c.get('SQ_ACTIVE_INST_ANY',0)/w:.3f}")
PY
tail -4 $OUT/a.out
