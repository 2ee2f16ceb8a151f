#!/bin/bash
# MFMA-busy and effective clock of the inference kernel over full 378 x 504 frames (tests/probes/r05_frame_split.py)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_frame_pmc; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/a -o pmc -- python3 tests/probes/r05_frame_split.py --frames 2 > $OUT/a.out 2> $OUT/a.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r05_frame_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlp_fwd_kernel" in k:
            d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            acc[(k.split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            acc[(k.split("(")[0], r["Dispatch_Id"])]["_dur"] = d
rows = [v for v in acc.values() if v["_dur"] > 5e6]     # the fine-network launches of a whole frame
for v in rows[-2:]:
    dur = v["_dur"]; clk = v["GRBM_GUI_ACTIVE"] / 8 / dur
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (dur * clk)
    w = v["SQ_WAVE_CYCLES"]
    print(f"launch {dur/1e6:.2f} ms  effective clock {clk:.3f} GHz  MFMA busy {busy*100:.1f} % of the elapsed cycles  "
          f"MFMAs {v['SQ_INSTS_MFMA']:.3e}  wave time: parked {v['SQ_WAIT_ANY']/w:.2f} issue-stalled {v['SQ_WAIT_INST_ANY']/w:.2f} issuing {v['SQ_ACTIVE_INST_ANY']/w:.2f}")
PY
