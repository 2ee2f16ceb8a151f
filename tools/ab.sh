#!/bin/bash
# A/B of library variants inside ONE gpurun call (box-to-box variance is +-2-3 %):  tools/ab.sh OUTDIR name1 name2 ...
# "base" = the in-tree library; other names = spin-nerf_amd/lib/ablate/libspinnerf_hip_<name>.so (tools/build_variant.py)
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
    python bench.py --no-cpu-baseline --no-hashgrid ${AB_ARGS:-} > $OUT/${v}_$rep.log 2>&1
  done
done
python - "$OUT" "$@" <<'PY'
import json, sys, glob
out = sys.argv[1]
for v in sys.argv[2:]:
    for f in sorted(glob.glob(f"{out}/{v}_*.log")):
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); k = d["kernels"]
                g = lambda n: k.get(n, {}).get("ms_per_step", 0.0)
                print(f"{v:10s} step {d['ms_per_step']:.4f} ms  frame {d.get('ms_per_frame_378x504') or 0:.2f} ms  fwd {g('mlp_fwd'):.4f} dgrad {g('mlp_dgrad'):.4f} "
                      f"wgrad_pair {g('mlp_wgrad_pair'):.4f} wgrad {g('mlp_wgrad'):.4f} reduce {g('mlp_wgrad_reduce'):.4f} pack {g('mlp_pack'):.4f} adam {g('adam'):.4f}")
PY
