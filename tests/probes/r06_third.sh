#!/bin/bash
# third GPU call of round 6: the bf16 tests against the updated emulation, the hash-grid backward's ablations, the one-GPU
# rehearsal of tools/scale_matrix.py, the hardened bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
SNR_POISON_WS=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py tests/test_gpu_fullsize.py tests/test_gpu_train_step.py tests/test_gpu_nccl_one_rank.py tests/test_gpu_spin_iter.py tests/test_path.py -q -m gpu --tb=short -rA -p no:cacheprovider > $OUT/r06_third_tests.txt 2>&1
echo "tests exit=$? : $(tail -1 $OUT/r06_third_tests.txt)"; grep "^FAILED\|^ERROR" $OUT/r06_third_tests.txt | head -30
: > $OUT/r06_hg_ablate.txt
for v in base hg1 hg2 hg16 hg18 hg4 hg8; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  echo "$v $(timeout 300 python tests/probes/hashgrid_prof.py 2>/dev/null | tail -1)" >> $OUT/r06_hg_ablate.txt
done
unset SNR_LIB
cat $OUT/r06_hg_ablate.txt
timeout 1500 python3 tools/scale_matrix.py --gpus 1,2 --same-device --steps 10 --warmup 3 --out $OUT/r06_scale_matrix_rehearsal.md > $OUT/r06_scale_matrix.log 2>&1; echo "scale_matrix exit $?"; cat $OUT/r06_scale_matrix_rehearsal.md
timeout 900 python bench.py > $OUT/r06_bench_try.json 2> $OUT/r06_bench_try.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_bench_try.json") if l.startswith("{")][0])
print({k:d[k] for k in ("value","ms_per_step","ms_per_frame_378x504","sustained","frame_roofline","cpu_baseline")})
print(d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
