cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3d
run() {
  timeout 200 python bench.py --no-cpu-baseline --no-hashgrid --no-frame --steps 20 --warmup 5 > gpurun_out/r3d/t.txt 2>&1
  python - "$1" <<PY
import json,sys
for l in open('gpurun_out/r3d/t.txt'):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print(sys.argv[1], 'step', round(d['ms_per_step'],4), 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4))
PY
}
export SNR_PAIR_KIND=0 SNR_PAIR_PAIR=1
for sl in 128 64 32; do SNR_PAIR_SLOTS=$sl run "slots=$sl base"; done
export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_pa207.so
for sl in 128 64 32; do SNR_PAIR_SLOTS=$sl run "slots=$sl pa207"; done
