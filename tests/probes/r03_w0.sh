cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-frame --no-hashgrid --blocks 3"
export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_d.so
for rep in 1 2; do
for w in 66 72 76 80 86; do
  export SNR_PAIR_W0=$w
  timeout 300 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('W0 $w:', round(d['ms_per_step'],4), 'pair kernel', round(k['mlp_wgrad_pair']['ms_per_step'],4), 'reduce', round(k['mlp_wgrad_reduce']['ms_per_step'],4))
" | tee -a gpurun_out/w0_result.txt
done; done
