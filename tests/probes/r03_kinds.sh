cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frame --no-hashgrid --blocks 1"
export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_d.so
for cfg in "x x" "0 x" "1 x" "0 1" "1 1" "0 0" "1 0"; do
  set -- $cfg
  unset SNR_PAIR_KIND SNR_PAIR_PAIR
  [ $1 != x ] && export SNR_PAIR_KIND=$1
  [ $2 != x ] && export SNR_PAIR_PAIR=$2
  timeout 300 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('kind $1 pair $2:', 'pair kernel', round(k['mlp_wgrad_pair']['ms_per_step'],4))
" | tee -a gpurun_out/kinds_result.txt
done
