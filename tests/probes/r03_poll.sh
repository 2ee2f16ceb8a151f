cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-frame --no-hashgrid --blocks 3"
for rep in 1 2 3; do
for cfg in "16 2" "8 2" "4 2" "8 1" "8 4" "0 2"; do
  set -- $cfg
  export SNR_PAIR_POLL=$1 SNR_PAIR_LEAD=$2
  timeout 300 $B 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('poll $1 lead $2: step', round(d['ms_per_step'],4), 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4), 'fwd', round(k['mlp_fwd']['ms_per_step'],4))
" | tee -a gpurun_out/poll_result.txt
done; done
