"""Round 5: compute-side cost of the data-parallel step's variants, two ranks on ONE GPU over gloo (the
collective itself cannot be timed here: gloo moves the buffers through the host; RCCL over xGMI will not).  What CAN be
measured: the GPU time of the step's kernels per variant (library events, rank 0) — the price of the launch structure each
variant implies — so that the first real 8-GPU run has a pre-stated expectation.
  SNR_MERGE_NETS=1 / 0        both networks' backward as one launch sequence / one per network
  SNR_SPLIT_ALLREDUCE=0 / 1   one all-reduce of both gradients / one per network
Run: python tests/probes/r05_dist_ab.py   (spawns 2 ranks per variant; output: profiles/r05_dist_ab.txt)"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r'''
import os, sys, importlib, time, torch
sys.path.insert(0, ROOT)
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
import spin_nerf_amd as S
L = S._lib
train = importlib.import_module("spin-nerf_amd.train")
torch.manual_seed(0)
nets = [S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda() for _ in range(2)]
def q(i, v, n): return S.run_network(i, v, n)
q._snr_fused = True
kw = dict(network_query_fn=q, perturb=1.0, N_importance=128, network_fine=nets[1], N_samples=64, network_fn=nets[0], use_viewdirs=True,
          white_bkgd=True, raw_noise_std=1.0, ndc=False, lindisp=True, near=1.2, far=9.0)
tr = train.RenderTrainer(kw, lrate=5e-4, world_size=world)
tr.broadcast_parameters()
H, W, focal = 378, 504, 400.0
g = torch.Generator().manual_seed(rank)
rays = torch.randn(2, 1024, 3, generator=g).cuda(); rays[0] = 0; rays[1, :, 2] = -1
target = torch.rand(1024, 3, generator=g).cuda()
for _ in range(5): tr.step(H, W, focal, rays, target)
torch.cuda.synchronize(); dist.barrier()
L.prof_enable(True)
t0 = time.perf_counter()
N = 20
for _ in range(N): tr.step(H, W, focal, rays, target)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
p = L.prof_read()
if rank == 0:
    tot = sum(v[0] for v in p.values()) / N
    print("RESULT kernels_ms_per_step %.4f wall_ms_per_step %.3f launches %d  %s" % (tot, wall * 1e3, sum(v[1] for v in p.values()) // N,
          {k: round(v[0] / N, 4) for k, v in p.items()}), flush=True)
dist.barrier(); dist.destroy_process_group()
'''

for merge, split in ((1, 0), (1, 1), (0, 0), (0, 1)):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2",
                   SNR_MERGE_NETS=str(merge), SNR_SPLIT_ALLREDUCE=str(split))
        procs.append(subprocess.Popen([sys.executable, "-c", "ROOT = %r\n" % ROOT + WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")]
    print(f"SNR_MERGE_NETS={merge} SNR_SPLIT_ALLREDUCE={split}: " + (line[0][7:] if line else "FAILED\n" + outs[0][-1500:] + outs[1][-1500:]), flush=True)
