"""Training step of the render path: render() -> MSE losses -> backward -> (RCCL all-reduce) -> Adam.

Mirrors what one optimisation step of the reference's train() does around the hot path
(DS_NeRF/run_nerf.py:1455-1490 render + img2mse(rgb)+img2mse(rgb0); :1611-1612 backward + Adam;
:1616-1622 exponential lr decay).  The data feed, LaMa/LPIPS terms and logging of train() are out
of scope.

Data parallelism (new capability): rays shard by rank, every rank holds a full
replica of both MLPs (2 x 2.4 MB), and the only collective is one all-reduce of each net's flat fp32
gradient buffer per step over RCCL ("nccl" backend); Adam then runs on the flat buffers with the
1/world_size scale folded into the kernel.
"""
import ctypes
import math
import os

import torch

from . import _lib, ops
from ._lib import check, ptr
from .render import render


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


class RenderTrainer:
    def __init__(self, render_kwargs_train, lrate=5e-4, lrate_decay=250, world_size=1, process_group=None,
                 optimizer=None, start=None, graph=False):
        """``optimizer`` / ``start`` = what create_nerf() returned: when it reloaded a checkpoint (run_nerf.py:448-462)
        the Adam moments, the step count and the learning rate it restored continue here (a resumed run must not
        restart the bias correction and the lr decay); an optimizer whose state does not cover every network raises."""
        self.kw = dict(render_kwargs_train)
        self.nets = [n for n in (self.kw.get('network_fn'), self.kw.get('network_fine')) if n is not None]
        self.lrate, self.lrate_decay = lrate, lrate_decay
        self.world_size, self.pg = world_size, process_group
        # collectives run when there is more than one rank — or when SNR_FORCE_COLLECTIVES=1 asks a ONE-rank group to exercise
        # the collective code path (backend construction, async all-reduce on the gradient buffer, wait): tests/test_gpu_nccl_one_rank.py
        self._dist = world_size > 1 or os.environ.get("SNR_FORCE_COLLECTIVES") == "1"
        self.global_step = 0          # completed optimisation steps (the reference's global_step after its increment)
        self.opt_step = 0             # Adam's own step count (bias correction)
        self._lr = lrate              # rate the NEXT step uses (run_nerf.py:1616-1622 sets it after each step)
        self.m = [torch.zeros_like(n.flat.data) for n in self.nets]
        self.v = [torch.zeros_like(n.flat.data) for n in self.nets]
        # in-kernel random draws (Philox key / running call counter); ranks draw different streams
        self._seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (self._rank() + 1)) & 0xFFFFFFFFFFFFFFFF
        self._draws = 0
        # graph=True (or SNR_STEP_GRAPH=1): single-process steps of the plain configuration are captured once into a HIP
        # graph and replayed (the per-step scalars live in a device-side snr_step_state)
        self._graph_on = bool(graph) or os.environ.get("SNR_STEP_GRAPH") == "1"
        self._graph, self._graph_warm = None, None
        if optimizer is not None:
            self._adopt(optimizer, start)
        # The all-reduce of a net's gradient starts the moment autograd has finished that net (the fine net's
        # runs under the coarse net's backward); apply_gradients() only waits.
        self._works = {}
        if self._dist:
            for i, n in enumerate(self.nets):
                n.flat.register_post_accumulate_grad_hook(lambda p, i=i: self._start_all_reduce(i, p))

    def _rank(self):
        if self._dist:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_rank(self.pg)
        return 0

    def _start_all_reduce(self, i, p):
        import torch.distributed as dist
        self._works[i] = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _adopt(self, optimizer, start):
        """continue from a torch.optim.Adam over the networks' flat parameters (create_nerf's return value)"""
        st = optimizer.state
        have = [n.flat in st and 'exp_avg' in st[n.flat] for n in self.nets]
        if any(have) and not all(have):
            raise RuntimeError("the optimizer holds Adam state for some of the networks only; cannot resume")
        if all(have) and self.nets:
            steps = []
            for i, n in enumerate(self.nets):
                self.m[i].copy_(st[n.flat]['exp_avg']); self.v[i].copy_(st[n.flat]['exp_avg_sq'])
                steps.append(int(st[n.flat]['step']))
            if len(set(steps)) != 1:
                raise RuntimeError(f"networks disagree on the Adam step count: {steps}")
            self.opt_step = steps[0]
            self._lr = float(optimizer.param_groups[0]['lr'])
        if start is not None:
            self.global_step = int(start)

    def broadcast_parameters(self, src=0):
        """identical replicas at start (rank `src`'s init wins)"""
        if self._dist:
            import torch.distributed as dist
            for n in self.nets:
                dist.broadcast(n.flat.data, src=src, group=self.pg)
                n.mark_weights_changed()   # writes through .data do not bump the version the pack cache keys on

    def current_lr(self):
        """learning rate of the next optimisation step.  The reference steps with the optimizer's current rate and then
        sets lrate * 0.1 ** (global_step / (lrate_decay * 1000)) from the not-yet-incremented global_step
        (run_nerf.py:1611-1622, 1703): steps 1 and 2 run at lrate, step k >= 2 at lrate * 0.1 ** ((k - 2) / decay_steps)."""
        return self._lr

    def step(self, H, W, focal, batch_rays, target_s, chunk=1024 * 32, **extra):
        """one optimisation step on this rank's ray shard; returns (loss, rgb) detached.

        The plain configuration (NeRF or NeRF_TCNN networks, no sigma_loss / pytest hook, one chunk) runs without torch autograd:
        every launch of run_nerf.py:1465-1490 + 1611-1612 is issued directly, the random draws are made in-kernel
        (or taken from ``randoms=``), and each network's compositing forward, loss term and compositing backward are
        one kernel (snr_composite_train).  Anything else goes through render() + autograd."""
        if self._direct_ok(batch_rays, chunk, extra):
            if (self._graph_on and not self._dist and not extra.get("randoms")
                    and os.environ.get("SNR_NO_FUSED_STEP") != "1"):
                return self._step_graph(H, W, focal, batch_rays, target_s)
            return self._step_direct(H, W, focal, batch_rays, target_s, extra.get("randoms"))
        for n in self.nets:
            n.flat.grad = None
        rgb, disp, acc, depth, extras = render(H, W, focal, chunk=chunk, rays=batch_rays, retraw=True,
                                               **extra, **self.kw)
        # loss = img2mse(rgb, target) + img2mse(rgb0, target) (run_nerf.py:1482-1490) and its gradients in one
        # launch; autograd takes over at the render outputs
        rgb0 = extras.get('rgb0')
        loss, _, g_rgb, g_rgb0 = ops.mse_pair(rgb, rgb0, target_s)
        if rgb0 is not None:
            torch.autograd.backward([rgb, rgb0], [g_rgb, g_rgb0])
        else:
            torch.autograd.backward([rgb], [g_rgb])
        self.apply_gradients()
        return loss, rgb.detach()

    def _backward_two(self, h, net_c, net_f):
        """Both networks' parameter gradients of a fused forward, and the start of their exchange between data-parallel ranks.
        The gradients are halves of ONE buffer.  Three ways to exchange them (tools/scale_matrix.py measures all three):
          default                  both backward passes as one launch sequence (snr_mlp_backward_multi), then ONE all-reduce of
                                   the 4.77 MB — the cheapest launch structure, the whole collective exposed;
          SNR_SPLIT_ALLREDUCE=1    the same launch sequence, one all-reduce per network (started by apply_gradients);
          SNR_OVERLAP_ALLREDUCE=1  the fine network's backward first, its all-reduce started at once, the coarse network's
                                   backward UNDER it (two launch sequences: ~0.05 ms more kernels per step), then the coarse
                                   network's 2.4 MB — half the bytes exposed.
        Which wins at 8 ranks depends on the collective's latency over xGMI, unmeasured so far."""
        n_c = net_c.flat.numel()
        g_both = torch.empty(n_c + net_f.flat.numel(), device=net_c.flat.device, dtype=net_c.flat.dtype)
        g_c, g_f = g_both[:n_c], g_both[n_c:]
        if os.environ.get("SNR_OVERLAP_ALLREDUCE") == "1":
            # (the launch structure follows the switch with or without ranks, so that a one-rank run of it is the bit-exact
            #  reference of the collective run: tests/test_gpu_nccl_one_rank.py)
            if self._dist:
                import torch.distributed as dist
            ops.fused_backward(h, g_c, g_f, passes=ops.PASS_FINE)
            net_f.flat.grad = g_f
            if self._dist:
                self._works[self.nets.index(net_f)] = dist.all_reduce(g_f, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            ops.fused_backward(h, g_c, g_f, passes=ops.PASS_COARSE)
            net_c.flat.grad = g_c
            if self._dist:
                self._works[self.nets.index(net_c)] = dist.all_reduce(g_c, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            return
        ops.fused_backward(h, g_c, g_f)
        net_f.flat.grad = g_f
        net_c.flat.grad = g_c
        if self._dist and os.environ.get("SNR_SPLIT_ALLREDUCE") != "1":
            import torch.distributed as dist
            work = dist.all_reduce(g_both, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works[self.nets.index(net_c)] = self._works[self.nets.index(net_f)] = work

    # ---- the autograd-free step ------------------------------------------------------------------------------------
    def _direct_ok(self, batch_rays, chunk, extra):
        from .nerf import NeRF
        kw = self.kw
        if os.environ.get("SNR_NO_DIRECT_STEP") == "1" or set(extra) - {"randoms"}:
            return False
        # (no network_fine with N_importance > 0 = network_fn evaluated twice, run_nerf.py:705: the direct step handles it)
        nets = [kw.get('network_fn')] + ([kw['network_fine']] if kw.get('N_importance', 0) > 0 and kw.get('network_fine') is not None
                                        else [])
        from .hashgrid import NeRF_TCNN
        if any(type(n) not in (NeRF, NeRF_TCNN) for n in nets) or kw.get('sigma_loss') is not None:
            return False
        if not getattr(kw.get('network_query_fn'), "_snr_fused", False) or not batch_rays[1].is_cuda:
            return False
        if isinstance(kw.get('near'), torch.Tensor) or isinstance(kw.get('far'), torch.Tensor):
            return False
        return batch_rays.shape[1] <= chunk and bool(kw.get('use_viewdirs')) == bool(nets[0].use_viewdirs)

    def _step_direct(self, H, W, focal, batch_rays, target_s, randoms):
        kw = self.kw
        rnd = randoms or {}
        Nc, Nf = kw['N_samples'], kw.get('N_importance', 0)
        perturb, std = kw.get('perturb', 0.), float(kw.get('raw_noise_std', 0.))
        white, lindisp = kw.get('white_bkgd', False), kw.get('lindisp', False)
        net_c = kw['network_fn']
        net_f = (kw.get('network_fine') or net_c) if Nf > 0 else None
        target = ops.f32c(target_s)
        seed = self._seed

        if os.environ.get("SNR_NO_FUSED_STEP") != "1":
            # the whole launch sequence as library calls: snr_render_step_prepare (ray rows + stratified z + the zero fill of the
            # loss: one launch), snr_render_rays_fused_forward / _backward
            two = Nf > 0 and net_f is not net_c
            loss = torch.empty(2, device=batch_rays.device)      # [0] = the step's loss, [1] = the final render's term alone
            prep = dict(rays_o=batch_rays[0], rays_d=batch_rays[1], H=H, W=W, focal=focal, ndc=kw.get('ndc', True),
                        near=float(kw.get('near', 0.)), far=float(kw.get('far', 1.)), use_viewdirs=kw.get('use_viewdirs', False))
            h = ops.fused_forward(net_c, net_f if two else None, None, Nc, Nf, lindisp, white, perturb, std, seed, self._draws,
                                  target, loss, randoms=rnd, prepare=prep)   # local mean: Adam folds 1 / world_size in
            self._draws += 4
            if two:
                # both backward passes as ONE launch sequence (snr_mlp_backward_multi: one chain launch, one weight-gradient
                # launch, one reduce for the two networks); the two gradients are halves of one buffer, so that data-parallel
                # ranks exchange them with a single all-reduce
                self._backward_two(h, net_c, net_f)
            else:
                g_c = torch.empty_like(net_c.flat.data)
                ops.fused_backward(h, g_c)
                net_c.flat.grad = g_c
            self.apply_gradients()
            return loss[0], h.rgb

        rays = ops.pack_rays(batch_rays[0], batch_rays[1], H, W, focal, ndc=kw.get('ndc', True), near=float(kw.get('near', 0.)),
                             far=float(kw.get('far', 1.)), use_viewdirs=kw.get('use_viewdirs', False))
        vd = rays[:, -3:] if rays.shape[1] > 9 else None
        loss = torch.zeros(2, device=rays.device)      # [0] = the step's loss, [1] = the final render's term alone

        def draw():
            self._draws += 1
            return self._draws

        # coarse pass (run_nerf.py:646-692)
        if perturb > 0.:
            z_c = ops.sample_coarse(rays, Nc, lindisp, rnd["t_rand"]) if rnd.get("t_rand") is not None \
                else ops.sample_coarse_rng(rays, Nc, lindisp, seed, draw())
        else:
            z_c = ops.sample_coarse(rays, Nc, lindisp, None)
        raw_c, sv_c = ops.mlp_train_forward(net_c, rays, z_c, vd)
        n_c = ops.f32c(rnd["noise_c"]) if rnd.get("noise_c") is not None else None
        out_c = ops.composite_train(raw_c, z_c, rays, target, loss[0:1], None if Nf > 0 else loss[1:2], noise=n_c,
                                    noise_std=std, seed=seed, offset=draw(), white_bkgd=white)
        rgb = out_c[0]
        if Nf > 0:                                      # run_nerf.py:694-713
            if rnd.get("u") is not None:
                z_f = ops.sample_fine(z_c, out_c[4], Nf, rnd["u"])[0]
            elif perturb > 0.:
                z_f = ops.sample_fine_rng(z_c, out_c[4], Nf, seed, draw())
            else:
                z_f = ops.sample_fine(z_c, out_c[4], Nf, None)[0]
            raw_f, sv_f = ops.mlp_train_forward(net_f, rays, z_f, vd)
            n_f = ops.f32c(rnd["noise_f"]) if rnd.get("noise_f") is not None else None
            out_f = ops.composite_train(raw_f, z_f, rays, target, loss[0:1], loss[1:2], noise=n_f, noise_std=std, seed=seed,
                                        offset=draw(), white_bkgd=white)
            rgb = out_f[0]
            from .nerf import NeRF
            if net_f is not net_c and type(net_f) is NeRF and type(net_c) is NeRF:
                # both backward passes as one launch sequence, exactly like the fused library call (fused.cpp)
                g_f, g_c = ops.mlp_train_backward_multi([net_f, net_c], [sv_f, sv_c], [out_f[5], out_c[5]])
                net_f.flat.grad = g_f
            else:
                g_f = ops.mlp_train_backward(net_f, sv_f, out_f[5])
                if net_f is not net_c:
                    net_f.flat.grad = g_f
                    if self._dist:     # like the autograd hook: the fine net's all-reduce runs under the coarse backward
                        self._start_all_reduce(self.nets.index(net_f), net_f.flat)
                g_c = ops.mlp_train_backward(net_c, sv_c, out_c[5])
            net_c.flat.grad = g_c + g_f if net_f is net_c else g_c
        else:
            net_c.flat.grad = ops.mlp_train_backward(net_c, sv_c, out_c[5])
        self.apply_gradients()
        return loss[0], rgb

    # ---- the same step as a captured HIP graph ---------------------------------------------------------------------------
    def _host_state(self):
        """snr_step_state of the step about to run, from the host counters (include/spinnerf_hip.h)"""
        import numpy as np

        def powi(b, n):        # csrc/render_ops.hip: powi — the same IEEE multiplications as the library's
            r = 1.0
            while n > 0:
                if n & 1:
                    r *= b
                b *= b
                n >>= 1
            return r
        k = self.opt_step + 1
        b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))      # the C ABI takes the betas as floats
        return _lib.StepState(int(self._draws), k, int(self.global_step), float(self._lr),
                              float(np.float32(1.0 - powi(b1, k))), float(np.float32(math.sqrt(1.0 - powi(b2, k)))), 0.0)

    def _upload_state(self, G):
        st = self._host_state()
        G["state"].copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))
        G["expect"] = (self._draws, self.opt_step, self.global_step)

    def _capture(self, H, W, focal, batch_rays, target_s, key):
        kw = self.kw
        lib = _lib.load()
        Nc, Nf = kw['N_samples'], kw.get('N_importance', 0)
        net_c = kw['network_fn']
        net_f = (kw.get('network_fine') or net_c) if Nf > 0 else None
        two = Nf > 0 and net_f is not net_c
        dev = batch_rays.device
        G = {"key": key, "rays": batch_rays.detach().clone(), "target": ops.f32c(target_s).clone(),
             "state": torch.zeros(ctypes.sizeof(_lib.StepState), dtype=torch.uint8, device=dev)}
        self._upload_state(G)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = torch.zeros(2, device=dev)
            rows = ops.pack_rays(G["rays"][0], G["rays"][1], H, W, focal, ndc=kw.get('ndc', True), near=float(kw.get('near', 0.)),
                                 far=float(kw.get('far', 1.)), use_viewdirs=kw.get('use_viewdirs', False))
            for n in self.nets:              # the pack kernels belong to every replay: Adam rewrote the parameters
                n.mark_weights_changed()
                n.packed_weights()
            h = ops.fused_forward(net_c, net_f if two else None, rows, Nc, Nf, kw.get('lindisp', False), kw.get('white_bkgd', False),
                                  kw.get('perturb', 0.), float(kw.get('raw_noise_std', 0.)), self._seed, 0, G["target"], loss,
                                  offset_base=G["state"])
            g_c = torch.empty_like(net_c.flat.data)
            g_f = torch.empty_like(net_f.flat.data) if two else None
            ops.fused_backward(h, g_c, g_f)
            grads = {id(net_c): g_c}
            if two:
                grads[id(net_f)] = g_f
            for n, m, v in zip(self.nets, self.m, self.v):
                gr = grads.get(id(n))
                if gr is None:
                    continue
                check(lib.snr_adam_step_dev(ptr(n.flat.data), ptr(gr), ptr(m), ptr(v), n.flat.numel(), ptr(G["state"]), 0.9, 0.999,
                                            1e-8, 1.0, _lib.stream()), "snr_adam_step_dev")
            check(lib.snr_step_state_advance(ptr(G["state"]), float(self.lrate), float(self.lrate_decay) * 1000.0, 0.9, 0.999, 4,
                                             _lib.stream()), "snr_step_state_advance")
        G.update(graph=g, loss=loss, rgb=h.rgb, handle=h, grads=grads)
        self._graph = G

    def _step_graph(self, H, W, focal, batch_rays, target_s):
        """RenderTrainer.step through a captured graph.  The first call with a given batch shape runs eagerly (it sizes
        the allocator and sets the kernels' LDS attributes, which a capture may not), the second captures, every later one
        copies the batch into the graph's input buffers and replays.  Returned tensors are the graph's own output buffers:
        they are overwritten by the next step."""
        # (the graph holds raw pointers: a parameter or moment buffer that moved — .to(), a re-created network — means a
        #  new capture, not a replay into freed memory)
        # ... and so does everything the capture bakes in: the render configuration, the rate schedule, the networks' precision
        kw = self.kw
        baked = tuple((k, float(kw[k]) if isinstance(kw.get(k), (int, float)) and not isinstance(kw.get(k), bool) else kw.get(k))
                      for k in ('N_samples', 'N_importance', 'perturb', 'raw_noise_std', 'white_bkgd', 'lindisp', 'near', 'far', 'ndc',
                                'use_viewdirs') if not isinstance(kw.get(k), torch.Tensor))
        key = (H, W, float(focal), tuple(batch_rays.shape), str(batch_rays.device),
               tuple(n.flat.data_ptr() for n in self.nets), tuple(m.data_ptr() for m in self.m + self.v),
               baked, float(self.lrate), float(self.lrate_decay),
               tuple(n.cfg.key() if hasattr(getattr(n, "cfg", None), "key") else None for n in self.nets))
        if self._graph is None or self._graph["key"] != key:
            if self._graph_warm != key:
                self._graph_warm = key
                return self._step_direct(H, W, focal, batch_rays, target_s, None)
            try:
                self._capture(H, W, focal, batch_rays, target_s, key)
            except _lib.HipLibraryError:
                raise                   # a library / launch error is a bug, not a capture limitation: never mask it
            except RuntimeError as e:   # what torch.cuda.graph raises when something in the step cannot be captured
                # a failed capture must not be retried on every step: say so, forget the warm-up state, take the eager route
                import warnings
                warnings.warn(f"RenderTrainer: HIP-graph capture of the step failed ({e}); continuing on the eager route")
                self._graph, self._graph_warm, self._graph_on = None, None, False
                return self._step_direct(H, W, focal, batch_rays, target_s, None)
        G = self._graph
        if G["expect"] != (self._draws, self.opt_step, self.global_step):   # eager steps / a checkpoint moved the counters
            self._upload_state(G)
        G["rays"].copy_(batch_rays)
        G["target"].copy_(target_s)
        G["graph"].replay()
        for n in self.nets:
            n.flat.grad = G["grads"].get(id(n))
            n.mark_weights_changed()
        self._draws += 4
        self.opt_step += 1
        self._lr = self.lrate * (0.1 ** (self.global_step / (self.lrate_decay * 1000)))
        self.global_step += 1
        G["expect"] = (self._draws, self.opt_step, self.global_step)
        return G["loss"][0], G["rgb"]

    def spin_loss(self, H, W, focal, batch_rays_clf, target_clf, batch_rays, target_s, batch_inp=None,
                  depth_inp=None, chunk=1024 * 32, randoms=None, batched=False, colmap_depth=None, lpips=None, **extra):
        """Loss of one SPIn-NeRF iteration in its default mode (run_nerf.py:1455-1521, no --masked_NeRF /
        --object_removal / --prepare / --no_geometry, LPIPS and COLMAP-depth terms excluded — SURVEY.md §8 f-1):

          render(unmasked-pixel rays)                         -> mse(rgb, target_clf) + mse(rgb0, target_clf)
          render(all-pixel rays, detach_weights=True)         -> mse(rgb, target_s)   + mse(rgb0, target_s)
          render(rays with an inpainted-disparity target)     -> MSE(disp, depth_inp) + MSE(disp0, depth_inp),
                                                                 skipped when NaN like the reference (:1518-1521)

        Optional terms of the same iteration:
          colmap_depth = dict(rays=[2,N,3], target=[N] depths, weights=[N] or None, depth_lambda=..., mode='mse' |
                         'weighted' | 'weighted_normalized' | 'relative', max_depth=...)
              -> render(rays, depths=target) and depth_lambda * the depth loss of run_nerf.py:1473-1507 (--colmap_depth
                 --depth_loss with --weighted_loss / --normalize_depth / --relative_loss); its randoms = randoms[3]
          lpips        = dict(fn=<perceptual distance on [-1,1] NCHW batches>, poses=[B,3,4], images=[B,H,W,3],
                         masks=[B,H,W], hwf=(H,W,focal), render_kwargs=<test kwargs>, render_factor=1, patch_len_factor=4)
              -> the patch renders with gradients of run_nerf.py:1523-1561 (render_path in patch mode, detach_weights=True)
                 and sum(fn(prediction, target patch)) / B / 100.  The LPIPS network itself is not part of this package
                 (its VGG weights are a download); any callable with that contract plugs in.

        ``randoms`` = optional list of injected-random dicts (one per render, as for render()) so that
        parity tests can pin the draws.  ``batched`` runs the first and the third render as one.
        Returns (loss, dict of the render outputs)."""
        rnd = list(randoms or [None, None, None])
        while len(rnd) < 4:
            rnd.append(None)
        kw = dict(chunk=chunk, retraw=True, **extra, **self.kw)
        ex_i = disp_i = None
        if batched and batch_inp is not None:
            # the two renders that do not detach the weights share one launch sequence (rays are independent,
            # chunking does not change results): 2 x N_rand rays through render(), outputs split afterwards
            n = batch_rays_clf.shape[1]
            both = torch.cat([batch_rays_clf, batch_inp], 1)
            r2 = None
            if rnd[0] is not None and rnd[2] is not None:
                r2 = {k: torch.cat([rnd[0][k], rnd[2][k]], 0) for k in rnd[0]}
            rgb2, disp2, acc2, depth2, ex2 = render(H, W, focal, rays=both, randoms=r2, **kw)
            rgb, disp, acc, depth = rgb2[:n], disp2[:n], acc2[:n], depth2[:n]
            ex = {k: v[:n] for k, v in ex2.items()}
            disp_i, ex_i = disp2[n:], {k: v[n:] for k, v in ex2.items()}
        else:
            rgb, disp, acc, depth, ex = render(H, W, focal, rays=batch_rays_clf, randoms=rnd[0], **kw)
        rgb_c, _, _, _, ex_c = render(H, W, focal, rays=batch_rays, detach_weights=True, randoms=rnd[1], **kw)
        loss = img2mse(rgb, target_clf) + img2mse(rgb_c, target_s)
        if 'rgb0' in ex_c:
            loss = loss + img2mse(ex_c['rgb0'], target_s)
        if 'rgb0' in ex:
            loss = loss + img2mse(ex['rgb0'], target_clf)
        outs = {"clf": (rgb, disp, acc, depth, ex), "complete": (rgb_c, ex_c)}
        if batch_inp is not None:
            if disp_i is None:
                _, disp_i, _, _, ex_i = render(H, W, focal, rays=batch_inp, randoms=rnd[2], **kw)
            inp = img2mse(disp_i, depth_inp)
            if 'disp0' in ex_i:
                inp = inp + img2mse(ex_i['disp0'], depth_inp)
            if not bool(torch.isnan(inp)):
                loss = loss + inp
            outs["inp"] = (disp_i, ex_i)
        if colmap_depth is not None:                               # run_nerf.py:1473-1477, 1492-1507
            cd = colmap_depth
            tgt = cd["target"]
            _, _, _, depth_col, ex_col = render(H, W, focal, rays=cd["rays"], depths=tgt, randoms=rnd[3], **kw)
            mode = cd.get("mode", "mse")
            if mode == "weighted":
                d_loss = torch.mean(((depth_col - tgt) ** 2) * cd["weights"])
            elif mode == "weighted_normalized":
                d_loss = torch.mean((((depth_col - tgt) / cd["max_depth"]) ** 2) * cd["weights"])
            elif mode == "relative":
                d_loss = torch.mean(((depth_col - tgt) / tgt) ** 2)
            else:
                d_loss = img2mse(depth_col, tgt)
            loss = loss + cd.get("depth_lambda", 0.1) * d_loss
            outs["colmap"] = (depth_col, ex_col)
        if lpips is not None:                                      # run_nerf.py:1523-1561
            loss = loss + self.lpips_term(chunk=chunk, **lpips)
        return loss, outs

    def lpips_term(self, fn, poses, images, masks, hwf, render_kwargs, render_factor=1, patch_len_factor=4,
                   chunk=1024 * 32):
        """Patch renders WITH gradients of `poses` (render_path's patch mode: corner drawn inside each mask's bounding
        box, detach_weights=True) against the same patch of the (resized) training images, mapped to [-1, 1] NCHW like
        the reference feeds its LPIPS network; returns sum(fn(pred, target).mean()) / B / 100 (run_nerf.py:1523-1561)."""
        from .path import render_path
        import torch.nn.functional as F
        Hh, Ww = int(hwf[0]) // render_factor, int(hwf[1]) // render_factor
        patch_len = (Hh // patch_len_factor, Ww // patch_len_factor)
        rgbs, _, (Xs, Ys) = render_path(poses, hwf, chunk, render_kwargs, render_factor=render_factor,
                                        rgb_require_grad=True, need_alpha=False, detach_weights=True,
                                        patch_len=patch_len, masks=masks)
        total = 0.
        for b in range(len(poses)):
            pred = ((rgbs[b] - 0.5) * 2).permute(2, 0, 1)[None, ...]
            tgt = ((images[b] - 0.5) * 2).permute(2, 0, 1)[None, ...]
            if tuple(tgt.shape[-2:]) != (Hh, Ww):     # torchvision.transforms.Resize (bilinear, antialiased) in the reference
                tgt = F.interpolate(tgt, size=(Hh, Ww), mode="bilinear", antialias=True, align_corners=False)
            tgt = tgt[:, :, Xs[b]:Xs[b] + patch_len[0], Ys[b]:Ys[b] + patch_len[1]]
            total = total + fn(pred, tgt.to(pred.device)).mean()
        return total / len(poses) / 100

    def export_disparities(self, poses, hwf, render_kwargs, masks, out_dir, render_factor=1, chunk=1024 * 32):
        """--prepare (run_nerf.py:1563-1609): render every training pose without gradients and write the disparity maps
        (x 255, 8-bit PNG) and the masks subsampled by render_factor (x 255) as the depth-inpainting input:
        <out_dir>/img{i:03}.png and <out_dir>/label/img{i:03}.png."""
        import os
        import numpy as np
        from .path import render_path, write_png
        os.makedirs(os.path.join(out_dir, "label"), exist_ok=True)
        with torch.no_grad():
            _, disps, _ = render_path(poses, hwf, chunk, render_kwargs, render_factor=render_factor)
        for i in range(len(poses)):
            d = np.nan_to_num(np.asarray(disps[i], dtype=np.float64) * 255, nan=0.0)   # cv2.imwrite casts with saturation
            write_png(os.path.join(out_dir, f"img{i:03d}.png"), np.clip(np.rint(d), 0, 255).astype(np.uint8))
            m = np.asarray(masks[i])[::render_factor, ::render_factor] * 255
            write_png(os.path.join(out_dir, "label", f"img{i:03d}.png"), np.clip(np.rint(m), 0, 255).astype(np.uint8))
        return disps

    def _spin_direct(self, H, W, focal, batch_rays_clf, target_clf, batch_rays, target_s, batch_inp, depth_inp, randoms=None):
        """The three renders of the iteration as ONE render of the concatenated rays (rays are independent), on the step's
        library route: snr_render_step_prepare, snr_render_rays_fused_forward_terms — the loss as three terms over the three ray
        ranges (unmasked pixels: rgb; all pixels: rgb through detached weights; inpainted-depth rays: disparity, NaN-guarded
        like run_nerf.py:1518-1521) —, ONE backward launch sequence over all rays for both networks, snr_adam_pack_multi.  No
        torch autograd; the sum of the three terms' gradients is what one backward over the concatenated batch computes."""
        kw = self.kw
        Nc, Nf = kw['N_samples'], kw.get('N_importance', 0)
        net_c = kw['network_fn']
        net_f = (kw.get('network_fine') or net_c) if Nf > 0 else None
        two = Nf > 0 and net_f is not net_c
        n1, n2, n3 = batch_rays_clf.shape[1], batch_rays.shape[1], batch_inp.shape[1]
        rays_all = torch.cat([batch_rays_clf, batch_rays, batch_inp], 1)
        rnd = None
        if randoms and all(r is not None for r in randoms[:3]):
            rnd = {k: torch.cat([ops.f32c(r[k]) for r in randoms[:3]], 0) for k in randoms[0]}
        loss = torch.empty(4, device=rays_all.device)   # [0] the iteration's loss, [1] mse(rgb, target_clf) alone, [2] the geometry term
        prep = dict(rays_o=rays_all[0], rays_d=rays_all[1], H=H, W=W, focal=focal, ndc=kw.get('ndc', True),
                    near=float(kw.get('near', 0.)), far=float(kw.get('far', 1.)), use_viewdirs=kw.get('use_viewdirs', False))
        terms = [dict(first=0, n=n1, kind=_lib.LOSS_RGB, target=target_clf, slot=0, slot_final=1),
                 dict(first=n1, n=n2, kind=_lib.LOSS_RGB_DETACHED, target=target_s, slot=0),
                 dict(first=n1 + n2, n=n3, kind=_lib.LOSS_DISP, target=depth_inp, slot=2)]
        h = ops.fused_forward(net_c, net_f if two else None, None, Nc, Nf, kw.get('lindisp', False), kw.get('white_bkgd', False),
                              kw.get('perturb', 0.), float(kw.get('raw_noise_std', 0.)), self._seed, self._draws, None, loss,
                              randoms=rnd, prepare=prep, loss_terms=terms, guard_term=2)
        self._draws += 4
        if two:
            self._backward_two(h, net_c, net_f)
        else:
            g_c = torch.empty_like(net_c.flat.data)
            ops.fused_backward(h, g_c)
            net_c.flat.grad = g_c
        self.apply_gradients()
        self._last_spin = h
        return loss[0], -10.0 * torch.log10(loss[1])

    def _spin_direct_ok(self, args, kwargs):
        if os.environ.get("SNR_NO_FUSED_STEP") == "1" or os.environ.get("SNR_NO_DIRECT_SPIN") == "1":
            return False
        names = ("H", "W", "focal", "batch_rays_clf", "target_clf", "batch_rays", "target_s", "batch_inp", "depth_inp")
        a = dict(zip(names, args))
        a.update(kwargs)
        if set(a) - set(names) - {"randoms", "chunk", "batched"} or a.get("batch_inp") is None or a.get("depth_inp") is None:
            return False      # COLMAP-depth render, LPIPS term, no geometry term: the general route
        n = a["batch_rays_clf"].shape[1] + a["batch_rays"].shape[1] + a["batch_inp"].shape[1]
        if not self._direct_ok(a["batch_rays_clf"], n, {}) or n > a.get("chunk", 1024 * 32) * 3:
            return False
        # the library route hands raw device pointers to the kernels: every tensor must live on the rays' device and hold exactly
        # its ray range's elements (anything else takes the general route, whose torch ops raise the usual device / shape errors)
        dev = a["batch_rays_clf"].device
        for rays, tgt, per_ray in ((a["batch_rays_clf"], a["target_clf"], 3), (a["batch_rays"], a["target_s"], 3),
                                   (a["batch_inp"], a["depth_inp"], 1)):
            if not (isinstance(tgt, torch.Tensor) and rays.is_cuda and rays.device == dev and tgt.device == dev
                    and tgt.numel() == per_ray * rays.shape[1]):
                return False
        rnd = a.get("randoms")
        if rnd:      # injected draws: for all three renders or for none (a partial list cannot be honoured by ONE concatenated render)
            have = [r is not None for r in list(rnd)[:3]] + [False] * (3 - len(list(rnd)[:3]))
            if any(have) and not all(have):
                return False
        from .nerf import NeRF
        kw = self.kw
        nets = [kw.get('network_fn')] + ([kw['network_fine']] if kw.get('N_importance', 0) > 0 and kw.get('network_fine') is not None else [])
        return all(type(x) is NeRF for x in nets)

    def spin_iteration(self, *args, **kwargs):
        """spin_loss + backward + all-reduce + Adam; returns (loss, psnr of the unmasked-pixel render).  The default
        iteration (three renders, MLP networks, no COLMAP-depth / LPIPS term) takes the step's library route without torch
        autograd (_spin_direct); everything else, or SNR_NO_DIRECT_SPIN=1, goes through spin_loss + autograd."""
        if self._spin_direct_ok(args, kwargs):
            names = ("H", "W", "focal", "batch_rays_clf", "target_clf", "batch_rays", "target_s", "batch_inp", "depth_inp")
            a = dict(zip(names, args))
            a.update({k: v for k, v in kwargs.items() if k in names or k == "randoms"})
            return self._spin_direct(**a)
        for n in self.nets:
            n.flat.grad = None
        loss, outs = self.spin_loss(*args, **kwargs)
        loss.backward()
        self.apply_gradients()
        rgb = outs["clf"][0].detach()
        target_clf = args[4] if len(args) > 4 else kwargs["target_clf"]
        mse = img2mse(rgb, target_clf)
        return loss.detach(), -10.0 * torch.log10(mse)

    # ---- checkpoints in the reference's format (run_nerf.py:1626-1636 written, :448-462 read) ----------
    def state_dict(self):
        """{'global_step', 'network_fn_state_dict', 'network_fine_state_dict', 'optimizer_state_dict'} with the
        optimizer state laid out as torch.optim.Adam over the per-layer parameters in `grad_vars` order
        (coarse net's parameters, then the fine net's: run_nerf.py:398-425) — loadable by the reference."""
        fn, fine = self.kw.get('network_fn'), self.kw.get('network_fine')
        state, idx = {}, 0
        for n, m, v in zip(self.nets, self.m, self.v):
            # param_views: the parameters the reference module registers (NeRF_RGB has no alpha_linear, helpers:183-185)
            for (k, mv), vv in zip(n.param_views(m).items(), n.param_views(v).values()):
                state[idx] = {'step': torch.tensor(float(self.opt_step)), 'exp_avg': mv.detach().clone(),
                              'exp_avg_sq': vv.detach().clone()}
                idx += 1
        group = {'lr': self.current_lr(), 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(idx))}
        return {'global_step': self.global_step,
                'network_fn_state_dict': fn.state_dict() if fn is not None else None,
                'network_fine_state_dict': fine.state_dict() if fine is not None else None,
                'optimizer_state_dict': {'state': state if self.opt_step > 0 else {}, 'param_groups': [group]}}

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)

    def load_state_dict(self, ckpt):
        """Accepts checkpoints written by save_checkpoint() or by the reference's train()."""
        fn, fine = self.kw.get('network_fn'), self.kw.get('network_fine')
        if fn is not None:
            fn.load_state_dict(ckpt['network_fn_state_dict'])
        if fine is not None and ckpt.get('network_fine_state_dict') is not None:
            fine.load_state_dict(ckpt['network_fine_state_dict'])
        self.global_step = int(ckpt['global_step'])
        osd = ckpt['optimizer_state_dict']
        st = osd['state']
        from .render import scatter_per_layer_adam
        for m, v in zip(self.m, self.v):
            m.zero_(); v.zero_()
        steps = set(scatter_per_layer_adam(st, self.nets, list(zip(self.m, self.v)))) if st else set()
        steps.discard(0.0) if len(steps) > 1 else None      # (a network all of whose layers are idle carries no step count)
        if len(steps) > 1:
            raise RuntimeError(f"checkpoint parameters disagree on the Adam step count: {sorted(steps)}")
        self.opt_step = int(steps.pop()) if steps else 0
        self._lr = float(osd['param_groups'][0]['lr']) if osd.get('param_groups') else self.lrate
        for n in self.nets:
            n.mark_weights_changed()

    def load_checkpoint(self, path, map_location=None):
        self.load_state_dict(torch.load(path, map_location=map_location, weights_only=False))

    # ---- exposed communication time (bench.py's `distributed.allreduce_ms_per_step_exposed`) ----
    def comm_reset(self):
        self._comm_events, self._comm_steps = [], 0

    def comm_ms_per_step(self):
        """Stream time the optimizer step spent waiting for the gradient all-reduce (events around the waits), per step
        since comm_reset(); None on one GPU."""
        if not self._dist or not getattr(self, "_comm_steps", 0):
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._comm_events) / self._comm_steps

    def apply_gradients(self):
        if self._dist:
            import torch.distributed as dist
            idle = []                            # networks without a gradient: every rank agrees on which (same code path)
            for i, n in enumerate(self.nets):   # anything the hooks did not see (gradients set by hand)
                if i not in self._works:
                    if n.flat.grad is None:     # a network that took no part in this step still joins the collective ...
                        n.flat.grad = torch.zeros_like(n.flat.data)
                        idle.append(n)
                    self._start_all_reduce(i, n.flat)
            timed = hasattr(self, "_comm_events") and self.nets[0].flat.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for w in self._works.values():
                w.wait()
            if timed:
                e1.record()
                self._comm_events.append((e0, e1))
                self._comm_steps += 1
            self._works.clear()
            for n in idle:                       # ... but takes no Adam step, exactly like the single-process path
                n.flat.grad = None
        self.opt_step += 1
        lr = self._lr
        from .nerf import NeRF
        live = [(n, m, v) for n, m, v in zip(self.nets, self.m, self.v) if n.flat.grad is not None]   # torch.optim.Adam skips parameters without a gradient
        fused = [t for t in live if type(t[0]) is NeRF and t[0].flat.is_cuda and os.environ.get("SNR_NO_ADAM_PACK") != "1"]
        done = []
        if fused:
            # Adam + the re-pack of the weights of both MLPs: ONE launch (csrc/adam_pack.hip) instead of two Adam and two pack launches.
            # Two networks share a call only when their configurations are equal (then the call is one launch, and
            # SNR_ERR_UNSUPPORTED — a shape the fused kernel does not cover — means nothing ran: the generic route below takes over)
            k = 0
            while k < len(fused):
                pair = k + 1 < len(fused) and fused[k][0].cfg.key() == fused[k + 1][0].cfg.key()
                grp = fused[k:k + (2 if pair else 1)]
                k += len(grp)
                if ops.adam_pack_step_([t[0] for t in grp], [t[0].flat.grad for t in grp], [t[1] for t in grp], [t[2] for t in grp],
                                       lr, self.opt_step, grad_scale=1.0 / self.world_size):
                    done.extend(t[0] for t in grp)
        for n, m, v in live:
            if any(n is d for d in done):
                continue
            ops.adam_step_(n.flat.data, n.flat.grad, m, v, lr, self.opt_step, grad_scale=1.0 / self.world_size)
            n.mark_weights_changed()   # written through a raw pointer: re-pack before the next forward
        # run_nerf.py:1616-1622 (rate for the next step, from the global_step before its increment), :1703
        self._lr = self.lrate * (0.1 ** (self.global_step / (self.lrate_decay * 1000)))
        self.global_step += 1
