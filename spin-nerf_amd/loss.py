"""Density loss of the COLMAP-depth mode (`--sigma_loss`; DS_NeRF/loss.py:8-44, called from render_rays,
run_nerf.py:728-731): N_samples points between `near` and the ray's known depth, one MLP query, and

    loss_ray = -exp(sigma_last) / (sum_i exp(sigma_i) + 1),   sigma = relu(raw[..., 3] + noise)

i.e. a soft-max that wants the density to peak at the known depth.  The query runs through the fused HIP MLP
(`run_func` is render()'s network_query_fn); the [N_rays, N_samples] element-wise part around it is not worth a
kernel.  `randoms` (t_rand, noise — the latter pre-scaled) pins the draws for parity tests."""
import torch


class SigmaLoss:
    def __init__(self, N_samples, perturb, raw_noise_std):
        self.N_samples = N_samples
        self.perturb = perturb
        self.raw_noise_std = raw_noise_std

    def calculate_loss(self, rays_o, rays_d, viewdirs, near, far, depths, run_func, network, randoms=None):
        randoms = randoms or {}
        n_rays, dev = rays_o.shape[0], rays_o.device
        t = torch.linspace(0., 1., steps=self.N_samples, device=dev).expand(n_rays, self.N_samples)
        z = near * (1. - t) + depths[:, None] * t                       # loss.py:20-22 (`far` is not used)
        if self.perturb > 0.:
            mids = .5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            t_rand = randoms.get("t_rand")
            if t_rand is None:
                t_rand = torch.rand(z.shape, device=dev)
            z = lower + (upper - lower) * t_rand
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        raw = run_func(pts, viewdirs, network)
        sigma_raw = raw[..., 3]
        if self.raw_noise_std > 0.:
            noise = randoms.get("noise")
            if noise is None:
                noise = torch.randn(sigma_raw.shape, device=dev) * self.raw_noise_std
            sigma_raw = sigma_raw + noise
        sigma = torch.relu(sigma_raw)
        e = torch.exp(sigma)
        return -e[:, -1] / (e.sum(1) + 1)
