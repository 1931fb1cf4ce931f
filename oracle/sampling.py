"""Oracle (test infrastructure): CPU restatement of the samplers / compositing helpers of
reference gpnerf/render_ray.py (lines 9-191)."""
from collections import OrderedDict

import torch


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False,
                            t_rand=None):
    """reference render_ray.py:58-102.  `t_rand` ([R,S] in [0,1)) replaces torch.rand_like when
    det=False so that stochastic sampling is testable."""
    near = depth_range[0, 0]
    far = depth_range[0, 1]
    assert near > 0 and far > 0 and far > near
    ones = torch.ones_like(ray_d[..., 0])
    near_d = (near * ones).to(ray_d.dtype)  # explicit cast: float64 depth ranges stay fp32
    far_d = (far * ones).to(ray_d.dtype)
    if inv_uniform:
        start = 1.0 / near_d
        step = (1.0 / far_d - start) / (N_samples - 1)
        z = 1.0 / torch.stack([start + i * step for i in range(N_samples)], dim=1)
    else:
        start = near_d
        step = (far_d - near_d) / (N_samples - 1)
        z = torch.stack([start + i * step for i in range(N_samples)], dim=1)
    if not det:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], dim=-1)
        lower = torch.cat([z[:, 0:1], mids], dim=-1)
        if t_rand is None:
            t_rand = torch.rand_like(z)
        z = lower + (upper - lower) * t_rand
    pts = z.unsqueeze(2) * ray_d.unsqueeze(1) + ray_o.unsqueeze(1)
    return pts, z


def sample_pdf(bins, weights, N_samples, det=False, u=None):
    """Inverse-CDF sampling, reference render_ray.py:9-55.  Does NOT mutate `weights` (the
    reference adds 1e-5 in place to its own clone).  `u` ([R,N]) replaces torch.rand."""
    M = weights.shape[1]
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[:, 0:1]), cdf], dim=-1)  # [R,M+1]
    if det:
        u = torch.linspace(0.0, 1.0, N_samples, device=bins.device)
        u = u.unsqueeze(0).repeat(bins.shape[0], 1)
    elif u is None:
        u = torch.rand(bins.shape[0], N_samples, device=bins.device)
    above = torch.zeros_like(u, dtype=torch.long)
    for i in range(M):
        above += (u >= cdf[:, i:i + 1]).long()
    below = torch.clamp(above - 1, min=0)
    idx = torch.stack((below, above), dim=2)
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(-1, N_samples, -1), -1, idx)
    bins_g = torch.gather(bins.unsqueeze(1).expand(-1, N_samples, -1), -1, idx)
    denom = cdf_g[:, :, 1] - cdf_g[:, :, 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[:, :, 0]) / denom
    return bins_g[:, :, 0] + t * (bins_g[:, :, 1] - bins_g[:, :, 0])


def sample_fine_pts(inv_uniform, N_importance, det, N_samples, ray_batch, weights, z_vals, u=None):
    """reference render_ray.py:162-191."""
    if inv_uniform:
        inv_z = 1.0 / z_vals
        inv_mid = 0.5 * (inv_z[:, 1:] + inv_z[:, :-1])
        w = weights[:, 1:-1]
        inv_s = sample_pdf(torch.flip(inv_mid, dims=[1]), torch.flip(w, dims=[1]),
                           N_importance, det=det, u=u)
        z_samples = 1.0 / inv_s
    else:
        mid = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
        z_samples = sample_pdf(mid, weights[:, 1:-1], N_importance, det=det, u=u)
    z = torch.cat((z_vals, z_samples), dim=-1)
    z, _ = torch.sort(z, dim=-1)
    pts = z.unsqueeze(2) * ray_batch["ray_d"].unsqueeze(1) + ray_batch["ray_o"].unsqueeze(1)
    return pts, z


def raw2outputs(raw, z_vals, mask, white_bkgd=False):
    """Classic volume compositing, reference render_ray.py:110-159 (model_type != 'gnt')."""
    rgb = raw[:, :, :3]
    sigma = raw[:, :, 3]
    alpha = 1.0 - torch.exp(-sigma)
    T = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)[:, :-1]
    T = torch.cat((torch.ones_like(T[:, 0:1]), T), dim=-1)
    weights = alpha * T
    rgb_map = torch.sum(weights.unsqueeze(2) * rgb, dim=1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - torch.sum(weights, dim=-1, keepdim=True))
    mask = mask.float().sum(dim=1) > 8
    depth_map = torch.sum(weights * z_vals, dim=-1)
    return OrderedDict([("rgb", rgb_map), ("depth", depth_map), ("weights", weights),
                        ("mask", mask), ("alpha", alpha), ("z_vals", z_vals)])
