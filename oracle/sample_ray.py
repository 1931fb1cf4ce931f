"""CPU restatement of the ray generation of reference gpnerf/sample_ray.py:108-132 (test
infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it).

Pinned by tests/golden/g6_sample_ray.npz, produced by the reference's own RaySamplerSingleImage
(tests/golden/make_golden.py)."""
import numpy as np
import torch


def get_rays_single_image(H, W, intrinsics, c2w, render_stride=1):
    """intrinsics, c2w [1,4,4] (or [1,3,3] / [1,3,4]) -> rays_o, rays_d [H'*W', 3] (reference :108-132)."""
    u, v = np.meshgrid(np.arange(W)[::render_stride], np.arange(H)[::render_stride])
    u = u.reshape(-1).astype(dtype=np.float32)
    v = v.reshape(-1).astype(dtype=np.float32)
    pixels = torch.from_numpy(np.stack((u, v, np.ones_like(u)), axis=0))
    batched_pixels = pixels.unsqueeze(0).repeat(intrinsics.shape[0], 1, 1)
    rays_d = (c2w[:, :3, :3].bmm(torch.inverse(intrinsics[:, :3, :3])).bmm(batched_pixels)).transpose(1, 2)
    rays_d = rays_d.reshape(-1, 3)
    rays_o = c2w[:, :3, 3].unsqueeze(1).repeat(1, rays_d.shape[0], 1).reshape(-1, 3)
    return rays_o, rays_d
