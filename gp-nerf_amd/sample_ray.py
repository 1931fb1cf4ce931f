"""`RaySamplerSingleImage` -- same constructor, attributes and return dictionaries as reference
gpnerf/sample_ray.py:38-214 (the step right before the hot path).

The reference builds all rays of the image in numpy on the host and copies 24 bytes per ray to the
GPU for every image (get_all :134-147).  Here the rays are GENERATED on the device by
`gpn_rays_single_image` (csrc/sampling.hip) from the 3x3 matrix c2w[:3,:3] . K[:3,:3]^-1 and the
camera centre, bit for bit the reference's values; nothing but 12 floats crosses PCIe.  The
per-image tensors that only ride along (rgb, labels, depth, source views) are moved with `.to()`.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib

rng = np.random.RandomState(234)  # reference sample_ray.py:6 (module-level, same seed)


def _split_camera(params, k_side, pose_rows):
    """Rows of [H, W, K (k_side x k_side, row-major), pose (pose_rows x 4, row-major)] ->
    (W, H, K [B,k,k], pose [B,rows,4]) -- the tuple order of the reference's parsers."""
    nk = k_side * k_side
    K = params[:, 2:2 + nk].reshape((-1, k_side, k_side))
    pose = params[:, 2 + nk:2 + nk + 4 * pose_rows].reshape((-1, pose_rows, 4))
    return params[:, 1], params[:, 0], K, pose


def parse_camera(params):
    """23-float cameras: 3x3 intrinsics, 3x4 pose (reference sample_ray.py:13-18)."""
    return _split_camera(params, 3, 3)


def parse_camera_ray(params):
    """34-float cameras: 4x4 intrinsics, 4x4 pose (reference sample_ray.py:20-25)."""
    return _split_camera(params, 4, 4)


def _dev(t, device):
    return t.to(device) if t is not None else None


class RaySamplerSingleImage(object):
    # per-pixel tensors of the query image: attribute name -> channels of a flattened row
    _PIXEL_FIELDS = (("rgb", 3), ("labels", 1), ("true_depth", 1), ("depth_mask", 1))

    def __init__(self, data, device, resize_factor=1, render_stride=1):
        """`data`: one item of a reference dataset (attributes as reference sample_ray.py:39-93)."""
        super().__init__()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RaySamplerSingleImage generates rays on a HIP device (got %s); there "
                               "is no CPU fallback" % (self.device,))
        self.render_stride = render_stride
        # optional entries default to None; the mandatory ones raise KeyError as in the reference
        for name in ("rgb", "labels", "src_rgbs", "src_cameras", "src_labels"):
            setattr(self, name, data.get(name))
        for name in ("camera", "rgb_path", "true_depth", "depth_mask", "depth_range"):
            setattr(self, name, data[name])
        parse = parse_camera_ray if self.camera.shape[-1] == 34 else parse_camera
        widths, heights, self.intrinsics, self.c2w_mat = parse(self.camera)
        self.batch_size = len(self.camera)
        if self.batch_size != 1:
            raise ValueError("one query image per sampler (the render path asserts batch size 1, "
                             "reference projection.py:97-101)")
        self.H, self.W = int(heights[0]), int(widths[0])
        if resize_factor != 1:  # resized output (reference :58-80): image size, intrinsics and pixel tensors
            self.H, self.W = int(self.H * resize_factor), int(self.W * resize_factor)
            self.intrinsics = self.intrinsics.clone()
            self.intrinsics[:, :2, :3] *= resize_factor
            for name, _ in self._PIXEL_FIELDS:
                t = getattr(self, name)
                if t is not None:
                    setattr(self, name, F.interpolate(t.permute(0, 3, 1, 2),
                                                      scale_factor=resize_factor).permute(0, 2, 3, 1))
        self.rays_o, self.rays_d = self.get_rays_single_image(self.H, self.W, self.intrinsics,
                                                              self.c2w_mat)
        for name, ch in self._PIXEL_FIELDS:  # one row per pixel, in ray order
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.reshape(-1, ch))

    # -------------------------------------------------------------------------------------------
    def _matrix(self, intrinsics, c2w):
        """M = c2w[:3,:3] . K[:3,:3]^-1 and o = c2w[:3,3] with the reference's own CPU ops (:124-130)."""
        intr = intrinsics.detach().float().cpu()
        c2w = c2w.detach().float().cpu()
        M = c2w[:, :3, :3].bmm(torch.inverse(intr[:, :3, :3]))[0].contiguous()
        o = c2w[0, :3, 3].contiguous()
        return M, o

    def _generate(self, H, W, intrinsics, c2w, inds=None):
        M, o = self._matrix(intrinsics, c2w)
        stride = self.render_stride
        if inds is None:
            n = len(range(0, H, stride)) * len(range(0, W, stride))
            ip, ni = None, 0
        else:
            inds = torch.as_tensor(inds, dtype=torch.int64, device=self.device).contiguous()
            n, ip, ni = inds.numel(), C.c_void_p(inds.data_ptr()), inds.numel()
        rays_o = torch.empty((n, 3), device=self.device)
        rays_d = torch.empty((n, 3), device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gpn_rays_single_image(
                C.c_void_p(M.data_ptr()), C.c_void_p(o.data_ptr()), H, W, stride, ip, ni,
                _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.stream()), "gpn_rays_single_image")
        return rays_o, rays_d

    def get_rays_single_image(self, H, W, intrinsics, c2w):
        """reference :108-132 -> (rays_o, rays_d) [H'*W', 3], on the device."""
        return self._generate(H, W, intrinsics, c2w)

    def get_all(self):
        """reference :134-147 (every entry on the device; rays are already there)."""
        d = self.device
        return {
            "ray_o": self.rays_o,
            "ray_d": self.rays_d,
            "depth_range": _dev(self.depth_range, d),
            "camera": _dev(self.camera, d),
            "true_depth": _dev(self.true_depth, d),
            "rgb": _dev(self.rgb, d),
            "src_rgbs": _dev(self.src_rgbs, d),
            "labels": _dev(self.labels, d),
            "src_labels": _dev(self.src_labels, d),
            "src_cameras": _dev(self.src_cameras, d),
        }

    def sample_random_pixel(self, N_rand, sample_mode, center_ratio=0.8):
        """Flat pixel indices of N_rand distinct pixels, drawn with the module's RandomState exactly as
        reference sample_ray.py:149-170 draws them (same stream, same values): 'uniform' over the image,
        'center' over the window that leaves a margin of (1 - center_ratio)/2 on every side."""
        if sample_mode == "uniform":
            return rng.choice(self.H * self.W, size=(N_rand,), replace=False)
        if sample_mode != "center":
            raise Exception("unknown sample mode!")
        top = int(self.H * (1 - center_ratio) / 2.0)
        left = int(self.W * (1 - center_ratio) / 2.0)
        n_rows, n_cols = self.H - 2 * top, self.W - 2 * left
        # the reference numbers the window's pixels column by column (np.meshgrid of (rows, cols) in
        # 'xy' order, flattened): candidate k is row top + k % n_rows, column left + k // n_rows
        k = rng.choice(n_rows * n_cols, size=(N_rand,), replace=False)
        return (left + k // n_rows) + self.W * (top + k % n_rows)

    def random_sample(self, N_rand, sample_mode, center_ratio=0.8):
        """reference :172-214: N_rand rays of the image (indices chosen on the host as in the
        reference, the rays themselves gathered on the device)."""
        select_inds = self.sample_random_pixel(N_rand, sample_mode, center_ratio)
        idx = torch.as_tensor(select_inds, dtype=torch.int64, device=self.device)
        d = self.device

        def pick(t):
            return _dev(t, d)[idx] if t is not None else None

        return {
            "ray_o": self.rays_o[idx],
            "ray_d": self.rays_d[idx],
            "camera": _dev(self.camera, d),
            "depth_range": _dev(self.depth_range, d),
            "rgb": pick(self.rgb),
            "true_depth": pick(self.true_depth),
            "depth_mask": pick(self.depth_mask) if self.true_depth is not None else None,
            "src_rgbs": _dev(self.src_rgbs, d),
            "labels": pick(self.labels),
            "src_labels": _dev(self.src_labels, d),
            "src_cameras": _dev(self.src_cameras, d),
            "selected_inds": select_inds,
        }
