"""Deterministic synthetic scenes for tests, smoke and bench.

Everything here is host logic: cameras in the reference's 34-float layout
(`[H, W, K(4x4 row-major), c2w(4x4 row-major)]`, reference
gpnerf/data_loaders/replica_dataset.py:297-299), rays via the pixel-grid math of
`RaySamplerSingleImage.get_rays_single_image` (reference gpnerf/sample_ray.py:108-132),
source maps of the shapes the path consumes, and GNT / semantic-head weights.

Random numbers come from a counter-based integer hash (splitmix64 of
(seed, stream, index)) so that inputs regenerate bit-identically on any machine and
any torch build; no transcendental function is involved (normals are an
Irwin-Hall sum of four uniforms scaled to unit variance).
"""
import math

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(seed, stream, shape):
    """float32 uniforms in [0,1) on a 24-bit grid; function of (seed, stream, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(idx ^ base)
    u = (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)
    return u.astype(np.float32).reshape(shape)


def hash_normal(seed, stream, shape):
    """Unit-variance, zero-mean pseudo-normal: Irwin-Hall sum of 4 uniforms."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += hash_uniform(seed, stream * 4 + k + 1000003, (n,)).astype(np.float64)
    out = (acc - 2.0) * math.sqrt(3.0)
    return out.astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------- cameras

REPLICA_K = [[320.0, 0.0, 320.0, 0.0], [0.0, 319.5, 229.5, 0.0], [0, 0, 1, 0], [0, 0, 0, 1]]
SCANNET_K = [[289.0, 0.0, 159.5, 0.0], [0.0, 289.0, 119.5, 0.0], [0, 0, 1, 0], [0, 0, 0, 1]]


def scaled_intrinsics(K, sx, sy):
    K = np.array(K, dtype=np.float64)
    K[0, :3] *= sx
    K[1, :3] *= sy
    return K


def make_camera(H, W, K, c2w):
    cam = np.zeros(34, dtype=np.float32)
    cam[0], cam[1] = H, W
    cam[2:18] = np.asarray(K, dtype=np.float64).reshape(-1).astype(np.float32)
    cam[18:34] = np.asarray(c2w, dtype=np.float64).reshape(-1).astype(np.float32)
    return cam


def _yaw(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    R = np.eye(4)
    R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
    return R


def source_rig(V, H, W, K, dx=0.15, dy=0.05, yaw_deg=-2.0):
    """Source view i: x-translate dx*(i-(V-1)/2), y-translate dy*((i%3)-1), yaw yaw_deg*(i-(V-1)/2)."""
    cams = []
    for i in range(V):
        o = i - (V - 1) / 2.0
        c2w = _yaw(yaw_deg * o)
        c2w[0, 3] = dx * o
        c2w[1, 3] = dy * ((i % 3) - 1)
        cams.append(make_camera(H, W, K, c2w))
    return np.stack(cams)


def rays_single_image(H, W, K, c2w, render_stride=1):
    """d = c2w[:3,:3] . K[:3,:3]^-1 . [u,v,1], o = c2w[:3,3] (reference sample_ray.py:116-132)."""
    u, v = np.meshgrid(np.arange(W)[::render_stride], np.arange(H)[::render_stride])
    u = u.reshape(-1).astype(np.float32)
    v = v.reshape(-1).astype(np.float32)
    pix = torch.from_numpy(np.stack((u, v, np.ones_like(u)), axis=0))[None]
    K = torch.as_tensor(np.asarray(K, dtype=np.float32)).reshape(1, 4, 4)
    c2w = torch.as_tensor(np.asarray(c2w, dtype=np.float32)).reshape(1, 4, 4)
    d = c2w[:, :3, :3].bmm(torch.inverse(K[:, :3, :3])).bmm(pix).transpose(1, 2).reshape(-1, 3)
    o = c2w[:, :3, 3].unsqueeze(1).repeat(1, d.shape[0], 1).reshape(-1, 3)
    return o.contiguous(), d.contiguous()


class Scene:
    """Holds one synthetic query view + V source views; tensors are CPU float32."""

    def __init__(self, H, W, V, K, depth_range, feat_ch=32, sem_ch=512, seed=1234,
                 rig=None, render_stride=1, map_source="hash"):
        self.H, self.W, self.V = H, W, V
        self.render_stride = render_stride
        rig = rig or {}
        self.camera = torch.from_numpy(make_camera(H, W, K, np.eye(4)))[None]  # [1,34]
        self.src_cameras = torch.from_numpy(source_rig(V, H, W, K, **rig))[None]  # [1,V,34]
        self.depth_range = torch.tensor([depth_range], dtype=torch.float32)  # [1,2]
        hf, wf, hs, ws = H // 4, W // 4, H // 2, W // 2
        if map_source == "hash":
            self.src_rgbs = torch.from_numpy(hash_uniform(seed, 1, (1, V, H, W, 3)))
            self.featmaps = torch.from_numpy(hash_normal(seed, 2, (V, feat_ch, hf, wf)))
            self.deep_semantics = torch.from_numpy(hash_normal(seed, 3, (V, sem_ch, hs, ws)))
        else:  # "torch": fast generator for bench-sized maps (not bit-stable across builds)
            g = torch.Generator().manual_seed(seed)
            self.src_rgbs = torch.rand((1, V, H, W, 3), generator=g)
            self.featmaps = torch.randn((V, feat_ch, hf, wf), generator=g)
            self.deep_semantics = torch.randn((V, sem_ch, hs, ws), generator=g)
        self.rays_o, self.rays_d = rays_single_image(H, W, K, np.eye(4), render_stride)

    def ray_batch(self, start=0, count=None, device="cpu"):
        sl = slice(start, None if count is None else start + count)
        return {
            "ray_o": self.rays_o[sl].to(device),
            "ray_d": self.rays_d[sl].to(device),
            "depth_range": self.depth_range.to(device),
            "camera": self.camera.to(device),
            "src_rgbs": self.src_rgbs.to(device),
            "src_cameras": self.src_cameras.to(device),
        }


def replica_scene(V=10, scale=1.0, seed=1234, render_stride=1, map_source="hash"):
    H, W = int(480 * scale), int(640 * scale)
    return Scene(H, W, V, scaled_intrinsics(REPLICA_K, scale, scale), (0.1, 10.0),
                 seed=seed, render_stride=render_stride, map_source=map_source)


def scannet_scene(V=10, scale=1.0, seed=1234, render_stride=1, map_source="hash"):
    H, W = int(240 * scale), int(320 * scale)
    return Scene(H, W, V, scaled_intrinsics(SCANNET_K, scale, scale), (0.1, 6.0),
                 seed=seed, render_stride=render_stride, map_source=map_source)


def c1_scene(seed=1234):
    """BASELINE config 1: 8 source views 320x240, tight rig (5 cm spacing, no rotation)."""
    return Scene(240, 320, 8, SCANNET_K, (0.1, 6.0), seed=seed,
                 rig=dict(dx=0.05, dy=0.0, yaw_deg=0.0))


# --------------------------------------------------------------------------- weights

def gnt_param_shapes(netwidth=64, trans_depth=8, in_feat_ch=32, posenc_dim=63, viewenc_dim=63,
                     sem_ch=512):
    """(name, shape) in the reference's state-dict order (transformer_network.py:222-294)."""
    C = netwidth
    out = [("rgbfeat_fc.0.weight", (C, in_feat_ch + 3)), ("rgbfeat_fc.0.bias", (C,)),
           ("rgbfeat_fc.2.weight", (C, C)), ("rgbfeat_fc.2.bias", (C,))]

    def norms_ff(p):
        return [(p + "attn_norm.weight", (C,)), (p + "attn_norm.bias", (C,)),
                (p + "ff_norm.weight", (C,)), (p + "ff_norm.bias", (C,)),
                (p + "ff.fc1.weight", (4 * C, C)), (p + "ff.fc1.bias", (4 * C,)),
                (p + "ff.fc2.weight", (C, 4 * C)), (p + "ff.fc2.bias", (C,))]

    for i in range(trans_depth):
        p = "view_selftrans.%d." % i
        out += norms_ff(p)
        out += [(p + "attn.q_fc.weight", (C, C)), (p + "attn.k_fc.weight", (C, C)),
                (p + "attn.v_fc.weight", (C, C)),
                (p + "attn.out_fc.weight", (C, C)), (p + "attn.out_fc.bias", (C,))]
    for i in range(trans_depth):
        p = "view_crosstrans.%d." % i
        out += norms_ff(p)
        out += [(p + "attn.q_fc.weight", (C, C)), (p + "attn.k_fc.weight", (C, C)),
                (p + "attn.v_fc.weight", (C, C)),
                (p + "attn.pos_fc.0.weight", (C // 8, 4)), (p + "attn.pos_fc.0.bias", (C // 8,)),
                (p + "attn.pos_fc.2.weight", (C, C // 8)), (p + "attn.pos_fc.2.bias", (C,)),
                (p + "attn.attn_fc.0.weight", (C // 8, C)), (p + "attn.attn_fc.0.bias", (C // 8,)),
                (p + "attn.attn_fc.2.weight", (C, C // 8)), (p + "attn.attn_fc.2.bias", (C,)),
                (p + "attn.out_fc.weight", (C, C)), (p + "attn.out_fc.bias", (C,))]
    for i in range(0, trans_depth, 2):
        p = "q_fcs.%d." % i
        out += [(p + "0.weight", (C, C + posenc_dim + viewenc_dim)), (p + "0.bias", (C,)),
                (p + "2.weight", (C, C)), (p + "2.bias", (C,))]
    out += [("norm.weight", (C,)), ("norm.bias", (C,)),
            ("rgb_fc.weight", (3, C)), ("rgb_fc.bias", (3,)),
            ("semantic_adaptor.0.weight", (sem_ch, sem_ch)), ("semantic_adaptor.0.bias", (sem_ch,))]
    return out


def gnt_state_dict(seed=5678, **kw):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear weights and biases; LayerNorm weight 1, bias 0."""
    sd = {}
    fan_in = {}
    shapes = gnt_param_shapes(**kw)
    for name, shape in shapes:
        if name.endswith(".weight") and len(shape) == 2:
            fan_in[name[: -len(".weight")]] = shape[1]
    for t, (name, shape) in enumerate(shapes):
        mod = name.rsplit(".", 1)[0]
        if "norm" in mod.split(".")[-1]:
            val = np.ones(shape, np.float32) if name.endswith("weight") else np.zeros(shape, np.float32)
        else:
            bound = 1.0 / math.sqrt(fan_in[mod])
            val = ((hash_uniform(seed, 100 + t, shape).astype(np.float64) * 2.0 - 1.0) * bound)
            val = val.astype(np.float32)
        sd[name] = torch.from_numpy(val)
    return sd


def sem_head_param_shapes(num_classes=20, conv_dims=128):
    """NeRFSemSegFPNHead state-dict (reference semantic_branch.py:13-43); strides [2,4,8,16]."""
    out = []
    for i, head_len in enumerate([1, 1, 2, 3]):
        for k in range(head_len):
            j = k if i == 0 else 2 * k  # an Upsample follows every conv when stride != 2
            p = "scale_heads.%d.%d." % (i, j)
            out += [(p + "weight", (conv_dims, conv_dims, 3, 3)),
                    (p + "norm.weight", (conv_dims,)), (p + "norm.bias", (conv_dims,))]
    out += [("predictor.weight", (num_classes + 1, conv_dims, 1, 1)),
            ("predictor.bias", (num_classes + 1,))]
    return out


def sem_head_state_dict(seed=9012, num_classes=20):
    """Kaiming-normal(fan_out, relu) conv weights (c2_msra_fill), GN weight 1 / bias 0, zero conv bias."""
    sd = {}
    for t, (name, shape) in enumerate(sem_head_param_shapes(num_classes)):
        if name.endswith("norm.weight"):
            val = np.ones(shape, np.float32)
        elif name.endswith("bias"):
            val = np.zeros(shape, np.float32)
        else:
            fan_out = shape[0] * shape[2] * shape[3]
            val = hash_normal(seed, 200 + t, shape) * np.float32(math.sqrt(2.0 / fan_out))
        sd[name] = torch.from_numpy(val.astype(np.float32))
    return sd


def hashed_state_dict(shapes, seed):
    """Deterministic weights for a conv net given [(name, shape)] in state-dict order: conv / linear
    weights U(+-1/sqrt(fan_in)), norm weights 1 + 0.1 n, every bias 0.1 n (n = hashed unit normal)."""
    sd = {}
    for t, (name, shape) in enumerate(shapes):
        shape = tuple(shape)
        if name.endswith("weight") and len(shape) >= 2:
            bound = 1.0 / math.sqrt(float(np.prod(shape[1:])))
            val = (hash_uniform(seed, 500 + t, shape).astype(np.float64) * 2.0 - 1.0) * bound
        elif name.endswith("weight"):
            val = 1.0 + 0.1 * hash_normal(seed, 500 + t, shape).astype(np.float64)
        else:
            val = 0.1 * hash_normal(seed, 500 + t, shape).astype(np.float64)
        sd[name] = torch.from_numpy(val.astype(np.float32))
    return sd
