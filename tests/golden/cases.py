"""Seeded input builders shared by make_golden.py (reference run) and the tests (oracle / HIP run).
All tensors are CPU float32 and regenerate bit-identically from gp_nerf_amd.synth's integer hash."""
import hashlib

import numpy as np
import torch

from gp_nerf_amd import synth


def sha(case):
    h = hashlib.sha256()

    def feed(v):
        if isinstance(v, torch.Tensor):
            h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
        elif isinstance(v, dict):
            for k in sorted(v):
                feed(v[k])

    feed(case)
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def weights():
    return synth.gnt_state_dict(seed=5678)


def _small_scene(H, W, V, seed, depth=(0.1, 6.0), rig=None):
    K = synth.scaled_intrinsics(synth.SCANNET_K, W / 320.0, H / 240.0)
    return synth.Scene(H, W, V, K, depth, seed=seed, rig=rig)


def g1_projector():
    """V=4, 64x48 image, R=64 rays x S=16 samples; view 3 sits 1.5 m ahead (points behind it) and
    the rig is yawed hard so many samples leave the frusta; a few points are planted exactly on the
    image border / principal axis."""
    H, W, V, R, S = 48, 64, 4, 64, 16
    sc = _small_scene(H, W, V, seed=1234, rig=dict(dx=0.4, dy=0.1, yaw_deg=-9.0))
    cams = sc.src_cameras.clone()
    cams[0, 3, 18 + 11] = 1.5  # c2w[2,3] of view 3: camera moved forward along +z
    sel = (torch.arange(R) * 47) % (H * W)
    ray_o, ray_d = sc.rays_o[sel], sc.rays_d[sel]
    z = 1.0 / torch.linspace(1.0 / 0.1, 1.0 / 6.0, S)
    pts = ray_o[:, None] + z[None, :, None] * ray_d[:, None]
    # planted points: on view 0's optical axis / exact borders (x = 0, x = W-1 when seen from view 0)
    Kinv = torch.inverse(cams[0, 0, 2:18].reshape(4, 4)[:3, :3])
    c2w0 = cams[0, 0, 18:34].reshape(4, 4)
    for n, (u, v, d) in enumerate([(0.0, 0.0, 2.0), (W - 1.0, H - 1.0, 2.0), (W - 1.0, 10.0, 3.0),
                                   (31.5, 23.5, 1.0), (W - 0.5, 10.0, 1.0)]):
        p_cam = d * (Kinv @ torch.tensor([u, v, 1.0]))
        pts[n, n % S] = c2w0[:3, :3] @ p_cam + c2w0[:3, 3]
    return dict(pts=pts.contiguous(), camera=sc.camera, src_rgbs=sc.src_rgbs, src_cameras=cams,
                featmaps=sc.featmaps, deep_semantics=sc.deep_semantics, H=H, W=W)


def g2_gnt(S, R=8, V=10):
    seed = 4242 + S
    rgb_feat = torch.from_numpy(synth.hash_normal(seed, 1, (R, S, V, 35)))
    deep = torch.from_numpy(synth.hash_normal(seed, 2, (R, S, V, 512)))
    rd = torch.from_numpy(synth.hash_normal(seed, 3, (R, S, V, 4)))
    rd[..., :3] = rd[..., :3] / rd[..., :3].norm(dim=-1, keepdim=True)
    rd[..., 3] = torch.tanh(rd[..., 3])
    mask = (torch.from_numpy(synth.hash_uniform(seed, 4, (R, S, V, 1))) < 0.6).float()
    mask[0, 5] = 0.0  # an all-masked sample -> uniform 1/V attention, not NaN
    mask[1, :, 0] = 0.0
    ray_o = torch.zeros(R, 3)
    ray_d = torch.from_numpy(synth.hash_normal(seed, 5, (R, 3))) * 0.3 + torch.tensor([0.0, 0.0, 1.0])
    z = 1.0 / torch.linspace(1.0 / 0.1, 1.0 / 10.0, S)
    pts = ray_o[:, None] + z[None, :, None] * ray_d[:, None]
    return dict(rgb_feat=rgb_feat, deep_sem=deep, ray_diff=rd, mask=mask, pts=pts.contiguous(),
                ray_d=ray_d)


def g3_samplers(R=96, S=48):
    seed = 777
    ray_o = torch.from_numpy(synth.hash_normal(seed, 1, (R, 3))) * 0.1
    ray_d = torch.from_numpy(synth.hash_normal(seed, 2, (R, 3))) * 0.4 + torch.tensor([0.0, 0.0, 1.0])
    w = torch.from_numpy(synth.hash_uniform(seed, 3, (R, S))) ** 4
    w[0] = 0.0           # flat pdf
    w[1] = 0.0
    w[1, 20] = 1.0       # a single spike
    w[2, :] = 1.0 / S
    w = w / w.sum(-1, keepdim=True).clamp(min=1e-9)
    raw = torch.from_numpy(synth.hash_normal(seed, 4, (R, S, 4)))
    raw[..., 3] = raw[..., 3].abs() * 0.5
    vis = torch.from_numpy(synth.hash_uniform(seed, 5, (R, S))) < 0.3
    return dict(ray_o=ray_o, ray_d=ray_d, depth_range=torch.tensor([[0.1, 6.0]]), weights=w,
                raw=raw, vis_mask=vis)


def g4_render_rays(R=64):
    sc = _small_scene(96, 128, 8, seed=2024, rig=dict(dx=0.05, dy=0.0, yaw_deg=-1.0))
    n = sc.rays_o.shape[0]
    sel = (torch.arange(R) * 193) % n
    rb = sc.ray_batch()
    rb["ray_o"], rb["ray_d"] = sc.rays_o[sel].contiguous(), sc.rays_d[sel].contiguous()
    return dict(ray_batch=rb, featmaps=sc.featmaps, deep_semantics=sc.deep_semantics)


def g5_render_image():
    sc = _small_scene(24, 32, 4, seed=31337, rig=dict(dx=0.1, dy=0.02, yaw_deg=-2.0))
    return dict(ray_batch=sc.ray_batch(), featmaps=sc.featmaps, deep_semantics=sc.deep_semantics,
                H=24, W=32)


def g6_sample_ray():
    """Two RaySamplerSingleImage inputs (reference sample_ray.py:38): a small ScanNet-shaped image with a
    yawed, translated camera at stride 1, and a Replica-shaped 480x640 image at stride 2."""
    out = {}
    for tag, (H, W, K, stride, i) in {"small": (48, 64, synth.scaled_intrinsics(synth.SCANNET_K, 0.2, 0.2), 1, 3),
                                      "replica": (480, 640, synth.REPLICA_K, 2, 7)}.items():
        cam = torch.from_numpy(synth.source_rig(10, H, W, K, dx=0.21, dy=0.07, yaw_deg=-3.5)[i])[None]
        out[tag] = dict(camera=cam, rgb=torch.from_numpy(synth.hash_uniform(77, 5, (1, H, W, 3))),
                        depth_range=torch.tensor([[0.1, 6.0]]), stride=torch.tensor(stride))
    return out


def g6_data(c):
    """The `data` dict RaySamplerSingleImage reads (keys of reference sample_ray.py:41-48)."""
    return {"rgb": c["rgb"], "camera": c["camera"], "rgb_path": "synthetic", "true_depth": None,
            "depth_mask": None, "depth_range": c["depth_range"]}


def _resunet_shapes():
    """ResUNet state-dict layout (reference feature_network.py:164-291, resnet34 / single_net / 32 ch)."""
    out = [("conv1.weight", (64, 3, 7, 7)), ("bn1.weight", (64,)), ("bn1.bias", (64,))]
    cin = 64
    for li, (cout, n) in enumerate(((64, 3), (128, 4), (256, 6)), start=1):
        for b in range(n):
            p = "layer%d.%d." % (li, b)
            out += [(p + "conv1.weight", (cout, cin if b == 0 else cout, 3, 3)), (p + "bn1.weight", (cout,)),
                    (p + "bn1.bias", (cout,)), (p + "conv2.weight", (cout, cout, 3, 3)),
                    (p + "bn2.weight", (cout,)), (p + "bn2.bias", (cout,))]
            if b == 0:
                out += [(p + "downsample.0.weight", (cout, cin, 1, 1)), (p + "downsample.1.weight", (cout,)),
                        (p + "downsample.1.bias", (cout,))]
        cin = cout
    for name, ci, co in (("upconv3.conv", 256, 128), ("iconv3", 256, 128), ("upconv2.conv", 128, 64),
                         ("iconv2", 128, 32)):
        out += [(name + ".conv.weight", (co, ci, 3, 3)), (name + ".conv.bias", (co,)),
                (name + ".bn.weight", (co,)), (name + ".bn.bias", (co,))]
    out += [("out_conv.weight", (32, 32, 1, 1)), ("out_conv.bias", (32,))]
    return out


def fpn_shapes(in_channels=(64, 64, 128, 256), out_channels=128):
    out = []
    for i, c in enumerate(in_channels):
        out += [("lateral_convs.%d.conv.weight" % i, (out_channels, c, 1, 1)),
                ("lateral_convs.%d.conv.bias" % i, (out_channels,))]
    for i, c in enumerate(in_channels):
        out += [("fpn_convs.%d.conv.weight" % i, (out_channels, out_channels, 3, 3)),
                ("fpn_convs.%d.conv.bias" % i, (out_channels,))]
    return out


def g7_encoders():
    """3 source images of 52x72 (odd H/16, W/16 sizes exercise skipconnect's padding) + hashed weights."""
    x = torch.from_numpy(synth.hash_uniform(4242, 9, (3, 3, 52, 72)))
    return dict(x=x, resunet=synth.hashed_state_dict(_resunet_shapes(), 777),
                fpn=synth.hashed_state_dict(fpn_shapes(), 778))


def g8_scannet_labels():
    """BASELINE.json configs[2] at full size: ScanNet-shaped 240x320 view, 10 source views, hashed maps; the
    19,200 rays of the half-resolution pixel grid the semantic head works on (feats_out[::2, ::2],
    reference eval_gpnerf.py:201-203) and a hashed semantic head (20 classes + 1)."""
    sc = synth.scannet_scene(V=10, seed=8086)
    H, W = sc.H, sc.W
    idx = (torch.arange(0, H, 2)[:, None] * W + torch.arange(0, W, 2)[None, :]).reshape(-1)
    rb = sc.ray_batch()
    rb["ray_o"], rb["ray_d"] = sc.rays_o[idx].contiguous(), sc.rays_d[idx].contiguous()
    return dict(ray_batch=rb, featmaps=sc.featmaps, deep_semantics=sc.deep_semantics, H=H, W=W,
                head=synth.hashed_state_dict(synth.sem_head_param_shapes(), 901))
