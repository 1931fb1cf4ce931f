"""CPU restatement of the 2-D encoders (test infrastructure: only tests/, smoke() and bench.py's
cpu_baseline may import it), written functionally over a state dict.

  resunet_forward : reference gpnerf/feature_network.py:293-313 (+ blocks :26-70, :133-161, :280-291).
                    Pinned by tests/golden/g7_resunet.npz, produced by the reference's own ResUNet.
  fpn_forward     : reference gpnerf/fpn.py:161-227 (mmdet FPN + concat_out).  The reference module needs
                    mmcv / mmengine / mmdet (absent here) -> PARITY UNPINNED: restated from the source text.
"""
import torch
import torch.nn.functional as F


def _conv(sd, name, x, stride=1, pad=0):
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(x, w, b, stride=stride)


def _inorm(sd, name, x):
    return F.instance_norm(x, weight=sd[name + ".weight"], bias=sd[name + ".bias"], eps=1e-5)


def _block(sd, p, x, stride):
    y = F.relu(_inorm(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    y = _inorm(sd, p + ".bn2", _conv(sd, p + ".conv2", y, 1, 1))
    if (p + ".downsample.0.weight") in sd:
        x = _inorm(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0))
    return F.relu(y + x)


def _cie(sd, p, x):
    return F.elu(_inorm(sd, p + ".bn", _conv(sd, p + ".conv", x, 1, 1)))


def _skip(skip, up):
    dy, dx = up.shape[2] - skip.shape[2], up.shape[3] - skip.shape[3]
    return torch.cat([up, F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))], dim=1)


@torch.no_grad()
def resunet_forward(sd, x, blocks=(3, 4, 6)):
    x0 = F.relu(_inorm(sd, "bn1", _conv(sd, "conv1", x, 2, 3)))
    feats = [x0]
    y = x0
    for li, n in enumerate(blocks, start=1):
        for b in range(n):
            y = _block(sd, "layer%d.%d" % (li, b), y, 2 if b == 0 else 1)
        feats.append(y)
    x1, x2, x3 = feats[1:]
    up = F.interpolate(x3, scale_factor=2, align_corners=True, mode="bilinear")
    y = _cie(sd, "iconv3", _skip(x2, _cie(sd, "upconv3.conv", up)))
    up = F.interpolate(y, scale_factor=2, align_corners=True, mode="bilinear")
    y = _cie(sd, "iconv2", _skip(x1, _cie(sd, "upconv2.conv", up)))
    out = _conv(sd, "out_conv", y)
    return out, out, feats


@torch.no_grad()
def fpn_forward(sd, inputs):
    n = len(inputs)
    lat = [F.conv2d(x, sd["lateral_convs.%d.conv.weight" % i], sd["lateral_convs.%d.conv.bias" % i])
           for i, x in enumerate(inputs)]
    for i in range(n - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    outs = [F.conv2d(x, sd["fpn_convs.%d.conv.weight" % i], sd["fpn_convs.%d.conv.bias" % i], padding=1)
            for i, x in enumerate(lat)]
    size = outs[0].shape[-2:]
    outs = [outs[0]] + [F.interpolate(o, size, mode="bilinear", align_corners=True) for o in outs[1:]]
    return torch.cat(outs, dim=1)
