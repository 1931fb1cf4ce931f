"""2-D encoders upstream of the ray path: `ResUNet` (reference
gpnerf/feature_network.py:164-313, ResNet-34 layout, InstanceNorm, reflect padding) and `FPN`
(reference gpnerf/fpn.py:67-227: mmdet's FPN with the reference's `concat_out` tail), with the
reference's parameter names so that a reference checkpoint's `feature_net` / `feature_fpn` state
dicts load unchanged.

They run once per image, on library convolutions (MIOpen through torch) -- not on the per-ray hot
path -- but they are laid out for it: the modules compute in channels_last, so the coarse feature map
[V,32,H/4,W/4] and the deep-semantic map [V,512,H/2,W/2] leave the encoder ALREADY in the NHWC memory
layout the projection / gather kernels read (`Projector` takes a channels_last map as is: no
NCHW->NHWC repack pass over the 1.6 GB of maps per image).

Inference only; inputs must be on a HIP device (no CPU fallback -- the CPU restatement used by the
tests lives in oracle/encoders.py).  mmcv's ConvModule(conv_cfg=None, norm_cfg=None, act_cfg=None) is
a bare nn.Conv2d with bias held under `.conv`; its Xavier-uniform init is reproduced.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_CL = torch.channels_last


def _need_hip(x, who):
    if not x.is_cuda:
        raise RuntimeError("%s: input is on %s; the encoders run on a HIP device only (no CPU fallback)"
                           % (who, x.device))


def _inorm(ch):
    return nn.InstanceNorm2d(ch, track_running_stats=False, affine=True)


def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False, padding_mode="reflect")


class _Res(nn.Module):
    """BasicBlock (reference :26-70): conv-IN-ReLU-conv-IN (+ 1x1/IN shortcut) -> add -> ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.bn1 = _c3(cin, cout, stride), _inorm(cout)
        self.conv2, self.bn2 = _c3(cout, cout), _inorm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, bias=False, padding_mode="reflect"), _inorm(cout))

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _CIE(nn.Module):
    """conv -> InstanceNorm -> ELU (reference `conv`, :133-149); children named conv / bn."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, 1, (k - 1) // 2, padding_mode="reflect")
        self.bn = _inorm(cout)

    def forward(self, x):
        return F.elu(self.bn(self.conv(x)))


class _Up(nn.Module):
    """x2 bilinear (align_corners=True) then conv-IN-ELU (reference `upconv`, :152-161)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _CIE(cin, cout, 3)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, align_corners=True, mode="bilinear"))


def _stage(cin, cout, n):
    return nn.Sequential(_Res(cin, cout, 2), *[_Res(cout, cout, 1) for _ in range(n - 1)])


def _skip(skip, up):
    """reference skipconnect (:280-291): pad the skip tensor to the upsampled size, cat [up, skip]."""
    dy, dx = up.shape[2] - skip.shape[2], up.shape[3] - skip.shape[3]
    skip = F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return torch.cat([up, skip], dim=1)


class ResUNet(nn.Module):
    def __init__(self, encoder="resnet34", coarse_out_ch=32, fine_out_ch=32, norm_layer=None,
                 single_net=True):
        super().__init__()
        if encoder not in ("resnet18", "resnet34"):
            raise ValueError("only the BasicBlock encoders are built (the reference's forward uses "
                             "BasicBlock for every encoder name, feature_network.py:203)")
        self.single_net = single_net
        self.coarse_out_ch = coarse_out_ch
        self.fine_out_ch = coarse_out_ch if single_net else fine_out_ch
        out_ch = coarse_out_ch if single_net else coarse_out_ch + fine_out_ch
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False, padding_mode="reflect")
        self.bn1 = _inorm(64)
        self.layer1, self.layer2, self.layer3 = _stage(64, 64, 3), _stage(64, 128, 4), _stage(128, 256, 6)
        self.upconv3, self.iconv3 = _Up(256, 128), _CIE(128 + 128, 128, 3)
        self.upconv2, self.iconv2 = _Up(128, 64), _CIE(64 + 64, out_ch, 3)
        self.out_conv = nn.Conv2d(out_ch, out_ch, 1, 1)
        self.to(memory_format=_CL)

    @torch.no_grad()
    def forward(self, x):
        """[V,3,H,W] -> (coarse [V,32,H/4,W/4], fine (same tensor when single_net),
        [x0 (64, H/2), x1 (64, H/4), x2 (128, H/8), x3 (256, H/16)]) -- reference :293-313."""
        _need_hip(x, "ResUNet")
        x = x.contiguous(memory_format=_CL)
        x0 = F.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        y = self.iconv3(_skip(x2, self.upconv3(x3)))
        y = self.iconv2(_skip(x1, self.upconv2(y)))
        out = self.out_conv(y)
        if self.single_net:
            return out, out, [x0, x1, x2, x3]
        return out[:, : self.coarse_out_ch], out[:, -self.fine_out_ch:], [x0, x1, x2, x3]


class _MMConv(nn.Module):
    """mmcv ConvModule without norm / activation: the conv lives under `.conv`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2)
        nn.init.xavier_uniform_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        return self.conv(x)


class FPN(nn.Module):
    """mmdet FPN as the reference configures it (model.py:52: in_channels [64,64,128,256],
    out_channels 128, num_outs 4, concat_out): lateral 1x1 convs, nearest-neighbour top-down sums,
    3x3 output convs, then every level resized (bilinear, align_corners=True) to level 0 and
    concatenated -> [V, 4*out_channels, H/2, W/2] (reference fpn.py:161-227)."""

    def __init__(self, in_channels=(64, 128, 256, 512), out_channels=64, num_outs=4, concat_out=True):
        super().__init__()
        if num_outs != len(in_channels):
            raise ValueError("extra output levels are not built (the reference uses num_outs == inputs)")
        self.in_channels, self.out_channels, self.concat_out = list(in_channels), out_channels, concat_out
        self.lateral_convs = nn.ModuleList(_MMConv(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(_MMConv(out_channels, out_channels, 3) for _ in in_channels)
        self.to(memory_format=_CL)

    @torch.no_grad()
    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        _need_hip(inputs[0], "FPN")
        lat = [conv(x.contiguous(memory_format=_CL)) for conv, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [conv(x) for conv, x in zip(self.fpn_convs, lat)]
        if not self.concat_out:
            return tuple(outs)
        size = outs[0].shape[-2:]
        outs = [outs[0]] + [F.interpolate(o, size, mode="bilinear", align_corners=True) for o in outs[1:]]
        return torch.cat(outs, dim=1).contiguous(memory_format=_CL)
