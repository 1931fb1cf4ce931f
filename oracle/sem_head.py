"""Oracle (test infrastructure): CPU restatement of the EVAL branch of reference
gpnerf/semantic_branch.py::NeRFSemSegFPNHead (ctor lines 13-43, forward lines 89-107).

PARITY UNPINNED: the reference module imports detectron2 and fvcore, neither of which is installed
here, and the reference ships no test/fixture for it.  Third-party pieces restated from their
published behaviour (detectron2 v0.6 API):
  detectron2.layers.Conv2d(norm=..., activation=...)  == conv2d -> norm -> activation
  get_norm('GN', 128)                                  == GroupNorm(32, 128)
  fvcore c2_msra_fill                                  == kaiming_normal_(fan_out, relu), bias 0
"""
import torch
import torch.nn.functional as F

HEAD_LENGTHS = (1, 1, 2, 3)  # strides [2,4,8,16] vs common stride 2 (semantic_branch.py:20-22)


def _scale_head(sd, i, x):
    for k in range(HEAD_LENGTHS[i]):
        j = k if i == 0 else 2 * k
        p = "scale_heads.%d.%d." % (i, j)
        x = F.conv2d(x, sd[p + "weight"], None, stride=1, padding=1)
        x = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
        x = F.relu(x)
        if i != 0:
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    return x


def forward(sd, deep_feats, hw=None):
    """deep_feats [1,512,h,w] -> logits [1,num_classes+1,2h,2w].

    The reference hard-codes (h, w) = (120, 160) in its reshape (semantic_branch.py:95); `hw`
    generalises it (default: taken from the input)."""
    h, w = hw if hw is not None else deep_feats.shape[-2:]
    flat = deep_feats.reshape(1, deep_feats.shape[1], -1).squeeze(0).permute(1, 0)  # [h*w,512]
    x = None
    for i, chunk in enumerate(torch.chunk(flat, 4, dim=1)):
        chunk = chunk.reshape(h, w, chunk.shape[-1]).permute(2, 0, 1).unsqueeze(0)
        if i == 0:
            x = _scale_head(sd, 0, chunk)
        else:
            chunk = F.interpolate(chunk, scale_factor=1 / (2 ** i), mode="bilinear",
                                  align_corners=True, recompute_scale_factor=True)
            x = x + _scale_head(sd, i, chunk)
    out = F.conv2d(x, sd["predictor.weight"], sd["predictor.bias"])
    return F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
