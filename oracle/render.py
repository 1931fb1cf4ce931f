"""Oracle (test infrastructure): CPU restatement of reference gpnerf/render_ray.py::render_rays
(lines 194-300, model_type='gnt') and gpnerf/render_image.py::render_single_image (lines 6-121)."""
from collections import OrderedDict

import torch

from . import gnt, projector, sampling


def _one_pass(sd, pts, z_vals, ray_batch, featmaps, deep_semantics, ret_alpha, trans_depth):
    rgb_feat, deep, ray_diff, mask = projector.compute(
        pts, ray_batch["camera"], ray_batch["src_rgbs"], ray_batch["src_cameras"],
        featmaps, deep_semantics)
    out, feats_out, feats_out_3d = gnt.forward(sd, rgb_feat, deep, ray_diff, mask, pts,
                                               ray_batch["ray_d"], trans_depth=trans_depth)
    if ret_alpha:
        rgb, weights = out[:, 0:3], out[:, 3:]
        depth = torch.sum(weights * z_vals, dim=-1)  # render_ray.py:255-256
    else:
        rgb, weights, depth = out, None, None
    return {"rgb": rgb, "weights": weights, "depth": depth, "feats_out": feats_out,
            "feats_out_3d": feats_out_3d}


def render_rays(ray_batch, sd, featmaps, ref_deep_semantics, N_samples, inv_uniform=False,
                N_importance=0, det=False, ret_alpha=False, trans_depth=8):
    """`sd` is the net_coarse state dict (single_net=True: the same net serves both passes)."""
    ret = {"outputs_coarse": None, "outputs_fine": None}
    pts, z_vals = sampling.sample_along_camera_ray(
        ray_batch["ray_o"], ray_batch["ray_d"], ray_batch["depth_range"], N_samples,
        inv_uniform=inv_uniform, det=det)
    ret["outputs_coarse"] = _one_pass(sd, pts, z_vals, ray_batch, featmaps, ref_deep_semantics,
                                      ret_alpha, trans_depth)
    if N_importance > 0:
        weights = ret["outputs_coarse"]["weights"].clone().detach()
        pts, z_vals = sampling.sample_fine_pts(inv_uniform, N_importance, det, N_samples,
                                               ray_batch, weights, z_vals)
        ret["outputs_fine"] = _one_pass(sd, pts, z_vals, ray_batch, featmaps, ref_deep_semantics,
                                        True, trans_depth)
    return ret


_SHARED_KEYS = ("camera", "depth_range", "src_rgbs", "src_cameras", "labels", "src_labels")


def render_single_image(H, W, ray_batch, sd, chunk_size, N_samples, inv_uniform=False,
                        N_importance=0, det=False, render_stride=1, featmaps=None,
                        deep_semantics=None, ret_alpha=False, trans_depth=8):
    """Chunk loop + merge (render_image.py:35-121). H, W are ray_sampler.H / .W."""
    all_ret = OrderedDict([("outputs_coarse", OrderedDict()), ("outputs_fine", OrderedDict())])
    N = ray_batch["ray_o"].shape[0]
    for i in range(0, N, chunk_size):
        chunk = OrderedDict()
        for k, v in ray_batch.items():
            if k in _SHARED_KEYS:
                chunk[k] = v
            elif v is not None:
                chunk[k] = v[i:i + chunk_size]
            else:
                chunk[k] = None
        ret = render_rays(chunk, sd, featmaps, deep_semantics, N_samples, inv_uniform,
                          N_importance, det, ret_alpha, trans_depth)
        for lvl in ("outputs_coarse", "outputs_fine"):
            if ret[lvl] is None:
                all_ret[lvl] = None
                continue
            for k, v in ret[lvl].items():
                if v is not None:
                    all_ret[lvl].setdefault(k, []).append(v.cpu())
    Hs = len(range(0, H, render_stride))
    Ws = len(range(0, W, render_stride))
    for lvl in ("outputs_coarse", "outputs_fine"):
        if all_ret[lvl] is None:
            continue
        for k in all_ret[lvl]:
            cat = torch.cat(all_ret[lvl][k], dim=0)
            if k == "feats_out":
                all_ret[lvl][k] = cat.reshape((Hs, Ws, 512, -1)).squeeze()
            else:
                all_ret[lvl][k] = cat.reshape((Hs, Ws, -1)).squeeze()
    return all_ret
