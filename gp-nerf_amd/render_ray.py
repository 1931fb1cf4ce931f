"""Ray-level rendering -- same interface as reference gpnerf/render_ray.py, computed by gfx950 HIP
kernels (csrc/sampling.hip, csrc/projector.hip, csrc/gnt.hip) through the C-ABI.

sample_pdf / sample_along_camera_ray / raw2outputs / sample_fine_pts / render_rays keep the
reference's names, argument order, defaults and return structures (reference render_ray.py:9-300).
Differences, all deliberate: caller tensors are never mutated (the reference adds 1e-5 in place to
its own clone of the weights); tensors must live on a HIP device (no CPU fallback).
"""
from collections import OrderedDict

import torch

from . import _lib


def _f(t, name):
    return _lib.dev_f32(_lib.as_f32c(t), name)


_range_cache = {}


def _depth_range(depth_range):
    """(near, far) as host floats; the device->host read happens once per tensor, not per chunk."""
    key = (depth_range.data_ptr(), depth_range._version, str(depth_range.device))
    hit = _range_cache.get(key)
    if hit is None:
        if len(_range_cache) > 64:
            _range_cache.clear()
        hit = (float(depth_range[0, 0]), float(depth_range[0, 1]), depth_range)
        _range_cache[key] = hit
    return hit[0], hit[1]


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False):
    """reference render_ray.py:58-102 -> pts [R,S,3], z_vals [R,S]."""
    near, far = _depth_range(depth_range)
    assert near > 0 and far > 0 and far > near
    ray_o, ray_d = _f(ray_o, "ray_o"), _f(ray_d, "ray_d")
    R = ray_d.shape[0]
    pts = torch.empty((R, N_samples, 3), device=ray_d.device)
    z = torch.empty((R, N_samples), device=ray_d.device)
    t_rand = None if det else torch.rand((R, N_samples), device=ray_d.device)
    _lib.check(_lib.lib().gpn_sample_along_ray(
        _lib.ptr(ray_o), _lib.ptr(ray_d), near, far, R, N_samples, int(bool(inv_uniform)),
        _lib.ptr(t_rand), _lib.ptr(pts), _lib.ptr(z), _lib.stream()), "gpn_sample_along_ray")
    return pts, z


_det_u_cache = {}


def _det_u(n, dev):
    """The deterministic quantiles of the inverse-CDF lookup: the reference's host formula
    (torch.linspace on the CPU, render_ray.py:36), copied to the device ONCE per (n, device) -- a
    pageable host-to-device copy per call would make the host wait for the whole coarse pass."""
    key = (int(n), str(dev))
    u = _det_u_cache.get(key)
    if u is None:
        u = _det_u_cache[key] = torch.linspace(0.0, 1.0, n).to(dev)
    return u


def _fine(ray_o, ray_d, weights, z_vals, N_importance, det, inv_uniform):
    ray_o, ray_d = _f(ray_o, "ray_o"), _f(ray_d, "ray_d")
    weights, z_vals = _f(weights, "weights"), _f(z_vals, "z_vals")
    R, S = z_vals.shape
    dev = z_vals.device
    if det:
        u = _det_u(N_importance, dev)
        stride = 0
    else:
        u = torch.rand((R, N_importance), device=dev)
        stride = N_importance
    T = S + N_importance
    pts = torch.empty((R, T, 3), device=dev)
    z = torch.empty((R, T), device=dev)
    _lib.check(_lib.lib().gpn_sample_fine(
        _lib.ptr(ray_o), _lib.ptr(ray_d), _lib.ptr(weights), _lib.ptr(z_vals), _lib.ptr(u), stride,
        R, S, N_importance, int(bool(inv_uniform)), _lib.ptr(pts), _lib.ptr(z), _lib.stream()),
        "gpn_sample_fine")
    return pts, z


def sample_fine_pts(inv_uniform, N_importance, det, N_samples, ray_batch, weights, z_vals):
    """reference render_ray.py:162-191 -> pts [R,S+N_imp,3], z_vals [R,S+N_imp] (sorted)."""
    return _fine(ray_batch["ray_o"], ray_batch["ray_d"], weights, z_vals, N_importance, det,
                 inv_uniform)


def sample_pdf(bins, weights, N_samples, det=False):
    """reference render_ray.py:9-55: bins [R,M+1], weights [R,M] -> samples [R,N_samples].
    `weights` is not modified (the reference adds 1e-5 to it in place)."""
    bins, weights = _f(bins, "bins"), _f(weights, "weights")
    R, M = weights.shape
    dev = bins.device
    if det:
        u, stride = _det_u(N_samples, dev), 0
    else:
        u, stride = torch.rand((R, N_samples), device=dev), N_samples
    out = torch.empty((R, N_samples), device=dev)
    _lib.check(_lib.lib().gpn_sample_pdf(_lib.ptr(bins), _lib.ptr(weights), _lib.ptr(u), stride, R,
                                         M, N_samples, _lib.ptr(out), _lib.stream()),
               "gpn_sample_pdf")
    return out


def raw2outputs(raw, z_vals, mask, white_bkgd=False):
    """reference render_ray.py:110-159 (classic compositing, model_type != 'gnt')."""
    raw, z_vals = _f(raw, "raw"), _f(z_vals, "z_vals")
    vis = _f(mask.float(), "mask")
    R, S = z_vals.shape
    dev = z_vals.device
    rgb = torch.empty((R, 3), device=dev)
    depth = torch.empty((R,), device=dev)
    weights = torch.empty((R, S), device=dev)
    alpha = torch.empty((R, S), device=dev)
    rmask = torch.empty((R,), device=dev)
    _lib.check(_lib.lib().gpn_raw2outputs(
        _lib.ptr(raw), _lib.ptr(z_vals), _lib.ptr(vis), R, S, int(bool(white_bkgd)), _lib.ptr(rgb),
        _lib.ptr(depth), _lib.ptr(weights), _lib.ptr(alpha), _lib.ptr(rmask), _lib.stream()),
        "gpn_raw2outputs")
    return OrderedDict([("rgb", rgb), ("depth", depth), ("weights", weights),
                        ("mask", rmask > 0.5), ("alpha", alpha), ("z_vals", z_vals)])


def _gnt_pass(net, projector, sv, pts, z_vals, ray_d, ret_alpha, want_3d=True, row_hint=None):
    """One coarse/fine pass on the fused path (gpn_render_pass): the projector runs fused into the
    network's first stage (no [R,S,V,35] tensor), the semantic aggregation samples the NHWC
    deep-semantic maps itself (no [R,S,V,512] tensor)."""
    deep_map = None
    if sv.deep is not None:
        deep_map = dict(map=sv.deep, proj=sv.proj, img_h=sv.img_h, img_w=sv.img_w)
    out, depth, feats_out, feats_out_3d = net.run(
        None, None, None, pts, ray_d, z_vals=z_vals if ret_alpha else None,
        deep_map=deep_map, want_sem3d=want_3d, proj=projector.fused_args(sv), row_hint=row_hint)
    if ret_alpha:
        rgb, weights = out[:, 0:3], out[:, 3:]
    else:
        rgb, weights, depth = out, None, None
    return {"rgb": rgb, "weights": weights, "depth": depth, "feats_out": feats_out,
            "feats_out_3d": feats_out_3d}


def render_rays(
    ray_batch,
    model,
    featmaps,
    ref_deep_semantics,
    projector,
    N_samples,
    inv_uniform=False,
    N_importance=0,
    det=False,
    white_bkgd=False,
    ret_alpha=False,
    single_net=True,
    save_feature=False,
    model_type="gnt",
    feats_out_3d=True,
):
    """reference render_ray.py:194-300.  Returns {'outputs_coarse': {...}, 'outputs_fine': {...}} with
    keys rgb / weights / depth / feats_out / feats_out_3d (None where the reference returns None).

    `feats_out_3d=False` (extension, default keeps the reference behaviour) skips the [R,S,512]
    tensor, which the reference computes and copies to the host but never reads in eval."""
    ret = {"outputs_coarse": None, "outputs_fine": None}
    ray_o, ray_d = ray_batch["ray_o"], ray_batch["ray_d"]
    pts, z_vals = sample_along_camera_ray(ray_o=ray_o, ray_d=ray_d,
                                          depth_range=ray_batch["depth_range"],
                                          N_samples=N_samples, inv_uniform=inv_uniform, det=det)
    sv = projector.prepare(ray_batch["camera"], ray_batch["src_rgbs"], ray_batch["src_cameras"],
                           featmaps, ref_deep_semantics)
    fused = model_type == "gnt" and hasattr(model.net_coarse, "run")
    # optional key (not in the reference's ray_batch): (rays per image row, column of ray 0) when the rays of this
    # batch are consecutive pixels in row-major order -- render_single_image sets it per chunk.  A scheduling
    # hint for the semantic pass only; results do not depend on it.
    row_hint = ray_batch.get("pixel_row")

    if fused:
        ret["outputs_coarse"] = _gnt_pass(model.net_coarse, projector, sv, pts, z_vals, ray_d,
                                          ret_alpha, feats_out_3d, row_hint)
    else:
        rgb_feat, deep, ray_diff, mask = projector.compute(
            pts, ray_batch["camera"], ray_batch["src_rgbs"], ray_batch["src_cameras"],
            featmaps=featmaps, deep_semantics=ref_deep_semantics)
        if model_type == "gnt":  # a foreign net with the reference call signature
            rgb_out, feats_out, f3d = model.net_coarse(rgb_feat, deep, ray_diff, mask, pts, ray_d)
            if ret_alpha:
                rgb_out, weights = rgb_out[:, 0:3], rgb_out[:, 3:]
                depth_map = torch.sum(weights * z_vals, dim=-1)
            else:
                weights = depth_map = None
            ret["outputs_coarse"] = {"rgb": rgb_out, "weights": weights, "depth": depth_map,
                                     "feats_out": feats_out, "feats_out_3d": f3d}
        else:
            pixel_mask = mask[..., 0].sum(dim=2) > 1
            raw_coarse, _, _ = model.net_coarse(rgb_feat, ray_diff, mask)
            ret["outputs_coarse"] = raw2outputs(raw_coarse, z_vals, pixel_mask,
                                                white_bkgd=white_bkgd)
            ret["outputs_coarse"]["sems"] = None

    if N_importance > 0:
        weights = ret["outputs_coarse"]["weights"].detach()  # never mutated here (no clone needed)
        pts, z_vals = sample_fine_pts(inv_uniform, N_importance, det, N_samples, ray_batch, weights,
                                      z_vals)
        net = model.net_coarse if single_net else model.net_fine
        if hasattr(net, "run"):
            ret["outputs_fine"] = _gnt_pass(net, projector, sv, pts, z_vals, ray_d, True,
                                            feats_out_3d, row_hint)
        else:
            rgb_feat, deep, ray_diff, mask = projector.compute(
                pts, ray_batch["camera"], ray_batch["src_rgbs"], ray_batch["src_cameras"],
                featmaps=featmaps, deep_semantics=ref_deep_semantics)
            if single_net:
                out = net(rgb_feat, deep, ray_diff, mask, pts, ray_d)
            else:
                out = net(rgb_feat, ray_diff, mask, pts, ray_d)
            rgb_out, feats_out, f3d = out
            rgb_out, weights = rgb_out[:, 0:3], rgb_out[:, 3:]
            ret["outputs_fine"] = {"rgb": rgb_out, "weights": weights,
                                   "depth": torch.sum(weights * z_vals, dim=-1),
                                   "feats_out": feats_out, "feats_out_3d": f3d}
    return ret
