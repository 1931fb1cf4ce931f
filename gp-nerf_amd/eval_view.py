"""`render_view` -- the compute part of the reference's `log_view` (eval_gpnerf.py:156-206; same in
train_gpnerf.py:339 / ft_gpnerf.py:342 / render.py:221): source-view encoding, full-image rendering,
semantic head and label image.  PNG dumps, wandb, LPIPS/SSIM/IoU scoring (eval_gpnerf.py:208-271) are
the drivers' business and stay out.

Everything between the ray sampler and the label image stays on the device; with `sharded=True`
(inside an initialised torch.distributed job) the image rows are split over the ranks and the
rgb / depth / feature tiles are all-gathered (render_image.render_image_sharded).
"""
import torch

from .render_image import render_image_sharded, render_single_image
from .semantic_branch import semantic_labels


@torch.no_grad()
def render_view(args, model, ray_sampler, projector, gt_img=None, render_stride=1, ret_alpha=False,
                single_net=True, sharded=False, feats_out_3d=True, shard_kw=None):
    """-> (ret, ray_batch): `ret` as render_single_image returns it plus, per level, 'sems'
    [1,H,W,num_classes+1] (reference :201-206) and 'labels' [H,W] int64 (loss.py:296's argmax).
    shard_kw: extra keyword arguments for render_image_sharded (group, gather_weights; render_fn in tests)."""
    model.switch_to_eval()
    ray_batch = ray_sampler.get_all()
    src = ray_batch["src_rgbs"].squeeze(0).permute(0, 3, 1, 2)
    ref_coarse_feats, _, ref_deep = model.feature_net(src)
    ref_deep_semantics = model.feature_fpn(ref_deep)
    device = ref_deep_semantics.device
    kw = dict(chunk_size=args.chunk_size, N_samples=args.N_samples, inv_uniform=args.inv_uniform, det=True,
              N_importance=args.N_importance, white_bkgd=args.white_bkgd, render_stride=render_stride,
              featmaps=ref_coarse_feats, deep_semantics=ref_deep_semantics, ret_alpha=ret_alpha,
              single_net=single_net)
    if sharded:
        ret = render_image_sharded(ray_sampler, ray_batch, model, projector, **kw, **(shard_kw or {}))
    else:
        ret = render_single_image(ray_sampler=ray_sampler, ray_batch=ray_batch, model=model,
                                  projector=projector, feats_out_3d=feats_out_3d, **kw)
    for lvl in ("outputs_coarse", "outputs_fine"):
        if ret[lvl] is None:
            continue
        feats = ret[lvl]["feats_out"]
        # reference :201-203: the head works on the half-resolution grid.  (The reference tests
        # args.render_stride there; every call site passes render_stride=args.render_stride, eval_gpnerf.py:128,
        # so the two agree -- the parameter is what the image was rendered with.)
        if render_stride == 1:
            feats = feats[::2, ::2, :]
        logits = model.sem_seg_head(feats.permute(2, 0, 1).unsqueeze(0).to(device), None, None)
        ret[lvl]["sems"] = logits.permute(0, 2, 3, 1)
        ret[lvl]["labels"] = semantic_labels(logits)
    return ret, ray_batch
