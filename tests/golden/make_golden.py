"""Generate the golden vectors in this directory by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules gpnerf/{projection,render_ray,render_image,transformer_network}.py are
imported unmodified from /root/reference and run on CPU (torch CPU fp32, eval mode, det=True) on
seeded synthetic inputs built by gp_nerf_amd.synth (counter-based hash -> bit-identical anywhere).
Only OUTPUTS are stored (plus a SHA-256 of the regenerated inputs); tests rebuild the inputs from
the same seeds via `cases.py`.  The semantic head is not covered: its reference module needs
detectron2/fvcore (absent) -> parity unpinned (see oracle/sem_head.py).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from gpnerf.projection import Projector  # noqa: E402  (reference)
from gpnerf.render_image import render_single_image  # noqa: E402  (reference)
from gpnerf.render_ray import (raw2outputs, render_rays, sample_along_camera_ray,  # noqa: E402
                               sample_fine_pts)
from gpnerf.transformer_network import GNT  # noqa: E402  (reference)

sys.path.insert(0, HERE)
import cases  # noqa: E402


def ref_gnt(sd):
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False,
                                 semantic_model="fpn", num_classes=20)
    net = GNT(args, in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=True)
    net.load_state_dict(sd, strict=True)
    return net.eval()


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


@torch.no_grad()
def g6():
    """G6: the reference's own RaySamplerSingleImage (sample_ray.py:38-147), CPU part only (get_all()
    calls .cuda(); rays_o / rays_d / rgb are read from the object)."""
    from gpnerf.sample_ray import RaySamplerSingleImage  # reference
    arrays = {}
    c = cases.g6_sample_ray()
    for tag, cc in c.items():
        s = RaySamplerSingleImage(cases.g6_data(cc), "cpu", render_stride=int(cc["stride"]))
        d, o = np_(s.rays_d), np_(s.rays_o)
        if tag == "small":
            arrays.update({"small_rays_d": d, "small_rays_o": o, "small_rgb": np_(s.rgb)})
        else:  # 76,800 rays: every 97th ray + fp64 checksums
            arrays.update({"replica_rays_d_sub": d[::97].copy(), "replica_rays_o_sub": o[::97].copy(),
                           "replica_n": np.int64(d.shape[0]),
                           "replica_d_sum": d.astype(np.float64).sum(0),
                           "replica_d_abs_sum": np.abs(d.astype(np.float64)).sum(0)})
    save("g6_sample_ray", sha=cases.sha(c), **arrays)


@torch.no_grad()
def g7():
    """G7: the reference's own ResUNet (feature_network.py:164-313), eval mode, CPU.  (Its FPN needs
    mmcv / mmdet -- absent -- so the FPN restatement stays unpinned.)"""
    from gpnerf.feature_network import ResUNet  # reference
    c = cases.g7_encoders()
    net = ResUNet(coarse_out_ch=32, fine_out_ch=32, single_net=True)
    net.load_state_dict(c["resunet"], strict=True)
    net.eval()
    coarse, fine, feats = net(c["x"])
    assert fine is coarse
    arrays = {"coarse": np_(coarse)}
    for i, f in enumerate(feats):
        arrays["x%d_shape" % i] = np.array(f.shape)
        arrays["x%d_head" % i] = np_(f[:, :4])
        arrays["x%d_sum" % i] = np_(f.double().sum(1))
    save("g7_resunet", sha=cases.sha(dict(x=c["x"], resunet=c["resunet"])), **arrays)


@torch.no_grad()
def g8():
    """G8: BASELINE configs[2] at full size -- the reference's own render_rays (two passes, 48 + 16) on the
    19,200 rays of the half-resolution grid of a 240x320 ScanNet-shaped view, then the semantic head and the
    label image (eval_gpnerf.py:201-206, loss.py:296).  The head is the CPU restatement oracle/sem_head.py
    (the reference's module needs detectron2 / fvcore: absent), so the labels are pinned to the reference
    up to the head's input and to the restatement from there.  ~15 minutes on 8 cores."""
    import time
    import oracle
    c = cases.g8_scannet_labels()
    net = ref_gnt(cases.weights())
    model = types.SimpleNamespace(net_coarse=net, net_fine=None)
    proj = Projector("cpu")
    rb = c["ray_batch"]
    n = rb["ray_o"].shape[0]
    keep = ("rgb", "depth", "feats_out")
    outs = {lvl: {k: [] for k in keep} for lvl in ("outputs_coarse", "outputs_fine")}
    t0 = time.time()
    for i in range(0, n, 240):
        chunk = dict(rb)
        chunk["ray_o"], chunk["ray_d"] = rb["ray_o"][i:i + 240], rb["ray_d"][i:i + 240]
        ret = render_rays(chunk, model, c["featmaps"], c["deep_semantics"], proj, N_samples=48, inv_uniform=True,
                          N_importance=16, det=True, ret_alpha=True, single_net=True)
        for lvl in outs:
            for k in keep:
                outs[lvl][k].append(ret[lvl][k].clone())
        print("g8: %d / %d rays, %.0f s" % (i + 240, n, time.time() - t0), flush=True)
    arrays = {}
    h, w = c["H"] // 2, c["W"] // 2
    for lvl, t in (("outputs_coarse", "coarse"), ("outputs_fine", "fine")):
        o = {k: torch.cat(v, dim=0) for k, v in outs[lvl].items()}
        feats = o["feats_out"].reshape(h, w, 512).permute(2, 0, 1).unsqueeze(0)
        logits = oracle.sem_head.forward(c["head"], feats)  # [1,21,240,320]
        top2 = logits[0].topk(2, dim=0).values
        arrays.update({t + "_rgb": np_(o["rgb"]), t + "_depth": np_(o["depth"]),
                       t + "_feats_sub": np_(o["feats_out"][::37]),
                       t + "_feats_sum": np_(o["feats_out"].double().sum(-1)),
                       t + "_labels": np_(logits[0].argmax(dim=0)).astype(np.uint8),
                       t + "_margin": np_(top2[0] - top2[1]),
                       t + "_logits_sub": np_(logits[0, :, ::4, ::4]),
                       t + "_logits_absmax": np.float64(logits.abs().max().item())})
    # sha: every input incl. the rays (torch CPU inverse / bmm: may differ in the last bit on another CPU);
    # sha_maps: the hash-generated inputs only (bit-identical anywhere) -- what the GPU box checks
    save("g8_scannet_labels", sha=cases.sha(dict(rb=rb, f=c["featmaps"], d=c["deep_semantics"], h=c["head"])),
         sha_maps=cases.sha(dict(f=c["featmaps"], d=c["deep_semantics"], h=c["head"], s=rb["src_rgbs"],
                                 c=rb["src_cameras"])), **arrays)


@torch.no_grad()
def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        return g8()
    if len(sys.argv) > 1 and sys.argv[1] == "g6":
        return g6()
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        return g7()
    torch.manual_seed(0)
    proj = Projector("cpu")

    # ---- G1 projector
    c = cases.g1_projector()
    rgb_feat, deep, ray_diff, mask = proj.compute(c["pts"], c["camera"], c["src_rgbs"],
                                                  c["src_cameras"], c["featmaps"],
                                                  c["deep_semantics"])
    pix, in_front = proj.compute_projections(c["pts"], c["src_cameras"][0])
    save("g1_projector", sha=cases.sha(c), pix=np_(pix), in_front=np_(in_front),
         rgb_feat=np_(rgb_feat), ray_diff=np_(ray_diff), mask=np_(mask),
         deep_head=np_(deep[:8]), deep_sum=np_(deep.double().sum(dim=-1)))

    # ---- G2 GNT
    sd = cases.weights()
    net = ref_gnt(sd)
    for S in (48, 64):
        c = cases.g2_gnt(S)
        out, sem, sem3d = net(c["rgb_feat"], c["deep_sem"], c["ray_diff"], c["mask"], c["pts"],
                              c["ray_d"])
        extra = dict(sem3d=np_(sem3d)) if S == 48 else dict(sem3d_sum=np_(sem3d.double().sum(-1)))
        save("g2_gnt_s%d" % S, sha=cases.sha(c), out=np_(out), sem=np_(sem), **extra)

    # ---- G3 samplers
    c = cases.g3_samplers()
    arrays = {}
    for inv in (False, True):
        pts, z = sample_along_camera_ray(c["ray_o"], c["ray_d"], c["depth_range"], 48,
                                         inv_uniform=inv, det=True)
        fpts, fz = sample_fine_pts(inv, 16, True, 48, c, c["weights"].clone(), z)
        tag = "inv" if inv else "lin"
        arrays.update({"pts_" + tag: np_(pts), "z_" + tag: np_(z), "fpts_" + tag: np_(fpts),
                       "fz_" + tag: np_(fz)})
    r2o = raw2outputs(c["raw"], arrays_z := torch.from_numpy(arrays["z_lin"]), c["vis_mask"])
    arrays.update({"r2o_" + k: np_(v) for k, v in r2o.items()})
    r2o_w = raw2outputs(c["raw"], arrays_z, c["vis_mask"], white_bkgd=True)
    arrays["r2o_rgb_white"] = np_(r2o_w["rgb"])
    save("g3_samplers", sha=cases.sha(c), **arrays)

    # ---- G4 render_rays (two-pass, C1-shaped, reduced)
    c = cases.g4_render_rays()
    model = types.SimpleNamespace(net_coarse=net, net_fine=None)
    ret = render_rays(c["ray_batch"], model, c["featmaps"], c["deep_semantics"], proj,
                      N_samples=48, inv_uniform=True, N_importance=16, det=True, ret_alpha=True,
                      single_net=True)
    arrays = {}
    for lvl in ("outputs_coarse", "outputs_fine"):
        o = ret[lvl]
        t = lvl.split("_")[1]
        arrays.update({t + "_rgb": np_(o["rgb"]), t + "_weights": np_(o["weights"]),
                       t + "_depth": np_(o["depth"]), t + "_feats_out": np_(o["feats_out"]),
                       t + "_f3d_head": np_(o["feats_out_3d"][:4]),
                       t + "_f3d_sum": np_(o["feats_out_3d"].double().sum(-1))})
    save("g4_render_rays", sha=cases.sha(c), **arrays)

    # ---- G5 render_single_image, two chunk sizes must agree
    c = cases.g5_render_image()
    sampler = types.SimpleNamespace(H=c["H"], W=c["W"])
    outs = []
    for chunk in (50, 768):
        r = render_single_image(sampler, c["ray_batch"], model, proj, chunk, N_samples=16,
                                inv_uniform=True, N_importance=8, det=True, render_stride=1,
                                featmaps=c["featmaps"], deep_semantics=c["deep_semantics"],
                                ret_alpha=True, single_net=True)
        outs.append(r)
    arrays = {}
    for lvl in ("outputs_coarse", "outputs_fine"):
        t = lvl.split("_")[1]
        a, b = outs[0][lvl], outs[1][lvl]
        for k in ("rgb", "depth", "weights", "feats_out"):
            arrays[t + "_" + k] = np_(b[k])
            arrays[t + "_" + k + "_chunkdiff"] = np.float64((a[k] - b[k]).abs().max().item())
        arrays[t + "_f3d_shape"] = np.array(b["feats_out_3d"].shape)
        arrays[t + "_f3d_sum"] = np_(b["feats_out_3d"].double().reshape(c["H"], c["W"], -1, 512).sum(-1))
    save("g5_render_image", sha=cases.sha(c), **arrays)
    g6()
    g7()
    g8()


if __name__ == "__main__":
    main()
