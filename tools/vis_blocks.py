"""How much view-attention work would a per-workgroup ray block cost?  
The view kernel computes a (tile, view) pair for every view that is live for SOME point of a 16-point
tile; gpn_vis_order sorts the points of the whole chunk by visibility pattern so that tiles are
homogeneous.  If every workgroup owned a block of B rays through all layers the sort could only be local
to the block.  This script counts (tile, view) pairs on the bench scene for: no sort, the global sort, and
local sorts over blocks of B rays (B = 8, 16, 32, 64), for both passes of a step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from gp_nerf_amd import render_ray as rr

dev = torch.device("cuda:0")
for rig in ("sparse", "dense"):
    scene, net, projector = bench.build(dev, rig=rig)
    rb = bench.ray_chunk(scene, 0, 1, dev)
    sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
    for S in (48, 64):
        pts, _ = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], S, inv_uniform=True, det=True)
        mask = projector.compute_fused(pts, sv)[2].reshape(pts.shape[0], S, -1) != 0  # [R,S,V]
        R, _, V = mask.shape
        key = (mask.long() << torch.arange(V, device=dev)).sum(-1)  # [R,S] visibility pattern
        allv = (1 << V) - 1
        eff = torch.where(key == 0, torch.full_like(key, allv), key)  # a point without a visible view needs all

        def pairs(keys_1d):
            n = keys_1d.numel()
            pad = (-n) % 16
            k = torch.cat([keys_1d, keys_1d[-1:].expand(pad)]) if pad else keys_1d
            t = k.reshape(-1, 16)
            u = t[:, 0]
            for i in range(1, 16):
                u = u | t[:, i]
            cnt = torch.zeros_like(u)
            for b in range(V):
                cnt += (u >> b) & 1
            return int(cnt.sum())

        def cost_sorted(keys_1d):  # order of gpn_vis_order: by pattern (any grouping of equal patterns does)
            return pairs(torch.sort(keys_1d)[0])
        flat = eff.reshape(-1)
        ideal = int(sum(((flat >> b) & 1).sum() for b in range(V))) / 16.0
        res = {"unsorted": pairs(flat), "global sort": cost_sorted(flat)}
        spread = {}
        for B in (8, 16, 32, 64):
            per = [cost_sorted(eff[r0:r0 + B].reshape(-1)) for r0 in range(0, R, B)]
            res["blocks of %d rays" % B] = sum(per)
            spread[B] = max(per) / (sum(per) / len(per))
        print("%s rig, S=%d: ideal (point-view pairs / 16) %.0f; (tile, view) pairs: %s" % (
            rig, S, ideal, ", ".join("%s %d (%.2fx)" % (k, v, v / res["global sort"]) for k, v in res.items())), flush=True)
        print("    busiest block / mean block (static ray ownership): %s" % ", ".join("B=%d %.2f" % kv for kv in spread.items()), flush=True)
