"""Dump per-layer view logits of one point in both precision modes (bench chunk, fine pass)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd import render_ray as rr
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
rb = bench.ray_chunk(scene, 0, 1, dev)
ray, smp = int(sys.argv[1]), int(sys.argv[2])
net.precision = "fp32"
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
ret = rr.render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=48, inv_uniform=True,
                     N_importance=16, det=True, ret_alpha=True, single_net=True)
sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
pts_c, z_c = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], 48, inv_uniform=True, det=True)
pts, z = rr.sample_fine_pts(True, 16, True, 48, rb, ret["outputs_coarse"]["weights"], z_c)
rgb_feat, ray_diff, mask = projector.compute_fused(pts, sv)
R, S, V, D = 4096, 64, 10, 8
N = R * S
al = lambda b: (b + 255) & ~255
off_logits = al(N * 128 * 4) + al(N * V * 64 * 4) + al(N * 64 * 4)
off_stats = off_logits + al(N * V * D * 64 * 4)
n = ray * S + smp
print("mask", mask.reshape(N, V)[n].tolist())
print("ray_diff", ray_diff.reshape(N, V, 4)[n].tolist())
print("pts", pts[ray, smp].tolist(), "z", float(z[ray, smp]))
res = {}
for prec in ("fp32", "bf16x3"):
    net.precision = prec
    out = net.run(rgb_feat, ray_diff, mask, pts, rb["ray_d"], z_vals=z,
                  deep_map=dict(map=sv.deep, proj=sv.proj, img_h=sv.img_h, img_w=sv.img_w))
    torch.cuda.synchronize()
    ws = net._ws
    lg = ws[off_logits:off_logits + N * V * D * 64 * 4].view(torch.float32).reshape(N, V, D, 64)[n].clone()
    st = ws[off_stats:off_stats + N * D * 128 * 4].view(torch.float32).reshape(N, D, 2, 64)[n].clone()
    res[prec] = (lg, st, out[3][ray, smp].clone())
vis = mask.reshape(N, V)[n] != 0
for l in range(D):
    a, b = res["fp32"][0][vis][:, l], res["bf16x3"][0][vis][:, l]
    print("layer", l, "logit max|.| %.3e  diff %.3e | stats max diff %.3e rsum diff %.3e" % (
        float(a.abs().max()), float((a - b).abs().max()),
        float((res["fp32"][1][l, 0] - res["bf16x3"][1][l, 0]).abs().max()),
        float((res["fp32"][1][l, 1] - res["bf16x3"][1][l, 1]).abs().max())))
print("sem3d diff", float((res["fp32"][2] - res["bf16x3"][2]).abs().max()))
