"""Why does one fine sample differ between modes?  Compare fine depths / masks of one ray."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd import render_ray as rr
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
rb = bench.ray_chunk(scene, 0, 1, dev)
ray = int(sys.argv[1])
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
pts_c, z_c = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], 48, inv_uniform=True, det=True)
zs, ms, f3, wc = {}, {}, {}, {}
for prec in ("fp32", "bf16x3"):
    net.precision = prec
    ret = rr.render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=48, inv_uniform=True,
                         N_importance=16, det=True, ret_alpha=True, single_net=True)
    w = ret["outputs_coarse"]["weights"]
    pts, z = rr.sample_fine_pts(True, 16, True, 48, rb, w, z_c)
    zs[prec], wc[prec] = z[ray].cpu(), w[ray].cpu()
    ms[prec] = projector.compute_fused(pts, sv)[2].reshape(4096, 64, 10)[ray].cpu()
    f3[prec] = ret["outputs_fine"]["feats_out_3d"][ray].cpu()
print("coarse weight diff", float((wc["fp32"] - wc["bf16x3"]).abs().max()))
dz = (zs["fp32"] - zs["bf16x3"]).abs()
print("fine z max diff %.3e at %d; z there %s %s" % (float(dz.max()), int(dz.argmax()), float(zs["fp32"][dz.argmax()]), float(zs["bf16x3"][dz.argmax()])))
print("z[48:53] fp32  ", zs["fp32"][48:53].tolist())
print("z[48:53] bf16x3", zs["bf16x3"][48:53].tolist())
print("mask rows differ at samples", torch.nonzero((ms["fp32"] != ms["bf16x3"]).any(-1)).flatten().tolist())
d = (f3["fp32"] - f3["bf16x3"]).abs().amax(-1)
print("f3d per-sample diff > 1e-2 at", torch.nonzero(d > 1e-2).flatten().tolist(), [round(float(x), 3) for x in d[d > 1e-2]])
