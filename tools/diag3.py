"""Compare the two GPU precision modes with each other and with the CPU oracle on the bench chunk."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd.render_ray import render_rays
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
rb = bench.ray_chunk(scene, 0, 1, dev)
outs = {}
for prec in ("fp32", "bf16x3"):
    net.precision = prec
    outs[prec] = render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=48, inv_uniform=True,
                             N_importance=16, det=True, ret_alpha=True, single_net=True)
    torch.cuda.synchronize()
def rel(x, y):
    return float((x - y).abs().max() / max(1.0, float(y.abs().max())))
for lvl in ("outputs_coarse", "outputs_fine"):
    for k in ("rgb", "depth", "weights", "feats_out", "feats_out_3d"):
        a, b = outs["bf16x3"][lvl][k], outs["fp32"][lvl][k]
        d = (a - b).abs()
        worst = int(d.reshape(d.shape[0], -1).max(-1)[0].argmax())
        print(lvl, k, "bf16x3 vs fp32: %.3e  (worst ray %d)" % (rel(a, b), worst))
lo, hi = int(sys.argv[1]), int(sys.argv[2])
rb2 = dict(rb)
rb2["ray_o"], rb2["ray_d"] = rb["ray_o"][lo:hi], rb["ray_d"][lo:hi]
want, _ = bench.cpu_baseline(scene, rb2, hi - lo)
for prec in ("fp32", "bf16x3"):
    for lvl in ("outputs_coarse", "outputs_fine"):
        for k in ("rgb", "depth", "weights", "feats_out", "feats_out_3d"):
            a, b = outs[prec][lvl][k][lo:hi].cpu(), want[lvl][k]
            d = (a - b).abs().reshape(hi - lo, -1).max(-1)[0]
            print(prec, lvl, k, "vs cpu: %.3e (worst ray %d)" % (rel(a, b), lo + int(d.argmax())))
    f3 = outs[prec]["outputs_fine"]["feats_out_3d"][lo:hi].cpu()
    d = (f3 - want["outputs_fine"]["feats_out_3d"]).abs().amax(-1)
    r = int(d.amax(-1).argmax())
    print(prec, "fine f3d per-sample err of worst ray", lo + r, [round(float(x), 4) for x in d[r]])
