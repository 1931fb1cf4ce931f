"""Full-image render timing (BASELINE configs[1]: Replica-shaped 640x480, 10 views, 48+16 samples) via
render_single_image, as a function of the chunk size.  feats_out_3d is not materialised on the host."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd.render_image import render_single_image
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, W = scene.H, scene.W
idx = (torch.arange(0, H, stride)[:, None] * W + torch.arange(0, W, stride)[None, :]).reshape(-1)
rb = {"ray_o": scene.rays_o[idx].to(dev), "ray_d": scene.rays_d[idx].to(dev), "depth_range": scene.depth_range,
      "camera": scene.camera, "src_rgbs": scene.src_rgbs, "src_cameras": scene.src_cameras}
sampler = types.SimpleNamespace(H=H, W=W)
for chunk in [int(c) for c in (sys.argv[2:] or ["4096", "8192", "16384"])]:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = render_single_image(sampler, rb, model, projector, chunk, 48, inv_uniform=True, N_importance=16, det=True,
                                  render_stride=stride, featmaps=scene.featmaps, deep_semantics=scene.deep, ret_alpha=True,
                                  single_net=True, feats_out_3d=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = idx.numel()
    print("stride %d chunk %6d: %d rays in %.3f s -> %.0f rays/s (outputs on the host, rgb %s)" % (stride, chunk, n, dt, n / dt, tuple(out["outputs_fine"]["rgb"].shape)), flush=True)
