"""Where a full image's time goes in render_single_image: enqueue of the chunk loop, rendering, host copies."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gp_nerf_amd import render_image as ri
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
idx = (torch.arange(0, scene.H, 2)[:, None] * scene.W + torch.arange(0, scene.W, 2)[None, :]).reshape(-1)
rb = {"ray_o": scene.rays_o[idx].to(dev), "ray_d": scene.rays_d[idx].to(dev), "depth_range": scene.depth_range,
      "camera": scene.camera, "src_rgbs": scene.src_rgbs, "src_cameras": scene.src_cameras}
args = (rb, model, projector, 4096, 48, True, 16, True, False, scene.featmaps, scene.deep, True, True, 0, idx.numel(), False)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc = ri._render_chunks(*args, row_width=320)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = ri._merge(acc, 240, 320)
    t3 = time.perf_counter()
    print("after the loop : enqueue %.1f ms, chunks done %.1f ms, merge+copy %.1f ms, total %.1f" % ((t1-t0)*1e3, (t2-t0)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sink = ri._render_chunks(*args, row_width=320, sink=ri._HostSink(idx.numel(), dev))
    t1 = time.perf_counter()
    ev = [p[2] for p in sink.pending]
    out = sink.finish(240, 320)
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("behind the chunk: enqueue %.1f ms, finish returns %.1f ms, device idle %.1f ms" % ((t1-t0)*1e3, (t2-t0)*1e3, (t3-t0)*1e3))
