"""How far the host runs ahead of the GPU: time to ENQUEUE one render_rays step (no synchronisation)
against the time the GPU needs for it (development aid)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd.render_ray import render_rays
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
rb = bench.ray_chunk(scene, 0, 1, dev)
for mode in ("fp32", "bf16x3"):
    net.precision = mode
    def step():
        return render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=48, inv_uniform=True,
                           N_importance=16, det=True, ret_alpha=True, single_net=True)
    for _ in range(3): step()
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # single step from an idle GPU: latency view
    lat = []
    for _ in range(5):
        torch.cuda.synchronize(); a = time.perf_counter(); step(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        lat.append((b - a, c - a))
    print("%s: enqueue %.2f ms/step, total %.2f ms/step; from idle: enqueue %.2f ms, done after %.2f ms"
          % (mode, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K, 1e3 * min(x[0] for x in lat), 1e3 * min(x[1] for x in lat)))
