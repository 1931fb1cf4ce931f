"""Shader clock under the bench's sustained load (development aid, -DGPN_STAMP build): s_memtime
ticks per 100 MHz wall-clock tick inside the last k_ff / k_view_attn launch of a run of bench steps."""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd import _lib
from gp_nerf_amd.render_ray import render_rays
dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
rb = bench.ray_chunk(scene, 0, 1, dev)
h = C.CDLL(_lib.LIB_PATH)
for steps in (1, 30, 100):
    for _ in range(steps):
        render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=48, inv_uniform=True,
                    N_importance=16, det=True, ret_alpha=True, single_net=True)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    h.gpn_debug_stamps(buf)
    s = list(buf)
    for name, a, b in (("k_ff", 8, 11), ("k_view_attn", 0, 5)):
        wall = (s[b] - s[a]) / 100.0  # us
        cyc = s[32 + b] - s[32 + a]
        print("after %3d steps: %s block 0: %.1f us, %d shader cycles -> %.0f MHz" % (steps, name, wall, cyc, cyc / wall), flush=True)
        t0 = s[a]
        print("    start/end relative to block 0's start (us): last block %.2f / %.2f, block grid/2+3 %.2f / %.2f"
              % ((s[16 + a] - t0) / 100.0, (s[16 + b] - t0) / 100.0, (s[48 + a] - t0) / 100.0, (s[48 + b] - t0) / 100.0), flush=True)
