"""Experiment: the bench's 4096-ray step issued alternately on two HIP streams (two chunks in flight: the drain of one
launch under the ramp of the other stream's) against one stream.  Usage: python tools/twostream.py [precision] [steps]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gp_nerf_amd.render_ray import render_rays
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
scene, net, projector = bench.build(dev)
net.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
model = types.SimpleNamespace(net_coarse=net, net_fine=None)
rbs = [bench.ray_chunk(scene, r, 2, dev) for r in range(2)]


def step(rb):
    return render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=48, inv_uniform=True,
                       N_importance=16, det=True, ret_alpha=True, single_net=True)


def run(nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    keep = []
    for w in range(4):
        with torch.cuda.stream(streams[w % nstreams]):
            keep.append(step(rbs[w % 2]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % nstreams]):
            keep.append(step(rbs[k % 2]))
        if len(keep) > 4:
            keep.pop(0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for n in (1, 2, 1, 2, 3):
    print("%s, %d stream(s): %.3f ms per 4096-ray step" % (net.precision, n, run(n)), flush=True)
