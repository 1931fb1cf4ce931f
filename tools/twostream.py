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


PRE = [torch.cuda.Stream() for _ in range(int(os.environ.get('PRE', '0')))]  # streams made before (queue mapping)


USED = []
for _ in range(int(os.environ.get('PREUSE', '0'))):  # streams made AND used before
    st_ = torch.cuda.Stream()
    with torch.cuda.stream(st_):
        torch.zeros(16, device=dev).add_(1)
    USED.append(st_)
torch.cuda.synchronize()


def run(nstreams):
    prio = os.environ.get('PRIO') == '1'  # second stream high priority: its own hardware-queue pool
    streams = [torch.cuda.Stream(priority=(-1 if prio and i % 2 else 0)) for i in range(nstreams)]
    print('   streams', [hex(s.cuda_stream) for s in streams])
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


if os.environ.get("PREIMG") == "1":  # an image through the product's chunk loop first (its side streams get used)
    from gp_nerf_amd.render_image import render_single_image
    idx = (torch.arange(0, scene.H, 2)[:, None] * scene.W + torch.arange(0, scene.W, 2)[None, :]).reshape(-1)
    rbi = dict(rbs[0])
    rbi["ray_o"], rbi["ray_d"] = scene.rays_o[idx].to(dev), scene.rays_d[idx].to(dev)
    render_single_image(types.SimpleNamespace(H=scene.H, W=scene.W), rbi, model, projector, 4096, 48, inv_uniform=True,
                        N_importance=16, det=True, render_stride=2, featmaps=scene.featmaps, deep_semantics=scene.deep,
                        ret_alpha=True, single_net=True, feats_out_3d=False)
    torch.cuda.synchronize()
    print("image rendered first")
if os.environ.get("PREBIG") == "1":  # a 16384-ray step first (bench leg chunk_16384), then empty_cache
    big = 16384
    rbb = dict(rbs[0])
    rbb["ray_o"], rbb["ray_d"] = scene.rays_o[:big].to(dev), scene.rays_d[:big].to(dev)
    step(rbb); step(rbb)
    torch.cuda.synchronize()
    del rbb
    torch.cuda.empty_cache()
    print("big step first")
for n in (1, 2, 2, 2, 2, 2, 2):
    print("%s, %d stream(s): %.3f ms per 4096-ray step" % (net.precision, n, run(n)), flush=True)
