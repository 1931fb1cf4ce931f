"""Microbench of gpn_view_attn on the bench scene's own visibility pattern (64-sample pass).

Variants separate what bounds the kernel: with / without the logits+stats side outputs, sorted
(gpn_vis_order) / identity order, real masks / all views visible."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from gp_nerf_amd import _lib
from gp_nerf_amd import render_ray as rr

dev = torch.device("cuda:0")
scene, net, projector = bench.build(dev)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
net.precision = prec; PC = _lib.precision_code(prec)
W = net._weights()
L = _lib.lib()
st = _lib.stream()
rb = bench.ray_chunk(scene, 0, 1, dev)
S, V = 64, 10
pts, z = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], S, inv_uniform=True, det=True)
sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
rgb_feat, ray_diff, mask = projector.compute_fused(pts, sv)
N = pts.shape[0] * S
mask = mask.reshape(N, V).contiguous()
ray_diff = ray_diff.reshape(N, V, 4).contiguous()
feat = torch.empty(N, V, 64, device=dev)
q = torch.empty(N, 64, device=dev)
rgb_feat = rgb_feat.reshape(N, V, -1).contiguous()
_lib.check(L.gpn_rgbfeat(C.byref(W.rgbfeat0), C.byref(W.rgbfeat2), _lib.ptr(rgb_feat), N, V, 35, rgb_feat.shape[-1],
                         None, _lib.ptr(feat), _lib.ptr(q), PC, st), "rgbfeat")
qo = torch.empty_like(q)
logits = torch.empty(N, V, 8, 8, device=dev)  # hidden side channel [N,V,depth,8]
stats = torch.empty(N, 8, 64, device=dev)
perm = torch.empty(2 * N, dtype=torch.int32, device=dev)
scratch = torch.empty(L.gpn_vis_order_scratch_bytes(), dtype=torch.uint8, device=dev)
vl = W.view[0]
print("precision", prec, "N", N, "visible fraction %.3f" % float(mask.mean()))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def run(m, use_perm, side, ld=8 * 64, hld=8 * 8):
    if use_perm:
        _lib.check(L.gpn_vis_order(_lib.ptr(m), N, V, _lib.ptr(perm), _lib.ptr(scratch), st), "vis")
    p = _lib.ptr(perm) if use_perm else None
    lg = _lib.ptr(logits) if side else None
    sp = _lib.ptr(stats) if side else None
    return timeit(lambda: _lib.check(L.gpn_view_attn(C.byref(vl), _lib.ptr(q), _lib.ptr(feat), _lib.ptr(ray_diff),
                                                     _lib.ptr(m), p, N, V, _lib.ptr(qo), lg, hld, sp, ld, PC, st), "va"))


ones = torch.ones_like(mask)
if len(sys.argv) > 2 and sys.argv[2] == "one":  # profiling mode: the bench configuration only
    print("real masks, sorted, side outputs: %.1f us" % run(mask, True, True))
    sys.exit(0)
for name, m, up, side in (("real masks, sorted, side outputs", mask, True, True),
                          ("real masks, sorted, no side outputs", mask, True, False),
                          ("real masks, identity order, side outputs", mask, False, True),
                          ("all visible, sorted, side outputs", ones, True, True),
                          ("all visible, sorted, no side outputs", ones, True, False),
                          ("all visible, identity, no side outputs", ones, False, False)):
    pairs = float(m.sum())
    t = run(m, up, side)
    print("%-45s %8.1f us   %.1f ns per visible pair-tile(16)" % (name, t, t * 1e3 / (pairs / 16)), flush=True)

print("dense side outputs (ld=64/8), real masks sorted: %.1f us" % run(mask, True, True, 64, 8))
print("dense side outputs (ld=64/8), all visible sorted: %.1f us" % run(ones, True, True, 64, 8))
