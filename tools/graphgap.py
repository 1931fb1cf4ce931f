"""Does a HIP graph shorten the gap between dependent launches?  84 back-to-back launches of k_ff on one
tile-sized and one bench-sized input, enqueued one by one vs replayed from a captured graph."""
import ctypes as C
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_nerf_amd import _lib, synth
from gp_nerf_amd.transformer_network import GNT

dev = torch.device("cuda:0")
args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn", num_classes=20)
net = GNT(args, 32, 63, 63, ret_alpha=True); net.load_state_dict(synth.gnt_state_dict()); net = net.to(dev).eval()
L = _lib.lib()
for prec in ("fp32", "bf16x3"):
    net.precision = prec
    PC = _lib.precision_code(prec)
    W = net._weights()
    vl = W.view[0]
    for N in (128, 196608):
        q = torch.randn(N, 64, device=dev); qo = torch.empty_like(q)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            st = _lib.stream()

            def chain():
                for i in range(42):
                    L.gpn_ff(vl.ff_norm_w, vl.ff_norm_b, C.byref(vl.ff1), C.byref(vl.ff2), _lib.ptr(q), N, _lib.ptr(qo), PC, st)
                    L.gpn_ff(vl.ff_norm_w, vl.ff_norm_b, C.byref(vl.ff1), C.byref(vl.ff2), _lib.ptr(qo), N, _lib.ptr(q), PC, st)
            chain(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(5): chain()
            b.record(side); torch.cuda.synchronize()
            t_plain = a.elapsed_time(b) / 5 / 84 * 1e3
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                st = _lib.stream()
                chain()
            g.replay(); torch.cuda.synchronize()
            a.record(side)
            for _ in range(5): g.replay()
            b.record(side); torch.cuda.synchronize()
            t_graph = a.elapsed_time(b) / 5 / 84 * 1e3
        print("%-6s N=%6d: %.2f us per launch enqueued one by one, %.2f us replayed from a graph" % (prec, N, t_plain, t_graph), flush=True)
