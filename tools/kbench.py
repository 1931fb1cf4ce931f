"""Kernel-level microbench: time per launch vs problem size (fixed overhead vs slope)."""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_nerf_amd import _lib, synth
from gp_nerf_amd.transformer_network import GNT
dev = torch.device("cuda:0")
args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn", num_classes=20)
net = GNT(args, 32, 63, 63, ret_alpha=True); net.load_state_dict(synth.gnt_state_dict()); net = net.to(dev).eval()
net.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
PC = _lib.precision_code(net.precision)
W = net._weights(); L = _lib.lib(); st = _lib.stream()
print("precision", net.precision)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
V = 10
for N in (128, 32768, 131072, 262144):
    q = torch.randn(N, 64, device=dev); qo = torch.empty_like(q)
    feat = torch.randn(N, V, 64, device=dev); rd = torch.randn(N, V, 4, device=dev)
    mask = torch.ones(N, V, device=dev); emb = torch.randn(N, 64, device=dev)
    vl = W.view[0]
    t_ff = timeit(lambda: L.gpn_ff(vl.ff_norm_w, vl.ff_norm_b, C.byref(vl.ff1), C.byref(vl.ff2), _lib.ptr(q), N, _lib.ptr(qo), PC, st))
    t_va = timeit(lambda: L.gpn_view_attn(C.byref(vl), _lib.ptr(q), _lib.ptr(feat), _lib.ptr(rd), _lib.ptr(mask), None, N, V, _lib.ptr(qo), None, 64, None, 64, PC, st))
    t_qp = timeit(lambda: L.gpn_qpos(C.byref(W.qfc0[0]), C.byref(W.qfc2[0]), _lib.ptr(q), _lib.ptr(emb), _lib.ptr(emb), N, 64, _lib.ptr(qo), PC, st))
    S = 64; R = max(1, N // S)
    x = torch.randn(R, S, 64, device=dev); xo = torch.empty_like(x); ab = torch.empty(R, S, 4, device=dev)
    t_ra = timeit(lambda: L.gpn_ray_attn(C.byref(W.ray[0]), _lib.ptr(x), R, S, _lib.ptr(xo), _lib.ptr(ab), 4, None, PC, st))
    print("N=%7d  ff %8.1f us  view_attn(dense,V=10) %8.1f us  qpos %8.1f us  ray_attn(S=64,R=%d) %8.1f us" % (N, t_ff, t_va, t_qp, R, t_ra), flush=True)


# optional: in-kernel cycle stamps of an instrumented build (tools only; -DGPN_STAMP)
import ctypes
h = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(h, "gpn_debug_stamps"):
    buf = (ctypes.c_ulonglong * 64)()
    h.gpn_debug_stamps(buf)
    st_ = list(buf)[:24]
    print("stamps:", [st_[i] - st_[0] for i in range(24)])
