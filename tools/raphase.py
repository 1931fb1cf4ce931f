"""Phase cycles of k_ray_attn per wave (a -DGPN_STAMP build): python tools/raphase.py <lib.so> [fp32|bf16x3] [S]"""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPN_LIB"] = os.path.abspath(sys.argv[1])
import torch
from gp_nerf_amd import _lib, synth
from gp_nerf_amd.transformer_network import GNT
dev = torch.device("cuda:0")
args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn", num_classes=20)
net = GNT(args, 32, 63, 63, ret_alpha=True); net.load_state_dict(synth.gnt_state_dict()); net = net.to(dev).eval()
net.precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
PC = _lib.precision_code(net.precision)
W = net._weights(); L = _lib.lib(); st = _lib.stream()
R = 4096
x = torch.randn(R, S, 64, device=dev); xo = torch.empty_like(x); ab = torch.empty(R, S, 4, device=dev)
for _ in range(30):
    L.gpn_ray_attn(C.byref(W.ray[0]), _lib.ptr(x), R, S, _lib.ptr(xo), _lib.ptr(ab), 4, None, PC, st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    L.gpn_ray_attn(C.byref(W.ray[0]), _lib.ptr(x), R, S, _lib.ptr(xo), _lib.ptr(ab), 4, None, PC, st)
b.record(); torch.cuda.synchronize()
print("%s S=%d: %.1f us per launch" % (net.precision, S, a.elapsed_time(b) / 50 * 1e3))
h = C.CDLL(os.environ["GPN_LIB"])
buf = (C.c_ulonglong * 128)()
h.gpn_debug_ra_phase(buf)
v = list(buf)
names = ["proj(P1)", "barrier A", "attn(P2)", "barrier B", "abar+loop", "iters"]
for w in range(8):
    row = v[w * 8: w * 8 + 6]
    it = max(1, row[5])
    print("wave %d: " % w + "  ".join("%s %6d" % (n, c // it) for n, c in zip(names[:5], row[:5])) + "  (cycles per iteration, %d iterations)" % row[5])
