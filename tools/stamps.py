"""In-kernel wall-clock stamps of a -DGPN_STAMP build (development aid): where the fixed cost of a
k_view_attn / k_ff launch goes.  Build: make -C gp-nerf_amd/csrc clean all EXTRA=-DGPN_STAMP"""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_nerf_amd import _lib, synth
from gp_nerf_amd.transformer_network import GNT
dev = torch.device("cuda:0")
args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn", num_classes=20)
net = GNT(args, 32, 63, 63, ret_alpha=True); net.load_state_dict(synth.gnt_state_dict()); net = net.to(dev).eval()
net.precision = "bf16x3"; PC = 1
W = net._weights(); L = _lib.lib(); st = _lib.stream()
h = C.CDLL(_lib.LIB_PATH)
V = 10
for N in (128, 32768, 65536):
    q = torch.randn(N, 64, device=dev); qo = torch.empty_like(q)
    feat = torch.randn(N, V, 64, device=dev); rd = torch.randn(N, V, 4, device=dev)
    mask = torch.ones(N, V, device=dev)
    vl = W.view[0]
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        L.gpn_view_attn(C.byref(vl), _lib.ptr(q), _lib.ptr(feat), _lib.ptr(rd), _lib.ptr(mask), None, N, V, _lib.ptr(qo), None, 64, None, 64, PC, st)
        b.record(); torch.cuda.synchronize()
        t_va = a.elapsed_time(b) * 1e3
        a.record()
        L.gpn_ff(vl.ff_norm_w, vl.ff_norm_b, C.byref(vl.ff1), C.byref(vl.ff2), _lib.ptr(q), N, _lib.ptr(qo), PC, st)
        b.record(); torch.cuda.synchronize()
        t_ff = a.elapsed_time(b) * 1e3
    buf = (C.c_ulonglong * 64)()
    h.gpn_debug_stamps(buf)
    s = list(buf)
    print("N=%6d view_attn %.1f us: staged(loads+stores) +%.2f, synced +%.2f, tile loop top +%.2f, views done +%.2f, end +%.2f us | ff %.1f us: stores done +%.2f synced +%.2f"
          % (N, t_va, (s[1]-s[0])/100, (s[2]-s[0])/100, (s[3]-s[0])/100, (s[4]-s[0])/100, (s[5]-s[0])/100, t_ff, (s[9]-s[8])/100, (s[10]-s[8])/100), flush=True)
