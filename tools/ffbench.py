"""k_ff launch time (events around 50 back-to-back launches) for builds that differ in the store form."""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_nerf_amd import _lib, synth
from gp_nerf_amd.transformer_network import GNT
dev = torch.device("cuda:0")
args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn", num_classes=20)
net = GNT(args, 32, 63, 63, ret_alpha=True); net.load_state_dict(synth.gnt_state_dict()); net = net.to(dev).eval()
net.precision = "bf16x3"; PC = 1
W = net._weights(); st = _lib.stream()
for path in sys.argv[1:]:
    L = C.CDLL(path)
    for name, (res, at) in _lib._SIGS.items():
        if hasattr(L, name):
            getattr(L, name).restype, getattr(L, name).argtypes = res, at
    for N in [int(v) for v in os.environ.get('FF_NS', '196608,262144').split(',')]:
        q = torch.randn(N, 64, device=dev); qo = torch.empty_like(q)
        vl = W.view[0]
        fn = lambda: L.gpn_ff(vl.ff_norm_w, vl.ff_norm_b, C.byref(vl.ff1), C.byref(vl.ff2), _lib.ptr(q), N, _lib.ptr(qo), PC, st)
        for _ in range(20): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        a.record()
        t0 = time.perf_counter()
        for _ in range(200): fn()
        t1 = time.perf_counter()
        b.record(); torch.cuda.synchronize()
        print("%s N=%d: %.1f us per launch (host time to enqueue: %.1f us per launch)"
              % (os.path.basename(path), N, a.elapsed_time(b) / 200 * 1e3, (t1 - t0) / 200 * 1e6), flush=True)

    if hasattr(L, "gpn_debug_wave_times"):  # -DGPN_STAMP build: per-wave start / end of the LAST launch
        for _ in range(20): fn()
        torch.cuda.synchronize()
        import numpy as np
        buf = (C.c_ulonglong * 8192)()
        L.gpn_debug_wave_times(buf)
        t = np.array(list(buf), dtype=np.int64).reshape(2, 4096)[:, :2048] / 100.0
        t0 = t[0].min()
        st_, en = t[0] - t0, t[1] - t0
        print("  N=%d waves: start min/max %.2f/%.2f us; end min/median/max %.2f/%.2f/%.2f us" % (N, st_.min(), st_.max(), en.min(), np.median(en), en.max()))
        blk_end = en.reshape(256, 8).max(axis=1)
        print("  block end by XCD (blockIdx %% 8) max:", " ".join("%.1f" % blk_end[k::8].max() for k in range(8)))
        print("  block end by XCD (blockIdx %% 8) min:", " ".join("%.1f" % blk_end[k::8].min() for k in range(8)))
        print("  block start by XCD max:", " ".join("%.1f" % st_.reshape(256, 8).min(axis=1)[k::8].max() for k in range(8)))
        dur = en - st_
        print("  wave duration min/median/max %.2f/%.2f/%.2f us" % (dur.min(), np.median(dur), dur.max()))

    if hasattr(L, "gpn_debug_ffc"):  # k_ff32 region stamps (shader cycles) of waves 0..3 of block 7, second tile
        buf = (C.c_ulonglong * 128)()
        L.gpn_debug_ffc(buf)
        import numpy as np
        t = np.array(list(buf), dtype=np.int64).reshape(4, 32)
        for w in range(4):
            d = t[w, 1:18] - t[w, 0:17]
            print("  wave %d: y init %d | regions A/B per chunk: %s | tile total %d cycles"
                  % (w, d[0], " ".join("%d/%d" % (d[1 + 2 * c], d[2 + 2 * c]) for c in range(8)), t[w, 17] - t[w, 0]))
