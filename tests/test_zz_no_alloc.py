"""gpn_gnt_forward_static (the allocation-free entry point, include/gpnerf_hip.h) against
gpn_gnt_forward on a caller workspace: same inputs through the C-ABI, outputs compared bit for bit,
two calls in a row on the one static workspace, the second on a smaller input (stale contents of the
first call must not leak into it).  Collected last (test_zz_*): a failure here must not hide others."""
import ctypes as C
import types

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _net(dev, prec):
    from gp_nerf_amd.transformer_network import GNT
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False,
                                 semantic_model="fpn", num_classes=20)
    n = GNT(args, in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=True)
    n.load_state_dict(cases.weights(), strict=True)
    n = n.to(dev).eval()
    n.precision = prec
    return n


def _forward(L, _lib, W, c, static, dev):
    """One GNT forward through the C-ABI on the dense deep-semantic tensor; returns the four outputs."""
    R, S, V = c["rgb_feat"].shape[:3]
    f = lambda t: t.to(dev).float().contiguous()
    rgbf, deep, rd = f(c["rgb_feat"]), f(c["deep_sem"]), f(c["ray_diff"])
    mk, pts, ray_d = f(c["mask"]).reshape(R, S, V), f(c["pts"]), f(c["ray_d"])
    z = torch.linspace(0.5, 4.0, S, device=dev)[None].repeat(R, 1).contiguous()
    out, depth = torch.empty(R, 3 + S, device=dev), torch.empty(R, device=dev)
    semf, sem3d = torch.empty(R, 512, device=dev), torch.empty(R, S, 512, device=dev)
    sem = _lib.SemArgs()
    sem.deep_dense, sem.sem3d = _lib.ptr(deep), _lib.ptr(sem3d)
    common = (C.byref(W), _lib.ptr(rgbf), 35, _lib.ptr(rd), _lib.ptr(mk), _lib.ptr(pts), _lib.ptr(ray_d),
              _lib.ptr(z), R, S, V, C.byref(sem), _lib.ptr(out), _lib.ptr(depth), _lib.ptr(semf))
    if static:
        rc = L.gpn_gnt_forward_static(*common, _lib.stream())
    else:
        need = L.gpn_gnt_workspace_bytes(R, S, V, 8)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        rc = L.gpn_gnt_forward(*common, _lib.ptr(ws), need, _lib.stream())
    _lib.check(rc, "gpn_gnt_forward_static" if static else "gpn_gnt_forward")
    torch.cuda.synchronize()
    return out, depth, semf, sem3d


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_static_forward_matches_workspace_forward(prec):
    from gp_nerf_amd import _lib
    assert torch.cuda.is_available(), "needs a HIP device"
    dev = torch.device("cuda:0")
    L = _lib.lib()
    net = _net(dev, prec)
    W = net._weights()
    # first call: 40 rays x 64 samples x 10 views; second call, right after it, on a smaller input
    # (24 rays x 48 samples, another seed): every buffer of the second call overlays the first call's
    for S, R in ((64, 40), (48, 24)):
        c = cases.g2_gnt(S, R=R)
        want = _forward(L, _lib, W, c, False, dev)
        got = _forward(L, _lib, W, c, True, dev)
        for name, a, b in zip(("out", "depth", "sem_feat", "sem_feat_3d"), got, want):
            assert bool(torch.isfinite(a).all()), name
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), \
                "%s differs between the static and the caller workspace (S=%d R=%d, %s)" % (name, S, R, prec)
    # the entry point's defining property: it allocates nothing.  With every tensor made beforehand and one
    # warm call behind us, neither the device's free memory (hipMemGetInfo: sees a hipMalloc by anyone in this
    # process) nor torch's allocator statistics move across further calls.
    c = cases.g2_gnt(48, R=24)
    R, S, V = c["rgb_feat"].shape[:3]
    f = lambda t: t.to(dev).float().contiguous()
    rgbf, deep, rd = f(c["rgb_feat"]), f(c["deep_sem"]), f(c["ray_diff"])
    mk, pts, ray_d = f(c["mask"]).reshape(R, S, V), f(c["pts"]), f(c["ray_d"])
    out, semf, sem3d = torch.empty(R, 3 + S, device=dev), torch.empty(R, 512, device=dev), torch.empty(R, S, 512, device=dev)
    sem = _lib.SemArgs()
    sem.deep_dense, sem.sem3d = _lib.ptr(deep), _lib.ptr(sem3d)
    st = _lib.stream()
    call = lambda: L.gpn_gnt_forward_static(C.byref(W), _lib.ptr(rgbf), 35, _lib.ptr(rd), _lib.ptr(mk), _lib.ptr(pts),
                                            _lib.ptr(ray_d), None, R, S, V, C.byref(sem), _lib.ptr(out), None,
                                            _lib.ptr(semf), st)
    assert call() == 0
    torch.cuda.synchronize()
    free0, alloc0, resv0 = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0, "gpn_gnt_forward_static changed the device's free memory"
    assert torch.cuda.memory_allocated(dev) == alloc0 and torch.cuda.memory_reserved(dev) == resv0
    # a chunk beyond the static workspace is refused with a message, before any launch
    big_R = 4096
    assert L.gpn_gnt_workspace_bytes(big_R, 64, 10, 8) > _lib.STATIC_WS_BYTES
    d = torch.zeros(16, device=dev)
    rc = L.gpn_gnt_forward_static(C.byref(W), _lib.ptr(d), 35, _lib.ptr(d), _lib.ptr(d), _lib.ptr(d),
                                  _lib.ptr(d), None, big_R, 64, 10, None, _lib.ptr(d), None, None,
                                  _lib.stream())
    assert rc == -1 and b"GPN_STATIC_WS_BYTES" in L.gpn_last_error()
