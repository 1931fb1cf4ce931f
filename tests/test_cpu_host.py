"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the
product refuses to run without a HIP device (no CPU fallback), host logic (synthetic scenes, ray
sharding, the world_size-2 all-gather path over gloo)."""
import ctypes
import os
import re
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from gp_nerf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gpnerf_hip.h")).read()
    declared = set(re.findall(r"\b(gpn_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gpn_linear_t"}
    assert len(declared) >= 20
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(h, n)]
    assert not missing, "declared in include/gpnerf_hip.h but not exported: %s" % missing
    assert set(_lib.EXPORTS) == declared, "ctypes bindings out of sync with the header"
    assert _lib.lib().gpn_abi_version() == 3


def test_ctypes_struct_layout_matches_header_sizes():
    """sizeof of the ctypes mirrors must equal the C structs (checked by compiling a tiny probe)."""
    import subprocess
    import tempfile
    from gp_nerf_amd import _lib
    src = ('#include <stdio.h>\n#include "gpnerf_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
           "sizeof(gpn_linear_t),sizeof(gpn_view_layer_t),sizeof(gpn_ray_layer_t),"
           "sizeof(gpn_gnt_weights_t),sizeof(gpn_project_args_t),sizeof(gpn_sem_args_t));return 0;}")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True,
                                                check=True).stdout.split()]
    mirrors = [_lib.Linear, _lib.ViewLayer, _lib.RayLayer, _lib.GntWeights, _lib.ProjectArgs,
               _lib.SemArgs]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_argument_errors_come_back_as_messages():
    from gp_nerf_amd import _lib
    L = _lib.lib()
    rc = L.gpn_sample_along_ray(None, None, 1.0, 2.0, 4, 8, 0, None, None, None, None)
    assert rc == -1 and b"null pointer" in L.gpn_last_error()
    rc = L.gpn_ray_attn(None, None, 0, 48, None, None, 4, None, 0, None)
    assert rc == 0  # empty batch is a no-op
    # argument checks run before any HIP call, so they are testable without a GPU
    assert L.gpn_view_fold(None, None, None) == -1 and b"null pointer" in L.gpn_last_error()
    assert L.gpn_view_attn(None, None, None, None, None, None, 16, 10, None, None, 8, None, 64, 1, None) == -1
    assert b"null pointer" in L.gpn_last_error()
    # ABI 2: the arithmetic mode is an argument (no process-global switch); unknown modes are refused
    assert not hasattr(L, "gpn_set_precision")
    assert L.gpn_ff(None, None, None, None, None, 16, None, 7, None) == -1
    assert b"unknown precision 7" in L.gpn_last_error()
    with pytest.raises(ValueError):
        _lib.precision_code("fp16")
    # N = R*S (and N*V) must fit a 32-bit int: refused instead of wrapping into a bad grid
    W = _lib.GntWeights()
    for lin in (W.ray[0].q, W.ray[0].k, W.ray[0].v, W.ray[0].out):
        lin.w = 16  # non-null, aligned: the call must stop at the range check, before any launch
    W.ray[0].attn_norm_w = W.ray[0].attn_norm_b = 16
    assert L.gpn_ray_attn(ctypes.byref(W.ray[0]), ctypes.c_void_p(16), 1 << 26, 128, ctypes.c_void_p(16),
                          None, 4, None, 0, None) == -1
    assert b"does not fit a 32-bit int" in L.gpn_last_error()
    assert L.gpn_gnt_workspace_bytes(4096, 64, 10, 8) > 4096 * 64 * 10 * (8 * 8 + 64) * 4  # hidden side channel + feat
    assert L.gpn_gnt_workspace_bytes(0, 0, 0, 0) == 0
    with pytest.raises(RuntimeError, match="gpn_sample_along_ray failed"):
        _lib.check(L.gpn_sample_along_ray(None, None, 1.0, 2.0, 4, 8, 0, None, None, None, None),
                   "gpn_sample_along_ray")


def test_product_refuses_cpu_tensors():
    """No CPU fallback: every public op raises on CPU tensors instead of computing elsewhere."""
    import cases
    from gp_nerf_amd.projection import Projector
    from gp_nerf_amd import render_ray as rr
    from gp_nerf_amd.transformer_network import GNT
    c = cases.g1_projector()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Projector("cpu").compute(c["pts"], c["camera"], c["src_rgbs"], c["src_cameras"],
                                 c["featmaps"], c["deep_semantics"])
    s = cases.g3_samplers()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rr.sample_along_camera_ray(s["ray_o"], s["ray_d"], s["depth_range"], 16, det=True)
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False,
                                 semantic_model="fpn", num_classes=20)
    net = GNT(args, 32, 63, 63, ret_alpha=True)
    g = cases.g2_gnt(48)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(g["rgb_feat"], g["deep_sem"], g["ray_diff"], g["mask"], g["pts"], g["ray_d"])
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "gp-nerf_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", txt, re.M), f


def test_gnt_state_dict_contract():
    from gp_nerf_amd import synth
    from gp_nerf_amd.transformer_network import GNT
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False,
                                 semantic_model="fpn", num_classes=20)
    net = GNT(args, 32, 63, 63, ret_alpha=True)
    names = [n for n, _ in synth.gnt_param_shapes()]
    assert list(net.state_dict().keys()) == names and len(names) == 298
    assert sum(p.numel() for p in net.parameters()) == 1145347
    net.load_state_dict(synth.gnt_state_dict(), strict=True)


def test_synth_is_deterministic_and_shaped():
    from gp_nerf_amd import synth
    a = synth.hash_uniform(1, 2, (5, 7))
    assert np.array_equal(a, synth.hash_uniform(1, 2, (5, 7))) and a.min() >= 0 and a.max() < 1
    assert not np.array_equal(a, synth.hash_uniform(1, 3, (5, 7)))
    n = synth.hash_normal(3, 1, (20000,))
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1.0) < 0.03
    sc = synth.replica_scene(V=3, scale=0.1)
    assert sc.src_cameras.shape == (1, 3, 34) and sc.camera.shape == (1, 34)
    assert sc.featmaps.shape == (3, 32, 12, 16) and sc.deep_semantics.shape == (3, 512, 24, 32)
    assert float(sc.camera[0, 0]) == 48 and float(sc.camera[0, 1]) == 64
    # rays: d = c2w[:3,:3] K^-1 [u,v,1]  (reference sample_ray.py:116-132); identity pose
    K = sc.camera[0, 2:18].reshape(4, 4)
    u, v = 10.0, 7.0
    d = torch.inverse(K[:3, :3]) @ torch.tensor([u, v, 1.0])
    assert torch.allclose(sc.rays_d[7 * 64 + 10], d, atol=1e-6)
    assert torch.equal(sc.rays_o, torch.zeros_like(sc.rays_o))


def test_sem_head_oracle_shapes_and_determinism():
    """a15 is parity-unpinned (detectron2 absent): shape / linearity-free self-consistency only."""
    import oracle
    from gp_nerf_amd import synth
    sd = synth.sem_head_state_dict(num_classes=20)
    x = torch.from_numpy(synth.hash_normal(11, 1, (1, 512, 24, 32)))
    out = oracle.sem_head.forward(sd, x)
    assert out.shape == (1, 21, 48, 64) and torch.isfinite(out).all()
    assert torch.equal(out, oracle.sem_head.forward(sd, x))


# ------------------------------------------------------------------ multi-GPU path on gloo (world 2 and 4)
N_S, N_I = 8, 3


def _fake_render(ray_batch, model, projector, chunk_size, N_samples, inv_uniform, N_importance, det,
                 white_bkgd, featmaps, deep_semantics, ret_alpha, single_net, start, stop, f3d):
    """Stands in for the per-rank chunk renderer: outputs are pure functions of the ray index, with the
    structure _render_chunks returns (a rank without rays returns empty dicts; no fine level without
    importance samples)."""
    from collections import OrderedDict
    acc = OrderedDict([("outputs_coarse", OrderedDict()), ("outputs_fine", OrderedDict())])
    if N_importance <= 0:
        acc["outputs_fine"] = None
    for i in range(start, stop, chunk_size):
        j = min(i + chunk_size, stop)
        idx = torch.arange(i, j, dtype=torch.float32)
        for lvl, off, S in (("outputs_coarse", 0.0, N_samples), ("outputs_fine", 0.5, N_samples + N_importance)):
            if acc[lvl] is None:
                continue
            acc[lvl].setdefault("rgb", []).append(torch.stack([idx, idx + 1, idx + 2], 1) + off)
            acc[lvl].setdefault("depth", []).append(idx * 0.25 + off)
            acc[lvl].setdefault("feats_out", []).append(idx[:, None] + torch.arange(512.0)[None] + off)
            acc[lvl].setdefault("weights", []).append(idx[:, None] * torch.ones(1, S) + off)
    return acc


class _FakeHead:
    """labels(feats [1,512,h,w]) -> uint8 [2h,2w]: a deterministic function of the whole feature image
    (like the real head's GroupNorm, it needs all rows: the image mean enters every label)."""

    def labels(self, feats, dtype):
        m = feats.mean()
        lab = ((feats[0, :21] - m).argmax(dim=0)).to(dtype)
        return lab.repeat_interleave(2, 0).repeat_interleave(2, 1)


def _worker(rank, world, port, H, W, kw, q):
    import torch.distributed as dist
    from gp_nerf_amd.render_image import render_image_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sampler = types.SimpleNamespace(H=H, W=W)
    rb = {"ray_o": torch.zeros(H * W, 3)}
    kw = dict(kw)
    if kw.pop("head", False):
        kw["sem_head"] = _FakeHead()
    out = render_image_sharded(sampler, rb, None, None, 50, N_S, render_fn=_fake_render,
                               deep_semantics=torch.zeros(1), ret_alpha=True, **kw)
    q.put((rank, {l: (None if out[l] is None else {k: v.numpy().copy() for k, v in out[l].items()})
                  for l in out}))  # by value
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, H, W, kw):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _single(H, W, kw):
    from gp_nerf_amd.render_image import render_image_sharded
    kw = dict(kw)
    if kw.pop("head", False):
        kw["sem_head"] = _FakeHead()
    return render_image_sharded(types.SimpleNamespace(H=H, W=W), {"ray_o": torch.zeros(H * W, 3)}, None, None,
                                50, N_S, render_fn=_fake_render, deep_semantics=torch.zeros(1),
                                ret_alpha=True, **kw)


@pytest.mark.parametrize("H", [6, 7])  # 7 rows over 2 ranks: the last tile is padded
def test_sharded_render_world2_gloo(H):
    from gp_nerf_amd.render_image import shard_rows
    W = 5
    rows, parts = shard_rows(H, 2)
    assert parts[0][0] == 0 and parts[-1][1] == H and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    kw = dict(N_importance=N_I)
    res = _run_world(2, H, W, kw)
    one = _single(H, W, kw)  # single-process result of the same fake renderer = ground truth
    for r in (0, 1):
        for lvl in ("outputs_coarse", "outputs_fine"):
            assert set(res[r][lvl]) == {"rgb", "depth", "feats_out"}
            for k in res[r][lvl]:
                assert np.array_equal(res[r][lvl][k], one[lvl][k].numpy()), (r, lvl, k)
    assert one["outputs_fine"]["rgb"].shape == (H, W, 3)
    assert one["outputs_fine"]["feats_out"].shape == (H, W, 512)
    assert float(one["outputs_coarse"]["depth"][H - 1, W - 1]) == (H * W - 1) * 0.25


@pytest.mark.parametrize("kw", [dict(N_importance=N_I, gather_weights=True),
                                dict(N_importance=0),
                                dict(N_importance=N_I, head=True, labels_only=True),
                                dict(N_importance=N_I, head=True)],
                         ids=["weights", "no-fine-level", "labels-only", "labels-everywhere"])
def test_sharded_render_world4_rank_without_rows(kw):
    """H' = 5 over 4 ranks: rows per rank = 2, so rank 2 owns one row and rank 3 owns NONE.  The tile
    layout comes from the arguments (never from what a rank happened to render), so the empty rank
    joins every collective with the right shape; without importance samples no rank enters a
    fine-level collective.  labels-only: features go to rank 0 only, 1-byte labels are broadcast."""
    from gp_nerf_amd.render_image import shard_rows
    H, W = 5, 6
    rows, parts = shard_rows(H, 4)
    assert rows == 2 and parts[3] == (5, 5) and parts[2] == (4, 5)
    res = _run_world(4, H, W, kw)
    one = _single(H, W, kw)
    want_keys = {"rgb", "depth", "feats_out"} | ({"weights"} if kw.get("gather_weights") else set()) \
        | ({"labels"} if kw.get("head") else set())
    for r in range(4):
        if kw["N_importance"] == 0:
            assert res[r]["outputs_fine"] is None and one["outputs_fine"] is None
        for lvl in ("outputs_coarse", "outputs_fine"):
            if one[lvl] is None:
                continue
            keys = set(want_keys)
            if kw.get("labels_only") and r != 0:
                keys.discard("feats_out")  # features stay on rank 0 in labels-only mode
            assert set(res[r][lvl]) == keys, (r, lvl, set(res[r][lvl]))
            for k in keys:
                assert np.array_equal(res[r][lvl][k], one[lvl][k].numpy()), (r, lvl, k)
            if "labels" in keys:
                assert res[r][lvl]["labels"].dtype == np.uint8 and res[r][lvl]["labels"].shape == (6, 6)
    if kw.get("gather_weights"):
        assert one["outputs_fine"]["weights"].shape == (H, W, N_S + N_I)
        assert one["outputs_coarse"]["weights"].shape == (H, W, N_S)


# ------------------------------------------------------------------ render_view(sharded=True) on gloo, world 4
class _FakeViewModel:
    """Duck-typed GPNeRFModel for the host logic of render_view: CPU stand-ins for the encoders and the head
    (the head, like the real one, needs the whole feature image: its mean enters every logit)."""

    def switch_to_eval(self):
        pass

    def feature_net(self, src):
        return torch.zeros(src.shape[0], 32, 2, 2), None, [torch.zeros(1)]

    def feature_fpn(self, feats):
        return torch.zeros(1, 512, 2, 2)

    def sem_seg_head(self, feats, a, b):
        x = feats[0, :21] - feats.mean()
        return torch.nn.functional.interpolate(x[None], scale_factor=2, mode="nearest")


def _view_worker(rank, world, port, H, W, q):
    import torch.distributed as dist
    from gp_nerf_amd import eval_view
    eval_view.semantic_labels = lambda logits: logits[0].argmax(dim=0)  # the HIP epilogue refuses CPU tensors
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    args = types.SimpleNamespace(chunk_size=50, N_samples=N_S, inv_uniform=True, N_importance=N_I, white_bkgd=False,
                                 render_stride=1)
    sampler = types.SimpleNamespace(H=H, W=W, get_all=lambda: {"ray_o": torch.zeros(H * W, 3),
                                                               "src_rgbs": torch.zeros(1, 2, 4, 4, 3)})
    ret, _ = eval_view.render_view(args, _FakeViewModel(), sampler, None, render_stride=1, ret_alpha=True,
                                   sharded=True, shard_kw=dict(render_fn=_fake_render))
    q.put((rank, {l: {k: v.numpy().copy() for k, v in ret[l].items()} for l in ret}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_render_view_sharded_world4_equals_one_rank():
    """render_view(sharded=True) -- encoders, sharded render, head, label image -- in a 4-rank gloo job gives,
    on EVERY rank, the image a 1-rank call gives: rgb / depth / feats_out / sems / labels bit for bit
    (H' = 6 over 4 ranks: the last rank owns no rows).  Reference eval_gpnerf.py:172-206 on
    render_image.py:39-62's loop, sharded."""
    H, W = 6, 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_view_worker, args=(r, 4, port, H, W, q)) for r in range(4)]
    procs.append(ctx.Process(target=_view_worker, args=(9, 1, 0, H, W, q)))  # the 1-rank image
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(5))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = res[9]
    for r in range(4):
        for lvl in ("outputs_coarse", "outputs_fine"):
            assert set(res[r][lvl]) == {"rgb", "depth", "feats_out", "sems", "labels"}
            for k in res[r][lvl]:
                assert np.array_equal(res[r][lvl][k], one[lvl][k]), (r, lvl, k)
    assert one["outputs_fine"]["labels"].shape == (H, W) and one["outputs_fine"]["sems"].shape == (1, H, W, 21)
