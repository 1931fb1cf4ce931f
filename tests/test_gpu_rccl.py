"""The RCCL calls of the multi-GPU path on real hardware, with the one GPU a test box has: a one-rank
`nccl` process group (backend "nccl" is RCCL on ROCm) and `render_image_sharded` forced to issue its
collectives anyway (all_gather_into_tensor of the rgb / depth / feature tiles, gather of the feature
tiles to rank 0, broadcast of the 1-byte label image).  What this cannot show is the exchange between
GPUs (tests/test_cpu_host.py covers the N > 1 logic on gloo); what it does show is that the library
loads, a communicator comes up on this device, and the calls accept these shapes and dtypes."""
import os
import socket
import types

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Head:
    """labels(feats [1,512,h,w]) -> uint8 [2h,2w]: a deterministic function of the whole feature image
    through the product's own label kernel."""

    def labels(self, feats, dtype):
        from gp_nerf_amd.semantic_branch import semantic_labels
        lab = semantic_labels((feats[:, :21] - feats.mean()).contiguous(), dtype)
        return lab.repeat_interleave(2, 0).repeat_interleave(2, 1)


def test_sharded_render_through_rccl_one_rank():
    import torch.distributed as dist
    from gp_nerf_amd import render_image as ri
    from gp_nerf_amd.projection import Projector
    from gp_nerf_amd.transformer_network import GNT
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                device_id=dev)
    except Exception as e:  # no RCCL on this box: nothing to show
        pytest.skip("RCCL process group did not come up: %r" % (e,))
    os.environ["GPN_SHARD_FORCE_COLLECTIVES"] = "1"
    try:
        args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn",
                                     num_classes=20)
        net = GNT(args, in_feat_ch=32, posenc_dim=63, viewenc_dim=63, ret_alpha=True)
        net.load_state_dict(cases.weights(), strict=True)
        net = net.to(dev).eval()
        model = types.SimpleNamespace(net_coarse=net, net_fine=None)
        head = _Head()  # (the real head hard-codes a 120x160 input, reference semantic_branch.py:95)
        c = cases.g5_render_image()
        rb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c["ray_batch"].items()}
        sampler = types.SimpleNamespace(H=c["H"], W=c["W"])
        P = Projector(dev)
        kw = dict(inv_uniform=True, N_importance=8, det=True, render_stride=1, featmaps=c["featmaps"].to(dev),
                  deep_semantics=c["deep_semantics"].to(dev), ret_alpha=True, single_net=True)
        with_coll = ri.render_image_sharded(sampler, rb, model, P, 50, 16, gather_weights=True, sem_head=head,
                                            labels_only=True, **kw)
        torch.cuda.synchronize()
        os.environ["GPN_SHARD_FORCE_COLLECTIVES"] = "0"
        plain = ri.render_image_sharded(sampler, rb, model, P, 50, 16, gather_weights=True, sem_head=head,
                                        labels_only=True, **kw)
        for lvl in ("outputs_coarse", "outputs_fine"):
            assert set(with_coll[lvl]) == set(plain[lvl])
            for k in plain[lvl]:
                assert torch.equal(with_coll[lvl][k], plain[lvl][k]), (lvl, k)
            assert with_coll[lvl]["labels"].dtype == torch.uint8
    finally:
        os.environ.pop("GPN_SHARD_FORCE_COLLECTIVES", None)
        dist.destroy_process_group()
