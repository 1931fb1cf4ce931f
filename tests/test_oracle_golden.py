"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs on CPU (-m "not gpu")."""
import numpy as np
import torch

import cases
import oracle


def _same_inputs(case, g):
    assert np.array_equal(cases.sha(case), g["sha"]), "regenerated inputs differ from the golden run"


def _close(a, b, rtol=1e-5, atol=1e-6):
    """atol is relative to the tensor's scale (max |b|, at least 1): fp32 reassociation noise."""
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale)


def test_g1_projector_bit_exact(golden):
    g = golden("g1_projector")
    c = cases.g1_projector()
    _same_inputs(c, g)
    rgb_feat, deep, ray_diff, mask, pix = oracle.projector.compute(
        c["pts"], c["camera"], c["src_rgbs"], c["src_cameras"], c["featmaps"],
        c["deep_semantics"], return_pix=True)
    _, in_front = oracle.projector.compute_projections(c["pts"], c["src_cameras"][0])
    # masks / pixel coordinates / gathered values: bit-exact vs the reference on this machine
    assert np.array_equal(pix.numpy(), g["pix"])
    assert np.array_equal(in_front.numpy(), g["in_front"])
    assert np.array_equal(mask.numpy(), g["mask"])
    assert np.array_equal(rgb_feat.numpy(), g["rgb_feat"])
    assert np.array_equal(deep[:8].numpy(), g["deep_head"])
    assert np.array_equal(ray_diff.numpy(), g["ray_diff"])
    # the case really exercises the edge conditions
    m = g["mask"][..., 0]
    assert 0.05 < m.mean() < 0.95 and (~g["in_front"]).any()


def test_g2_gnt(golden):
    sd = cases.weights()
    for S in (48, 64):
        g = golden("g2_gnt_s%d" % S)
        c = cases.g2_gnt(S)
        _same_inputs(c, g)
        rec = {}
        out, sem, sem3d = oracle.gnt.forward(sd, c["rgb_feat"], c["deep_sem"], c["ray_diff"],
                                             c["mask"], c["pts"], c["ray_d"], intermediates=rec)
        _close(out, g["out"])
        _close(sem, g["sem"], rtol=1e-5, atol=1e-5)
        if S == 48:
            _close(sem3d, g["sem3d"], rtol=1e-5, atol=1e-5)
        else:
            _close(sem3d.double().sum(-1), g["sem3d_sum"], rtol=1e-5, atol=1e-4)
        assert torch.isfinite(out).all()
        _close(out[:, 3:].sum(-1), np.ones(out.shape[0], np.float32), atol=1e-5)
        # fused semantic aggregation == literal dataflow (what the HIP path relies on)
        W3d, W2d = oracle.gnt.semantic_weights(rec["attn2d"], rec["A"])
        sem_f, sem3d_f, pre = oracle.gnt.fused_semantics(sd, c["deep_sem"], W3d, W2d)
        _close(sem3d_f, sem3d.numpy(), rtol=1e-4, atol=1e-4)
        _close(pre, rec["pre_adaptor"].numpy(), rtol=1e-4, atol=1e-4)
        _close(sem_f, sem.numpy(), rtol=1e-4, atol=1e-4)
        # all-masked sample -> uniform attention over views
        a = rec["attn2d"][0][0, 5]
        _close(a, np.full(a.shape, 1.0 / a.shape[0], np.float32), atol=1e-6)


def test_g3_samplers(golden):
    g = golden("g3_samplers")
    c = cases.g3_samplers()
    _same_inputs(c, g)
    for inv, tag in ((False, "lin"), (True, "inv")):
        pts, z = oracle.sampling.sample_along_camera_ray(c["ray_o"], c["ray_d"], c["depth_range"],
                                                         48, inv_uniform=inv, det=True)
        assert np.array_equal(z.numpy(), g["z_" + tag])
        assert np.array_equal(pts.numpy(), g["pts_" + tag])
        w0 = c["weights"].clone()
        fpts, fz = oracle.sampling.sample_fine_pts(inv, 16, True, 48, c, c["weights"], z)
        assert torch.equal(w0, c["weights"]), "oracle must not mutate caller weights"
        assert np.array_equal(fz.numpy(), g["fz_" + tag])
        assert np.array_equal(fpts.numpy(), g["fpts_" + tag])
    z = torch.from_numpy(g["z_lin"])
    r = oracle.sampling.raw2outputs(c["raw"], z, c["vis_mask"])
    for k, v in r.items():
        assert np.array_equal(v.numpy(), g["r2o_" + k]), k
    rw = oracle.sampling.raw2outputs(c["raw"], z, c["vis_mask"], white_bkgd=True)
    assert np.array_equal(rw["rgb"].numpy(), g["r2o_rgb_white"])


def test_g4_render_rays(golden):
    g = golden("g4_render_rays")
    c = cases.g4_render_rays()
    _same_inputs(c, g)
    ret = oracle.render.render_rays(c["ray_batch"], cases.weights(), c["featmaps"],
                                    c["deep_semantics"], N_samples=48, inv_uniform=True,
                                    N_importance=16, det=True, ret_alpha=True)
    for lvl, t in (("outputs_coarse", "coarse"), ("outputs_fine", "fine")):
        o = ret[lvl]
        _close(o["rgb"], g[t + "_rgb"])
        _close(o["weights"], g[t + "_weights"])
        _close(o["depth"], g[t + "_depth"])
        _close(o["feats_out"], g[t + "_feats_out"], atol=1e-5)
        _close(o["feats_out_3d"][:4], g[t + "_f3d_head"], atol=1e-5)
        _close(o["feats_out_3d"].double().sum(-1), g[t + "_f3d_sum"], atol=1e-4)


def test_g5_render_single_image_chunk_invariance(golden):
    g = golden("g5_render_image")
    c = cases.g5_render_image()
    _same_inputs(c, g)
    outs = []
    for chunk in (50, 768):
        outs.append(oracle.render.render_single_image(
            c["H"], c["W"], c["ray_batch"], cases.weights(), chunk, N_samples=16,
            inv_uniform=True, N_importance=8, det=True, featmaps=c["featmaps"],
            deep_semantics=c["deep_semantics"], ret_alpha=True))
    for lvl, t in (("outputs_coarse", "coarse"), ("outputs_fine", "fine")):
        for k in ("rgb", "depth", "weights", "feats_out"):
            _close(outs[1][lvl][k], g[t + "_" + k], atol=1e-5)
            assert outs[1][lvl][k].shape == g[t + "_" + k].shape
            _close(outs[0][lvl][k], outs[1][lvl][k].numpy(), atol=1e-5)
            assert float(g[t + "_" + k + "_chunkdiff"]) < 1e-4
        f3 = outs[1][lvl]["feats_out_3d"]
        assert tuple(f3.shape) == tuple(g[t + "_f3d_shape"])
        _close(f3.double().reshape(c["H"], c["W"], -1, 512).sum(-1), g[t + "_f3d_sum"], atol=1e-4)


def test_g6_sample_ray_bit_exact(golden):
    """Ray generation (reference sample_ray.py:108-132): the restatement equals the reference's own
    RaySamplerSingleImage bit for bit."""
    g = golden("g6_sample_ray")
    c = cases.g6_sample_ray()
    _same_inputs(c, g)
    for tag, cc in c.items():
        cam = cc["camera"]
        H, W = int(cam[0, 0]), int(cam[0, 1])
        o, d = oracle.sample_ray.get_rays_single_image(H, W, cam[:, 2:18].reshape(-1, 4, 4),
                                                       cam[:, 18:34].reshape(-1, 4, 4), int(cc["stride"]))
        if tag == "small":
            assert np.array_equal(d.numpy(), g["small_rays_d"]) and np.array_equal(o.numpy(), g["small_rays_o"])
        else:
            assert d.shape[0] == int(g["replica_n"])
            assert np.array_equal(d.numpy()[::97], g["replica_rays_d_sub"])
            assert np.array_equal(o.numpy()[::97], g["replica_rays_o_sub"])
            assert np.array_equal(d.numpy().astype(np.float64).sum(0), g["replica_d_sum"])


def test_g7_resunet(golden):
    """2-D encoder restatement vs the reference's own ResUNet (same CPU conv library: 1e-5)."""
    g = golden("g7_resunet")
    c = cases.g7_encoders()
    assert np.array_equal(cases.sha(dict(x=c["x"], resunet=c["resunet"])), g["sha"])
    coarse, fine, feats = oracle.encoders.resunet_forward(c["resunet"], c["x"])
    _close(coarse, g["coarse"], rtol=1e-4, atol=1e-5)
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["x%d_shape" % i])
        _close(f[:, :4], g["x%d_head" % i], rtol=1e-4, atol=1e-5)
        _close(f.double().sum(1), g["x%d_sum" % i], rtol=1e-4, atol=1e-5)
    # FPN: parity unpinned (mmcv / mmdet absent); shape + determinism only
    out = oracle.encoders.fpn_forward(c["fpn"], feats)
    assert tuple(out.shape) == (3, 512, 26, 36) and torch.isfinite(out).all()


def test_oracle_on_the_full_size_scannet_view(golden):
    """Golden g8 (the reference's renderer on the 19,200 half-grid rays of a 240x320 ScanNet-shaped view,
    BASELINE.json configs[2]): the oracle on every 800th of those rays, two passes, against the reference's
    rgb / depth / feats_out; and the label image the golden holds is the argmax the restated head gives
    on the golden's own logits subsample."""
    g = golden("g8_scannet_labels")
    c = cases.g8_scannet_labels()
    assert np.array_equal(cases.sha(dict(rb=c["ray_batch"], f=c["featmaps"], d=c["deep_semantics"], h=c["head"])),
                          g["sha"])
    sel = torch.arange(0, 19200, 800)
    rb = dict(c["ray_batch"])
    rb["ray_o"], rb["ray_d"] = rb["ray_o"][sel].contiguous(), rb["ray_d"][sel].contiguous()
    out = oracle.render.render_rays(rb, cases.weights(), c["featmaps"], c["deep_semantics"], N_samples=48,
                                    inv_uniform=True, N_importance=16, det=True, ret_alpha=True)
    for lvl, t in (("outputs_coarse", "coarse"), ("outputs_fine", "fine")):
        for k in ("rgb", "depth"):
            want = torch.from_numpy(g[t + "_" + k])[sel]
            err = float((out[lvl][k] - want).abs().max() / max(1.0, float(want.abs().max())))
            assert err < (2e-5 if t == "coarse" else 1e-3), (lvl, k, err)
        want = torch.from_numpy(g[t + "_feats_sum"])[sel]
        err = float((out[lvl]["feats_out"].double().sum(-1) - want).abs().max() / max(1.0, float(want.abs().max())))
        assert err < (2e-5 if t == "coarse" else 1e-2), (lvl, err)
        lab = g[t + "_labels"]
        assert lab.shape == (240, 320) and lab.max() <= 20
        assert np.array_equal(g[t + "_logits_sub"].argmax(0), lab[::4, ::4])
        assert (g[t + "_margin"] >= 0).all()
