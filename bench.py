#!/usr/bin/env python
"""Benchmark of the GP-NeRF ray-rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], named in config.workload): Replica-shaped synthetic scene,
640x480 image, 10 source views (RGB [10,480,640,3], features [10,32,120,160], deep semantics
[10,512,240,320]), random-init GNT (netwidth 64, depth 8).  One step = one `render_rays` call on a
chunk of 4096 rays with the reference's eval settings (48 coarse + 16 importance samples -> a 48-sample
pass and a 64-sample pass, inv_uniform, det, single_net, ret_alpha, feats_out_3d materialised), all
inputs resident in HBM, outputs left on the device.  With N>1 every rank renders its own 4096-ray
chunk from its row shard of the image (weak scaling) and the rgb/depth/feats_out tiles of both
levels are all-gathered over RCCL inside the timed region.

Arithmetic: the default mode is "bf16x3" -- every fp32 product of the GNT linear layers is evaluated
as three bf16 MFMAs on split operands with fp32 accumulation (include/gpnerf_hip.h); it meets the
same 1e-4 parity bar as the exact-fp32 mode (tests/test_gpu_parity.py runs both).  The exact mode
(fp32 MFMA, reference operation order) is timed too and reported under "fp32_exact".

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_view_attn), which is
HBM-bound: algorithmic bytes per launch / mean launch duration measured with HIP events inside the
timed steps, against the 8 TB/s HBM peak; `traffic` = measured HBM bytes per launch (PMC passes,
profiles/); `roofline.mfma` carries the same launch against the MFMA peak of the mode (SURVEY.md 8(d)
reference-algorithm GEMM flops).  `cpu_baseline` is the CPU oracle (a port of the reference path) on a
bounded ray sample.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = int(os.environ.get("GPN_BENCH_CHUNK", "4096"))  # rays per step per GPU (4096: the north-star chunk)
N_SAMPLES, N_IMPORTANCE, VIEWS = 48, 16, 10
FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense; a bf16x3 product costs three MFMA flops per algorithmic flop
HBM_PEAK_GBPS = 8000.0
# what a bare copy kernel reaches on this GPU for k_view_attn's access shape (random 256-byte rows,
# read + written; tools/micro/reqbench.hip, profiles/r01_reqbench.txt); sequential read-only: 6250
RANDOM_ROW_RW_CEILING_GBPS = 4700.0


def view_attn_flops_per_point(V):
    """Algorithmic flops (MAC = 2) of the attention half of one view-transformer layer per ray
    sample: per view k_fc + v_fc (2 x 8192), pos_fc (4->8->64: 1088), attn_fc (64->8->64: 2048);
    per sample q_fc + out_fc (2 x 8192).  (SURVEY.md 8(d) minus the deep-semantic MAC and the FF,
    which run in other kernels here.)"""
    return V * (2 * 8192 + 1088 + 2048) + 2 * 8192


# MFMA flops the kernel executes (MAC = 2; every bf16x3 product counts its three MFMAs).
# bf16x3 (folded weights): per computed pair (Wv Wk) 8192 + pos_fc[2] 1024 + (A0 Wk | A0 P2) 1152 +
# attn_fc[2] 1024 = 11392, x3; + fp32 pos_fc[0] 64.  Per point q_fc + out_fc + A0 q = 17408, x3.
# fp32: the reference order, k 8192 + v 8192 + pos 64 + 1024 + attn 1024 + 1024 per pair, 16384 per point.
VIEW_EXEC_FLOPS_PER_PAIR = {"bf16x3": 3 * 11392 + 64, "fp32": 19520}
VIEW_EXEC_FLOPS_PER_POINT = {"bf16x3": 3 * 17408, "fp32": 16384}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 as the
    gfx950 correction + WRITE_SIZE, separate runs of this same command): profiles/r01_pmc_traffic.json.
    None if the file is missing."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def visibility_stats(scene, rb, ret, projector):
    """Fractions of (sample, view) pairs the view transformer has to compute / that are visible, over
    the two passes of one step (coarse samples + the fine samples derived from the coarse weights)."""
    from gp_nerf_amd import render_ray as rr
    sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
    pts_c, z_c = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], N_SAMPLES,
                                            inv_uniform=True, det=True)
    pts_f, _ = rr.sample_fine_pts(True, N_IMPORTANCE, True, N_SAMPLES, rb,
                                  ret["outputs_coarse"]["weights"], z_c)
    vis_pairs = comp_pairs = pts = 0
    for p in (pts_c, pts_f):
        _, _, mask = projector.compute_fused(p, sv)
        m = mask.reshape(-1, VIEWS) != 0
        nvis = m.sum(-1)
        vis_pairs += int(nvis.sum())
        comp_pairs += int(torch.where(nvis == 0, torch.full_like(nvis, VIEWS), nvis).sum())
        pts += m.shape[0]
    return {"visible_pair_fraction": vis_pairs / (pts * VIEWS),
            "computed_pair_fraction": comp_pairs / (pts * VIEWS)}


def build(dev, seed=1234):
    from gp_nerf_amd import synth
    from gp_nerf_amd.projection import Projector
    from gp_nerf_amd.transformer_network import GNT
    import numpy as np

    H, W = 480, 640
    K = synth.REPLICA_K
    g = torch.Generator(device=dev).manual_seed(seed)
    scene = types.SimpleNamespace()
    scene.H, scene.W = H, W
    scene.camera = torch.from_numpy(synth.make_camera(H, W, K, np.eye(4)))[None].to(dev)
    scene.src_cameras = torch.from_numpy(synth.source_rig(VIEWS, H, W, K))[None].to(dev)
    scene.depth_range = torch.tensor([[0.1, 10.0]], device=dev)
    scene.src_rgbs = torch.rand((1, VIEWS, H, W, 3), device=dev, generator=g)
    scene.featmaps = torch.randn((VIEWS, 32, H // 4, W // 4), device=dev, generator=g)
    scene.deep = torch.randn((VIEWS, 512, H // 2, W // 2), device=dev, generator=g)
    scene.rays_o, scene.rays_d = synth.rays_single_image(H, W, K, np.eye(4), 1)
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False,
                                 semantic_model="fpn", num_classes=20)
    net = GNT(args, 32, 63, 63, ret_alpha=True)
    net.load_state_dict(synth.gnt_state_dict())
    net = net.to(dev).eval()
    return scene, net, Projector(dev)


def ray_chunk(scene, rank, world, dev):
    """4096 rays from the middle of this rank's row shard (rank 0 of 1: pixel (H/2)*W onwards)."""
    from gp_nerf_amd.render_image import shard_rows
    rows, parts = shard_rows(scene.H, world)
    r0, r1 = parts[rank]
    start = ((r0 + r1) // 2) * scene.W
    start = min(start, scene.H * scene.W - CHUNK)
    sl = slice(start, start + CHUNK)
    return {"ray_o": scene.rays_o[sl].to(dev), "ray_d": scene.rays_d[sl].to(dev),
            "depth_range": scene.depth_range, "camera": scene.camera,
            "src_rgbs": scene.src_rgbs, "src_cameras": scene.src_cameras}


def cpu_baseline(scene, rb, n_rays):
    """The CPU oracle (port of the reference path) on the first `n_rays` rays of the same chunk."""
    import oracle
    from gp_nerf_amd import synth
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = synth.gnt_state_dict()
    rbc = {k: v.cpu() for k, v in rb.items()}
    rbc["ray_o"], rbc["ray_d"] = rbc["ray_o"][:n_rays], rbc["ray_d"][:n_rays]
    feat, deep = scene.featmaps.cpu(), scene.deep.cpu()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = oracle.render.render_rays(rbc, sd, feat, deep, N_samples=N_SAMPLES, inv_uniform=True,
                                        N_importance=N_IMPORTANCE, det=True, ret_alpha=True)
    dt = time.perf_counter() - t0
    return out, {"value": n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
                 "sample": "%d rays of the same chunk, full two-pass (48+16) render_rays, torch-CPU "
                           "fp32 oracle, %.1f s" % (n_rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-rays", type=int, default=1536, help="rays of the CPU baseline sample (0: skip)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-stage time table to stderr")
    ap.add_argument("--precision", default="bf16x3", choices=("fp32", "bf16x3"),
                    help="arithmetic of the GNT linear layers (see include/gpnerf_hip.h)")
    ap.add_argument("--only-timed", action="store_true",
                    help="skip the other arithmetic mode, the full-image run and the CPU baseline "
                         "(tools/profile_round.sh: the profiled kernels are then the timed workload only)")
    a = ap.parse_args()
    if a.only_timed:
        a.cpu_rays = 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per "
                     "GPU); WORLD_SIZE is 1" % a.gpus)
        a.gpus = world
    import torch.distributed as dist
    # rehearsal aid for a one-GPU box (tools/rehearse_ranks.sh): every rank on cuda:0 over gloo -- RCCL
    # refuses two ranks on one device.  Never set by the driver; numbers from such a run mean nothing.
    rehearse = os.environ.get("GPN_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)  # a hang names its line
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from gp_nerf_amd import _lib
    from gp_nerf_amd.render_ray import render_rays
    _lib.lib()
    scene, net, projector = build(dev)
    net.precision = a.precision
    model = types.SimpleNamespace(net_coarse=net, net_fine=None)
    rb = ray_chunk(scene, rank, world, dev)
    tile_w = 2 * (3 + 1 + 512)
    gathered = torch.empty((world * CHUNK, tile_w), device=dev) if world > 1 else None

    def step(collective=True):
        ret = render_rays(rb, model, scene.featmaps, scene.deep, projector, N_samples=N_SAMPLES,
                          inv_uniform=True, N_importance=N_IMPORTANCE, det=True, ret_alpha=True,
                          single_net=True)
        if world > 1 and collective:  # rgb / depth / semantic tiles of both levels, one collective
            c, f = ret["outputs_coarse"], ret["outputs_fine"]
            tile = torch.cat([c["rgb"], c["depth"][:, None], c["feats_out"],
                              f["rgb"], f["depth"][:, None], f["feats_out"]], dim=1)
            if rehearse:  # gloo stages device tensors through the host unreliably: gather host copies
                host = torch.empty((world * CHUNK, tile_w))
                dist.all_gather_into_tensor(host, tile.cpu())
                gathered.copy_(host)
            else:
                dist.all_gather_into_tensor(gathered, tile)
        return ret

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(precision, steps, warmup):
        """W warm-up steps, then K steps between fences; max over ranks; k_view_attn event-timed."""
        net.precision = precision
        for _ in range(warmup):
            r = step()
        fence()
        t0 = time.perf_counter()
        with _lib.stage_timing(stages=("view_attn",), max_records=32 * steps + 64) as tm_:
            for _ in range(steps):
                r = step()
            fence()
            dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return r, dt, tm_

    ret, elapsed, tm = timed(a.precision, a.steps, a.warmup)

    # dominant kernel: k_view_attn, 8 launches on N=4096*48 points + 8 on N=4096*64 per step.
    # It is HBM-bound (51 % of the 8 TB/s peak in measured traffic, 86 % of the
    # ceiling a bare copy kernel reaches for the same access shape -- random 256-byte rows read and
    # written; its time follows its bytes, not its instruction count), so the roofline is the HBM one:
    # achieved = ALGORITHMIC bytes per launch / mean launch time.
    launches = tm.launches["view_attn"]
    mean_ms = tm.ms["view_attn"] / max(1, launches)
    pts_per_launch = CHUNK * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) / 2.0
    vstat = visibility_stats(scene, rb, ret, projector)
    live = vstat["computed_pair_fraction"]
    # per computed (point, view) pair: feature row 256 B + ray_diff 16 B; per visible pair: the 8
    # hidden values of attn_fc 32 B; per point: q in 256 + q out 256 + softmax stats 256 + order /
    # visibility words 8
    alg_bytes = (VIEWS * live * (256 + 16) + VIEWS * vstat["visible_pair_fraction"] * 32 +
                 256 + 256 + 256 + 8) * pts_per_launch
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
    traffic = pmc_traffic("k_view_attn")
    roofline = {"bound": "hbm", "kernel": "k_view_attn", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "launches": launches, "mean_launch_ms": mean_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_GBps": (traffic / (mean_ms * 1e-3) / 1e9) if traffic else None,
                "access_shape_ceiling_GBps": RANDOM_ROW_RW_CEILING_GBPS,
                "visibility": vstat}
    # the same launch against the MFMA roofline (reference-algorithm GEMM flops, whether or not the
    # kernel executes them; executed flops alongside)
    flops_per_launch = view_attn_flops_per_point(VIEWS) * pts_per_launch
    if a.precision == "bf16x3":
        peak, peak_note = BF16_MATRIX_PEAK_TFLOPS / 3.0, "dense bf16 MFMA peak / 3 (three bf16 MFMAs per product)"
    else:
        peak, peak_note = FP32_MATRIX_PEAK_TFLOPS, "fp32 MFMA peak"
    ex_flops = (VIEWS * live * VIEW_EXEC_FLOPS_PER_PAIR[a.precision] +
                VIEW_EXEC_FLOPS_PER_POINT[a.precision]) * pts_per_launch
    roofline["mfma"] = {"achieved_TFLOPs": flops_per_launch / (mean_ms * 1e-3) / 1e12, "peak_TFLOPs": peak,
                        "frac": flops_per_launch / (mean_ms * 1e-3) / 1e12 / peak, "peak_note": peak_note,
                        "flops_per_launch": flops_per_launch,
                        "executed_mfma_TFLOPs": ex_flops / (mean_ms * 1e-3) / 1e12}

    other = None
    if not a.only_timed:  # the other arithmetic mode on the same workload (shorter run; every rank takes part)
        omode = "fp32" if a.precision == "bf16x3" else "bf16x3"
        _, dt_o, _ = timed(omode, max(3, a.steps // 2), 2)
        other = {"precision": omode, "value": world * CHUNK * max(3, a.steps // 2) / dt_o,
                 "ms_per_step": 1e3 * dt_o / max(3, a.steps // 2)}
        net.precision = a.precision

    stages = None
    if a.breakdown or rank == 0:
        with _lib.stage_timing() as tb:  # one extra, untimed step with every stage instrumented
            step(collective=False)  # rank-local: not every rank runs this block
            torch.cuda.synchronize()
        stages = {k: round(v, 4) for k, v in tb.ms.items() if tb.launches[k]}
        if a.breakdown and rank == 0:
            print("per-stage ms (one step):", json.dumps(stages), file=sys.stderr)

    # secondary metric: seconds per full image through render_single_image, results
    # on the host (BASELINE configs[1]: Replica 640x480, render_stride 2 as in gpnerf_replica.txt)
    full_image = None
    if rank == 0 and world == 1 and not a.only_timed:
        from gp_nerf_amd.render_image import render_single_image
        stride = 2
        idx = (torch.arange(0, scene.H, stride)[:, None] * scene.W +
               torch.arange(0, scene.W, stride)[None, :]).reshape(-1)
        rbi = dict(rb)
        rbi["ray_o"], rbi["ray_d"] = scene.rays_o[idx].to(dev), scene.rays_d[idx].to(dev)
        sampler = types.SimpleNamespace(H=scene.H, W=scene.W)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            render_single_image(sampler, rbi, model, projector, CHUNK, N_SAMPLES, inv_uniform=True,
                                N_importance=N_IMPORTANCE, det=True, render_stride=stride,
                                featmaps=scene.featmaps, deep_semantics=scene.deep, ret_alpha=True,
                                single_net=True, feats_out_3d=False)
            torch.cuda.synchronize()
            dt_i = time.perf_counter() - t0
            best = dt_i if best is None else min(best, dt_i)
        full_image = {"workload": "render_single_image, replica 640x480 render_stride 2 (%d rays), chunk %d, "
                                  "rgb/depth/weights/feats_out of both levels copied to the host; "
                                  "feats_out_3d not materialised" % (idx.numel(), CHUNK),
                      "seconds_per_image": best, "rays_per_sec": idx.numel() / best}

    cpu = None
    if rank == 0 and world == 1 and a.cpu_rays > 0:
        want, cpu = cpu_baseline(scene, rb, a.cpu_rays)
        # The 16 importance samples are an inverse-CDF lookup of the coarse weights, and visibility is
        # a hard in-frustum test: a 1e-7 change of the coarse weights can move a sample across a CDF
        # bin edge or across the border of a source image (reference behaviour, any two platforms
        # differ this way).  Rays whose fine depths or visibility masks differ between the GPU run
        # and the CPU run are counted and left out of the fine-level comparison.
        import oracle
        nr = a.cpu_rays
        rbc = {k: v.cpu() for k, v in rb.items()}
        rbc["ray_o"], rbc["ray_d"] = rbc["ray_o"][:nr], rbc["ray_d"][:nr]
        _, zc = oracle.sampling.sample_along_camera_ray(rbc["ray_o"], rbc["ray_d"], rbc["depth_range"],
                                                        N_SAMPLES, inv_uniform=True, det=True)
        zf = [oracle.sampling.sample_fine_pts(True, N_IMPORTANCE, True, N_SAMPLES, rbc, w.clone(), zc)[1]
              for w in (ret["outputs_coarse"]["weights"][:nr].cpu(), want["outputs_coarse"]["weights"])]
        same = ((zf[0] - zf[1]).abs().amax(-1) <= 1e-4 * zf[1].abs().amax(-1))
        # visibility of the fine samples: the GPU pipeline's own points vs the oracle's
        from gp_nerf_amd import render_ray as rr
        sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
        rbg = dict(rb)
        rbg["ray_o"], rbg["ray_d"] = rb["ray_o"][:nr], rb["ray_d"][:nr]
        pg, _ = rr.sample_fine_pts(True, N_IMPORTANCE, True, N_SAMPLES, rbg,
                                   ret["outputs_coarse"]["weights"][:nr], zc.to(dev))
        m_gpu = (projector.compute_fused(pg, sv)[2].reshape(nr, -1, VIEWS) != 0).cpu()
        pc, _ = oracle.sampling.sample_fine_pts(True, N_IMPORTANCE, True, N_SAMPLES, rbc,
                                                want["outputs_coarse"]["weights"].clone(), zc)
        pix, front = oracle.projector.compute_projections(pc, rbc["src_cameras"][0])
        m_cpu = (oracle.projector.inbound(pix, float(scene.H), float(scene.W)) & front).permute(1, 2, 0)
        same &= (m_gpu == m_cpu).reshape(nr, -1).all(dim=-1)
        worst = 0.0
        for lvl in ("outputs_coarse", "outputs_fine"):
            for k in ("rgb", "depth", "weights", "feats_out"):
                x, y = ret[lvl][k][:nr].cpu(), want[lvl][k]
                if lvl == "outputs_fine":
                    x, y = x[same], y[same]
                worst = max(worst, float((x - y).abs().max() / max(1.0, float(y.abs().max()))))
        cpu["max_rel_err_gpu_vs_cpu"] = worst
        cpu["fine_rays_resampled"] = int((~same).sum())

    if rank == 0:
        rays = world * CHUNK * a.steps
        line = {
            "metric": "rays_per_sec", "value": rays / elapsed, "unit": "rays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == "fp32" else "f32 (bf16x3 split-operand MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "replica_640x480_10views render_rays chunk=%d rays/GPU, "
                                   "coarse 48 + importance 16 (48- and 64-sample passes)" % CHUNK,
                       "precision": a.precision,
                       "rays_per_step_per_gpu": CHUNK, "samples": [N_SAMPLES, N_SAMPLES + N_IMPORTANCE],
                       "views": VIEWS, "parallelism": "ray-sharded x%d" % world},
            "roofline": roofline, "cpu_baseline": cpu,
            "fp32_exact" if a.precision == "bf16x3" else "bf16x3": other,
            "stage_ms_one_step": stages, "full_image": full_image,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
