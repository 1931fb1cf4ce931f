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

Arithmetic.  The HEADLINE (`value`, `dtype`, `roofline`, `path_mfma`) is the exact-fp32 mode: fp32 MFMA,
fp32 accumulation, the reference's precision.  The faster "bf16x3" mode (every fp32 product as three
bf16 MFMAs on split operands, fp32 accumulation, ~2^-16 per product; include/gpnerf_hip.h) meets the same
1e-4 parity bar (tests/test_gpu_parity.py runs both) and is measured on the same workload under the key
"bf16x3", with its own roofline blocks.  `--precision bf16x3` swaps the two.

Prints ONE JSON line on stdout (rank 0), at most 4 KB -- the driver reads the tail of stdout -- with one
pricing per item; everything else (both pricings of every candidate kernel, the stage tables, the full block
of every leg) goes to stderr and to `bench_detail.json` (`detail_file`):
  roofline      the kernel with the largest share of the step in the headline mode.  fp32: an MFMA-bound
                kernel -> bound "mfma", achieved = ALGORITHMIC flops per launch (one flop per fp32 product the
                kernel's algorithm needs: computed pairs only, folded products) / mean launch time (HIP events
                inside the timed steps, on the launch stream), peak = fp32 matrix peak.  bf16x3: k_view_attn,
                HBM-bound -> algorithmic bytes per launch / launch time against 8 TB/s; `traffic` = HBM bytes
                per launch from the committed PMC passes under profiles/ (NOT measured in this run:
                `traffic_source` says which file).
  path_vs_reference_flops
                SURVEY.md 8(d) whole-path figure: F_ref (1.806 TFLOP per 4096-ray two-pass step, the
                reference algorithm's GEMM-class flops whether or not the kernels execute them) / step
                time, and its ratio to the MFMA peak of the mode.  NOT a roofline fraction: the kernels skip
                views with zero attention weight exactly and fold weight products, so the ratio exceeds 1 on
                the sparse rig; legs.dense_visibility is the rig where nothing is skipped.
  other_modes   the other arithmetic modes on the same workload (rays/s, step time, dominant-kernel fraction).
  legs          one or two numbers per leg: single_pass_64 (N_samples=64, N_importance=0: the
                literal "4096 rays x 64 samples x 10 views"), dense_visibility (5 cm / no-yaw rig, near
                plane 1 m: ~96 % of the (sample, view) pairs visible instead of ~41 %), projector_standalone
                (Projector.compute with the 512-channel gather materialised, B_proj of SURVEY 8(d)),
                stress_shape (BASELINE configs[4] on one GPU: 16 views, 1296x968 images, 96 + 32 samples),
                scannet_eval_view (BASELINE configs[2]: encoders + 76,800 rays + semantic head, seconds per view),
                chunk_16384 (the headline workload at four times the rays per step), full image seconds.
  cpu_baseline  the CPU oracle (a port of the reference path) on a bounded ray sample + max relative error of
                the GPU result against it.
  N > 1         rccl_world (an all-reduce of ones over the job's backend: the number of ranks RCCL saw),
                rank_ms_per_step (every rank's own clock), the collective inside the timed steps.
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
TRAFFIC_FILE = "profiles/r03_pmc_traffic.json"  # tools/profile_round.sh r03 + tools/profile_collect.py r03

RIGS = {
    # 15 cm / 5 cm offsets, 2 degrees of yaw per view, depth 0.1-10 m (mask mean ~0.41)
    "sparse": dict(rig=dict(), depth=(0.1, 10.0)),
    # the survey's tight rig: 5 cm spacing, no rotation; near plane at 1 m (mask mean ~0.96)
    "dense": dict(rig=dict(dx=0.05, dy=0.0, yaw_deg=0.0), depth=(1.0, 6.0)),
    # BASELINE.json configs[4] (stress shape: 16 views, 1296x968 ScanNet-sized images, 128 samples)
    "stress": dict(rig=dict(dx=0.06, dy=0.02, yaw_deg=-1.5), depth=(0.1, 6.0)),
}
STRESS = dict(views=16, hw=(968, 1296), n_samples=96, n_importance=32)


# ---------------------------------------------------------------------------------------- flop / byte models
def f_ref_per_sample(V, S):
    """reference-algorithm GEMM-class flops (MAC = 2) of one sample evaluation in a
    pass of S samples: rgbfeat + 8 x (view layer + ray layer) + 4 x q positional MLP."""
    return V * 12672 + 8 * (V * 20544 + 81920 + 98304 + 1280 * S) + 4 * 32512


def f_ref_per_ray(V, samples):
    """One render_rays call = one pass per entry of `samples`; + the 512x512 adaptor per pass."""
    return sum(S * f_ref_per_sample(V, S) + 524288 for S in samples)


def view_attn_flops_per_point(V):
    """Algorithmic flops (MAC = 2) of the attention half of one view-transformer layer per ray
    sample: per view k_fc + v_fc (2 x 8192), pos_fc (4->8->64: 1088), attn_fc (64->8->64: 2048);
    per sample q_fc + out_fc (2 x 8192).  (SURVEY.md 8(d) minus the deep-semantic MAC and the FF,
    which run in other kernels here.)"""
    return V * (2 * 8192 + 1088 + 2048) + 2 * 8192


FF_FLOPS_PER_POINT = 2 * 2 * 64 * 256  # fc1 + fc2 of one FeedForward (the FF share of 81,920 / 98,304)

# ALGORITHMIC flops of k_view_attn (MAC = 2; one flop per fp32 product, whatever the MFMA type; folded weights):
# per computed pair (Wv Wk) 8192 + pos_fc[2] 1024 + (A0 Wk | A0 P2) 1152 + attn_fc[2] 1024 = 11392, + fp32 pos_fc[0]
# 64; per point q_fc + out_fc + A0 q = 17408.  (A bf16x3 product issues three MFMAs per flop: `mult` in rooflines.)
VIEW_ALG_FLOPS_PER_PAIR = 11392 + 64
VIEW_ALG_FLOPS_PER_POINT = 17408
QPOS_FLOPS_PER_POINT = 2 * (192 * 64 + 64 * 64)  # q MLP on [q | emb_pt | emb_ray] (190 of the 192 inputs are real)


def mfma_peak(precision):
    if precision == "bf16":
        return BF16_MATRIX_PEAK_TFLOPS, "dense bf16 MFMA peak (one bf16 MFMA per product)"
    if precision == "bf16x3":
        return BF16_MATRIX_PEAK_TFLOPS / 3.0, "dense bf16 MFMA peak / 3 (three bf16 MFMAs per fp32 product)"
    return FP32_MATRIX_PEAK_TFLOPS, "fp32 MFMA peak"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 as the
    gfx950 correction + WRITE_SIZE, separate runs of this same command).  None if the file is missing."""
    for name in (TRAFFIC_FILE, "profiles/r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, name)) as f:
                return json.load(f)[kernel]["hbm_bytes_per_launch"], name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


# ---------------------------------------------------------------------------------------- scene
def build(dev, seed=1234, rig="sparse", views=VIEWS, hw=(480, 640), K=None):
    from gp_nerf_amd import synth
    from gp_nerf_amd.projection import Projector
    from gp_nerf_amd.transformer_network import GNT
    import numpy as np

    H, W = hw
    K = synth.REPLICA_K if K is None else K
    cfg = RIGS[rig]
    g = torch.Generator(device=dev).manual_seed(seed)
    scene = types.SimpleNamespace()
    scene.H, scene.W, scene.rig, scene.views = H, W, rig, views
    scene.camera = torch.from_numpy(synth.make_camera(H, W, K, np.eye(4)))[None].to(dev)
    scene.src_cameras = torch.from_numpy(synth.source_rig(views, H, W, K, **cfg["rig"]))[None].to(dev)
    scene.depth_range = torch.tensor([list(cfg["depth"])], device=dev)
    scene.src_rgbs = torch.rand((1, views, H, W, 3), device=dev, generator=g)
    scene.featmaps = torch.randn((views, 32, H // 4, W // 4), device=dev, generator=g)
    scene.deep = torch.randn((views, 512, H // 2, W // 2), device=dev, generator=g)
    scene.rays_o, scene.rays_d = synth.rays_single_image(H, W, K, np.eye(4), 1)
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False,
                                 semantic_model="fpn", num_classes=20)
    net = GNT(args, 32, 63, 63, ret_alpha=True)
    net.load_state_dict(synth.gnt_state_dict())
    net = net.to(dev).eval()
    return scene, net, Projector(dev)


def build_stress(dev):
    """BASELINE.json configs[4] scene: 16 views of 1296x968 (ScanNet-sized) images."""
    from gp_nerf_amd import synth
    H, W = STRESS["hw"]
    return build(dev, rig="stress", views=STRESS["views"], hw=(H, W),
                 K=synth.scaled_intrinsics(synth.SCANNET_K, W / 320.0, H / 240.0))


def rerig(scene, rig, dev):
    """The same maps / weights seen through another source rig (cameras and depth range only)."""
    from gp_nerf_amd import synth
    cfg = RIGS[rig]
    other = types.SimpleNamespace(**vars(scene))
    other.rig = rig
    other.src_cameras = torch.from_numpy(synth.source_rig(VIEWS, scene.H, scene.W, synth.REPLICA_K,
                                                          **cfg["rig"]))[None].to(dev)
    other.depth_range = torch.tensor([list(cfg["depth"])], device=dev)
    return other


def ray_chunk(scene, rank, world, dev):
    """4096 rays from the middle of this rank's row shard (rank 0 of 1: pixel (H/2)*W onwards)."""
    from gp_nerf_amd.render_image import shard_rows
    rows, parts = shard_rows(scene.H, world)
    r0, r1 = parts[rank]
    start = ((r0 + r1) // 2) * scene.W
    start = min(start, scene.H * scene.W - CHUNK)
    sl = slice(start, start + CHUNK)
    # pixel_row: the chunk is consecutive pixels of the image (render_single_image sets the same hint per chunk)
    return {"ray_o": scene.rays_o[sl].to(dev), "ray_d": scene.rays_d[sl].to(dev),
            "depth_range": scene.depth_range, "camera": scene.camera,
            "src_rgbs": scene.src_rgbs, "src_cameras": scene.src_cameras,
            "pixel_row": (scene.W, start % scene.W)}


def visibility_stats(scene, rb, projector, n_samples, n_importance, coarse_weights=None):
    """Fractions of (sample, view) pairs the view transformer has to compute / that are visible, over
    the passes of one step (coarse samples + the fine samples derived from the coarse weights)."""
    from gp_nerf_amd import render_ray as rr
    sv = projector.prepare(rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps, scene.deep)
    pts_c, z_c = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], n_samples,
                                            inv_uniform=True, det=True)
    passes = [pts_c]
    if n_importance > 0:
        passes.append(rr.sample_fine_pts(True, n_importance, True, n_samples, rb, coarse_weights, z_c)[0])
    vis_pairs = comp_pairs = pts = 0
    for p in passes:
        _, _, mask = projector.compute_fused(p, sv)
        m = mask.reshape(-1, VIEWS) != 0
        nvis = m.sum(-1)
        vis_pairs += int(nvis.sum())
        comp_pairs += int(torch.where(nvis == 0, torch.full_like(nvis, VIEWS), nvis).sum())
        pts += m.shape[0]
    return {"visible_pair_fraction": vis_pairs / (pts * VIEWS),
            "computed_pair_fraction": comp_pairs / (pts * VIEWS)}


def cpu_baseline(scene, rb, n_rays):
    """The CPU oracle (port of the reference path) on the first `n_rays` rays of the same chunk."""
    import oracle
    from gp_nerf_amd import synth
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = synth.gnt_state_dict()
    rbc = {k: v.cpu() for k, v in rb.items() if isinstance(v, torch.Tensor)}
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


# the legs of the default run (development aid: GPN_BENCH_LEGS=a,b runs a subset)
LEG_NAMES = tuple(n for n in os.environ.get(
    "GPN_BENCH_LEGS", "single_pass_64,dense_visibility,projector_standalone,stress_shape,scannet_eval_view,chunk_16384,"
    "chunks_in_flight_2").split(",") if n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-rays", type=int, default=1536, help="rays of the CPU baseline sample (0: skip)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-stage time table to stderr")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16x3", "bf16"),
                    help="arithmetic of the headline numbers (the other mode is reported under its own key)")
    ap.add_argument("--only-timed", action="store_true",
                    help="skip the other arithmetic mode, the extra legs, the full-image run and the CPU baseline "
                         "(tools/profile_round.sh: the profiled kernels are then the timed workload only)")
    ap.add_argument("--leg", default=None, choices=("single_pass_64", "dense_visibility", "projector_standalone", "stress_shape",
                                                 "scannet_eval_view", "chunk_16384", "chunks_in_flight_2"),
                    help="run ONLY this leg as the timed workload (profiling aid; prints the leg's block)")
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
    # development aid: GPN_BENCH_FORCE_DIST=1 runs the N>1 code path (RCCL process group, barrier,
    # all-reduce of the step time, all-gather of the tiles inside the timed steps) in a ONE-rank job, so that
    # those calls can be exercised on a one-GPU box.  Never set by the driver.
    multi = world > 1 or os.environ.get("GPN_BENCH_FORCE_DIST") == "1"
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from gp_nerf_amd import _lib
    from gp_nerf_amd.render_ray import render_rays
    _lib.lib()
    scene, net, projector = build(dev)
    model = types.SimpleNamespace(net_coarse=net, net_fine=None)
    rb = ray_chunk(scene, rank, world, dev)
    tile_w = 2 * (3 + 1 + 512)
    gathered = torch.empty((world * CHUNK, tile_w), device=dev) if multi else None

    def make_step(sc, batch, n_samples, n_importance):
        def step(collective=True):
            ret = render_rays(batch, model, sc.featmaps, sc.deep, projector, N_samples=n_samples,
                              inv_uniform=True, N_importance=n_importance, det=True, ret_alpha=True,
                              single_net=True)
            if multi and collective:  # rgb / depth / semantic tiles of both levels, one collective
                c, f = ret["outputs_coarse"], ret["outputs_fine"] or ret["outputs_coarse"]
                tile = torch.cat([c["rgb"], c["depth"][:, None], c["feats_out"],
                                  f["rgb"], f["depth"][:, None], f["feats_out"]], dim=1)
                if rehearse:  # gloo stages device tensors through the host unreliably: gather host copies
                    host = torch.empty((world * CHUNK, tile_w))
                    dist.all_gather_into_tensor(host, tile.cpu())
                    gathered.copy_(host)
                else:
                    # asynchronous: RCCL runs the gather of step k on its own stream while the kernels of step
                    # k + 1 run on the launch stream (a chunk loop over an image pipelines the same way); the
                    # launch stream waits for it only before the next gather reuses `gathered`, or at the fence --
                    # every collective still completes INSIDE the timed region (fence() waits for the last one)
                    if pending:
                        pending.pop().wait()
                    pending.append(dist.all_gather_into_tensor(gathered, tile, async_op=True))
            return ret
        return step

    pending = []  # the all-gather in flight (at most one)

    def fence():
        while pending:
            pending.pop().wait()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    CANDIDATES = ("view_attn", "ray_attn")  # the two launches of a layer: the candidates for the dominant kernel

    def timed(step, precision, steps, warmup, stages):
        """W warm-up steps, then K steps between fences; max over ranks.  The launches of `stages` (the
        dominant kernel, chosen from an instrumented step BEFORE the timed region) are event-timed on the
        launch stream inside the timed steps.  Only that one kernel: an event pair around every launch of
        both candidates costs 0.3 ms per step (measured: 7.26 -> 6.97 ms bf16x3 without any), one stage
        0.07-0.2 ms."""
        net.precision = precision
        for _ in range(warmup):
            r = step()
        fence()
        t0 = time.perf_counter()
        with _lib.stage_timing(stages=stages, max_records=96 * steps + 64) as tm_:
            for _ in range(steps):
                r = step()
            fence()
            dt = time.perf_counter() - t0
        timed.rank_dt = dt  # this rank's own clock; `dt` below is the maximum over the ranks
        if multi:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return r, dt, tm_

    def stage_breakdown(step, precision, full=False):
        net.precision = precision
        with _lib.stage_timing() as tb:  # one extra, untimed step with every stage instrumented
            step(collective=False)  # rank-local: not every rank runs this block
            torch.cuda.synchronize()
        table = {k: round(v, 4) for k, v in tb.ms.items() if tb.launches[k]}
        return (table, tb) if full else table

    def rooflines(precision, tm, dt, steps, samples, vstat, cal, dom_stage):
        """roofline of the dominant kernel of this mode + the whole-path MFMA figure.  `tm`: the event
        times of the dominant kernel inside the timed steps; the other candidate is priced from `cal`, the
        instrumented step that ran before the timed region (marked `timing_source`).

        The two candidates are the two launches of a layer: k_view_attn = view
        attention + the stage's feed-forward block (+ the q MLP on even layers) as phases of ONE launch,
        k_ray1 / k_ray_attn = ray attention + feed-forward block.  Algorithmic flops = one flop per fp32
        product the kernel's algorithm needs (folded weight products, computed pairs only); algorithmic
        bytes = every input / output row once per phase."""
        class _Both:  # per-stage totals: in-step events for the dominant stage, the calibration step otherwise
            ms = {k: (tm.ms[k] if k == dom_stage else cal.ms[k] * steps) for k in CANDIDATES}
            launches = {k: (tm.launches[k] if k == dom_stage else cal.launches[k] * steps) for k in CANDIDATES}
        tm = _Both
        pts_per_launch = CHUNK * sum(samples) / float(len(samples))
        s_mean = sum(samples) / float(len(samples))
        live = vstat["computed_pair_fraction"]
        peak, peak_note = mfma_peak(precision)
        mult = 3 if precision == "bf16x3" else 1  # MFMA flops issued per algorithmic flop (fp32, bf16: one)
        blocks = {}
        # ---- the view launch: attention (HBM-bound in bf16x3) + feed-forward + q MLP on every second layer
        n_va = max(1, tm.launches["view_attn"])
        va_ms = max(tm.ms["view_attn"] / n_va, 1e-9)
        # attention phase, per computed (point, view) pair: feature row 256 B + ray_diff 16 B; per visible pair:
        # the 8 hidden values of attn_fc 32 B; per point: q in 256 + q out 256 + order / visibility words 8
        # (+ 256 of softmax statistics: VIEW_STATS_BYTES); feed-forward phase: q in + out 512; q MLP phase
        # (half of the launches): q in + out + the point's encoding 768
        alg_bytes = (VIEWS * live * (256 + 16) + VIEWS * vstat["visible_pair_fraction"] * 32 +
                     256 + 256 + VIEW_STATS_BYTES + 8 + 512 + 0.5 * 768) * pts_per_launch
        traffic, tsrc = pmc_traffic("k_view_attn" if precision != "fp32" else "k_view_attn_fp32")
        va_alg = (VIEWS * live * VIEW_ALG_FLOPS_PER_PAIR + VIEW_ALG_FLOPS_PER_POINT + FF_FLOPS_PER_POINT +
                  0.5 * QPOS_FLOPS_PER_POINT) * pts_per_launch
        va_ref = (view_attn_flops_per_point(VIEWS) + FF_FLOPS_PER_POINT + 0.5 * 32512) * pts_per_launch
        va_common = {"kernel": "k_view_attn (attention + feed-forward + q MLP phases)", "launches": tm.launches["view_attn"],
                     "mean_launch_ms": va_ms, "traffic": traffic,
                     "traffic_source": ("%s (rocprofv3 PMC passes of this command; NOT measured in this run)" % tsrc)
                     if tsrc else None}
        blocks["view_attn"] = {
            "hbm": dict(va_common, bound="hbm", achieved=alg_bytes / (va_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBPS,
                        unit="GB/s", frac=alg_bytes / (va_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        algorithmic_bytes_per_launch=alg_bytes,
                        traffic_GBps=(traffic / (va_ms * 1e-3) / 1e9) if traffic else None),
            "mfma": dict(va_common, bound="mfma", achieved=va_alg / (va_ms * 1e-3) / 1e12, peak=peak, unit="TFLOP/s",
                         frac=va_alg / (va_ms * 1e-3) / 1e12 / peak, peak_note=peak_note,
                         algorithmic_flops_per_launch=va_alg, issued_mfma_TFLOPs=mult * va_alg / (va_ms * 1e-3) / 1e12,
                         reference_algorithm_TFLOPs=va_ref / (va_ms * 1e-3) / 1e12),
            "step_share": tm.ms["view_attn"] / steps / (1e3 * dt / steps)}
        # ---- the ray launch: attention over the samples of a ray + feed-forward
        n_ra = max(1, tm.launches["ray_attn"])
        ra_ms = max(tm.ms["ray_attn"] / n_ra, 1e-9)
        ra_alg = (4 * 8192 + 256 * s_mean + FF_FLOPS_PER_POINT) * pts_per_launch
        ra_bytes = (256 + 256 + 16 + 512) * pts_per_launch  # x in / out, mean attention row 16; feed-forward 512
        sfx = "" if precision != "fp32" else "_fp32"
        rt = [pmc_traffic(k + sfx)[0] for k in ("k_ray1_s48", "k_ray_attn_s64")]  # the two passes of a step
        rtraffic = (sum(rt) / len(rt)) if all(t is not None for t in rt) and len(samples) == 2 else None
        blocks["ray_attn"] = {
            "mfma": {"bound": "mfma", "kernel": "k_ray1 / k_ray_attn (attention + feed-forward phases)",
                     "achieved": ra_alg / (ra_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                     "frac": ra_alg / (ra_ms * 1e-3) / 1e12 / peak, "peak_note": peak_note, "traffic": rtraffic,
                     "launches": tm.launches["ray_attn"], "mean_launch_ms": ra_ms,
                     "algorithmic_flops_per_launch": ra_alg, "algorithmic_bytes_per_launch": ra_bytes,
                     "issued_mfma_TFLOPs": mult * ra_alg / (ra_ms * 1e-3) / 1e12},
            "step_share": tm.ms["ray_attn"] / steps / (1e3 * dt / steps)}
        # dominant = larger share of the step; priced against the roofline that bounds its largest phase in
        # this mode (bf16x3: the attention phase of k_view_attn is HBM-bound; fp32: MFMA)
        if dom_stage == "view_attn":
            main_ = dict(blocks["view_attn"]["hbm" if precision != "fp32" else "mfma"])
            main_["mfma" if precision != "fp32" else "hbm"] = blocks["view_attn"]["mfma" if precision != "fp32" else "hbm"]
        else:
            main_ = dict(blocks["ray_attn"]["mfma"])
        main_["step_share"] = blocks[dom_stage]["step_share"]
        main_["timing_source"] = "HIP events around every launch of this kernel inside the timed steps"
        main_["visibility"] = vstat
        other = "ray_attn" if dom_stage == "view_attn" else "view_attn"
        main_["second_kernel"] = dict(blocks[other]["mfma"], step_share=blocks[other]["step_share"],
                                      timing_source="HIP events in one instrumented step before the timed region "
                                                    "(scaled to the timed steps)")
        f_ref = f_ref_per_ray(VIEWS, samples) * CHUNK * world
        t_step = dt / steps
        path = {"F_ref_TFLOP_per_step": f_ref / 1e12, "achieved_TFLOPs": f_ref / t_step / 1e12,
                "peak_TFLOPs": peak * world, "frac": f_ref / t_step / 1e12 / (peak * world), "peak_note": peak_note,
                "note": "SURVEY.md 8(d): reference-algorithm GEMM-class flops per step / step time; the kernels "
                        "execute fewer (views with zero attention weight are skipped exactly: "
                        "computed_pair_fraction)"}
        return main_, path

    def measure(step, precision, steps, warmup, samples, sc, batch, n_s, n_i):
        net.precision = precision
        step(collective=False)  # (first touch of this shape / mode: workspace, folded weights)
        table, cal = stage_breakdown(step, precision, full=True)
        dom_stage = max(CANDIDATES, key=lambda k: cal.ms[k])
        ret, dt, tm = timed(step, precision, steps, warmup, (dom_stage,))
        cw = ret["outputs_coarse"]["weights"] if n_i > 0 else None
        vstat = visibility_stats(sc, batch, projector, n_s, n_i, cw)
        roof, path = rooflines(precision, tm, dt, steps, samples, vstat, cal, dom_stage)
        return ret, {"precision": precision, "value": world * CHUNK * steps / dt, "unit": "rays/s",
                     "ms_per_step": 1e3 * dt / steps, "rank_ms_per_step": 1e3 * timed.rank_dt / steps,
                     "roofline": roof, "path_mfma": path, "stage_ms_one_step": table}

    two_pass = make_step(scene, rb, N_SAMPLES, N_IMPORTANCE)
    samples2 = [N_SAMPLES, N_SAMPLES + N_IMPORTANCE]

    # ------------------------------------------------------------------ profiling aid: one leg only
    if a.leg is not None:
        print(json.dumps(run_leg(a.leg, a, dev, scene, rb, net, model, projector, make_step, measure,
                                 stage_breakdown, world, rank)))
        return

    ret, head = measure(two_pass, a.precision, a.steps, a.warmup, samples2, scene, rb, N_SAMPLES, N_IMPORTANCE)
    stages = head.pop("stage_ms_one_step")
    if a.breakdown and rank == 0:
        print("per-stage ms (one step, %s):" % a.precision, json.dumps(stages), file=sys.stderr)

    # the other arithmetic modes on the same workload (every rank takes part)
    omodes = [m for m in ("fp32", "bf16x3", "bf16") if m != a.precision]
    others, rets_o = {}, {}
    if not a.only_timed:
        for om in omodes:
            rets_o[om], others[om] = measure(two_pass, om, a.steps, 2, samples2, scene, rb, N_SAMPLES, N_IMPORTANCE)
        net.precision = a.precision

    legs = None
    if rank == 0 and world == 1 and not a.only_timed:
        legs = {name: run_leg(name, a, dev, scene, rb, net, model, projector, make_step, measure,
                              stage_breakdown, world, rank)
                for name in LEG_NAMES}
        net.precision = a.precision

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
        full_image = {"workload": "render_single_image, replica 640x480 render_stride 2 (%d rays), chunk %d, "
                                  "rgb/depth/weights/feats_out of both levels copied to the host; "
                                  "feats_out_3d not materialised" % (idx.numel(), CHUNK)}
        for mode in [a.precision] + omodes:
            net.precision = mode
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
            full_image[mode] = {"seconds_per_image": best, "rays_per_sec": idx.numel() / best}
        net.precision = a.precision

    cpu = None
    if rank == 0 and world == 1 and a.cpu_rays > 0:
        want, cpu = cpu_baseline(scene, rb, a.cpu_rays)
        for mode, r in [(a.precision, ret)] + list(rets_o.items()):
            if r is not None:
                err, resampled = compare_with_cpu(scene, rb, projector, r, want, a.cpu_rays, dev)
                cpu["max_rel_err_gpu_vs_cpu_" + mode] = err
                cpu["fine_rays_resampled_" + mode] = resampled

    rccl_world = None
    if multi:  # proof that the collective backend saw every rank: an all-reduce of ones (+ per-rank step times)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_world = int(ones.item())
        mine = torch.tensor([head["rank_ms_per_step"]], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [round(float(t.item()), 4) for t in every]
    else:
        rank_ms = None

    if rank == 0:
        detail = dict({"headline": dict(head, stage_ms_one_step=stages), "legs": legs,
                       "full_image": full_image, "cpu_baseline": cpu}, **others)
        legs_file = None if a.only_timed else write_detail(detail)  # (profiling runs keep the last full run's file)
        line = {
            "metric": "rays_per_sec", "value": head["value"], "unit": "rays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[a.precision], "headline_precision": a.precision, "data": "synthetic",
            "config": {"workload": "replica_640x480_10views render_rays chunk=%d rays/GPU, "
                                   "coarse 48 + importance 16 (48- and 64-sample passes)" % CHUNK,
                       "rays_per_step_per_gpu": CHUNK, "samples": samples2,
                       "views": VIEWS, "parallelism": "ray-sharded x%d" % world},
            "roofline": compact_roofline(head["roofline"]),
            "path_vs_reference_flops": compact_path(head["path_mfma"], head["roofline"]),
            "cpu_baseline": compact_cpu(cpu, a.precision),
            "other_modes": {m: compact_mode(o) for m, o in others.items()} or None,
            "legs": compact_legs(legs, full_image),
            "stage_ms_one_step": {k: round(v, 3) for k, v in stages.items()},
            "detail_file": legs_file,
        }
        if multi:
            line["rccl_world"] = rccl_world
            line["backend"] = dist.get_backend()
            line["rank_ms_per_step"] = rank_ms
            line["collective"] = ("all_gather_into_tensor of [rays, %d] fp32 tiles per step, asynchronous (overlaps the "
                                  "next step's kernels), all completed inside the timed region" % tile_w)
        text = json.dumps(round_floats(line))
        for drop in ("stage_ms_one_step", "legs", "other_modes", "path_vs_reference_flops"):  # never again an unparseable line
            if len(text) > 4000:
                line[drop] = "see " + str(legs_file)
                text = json.dumps(round_floats(line))
        sys.stdout.flush()
        print(text, flush=True)
    if multi:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------- the printed line
DTYPES = {"fp32": "f32", "bf16x3": "f32 (bf16x3 split-operand MFMA, fp32 accumulate)",
          "bf16": "bf16 (one bf16 MFMA per product, fp32 accumulate; softmax / LayerNorm / residuals fp32)"}
DETAIL_FILE = "bench_detail.json"


def round_floats(x, digits=5):
    """Floats to `digits` significant figures (the printed line must stay short)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: round_floats(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [round_floats(v, digits) for v in x]
    return x


def write_detail(detail):
    """Everything that used to ride on the printed line (per-mode roofline blocks with their alternative
    pricings, the stage tables, the full block of every leg) goes to a file beside bench.py (or the
    working directory) and to stderr; the printed line keeps one number per item.  -> the file's name."""
    text = json.dumps(round_floats(detail, 6), indent=1)
    print(text, file=sys.stderr)
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT, os.getcwd()):
        try:
            if not os.path.isdir(d):
                continue
            with open(os.path.join(d, DETAIL_FILE), "w") as f:
                f.write(text)
            return os.path.relpath(os.path.join(d, DETAIL_FILE), ROOT)
        except OSError:
            continue
    return None


def compact_roofline(r):
    """The contract's roofline object for the dominant kernel (one pricing: the roof that bounds it)."""
    keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "launches",
            "mean_launch_ms", "step_share", "timing_source", "algorithmic_flops_per_launch",
            "algorithmic_bytes_per_launch")
    out = {k: r[k] for k in keys if r.get(k) is not None}
    out.setdefault("traffic", None)
    if out.get("traffic_source"):
        out["traffic_source"] = out["traffic_source"].split(" (")[0] + " (committed PMC passes, not this run)"
    out["timing_source"] = "HIP events inside the timed steps"
    out["computed_pair_fraction"] = r["visibility"]["computed_pair_fraction"]
    sk = r.get("second_kernel")
    if sk:
        out["second_kernel"] = {"kernel": sk["kernel"], "bound": sk["bound"], "frac": sk["frac"],
                                "mean_launch_ms": sk["mean_launch_ms"], "step_share": sk["step_share"]}
    return out


def compact_path(p, r):
    """SURVEY 8(d)'s whole-path figure.  NOT a roofline fraction: the numerator is the REFERENCE algorithm's
    flop count, and the kernels skip views with zero attention weight exactly and use folded weight
    products, so the ratio to the MFMA peak exceeds 1 on scenes with few visible pairs."""
    return {"reference_TFLOP_per_step": p["F_ref_TFLOP_per_step"], "reference_TFLOPs_rate": p["achieved_TFLOPs"],
            "mfma_peak_TFLOPs": p["peak_TFLOPs"], "ratio_to_mfma_peak": p["frac"],
            "computed_pair_fraction": r["visibility"]["computed_pair_fraction"],
            "note": "reference-algorithm flops / step time; can exceed 1 (skipped pairs, folded products): "
                    "not a roofline fraction -- see legs.dense_visibility for the rig where nothing is skipped"}


def compact_cpu(cpu, precision):
    if not cpu:
        return None
    out = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample")}
    out["max_rel_err"] = cpu.get("max_rel_err_gpu_vs_cpu_" + precision)
    out["fine_rays_resampled"] = cpu.get("fine_rays_resampled_" + precision)
    out["max_rel_err_other_modes"] = {k[len("max_rel_err_gpu_vs_cpu_"):]: v for k, v in cpu.items()
                                      if k.startswith("max_rel_err_gpu_vs_cpu_") and not k.endswith("_" + precision)}
    return out


def compact_mode(m):
    """One arithmetic mode on the headline workload: rays/s, step time, the dominant kernel's roofline fraction."""
    if not m:
        return None
    r = m["roofline"]
    return {"value": m["value"], "ms_per_step": m["ms_per_step"], "roofline_kernel": r["kernel"].split(" (")[0],
            "roofline_bound": r["bound"], "roofline_frac": r["frac"], "mean_launch_ms": r["mean_launch_ms"]}


def compact_legs(legs, full_image):
    """One or two numbers per leg (rays/s unless the key says otherwise)."""
    if not legs:
        return None
    out = {}
    for name, leg in legs.items():
        if name == "projector_standalone":
            out[name] = {"ms_per_call": leg["ms_per_call"], "hbm_frac": leg["roofline"]["frac"],
                         "achieved_GBps": leg["roofline"]["achieved"]}
        elif name == "scannet_eval_view":
            out[name] = {m: leg[m]["seconds_per_view"] for m in leg if isinstance(leg[m], dict)}
            out[name]["unit"] = "s/view"
        elif name == "chunk_16384":
            out[name] = {m: leg[m]["rays_%d" % (4 * CHUNK)]["value"] for m in leg if isinstance(leg[m], dict)}
        else:
            out[name] = {m: leg[m]["value"] for m in leg if isinstance(leg[m], dict)}
            if name == "dense_visibility":
                out[name]["fp32_ratio_to_mfma_peak"] = leg["fp32"]["path_mfma"]["frac"]
                out[name]["fp32_kernel_frac"] = leg["fp32"]["roofline"]["frac"]
    if full_image:
        out["full_image_640x480_stride2"] = {m: full_image[m]["seconds_per_image"] for m in full_image
                                             if isinstance(full_image[m], dict)}
        out["full_image_640x480_stride2"]["unit"] = "s/image incl. copy to host"
    return out


# softmax statistics written per point by gpn_view_attn for the semantic pass (0 once that side
# channel is gone: gpn_sem_aggregate then rebuilds the log-sum-exp from the hidden values)
VIEW_STATS_BYTES = 256


def run_leg(name, a, dev, scene, rb, net, model, projector, make_step, measure, stage_breakdown, world, rank):
    """One of the SURVEY.md 8(d) legs beside the headline workload; both arithmetic modes where the
    GNT is involved.  Returns the leg's JSON block."""
    from gp_nerf_amd import _lib
    steps = max(5, a.steps // 2)
    if name == "single_pass_64":
        step = make_step(scene, rb, 64, 0)
        out = {"workload": "render_rays, N_samples=64, N_importance=0 (one 64-sample pass: 4096 rays x 64 samples "
                           "x 10 views), same scene and chunk"}
        for mode in ("fp32", "bf16x3"):
            _, m = measure(step, mode, steps, 2, [64], scene, rb, 64, 0)
            out[mode] = m
        return out
    if name == "dense_visibility":
        dense = rerig(scene, "dense", dev)
        rbd = dict(rb)
        rbd["depth_range"], rbd["src_cameras"] = dense.depth_range, dense.src_cameras
        step = make_step(dense, rbd, N_SAMPLES, N_IMPORTANCE)
        out = {"workload": "two-pass render_rays as the headline, source rig 5 cm apart / no rotation, depth "
                           "1-6 m: nearly every (sample, view) pair is visible (roofline.visibility)"}
        for mode in ("fp32", "bf16x3"):
            _, m = measure(step, mode, steps, 2, [N_SAMPLES, N_SAMPLES + N_IMPORTANCE], dense, rbd,
                           N_SAMPLES, N_IMPORTANCE)
            out[mode] = m
        return out
    if name == "projector_standalone":
        # north_star subsystem 1: Projector.compute with the 512-channel deep-semantic gather
        # materialised (reference projection.py:84-139) at the north-star shape
        from gp_nerf_amd import render_ray as rr
        S = 64
        pts, _ = rr.sample_along_camera_ray(rb["ray_o"], rb["ray_d"], rb["depth_range"], S, inv_uniform=True,
                                            det=True)
        R = pts.shape[0]
        H, W = scene.H, scene.W

        def call():
            return projector.compute(pts, rb["camera"], rb["src_rgbs"], rb["src_cameras"], scene.featmaps,
                                     scene.deep)
        for _ in range(2):
            out4 = call()
        torch.cuda.synchronize()
        mask_mean = float(out4[3].mean())
        del out4
        t0 = time.perf_counter()
        with _lib.stage_timing(stages=("project", "project_deep"), max_records=4 * steps + 8) as tm:
            for _ in range(steps):
                o = call()
                del o
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        # SURVEY.md 8(d) B_proj: outputs written once + every map byte read once + the points
        b_out = R * S * VIEWS * (35 + 512 + 4 + 1) * 4
        b_maps = VIEWS * 4 * (3 * H * W + 32 * (H // 4) * (W // 4) + 512 * (H // 2) * (W // 2))
        b_proj = b_out + b_maps + R * S * 12
        k_ms = {k: tm.ms[k] / max(1, tm.launches[k]) for k in ("project", "project_deep")}
        t_k = (k_ms["project"] + k_ms["project_deep"]) * 1e-3
        deep_bytes = R * S * VIEWS * 512 * 4
        traffic, tsrc = pmc_traffic("k_project_deep")
        return {"workload": "Projector.compute, R=%d S=%d V=%d, maps of a 640x480 image, deep-semantic gather "
                            "materialised ([R,S,V,512] = %.2f GB written)" % (R, S, VIEWS, deep_bytes / 1e9),
                "mask_mean": mask_mean, "ms_per_call": 1e3 * dt / steps, "rays_per_sec": R * steps / dt,
                "kernel_ms": {"k_project_feat": k_ms["project"], "k_project_deep": k_ms["project_deep"]},
                "roofline": {"bound": "hbm", "kernel": "k_project_feat + k_project_deep", "achieved": b_proj / t_k / 1e9,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_proj / t_k / 1e9 / HBM_PEAK_GBPS,
                             "algorithmic_bytes_per_call": b_proj, "traffic": traffic,
                             "traffic_source": ("%s (k_project_deep only; rocprofv3 PMC passes, NOT measured in this "
                                                "run)" % tsrc) if tsrc else None,
                             "k_project_deep_alone_GBps": (deep_bytes + VIEWS * 4 * 512 * (H // 2) * (W // 2)) /
                                                          (k_ms["project_deep"] * 1e-3) / 1e9}}
    if name == "stress_shape":
        # BASELINE.json configs[4] on ONE GPU: 16 source views, 1296x968 images (maps 324x242 / 648x484),
        # 96 coarse + 32 importance samples (a 96- and a 128-sample pass), one 4096-ray chunk per step
        from gp_nerf_amd.render_ray import render_rays
        V, (H, W) = STRESS["views"], STRESS["hw"]
        ns, ni = STRESS["n_samples"], STRESS["n_importance"]
        sc, _, proj = build_stress(dev)
        rbs = ray_chunk(sc, 0, 1, dev)

        def step(collective=False):
            return render_rays(rbs, model, sc.featmaps, sc.deep, proj, N_samples=ns, inv_uniform=True,
                               N_importance=ni, det=True, ret_alpha=True, single_net=True)
        out = {"workload": "BASELINE configs[4] on one GPU: two-pass render_rays, %d views, %dx%d images, %d + %d "
                           "samples (%d- and %d-sample passes), chunk %d rays"
                           % (V, W, H, ns, ni, ns, ns + ni, CHUNK)}
        f_ref = f_ref_per_ray(V, [ns, ns + ni]) * CHUNK
        for mode in ("fp32", "bf16x3", "bf16"):  # "bf16": configs[4] names "bf16 attention"
            net.precision = mode
            for _ in range(2):
                r = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            peak, note = mfma_peak(mode)
            fin = all(bool(torch.isfinite(r[lvl][k]).all()) for lvl in ("outputs_coarse", "outputs_fine")
                      for k in ("rgb", "depth", "weights", "feats_out"))
            out[mode] = {"precision": mode, "value": CHUNK / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt,
                         "finite": fin,
                         "path_mfma": {"F_ref_TFLOP_per_step": f_ref / 1e12, "achieved_TFLOPs": f_ref / dt / 1e12,
                                       "peak_TFLOPs": peak, "frac": f_ref / dt / 1e12 / peak, "peak_note": note},
                         "stage_ms_one_step": stage_breakdown(step, mode)}
        del sc, proj, rbs, r
        torch.cuda.empty_cache()
        return out
    if name == "chunk_16384":
        # the headline workload with two and four times the rays per step: what the fixed cost per launch (93
        # launches per step, ~11 us each + ~3.5 us between them) costs at the reference's default chunk of 4096
        from gp_nerf_amd.render_ray import render_rays
        out = {"workload": "the headline workload with 8192 / 16384 rays per step (chunk_size is the caller's choice, "
                           "reference render_image.py:39; results are chunk invariant bit for bit)"}
        for big in (2 * CHUNK, 4 * CHUNK):
            start = min((scene.H // 2) * scene.W, scene.H * scene.W - big)
            rbb = dict(rb)
            rbb["ray_o"], rbb["ray_d"] = scene.rays_o[start:start + big].to(dev), scene.rays_d[start:start + big].to(dev)

            def step(collective=False):
                return render_rays(rbb, model, scene.featmaps, scene.deep, projector, N_samples=N_SAMPLES,
                                   inv_uniform=True, N_importance=N_IMPORTANCE, det=True, ret_alpha=True,
                                   single_net=True)
            for mode in ("fp32", "bf16x3"):
                net.precision = mode
                for _ in range(2):
                    r = step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    r = step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                out.setdefault(mode, {})["rays_%d" % big] = {"value": big / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt}
            del r, rbb
            torch.cuda.empty_cache()
        return out
    if name == "chunks_in_flight_2":
        # the headline step (4096 rays, two passes) issued alternately on TWO streams, two different chunks: what the
        # chunk loop of render_single_image does (render_image.CHUNK_STREAMS) -- the drain of every launch of one
        # chunk runs under the other chunk's kernels.  Kernel durations overlap here, so this leg carries no roofline.
        from gp_nerf_amd.render_ray import render_rays
        out = {"workload": "the headline workload, %d-ray steps alternating over two HIP streams (two chunks in flight, "
                           "as in render_single_image's chunk loop)" % CHUNK}
        rbs = [ray_chunk(scene, r, 2, dev) for r in range(2)]
        from gp_nerf_amd.render_image import _side_streams
        streams = _side_streams(dev, 2)  # the very streams the product's chunk loop uses in this process

        def step2(k):
            with torch.cuda.stream(streams[k % 2]):
                return render_rays(rbs[k % 2], model, scene.featmaps, scene.deep, projector, N_samples=N_SAMPLES,
                                   inv_uniform=True, N_importance=N_IMPORTANCE, det=True, ret_alpha=True,
                                   single_net=True)
        for mode in ("fp32", "bf16x3"):
            net.precision = mode
            torch.cuda.synchronize()
            keep = [step2(k) for k in range(4)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                keep[k % 4] = step2(k)  # (results stay referenced until their stream has run them)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[mode] = {"value": CHUNK / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt}
            del keep
        return out
    if name == "scannet_eval_view":
        # BASELINE.json configs[2]: one ScanNet-shaped evaluation view through the whole chain of
        # eval_gpnerf.py:172-206 -- ResUNet + FPN on the 10 source images, render_single_image over the
        # 240x320 image (render_stride 1, chunk_size 2000 as in configs/gpnerf_scannet.txt), semantic
        # head on the half-resolution feature image, label image.  Results on the host as the driver
        # receives them.  Random-init encoders / head (torch default init), GNT as in the headline.
        out = {"workload": "render_view: ResUNet+FPN encoders on 10 source images 320x240, 76,800 rays in chunks of "
                           "2000, 48 + 16 samples, semantic head (20 classes + 1) and labels; seconds per view"}
        view = scannet_view(dev)
        for mode in ("fp32", "bf16x3"):
            out[mode] = view(mode, repeats=3)
        return out
    raise ValueError(name)


def scannet_view(dev, chunk_size=2000):
    """-> run(mode, repeats) for the scannet_eval_view leg (also used by the full-size test)."""
    from gp_nerf_amd import synth
    from gp_nerf_amd.eval_view import render_view
    from gp_nerf_amd.model import GPNeRFModel
    from gp_nerf_amd.projection import Projector
    from gp_nerf_amd.sample_ray import RaySamplerSingleImage
    torch.manual_seed(4321)
    args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn",
                                 num_classes=20, coarse_feat_dim=32, fine_feat_dim=32, single_net=True,
                                 N_importance=N_IMPORTANCE, N_samples=N_SAMPLES, inv_uniform=True, white_bkgd=False,
                                 chunk_size=chunk_size, render_stride=1, local_rank=dev.index or 0, ckpt_path=None,
                                 no_reload=True, rootdir="/nonexistent", expname="none", n_p=0)
    model = GPNeRFModel(args)
    model.net_coarse.load_state_dict(synth.gnt_state_dict())
    sc = synth.scannet_scene(V=VIEWS, map_source="torch")
    data = {"rgb": sc.src_rgbs[0, :1].clone(), "camera": sc.camera, "rgb_path": "synthetic", "true_depth": None,
            "depth_mask": None, "depth_range": sc.depth_range, "src_rgbs": sc.src_rgbs, "src_cameras": sc.src_cameras}
    sampler = RaySamplerSingleImage(data, dev, render_stride=1)
    projector = Projector(dev)

    def run(mode, repeats=1, chunk=None):
        model.net_coarse.precision = mode
        args.chunk_size = chunk or chunk_size
        best, ret = None, None
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret, _ = render_view(args, model, sampler, projector, render_stride=1, ret_alpha=True,
                                 single_net=True, feats_out_3d=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        n = sc.H * sc.W
        run.last = ret
        return {"precision": mode, "seconds_per_view": best, "rays_per_sec": n / best, "rays": n,
                "labels_shape": list(ret["outputs_fine"]["labels"].shape)}
    run.model, run.sampler, run.projector, run.args = model, sampler, projector, args
    return run


def compare_with_cpu(scene, rb, projector, ret, want, nr, dev):
    """max relative error of one GPU result against the CPU oracle's on the first nr rays.
    The 16 importance samples are an inverse-CDF lookup of the coarse weights, and visibility is a hard
    in-frustum test: a 1e-7 change of the coarse weights can move a sample across a CDF bin edge or across
    the border of a source image (reference behaviour, any two platforms differ this way).  Rays whose
    fine depths or visibility masks differ between the GPU run and the CPU run are counted and left out
    of the fine-level comparison."""
    import oracle
    from gp_nerf_amd import render_ray as rr
    rbc = {k: v.cpu() for k, v in rb.items() if isinstance(v, torch.Tensor)}
    rbc["ray_o"], rbc["ray_d"] = rbc["ray_o"][:nr], rbc["ray_d"][:nr]
    _, zc = oracle.sampling.sample_along_camera_ray(rbc["ray_o"], rbc["ray_d"], rbc["depth_range"],
                                                    N_SAMPLES, inv_uniform=True, det=True)
    zf = [oracle.sampling.sample_fine_pts(True, N_IMPORTANCE, True, N_SAMPLES, rbc, w.clone(), zc)[1]
          for w in (ret["outputs_coarse"]["weights"][:nr].cpu(), want["outputs_coarse"]["weights"])]
    same = ((zf[0] - zf[1]).abs().amax(-1) <= 1e-4 * zf[1].abs().amax(-1))
    # visibility of the fine samples: the GPU pipeline's own points vs the oracle's
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
    return worst, int((~same).sum())


if __name__ == "__main__":
    main()
