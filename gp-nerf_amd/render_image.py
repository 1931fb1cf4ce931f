"""Image-level rendering -- same interface as reference gpnerf/render_image.py:6-121.

`render_single_image` keeps the reference's chunk loop semantics and return structure (CPU tensors
reshaped to [H', W', .]), but accumulates chunk results on the device and copies each output to the
host once at the end instead of once per chunk.  `render_image_sharded` is the multi-GPU form
(north_star subsystem 4): ranks render disjoint row blocks of the strided pixel grid and the
rgb / depth / semantic tiles are all-gathered with RCCL (`torch.distributed`, backend "nccl").
"""
import contextlib
import os
from collections import OrderedDict

import torch

from .render_ray import render_rays

_SHARED_KEYS = ("camera", "depth_range", "src_rgbs", "src_cameras", "labels", "src_labels")


_SIDE_STREAMS = {}  # device index -> [torch.cuda.Stream]: the chunk loop's streams, made once per device
# Chunks in flight: 2 = consecutive chunks alternate between two HIP streams, so that the drain of every launch of
# one chunk runs under the other chunk's kernels (a 4096-ray chunk leaves ~5 % of the GPU idle at its 41 launch
# boundaries: 10.5 -> 10.0 ms per chunk in fp32, tools/twostream.py).  1 = the reference's strictly serial loop.
CHUNK_STREAMS = int(os.environ.get("GPN_CHUNK_STREAMS", "2"))
# render_single_image copies every chunk's results to the host behind the chunk (_HostSink); 0: after the last chunk
STREAM_TO_HOST = os.environ.get("GPN_STREAM_TO_HOST", "1") != "0"


def _side_streams(dev, n):
    have = _SIDE_STREAMS.setdefault(dev.index, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def _render_chunks(ray_batch, model, projector, chunk_size, N_samples, inv_uniform, N_importance,
                   det, white_bkgd, featmaps, deep_semantics, ret_alpha, single_net, start, stop,
                   feats_out_3d=True, row_width=None, streams=None, sink=None):
    """Chunk loop (reference render_image.py:39-85) over rays [start, stop); results stay on the
    device: {'outputs_coarse': {k: [chunks]}, 'outputs_fine': ... or None}.
    row_width: rays per row when ray_batch holds the whole strided pixel grid in row-major order; every chunk
    then carries the scheduling hint 'pixel_row' = (row_width, column of its first ray) for render_rays.
    streams: chunks in flight (default CHUNK_STREAMS).  The first chunk runs on the caller's stream (it fills the
    per-image caches: repacked maps, cameras, folded weights); the others alternate over side streams that wait
    for it, and the caller's stream waits for all of them before this function returns -- to the caller the loop
    is still one ordered piece of work on its stream, and the values do not depend on `streams`.
    sink: a _HostSink; every chunk's results are then handed to it inside the chunk's stream context (it queues
    their copies to the host right behind the chunk's kernels) instead of being collected on the device."""
    acc = OrderedDict([("outputs_coarse", OrderedDict()), ("outputs_fine", OrderedDict())])
    starts = list(range(start, stop, chunk_size))
    dev = ray_batch["ray_o"].device
    n_side = CHUNK_STREAMS if streams is None else int(streams)
    side = _side_streams(dev, n_side) if dev.type == "cuda" and n_side > 1 and len(starts) > 2 else []
    cur = torch.cuda.current_stream(dev) if side else None
    for idx, i in enumerate(starts):
        j = min(i + chunk_size, stop)
        chunk = OrderedDict()
        for k in ray_batch:
            if k == "pixel_row":
                continue
            if k in _SHARED_KEYS:
                chunk[k] = ray_batch[k]
            elif ray_batch[k] is not None:
                chunk[k] = ray_batch[k][i:j]
            else:
                chunk[k] = None
        if row_width:
            chunk["pixel_row"] = (row_width, i % row_width)
        if side and idx == 1:
            for st in side:  # everything enqueued so far (inputs, chunk 0 and the caches it filled)
                st.wait_stream(cur)
        st = side[(idx - 1) % len(side)] if side and idx > 0 else None
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            ret = render_rays(chunk, model, featmaps, ref_deep_semantics=deep_semantics,
                              projector=projector, N_samples=N_samples, inv_uniform=inv_uniform,
                              N_importance=N_importance, det=det, white_bkgd=white_bkgd,
                              ret_alpha=ret_alpha, single_net=single_net, feats_out_3d=feats_out_3d)
            if sink is not None:
                sink.take(i - start, j - start, ret, cur if st is not None else None)
        if sink is not None:
            continue
        for lvl in ("outputs_coarse", "outputs_fine"):
            if ret[lvl] is None:
                acc[lvl] = None
                continue
            for k, v in ret[lvl].items():
                if v is not None:
                    if st is not None and torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)  # made on a side stream, read on the caller's from here on
                    acc[lvl].setdefault(k, []).append(v)
    for st in side:
        cur.wait_stream(st)
    return acc if sink is None else sink


def _strided_hw(ray_sampler, render_stride):
    return (len(range(0, ray_sampler.H, render_stride)), len(range(0, ray_sampler.W, render_stride)))


_PINNED = {}  # (level, key) -> pinned host buffer reused across images (page-locked D2H runs at PCIe speed)
# Buffers above this size are not kept between images (feats_out_3d of a 640x480 image is ~70 GB: it
# is staged through pageable memory once rather than pinned for the life of the process).
PINNED_KEEP_BYTES = 1 << 30


def _to_host(t, tag):
    """One asynchronous copy into a cached page-locked buffer (the reference's per-chunk `.cpu()` into
    pageable memory, render_image.py:78-85, costs more than the rendering of a small image).  The
    returned CPU tensor is a fresh clone, so results of successive images do not alias."""
    nbytes = t.numel() * t.element_size()
    if nbytes > PINNED_KEEP_BYTES:
        return t.cpu(), None, False  # synchronous copy into a fresh pageable tensor; nothing retained
    view = _pinned(tag, tuple(t.shape), t.dtype)
    view.copy_(t, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return view, done, True


def _finish(out, lvl, k, flat, Hs, Ws):
    if k == "feats_out":
        out[lvl][k] = flat.reshape((Hs, Ws, 512, -1)).squeeze()
    else:
        out[lvl][k] = flat.reshape((Hs, Ws, -1)).squeeze()


class _HostSink:
    """Host side of render_single_image's chunk loop: every chunk's outputs are copied into page-locked staging
    buffers by asynchronous copies queued on the chunk's own stream right behind its kernels, and moved from there
    into the result tensors by the host while the GPU renders the following chunks -- the copies and the host-side
    moves of a 640x480 / stride-2 image (350 MB, 19 of 207 ms when done after the last chunk) disappear behind the
    rendering, except the last chunk's.  Same values, same result structure as _merge.  Outputs too large to
    stage (feats_out_3d of a big image, > PINNED_KEEP_BYTES) stay on the device and take _merge's path."""

    def __init__(self, n_rays, dev):
        self.n = n_rays
        # the copies run on a stream of their own, behind an event of the chunk: the chunk's stream goes straight on
        # to its next chunk (copies queued on the rendering streams themselves cost 12 of the 19 ms they were to save)
        self.copy_stream = _side_streams(dev, 3)[2]
        self.out = OrderedDict([("outputs_coarse", OrderedDict()), ("outputs_fine", OrderedDict())])
        self.stage, self.big, self.pending = {}, OrderedDict(), []

    def take(self, i, j, ret, reader_stream):
        staged = []
        ready = torch.cuda.Event()
        ready.record()  # on the chunk's stream: behind its kernels
        self.copy_stream.wait_event(ready)
        for lvl in ("outputs_coarse", "outputs_fine"):
            if ret[lvl] is None:
                self.out[lvl] = None
                continue
            for k, v in ret[lvl].items():
                if v is None or k == "random_sigma":
                    continue
                tag = (lvl, k)
                if tag not in self.stage and tag not in self.big and k not in self.out[lvl]:  # first chunk
                    shape = (self.n,) + tuple(v.shape[1:])
                    if v.is_cuda and v[:1].numel() * self.n * v.element_size() > PINNED_KEEP_BYTES:
                        self.big[tag] = []
                    else:
                        if v.is_cuda:
                            self.stage[tag] = _pinned(tag, shape, v.dtype)
                        self.out[lvl][k] = torch.empty(shape, dtype=v.dtype)
                    if tag in self.big:
                        self.out[lvl][k] = None  # (keeps the key order of the result)
                if tag in self.big:
                    if reader_stream is not None:
                        v.record_stream(reader_stream)
                    self.big[tag].append(v)
                elif not v.is_cuda:
                    self.out[lvl][k][i:j] = v
                else:
                    with torch.cuda.stream(self.copy_stream):
                        self.stage[tag][i:j].copy_(v, non_blocking=True)
                    staged.append((lvl, k, v))  # (v stays referenced until its copy has run: finish())
        if staged:
            done = torch.cuda.Event()
            done.record(self.copy_stream)
            self.pending.append((i, j, done, staged))

    def finish(self, Hs, Ws):
        for i, j, done, staged in self.pending:  # in chunk order, while later chunks are still being rendered
            done.synchronize()
            for lvl, k, _ in staged:
                self.out[lvl][k][i:j].copy_(self.stage[(lvl, k)][i:j])
        self.pending = []
        late = [(lvl, k) + _to_host(torch.cat(vs, dim=0), (lvl, k)) for (lvl, k), vs in self.big.items()]
        for lvl, k, host, done, shared in late:
            if done is not None:
                done.synchronize()
            self.out[lvl][k] = host.clone() if shared else host
        for lvl in ("outputs_coarse", "outputs_fine"):
            if self.out[lvl] is not None:
                for k in list(self.out[lvl]):
                    _finish(self.out, lvl, k, self.out[lvl][k], Hs, Ws)
        return self.out


def _pinned(tag, shape, dtype):
    numel = 1
    for d in shape:
        numel *= d
    buf = _PINNED.get(tag)
    if buf is None or buf.numel() < numel or buf.dtype != dtype:
        buf = torch.empty(numel, dtype=dtype, pin_memory=True)
        _PINNED[tag] = buf
    return buf[:numel].view(shape)


def _merge(acc, Hs, Ws):
    """concat + reshape (reference render_image.py:87-119)."""
    staged = []
    for lvl in ("outputs_coarse", "outputs_fine"):
        if acc[lvl] is None:
            continue
        for k in list(acc[lvl]):
            if k == "random_sigma":
                continue
            cat = torch.cat(acc[lvl][k], dim=0)
            staged.append((lvl, k) + (_to_host(cat, (lvl, k)) if cat.is_cuda else (cat, None, False)))
    # every copy has been queued; each output is cloned out of its page-locked buffer as soon as ITS
    # copy has landed, while the later ones are still in flight (the clones cost more than the copies)
    for lvl, k, host, done, shared in staged:
        if done is not None:
            done.synchronize()
        _finish(acc, lvl, k, host.clone() if shared else host, Hs, Ws)
    return acc


def render_single_image(
    ray_sampler,
    ray_batch,
    model,
    projector,
    chunk_size,
    N_samples,
    inv_uniform=False,
    N_importance=0,
    det=False,
    white_bkgd=False,
    render_stride=1,
    featmaps=None,
    deep_semantics=None,
    ret_alpha=False,
    single_net=False,
    feats_out_3d=True,
):
    """reference render_image.py:6-121 -> {'outputs_coarse': {...}, 'outputs_fine': {...}} of CPU
    tensors: rgb [H',W',3], depth [H',W'], weights [H',W',S], feats_out [H',W',512],
    feats_out_3d [H',W',S*512] (skipped with the extension flag feats_out_3d=False)."""
    N_rays = ray_batch["ray_o"].shape[0]
    Hs, Ws = _strided_hw(ray_sampler, render_stride)
    on_gpu = ray_batch["ray_o"].is_cuda
    acc = _render_chunks(ray_batch, model, projector, chunk_size, N_samples, inv_uniform,
                         N_importance, det, white_bkgd, featmaps, deep_semantics, ret_alpha,
                         single_net, 0, N_rays, feats_out_3d, row_width=Ws if N_rays == Hs * Ws else None,
                         sink=_HostSink(N_rays, ray_batch["ray_o"].device) if on_gpu and STREAM_TO_HOST else None)
    return acc.finish(Hs, Ws) if isinstance(acc, _HostSink) else _merge(acc, Hs, Ws)


def shard_rows(Hs, world_size):
    """Row-aligned partition of the strided pixel grid: every rank gets ceil(Hs/G) rows (the last
    ranks may own fewer / zero real rows; tiles are padded so that all-gather sizes are equal)."""
    rows = (Hs + world_size - 1) // world_size
    return rows, [(min(r * rows, Hs), min((r + 1) * rows, Hs)) for r in range(world_size)]


GATHER_KEYS = ("rgb", "depth", "feats_out")
_KEY_WIDTH = {"rgb": 3, "depth": 1, "feats_out": 512}


def tile_layout(N_samples, N_importance, ret_alpha, has_semantics, gather_weights):
    """Columns of the all-gathered tile per output level, derived from the call's ARGUMENTS only --
    every rank must agree on them, including a rank that owns no rows and therefore renders nothing.
    -> {level: [(key, width), ...] or None (level absent)}."""
    out = OrderedDict()
    for lvl in ("outputs_coarse", "outputs_fine"):
        fine = lvl == "outputs_fine"
        if fine and N_importance <= 0:
            out[lvl] = None
            continue
        with_alpha = fine or ret_alpha  # the fine pass always runs with ret_alpha (reference render_ray.py:290)
        cols = [("rgb", 3)]
        if with_alpha:
            cols.append(("depth", 1))
        if has_semantics:
            cols.append(("feats_out", 512))
        if gather_weights and with_alpha:
            cols.append(("weights", N_samples + (N_importance if fine else 0)))
        out[lvl] = cols
    return out


def render_rank_tiles(rank, G, ray_sampler, ray_batch, model, projector, chunk_size, N_samples,
                      inv_uniform=False, N_importance=0, det=False, white_bkgd=False, render_stride=1,
                      featmaps=None, deep_semantics=None, ret_alpha=False, single_net=False,
                      gather_weights=False, render_fn=None, split_feats=False):
    """The rank-local half of render_image_sharded: render rows [rank*rows, (rank+1)*rows) of the
    strided grid and pack them into equal-size tiles, zero-padded (a rank without rows returns all
    zeros).  -> (layout, tiles): tiles[level] = None (level absent) or a dict with 'shared'
    [rows*W', sum of the shared columns] and, with split_feats, 'feats' [rows*W', 512]."""
    Hs, Ws = _strided_hw(ray_sampler, render_stride)
    rows, parts = shard_rows(Hs, G)
    r0, r1 = parts[rank]
    extra = {} if render_fn is not None or ray_batch["ray_o"].shape[0] != Hs * Ws else {"row_width": Ws}
    fn = render_fn or _render_chunks
    acc = fn(ray_batch, model, projector, chunk_size, N_samples, inv_uniform, N_importance, det,
             white_bkgd, featmaps, deep_semantics, ret_alpha, single_net, r0 * Ws, r1 * Ws, False, **extra)
    dev = ray_batch["ray_o"].device
    layout = tile_layout(N_samples, N_importance, ret_alpha, deep_semantics is not None, gather_weights)
    tiles = OrderedDict()
    for lvl, cols in layout.items():
        if cols is None:
            tiles[lvl] = None
            continue

        def build(spec):
            tile = torch.zeros((rows * Ws, sum(w for _, w in spec)), device=dev)
            if r1 > r0:
                pieces = []
                for k, w in spec:
                    t = torch.cat(acc[lvl][k], dim=0)
                    t = t.reshape(t.shape[0], -1)
                    if tuple(t.shape) != ((r1 - r0) * Ws, w):
                        raise RuntimeError("rank %d: %s/%s has shape %s, the tile layout expects [%d, %d]"
                                           % (rank, lvl, k, tuple(t.shape), (r1 - r0) * Ws, w))
                    pieces.append(t)
                tile[: (r1 - r0) * Ws] = torch.cat(pieces, dim=1)
            return tile

        apart = split_feats and any(k == "feats_out" for k, _ in cols)
        shared = [(k, w) for k, w in cols if not (apart and k == "feats_out")]
        tiles[lvl] = {"shared": build(shared), "shared_cols": shared,
                      "feats": build([("feats_out", 512)]) if apart else None}
    return layout, tiles


def merge_rank_tiles(cols, full, Hs, Ws):
    """Rank-ordered concatenation of equal-size tiles [G*rows*W', width] -> {key: [H',W',.]} (the
    padding rows of the last ranks are cut off)."""
    full = full[: Hs * Ws]
    out, o = OrderedDict(), 0
    for k, wdt in cols:
        out[k] = full[:, o:o + wdt].reshape(Hs, Ws, wdt).squeeze()
        o += wdt
    return out


def render_image_sharded(ray_sampler, ray_batch, model, projector, chunk_size, N_samples,
                         inv_uniform=False, N_importance=0, det=False, white_bkgd=False,
                         render_stride=1, featmaps=None, deep_semantics=None, ret_alpha=False,
                         single_net=False, group=None, gather_weights=False, render_fn=None,
                         sem_head=None, labels_only=False):
    """Multi-GPU render_single_image: rank k renders rows [k*rows, (k+1)*rows) of the strided grid
    (source maps, cameras and weights replicated), then one all_gather_into_tensor per output level
    moves rgb[3] + depth[1] + feats_out[512] (+ weights[S]) per ray.  feats_out_3d is never
    gathered.  Returns device tensors shaped like render_single_image's outputs on every rank.
    A rank whose row block is empty (H' < world size) renders nothing and contributes a zero tile.

    sem_head (a NeRFSemSegFPNHead on this rank's device): adds 'labels' (uint8 [2h,2w], the label
    image of the head's logits: reference eval_gpnerf.py:201-206 + loss.py:296) to every level.
    With labels_only=True the 2 KB/ray feature tiles do not go to every rank: they are gathered to
    rank 0 only, rank 0 runs the head (its GroupNorm needs the whole image) and the 1 B/pixel label
    image is broadcast; rgb / depth are still all-gathered; 'feats_out' is then present on rank 0 only.
    `render_fn` (tests only) replaces the per-rank chunk renderer."""
    import torch.distributed as dist

    multi = dist.is_initialized()
    G = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    # the collectives are issued in a one-rank job too when asked for (tests: the RCCL calls of this
    # function run on real hardware with a single GPU)
    coll = G > 1 or (multi and os.environ.get("GPN_SHARD_FORCE_COLLECTIVES") == "1")
    root = dist.get_global_rank(group, 0) if (multi and group is not None) else 0
    if labels_only and sem_head is None:
        raise ValueError("labels_only=True needs a sem_head")
    Hs, Ws = _strided_hw(ray_sampler, render_stride)
    layout, tiles = render_rank_tiles(
        rank, G, ray_sampler, ray_batch, model, projector, chunk_size, N_samples, inv_uniform,
        N_importance, det, white_bkgd, render_stride, featmaps, deep_semantics, ret_alpha, single_net,
        gather_weights, render_fn, split_feats=labels_only)
    dev = ray_batch["ray_o"].device
    out = OrderedDict()
    for lvl, cols in layout.items():
        if cols is None:
            out[lvl] = None
            continue
        tile = tiles[lvl]["shared"]
        if coll:
            full = torch.empty((G * tile.shape[0], tile.shape[1]), device=dev)
            dist.all_gather_into_tensor(full, tile, group=group)
        else:
            full = tile
        out[lvl] = merge_rank_tiles(tiles[lvl]["shared_cols"], full, Hs, Ws)
        ftile = tiles[lvl]["feats"]
        if ftile is not None:  # labels_only: the feature tiles go to rank 0 only
            if coll:
                bufs = [torch.empty_like(ftile) for _ in range(G)] if rank == 0 else None
                dist.gather(ftile, bufs, dst=root, group=group)
                if rank == 0:
                    out[lvl]["feats_out"] = torch.cat(bufs, dim=0)[: Hs * Ws].reshape(Hs, Ws, 512)
            else:
                out[lvl]["feats_out"] = ftile[: Hs * Ws].reshape(Hs, Ws, 512)
        if sem_head is not None and "feats_out" in dict(cols):
            lab = None
            if not labels_only or rank == 0:
                feats = out[lvl]["feats_out"]
                if render_stride == 1:  # reference eval_gpnerf.py:201-203: the head works on the half grid
                    feats = feats[::2, ::2, :]
                lab = sem_head.labels(feats.permute(2, 0, 1).unsqueeze(0).contiguous(), torch.uint8)
            if labels_only and coll:
                hh, ww = (len(range(0, Hs, 2)), len(range(0, Ws, 2))) if render_stride == 1 else (Hs, Ws)
                if lab is None:
                    lab = torch.empty((2 * hh, 2 * ww), dtype=torch.uint8, device=dev)
                dist.broadcast(lab, src=root, group=group)
            out[lvl]["labels"] = lab
    return out
