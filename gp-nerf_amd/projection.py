"""`Projector` -- same interface as reference gpnerf/projection.py:5-139, computed by the fused
gfx950 kernel `gpn_project_gather` (csrc/projector.hip).

Host side does only what must match the reference's CPU arithmetic bit for bit and is O(V): the
projection matrices P_v = K_v . inv(c2w_v) are formed with the same torch CPU ops the reference uses
(projection.py:41-44) and cached per camera tensor; NCHW source maps are repacked once to NHWC on
the device (cached per map tensor).  Everything per point runs on the GPU.
"""
import ctypes as C

import torch

from . import _lib


def _key(t):
    """Identity of a device tensor's current contents.  Only valid while the tensor is alive: every
    cache below stores the keyed tensor next to its entry, so that the caching allocator cannot hand
    the same address to a later tensor (another image's cameras have the same shape and _version 0)
    while the entry exists."""
    return (t.data_ptr(), t._version, tuple(t.shape), str(t.device))


class SourceViews:
    """Device-resident, kernel-ready view of one image's source data (built by Projector.prepare)."""

    def __init__(self):
        self.V = 0
        self.img_h = self.img_w = 0.0
        self.proj = self.src_pos = None       # [V,4,4], [V,3] device
        self.proj_cpu = None
        self.qry_pos = (0.0, 0.0, 0.0)
        self.rgb = None                        # [V,H,W,3]
        self.feat = None                       # [V,Hf,Wf,32] NHWC
        self.deep = None                       # [V,Hs,Ws,512] NHWC or None


class Projector:
    def __init__(self, device):
        self.device = device
        self._cam_cache = {}  # key -> (..., keyed tensor)
        self._map_cache = {}  # role ('feat' / 'deep') -> (key, NHWC tensor, keyed tensor): one image's worth

    # ------------------------------------------------------------------ small helpers (API parity)
    def inbound(self, pixel_locations, h, w):
        """reference projection.py:9-22 (elementwise; stays a tensor expression)."""
        x, y = pixel_locations[..., 0], pixel_locations[..., 1]
        return (x <= w - 1.0) & (x >= 0) & (y <= h - 1.0) & (y >= 0)

    def normalize(self, pixel_locations, h, w):
        """reference projection.py:24-29."""
        size = torch.tensor([float(w) - 1.0, float(h) - 1.0], device=pixel_locations.device)
        return 2 * pixel_locations / size[None, None, :] - 1.0

    # ------------------------------------------------------------------ host-side O(V) preparation
    def _cameras(self, train_cameras):
        """train_cameras [V,34] (device) -> (P [V,4,4] dev, centres [V,3] dev, h, w); cached."""
        k = _key(train_cameras)
        hit = self._cam_cache.get(k)
        if hit is None:
            cams = train_cameras.detach().float().cpu()
            K = cams[:, 2:18].reshape(-1, 4, 4)
            c2w = cams[:, -16:].reshape(-1, 4, 4)
            P = K.bmm(torch.inverse(c2w)).contiguous()  # same CPU ops as the reference
            pos = c2w[:, :3, 3].contiguous()
            dev = train_cameras.device
            # (train_cameras rides along: it keeps the keyed address from being recycled)
            hit = (P.to(dev), pos.to(dev), float(cams[0, 0]), float(cams[0, 1]), P, train_cameras)
            if len(self._cam_cache) > 64:
                self._cam_cache.clear()
            self._cam_cache[k] = hit
        return hit[:5]

    def _query_pos(self, query_camera):
        """Camera centre of the query view as host floats; cached like _cameras."""
        k = ("q",) + _key(query_camera)
        hit = self._cam_cache.get(k)
        if hit is None:
            qc = query_camera.detach().float().cpu().reshape(-1)[-16:].reshape(4, 4)
            hit = ((float(qc[0, 3]), float(qc[1, 3]), float(qc[2, 3])), query_camera)
            if len(self._cam_cache) > 64:
                self._cam_cache.clear()
            self._cam_cache[k] = hit
        return hit[0]

    def _nhwc(self, maps, role="feat"):
        """[V,C,H,W] device tensor -> cached NHWC copy.  One entry per role: when the next image's
        maps arrive the previous image's copy (1.6 GB for a 640x480 deep-semantic stack) is dropped."""
        k = _key(maps)
        hit = self._map_cache.get(role)
        if hit is not None and hit[0] == k:
            return hit[1]
        if maps.dtype == torch.float32 and maps.is_cuda and maps.dim() == 4 and \
                maps.permute(0, 2, 3, 1).is_contiguous():
            # channels_last map (the encoders of feature_network.py emit this): already NHWC in memory
            nhwc = maps.permute(0, 2, 3, 1)
        else:
            src = _lib.dev_f32(_lib.as_f32c(maps), "feature maps")
            V, Cc, H, W = src.shape
            nhwc = torch.empty((V, H, W, Cc), dtype=torch.float32, device=src.device)
            _lib.check(_lib.lib().gpn_repack_nhwc(_lib.ptr(src), _lib.ptr(nhwc), V, Cc, H, W,
                                                  _lib.stream()), "gpn_repack_nhwc")
        self._map_cache[role] = (k, nhwc, maps)  # `maps` rides along: its address cannot be recycled
        return nhwc

    def prepare(self, query_camera, train_imgs, train_cameras, featmaps, deep_semantics):
        """Validate the reference-shaped inputs once and return a SourceViews for the kernels."""
        assert (train_imgs.shape[0] == 1) and (train_cameras.shape[0] == 1) and (
            query_camera.shape[0] == 1), "only support batch_size=1 for now"
        sv = SourceViews()
        cams = train_cameras.squeeze(0)
        sv.proj, sv.src_pos, sv.img_h, sv.img_w, sv.proj_cpu = self._cameras(cams)
        sv.V = cams.shape[0]
        sv.qry_pos = self._query_pos(query_camera)
        sv.rgb = _lib.dev_f32(_lib.as_f32c(train_imgs.squeeze(0)), "train_imgs")
        if sv.rgb.shape[0] != sv.V or sv.rgb.shape[-1] != 3:
            raise ValueError("train_imgs must be [1,V,H,W,3]")
        if featmaps is not None:
            sv.feat = self._nhwc(featmaps, "feat")
            if sv.feat.shape[0] != sv.V:
                raise ValueError("featmaps must have one map per source view")
        if deep_semantics is not None:
            sv.deep = self._nhwc(deep_semantics, "deep")
            if sv.deep.shape[0] != sv.V:
                raise ValueError("deep_semantics must have one map per source view")
        return sv

    def _launch(self, xyz, sv, want_feat, want_deep, want_geom=True, want_pix=False, feat_ld=0):
        pts = _lib.dev_f32(_lib.as_f32c(xyz), "xyz").reshape(-1, 3)
        lead = tuple(xyz.shape[:-1])
        N, V, dev = pts.shape[0], sv.V, pts.device
        a = _lib.ProjectArgs()
        a.xyz, a.proj, a.src_pos = _lib.ptr(pts), _lib.ptr(sv.proj), _lib.ptr(sv.src_pos)
        a.qry_pos = (C.c_float * 3)(*sv.qry_pos)
        a.img_h, a.img_w, a.N, a.V = sv.img_h, sv.img_w, N, V
        a.samples_per_ray = xyz.shape[-2] if xyz.dim() == 3 else 0  # scheduling hint of the deep gather
        a.rgb, a.rgb_h, a.rgb_w = _lib.ptr(sv.rgb), sv.rgb.shape[1], sv.rgb.shape[2]
        out = {}
        if want_feat:
            a.feat = _lib.ptr(sv.feat)
            a.feat_h, a.feat_w, a.feat_c = sv.feat.shape[1:]
            out["rgb_feat"] = torch.empty(lead + (V, feat_ld or 3 + sv.feat.shape[3]), device=dev)
            a.rgb_feat, a.rgb_feat_ld = _lib.ptr(out["rgb_feat"]), feat_ld
        if want_deep:
            a.deep = _lib.ptr(sv.deep)
            a.deep_h, a.deep_w, a.deep_c = sv.deep.shape[1:]
            out["deep"] = torch.empty(lead + (V, sv.deep.shape[3]), device=dev)
            a.deep_out = _lib.ptr(out["deep"])
        if want_geom:
            out["ray_diff"] = torch.empty(lead + (V, 4), device=dev)
            out["mask"] = torch.empty(lead + (V, 1), device=dev)
            a.ray_diff, a.mask = _lib.ptr(out["ray_diff"]), _lib.ptr(out["mask"])
        if want_pix:
            out["pix"] = torch.empty((V,) + lead + (2,), device=dev)
            out["taps"] = torch.empty((V,) + lead + (2,), device=dev, dtype=torch.int32)
            out["front"] = torch.empty((V,) + lead, device=dev, dtype=torch.uint8)
            a.pix, a.taps, a.front = _lib.ptr(out["pix"]), _lib.ptr(out["taps"]), _lib.ptr(out["front"])
        _lib.check(_lib.lib().gpn_project_gather(C.byref(a), _lib.stream()), "gpn_project_gather")
        return out

    def fused_args(self, sv):
        """gpn_project_args_t of the source views for gpn_render_pass (projector fused into the network's
        first stage): maps and cameras only -- points, N and the outputs are filled in by the callee."""
        if sv.feat is None:
            raise ValueError("the fused path needs feature maps")
        a = _lib.ProjectArgs()
        a.proj, a.src_pos = _lib.ptr(sv.proj), _lib.ptr(sv.src_pos)
        a.qry_pos = (C.c_float * 3)(*sv.qry_pos)
        a.img_h, a.img_w, a.V = sv.img_h, sv.img_w, sv.V
        a.rgb, a.rgb_h, a.rgb_w = _lib.ptr(sv.rgb), sv.rgb.shape[1], sv.rgb.shape[2]
        a.feat = _lib.ptr(sv.feat)
        a.feat_h, a.feat_w, a.feat_c = sv.feat.shape[1:]
        return a

    # ------------------------------------------------------------------ reference API
    def compute_projections(self, xyz, train_cameras):
        """reference projection.py:31-55: pix [V,...,2], in-front mask [V,...] (bool)."""
        sv = SourceViews()
        sv.proj, sv.src_pos, sv.img_h, sv.img_w, sv.proj_cpu = self._cameras(train_cameras)
        sv.V = train_cameras.shape[0]
        sv.rgb = torch.empty((sv.V, 2, 2, 3), device=xyz.device)  # geometry only: maps unused
        out = self._launch(xyz, sv, False, False, want_geom=False, want_pix=True)
        return out["pix"], out["front"].bool()

    def compute_angle(self, xyz, query_camera, train_cameras):
        """reference projection.py:57-82: [V,...,4]."""
        sv = SourceViews()
        sv.proj, sv.src_pos, sv.img_h, sv.img_w, sv.proj_cpu = self._cameras(train_cameras)
        sv.V = train_cameras.shape[0]
        sv.qry_pos = self._query_pos(query_camera)
        sv.rgb = torch.empty((sv.V, 2, 2, 3), device=xyz.device)
        out = self._launch(xyz, sv, False, False, want_geom=True)
        nd = out["ray_diff"].dim()
        return out["ray_diff"].permute(nd - 2, *range(nd - 2), nd - 1)

    def compute(self, xyz, query_camera, train_imgs, train_cameras, featmaps, deep_semantics):
        """reference projection.py:84-139.  Returns rgb_feat [R,S,V,3+F], deep_sem [R,S,V,D] (None
        if deep_semantics is None), ray_diff [R,S,V,4], mask [R,S,V,1]."""
        sv = self.prepare(query_camera, train_imgs, train_cameras, featmaps, deep_semantics)
        out = self._launch(xyz, sv, True, deep_semantics is not None)
        return out["rgb_feat"], out.get("deep"), out["ray_diff"], out["mask"]

    def compute_fused(self, xyz, sv):
        """Fast path used by render_rays: no [R,S,V,512] materialisation (the deep-semantic maps are
        sampled later, inside gpn_sem_aggregate).  rgb_feat comes back with rows padded to 36 floats
        (16-byte aligned rows, channels 35.. zero): the layout gpn_rgbfeat reads as whole 16-byte units;
        GNT.run takes either width."""
        out = self._launch(xyz, sv, True, False, feat_ld=36 if sv.feat.shape[3] == 32 else 0)
        return out["rgb_feat"], out["ray_diff"], out["mask"]
