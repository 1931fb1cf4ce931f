"""Oracle (test infrastructure): CPU restatement of reference gpnerf/projection.py.

Follows the reference op for op (same torch CPU ops in the same order) so that masks and
bilinear tap indices are bit-identical to the reference on the same machine.
"""
import torch
import torch.nn.functional as F


def projection_matrices(train_cameras):
    """P_v = K_v . inv(c2w_v), [V,4,4]  (reference projection.py:41-44)."""
    K = train_cameras[:, 2:18].reshape(-1, 4, 4)
    c2w = train_cameras[:, -16:].reshape(-1, 4, 4)
    return K.bmm(torch.inverse(c2w))


def compute_projections(xyz, train_cameras):
    """World points -> pixel coordinates in every source view (reference projection.py:31-55).

    xyz [R,S,3], train_cameras [V,34] -> pix [V,R,S,2] (clamped to +-1e6), in_front [V,R,S] bool.
    """
    lead = xyz.shape[:2]
    pts = xyz.reshape(-1, 3)
    V = train_cameras.shape[0]
    hom = torch.cat([pts, torch.ones_like(pts[:, :1])], dim=-1)  # [N,4]
    proj = projection_matrices(train_cameras).bmm(hom.t()[None].repeat(V, 1, 1))  # [V,4,N]
    proj = proj.permute(0, 2, 1)
    pix = proj[..., :2] / torch.clamp(proj[..., 2:3], min=1e-8)
    pix = torch.clamp(pix, min=-1e6, max=1e6)
    in_front = proj[..., 2] > 0
    return pix.reshape((V,) + lead + (2,)), in_front.reshape((V,) + lead)


def inbound(pix, h, w):
    """0 <= x <= w-1 and 0 <= y <= h-1 (reference projection.py:9-22)."""
    x, y = pix[..., 0], pix[..., 1]
    return (x <= w - 1.0) & (x >= 0) & (y <= h - 1.0) & (y >= 0)


def normalize(pix, h, w):
    """n = 2*pix/[w-1,h-1] - 1 (reference projection.py:24-29)."""
    size = torch.tensor([w - 1.0, h - 1.0]).to(pix.device)[None, None, :]
    return 2 * pix / size - 1.0


def compute_angle(xyz, query_camera, train_cameras):
    """Direction-difference feature [V,R,S,4] (reference projection.py:57-82)."""
    lead = xyz.shape[:2]
    pts = xyz.reshape(-1, 3)
    src_o = train_cameras[:, -16:].reshape(-1, 4, 4)[:, :3, 3]  # [V,3]
    V = src_o.shape[0]
    qry_o = query_camera[-16:].reshape(-1, 4, 4).repeat(V, 1, 1)[:, :3, 3]  # [V,3]
    a = qry_o.unsqueeze(1) - pts.unsqueeze(0)
    a = a / (torch.norm(a, dim=-1, keepdim=True) + 1e-6)
    b = src_o.unsqueeze(1) - pts.unsqueeze(0)
    b = b / (torch.norm(b, dim=-1, keepdim=True) + 1e-6)
    diff = a - b
    nrm = torch.norm(diff, dim=-1, keepdim=True)
    dot = torch.sum(a * b, dim=-1, keepdim=True)
    out = torch.cat([diff / torch.clamp(nrm, min=1e-6), dot], dim=-1)
    return out.reshape((V,) + lead + (4,))


def compute(xyz, query_camera, train_imgs, train_cameras, featmaps, deep_semantics,
            return_pix=False):
    """Reference Projector.compute (projection.py:84-139).

    xyz [R,S,3]; query_camera [1,34]; train_imgs [1,V,H,W,3]; train_cameras [1,V,34];
    featmaps [V,F,H/4,W/4]; deep_semantics [V,D,H/2,W/2] (may be None -> deep_sem None).
    Returns rgb_feat [R,S,V,3+F], deep_sem [R,S,V,D], ray_diff [R,S,V,4], mask [R,S,V,1] float.
    """
    assert train_imgs.shape[0] == 1 and train_cameras.shape[0] == 1 and query_camera.shape[0] == 1, \
        "only support batch_size=1 for now"
    imgs = train_imgs.squeeze(0).permute(0, 3, 1, 2)
    cams = train_cameras.squeeze(0)
    qcam = query_camera.squeeze(0)
    h, w = cams[0][:2]

    pix, in_front = compute_projections(xyz, cams)
    grid = normalize(pix, h, w)
    rgb = F.grid_sample(imgs, grid, align_corners=True).permute(2, 3, 0, 1)
    feat = F.grid_sample(featmaps, grid, align_corners=True).permute(2, 3, 0, 1)
    rgb_feat = torch.cat([rgb, feat], dim=-1)
    deep = None
    if deep_semantics is not None:
        deep = F.grid_sample(deep_semantics, grid, align_corners=True).permute(2, 3, 0, 1)
    inb = inbound(pix, h, w)
    ray_diff = compute_angle(xyz, qcam, cams).permute(1, 2, 0, 3)
    mask = (inb * in_front).float().permute(1, 2, 0)[..., None]
    if return_pix:
        return rgb_feat, deep, ray_diff, mask, pix
    return rgb_feat, deep, ray_diff, mask


def tap_indices(pix, h, w, hf, wf):
    """floor() of the un-normalised sampling coordinate on an (hf, wf) map, int32 [...,2] (x0,y0).

    Same float sequence as torch's grid_sampler_unnormalize with align_corners=True:
    ix = ((n + 1) / 2) * (wf - 1) with n from `normalize`.
    """
    n = normalize(pix, h, w)
    ix = ((n[..., 0] + 1.0) / 2.0) * (wf - 1)
    iy = ((n[..., 1] + 1.0) / 2.0) * (hf - 1)
    return torch.stack([torch.floor(ix), torch.floor(iy)], dim=-1).to(torch.int32)
