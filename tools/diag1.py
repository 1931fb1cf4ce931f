import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import cases, oracle
from gp_nerf_amd import _lib
from gp_nerf_amd.projection import Projector
from gp_nerf_amd import render_ray as rr
dev = torch.device("cuda:0")
to = lambda d: {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}

print("== posenc S=48")
c = cases.g2_gnt(48)
R, S = 8, 48
N = R * S
emb = torch.empty(N, 128, device=dev)
_lib.check(_lib.lib().gpn_posenc(_lib.ptr(c["pts"].to(dev).contiguous()), _lib.ptr(c["ray_d"].to(dev).contiguous()), R, S, _lib.ptr(emb), _lib.stream()), "posenc")
pe, ve = oracle.gnt.encodings(c["pts"], c["ray_d"])
want = torch.cat([pe.reshape(N, 63), ve[:, None].expand(R, S, 63).reshape(N, 63), torch.zeros(N, 2)], -1)
d = (emb.cpu() - want).abs()
idx = torch.nonzero(d > 1e-4)
print("n bad", idx.shape[0], "of", d.numel())
for n, ch in idx[:12].tolist():
    k = ch if ch < 63 else ch - 63
    src = c["pts"].reshape(N, 3)[n] if ch < 63 else (c["ray_d"][n // S] / c["ray_d"][n // S].norm())
    if k >= 3:
        f = (k - 3) // 6; comp = (k - 3) % 3
        arg = float(src[comp]) * 2 ** f
        import math
        print(n, ch, "f", f, "arg", arg, "got", float(emb[n, ch]), "want", float(want[n, ch]), "f64 sin/cos", math.sin(np.float32(src[comp]) * np.float32(2 ** f)), math.cos(np.float32(src[comp]) * np.float32(2 ** f)))

print("== samplers")
g = np.load("tests/golden/g3_samplers.npz")
c = cases.g3_samplers(); cd = to(c)
for inv, tag in ((False, "lin"), (True, "inv")):
    pts, z = rr.sample_along_camera_ray(cd["ray_o"], cd["ray_d"], cd["depth_range"], 48, inv_uniform=inv, det=True)
    fpts, fz = rr.sample_fine_pts(inv, 16, True, 48, cd, cd["weights"], z)
    e = np.abs(fz.cpu().numpy() - g["fz_" + tag]).max(-1)
    bad = np.nonzero(e > 1e-5)[0]
    print(tag, "bad rays", bad[:20], e[bad][:10])
    if len(bad):
        r = int(bad[0])
        print(" got ", fz[r].cpu().numpy()); print(" want", g["fz_" + tag][r]); print(" w", c["weights"][r].numpy())

print("== projector ragged")
c = cases.g1_projector(); cd = to(c)
P = Projector(dev)
for R in (1, 3, 37, 64):
    for Ssub in (5, 16):
        pts = cd["pts"][:R, :Ssub].contiguous()
        got = P.compute(pts, cd["camera"], cd["src_rgbs"], cd["src_cameras"], cd["featmaps"], cd["deep_semantics"])
        want = oracle.projector.compute(c["pts"][:R, :Ssub].contiguous(), c["camera"], c["src_rgbs"], c["src_cameras"], c["featmaps"], c["deep_semantics"])
        want_nc = oracle.projector.compute(c["pts"][:R, :Ssub], c["camera"], c["src_rgbs"], c["src_cameras"], c["featmaps"], c["deep_semantics"])
        full = oracle.projector.compute(c["pts"], c["camera"], c["src_rgbs"], c["src_cameras"], c["featmaps"], c["deep_semantics"])
        msg = []
        for gt, wt, wn, fu, nm in zip(got, want, want_nc, full, ("rgb_feat", "deep", "ray_diff", "mask")):
            gtc = gt.cpu()
            msg.append("%s: vs_oracle %d vs_oracle_noncontig %d vs_full_slice %d | oracle_vs_fullslice %d maxdiff %.3g" % (
                nm, int((gtc != wt).sum()), int((gtc != wn).sum()), int((gtc != fu[:R, :Ssub]).sum()), int((wt != fu[:R, :Ssub]).sum()), float((gtc - wt).abs().max())))
        print(R, Ssub, " ; ".join(msg))
