import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import cases
from gp_nerf_amd.transformer_network import GNT
dev = torch.device("cuda:0")
args = types.SimpleNamespace(netwidth=64, trans_depth=8, save_feature=False, semantic_model="fpn", num_classes=20)
net = GNT(args, 32, 63, 63, ret_alpha=True); net.load_state_dict(cases.weights()); net = net.to(dev).eval()
def rel(a, b):
    a = a.float().cpu().numpy(); b = np.asarray(b)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
for S in (48, 64):
    g = np.load("tests/golden/g2_gnt_s%d.npz" % S)
    c = {k: v.to(dev) for k, v in cases.g2_gnt(S).items()}
    for prec in ("fp32", "bf16x3"):
        net.precision = prec
        out, sem, sem3d = net(c["rgb_feat"], c["deep_sem"], c["ray_diff"], c["mask"], c["pts"], c["ray_d"])
        print(S, prec, "out", rel(out, g["out"]), "sem", rel(sem, g["sem"]), "finite", bool(torch.isfinite(out).all()))
