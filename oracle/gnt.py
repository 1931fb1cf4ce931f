"""Oracle (test infrastructure): functional CPU restatement of reference
gpnerf/transformer_network.py (GNT.forward and its sub-modules), driven by a plain state dict with
the reference's parameter names.  Eval mode: every Dropout is the identity.

Besides `forward` (the literal reference dataflow, which materialises the [R,S,V,512] deep-semantic
products) this file restates the algebraically identical *fused* semantic aggregation the HIP
path uses (SURVEY.md section 2a "key structural finding"): sem outputs are linear in deep_sem, so
they can be formed from per-layer attention weights after the transformer has run.
"""
import math

import torch
import torch.nn.functional as F


def embed(x, num_freqs=10, max_freq_log2=9):
    """Embedder(include_input, log-sampled freqs, [sin, cos]) (transformer_network.py:6-37)."""
    freqs = 2.0 ** torch.linspace(0.0, max_freq_log2, steps=num_freqs)
    out = [x]
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def feed_forward(sd, p, x):
    """FeedForward (transformer_network.py:40-51)."""
    return _lin(sd, p + ".fc2", F.relu(_lin(sd, p + ".fc1", x)))


def attention2d(sd, p, q, k, pos, mask):
    """Attention2D minus the deep-semantic product (transformer_network.py:74-88).
    Returns x [R,S,C] and the per-channel view softmax attn [R,S,V,C]."""
    q = _lin(sd, p + ".q_fc", q)
    k = _lin(sd, p + ".k_fc", k)
    v = _lin(sd, p + ".v_fc", k)  # quirk: v_fc applied to the projected k (:77)
    pos = _lin(sd, p + ".pos_fc.2", F.relu(_lin(sd, p + ".pos_fc.0", pos)))
    attn = k - q[:, :, None, :] + pos
    attn = _lin(sd, p + ".attn_fc.2", F.relu(_lin(sd, p + ".attn_fc.0", attn)))
    if mask is not None:
        attn = attn.masked_fill(mask == 0, -1e9)
    attn = torch.softmax(attn, dim=-2)
    x = ((v + pos) * attn).sum(dim=2)
    x = _lin(sd, p + ".out_fc", x)
    return x, attn


def view_transformer(sd, p, q, k, pos, mask):
    """Transformer2D (transformer_network.py:106-122). Returns q', attn, q_after_attn."""
    x, attn = attention2d(sd, p + ".attn", _ln(sd, p + ".attn_norm", q, 1e-6), k, pos, mask)
    x = x + q
    mid = x
    x = x + feed_forward(sd, p + ".ff", _ln(sd, p + ".ff_norm", x, 1e-6))
    return x, attn, mid


def ray_attention(sd, p, x, n_heads=4):
    """Attention, attn_mode='qk' (transformer_network.py:149-178). Returns out [R,S,C], A [R,H,S,S]."""
    R, S, C = x.shape
    q = _lin(sd, p + ".q_fc", x).view(R, S, n_heads, -1).permute(0, 2, 1, 3)
    k = _lin(sd, p + ".k_fc", x).view(R, S, n_heads, -1).permute(0, 2, 1, 3)
    v = _lin(sd, p + ".v_fc", x).view(R, S, n_heads, -1).permute(0, 2, 1, 3)
    A = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.shape[-1]), dim=-1)
    out = torch.matmul(A, v).permute(0, 2, 1, 3).contiguous().view(R, S, C)
    return _lin(sd, p + ".out_fc", out), A


def ray_transformer(sd, p, x):
    """Transformer (transformer_network.py:202-218). Returns x', A, x_after_attn."""
    o, A = ray_attention(sd, p + ".attn", _ln(sd, p + ".attn_norm", x, 1e-6))
    x = x + o
    mid = x
    x = x + feed_forward(sd, p + ".ff", _ln(sd, p + ".ff_norm", x, 1e-6))
    return x, A, mid


def encodings(pts, ray_d):
    """pos/view encodings (transformer_network.py:297-306): [R,S,63], [R,63]."""
    vd = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
    vd = embed(vd.reshape(-1, 3).float())
    pe = embed(pts.reshape(-1, 3).float()).reshape(pts.shape[:-1] + (-1,))
    return pe, vd


def forward(sd, rgb_feat, deep_sem_feat, ray_diff, mask, pts, ray_d, trans_depth=8,
            intermediates=None):
    """Literal GNT.forward with ret_alpha=True (transformer_network.py:295-338).

    Returns (cat[rgb(3), weights(S)] [R,3+S], sem_feat [R,512], sem_feat_3d [R,S,512]).
    deep_sem_feat may be None -> the two semantic outputs are None.
    If `intermediates` is a dict it is filled with per-stage tensors for kernel-level tests.
    """
    pe, ve = encodings(pts, ray_d)
    ve_b = ve[:, None].expand(pe.shape)
    feat = _lin(sd, "rgbfeat_fc.2", F.relu(_lin(sd, "rgbfeat_fc.0", rgb_feat)))
    q = feat.max(dim=2)[0]
    rec = intermediates
    if rec is not None:
        rec.update(pos_enc=pe, view_enc=ve, feat=feat, q0=q, attn2d=[], A=[], q_view_attn=[],
                   q_view=[], q_pos=[], q_ray_attn=[], q_ray=[])
    sem2d, sem3d = [], []
    A = None
    for i in range(trans_depth):
        q, attn, mid = view_transformer(sd, "view_crosstrans.%d" % i, q, feat, ray_diff, mask)
        if rec is not None:
            rec["attn2d"].append(attn)
            rec["q_view_attn"].append(mid)
            rec["q_view"].append(q)
        view_sem = None
        if deep_sem_feat is not None:
            view_sem = (deep_sem_feat * attn.repeat_interleave(8, dim=-1)).sum(dim=2)  # (:90)
        if i % 2 == 0:
            q = torch.cat((q, pe, ve_b), dim=-1)
            q = _lin(sd, "q_fcs.%d.2" % i, F.relu(_lin(sd, "q_fcs.%d.0" % i, q)))
            if rec is not None:
                rec["q_pos"].append(q)
        q, A, mid = ray_transformer(sd, "view_selftrans.%d" % i, q)
        if rec is not None:
            rec["A"].append(A)
            rec["q_ray_attn"].append(mid)
            rec["q_ray"].append(q)
        if view_sem is not None:
            R, S, D = view_sem.shape
            H = A.shape[1]
            vs = view_sem.view(R, S, H, -1).permute(0, 2, 1, 3)
            rs = torch.matmul(A, vs).permute(0, 2, 1, 3).contiguous().view(R, S, D)  # (:180-184)
            sem2d.append(rs)
            sem3d.append(view_sem)
    h = _ln(sd, "norm", q, 1e-5)
    rgb = _lin(sd, "rgb_fc", h.mean(dim=1))
    weights = A.mean(dim=1)[:, 0]  # head-mean attention, query row 0 only (:216)
    out = torch.cat([rgb, weights], dim=1)
    if deep_sem_feat is None:
        return out, None, None
    sem_feat_3d = torch.stack(sem3d, dim=0).sum(dim=0)
    pre = torch.stack(sem2d, dim=0).sum(dim=0).mean(dim=1)
    if rec is not None:
        rec["pre_adaptor"] = pre
    sem_feat = F.relu(_lin(sd, "semantic_adaptor.0", pre))
    return out, sem_feat, sem_feat_3d


def semantic_weights(attn2d_layers, A_layers):
    """W3d[r,s,v,g] = sum_l attn_l ;  W2d[r,s,v,g] = sum_l abar_l[r, g//16, s] * attn_l
    with abar_l[r,h,s'] = mean_s A_l[r,h,s,s']."""
    W3d = torch.zeros_like(attn2d_layers[0])
    W2d = torch.zeros_like(attn2d_layers[0])
    C = W3d.shape[-1]
    for attn, A in zip(attn2d_layers, A_layers):
        H = A.shape[1]
        abar = A.mean(dim=2)  # [R,H,S']
        per_ch = abar.permute(0, 2, 1).repeat_interleave(C // H, dim=-1)  # [R,S,C]
        W3d = W3d + attn
        W2d = W2d + attn * per_ch[:, :, None, :]
    return W3d, W2d


def fused_semantics(sd, deep_sem_feat, W3d, W2d):
    """sem_feat_3d = sum_v deep*W3d ; sem_feat = relu(adaptor(sum_{s,v} deep*W2d))."""
    rep = deep_sem_feat.shape[-1] // W3d.shape[-1]
    sem3d = (deep_sem_feat * W3d.repeat_interleave(rep, dim=-1)).sum(dim=2)
    pre = (deep_sem_feat * W2d.repeat_interleave(rep, dim=-1)).sum(dim=(1, 2))
    return F.relu(_lin(sd, "semantic_adaptor.0", pre)), sem3d, pre


def flops_per_ray(S, V, trans_depth=8):
    """Algorithmic GNT flops (MAC = 2) per ray for a pass of S samples."""
    per_sample = (V * 12672 + trans_depth * (V * 20544 + 81920 + 98304 + 1280 * S)
                  + (trans_depth // 2) * 32512)
    return S * per_sample + 524288
