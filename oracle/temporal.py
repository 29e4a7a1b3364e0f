"""Oracle: structure-prior-guided temporal cross-attention over keypoint-centred windows.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates
  * get_topk_index                     sgtapose/lib/model/networks/dla.py:898-913
  * get_topk_features_scale            dla.py:915-968   (fp32 coordinate math, truncation, x-major tokens)
  * MHCA_ein.forward                   dla.py:868-887
  * TransformerEncoderLayer / Encoder  dla.py:728-743, :791-803 (3 iterations of ONE shared layer, :788-789)
  * substitute_topk_features_scale     dla.py:1006-1018 (duplicate targets: last token wins on CPU)
"""
import math

import torch
import torch.nn.functional as F


def topk_index(hm_cls):
    """Per (b, class) arg-max (K=1) of a prior map [B,K,h,w] -> float32 [B,K,2] = (x, y).
    torch.topk on CPU: ties / all-zero maps give the lowest flat index."""
    B, K, h, w = hm_cls.shape
    flat = torch.topk(hm_cls.flatten(2), 1, dim=-1)[1].view(B, K)
    out = torch.zeros(B, K, 2)
    out[:, :, 0] = flat % w
    out[:, :, 1] = flat // w
    return out


def window_coords(idx_xy, scale, ks, H, W):
    """idx_xy float32 [B,K,2]; returns feat_id int64 [B, K*w*w] with w = 1 + 2*(ks//2).

    Everything up to the final truncation is float32, exactly as dla.py:943-957:
    coord = idx*scale + offset, clamp to [0, H-1] (BOTH x and y use H-1, dla.py:948-950),
    id = trunc(y*W + x).  Token order inside a window: x offset is the slow index, y the fast one."""
    r = ks // 2
    off = torch.arange(-r, r + 1, dtype=torch.float32)
    w = off.numel()
    ox = off.view(w, 1).expand(w, w).reshape(-1)      # slow
    oy = off.view(1, w).expand(w, w).reshape(-1)      # fast
    x = idx_xy[:, :, None, 0] * scale + ox            # [B,K,N] float32
    y = idx_xy[:, :, None, 1] * scale + oy
    x = torch.clamp(x, min=0, max=H - 1)
    y = torch.clamp(y, min=0, max=H - 1)
    fid = (y * W + x).reshape(idx_xy.shape[0], -1).type(torch.long)
    return fid


def window_gather(feats, fid):
    """feats [B,C,H,W], fid [B,n] -> tokens [B,n,C] (channel-last), straight from NCHW."""
    B, C, H, W = feats.shape
    flat = feats.reshape(B, C, H * W)
    tok = torch.gather(flat, 2, fid.view(B, 1, -1).expand(B, C, fid.shape[1]))
    return tok.permute(0, 2, 1).contiguous()


def mhca_forward(q, k, v, sd, p, n_heads=8, use_pos=True):
    """MHCA_ein (dla.py:868-887): bias-free Q/K/V projections C->2C, 8 heads,
    energy = QK^T / sqrt(d) + pos_embed[h], softmax over keys, AV, fc (2C->C, bias)."""
    B, n, C = q.shape
    Q = F.linear(q, sd[p + ".w_q.weight"])
    K = F.linear(k, sd[p + ".w_k.weight"])
    V = F.linear(v, sd[p + ".w_v.weight"])
    hid = Q.shape[-1]
    d = hid // n_heads
    Q = Q.view(B, n, n_heads, d).permute(0, 2, 1, 3)
    K = K.view(B, n, n_heads, d).permute(0, 2, 1, 3)
    V = V.view(B, n, n_heads, d).permute(0, 2, 1, 3)
    energy = torch.matmul(Q, K.transpose(-1, -2)) / math.sqrt(d)
    if use_pos:
        energy = energy + sd[p + ".pos_embed"]
    attn = torch.softmax(energy, dim=-1)
    out = torch.matmul(attn, V).permute(0, 2, 1, 3).reshape(B, n, hid)
    return F.linear(out, sd[p + ".fc.weight"], sd[p + ".fc.bias"])


def encoder_layer_forward(q, k, v, sd, p, use_pos=True):
    """TransformerEncoderLayer.forward (dla.py:734-743), dropout = identity in eval mode:
    q = LN1(attn(q,k,v) + q);  q = LN3(q + W2 relu(W1 q))."""
    C = q.shape[-1]
    t = mhca_forward(q, k, v, sd, p + ".cross_attn", use_pos=use_pos)
    q = F.layer_norm(t + q, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    t = F.linear(F.relu(F.linear(q, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return F.layer_norm(q + t, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)


def encoder_forward(q, k, v, sd, p, num_layers=3, use_pos=True):
    """TransformerEncoder.forward (dla.py:798-803).  layers.0/1/2 alias one module
    (dla.py:788-789) so ``layers.0`` is used for every iteration; key/value never change."""
    out = q
    for _ in range(num_layers):
        out = encoder_layer_forward(out, k, v, sd, p + ".layers.0", use_pos=use_pos)
    return out


def substitute_tokens(tokens, cur_feats, fid, sd, p):
    """substitute_topk_features_scale (dla.py:1006-1018):
    cur[b, :, fid] = W2 relu(W1 cat[tokens, cur[b, :, fid]]); other pixels unchanged.
    Duplicate fids: the LAST token (flattened order) wins, as torch's CPU index_put does."""
    B, C, H, W = cur_feats.shape
    cur_tok = window_gather(cur_feats, fid)
    z = torch.cat([tokens, cur_tok], dim=-1)
    z = F.linear(F.relu(F.linear(z, sd[p + ".0.weight"], sd[p + ".0.bias"])),
                 sd[p + ".2.weight"], sd[p + ".2.bias"])
    out = cur_feats.reshape(B, C, H * W).clone()
    for b in range(B):
        f = fid[b]
        n = f.numel()
        # winner per destination pixel = largest token index writing to it
        owner = torch.full((H * W,), -1, dtype=torch.long)
        owner.scatter_reduce_(0, f, torch.arange(n), reduce="amax", include_self=True)
        win = owner[f] == torch.arange(n)
        out[b][:, f[win]] = z[b][win].t()
    return out.view(B, C, H, W)
