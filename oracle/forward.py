"""Oracle: the whole hot path -- DLA_PlanAWindow_l3new forward + heads + sigmoid + decode.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates
  * DLA_PlanAWindow_l3new.imgpre2feats   sgtapose/lib/model/networks/dla.py:1505-1554
  * BaseModelPlanA.forward (heads)       sgtapose/lib/model/networks/base_model.py:170-200
  * SGTADetector.process                 sgtapose/lib/sgta_detector.py:881-927 (forward, sigmoid_, decode)
"""
import torch
import torch.nn.functional as F

from .dla34 import dla34_backbone, dla_up_forward, ida_up_forward
from .temporal import topk_index, window_coords, window_gather, encoder_forward, substitute_tokens
from .decode import dream_decode

SCALE_LIST = (4, 2, 1, 1 / 2, 1 / 4, 1 / 8)       # dla.py:1493
KS_DEFAULT = (12, 6, 3, 1, 1, 1)                  # opts_parallel.py:131-142 (ks1..ks6)
HEADS = ("hm", "reg", "tracking")                  # 'wh' is skipped: base_model.py:114-115, :194-195


def temporal_levels(x_pre, x_cur, pre_hm_cls, repro_hm_cls, sd, ks=KS_DEFAULT, use_pos=True,
                    levels=range(6)):
    """Levels 0-2: windowed cross-attention encoder then cat-MLP substitution; levels 3-5: the
    previous frame's tokens go straight into the cat-MLP (dla.py:1513-1543).  K_i = 1 everywhere."""
    pre_idx = topk_index(pre_hm_cls)
    cur_idx = topk_index(repro_hm_cls)
    out = list(x_cur)
    fids = {}
    for i in levels:
        pre_f, cur_f = x_pre[i], x_cur[i]
        H, W = cur_f.shape[2:]
        pre_fid = window_coords(pre_idx, SCALE_LIST[i], ks[i], H, W)
        cur_fid = window_coords(cur_idx, SCALE_LIST[i], ks[i], H, W)
        key = window_gather(pre_f, pre_fid)
        if i <= 2:
            qry = window_gather(cur_f, cur_fid)
            tok = encoder_forward(qry, key, key, sd, "transformer.%d" % i, use_pos=use_pos)
        else:
            tok = key
        out[i] = substitute_tokens(tok, cur_f, cur_fid, sd, "cat_layer.%d" % i)
        fids[i] = (pre_fid, cur_fid)
    return out, fids


def head_forward(feat, sd, name):
    """conv3x3(64->256, bias) + ReLU + conv1x1(256->classes, bias)  (base_model.py:121-135)."""
    h = F.relu(F.conv2d(feat, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1))
    return F.conv2d(h, sd[name + ".2.weight"], sd[name + ".2.bias"])


def plan_a_window_forward(sd, x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls,
                          ks=KS_DEFAULT, use_pos=True, return_feats=False):
    """Returns {'hm': logits [B,7,h,w], 'reg': [B,2,h,w], 'tracking': [B,2,h,w]} (what model(...)[-1] is)."""
    with torch.no_grad():
        x_pre = dla34_backbone(pre_img, pre_hm, sd)
        x_cur = dla34_backbone(x, repro_hm, sd)
        x_out, fids = temporal_levels(x_pre, x_cur, pre_hm_cls, repro_hm_cls, sd, ks, use_pos)
        ups = dla_up_forward(x_out, sd)
        y = [ups[0].clone(), ups[1].clone(), ups[2].clone()]
        ida_up_forward(y, 0, 3, sd, "ida_up")
        feat = y[-1]
        out = {h: head_forward(feat, sd, h) for h in HEADS}
        if return_feats:
            out["_feat"] = feat
            out["_x_out"] = x_out
            out["_fids"] = fids
        return out


# ---- DLA_PlanAWindow_Three: two previous frames ------------------------------------------
HEADS_THREE = ("hm", "reg", "wh", "tracking")      # BaseModelPlanA_Three keeps 'wh' (base_model.py:213-262)


def temporal_levels_three(x_ppre, x_pre, x_cur, ppre_hm_cls, pre_hm_cls, repro_hm_cls, sd, ks=KS_DEFAULT,
                          use_pos=True, shared_ca=False):
    """DLA_PlanAWindow_Three.imgpre2feats, temporal part (dla.py:1411-1447).  Levels 0-2: the current-frame window
    tokens attend to the windows of BOTH previous frames (one encoder per frame unless ``shared_ca``), the two
    outputs are concatenated (2C) and substituted through cat_layer (3C -> 6C -> C).  Levels 3-5: cat_layer runs
    on EVERY pixel of cat[pre, pre, cur] -- the reference assigns ``ppre_f = pre_feats`` (:1435), kept."""
    ppre_idx, pre_idx, cur_idx = topk_index(ppre_hm_cls), topk_index(pre_hm_cls), topk_index(repro_hm_cls)
    out = list(x_cur)
    for i in range(6):
        ppre_f, pre_f, cur_f = x_ppre[i], x_pre[i], x_cur[i]
        H, W = cur_f.shape[2:]
        if i <= 2:
            scale = (4, 2, 1)[i]
            key_pp = window_gather(ppre_f, window_coords(ppre_idx, scale, ks[i], H, W))
            key_p = window_gather(pre_f, window_coords(pre_idx, scale, ks[i], H, W))
            cur_fid = window_coords(cur_idx, scale, ks[i], H, W)
            qry = window_gather(cur_f, cur_fid)
            tf_pp = "prev_transformer.%d" % i if shared_ca else "pprev_transformer.%d" % i
            o_pp = encoder_forward(qry, key_pp, key_pp, sd, tf_pp, use_pos=use_pos)
            o_p = encoder_forward(qry, key_p, key_p, sd, "prev_transformer.%d" % i, use_pos=use_pos)
            out[i] = substitute_tokens(torch.cat([o_pp, o_p], dim=-1), cur_f, cur_fid, sd, "cat_layer.%d" % i)
        else:
            p = "cat_layer.%d" % i
            pf, cf = pre_f.permute(0, 2, 3, 1).contiguous(), cur_f.permute(0, 2, 3, 1).contiguous()
            z = torch.cat([pf, pf, cf], dim=-1)
            z = F.linear(F.relu(F.linear(z, sd[p + ".0.weight"], sd[p + ".0.bias"])), sd[p + ".2.weight"], sd[p + ".2.bias"])
            out[i] = z.permute(0, 3, 1, 2).contiguous()
    return out


def plan_a_window_three_forward(sd, x, ppre_img, pre_img, ppre_hm, pre_hm, repro_hm, ppre_hm_cls, pre_hm_cls,
                                repro_hm_cls, ks=KS_DEFAULT, use_pos=True, shared_ca=False):
    """BaseModelPlanA_Three.forward (base_model.py:267-299) over DLA_PlanAWindow_Three.imgpre2feats (dla.py:1400-1456)
    -> {'hm','reg','wh','tracking'} logits."""
    with torch.no_grad():
        x_ppre = dla34_backbone(ppre_img, ppre_hm, sd)
        x_pre = dla34_backbone(pre_img, pre_hm, sd)
        x_cur = dla34_backbone(x, repro_hm, sd)
        x_out = temporal_levels_three(x_ppre, x_pre, x_cur, ppre_hm_cls, pre_hm_cls, repro_hm_cls, sd, ks, use_pos,
                                      shared_ca)
        ups = dla_up_forward(x_out, sd)
        y = [ups[0].clone(), ups[1].clone(), ups[2].clone()]
        ida_up_forward(y, 0, 3, sd, "ida_up")
        return {h: head_forward(y[-1], sd, h) for h in HEADS_THREE}


# ---- T-frame generalisation (BASELINE configs[4]; no reference beyond T = 3) ------------------------------------------------
def seq_keys_from_three(sd):
    """state_dict of DLA_PlanAWindow_Three -> the key names of the T = 3 instance of the generalised model."""
    out = {}
    for k, v in sd.items():
        k2 = k.replace("pprev_transformer.", "frame_transformer.0.") if k.startswith("pprev_transformer.") else \
            (k.replace("prev_transformer.", "frame_transformer.1.") if k.startswith("prev_transformer.") else k)
        out[k2] = v
    return out


def plan_a_window_seq_forward(sd, x, pre_imgs, pre_hms, repro_hm, pre_hm_clss, repro_hm_cls, ks=KS_DEFAULT,
                              use_pos=True, legacy_slip=False):
    """DLA_PlanAWindow_Three.imgpre2feats (dla.py:1400-1456) generalised to T - 1 = len(pre_imgs) previous frames
    (oldest first): one encoder stream per previous frame against the same current-frame queries at levels 0-2
    (``frame_transformer.k.i``), outputs concatenated with the current tokens -> cat_layer (T*C -> 2T*C -> C); levels 3-5:
    cat_layer on every pixel of cat[frame_0 .. frame_{T-2}, cur].  ``legacy_slip`` (T = 3 only) = the reference's
    ``ppre_f = pre_feats`` at levels 3-5 (dla.py:1435): with it and ``seq_keys_from_three`` this IS
    plan_a_window_three_forward, which is pinned against the reference.  -> {'hm','reg','wh','tracking'} logits."""
    n_prev = len(pre_imgs)
    assert n_prev >= 2 and len(pre_hms) == n_prev and len(pre_hm_clss) == n_prev and not (legacy_slip and n_prev != 2)
    with torch.no_grad():
        xs_prev = [dla34_backbone(pre_imgs[k], pre_hms[k], sd) for k in range(n_prev)]
        x_cur = dla34_backbone(x, repro_hm, sd)
        idx_prev = [topk_index(c) for c in pre_hm_clss]
        cur_idx = topk_index(repro_hm_cls)
        out = list(x_cur)
        for i in range(6):
            cur_f = x_cur[i]
            H, W = cur_f.shape[2:]
            if i <= 2:
                cur_fid = window_coords(cur_idx, SCALE_LIST[i], ks[i], H, W)
                qry = window_gather(cur_f, cur_fid)
                toks = []
                for k in range(n_prev):
                    key = window_gather(xs_prev[k][i], window_coords(idx_prev[k], SCALE_LIST[i], ks[i], H, W))
                    toks.append(encoder_forward(qry, key, key, sd, "frame_transformer.%d.%d" % (k, i), use_pos=use_pos))
                out[i] = substitute_tokens(torch.cat(toks, dim=-1), cur_f, cur_fid, sd, "cat_layer.%d" % i)
            else:
                p = "cat_layer.%d" % i
                prev = [xs_prev[-1][i]] * n_prev if legacy_slip else [xs_prev[k][i] for k in range(n_prev)]
                z = torch.cat([f.permute(0, 2, 3, 1) for f in prev] + [cur_f.permute(0, 2, 3, 1)], dim=-1).contiguous()
                z = F.linear(F.relu(F.linear(z, sd[p + ".0.weight"], sd[p + ".0.bias"])), sd[p + ".2.weight"], sd[p + ".2.bias"])
                out[i] = z.permute(0, 3, 1, 2).contiguous()
        ups = dla_up_forward(out, sd)
        y = [ups[0].clone(), ups[1].clone(), ups[2].clone()]
        ida_up_forward(y, 0, 3, sd, "ida_up")
        return {h: head_forward(y[-1], sd, h) for h in HEADS_THREE}


def process_reference(sd, x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls, K=7, **kw):
    """SGTADetector.process for phase PlanA_win: forward, in-place sigmoid on 'hm', decode.
    Returns (output dict with post-sigmoid hm, dets dict of numpy arrays)."""
    out = plan_a_window_forward(sd, x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls, **kw)
    out["hm"] = out["hm"].sigmoid_()
    dets = dream_decode(out, K=K)
    return out, {k: v.detach().cpu().numpy() for k, v in dets.items()}
