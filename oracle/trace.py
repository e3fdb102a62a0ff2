"""ORACLE — test infrastructure only.  Forward of oracle/wav2vec2.py with every intermediate the HIP
model can dump (paa_model_debug_read) kept, and their gradients retained, so a GPU mismatch can be
localised to the first diverging stage."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import wav2vec2 as W


def forward_trace(sd, a, x, labels, direction=1.0):
    """Returns (loss, dict name -> tensor in (B, T, C) layout), after backward (grads in ``.grad``)."""
    tr = {}

    def keep(name, t):
        if t.requires_grad:
            t.retain_grad()
        tr[name] = t
        return t

    h = x[:, None, :]
    for i, s in enumerate(a.conv_stride):
        pre = f"wav2vec2.feature_extractor.conv_layers.{i}"
        h = F.conv1d(h, sd[f"{pre}.conv.weight"], sd.get(f"{pre}.conv.bias"), stride=s)
        if a.feat_extract_norm == "layer" and i > 0:
            keep(f"conv{i}.cv", h)
        c = h.shape[1]
        if a.feat_extract_norm == "group" and i == 0:
            h = F.group_norm(h, c, sd[f"{pre}.layer_norm.weight"], sd[f"{pre}.layer_norm.bias"], eps=1e-5)
        elif a.feat_extract_norm == "layer":
            h = F.layer_norm(h.transpose(1, 2), (c,), sd[f"{pre}.layer_norm.weight"], sd[f"{pre}.layer_norm.bias"],
                             eps=1e-5).transpose(1, 2)
        keep(f"conv{i}.pre", h)
        h = keep(f"conv{i}.act", F.gelu(h))
    f = h.transpose(1, 2)
    C = f.shape[-1]
    fn = keep("fn", F.layer_norm(f, (C,), sd["wav2vec2.feature_projection.layer_norm.weight"],
                                 sd["wav2vec2.feature_projection.layer_norm.bias"], eps=a.layer_norm_eps))
    h0 = keep("h0", F.linear(fn, sd["wav2vec2.feature_projection.projection.weight"],
                             sd["wav2vec2.feature_projection.projection.bias"]))
    K = a.num_conv_pos_embeddings
    pc = "wav2vec2.encoder.pos_conv_embed.conv"
    pos = F.conv1d(h0.transpose(1, 2), W.pos_conv_weight(sd), sd[f"{pc}.bias"], padding=K // 2,
                   groups=a.num_conv_pos_embedding_groups)
    if K % 2 == 0:
        pos = pos[:, :, :-1]
    keep("pos_pre", pos.transpose(1, 2))
    hsum = keep("hsum", h0 + F.gelu(pos).transpose(1, 2))
    H = a.hidden_size

    def ln(t, name):
        return F.layer_norm(t, (H,), sd[f"{name}.weight"], sd[f"{name}.bias"], eps=a.layer_norm_eps)

    hcur = hsum if a.do_stable_layer_norm else ln(hsum, "wav2vec2.encoder.layer_norm")
    B, T, _ = hcur.shape
    nh, hd = a.num_attention_heads, a.head_dim
    for l in range(a.num_hidden_layers):
        pre = f"wav2vec2.encoder.layers.{l}"
        xin = hcur
        if a.do_stable_layer_norm:
            keep(f"L{l}.ln1_in", xin)
            att_in = ln(xin, f"{pre}.layer_norm")
        else:
            att_in = xin
        q = F.linear(att_in, sd[f"{pre}.attention.q_proj.weight"], sd[f"{pre}.attention.q_proj.bias"])
        k = F.linear(att_in, sd[f"{pre}.attention.k_proj.weight"], sd[f"{pre}.attention.k_proj.bias"])
        v = F.linear(att_in, sd[f"{pre}.attention.v_proj.weight"], sd[f"{pre}.attention.v_proj.bias"])
        keep(f"L{l}.qkv", torch.cat([q, k, v], -1))
        qh, kh, vh = (t.view(B, T, nh, hd).transpose(1, 2) for t in (q, k, v))
        P = keep(f"L{l}.P", torch.softmax((qh @ kh.transpose(-1, -2)) * (hd ** -0.5), dim=-1))
        o = (P @ vh).transpose(1, 2).reshape(B, T, H)
        att = F.linear(o, sd[f"{pre}.attention.out_proj.weight"], sd[f"{pre}.attention.out_proj.bias"])
        r1 = xin + att
        if a.do_stable_layer_norm:
            keep(f"L{l}.ln2_in", r1)
            y = ln(r1, f"{pre}.final_layer_norm")
        else:
            keep(f"L{l}.ln1_in", r1)
            y = ln(r1, f"{pre}.layer_norm")
        fpre = keep(f"L{l}.fpre", F.linear(y, sd[f"{pre}.feed_forward.intermediate_dense.weight"],
                                           sd[f"{pre}.feed_forward.intermediate_dense.bias"]))
        ff = F.linear(F.gelu(fpre), sd[f"{pre}.feed_forward.output_dense.weight"], sd[f"{pre}.feed_forward.output_dense.bias"])
        if a.do_stable_layer_norm:
            hcur = r1 + ff
        else:
            r2 = keep(f"L{l}.ln2_in", y + ff)
            hcur = ln(r2, f"{pre}.final_layer_norm")
    if a.do_stable_layer_norm:
        hcur = ln(hcur, "wav2vec2.encoder.layer_norm")
    keep("xfinal", hcur)
    logits = keep("logits", F.linear(hcur, sd["lm_head.weight"], sd["lm_head.bias"]))
    loss = W.ctc_loss_of(a, logits, labels)
    (direction * loss).backward()
    return loss.detach(), tr
