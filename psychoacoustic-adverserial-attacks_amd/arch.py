"""Wav2Vec2 architecture description + rule-generated weights for the PGD step.

Field names follow HuggingFace ``Wav2Vec2Config`` (transformers, third-party dependency of the
reference: src/training_utils/build.py:229-230 loads ``Wav2Vec2ForCTC`` by name); state-dict key
names follow ``Wav2Vec2ForCTC.state_dict()`` so that a local checkpoint can be mapped 1:1.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import synth


@dataclass(frozen=True)
class Wav2Vec2Arch:
    conv_dim: tuple = (512,) * 7
    conv_kernel: tuple = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: tuple = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"          # "group" (base) | "layer" (large-lv60)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    do_stable_layer_norm: bool = False
    layer_norm_eps: float = 1e-5
    vocab_size: int = 32
    pad_token_id: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def feat_lengths(self, length: int) -> list:
        """Conv output lengths, floor((L - k) / s) + 1 per layer (HF ``_get_feat_extract_output_lengths``)."""
        out = []
        for k, s in zip(self.conv_kernel, self.conv_stride):
            length = (length - k) // s + 1
            out.append(length)
        return out

    def fwd_flops_per_clip(self, length: int) -> float:
        """Algorithmic forward FLOPs (2·MAC) per clip: conv + linear + attention."""
        t = self.feat_lengths(length)
        cin = (1,) + self.conv_dim[:-1]
        fl = sum(2.0 * t[i] * self.conv_dim[i] * cin[i] * self.conv_kernel[i] for i in range(len(t)))
        T, H, F = t[-1], self.hidden_size, self.intermediate_size
        fl += 2.0 * T * self.conv_dim[-1] * H
        fl += 2.0 * T * H * (H // self.num_conv_pos_embedding_groups) * self.num_conv_pos_embeddings
        fl += self.num_hidden_layers * (2.0 * T * (4 * H * H + 2 * H * F) + 4.0 * T * T * H)
        fl += 2.0 * T * H * self.vocab_size
        return fl


BASE = Wav2Vec2Arch()
LARGE_LV60 = Wav2Vec2Arch(conv_bias=True, feat_extract_norm="layer", hidden_size=1024, num_hidden_layers=24,
                           num_attention_heads=16, intermediate_size=4096, do_stable_layer_norm=True)


def tiny(feat_extract_norm: str = "group", stable: bool = False, **kw) -> Wav2Vec2Arch:
    """Small config used by the parity tests and goldens."""
    d = dict(conv_dim=(32,) * 7, conv_bias=(feat_extract_norm == "layer"), feat_extract_norm=feat_extract_norm,
             hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
             num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, do_stable_layer_norm=stable)
    d.update(kw)
    return Wav2Vec2Arch(**d)


def state_dict_spec(a: Wav2Vec2Arch) -> list:
    """[(hf_key, shape, kind)] for every tensor the PGD step reads."""
    spec = []
    cin = (1,) + tuple(a.conv_dim[:-1])
    for i, (co, k) in enumerate(zip(a.conv_dim, a.conv_kernel)):
        pre = f"wav2vec2.feature_extractor.conv_layers.{i}"
        spec.append((f"{pre}.conv.weight", (co, cin[i], k), "w"))
        if a.conv_bias:
            spec.append((f"{pre}.conv.bias", (co,), "b"))
        if (a.feat_extract_norm == "group" and i == 0) or a.feat_extract_norm == "layer":
            spec.append((f"{pre}.layer_norm.weight", (co,), "g"))
            spec.append((f"{pre}.layer_norm.bias", (co,), "b"))
    H, F, C = a.hidden_size, a.intermediate_size, a.conv_dim[-1]
    spec += [("wav2vec2.feature_projection.layer_norm.weight", (C,), "g"),
             ("wav2vec2.feature_projection.layer_norm.bias", (C,), "b"),
             ("wav2vec2.feature_projection.projection.weight", (H, C), "w"),
             ("wav2vec2.feature_projection.projection.bias", (H,), "b")]
    K, G = a.num_conv_pos_embeddings, a.num_conv_pos_embedding_groups
    pc = "wav2vec2.encoder.pos_conv_embed.conv"
    spec += [(f"{pc}.bias", (H,), "b"),
             (f"{pc}.parametrizations.weight.original0", (1, 1, K), "g"),
             (f"{pc}.parametrizations.weight.original1", (H, H // G, K), "w")]
    spec += [("wav2vec2.encoder.layer_norm.weight", (H,), "g"), ("wav2vec2.encoder.layer_norm.bias", (H,), "b")]
    for l in range(a.num_hidden_layers):
        pre = f"wav2vec2.encoder.layers.{l}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            spec += [(f"{pre}.attention.{n}.weight", (H, H), "w"), (f"{pre}.attention.{n}.bias", (H,), "b")]
        spec += [(f"{pre}.layer_norm.weight", (H,), "g"), (f"{pre}.layer_norm.bias", (H,), "b"),
                 (f"{pre}.feed_forward.intermediate_dense.weight", (F, H), "w"),
                 (f"{pre}.feed_forward.intermediate_dense.bias", (F,), "b"),
                 (f"{pre}.feed_forward.output_dense.weight", (H, F), "w"),
                 (f"{pre}.feed_forward.output_dense.bias", (H,), "b"),
                 (f"{pre}.final_layer_norm.weight", (H,), "g"), (f"{pre}.final_layer_norm.bias", (H,), "b")]
    spec += [("lm_head.weight", (a.vocab_size, H), "w"), ("lm_head.bias", (a.vocab_size,), "b")]
    return spec


def rule_weights(a: Wav2Vec2Arch, seed: int = 0) -> dict:
    """Deterministic random-init weights keyed by HF state-dict names (numpy float32).

    Rule: weights ~ N(0, gain / fan_in) from the counter-based generator in ``synth`` (keyed by
    tensor name, so independent of tensor order); norm gains 1 + 0.1·N(0,1); biases 0.02·N(0,1).
    There is no pretrained checkpoint offline, so benches and goldens use these.
    """
    out = {}
    for key, shape, kind in state_dict_spec(a):
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            gain = 2.0 if "feature_extractor" in key or "pos_conv" in key or "intermediate_dense" in key else 1.0
            out[key] = synth.tensor_normal(key, shape, std=float(np.sqrt(gain / fan_in)), seed=seed)
        elif kind == "g":
            out[key] = (1.0 + 0.1 * synth.tensor_normal(key, shape, seed=seed)).astype(np.float32)
        else:
            out[key] = synth.tensor_normal(key, shape, std=0.02, seed=seed)
    return out


def pos_conv_weight(sd: dict) -> np.ndarray:
    """Fold torch weight-norm (dim=2): w = g · v / ‖v‖ with the norm over dims (0, 1) per tap
    (HF Wav2Vec2PositionalConvEmbedding)."""
    pc = "wav2vec2.encoder.pos_conv_embed.conv"
    if f"{pc}.weight" in sd:
        return np.asarray(sd[f"{pc}.weight"], dtype=np.float32)
    # torch >= 2.1 parametrization keys, or the legacy weight_g / weight_v pair of the published 960h checkpoint files
    kg = f"{pc}.parametrizations.weight.original0" if f"{pc}.parametrizations.weight.original0" in sd else f"{pc}.weight_g"
    kv = f"{pc}.parametrizations.weight.original1" if f"{pc}.parametrizations.weight.original1" in sd else f"{pc}.weight_v"
    if kg not in sd or kv not in sd:
        raise KeyError(f"positional conv weight: none of '{pc}.weight', '...parametrizations.weight.original0/1', "
                       f"'{pc}.weight_g/weight_v' found in the state dict")
    g = np.asarray(sd[kg], dtype=np.float64)
    v = np.asarray(sd[kv], dtype=np.float64)
    nrm = np.sqrt((v * v).sum(axis=(0, 1), keepdims=True))
    return (g * v / nrm).astype(np.float32)
