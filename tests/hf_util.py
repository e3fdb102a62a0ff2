"""Test helper: a HuggingFace ``Wav2Vec2ForCTC`` (transformers is a third-party dependency of the reference,
src/training_utils/build.py:229-230) built from an ``arch.Wav2Vec2Arch`` and a state dict keyed like its own — no hub access,
no reference import."""
import torch


def hf_config(a):
    from transformers import Wav2Vec2Config
    return Wav2Vec2Config(vocab_size=a.vocab_size, hidden_size=a.hidden_size, num_hidden_layers=a.num_hidden_layers,
                          num_attention_heads=a.num_attention_heads, intermediate_size=a.intermediate_size,
                          conv_dim=list(a.conv_dim), conv_kernel=list(a.conv_kernel), conv_stride=list(a.conv_stride),
                          conv_bias=a.conv_bias, feat_extract_norm=a.feat_extract_norm,
                          num_conv_pos_embeddings=a.num_conv_pos_embeddings,
                          num_conv_pos_embedding_groups=a.num_conv_pos_embedding_groups,
                          do_stable_layer_norm=a.do_stable_layer_norm, layer_norm_eps=a.layer_norm_eps,
                          pad_token_id=a.pad_token_id)


def hf_model(a, sd_np):
    from transformers import Wav2Vec2ForCTC
    m = Wav2Vec2ForCTC(hf_config(a)).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("masked_spec_embed" in k for k in missing), missing
    return m
