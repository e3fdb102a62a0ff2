"""Host-logic checks (no GPU): the GEMM descriptors csrc/model.hip builds for the strided convolutions,
their input gradients and the grouped positional convolution, emulated in numpy (tests/gemm_ref.py)
on the packed weights of paa_amd.model.pack_weights, reproduce torch's conv1d and its backward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gemm_ref import emulate, padded_rows
from paa_amd import arch as A, synth
from paa_amd.model import pack_weights

GUARD = 8


@pytest.mark.parametrize("k,s", [(3, 2), (2, 2)])
def test_conv_as_gemm_and_dgrad(k, s):
    a = A.tiny(conv_kernel=(10, k, 3, 3, 3, 2, 2), conv_stride=(5, s, 2, 2, 2, 2, 2))
    sd = A.rule_weights(a)
    pk = pack_weights(a, sd)
    B, L = 3, 4000
    T, P = padded_rows(a, L)
    C = a.conv_dim[0]
    rng = np.random.default_rng(0)
    x = rng.normal(size=(B, C, T[0])).astype(np.float32)                 # conv0 output (channel-first for torch)
    W = torch.from_numpy(sd["wav2vec2.feature_extractor.conv_layers.1.conv.weight"])
    xt = torch.from_numpy(x).requires_grad_(True)
    y = F.conv1d(xt, W, stride=s)
    assert y.shape[-1] == T[1]
    gy = rng.normal(size=tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(gy))
    # channel-last padded layout with zero pad rows + slack
    act = np.zeros((B * P[0] + GUARD, C), np.float32)
    act.reshape(-1)[: B * P[0] * C].reshape(B, P[0], C)[:, :T[0]] = x.transpose(0, 2, 1)
    out = np.full((B * P[1] + GUARD, C), np.nan, np.float32)
    d = dict(M=B * P[1], N=C, K=k * C, lda=s * C, ldb=k * C, ldc=C, row_period=P[1], row_valid=T[1])
    emulate(d, act.reshape(-1), pk["c1.w"].reshape(-1), out.reshape(-1))
    got = out[: B * P[1]].reshape(B, P[1], C)
    np.testing.assert_allclose(got[:, :T[1]], y.detach().numpy().transpose(0, 2, 1), rtol=1e-4, atol=1e-5)
    assert np.all(got[:, T[1]:] == 0)
    # input gradient: one GEMM per residue class of the stride, reading Q guard rows back
    gyl = np.zeros((GUARD + B * P[1] + GUARD, C), np.float32)
    gyl[GUARD: GUARD + B * P[1]].reshape(B, P[1], C)[:, :T[1]] = gy.transpose(0, 2, 1)
    gx = np.full((B * P[0], C), np.nan, np.float32)
    zeros_aux = np.zeros_like(gx)
    for rho in range(s):
        if rho > k - 1:
            continue
        Q = (k - 1 - rho) // s
        wd = pk[f"c1.wd{rho}"]
        d = dict(M=B * P[1], N=C, K=(Q + 1) * C, lda=C, ldb=(Q + 1) * C, ldc=s * C)
        emulate(d, gyl.reshape(-1), wd.reshape(-1), gx.reshape(-1), a_off=(GUARD - Q) * C, c_off=rho * C)
    gxr = gx.reshape(B, P[0], C)
    np.testing.assert_allclose(gxr[:, :T[0]], xt.grad.numpy().transpose(0, 2, 1), rtol=1e-4, atol=1e-5)
    assert np.all(gxr[:, T[0]:] == 0)


def test_pos_conv_as_windowed_gemm():
    a = A.tiny()
    sd = A.rule_weights(a)
    pk = pack_weights(a, sd)
    H, G, K = a.hidden_size, a.num_conv_pos_embedding_groups, a.num_conv_pos_embeddings
    Hg = H // G
    B, T, P = 2, 24, 25
    rng = np.random.default_rng(1)
    h = rng.normal(size=(B, T, H)).astype(np.float32)
    Wp = torch.from_numpy(A.pos_conv_weight(sd))
    ht = torch.from_numpy(h).requires_grad_(True)
    y = F.conv1d(ht.transpose(1, 2), Wp, torch.from_numpy(sd["wav2vec2.encoder.pos_conv_embed.conv.bias"]),
                 padding=K // 2, groups=G)[:, :, :-1].transpose(1, 2)
    gy = rng.normal(size=(B, T, H)).astype(np.float32)
    y.backward(torch.from_numpy(gy))
    hp = np.zeros((B, P, H), np.float32); hp[:, :T] = h
    out = np.zeros((B, P, H), np.float32)
    d = dict(M=T, N=Hg, K=K * Hg, lda=H, ldb=K * Hg, ldc=H, a_kseg=Hg, a_kseg_stride=H, a_window=1, a_pad=K // 2,
             a_rows_valid=T, batch=B * G, batch2=G, a_s1=P * H, a_s2=Hg, b_s1=0, b_s2=Hg * K * Hg, c_s1=P * H, c_s2=Hg,
             bias_s2=Hg)
    emulate(d, hp.reshape(-1), pk["pc.w"].reshape(-1), out.reshape(-1), bias=pk["pc.b"])
    np.testing.assert_allclose(out[:, :T], y.detach().numpy(), rtol=1e-4, atol=1e-5)
    gp = np.zeros((B, P, H), np.float32); gp[:, :T] = gy
    gh = np.zeros((B, P, H), np.float32)
    d2 = dict(d)
    d2.update(a_pad=K - 1 - K // 2, bias_s2=0)
    emulate(d2, gp.reshape(-1), pk["pc.wd"].reshape(-1), gh.reshape(-1))
    np.testing.assert_allclose(gh[:, :T], ht.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_padded_layout_base():
    T, P = padded_rows(A.BASE, 160000)
    assert T == [31999, 15999, 7999, 3999, 1999, 999, 499]
    assert P == [32000, 16000, 8000, 4000, 2000, 1000, 500]
    T, P = padded_rows(A.BASE, 480000)
    assert T[-1] == 1499 and all(p >= t for p, t in zip(P, T))


def test_pos_conv_weight_key_aliases():
    """The weight-norm fold accepts the folded weight, the torch >= 2.1 parametrization keys and the legacy
    weight_g / weight_v pair (raw 960h checkpoint files) — same result."""
    a = A.tiny()
    sd = A.rule_weights(a)
    pc = "wav2vec2.encoder.pos_conv_embed.conv"
    w = A.pos_conv_weight(sd)
    legacy = {k: v for k, v in sd.items() if "parametrizations" not in k}
    legacy[f"{pc}.weight_g"] = sd[f"{pc}.parametrizations.weight.original0"]
    legacy[f"{pc}.weight_v"] = sd[f"{pc}.parametrizations.weight.original1"]
    np.testing.assert_array_equal(A.pos_conv_weight(legacy), w)
    folded = {f"{pc}.weight": w}
    np.testing.assert_array_equal(A.pos_conv_weight(folded), w)
    import pytest
    with pytest.raises(KeyError):
        A.pos_conv_weight({})


# ---- row (g): local-checkpoint loading (src/training_utils/build.py:225-231 loads Wav2Vec2ForCTC by name; here a LOCAL directory) ----
def test_arch_from_hf_config_both_topologies():
    """The two published topologies, as their config.json files describe them: wav2vec2-base-960h is Wav2Vec2Config's default
    (group-norm extractor without conv bias, post-LN encoder); wav2vec2-large-960h-lv60-self is the layer-norm extractor with conv
    bias and the pre-LN ("stable") encoder.  A wrong feat_extract_norm / do_stable_layer_norm mapping would pick the wrong kernels."""
    pytest.importorskip("transformers")
    from transformers import Wav2Vec2Config
    from paa_amd.model import arch_from_hf_config, arch_struct
    assert arch_from_hf_config(Wav2Vec2Config()) == A.BASE
    large = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, conv_bias=True,
                           feat_extract_norm="layer", do_stable_layer_norm=True)
    assert arch_from_hf_config(large) == A.LARGE_LV60
    s = arch_struct(arch_from_hf_config(large))
    assert (s.feat_norm_layer, s.stable_ln, s.conv_bias, s.hidden, s.layers, s.heads, s.ffn) == (1, 1, 1, 1024, 24, 16, 4096)
    s = arch_struct(arch_from_hf_config(Wav2Vec2Config()))
    assert (s.feat_norm_layer, s.stable_ln, s.conv_bias, s.hidden, s.layers, s.heads, s.ffn, s.vocab, s.blank) == (0, 0, 0, 768, 12, 12, 3072, 32, 0)
    assert list(s.conv_kernel)[:7] == [10, 3, 3, 3, 3, 2, 2] and list(s.conv_stride)[:7] == [5, 2, 2, 2, 2, 2, 2]


@pytest.mark.parametrize("norm,stable", [("group", False), ("layer", True)])
def test_pack_weights_of_hf_state_dict(norm, stable):
    """pack_weights over an HF module's own state_dict() — parametrised weight-norm keys (torch >= 2.1) and the weight_g / weight_v
    names of the published checkpoint files — equals pack_weights over the rule weights the module was loaded from, and the folded
    positional-conv weight equals the one HF's module computes."""
    pytest.importorskip("transformers")
    from hf_util import hf_model
    from paa_amd.arch import pos_conv_weight
    from paa_amd.model import arch_from_hf_config
    a = A.tiny(norm, stable)
    sd = A.rule_weights(a)
    hf = hf_model(a, sd)
    assert arch_from_hf_config(hf.config) == a
    ref = pack_weights(a, sd)
    hsd = {k: v.detach().numpy() for k, v in hf.state_dict().items()}
    pc = "wav2vec2.encoder.pos_conv_embed.conv"
    assert f"{pc}.parametrizations.weight.original0" in hsd or f"{pc}.weight_g" in hsd
    legacy = {k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v"): v
              for k, v in hsd.items()}
    assert f"{pc}.weight_g" in legacy and f"{pc}.weight_v" in legacy
    for name, d in (("hf", hsd), ("legacy", legacy)):
        got = pack_weights(arch_from_hf_config(hf.config), d)
        assert got.keys() == ref.keys(), name
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{name}:{k}")
    np.testing.assert_allclose(pos_conv_weight(hsd), hf.wav2vec2.encoder.pos_conv_embed.conv.weight.detach().numpy(), rtol=2e-6, atol=1e-7)
    with pytest.raises(KeyError):
        pos_conv_weight({k: v for k, v in hsd.items() if "pos_conv_embed" not in k})


def test_shard_batches_and_objective():
    """build.shard_batches: contiguous shards whose sizes differ by at most one clip, together covering every global batch; a batch
    with fewer clips than ranks is dropped on every rank."""
    from paa_amd.training_utils import build
    batches = [(torch.arange(7 * 3).reshape(7, 3).float(), [f"t{i}" for i in range(7)]),
               (torch.arange(2 * 3).reshape(2, 3).float(), ["a", "b"])]
    for world in (1, 2, 3):
        shards = [build.shard_batches(batches, r, world) for r in range(world)]
        n_steps = {len(s) for s in shards}
        assert len(n_steps) == 1                                   # every rank runs the same number of steps
        for step in range(n_steps.pop()):
            xs = torch.cat([s[step][0] for s in shards])
            ts = sum((s[step][1] for s in shards), [])
            assert torch.equal(xs, batches[step][0]) and ts == batches[step][1]
            sizes = [len(s[step][1]) for s in shards]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    assert len(build.shard_batches(batches, 0, 3)) == 1            # the 2-clip batch cannot feed 3 ranks in training
    ev = [build.shard_batches(batches, r, 3, keep_empty=True) for r in range(3)]
    assert [len(e) for e in ev] == [2, 2, 2] and sorted(len(e[1][1]) for e in ev) == [0, 1, 1]
    assert sum((e[1][1] for e in ev), []) == ["a", "b"]
