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
