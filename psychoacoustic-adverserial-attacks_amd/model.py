"""Host side of the Wav2Vec2 forward/backward: weight packing and the ``paa_model`` handle.

The reference obtains the model with ``Wav2Vec2ForCTC.from_pretrained(<name>)``
(src/training_utils/build.py:229-230) and differentiates through it with autograd
(train.py:158).  Here the weights (a HuggingFace state dict from a LOCAL checkpoint, an in-memory HF
module, or ``arch.rule_weights``) are re-laid-out once for the MFMA GEMMs and handed to
``paa_model_create``; one call then yields loss, logits and the gradient wrt the waveform.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, runtime
from .arch import Wav2Vec2Arch, pos_conv_weight


def _np(v):
    return v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)


def pack_weights(a: Wav2Vec2Arch, sd: dict) -> dict:
    """HF state dict -> {packed name: float32 ndarray} in the layouts csrc/model.hip expects.

    GEMM B operands are stored [N][K] (K contiguous).  For a strided conv with weight W[o][c][j]:
      c{i}.w       [O][k*I]           K index j*I + c           (forward, channel-last im2col)
      c{i}.wd{r}   [I][(Q+1)*O]       K index q*O + o  <->  tap j = r + s*(Q - q)
                   (input gradient of the residue class r of the stride, Q = (k-1-r)//s)
    For the grouped positional conv (weight-norm folded, G groups of Hg channels, K taps):
      pc.w   [G][Hg][K*Hg]   K index j*Hg + c       pc.wd  [G][Hg][K*Hg]  K index j'*Hg + o, tap K-1-j'
    Linear layers keep W ([out][in]) for the forward and W^T for the input gradient.
    """
    sd = {k: _np(v) for k, v in sd.items()}
    out = {}
    fe = "wav2vec2.feature_extractor.conv_layers"
    for i, (k, s) in enumerate(zip(a.conv_kernel, a.conv_stride)):
        W = sd[f"{fe}.{i}.conv.weight"]                          # [O][I][k]
        O, I, _ = W.shape
        if i == 0:
            out["c0.w"] = W.reshape(O, k)
        else:
            out[f"c{i}.w"] = W.transpose(0, 2, 1).reshape(O, k * I)
            for r in range(s):
                if r > k - 1:
                    continue
                Q = (k - 1 - r) // s
                blocks = [W[:, :, r + s * (Q - q)].T for q in range(Q + 1)]     # each [I][O]
                out[f"c{i}.wd{r}"] = np.concatenate(blocks, axis=1)
        if a.conv_bias:
            out[f"c{i}.b"] = sd[f"{fe}.{i}.conv.bias"]
        if (a.feat_extract_norm == "group" and i == 0) or a.feat_extract_norm == "layer":
            out[f"c{i}.g"] = sd[f"{fe}.{i}.layer_norm.weight"]
            out[f"c{i}.beta"] = sd[f"{fe}.{i}.layer_norm.bias"]
    fp = "wav2vec2.feature_projection"
    out["fp.ln_g"], out["fp.ln_b"] = sd[f"{fp}.layer_norm.weight"], sd[f"{fp}.layer_norm.bias"]
    out["fp.w"], out["fp.b"] = sd[f"{fp}.projection.weight"], sd[f"{fp}.projection.bias"]
    out["fp.wt"] = out["fp.w"].T
    Wp = pos_conv_weight(sd)                                     # [H][Hg][K]
    H, Hg, K = Wp.shape
    G = H // Hg
    Wg = Wp.reshape(G, Hg, Hg, K)                                # [g][o][c][j]
    out["pc.w"] = Wg.transpose(0, 1, 3, 2).reshape(G, Hg, K * Hg)               # [g][o][j*Hg + c]
    out["pc.wd"] = Wg[:, :, :, ::-1].transpose(0, 2, 3, 1).reshape(G, Hg, K * Hg)   # [g][c][j'*Hg + o]
    out["pc.b"] = sd["wav2vec2.encoder.pos_conv_embed.conv.bias"]
    out["enc.ln_g"], out["enc.ln_b"] = sd["wav2vec2.encoder.layer_norm.weight"], sd["wav2vec2.encoder.layer_norm.bias"]
    for l in range(a.num_hidden_layers):
        p = f"wav2vec2.encoder.layers.{l}"
        wq, wk, wv = (sd[f"{p}.attention.{n}_proj.weight"] for n in "qkv")
        bq, bk, bv = (sd[f"{p}.attention.{n}_proj.bias"] for n in "qkv")
        o = f"L{l}"
        out[f"{o}.wqkv"] = np.concatenate([wq, wk, wv], 0)
        out[f"{o}.bqkv"] = np.concatenate([bq, bk, bv], 0)
        out[f"{o}.wqkv_t"] = out[f"{o}.wqkv"].T
        out[f"{o}.wo"], out[f"{o}.bo"] = sd[f"{p}.attention.out_proj.weight"], sd[f"{p}.attention.out_proj.bias"]
        out[f"{o}.wo_t"] = out[f"{o}.wo"].T
        out[f"{o}.ln1_g"], out[f"{o}.ln1_b"] = sd[f"{p}.layer_norm.weight"], sd[f"{p}.layer_norm.bias"]
        out[f"{o}.w1"], out[f"{o}.b1"] = sd[f"{p}.feed_forward.intermediate_dense.weight"], sd[f"{p}.feed_forward.intermediate_dense.bias"]
        out[f"{o}.w1_t"] = out[f"{o}.w1"].T
        out[f"{o}.w2"], out[f"{o}.b2"] = sd[f"{p}.feed_forward.output_dense.weight"], sd[f"{p}.feed_forward.output_dense.bias"]
        out[f"{o}.w2_t"] = out[f"{o}.w2"].T
        out[f"{o}.ln2_g"], out[f"{o}.ln2_b"] = sd[f"{p}.final_layer_norm.weight"], sd[f"{p}.final_layer_norm.bias"]
    out["lm.w"], out["lm.b"] = sd["lm_head.weight"], sd["lm_head.bias"]
    out["lm.wt"] = out["lm.w"].T
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


def is_gemm_weight(name: str) -> bool:
    """Packed tensors that are GEMM B operands (stored as bf16 planes on the device); the rest stay f32."""
    if name == "c0.w":
        return False
    leaf = name.split(".", 1)[1]
    return leaf in ("w", "wt", "wd", "wqkv", "wqkv_t", "wo", "wo_t", "w1", "w1_t", "w2", "w2_t") or leaf.startswith("wd")


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round-to-nearest-even (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def interleave_planes(hi: np.ndarray, lo: np.ndarray) -> np.ndarray:
    """[N][K] hi / lo planes -> [N][2K] with the planes interleaved per 32-element K group (paa_gemm_desc.B_il)."""
    n, k = hi.shape
    return np.ascontiguousarray(np.stack([hi.reshape(n, k // 32, 32), lo.reshape(n, k // 32, 32)], axis=2).reshape(n, 2 * k))


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def split_bf16(x: np.ndarray):
    """x ~= hi + lo with both bf16: the operand planes of the fp32-parity (3-pass) GEMM."""
    hi = bf16_bits(x)
    lo = bf16_bits(np.asarray(x, dtype=np.float32) - bf16_to_f32(hi))
    return hi, lo


def arch_struct(a: Wav2Vec2Arch) -> _lib.PaaArch:
    s = _lib.PaaArch()
    s.n_conv = len(a.conv_dim)
    for i in range(s.n_conv):
        s.conv_dim[i], s.conv_kernel[i], s.conv_stride[i] = a.conv_dim[i], a.conv_kernel[i], a.conv_stride[i]
    s.conv_bias = int(a.conv_bias)
    s.feat_norm_layer = int(a.feat_extract_norm == "layer")
    s.hidden, s.layers, s.heads, s.ffn = a.hidden_size, a.num_hidden_layers, a.num_attention_heads, a.intermediate_size
    s.pos_k, s.pos_groups = a.num_conv_pos_embeddings, a.num_conv_pos_embedding_groups
    s.stable_ln = int(a.do_stable_layer_norm)
    s.vocab, s.blank = a.vocab_size, a.pad_token_id
    s.ln_eps = a.layer_norm_eps
    return s


def arch_from_hf_config(cfg) -> Wav2Vec2Arch:
    return Wav2Vec2Arch(conv_dim=tuple(cfg.conv_dim), conv_kernel=tuple(cfg.conv_kernel), conv_stride=tuple(cfg.conv_stride),
                        conv_bias=bool(cfg.conv_bias), feat_extract_norm=cfg.feat_extract_norm, hidden_size=cfg.hidden_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                        num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups,
                        do_stable_layer_norm=bool(cfg.do_stable_layer_norm), layer_norm_eps=float(cfg.layer_norm_eps),
                        vocab_size=cfg.vocab_size, pad_token_id=cfg.pad_token_id)


PRECISIONS = {"bf16": 0, "fp32": 1}


class PaaModel:
    """Wav2Vec2ForCTC forward + CTC + input gradient on one GPU."""

    def __init__(self, arch: Wav2Vec2Arch, state_dict: dict, max_batch: int, length: int, dtype: str = "bf16",
                 device="cuda"):
        if dtype not in PRECISIONS:
            raise ValueError(f"dtype must be one of {list(PRECISIONS)}")
        self.arch, self.max_batch, self.length, self.dtype = arch, int(max_batch), int(length), dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PaaModel needs a GPU; there is no CPU fallback")
        packed = pack_weights(arch, state_dict)
        # GEMM weights become bf16 planes (hi, and lo for the fp32-parity mode); everything else stays f32
        planes = {}
        for k, v in packed.items():
            if is_gemm_weight(k):
                hi, lo = split_bf16(v)
                planes[k] = hi
                if dtype == "fp32":
                    planes[k + ".lo"] = lo
                    if hi.ndim == 2 and hi.shape[1] % 32 == 0:
                        planes[k + ".il"] = interleave_planes(hi, lo)
            else:
                planes[k] = v
        # one device byte buffer for all packed weights, 256-byte aligned slices
        offs, total = {}, 0
        for k, v in planes.items():
            offs[k] = total
            total += (v.nbytes + 255) // 256 * 256
        host = np.zeros(total, dtype=np.uint8)
        for k, v in planes.items():
            host[offs[k]:offs[k] + v.nbytes] = v.reshape(-1).view(np.uint8)
        self._weights = torch.from_numpy(host).to(self.device)
        names = [k.encode() for k in planes]
        arr = (_lib.PaaTensor * len(planes))()
        base = self._weights.data_ptr()
        for i, (k, v) in enumerate(planes.items()):
            arr[i].name = names[i]
            arr[i].d_ptr = base + offs[k]
            arr[i].numel = v.size
        self._names = names
        packed = planes
        h = C.c_void_p()
        st = arch_struct(arch)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().paa_model_create(C.byref(h), C.byref(st), arr, len(packed), self.max_batch, self.length,
                                                   PRECISIONS[dtype]))
        self.h = h
        self.frames = _lib.lib().paa_model_frames(h)
        self.workspace_bytes = _lib.lib().paa_model_workspace_bytes(h)

    @classmethod
    def from_hf(cls, hf_model, max_batch, length, dtype="bf16", device="cuda"):
        return cls(arch_from_hf_config(hf_model.config), hf_model.state_dict(), max_batch, length, dtype, device)

    def _checked(self, clean, p):
        """Raw pointers cross the C ABI: refuse anything that is not a contiguous float32 tensor on this model's GPU
        with the configured length (a float64 / CPU / strided batch would be read as garbage device memory)."""
        clean = runtime.as_f32_cuda(clean, "clean_audio")
        if clean.dim() != 2:
            raise ValueError(f"clean_audio must be (B, L), got {tuple(clean.shape)}")
        if clean.device != self.device and not (self.device.index is None and clean.device.type == "cuda"):
            raise RuntimeError(f"clean_audio lives on {clean.device}, the model on {self.device}")
        if clean.shape[1] != self.length:
            raise ValueError(f"Loaded perturbation length {clean.shape[1]} != expected {self.length}")
        if clean.shape[0] < 1 or clean.shape[0] > self.max_batch:
            raise ValueError(f"batch {clean.shape[0]} exceeds the model's max_batch {self.max_batch}")
        if p is not None:
            p = runtime.as_f32_cuda(p, "p")
            if p.numel() != self.length:
                raise ValueError(f"Loaded perturbation length {p.numel()} != expected {self.length}")
            if p.device != clean.device:
                raise RuntimeError(f"p lives on {p.device}, clean_audio on {clean.device}")
        return clean, p

    def fwd_bwd(self, clean, p, labels, direction=+1, want_grad=True, want_logits=True, out=None):
        """clean (B, L) f32 cuda; p (1, L) or None; labels (B, S) integer tensor, negatives = padding.
        Returns dict(loss=0-d tensor, logits=(B, T_e, V) | None, grad=(1, L) | None, stats=(8,))."""
        clean, p = self._checked(clean, p)
        B, L = clean.shape
        dev = self.device
        lab = None if labels is None else labels.to(device=dev, dtype=torch.int32).contiguous()
        S = 0 if lab is None else lab.shape[1]
        out = out or {}
        grad = out.get("grad") if want_grad else None
        if want_grad and grad is None:
            grad = torch.empty(1, L, dtype=torch.float32, device=dev)
        logits = out.get("logits") if want_logits else None
        if want_logits and logits is None:
            logits = torch.empty(B, self.frames, self.arch.vocab_size, dtype=torch.float32, device=dev)
        stats = out.get("stats")
        if stats is None:
            stats = torch.zeros(8, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().paa_model_fwd_bwd(self.h, _lib.ptr(clean), _lib.ptr(p), _lib.ptr(lab), B, S, int(direction),
                                                    _lib.ptr(grad), _lib.ptr(logits), _lib.ptr(stats), _lib.stream_ptr()))
        return dict(loss=stats[0], logits=logits, grad=grad, stats=stats, labels=lab)

    def forward(self, clean, p, labels, clamp=False, want_logits=True):
        """Forward + CTC loss only.  ``clamp=False`` composes ``clean + p`` as the reference's evaluation does
        (evaluation.py:16); ``p=None`` evaluates the clean batch.  Returns dict(loss, logits)."""
        clean, p = self._checked(clean, p)
        B, L = clean.shape
        dev = self.device
        lab = None if labels is None else labels.to(device=dev, dtype=torch.int32).contiguous()
        logits = torch.empty(B, self.frames, self.arch.vocab_size, dtype=torch.float32, device=dev) if want_logits else None
        stats = torch.zeros(8, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().paa_model_forward(self.h, _lib.ptr(clean), _lib.ptr(p), int(bool(clamp)), _lib.ptr(lab), B,
                                                    0 if lab is None else lab.shape[1], _lib.ptr(logits), _lib.ptr(stats),
                                                    _lib.stream_ptr()))
        return dict(loss=stats[0], logits=logits)

    def debug_read(self, name: str, B: int) -> np.ndarray:
        """Copy a named internal activation to the host (tests / diagnostics only)."""
        n = _lib.lib().paa_model_debug_read(self.h, name.encode(), None, 0, B)
        if n <= 0:
            raise KeyError(name)
        buf = np.empty(n, dtype=np.float32)
        got = _lib.lib().paa_model_debug_read(self.h, name.encode(), buf.ctypes.data_as(C.c_void_p), n, B)
        if got != n:
            raise RuntimeError(f"debug_read({name}) failed: {got}")
        return buf

    def layout(self, i: int) -> int:
        return _lib.lib().paa_model_layout(self.h, i)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().paa_model_destroy(self.h)
        except Exception:
            pass
