"""A/B of the paa_gemm kernel configurations on the model's large shapes (run on the GPU box): every ring configuration
against the register-staged kernels of round 1, interleaved rounds in ONE process on the same random operands, results
compared bit for bit (same K order => identical f32 accumulators)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paa_amd import _lib

L = _lib.lib()
# (name, M, N, K, lda, epilogue) — B=32 x 10 s, wav2vec2-base
# "pitch" probes: the same products with the A row pitch moved off a multiple of 2 KB (L2 channel spread)
PITCH = [("pitch conv1 lda1024", 512000, 512, 1536, 1024, "gelu"), ("pitch conv1 lda1088", 512000, 512, 1536, 1088, "gelu"),
         ("pitch conv1 lda1056", 512000, 512, 1536, 1056, "gelu"),
         ("pitch ffn2 lda3072", 16000, 768, 3072, 3072, "res"), ("pitch ffn2 lda3136", 16000, 768, 3072, 3136, "res"),
         ("pitch dqkv lda2304", 16000, 768, 2304, 2304, "res"), ("pitch dqkv lda2368", 16000, 768, 2304, 2368, "res"),
         ("pitch ffn1 lda768", 16000, 3072, 768, 768, "gelu"), ("pitch ffn1 lda832", 16000, 3072, 768, 832, "gelu")]
PROBES = [("probe conv1 bf16-out", 512000, 512, 1536, 1024, "bf"), ("probe conv1 bf16-out ALIASED rows", 512000, 512, 1536, 0, "bf"),
          ("probe ffn1 bf16-out", 16000, 3072, 768, None, "bf"), ("probe ffn1 bf16-out ALIASED rows", 16000, 3072, 768, 0, "bf"),
          ("probe sq8192 bf16-out", 8192, 8192, 8192, None, "bf")]
# " kg" in the name: gemm.h's k_group order of the K slabs (channel-slab-major, tap-minor), k_group = 512
SHAPES = [("conv1 fwd gelu", 512000, 512, 1536, 1024, "gelu"), ("conv1 fwd gelu kg", 512000, 512, 1536, 1024, "gelu"),
          ("conv1 dgrad even", 512000, 512, 1024, 512, "gg"), ("conv1 dgrad even kg", 512000, 512, 1024, 512, "gg"),
          ("conv2 fwd gelu", 256000, 512, 1536, 1024, "gelu"), ("conv4 fwd gelu", 64000, 512, 1536, 1024, "gelu"),
          ("ffn1 gelu", 16000, 3072, 768, None, "gelu"), ("ffn2 resid", 16000, 768, 3072, None, "res"),
          ("qkv", 16000, 2304, 768, None, "bf"), ("dffn (w2_t) gg", 16000, 3072, 768, None, "gg"),
          ("dqkv (wqkv_t) resid", 16000, 768, 2304, None, "res"), ("outproj resid", 16000, 768, 768, None, "res")]


def build(M, N, K, lda, ep, prec, kg=0):
    alias = lda == 0              # every A / B row aliases row 0: operands always hit in cache (memory-system probe)
    lda = K if (lda is None or alias) else lda
    def rnd16(n, scale=1.0):          # random bf16 bit patterns of N(0, scale^2) values (full-range mantissas and signs)
        return (torch.randn(n, device="cuda") * scale).to(torch.bfloat16).view(torch.int16)
    A, B = rnd16(M * lda + K + 64), rnd16(N * K)
    Al, Bl = rnd16(M * lda + K + 64, 2.0 ** -9), rnd16(N * K, 2.0 ** -9)
    aux = torch.randn(M * N, device="cuda")
    aux16 = rnd16(M * N)
    d = _lib.PaaGemmDesc()
    d.A, d.B, d.A_lo, d.B_lo = A.data_ptr(), B.data_ptr(), Al.data_ptr(), Bl.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, (0 if alias else lda), (0 if alias else K), N
    d.a_kcontig = d.b_kcontig = 1
    d.batch = d.batch2 = 1
    d.alpha = 0.03
    d.operand_bf16 = 1
    d.precision = prec
    d.k_group = kg
    keep = [A, B, Al, Bl, aux, aux16]
    outs = {}

    def out(name, dtype):
        t = torch.zeros(M * N, dtype=dtype, device="cuda")
        outs[name] = t
        return t.data_ptr()
    if ep == "gelu":            # conv / ffn1 forward: GELU, pre-activation kept, bf16 planes out
        d.act = 1
        d.C_pre = out("pre", torch.int16 if not prec else torch.float32)
        d.aux_bf16 = d.aux_gate = 0 if prec else 1
        d.Cb = out("cb", torch.int16)
        if prec:
            d.Cb_lo = out("cbl", torch.int16)
    elif ep == "gg":            # dgrad through a GELU: multiply by the kept array, bf16 planes out
        d.act = 2
        d.aux = (aux if prec else aux16).data_ptr()
        d.ld_aux = N
        d.aux_bf16 = d.aux_gate = 0 if prec else 1
        d.Cb = out("cb", torch.int16)
        if prec:
            d.Cb_lo = out("cbl", torch.int16)
    elif ep == "res":
        d.residual, d.ld_res = aux.data_ptr(), N
        d.C = out("c", torch.float32)
    else:
        d.Cb = out("cb", torch.int16)
        if prec:
            d.Cb_lo = out("cbl", torch.int16)
    return d, outs, keep


def timeit(d, iters):
    st = _lib.stream_ptr()
    _lib.check(L.paa_gemm(C.byref(d), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.paa_gemm(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    cfgs = {0: [2, 21, 19, 23], 1: [17, 20, 18, 22]} if os.environ.get('PAA_R2_PROBE') else {0: [1, 2, 8, 19], 1: [1, 7, 17, 18]} if os.environ.get('PAA_SQ_PROBE') else {0: [1, 2, 14, 15], 1: [1, 7, 16]} if os.environ.get('PAA_W4_PROBE') else {0: [1, 2, 8], 1: [1, 4, 17]} if os.environ.get('PAA_KG_PROBE') else {0: [1, 8, 10], 1: [1, 7, 9]} if os.environ.get('PAA_MF16_PROBE') else {0: [1, 8], 1: [1, 7, 13]} if os.environ.get('PAA_DEEP_PROBE') else {0: [1, 2, 3, 5, 8], 1: [1, 4, 6, 7]}
    argv = sys.argv[1:]
    pick = None
    if "--one" in argv:               # e.g. --one conv1  (profiling runs: one shape family, few launches)
        pick = argv[argv.index("--one") + 1]
        argv = [a for a in argv if a not in ("--one", pick)]
    exact = "--exact" in argv         # --one matches the whole shape name
    argv = [a for a in argv if a != "--exact"]
    only = [int(a) for a in argv]
    for prec in (0, 1):
        if only and prec not in only:
            continue
        for (nm, M, N, K, lda, ep) in (PROBES if pick == "probe" else PITCH if pick == "pitch" else SHAPES):
            if pick and not (nm == pick if exact else nm.startswith(pick)):
                continue
            d, outs, keep = build(M, N, K, lda, ep, prec, 512 if nm.endswith(" kg") else 0)
            iters = 3 if M > 100000 else 10
            ref = None
            best = {}
            for rnd in range(3):
                for cfg in cfgs[prec]:
                    L.paa_gemm_config(cfg)
                    ms = timeit(d, iters)
                    best[cfg] = min(best.get(cfg, 1e9), ms)
                    if rnd == 0:
                        snap = {k: v.clone() for k, v in outs.items()}
                        if ref is None:
                            ref = snap
                        else:
                            for k in snap:
                                if not torch.equal(snap[k], ref[k]):
                                    bad = (snap[k] != ref[k]).float().mean().item()
                                    print(f"   MISMATCH cfg {cfg} output {k}: {bad:.3e} of elements differ", flush=True)
            fl = 2.0 * M * N * K
            print(f"prec={prec} {nm:22s} M={M:7d} N={N:5d} K={K:5d} " +
                  "  ".join(f"cfg{c}: {best[c] * 1e3:8.1f} us {fl / best[c] / 1e9:7.1f} TF" for c in cfgs[prec]), flush=True)
            del d, outs, keep
            torch.cuda.empty_cache()
    L.paa_gemm_config(0)


if __name__ == "__main__":
    main()
