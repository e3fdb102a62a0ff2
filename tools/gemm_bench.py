"""Micro-benchmark of paa_gemm shapes taken from the base model at B=32 x 10 s (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from paa_amd import _lib

L = _lib.lib()


def run(name, M, N, K, lda=None, act=0, pre=False, outf=True, outb=False, prec=0, resid=False, iters=20, rowmask=0, alias=False):
    lda = lda or K
    rows = M * lda // K + 8 if lda < K else M
    A = torch.randint(-2000, 2000, (max(M * lda + K, rows * K),), dtype=torch.int16, device="cuda")
    B = torch.randint(-2000, 2000, (N * K,), dtype=torch.int16, device="cuda")
    Al, Bl = A.clone(), B.clone()
    Cf = torch.zeros(M * N, device="cuda")
    Cp = torch.zeros(M * N, device="cuda")
    Cb = torch.zeros(M * N, dtype=torch.int16, device="cuda")
    Cbl = torch.zeros(M * N, dtype=torch.int16, device="cuda")
    aux = torch.randn(M * N, device="cuda")
    d = _lib.PaaGemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.A_lo, d.B_lo = Al.data_ptr(), Bl.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, (0 if alias else lda), (0 if alias else K), N
    d.a_kcontig = d.b_kcontig = 1
    d.batch = d.batch2 = 1
    d.alpha = 1.0
    d.operand_bf16 = 1
    d.precision = prec
    d.act = act
    if act == 2:
        d.aux, d.ld_aux = aux.data_ptr(), N
    if pre:
        d.C_pre = Cp.data_ptr()
    if outf:
        d.C = Cf.data_ptr()
    if outb:
        d.Cb = Cb.data_ptr()
        if prec:
            d.Cb_lo = Cbl.data_ptr()
    if resid:
        d.residual, d.ld_res = aux.data_ptr(), N
    if rowmask:
        d.row_period, d.row_valid = rowmask, rowmask - 1
    st = _lib.stream_ptr()
    for _ in range(3):
        _lib.check(L.paa_gemm(C.byref(d), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.paa_gemm(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} M={M:7d} N={N:5d} K={K:5d} prec={prec} {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    import sys
    modes = [int(a) for a in sys.argv[1:]] or [1]
    if modes == [3]:      # memory-latency probe: every A / B row aliases row 0 (lda = ldb = 0): operands always hit in cache
        for (nm, M, N, K, it) in (("ffn2", 16000, 768, 3072, 20), ("ffn1", 16000, 3072, 768, 20), ("conv1", 512000, 512, 1536, 5)):
            run(nm + " bf16 out", M, N, K, outf=False, outb=True, iters=it)
            run(nm + " bf16 out, rows aliased", M, N, K, lda=8 * 0 + 0, outf=False, outb=True, iters=it, alias=True)
        sys.exit(0)
    if modes == [2]:      # epilogue share: same product, different outputs
        for (nm, M, N, K, lda, it) in (("ffn1", 16000, 3072, 768, None, 20), ("qkv", 16000, 2304, 768, None, 20),
                                       ("outproj", 16000, 768, 768, None, 20), ("ffn2", 16000, 768, 3072, None, 20),
                                       ("conv1", 512000, 512, 1536, 1024, 5)):
            run(nm + " bf16 out only", M, N, K, lda=lda, outf=False, outb=True, iters=it)
            run(nm + " f32 out only", M, N, K, lda=lda, iters=it)
            run(nm + " f32 + resid", M, N, K, lda=lda, resid=True, iters=it)
            run(nm + " gelu pre+bf16", M, N, K, lda=lda, act=1, pre=True, outf=False, outb=True, iters=it)
            run(nm + " gelugrad f32", M, N, K, lda=lda, act=2, iters=it)
        sys.exit(0)
    for mode in modes:
      for prec in (0, 1):
        run("w1_t (N=768,K=3072) resid", 16000, 768, 3072, resid=True, prec=prec)
        run("wqkv_t (N=768,K=2304) resid", 16000, 768, 2304, resid=True, prec=prec)
        run("conv2 gelu pre+bf16 mask", 256000, 512, 1536, lda=1024, act=1, pre=True, outf=False, outb=True, rowmask=8000, prec=prec, iters=5)
        run("ffn1 plain f32 out", 16000, 3072, 768, prec=prec)
        run("ffn1 gelu pre+bf16 (model)", 16000, 3072, 768, act=1, pre=True, outf=False, outb=True, prec=prec)
        run("ffn2 resid f32", 16000, 768, 3072, resid=True, prec=prec)
        run("qkv f32", 16000, 2304, 768, prec=prec)
        run("outproj", 16000, 768, 768, resid=True, prec=prec)
        run("conv1 plain f32", 512000, 512, 1536, lda=1024, prec=prec, iters=5)
        run("conv1 gelu pre+bf16 mask (model)", 512000, 512, 1536, lda=1024, act=1, pre=True, outf=False, outb=True, rowmask=16000, prec=prec, iters=5)
        run("conv1 dgrad-even gelugrad f32", 512000, 512, 1024, lda=512, act=2, prec=prec, iters=5)
        run("square 4096", 4096, 4096, 4096, prec=prec, iters=10)
        run("square 8192 bf16 out only", 8192, 8192, 8192, outf=False, outb=True, prec=prec, iters=5)
