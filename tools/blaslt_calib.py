"""Calibration only (not part of the product path): hipBLASLt (torch.matmul, bf16) on the model's GEMM shapes,
to see how far the hand-written kernel is from the vendor library on the same device."""
import torch

def run(name, M, N, K, iters=20):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:20s} M={M:7d} N={N:5d} K={K:5d} {ms*1e3:9.1f} us {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s (bf16 out)", flush=True)

if __name__ == "__main__":
    run("w1_t", 16000, 768, 3072)
    run("wqkv_t", 16000, 768, 2304)
    run("ffn1", 16000, 3072, 768)
    run("qkv", 16000, 2304, 768)
    run("outproj", 16000, 768, 768)
    run("conv1-like", 512000, 512, 1536, iters=5)
    run("conv-dgrad-like", 512000, 512, 1024, iters=5)
    run("square4096", 4096, 4096, 4096)
    run("square8192", 8192, 8192, 8192, iters=5)
