"""Does HBM care about the access pattern of the GEMM's A operand?  Copies 128..2048-byte segments out of 2 KB rows (the conv
activations' layout as a K slab sees it) and the same number of bytes from a contiguous array, on the GPU box:
    python tools/stride_probe.py
Result (MI355X): 2.5-3.1 TB/s read (+ the same written) either way — strided 128-byte segments cost nothing extra."""
import torch, time
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rows = 2 * 1024 * 1024
x = torch.randn(rows, 1024, device="cuda").to(torch.bfloat16)       # row stride 2 KB, 4 GB
for w in (64, 128, 256, 512, 1024):
    y = torch.empty(rows, w, device="cuda", dtype=torch.bfloat16)
    ms = bench(lambda: y.copy_(x[:, :w]))
    rb = rows * w * 2
    print(f"strided read {w*2:5d} B per 2 KB row: {ms*1e3:8.1f} us  read {rb/ms/1e6:8.1f} GB/s  read+write {2*rb/ms/1e6:8.1f} GB/s")
# same bytes, contiguous
for w in (64, 256):
    src = torch.randn(rows, w, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(src)
    ms = bench(lambda: y.copy_(src))
    rb = rows * w * 2
    print(f"contiguous {rb/1e6:.0f} MB: {ms*1e3:8.1f} us  read {rb/ms/1e6:8.1f} GB/s")
