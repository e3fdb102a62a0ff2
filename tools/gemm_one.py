"""Run a few launches of one GEMM shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run
if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "square"
    if which == "square":
        run("square 4096", 4096, 4096, 4096, iters=5)
    elif which == "ffn1":
        run("ffn1", 16000, 3072, 768, act=1, pre=True, outf=False, outb=True, iters=5)
    elif which == "conv1":
        run("conv1", 512000, 512, 1536, lda=1024, act=1, pre=True, outf=False, outb=True, rowmask=16000, iters=3)
