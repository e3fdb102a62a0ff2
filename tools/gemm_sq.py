import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run
run("square 8192 bf16 out only", 8192, 8192, 8192, outf=False, outb=True, prec=0, iters=5)
run("square 4096", 4096, 4096, 4096, iters=10)
