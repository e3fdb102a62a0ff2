"""Where an iteration of k_spec_run goes (diagnostic build with s_memtime stamps; run on the GPU box):

    PAA_EXTRA_HIPCC_FLAGS="-DPAA_EXPERIMENTS -DPAA_SPEC_STAMP" python tools/fft_stamps.py

Builds the diagnostic library next to the shipped one, runs min_max_freqs at (32, 160000) and prints, for workgroup (0, 0), per wave
and iteration: frame time, wait at barrier A, overlap-add, wait at barrier B (shader cycles)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PAA_EXTRA_HIPCC_FLAGS", "-DPAA_EXPERIMENTS -DPAA_SPEC_STAMP")
import numpy as np
import torch

from paa_amd import _lib, build_ext, runtime
from paa_amd.training_utils import build, parser

build_ext.build(verbose=False)
lib = _lib.lib()
rows, L = 32, 160000
x = (torch.randn(rows, L, device="cuda") * 0.05).contiguous()
y = torch.empty_like(x)
args = parser.create_arg_parser().parse_args(["--norm_type", "min_max_freqs", "--device", "cuda"])
pr = runtime.get_proj(args, x.device, rows, L)
pr.set_spl_thresh(build.init_phon_threshold_tensor(args))
prm = runtime.params_of(args)
for _ in range(20):
    _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(x), _lib.ptr(y), rows, None, 0, L, _lib.stream_ptr()))
torch.cuda.synchronize()
n = 64 + 12 * 8 * 5
buf = (ctypes.c_longlong * n)()
lib.paa_debug_spec_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
_lib.check(lib.paa_debug_spec_stamps(pr.h, buf, n))
s = np.array(buf[64:], dtype=np.int64).reshape(12, 8, 5)
t0 = s[:, 0, 0].min()
print("wave it | top(rel) frame waitA ola waitB | next-top gap")
for w in range(12):
    for it in range(7):
        a = s[w, it]
        nxt = s[w, it + 1, 0] - a[4] if it + 1 < 7 else 0
        print(f"{w:3d} {it:2d} | {a[0] - t0:7d} {a[1] - a[0]:6d} {a[2] - a[1]:6d} {a[3] - a[2]:5d} {a[4] - a[3]:6d} | {nxt:5d}")
it_len = s[:, 1:7, 0] - s[:, 0:6, 0]
print("iteration length per wave (cycles): mean %.0f min %d max %d" % (it_len.mean(), it_len.min(), it_len.max()))
print("frame %.0f  waitA %.0f  ola %.0f  waitB %.0f" % ((s[:, :7, 1] - s[:, :7, 0]).mean(), (s[:, :7, 2] - s[:, :7, 1]).mean(), (s[:, :7, 3] - s[:, :7, 2]).mean(), (s[:, :7, 4] - s[:, :7, 3]).mean()))
