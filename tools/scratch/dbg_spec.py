import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
from paa_amd import _lib, runtime, synth
from paa_amd.training_utils import build, parser
lib = _lib.lib()
rows, L = 32, 160000
x = torch.from_numpy(synth.clean_audio(rows, L)).cuda() * 0.3
for norm in ("min_max_freqs", "max_phon"):
    args = parser.create_arg_parser().parse_args(["--norm_type", norm, "--device", "cuda"])
    pr = runtime.get_proj(args, x.device, rows, L)
    pr.set_spl_thresh(build.init_phon_threshold_tensor(args))
    prm = runtime.params_of(args)
    many = torch.empty(rows, L, device="cuda")
    _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(x), _lib.ptr(many), rows, None, 0, L, _lib.stream_ptr()))
    for r in (0, 5, 31):
        one = torch.empty(1, L, device="cuda")
        _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(x[r:r+1].contiguous()), _lib.ptr(one), 1, None, 0, L, _lib.stream_ptr()))
        d = (one[0] - many[r]).abs()
        nz = torch.nonzero(d > 0).flatten()
        print(norm, r, "max diff", float(d.max()), "scale", float(one.abs().max()), "n diff", nz.numel(), "first", nz[:8].tolist(), "last", nz[-4:].tolist())
        if nz.numel():
            blk = (nz // 256).unique()
            print("   blocks with diffs:", blk.numel(), blk[:20].tolist())
