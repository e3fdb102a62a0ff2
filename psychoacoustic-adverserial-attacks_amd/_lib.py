"""ctypes binding of libpaa_hip.so (include/paa_hip.h).  There is NO fallback: if the library is
missing or a call fails, the host side raises — the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PAA_EXTRA_HIPCC_FLAGS (tools/ only: diagnostic builds, see build_ext.py) selects the diagnostic library built next to the shipped one
LIB_PATH = os.path.join(HERE, "libpaa_hip_exp.so" if os.environ.get("PAA_EXTRA_HIPCC_FLAGS", "").strip() else "libpaa_hip.so")

ABI_VERSIONS = (300, 301)      # include/paa_hip.h paa_version: 301 = the same ABI built with -DPAA_EXPERIMENTS

PAA_OK, PAA_ERR_BAD_NORM, PAA_ERR_NEED_CLEAN, PAA_ERR_SIZE, PAA_ERR_HIP, PAA_ERR_ARG, PAA_ERR_MISSING = range(7)

NORM_IDS = {"l2": 0, "linf": 1, "snr": 2, "tv": 3, "fletcher_munson": 4, "min_max_freqs": 5, "max_phon": 6}


class PaaParams(C.Structure):
    _fields_ = [("norm_type", C.c_int32), ("l2_size", C.c_float), ("linf_size", C.c_float), ("snr_db", C.c_float),
                ("tv_epsilon", C.c_float), ("fm_epsilon", C.c_float), ("min_freq_attack", C.c_float),
                ("max_freq_attack", C.c_float), ("phon_reference_db", C.c_float), ("lr", C.c_float),
                ("direction", C.c_int32)]


class PaaArch(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("conv_dim", C.c_int32 * 8), ("conv_kernel", C.c_int32 * 8),
                ("conv_stride", C.c_int32 * 8), ("conv_bias", C.c_int32), ("feat_norm_layer", C.c_int32),
                ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32),
                ("pos_k", C.c_int32), ("pos_groups", C.c_int32), ("stable_ln", C.c_int32), ("vocab", C.c_int32),
                ("blank", C.c_int32), ("ln_eps", C.c_float)]


class PaaTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("d_ptr", C.c_void_p), ("numel", C.c_int64)]


class PaaGemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int64), ("a_kcontig", C.c_int32), ("a_kseg", C.c_int32), ("a_kseg_stride", C.c_int64),
                ("a_window", C.c_int32), ("a_pad", C.c_int32), ("a_rows_valid", C.c_int32),
                ("ldb", C.c_int64), ("b_kcontig", C.c_int32), ("ldc", C.c_int64),
                ("batch", C.c_int32), ("batch2", C.c_int32),
                ("a_s1", C.c_int64), ("a_s2", C.c_int64), ("b_s1", C.c_int64), ("b_s2", C.c_int64),
                ("c_s1", C.c_int64), ("c_s2", C.c_int64),
                ("alpha", C.c_float), ("bias", C.c_void_p), ("bias_s2", C.c_int64), ("act", C.c_int32),
                ("C_pre", C.c_void_p), ("aux", C.c_void_p), ("ld_aux", C.c_int64), ("aux_s1", C.c_int64),
                ("aux_s2", C.c_int64), ("residual", C.c_void_p), ("ld_res", C.c_int64), ("res_s1", C.c_int64),
                ("res_s2", C.c_int64), ("row_period", C.c_int32), ("row_valid", C.c_int32),
                ("accumulate", C.c_int32), ("precision", C.c_int32),
                ("operand_bf16", C.c_int32), ("A_lo", C.c_void_p), ("B_lo", C.c_void_p), ("Cb", C.c_void_p),
                ("Cb_lo", C.c_void_p), ("aux_bf16", C.c_int32), ("aux_gate", C.c_int32), ("k_group", C.c_int32), ("B_il", C.c_void_p),
                ("A_il", C.c_void_p), ("Cb_il", C.c_void_p),
                ("res_ln_stats", C.c_void_p), ("res_ln_g", C.c_void_p), ("res_ln_b", C.c_void_p)]


_SIGS = {
    "paa_last_error": (C.c_char_p, []),
    "paa_version": (C.c_int, []),
    "paa_abi_sizes": (None, [C.POINTER(C.c_int32)]),
    "paa_proj_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int]),
    "paa_proj_destroy": (None, [C.c_void_p]),
    "paa_proj_set_spl_thresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "paa_project": (C.c_int, [C.c_void_p, C.POINTER(PaaParams), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                              C.c_void_p]),
    "paa_project_to": (C.c_int, [C.c_void_p, C.POINTER(PaaParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p]),
    "paa_spectrum_project": (C.c_int, [C.c_void_p, C.POINTER(PaaParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "paa_fm_weighted_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "paa_project_ext": (C.c_int, [C.c_void_p, C.POINTER(PaaParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                  C.c_int, C.c_void_p]),
    "paa_batch_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "paa_stft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "paa_istft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "paa_sign_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "paa_clamp": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "paa_compose_clamp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "paa_model_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(PaaArch), C.POINTER(PaaTensor), C.c_int, C.c_int,
                                   C.c_int, C.c_int]),
    "paa_model_destroy": (None, [C.c_void_p]),
    "paa_model_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "paa_model_frames": (C.c_int, [C.c_void_p]),
    "paa_model_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "paa_model_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "paa_argmax_ids": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "paa_model_debug_read": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int]),
    "paa_model_layout": (C.c_int, [C.c_void_p, C.c_int]),
    "paa_gemm": (C.c_int, [C.POINTER(PaaGemmDesc), C.c_void_p]),
    "paa_gemm_config": (None, [C.c_int]),
    "paa_test_option": (C.c_int, [C.c_int, C.c_int]),
    "paa_prof_enable": (C.c_int, [C.c_int]),
    "paa_prof_read": (C.c_int, [C.c_void_p]),
    "paa_prof_pause": (C.c_int, [C.c_int]),
    "paa_attn_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "paa_attn_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 6 + [C.c_void_p]),
    "paa_attn_fwd_split": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    "paa_attn_bwd_split": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 6 + [C.c_void_p]),
    "paa_layernorm_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "paa_layernorm_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p]),
    "paa_softmax_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "paa_softmax_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "paa_ctc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "paa_ctc_work_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
}

_lib = None


class PaaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libpaa_hip status {status}: {msg}")
        self.status = status
        self.msg = msg


def _check_current():
    """A library built from other sources than the ones next to it is refused (content hash, build_ext.source_key): a stale
    kernel would pass the ABI-size check and silently compute with old code."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("paa_build_ext", os.path.join(HERE, "build_ext.py"))
    be = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(be)
    if not be.is_current():
        raise RuntimeError(f"{LIB_PATH} was not built from the sources in {be.CSRC} (content hash mismatch): run "
                           "`python __graft_entry__.py build`")


def lib():
    """Load libpaa_hip.so (built in-tree by paa_amd.build_ext / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                               "There is no CPU fallback for the product path.")
        _check_current()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        ver = L.paa_version()
        if ver not in ABI_VERSIONS:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {ver}, this binding needs one of {ABI_VERSIONS}: rebuild "
                               "(`python __graft_entry__.py build`)")
        sizes = (C.c_int32 * 4)()
        L.paa_abi_sizes(sizes)
        mine = [C.sizeof(PaaParams), C.sizeof(PaaArch), C.sizeof(PaaTensor), C.sizeof(PaaGemmDesc)]
        if list(sizes) != mine:
            raise RuntimeError(f"ABI mismatch between _lib.py and libpaa_hip.so: {list(sizes)} vs {mine}")
        _lib = L
    return _lib


def exported_symbols():
    return list(_SIGS)


def check(status: int):
    """Map a paa_status to the exception type the reference raises for the same condition
    (SURVEY §8b 'Errors')."""
    if status == PAA_OK:
        return
    msg = lib().paa_last_error().decode(errors="replace")
    if status in (PAA_ERR_BAD_NORM, PAA_ERR_NEED_CLEAN, PAA_ERR_SIZE):
        raise ValueError(msg)                       # train.py:91,95,98; build.py:315
    raise PaaError(status, msg)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
