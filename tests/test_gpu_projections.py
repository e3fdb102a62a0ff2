"""-m gpu: the HIP projection path against (a) the goldens produced by the reference itself and
(b) the oracle at BASELINE sizes, plus size-independent properties (idempotence, STFT round trip)."""
import types

import numpy as np
import pytest
import torch

from gpu_util import rel_err
from oracle import projections as OP
from oracle.gen_cases import AMPS, LENGTHS, NORM_CASES, case_name, cli_to_args
from paa_amd import synth
from paa_amd.core import fourier_transforms, projections
from paa_amd.training_utils import build, train

pytestmark = pytest.mark.gpu
TOL = 2e-5      # |got - ref| <= TOL * max|ref|  (fp32 FFT / reduction-order differences)


def _args(norm, extra):
    a = cli_to_args(norm, extra)
    a.device = "cuda"
    return a


@pytest.mark.parametrize("norm,extra", NORM_CASES)
def test_constraint_vs_reference_goldens(gold, norm, extra):
    g = gold("projections.npz")
    args = _args(norm, extra)
    spl = build.init_phon_threshold_tensor(args)
    worst = 0.0
    for L in LENGTHS + [16000]:
        for B in ([1, 3] if norm in ("snr", "tv") else [1]):
            for amp in AMPS:
                clean = torch.from_numpy(synth.clean_audio(B, L)).cuda()
                p = torch.from_numpy(synth.perturbation(L) * np.float32(amp)).cuda()
                q = train.perturbation_constraint(p, clean, args, None, spl).cpu().numpy()
                name = case_name(norm, extra, L, B, amp)
                ref, got = (g[name + "|samples"], q[0, ::7]) if L == 16000 else (g[name], q)
                e = rel_err(got, ref) if np.abs(ref).max() > 0 else float(np.abs(got).max())
                worst = max(worst, e)
                assert e <= TOL, (name, e)
    print(f"{norm} {extra}: worst rel err {worst:.2e}")


def test_stft_istft_vs_goldens(gold):
    g = gold("projections.npz")
    args = _args("max_phon", [])
    for L in (4096, 5000):
        p = torch.from_numpy(np.concatenate([synth.perturbation(L) * np.float32(1e-2), synth.clean_audio(1, L)], 0)).cuda()
        S = fourier_transforms.compute_stft(p, args)
        ref = torch.view_as_complex(torch.from_numpy(g[f"stft|L{L}"]))
        assert tuple(S.shape) == tuple(ref.shape)
        e = float((S.cpu() - ref).abs().max() / ref.abs().max())
        y = fourier_transforms.compute_istft(ref.cuda(), args).cpu().numpy()
        e2 = float(np.abs(y - g[f"istft|L{L}"]).max())
        print(f"stft L={L}: rel {e:.2e}; istft abs {e2:.2e}")
        assert e < 5e-6 and e2 < 5e-7


@pytest.mark.parametrize("norm,extra", [("snr", ["--snr_db", "40"]), ("tv", []), ("l2", []), ("linf", []),
                                        ("fletcher_munson", ["--fm_epsilon", "2.0"]), ("max_phon", []), ("min_max_freqs", [])])
def test_constraint_full_size_vs_oracle(norm, extra):
    """BASELINE size: (1, 160000) perturbation, batch of 10 s clips."""
    args = _args(norm, extra)
    L, B = 160000, 4
    clean = torch.from_numpy(synth.clean_audio(B, L))
    for amp in (1e-3, 0.2):
        p = torch.from_numpy(synth.perturbation(L) * np.float32(amp))
        ref = OP.perturbation_constraint(p, clean, args, OP.spl_thresh_tensor(args)).numpy()
        spl = build.init_phon_threshold_tensor(args)
        got = train.perturbation_constraint(p.cuda(), clean.cuda(), args, None, spl)
        e = rel_err(got.cpu().numpy(), ref)
        print(f"{norm} amp={amp}: rel err {e:.2e}")
        assert e <= TOL
        # idempotence of the feasible-set projections (scale-type norms and linf): projecting twice changes nothing more
        if norm in ("l2", "linf", "tv"):
            again = train.perturbation_constraint(got, clean.cuda(), args, None, spl)
            assert rel_err(again.cpu().numpy(), got.cpu().numpy()) <= 2e-6


def test_batched_rows_and_errors():
    args = _args("l2", [])
    x = torch.from_numpy(synth.clean_audio(4, 8192)).cuda()
    q = projections.project_l2(x, 0.05)
    assert abs(float(q.norm()) - 0.05) < 1e-6
    with pytest.raises(ValueError):
        train.perturbation_constraint(x[:1], None, _args("snr", []), None, None)
    with pytest.raises(ValueError):
        train.perturbation_constraint(x[:1], None, _args("tv", []), None, None)
    bad = types.SimpleNamespace(**vars(args)); bad.norm_type = "l1"
    with pytest.raises(ValueError):
        train.perturbation_constraint(x[:1], x, bad, None, None)
    # STFT -> iSTFT round trip at full length (size-independent property)
    a2 = _args("max_phon", [])
    p = torch.from_numpy(synth.perturbation(160000) * np.float32(1e-2)).cuda()
    y = fourier_transforms.compute_istft(fourier_transforms.compute_stft(p, a2), a2)
    assert float((y - p[:, : y.shape[1]]).abs().max()) < 2e-8 * 100


def test_compose_clamp_clamp_box_and_argmax():
    """The small boundary entries: train.py:136 compose + clamp, projections.py:37-39 with an asymmetric box,
    loss_helpers.py:26 argmax — each against the torch expression the reference evaluates."""
    from paa_amd import _lib
    from paa_amd.core import loss_helpers
    lib = _lib.lib()
    B, L = 3, 5000
    x = torch.from_numpy(synth.clean_audio(B, L)).cuda() * 12.0          # pushes samples outside [-1, 1]
    p = torch.from_numpy(synth.perturbation(L) * np.float32(0.3)).cuda()
    out = torch.empty_like(x)
    _lib.check(lib.paa_compose_clamp(_lib.ptr(x), _lib.ptr(p), _lib.ptr(out), B, L, _lib.stream_ptr()))
    ref = (x + p).clamp_(-1.0, 1.0)
    assert torch.equal(out, ref) and float((ref.abs() == 1).float().mean()) > 0.01
    q = projections.project_linf(p, -0.05, 0.2)
    assert torch.equal(q, torch.clamp(p, -0.05, 0.2)) and float(q.min()) == pytest.approx(-0.05) and float(q.max()) == pytest.approx(0.2)
    assert torch.equal(projections.project_linf(p, 0.3, 0.1), torch.clamp(p, 0.3, 0.1))     # min > max: everything becomes max
    torch.manual_seed(0)
    lg = torch.randn(4, 499, 32, device="cuda")
    lg[0, 5, 7] = lg[0, 5, 9] = 50.0                                      # a tie: the first maximum wins
    ids = loss_helpers.argmax_ids(lg)
    assert ids.dtype == torch.int16 and torch.equal(ids.long(), torch.argmax(lg, dim=-1)) and int(ids[0, 5]) == 7


def test_spectrum_level_functions_vs_reference_goldens(gold):
    """core/projections.py:68-159 on a complex (B, F, T) tensor, with the reference's argument order
    (tests/golden/spectrum.npz holds what the reference returns for the same calls)."""
    g = gold("spectrum.npz")
    from paa_amd.core import iso
    interp = iso.build_weight_interpolator()
    for L, amp in ((4096, 0.3), (5000, 1e-2)):
        args = _args("fletcher_munson", ["--fm_epsilon", "0.05"])
        spl = build.init_phon_threshold_tensor(args)
        p = torch.from_numpy(np.concatenate([synth.perturbation(L) * np.float32(amp), synth.clean_audio(1, L) * np.float32(4.0)], 0)).cuda()
        S = fourier_transforms.compute_stft(p, args)
        tag = f"L{L}|a{amp:g}"

        def cmp(got, key, tol=TOL):
            ref = g[key]
            assert tuple(got.shape) == tuple(ref.shape[:-1])
            e = float(np.abs(torch.view_as_real(got.contiguous()).cpu().numpy() - ref).max() / np.abs(ref).max())
            print(f"{key}: {e:.2e}")
            assert e <= tol, (key, e)
        cmp(projections.project_min_max_freqs(args, S, 120.0, 20000.0), f"minmax|{tag}")
        cmp(projections.project_min_max_freqs(args, S, 500.0, 3000.0), f"minmax_500_3000|{tag}")
        n = float(projections.compute_fm_weighted_norm_interp(S, interp, args))
        assert n == pytest.approx(float(g[f"fm_norm|{tag}"][0]), rel=2e-5)
        cmp(projections.project_fm_norm(S, args, interp), f"fm|{tag}")
        big = _args("fletcher_munson", ["--fm_epsilon", "1e9"])
        cmp(projections.project_fm_norm(S, big, interp), f"fm_inactive|{tag}")
        cmp(projections.project_phon_level(S, args, spl), f"phon|{tag}")
        # spectrum-level op between the library's own STFT and iSTFT == the fused dispatcher path
        a2 = _args("max_phon", [])
        y = fourier_transforms.compute_istft(projections.project_phon_level(S, a2, spl), a2)
        q = train.perturbation_constraint(p, None, a2, None, spl)
        assert tuple(y.shape) == tuple(q.shape) and float((y - q).abs().max()) <= 2e-6 * float(q.abs().max())


def test_out_of_place_equals_in_place_and_batched_rows():
    """paa_project_to (one fused launch) against paa_project (workspace + copy-back) on the same input, (1, L) and a
    (32, L) batch of rows — bit for bit — and the fused kernel's two workgroup shapes (8 / 16 frames) against each other."""
    from paa_amd import _lib, runtime
    lib = _lib.lib()
    L = 160000
    for norm, extra in (("min_max_freqs", []), ("max_phon", []), ("fletcher_munson", ["--fm_epsilon", "0.5"])):
        args = _args(norm, extra)
        spl = build.init_phon_threshold_tensor(args)
        x = torch.from_numpy(synth.clean_audio(32, L)).cuda() * 0.3
        pr = runtime.get_proj(args, x.device, 32, L)
        pr.set_spl_thresh(spl)
        prm = runtime.params_of(args)
        for rows in (1, 32):
            src = x[:rows].contiguous()
            a = src.clone()
            _lib.check(lib.paa_project(pr.h, prm, _lib.ptr(a), rows, None, 0, L, _lib.stream_ptr()))
            b = torch.empty_like(src)
            _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(src), _lib.ptr(b), rows, None, 0, L, _lib.stream_ptr()))
            assert torch.equal(a, b), (norm, rows)
        if norm != "fletcher_munson":          # row 0 of the batch (run kernel) against the single row (slab kernel): the same arithmetic,
            # written as explicit packed instructions (spec_pk.h): bit for bit
            one = torch.empty(1, L, device="cuda"); many = torch.empty(32, L, device="cuda")
            _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(x[:1].contiguous()), _lib.ptr(one), 1, None, 0, L, _lib.stream_ptr()))
            _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(x), _lib.ptr(many), 32, None, 0, L, _lib.stream_ptr()))
            assert torch.equal(one[0], many[0]), norm
        with pytest.raises(Exception):
            _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(x), _lib.ptr(x), 32, None, 0, L, _lib.stream_ptr()))


@pytest.mark.parametrize("rows,L", [(32, 160000), (24, 40001), (9, 480000), (40, 33000), (300, 2048), (64, 5000), (33, 1100), (600, 700)])
def test_run_kernel_rows_vs_single_row_kernel_and_oracle(rows, L):
    """Batched shapes go through the run-walking kernel (k_spec_run: carried frames, prefetched samples, per-wave
    overlap-add); a single row goes through the slab kernel (k_spec_fused, 8 frames).  Same per-frame arithmetic — written
    as explicit packed instructions (spec_pk.h), so no compiler decision about fmas comes between the two kernels — and the
    same overlap-add order: every checked row of the batch must equal that row projected alone BIT FOR BIT, and the oracle to TOL: odd L (reflect path on every frame, scalar stores), L not a multiple of the hop (zero tail), 30 s rows
    (11 iterations per run), more rows than a round of workgroups, rows of a few frames only (one short run per row, every frame
    touching the reflect padding)."""
    from paa_amd import _lib, runtime
    lib = _lib.lib()
    g = torch.Generator().manual_seed(rows * 1000003 + L)
    x = (torch.randn(rows, L, generator=g) * 0.05)
    xc = x.cuda()
    for norm, extra in (("min_max_freqs", []), ("max_phon", []), ("fletcher_munson", ["--fm_epsilon", "0.5"])):
        args = _args(norm, extra)
        spl = build.init_phon_threshold_tensor(args)
        pr = runtime.get_proj(args, xc.device, rows, L)
        pr.set_spl_thresh(spl)
        prm = runtime.params_of(args)
        many = torch.full((rows, L), float("nan"), device="cuda")
        _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(xc), _lib.ptr(many), rows, None, 0, L, _lib.stream_ptr()))
        assert bool(torch.isfinite(many).all()), norm
        picked = sorted({0, 1, rows // 2, rows - 1})
        for r in picked:
            one = torch.full((1, L), float("nan"), device="cuda")
            _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(xc[r:r + 1].contiguous()), _lib.ptr(one), 1, None, 0, L, _lib.stream_ptr()))
            if norm != "fletcher_munson":
                assert torch.equal(one[0], many[r]), (norm, r, float((one[0] - many[r]).abs().max()))
        if norm == "fletcher_munson":
            # ONE factor for the whole tensor (projections.py:115-133): the oracle on the whole batch would take minutes at 30 s, so
            # check that factor row against row through the STFT -> iSTFT identity (many = s x on the iSTFT's support) ...
            V = 256 * (L // 256)                      # iSTFT length; the tail behind it is zero-filled
            ratio = many[0, :V].double().norm() / xc[0, :V].double().norm()
            for r in (1, rows - 1):
                rr = many[r, :V].double().norm() / xc[r, :V].double().norm()
                assert abs(float(rr / ratio) - 1.0) < 1e-5, (r, float(rr), float(ratio))
            if L > 200000:
                continue
            # ... and the factor itself against the oracle on the whole batch
            ref = OP.perturbation_constraint(x, x, args, OP.spl_thresh_tensor(args)).numpy()
            assert rel_err(many.cpu().numpy(), ref) <= TOL, norm
        else:
            # per-row ops: the oracle on the picked rows alone
            sel = x[picked]
            ref = OP.perturbation_constraint(sel, sel, args, OP.spl_thresh_tensor(args)).numpy()
            assert rel_err(many[picked].cpu().numpy(), ref) <= TOL, norm


def test_out_of_place_equals_in_place_scale_norms():
    """l2 / snr / tv / linf: paa_project_to reduces over and scales the SOURCE directly (no copy in front); it must equal
    paa_project on a clone bit for bit, at (1, L), (32, L) and a length that is not a multiple of four (scalar tail of the
    16-byte scaling loop)."""
    from paa_amd import _lib, runtime
    lib = _lib.lib()
    for L in (160000, 40001):
        x = torch.from_numpy(synth.clean_audio(32, L)).cuda() * 0.3
        clean = torch.from_numpy(synth.clean_audio(4, L)).cuda()
        for norm, extra in (("l2", []), ("snr", ["--snr_db", "40"]), ("tv", []), ("linf", ["--linf_size", "1e-3"])):
            args = _args(norm, extra)
            pr = runtime.get_proj(args, x.device, 32, L)
            prm = runtime.params_of(args)
            for rows in (1, 32):
                src = x[:rows].contiguous()
                a = src.clone()
                _lib.check(lib.paa_project(pr.h, prm, _lib.ptr(a), rows, _lib.ptr(clean), 4, L, _lib.stream_ptr()))
                b = torch.full_like(src, float("nan"))
                _lib.check(lib.paa_project_to(pr.h, prm, _lib.ptr(src), _lib.ptr(b), rows, _lib.ptr(clean), 4, L, _lib.stream_ptr()))
                assert torch.equal(a, b), (norm, rows, L)
                assert torch.equal(src, x[:rows]), "the source must stay untouched"
                assert float((a - src).abs().max()) > 0, (norm, rows, "projection did nothing: pick a smaller bound")
