// Fused STFT -> per-bin projection -> iSTFT path for the default frame geometry (n_fft = win = 1024, hop = 256):
// see spec_kernels.hip.  Other geometries stay on the generic one-frame-per-workgroup kernels of proj_kernels.hip.
#pragma once
#include "paa_common.h"

namespace paa {

enum SpecOp { SOP_NONE = 0, SOP_MINMAX = 1, SOP_PHON = 2, SOP_FM = 3 };

struct SpecArgs {
    const float* x;        // (rows, L) waveform in                       [waveform source]
    const float* S_in;     // (rows, T, F) complex64 in                   [spectrum source]
    float* out;            // (rows, out_len) waveform out                [waveform destination]
    float* S_out;          // (rows, T, F) complex64 out                  [spectrum destination]
    double* part;          // FM: one partial sum of |S|^2 w per (row, workgroup)
    const float2* tw;      // e^{-2 pi i m / 1024}, m < 1024
    const float* win;      // periodic Hann window, 1024
    const float* fm;       // [10][513] Fletcher-Munson weights lerped to the bins (< 0: outside the interpolator)
    const float* thr;      // [513] max_phon contour
    const float* thr_max;  // [1]
    int L, T, out_len;     // samples per row in, frames, samples per row out (>= 256 (T - 1); the tail is zero-filled)
    float bin_hz, min_f, max_f, phon_ref;
};

// rows x (L) waveform -> per-bin op -> waveform (train.py:38-66 _project_frequency_domain with _align_to), in place allowed
// when out == x is NOT used by a later workgroup — callers pass a separate output buffer.
paa_status spec_project(const SpecArgs& a, int op, int rows, int* n_part, hipStream_t st);
paa_status spec_stft(const SpecArgs& a, int rows, hipStream_t st);     // fourier_transforms.py:20-29
paa_status spec_istft(const SpecArgs& a, int rows, hipStream_t st);    // fourier_transforms.py:31-41
// per-bin op on a caller-supplied spectrum (projections.py:68-159 called on a (B, F, T) tensor), frame-major storage
paa_status spec_apply(const SpecArgs& a, int op, int rows, const float* scale, int* n_part, hipStream_t st);
int spec_groups(int T, int rows, int op, bool src_spec);        // workgroups per row spec_project / spec_istft launch (size of the FM partial array / rows)

}  // namespace paa
