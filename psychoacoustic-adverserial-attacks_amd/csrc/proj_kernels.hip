// Psychoacoustic constraint projections for gfx950: LDS-staged Stockham FFT (STFT / iSTFT),
// fused spectral projections, wave-level reductions for the time-domain norms.
//
// Reference behaviour reproduced (paths into the reference's src/):
//   core/fourier_transforms.py:4-41      compute_stft / compute_istft
//   core/projections.py:11-159           project_{snr,linf,l2,tv,min_max_freqs,fm_norm,phon_level}
//   training_utils/train.py:27-99        _align_to, _project_frequency_domain, perturbation_constraint
//
// Data layout: waveforms (rows, L) f32 row-major.  Spectra are frame-major (rows, T, F) complex64
// so that one workgroup = one frame writes F contiguous bins.  The frequency-domain projections
// never materialise the spectrum: one kernel does window -> FFT -> per-bin op -> inverse FFT ->
// window and leaves windowed frames (rows, T, n_fft) for the overlap-add kernel.  All data-dependent
// branches of the reference ("scale only if over epsilon") become a predicated scale factor computed
// on the device, so a projection has no host synchronisation.
#include <math.h>

#include <algorithm>
#include <vector>

#include "paa_common.h"
#include "spec_kernels.h"

namespace paa {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

constexpr int FFT_NT = 256;   // threads per frame workgroup
constexpr int RED_NT = 256;
constexpr int MAX_PART = 4096;

enum FrameOp { OP_STFT = 0, OP_MINMAX = 1, OP_PHON = 2, OP_FM = 3, OP_ISTFT = 4 };

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-LDS Stockham autosort FFT, radix 4 with one radix-2 pass when log2(N) is odd.
// INV=false: X[k] = sum x[n] e^{-2 pi i nk/N};  INV=true: unnormalised inverse.
// tw[m] = e^{-2 pi i m / N}.  Returns the buffer that holds the result (natural order).
template <bool INV>
__device__ float2* fft_lds(float2* src, float2* dst, const float2* __restrict__ tw, int N, int log2n) {
    const int tid = threadIdx.x;
    int Ns = 1;
    int rem = log2n;
    while (rem >= 2) {
        const int q = N >> 2;
        const int tstep = N / (Ns * 4);
        for (int j = tid; j < q; j += FFT_NT) {
            const int k = j & (Ns - 1);
            const int m = k * tstep;
            float2 v0 = src[j], v1 = src[j + q], v2 = src[j + 2 * q], v3 = src[j + 3 * q];
            if (k) {
                float2 w1 = tw[m], w2 = tw[2 * m], w3 = tw[3 * m];
                if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
            }
            const float2 a = make_float2(v0.x + v2.x, v0.y + v2.y);
            const float2 b = make_float2(v0.x - v2.x, v0.y - v2.y);
            const float2 c = make_float2(v1.x + v3.x, v1.y + v3.y);
            float2 d = make_float2(v1.x - v3.x, v1.y - v3.y);
            // forward: y1 = b - i d, y3 = b + i d ; inverse: swapped
            float2 id = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);   // (-i d) fwd, (+i d) inv
            const int o = ((j - k) << 2) + k;
            dst[o] = make_float2(a.x + c.x, a.y + c.y);
            dst[o + Ns] = make_float2(b.x + id.x, b.y + id.y);
            dst[o + 2 * Ns] = make_float2(a.x - c.x, a.y - c.y);
            dst[o + 3 * Ns] = make_float2(b.x - id.x, b.y - id.y);
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
        Ns <<= 2;
        rem -= 2;
    }
    if (rem == 1) {
        const int h = N >> 1;
        const int tstep = N / (Ns * 2);
        for (int j = tid; j < h; j += FFT_NT) {
            const int k = j & (Ns - 1);
            float2 v0 = src[j], v1 = src[j + h];
            if (k) {
                float2 w1 = tw[k * tstep];
                if (INV) w1.y = -w1.y;
                v1 = cmul(v1, w1);
            }
            const int o = ((j - k) << 1) + k;
            dst[o] = make_float2(v0.x + v1.x, v0.y + v1.y);
            dst[o + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;
}

struct FrameArgs {
    const float* x;      // (rows, L) waveform            [all ops but ISTFT]
    const float* S_in;   // (rows, T, F) complex64        [ISTFT]
    float* S_out;        // (rows, T, F) complex64        [STFT]
    float* frames;       // (rows, T, N) windowed inverse frames
    double* part;        // (rows*T) per-frame partial sums [FM]
    const float2* tw;
    const float* win;
    const float* fm;     // [10][F], negative => out of the interpolator's frequency range
    const float* thr;    // [F] spl_thresh
    const float* thr_max;  // [1]
    int L, T, N, log2n, hop, F;
    float bin_hz, min_f, max_f, phon_ref;
};

// One workgroup = one STFT frame: window -> FFT -> OP on the F one-sided bins -> inverse FFT -> window.
template <int OP>
__global__ __launch_bounds__(FFT_NT) void k_frame(FrameArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* buf0 = reinterpret_cast<float2*>(smem_raw);
    float2* buf1 = buf0 + a.N;
    __shared__ double red[FFT_NT / 64];
    const int t = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
    const int N = a.N, F = a.F, half = N >> 1;
    float2* X;
    if (OP != OP_ISTFT) {
        const float* x = a.x + (size_t)row * a.L;
        for (int n = tid; n < N; n += FFT_NT) {
            int s = t * a.hop + n - half;                  // center=True: reflect pad n_fft/2
            if (s < 0) s = -s;
            if (s >= a.L) s = 2 * (a.L - 1) - s;
            buf0[n] = make_float2(x[s] * a.win[n], 0.f);
        }
        __syncthreads();
        X = fft_lds<false>(buf0, buf1, a.tw, N, a.log2n);
    } else {
        const float2* S = reinterpret_cast<const float2*>(a.S_in) + ((size_t)row * a.T + t) * F;
        for (int k = tid; k < F; k += FFT_NT) buf0[k] = S[k];
        __syncthreads();
        X = buf0;
    }
    float2* Y = (X == buf0) ? buf1 : buf0;

    if (OP == OP_STFT) {
        float2* S = reinterpret_cast<float2*>(a.S_out) + ((size_t)row * a.T + t) * F;
        for (int k = tid; k < F; k += FFT_NT) S[k] = X[k];
        return;
    }
    double acc = 0.0;
    for (int k = tid; k < F; k += FFT_NT) {
        float2 v = X[k];
        if (OP == OP_MINMAX) {
            // projections.py:68-80: keep bins OUTSIDE [min, max]
            const float f = (float)k * a.bin_hz;
            const float m = ((f < a.min_f) || (f > a.max_f)) ? 1.f : 0.f;
            v.x *= m; v.y *= m;
        } else if (OP == OP_PHON) {
            // projections.py:138-159
            const float mag = hypotf(v.x, v.y);
            const float mag_db = 20.f * log10f(mag + 1e-8f);
            const float thr = (a.thr[k] - a.thr_max[0]) + a.phon_ref;
            const float db = (mag_db > thr) ? thr : mag_db;
            const float mc = exp10f(db / 20.f);
            const float ang = atan2f(v.y, v.x);
            float sn, cs;
            sincosf(ang, &sn, &cs);
            v = make_float2(mc * cs, mc * sn);
        } else if (OP == OP_FM) {
            // projections.py:83-113: weight = bilinear(iso grid)(10 log10(|S|^2 + 1e-10), f_bin); OOB -> 1
            const float mag = hypotf(v.x, v.y);
            const float pw = mag * mag;
            const float s = 10.f * log10f(pw + 1e-10f);
            float w = 1.f;
            const float w0 = a.fm[k];
            if (w0 >= 0.f && s >= 0.f && s <= 90.f) {
                int i = (int)floorf(s * 0.1f);
                i = i > 8 ? 8 : i;
                if (s <= 10.f * (float)i && i > 0) i -= 1;       // searchsorted(side='left') - 1
                const float ys = (s - 10.f * (float)i) * 0.1f;
                w = a.fm[i * F + k] * (1.f - ys) + a.fm[(i + 1) * F + k] * ys;
            }
            acc += (double)(pw * w);
        }
        if (k == 0 || k == half) v.y = 0.f;               // irfft ignores Im(DC), Im(Nyquist)
        Y[k] = v;
        if (k > 0 && k < half) Y[N - k] = make_float2(v.x, -v.y);
    }
    if (OP == OP_FM) {
        const double tot = block_sum<double, FFT_NT>(acc, red);
        if (tid == 0) a.part[(size_t)row * a.T + t] = tot;
    }
    __syncthreads();
    float2* y = fft_lds<true>(Y, X, a.tw, N, a.log2n);
    float* fr = a.frames + ((size_t)row * a.T + t) * N;
    const float inv = 1.f / (float)N;
    for (int n = tid; n < N; n += FFT_NT) fr[n] = y[n].x * inv * a.win[n];
}

// Overlap-add + envelope division + trim + _align_to (train.py:27-35): out (rows, out_len).
// Samples >= hop*(T-1) are zero (right zero-pad).  scal[0] (if non-null) is a uniform scale factor.
__global__ void k_ola(const float* __restrict__ frames, const float* __restrict__ win, const float* __restrict__ scal,
                      float* __restrict__ out, int T, int N, int hop, int out_len) {
    const int row = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= out_len) return;
    float v = 0.f;
    if (m < hop * (T - 1)) {
        const int s = m + (N >> 1);
        int t1 = s / hop;
        if (t1 > T - 1) t1 = T - 1;
        int t0 = (s - N + hop) / hop;                      // ceil((s - N + 1) / hop) for s-N+1 >= 0
        if (s - N + 1 <= 0) t0 = 0;
        float sum = 0.f, env = 0.f;
        const float* fr = frames + (size_t)row * T * N;
        for (int t = t0; t <= t1; ++t) {
            const int n = s - t * hop;
            sum += fr[(size_t)t * N + n];
            const float w = win[n];
            env += w * w;
        }
        v = sum / env;
        if (scal) v *= scal[0];
    }
    out[(size_t)row * out_len + m] = v;
}

// projections.py:116-133 project_fm_norm: scale = eps / max(norm, 1e-8) if norm > eps else 1
__global__ __launch_bounds__(RED_NT) void k_fm_finalize(const double* __restrict__ part, int n, float eps,
                                                      float* __restrict__ scal) {
    __shared__ double red[RED_NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += RED_NT) s += part[i];
    s = block_sum<double, RED_NT>(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf((float)s);
        scal[0] = (norm <= eps) ? 1.f : eps / fmaxf(norm, 1e-8f);
        scal[1] = norm;
    }
}

// Second launch of a fused spectral projection: dst = src * scale, the scale being the predicated FM factor computed from
// the per-workgroup partial sums of the first launch (every block re-sums the few hundred partials: no third launch);
// npart = 0: plain copy (min_max_freqs / max_phon, in-place form only).
__global__ __launch_bounds__(RED_NT) void k_spec_finish(const float* __restrict__ src, float* __restrict__ dst, int64_t n,
                                                      const double* __restrict__ part, int npart, float eps, float* __restrict__ scal) {
    __shared__ double red[RED_NT / 64];
    float scale = 1.f;
    if (npart > 0) {
        double s = 0.0;
        for (int i = threadIdx.x; i < npart; i += RED_NT) s += part[i];
        s = block_sum<double, RED_NT>(s, red);
        const float norm = sqrtf((float)s);
        scale = (norm <= eps) ? 1.f : eps / fmaxf(norm, 1e-8f);
        if (blockIdx.x == 0 && threadIdx.x == 0 && scal) { scal[0] = scale; scal[1] = norm; }
    }
    const int64_t n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        for (int64_t i = (int64_t)blockIdx.x * RED_NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RED_NT) {
            float4 v = s4[i];
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            d4[i] = v;
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * RED_NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_NT) dst[i] = src[i] * scale;
    } else {
        for (int64_t i = (int64_t)blockIdx.x * RED_NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_NT) dst[i] = src[i] * scale;
    }
}

// ---- time-domain reductions --------------------------------------------------------------------
enum RedMode { RED_SQ = 0, RED_TV = 1 };

// Blocks [0, g1) reduce x1 (n1 elements, rows of length L1), blocks [g1, g1+g2) reduce x2.
template <int MODE>
__global__ __launch_bounds__(RED_NT) void k_reduce2(const float* __restrict__ x1, int64_t n1, int L1, int g1,
                                                  const float* __restrict__ x2, int64_t n2, int L2, int g2,
                                                  double* __restrict__ part) {
    __shared__ double red[RED_NT / 64];
    const bool first = (int)blockIdx.x < g1;
    const float* x = first ? x1 : x2;
    const int64_t n = first ? n1 : n2;
    const int L = first ? L1 : L2;
    const int g = first ? g1 : g2;
    const int b = first ? blockIdx.x : blockIdx.x - g1;
    float acc = 0.f;
    double dacc = 0.0;
    int cnt = 0;
    for (int64_t i = (int64_t)b * RED_NT + threadIdx.x; i < n; i += (int64_t)g * RED_NT) {
        if (MODE == RED_SQ) {
            const float v = x[i];
            acc += v * v;
        } else {
            if ((int)(i % L) != L - 1) acc += fabsf(x[i + 1] - x[i]);
        }
        if (++cnt == 64) { dacc += (double)acc; acc = 0.f; cnt = 0; }
    }
    dacc += (double)acc;
    dacc = block_sum<double, RED_NT>(dacc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = dacc;
}

struct ApplyArgs {
    const double* part;
    int g1, g2;
    float eps;          // l2_size | tv_epsilon
    float snr_db, snr_linear;
    double numel_clean, numel_p;
    float* scal;        // [0] scale applied, [1..] diagnostics
    const float* ext;   // optional device [sum clean^2, TV(clean)] supplied by the caller (data-parallel runs)
    const float* ext_clips;   // optional device [1]: number of clean clips over ALL ranks (an exact small integer in f32);
    double clip_len;          //   numel_clean = ext_clips[0] * clip_len then replaces the host value above
};

// p = src * scale (src == p: in place; otherwise the out-of-place form reads the caller's source directly — no copy in front)
template <int NORM>
__global__ __launch_bounds__(RED_NT) void k_apply_scale(const float* src, float* p, int64_t n, ApplyArgs a) {
    __shared__ double red[RED_NT / 64];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < a.g1; i += RED_NT) s1 += a.part[i];
    for (int i = threadIdx.x; i < a.g2; i += RED_NT) s2 += a.part[a.g1 + i];
    s1 = block_sum<double, RED_NT>(s1, red);
    s2 = block_sum<double, RED_NT>(s2, red);
    if (a.ext) s1 = (double)a.ext[NORM == PAA_NORM_TV ? 1 : 0];
    const double numel_clean = a.ext_clips ? (double)a.ext_clips[0] * a.clip_len : a.numel_clean;
    float scale = 1.f;
    if (NORM == PAA_NORM_L2) {                     // projections.py:41-46 (s2 = sum p^2)
        const float norm = sqrtf((float)s2);
        if (norm > a.eps) scale = a.eps / norm;
    } else if (NORM == PAA_NORM_SNR) {             // projections.py:11-35 (s1 = sum clean^2, s2 = sum p^2)
        const float sp = (float)(s1 / numel_clean);
        const float np_ = (float)(s2 / a.numel_p);
        const float cur = 10.f * log10f(sp / (np_ + 1e-12f));
        if (!(cur >= a.snr_db)) {
            const float target = sqrtf(sp / a.snr_linear * (float)numel_clean);
            const float cn = sqrtf((float)s2);
            if (!(cn < 1e-8f)) scale = target / cn;
        }
    } else if (NORM == PAA_NORM_TV) {              // projections.py:56-66 (s1 = TV(clean), s2 = TV(p))
        const float eps = a.eps * (float)s1;
        const float tv = (float)s2;
        if (tv > eps) scale = eps / tv;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.scal) { a.scal[0] = scale; a.scal[1] = (float)s1; a.scal[2] = (float)s2; }
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(p)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* p4 = reinterpret_cast<float4*>(p);
        for (int64_t i = (int64_t)blockIdx.x * RED_NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RED_NT) {
            float4 v = s4[i];
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            p4[i] = v;
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * RED_NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_NT) p[i] = src[i] * scale;
    } else {
        for (int64_t i = (int64_t)blockIdx.x * RED_NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_NT) p[i] = src[i] * scale;
    }
}

__global__ void k_clamp(const float* src, float* p, int64_t n, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = fminf(fmaxf(src[i], lo), hi);       // torch.clamp propagates NaN; fminf/fmaxf do not: see DESIGN.md
}

__global__ void k_sign_step(float* __restrict__ p, const float* __restrict__ g, float lr, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = g[i];
    const float s = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : v);   // sign(0) = 0, sign(NaN) = NaN
    p[i] = p[i] + lr * s;
}

__global__ void k_compose_clamp(const float* __restrict__ x, const float* __restrict__ p, float* __restrict__ out,
                                int B, int L) {
    const int64_t n = (int64_t)B * L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(x[i] + p[i % L], -1.f), 1.f);
}

}  // namespace paa

using namespace paa;

struct paa_proj {
    int n_fft, hop, win, sr, F, log2n, max_batch, max_len;
    float2* d_tw = nullptr;
    float* d_win = nullptr;
    float* d_fm = nullptr;
    float* d_thr = nullptr;
    float* d_thr_max = nullptr;
    float* d_frames = nullptr;
    double* d_part = nullptr;
    float* d_scal = nullptr;
    size_t frames_floats = 0;
};

extern "C" const char* paa_last_error(void) { return paa::g_err.c_str(); }
// 300 + 1 if the library was built with -DPAA_EXPERIMENTS (diagnostic configurations and environment switches present)
extern "C" int paa_version(void) {
#ifdef PAA_EXPERIMENTS
    return 301;
#else
    return 300;
#endif
}

extern "C" paa_status paa_proj_set_spl_thresh(paa_proj* h, const float* spl) {
    if (!h || !spl) PAA_FAIL(PAA_ERR_ARG, "paa_proj_set_spl_thresh: null argument");
    float mx = spl[0];
    for (int i = 1; i < h->F; ++i) mx = spl[i] > mx ? spl[i] : mx;
    PAA_HIP(hipMemcpy(h->d_thr, spl, sizeof(float) * h->F, hipMemcpyHostToDevice));
    PAA_HIP(hipMemcpy(h->d_thr_max, &mx, sizeof(float), hipMemcpyHostToDevice));
    return PAA_OK;
}

extern "C" paa_status paa_proj_create(paa_proj** out, int n_fft, int hop, int win, int sr, const double* fm_table,
                                      const float* spl_thresh, int max_batch, int max_len) {
    if (!out) PAA_FAIL(PAA_ERR_ARG, "paa_proj_create: out is null");
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096)
        PAA_FAIL(PAA_ERR_ARG, "n_fft=%d unsupported: must be a power of two in [64, 4096]", n_fft);
    if (win != n_fft) PAA_FAIL(PAA_ERR_ARG, "win_length=%d != n_fft=%d is not supported", win, n_fft);
    if (hop <= 0 || hop > n_fft) PAA_FAIL(PAA_ERR_ARG, "hop_length=%d out of range", hop);
    if (max_batch < 1 || max_len <= n_fft / 2) PAA_FAIL(PAA_ERR_SIZE, "max_batch/max_len too small");
    paa_proj* h = new paa_proj();
    h->n_fft = n_fft; h->hop = hop; h->win = win; h->sr = sr; h->F = n_fft / 2 + 1; h->log2n = log2n;
    h->max_batch = max_batch; h->max_len = max_len;
    const int F = h->F;
    std::vector<float2> tw(n_fft);
    std::vector<float> w(n_fft);
    for (int m = 0; m < n_fft; ++m) {
        const double ang = -2.0 * M_PI * (double)m / (double)n_fft;
        tw[m] = make_float2((float)cos(ang), (float)sin(ang));
        w[m] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)m / (double)n_fft));   // torch.hann_window (periodic)
    }
    // torch.istft refuses windows whose overlap-added squared envelope touches zero (NOLA); so does this library
    // (hop == n_fft with a Hann window would divide by zero in the overlap-add)
    for (int r = 0; r < hop; ++r) {
        double env = 0.0;
        for (int n = r; n < n_fft; n += hop) env += (double)w[n] * (double)w[n];
        if (env < 1e-11) { delete h; PAA_FAIL(PAA_ERR_ARG, "window overlap-add envelope is zero at offset %d (hop_length=%d, n_fft=%d): iSTFT undefined", r, hop, n_fft); }
    }
    std::vector<float> fm(10 * F, 1.f), thr(F, 0.f);
    if (fm_table) for (int i = 0; i < 10 * F; ++i) fm[i] = (float)fm_table[i];
    const int Tmax = 1 + max_len / hop;
    h->frames_floats = (size_t)max_batch * Tmax * n_fft;
#define PC(e) do { hipError_t _e = (e); if (_e != hipSuccess) { paa::set_error(std::string(#e) + ": " + hipGetErrorString(_e)); paa_proj_destroy(h); return PAA_ERR_HIP; } } while (0)
    PC(hipMalloc(&h->d_tw, sizeof(float2) * n_fft));
    PC(hipMalloc(&h->d_win, sizeof(float) * n_fft));
    PC(hipMalloc(&h->d_fm, sizeof(float) * 10 * F));
    PC(hipMalloc(&h->d_thr, sizeof(float) * F));
    PC(hipMalloc(&h->d_thr_max, sizeof(float)));
    PC(hipMalloc(&h->d_frames, sizeof(float) * h->frames_floats));
    PC(hipMalloc(&h->d_part, sizeof(double) * (MAX_PART + (size_t)max_batch * Tmax)));
    PC(hipMalloc(&h->d_scal, sizeof(float) * 8));
    PC(hipMemcpy(h->d_tw, tw.data(), sizeof(float2) * n_fft, hipMemcpyHostToDevice));
    PC(hipMemcpy(h->d_win, w.data(), sizeof(float) * n_fft, hipMemcpyHostToDevice));
    PC(hipMemcpy(h->d_fm, fm.data(), sizeof(float) * 10 * F, hipMemcpyHostToDevice));
    PC(hipMemset(h->d_scal, 0, sizeof(float) * 8));
#undef PC
    if (spl_thresh) {
        paa_status s = paa_proj_set_spl_thresh(h, spl_thresh);
        if (s != PAA_OK) { paa_proj_destroy(h); return s; }
    } else {
        thr.assign(F, 0.f);
        paa_proj_set_spl_thresh(h, thr.data());
    }
    *out = h;
    return PAA_OK;
}

#ifdef PAA_SPEC_STAMP
// diagnostic builds only (tools/fft_stamps.py): the per-iteration s_memtime stamps k_spec_run left in the FM partial array
extern "C" paa_status paa_debug_spec_stamps(paa_proj* h, long long* host, int n) {
    PAA_HIP(hipDeviceSynchronize());
    PAA_HIP(hipMemcpy(host, h->d_part + MAX_PART, sizeof(long long) * n, hipMemcpyDeviceToHost));
    return PAA_OK;
}
#endif

extern "C" void paa_proj_destroy(paa_proj* h) {
    if (!h) return;
    void* ptrs[] = {h->d_tw, h->d_win, h->d_fm, h->d_thr, h->d_thr_max, h->d_frames, h->d_part, h->d_scal};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    delete h;
}

static paa_status check_rows(const paa_proj* h, int rows, int L, const char* who) {
    if (rows < 1 || rows > h->max_batch) PAA_FAIL(PAA_ERR_SIZE, "%s: rows=%d exceeds max_batch=%d", who, rows, h->max_batch);
    if (L > h->max_len) PAA_FAIL(PAA_ERR_SIZE, "%s: L=%d exceeds max_len=%d", who, L, h->max_len);
    if (L <= h->n_fft / 2) PAA_FAIL(PAA_ERR_SIZE, "%s: L=%d must exceed n_fft/2=%d (reflect padding)", who, L, h->n_fft / 2);
    return PAA_OK;
}

static FrameArgs frame_args(const paa_proj* h, int L, int T) {
    FrameArgs a{};
    a.tw = h->d_tw; a.win = h->d_win; a.fm = h->d_fm; a.thr = h->d_thr; a.thr_max = h->d_thr_max;
    a.frames = h->d_frames; a.part = h->d_part + MAX_PART;
    a.L = L; a.T = T; a.N = h->n_fft; a.log2n = h->log2n; a.hop = h->hop; a.F = h->F;
    a.bin_hz = (float)((double)h->sr / (double)h->n_fft);
    return a;
}

static bool fused_geometry(const paa_proj* h) { return h->n_fft == 1024 && h->hop == 256 && h->win == 1024; }
static SpecArgs spec_args(const paa_proj* h, int L, int T, int out_len) {
    SpecArgs a{};
    a.tw = h->d_tw; a.win = h->d_win; a.fm = h->d_fm; a.thr = h->d_thr; a.thr_max = h->d_thr_max;
    a.part = h->d_part + MAX_PART;
    a.L = L; a.T = T; a.out_len = out_len;
    a.bin_hz = (float)((double)h->sr / (double)h->n_fft);
    return a;
}
static int spec_op_of(int norm) {
    return norm == PAA_NORM_MIN_MAX_FREQS ? SOP_MINMAX : norm == PAA_NORM_MAX_PHON ? SOP_PHON : norm == PAA_NORM_FLETCHER_MUNSON ? SOP_FM : SOP_NONE;
}

template <int OP>
static paa_status launch_frames(const paa_proj* h, const FrameArgs& a, int rows, hipStream_t st) {
    const size_t lds = 2 * sizeof(float2) * h->n_fft;
    hipLaunchKernelGGL(k_frame<OP>, dim3(a.T, rows), dim3(FFT_NT), lds, st, a);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

extern "C" paa_status paa_stft(paa_proj* h, const float* d_x, int B, int L, float* d_out, void* stream) {
    if (!h || !d_x || !d_out) PAA_FAIL(PAA_ERR_ARG, "paa_stft: null argument");
    PAA_TRY(check_rows(h, B, L, "paa_stft"));
    if (fused_geometry(h)) {
        SpecArgs sa = spec_args(h, L, 1 + L / h->hop, 0);
        sa.x = d_x; sa.S_out = d_out;
        return spec_stft(sa, B, (hipStream_t)stream);
    }
    FrameArgs a = frame_args(h, L, 1 + L / h->hop);
    a.x = d_x; a.S_out = d_out;
    return launch_frames<OP_STFT>(h, a, B, (hipStream_t)stream);
}

extern "C" paa_status paa_istft(paa_proj* h, const float* d_S, int B, int T, float* d_out, void* stream) {
    if (!h || !d_S || !d_out) PAA_FAIL(PAA_ERR_ARG, "paa_istft: null argument");
    if (T < 2) PAA_FAIL(PAA_ERR_SIZE, "paa_istft: T=%d", T);
    if (B < 1 || B > h->max_batch || (size_t)B * T * h->n_fft > h->frames_floats)
        PAA_FAIL(PAA_ERR_SIZE, "paa_istft: B=%d T=%d exceeds the workspace", B, T);
    hipStream_t st = (hipStream_t)stream;
    if (fused_geometry(h)) {
        SpecArgs sa = spec_args(h, 0, T, h->hop * (T - 1));
        sa.S_in = d_S; sa.out = d_out;
        return spec_istft(sa, B, st);
    }
    FrameArgs a = frame_args(h, 0, T);
    a.S_in = d_S;
    PAA_TRY(launch_frames<OP_ISTFT>(h, a, B, st));
    const int out_len = h->hop * (T - 1);
    hipLaunchKernelGGL(k_ola, dim3(cdiv(out_len, 256), B), dim3(256), 0, st, h->d_frames, h->d_win, (const float*)nullptr,
                       d_out, T, h->n_fft, h->hop, out_len);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

// d_src == nullptr: in place on d_p.  Otherwise out of place: reads d_src, writes d_p (the two must not overlap).
static paa_status project_impl(paa_proj* h, const paa_params* prm, float* d_p, int rows_p, const float* d_clean, int B, int L,
                               const float* d_ext, double ext_numel, void* stream, const float* d_src = nullptr,
                               const float* d_ext_clips = nullptr) {
    if (!h || !prm || !d_p) PAA_FAIL(PAA_ERR_ARG, "paa_project: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int nt = prm->norm_type;
    const int64_t n = (int64_t)rows_p * L;
    if (rows_p < 1 || L < 2) PAA_FAIL(PAA_ERR_SIZE, "paa_project: rows_p=%d L=%d", rows_p, L);
    if (nt < PAA_NORM_L2 || nt > PAA_NORM_MAX_PHON) PAA_FAIL(PAA_ERR_BAD_NORM, "Unknown norm_type: %d", nt);
    const bool spectral = nt == PAA_NORM_FLETCHER_MUNSON || nt == PAA_NORM_MIN_MAX_FREQS || nt == PAA_NORM_MAX_PHON;
    if (spectral && fused_geometry(h)) {
        // one fused launch STFT -> per-bin op -> iSTFT + overlap-add (spec_kernels.hip), then the scale / copy-back launch
        PAA_TRY(check_rows(h, rows_p, L, "paa_project"));
        const int T = 1 + L / h->hop;
        SpecArgs a = spec_args(h, L, T, L);
        a.min_f = prm->min_freq_attack; a.max_f = prm->max_freq_attack; a.phon_ref = prm->phon_reference_db;
        a.x = d_src ? d_src : d_p;
        a.out = d_src ? d_p : h->d_frames;                     // in place: through the workspace (neighbouring workgroups re-read the halo)
        int npart = 0;
        PAA_TRY(spec_project(a, spec_op_of(nt), rows_p, &npart, st));
        const bool fm = nt == PAA_NORM_FLETCHER_MUNSON;
        if (fm || !d_src) {
            hipLaunchKernelGGL(k_spec_finish, dim3(std::min(cdiv(n, (int64_t)RED_NT * 4), 1024)), dim3(RED_NT), 0, st,
                               (const float*)a.out, d_p, n, (const double*)a.part, fm ? npart : 0, prm->fm_epsilon, h->d_scal);
            PAA_LAUNCH_CHECK();
        }
        return PAA_OK;
    }
    const bool scale_type = nt == PAA_NORM_L2 || nt == PAA_NORM_SNR || nt == PAA_NORM_TV || nt == PAA_NORM_LINF;
    if (d_src && !scale_type) {          // the generic frame kernels work in place: copy first
        PAA_HIP(hipMemcpyAsync(d_p, d_src, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
    }
    const float* d_in = d_src ? d_src : d_p;      // scale-type norms and linf: the reduction and the scaling pass read the source directly
    switch (nt) {
        case PAA_NORM_FLETCHER_MUNSON:
        case PAA_NORM_MIN_MAX_FREQS:
        case PAA_NORM_MAX_PHON: {
            PAA_TRY(check_rows(h, rows_p, L, "paa_project"));
            const int T = 1 + L / h->hop;
            FrameArgs a = frame_args(h, L, T);
            a.x = d_p;
            a.min_f = prm->min_freq_attack; a.max_f = prm->max_freq_attack; a.phon_ref = prm->phon_reference_db;
            const float* scal = nullptr;
            if (nt == PAA_NORM_MIN_MAX_FREQS) PAA_TRY(launch_frames<OP_MINMAX>(h, a, rows_p, st));
            else if (nt == PAA_NORM_MAX_PHON) PAA_TRY(launch_frames<OP_PHON>(h, a, rows_p, st));
            else {
                PAA_TRY(launch_frames<OP_FM>(h, a, rows_p, st));
                hipLaunchKernelGGL(k_fm_finalize, dim3(1), dim3(RED_NT), 0, st, (const double*)a.part, rows_p * T,
                                   prm->fm_epsilon, h->d_scal);
                PAA_LAUNCH_CHECK();
                scal = h->d_scal;
            }
            hipLaunchKernelGGL(k_ola, dim3(cdiv(L, 256), rows_p), dim3(256), 0, st, h->d_frames, h->d_win, scal, d_p, T,
                               h->n_fft, h->hop, L);
            PAA_LAUNCH_CHECK();
            return PAA_OK;
        }
        case PAA_NORM_LINF: {
            hipLaunchKernelGGL(k_clamp, dim3(std::min(cdiv(n, 256), 2048)), dim3(256), 0, st, d_in, d_p, n, -prm->linf_size,
                               prm->linf_size);
            PAA_LAUNCH_CHECK();
            return PAA_OK;
        }
        case PAA_NORM_L2:
        case PAA_NORM_SNR:
        case PAA_NORM_TV: {
            if (nt != PAA_NORM_L2 && !d_ext && (!d_clean || B < 1)) {
                if (nt == PAA_NORM_SNR) PAA_FAIL(PAA_ERR_NEED_CLEAN, "SNR projection requires clean_audio ro compare to");
                PAA_FAIL(PAA_ERR_NEED_CLEAN, "TV projection can benefit from clean_audio for bounds");
            }
            const int64_t nc = (nt == PAA_NORM_L2 || d_ext) ? 0 : (int64_t)B * L;
            const int g1 = nc ? std::min(cdiv(nc, (int64_t)RED_NT * 16), 2048) : 0;
            const int g2 = std::min(cdiv(n, (int64_t)RED_NT * 8), 1024);
            if (nt == PAA_NORM_TV)
                hipLaunchKernelGGL(k_reduce2<RED_TV>, dim3(g1 + g2), dim3(RED_NT), 0, st, d_clean, nc, L, g1, d_in,
                                   n, L, g2, h->d_part);
            else
                hipLaunchKernelGGL(k_reduce2<RED_SQ>, dim3(g1 + g2), dim3(RED_NT), 0, st, d_clean, nc, L, g1, d_in,
                                   n, L, g2, h->d_part);
            PAA_LAUNCH_CHECK();
            ApplyArgs a{};
            a.part = h->d_part; a.g1 = g1; a.g2 = g2; a.scal = h->d_scal;
            a.numel_clean = d_ext ? ext_numel : (double)nc; a.numel_p = (double)n; a.ext = d_ext;
            a.ext_clips = d_ext ? d_ext_clips : nullptr; a.clip_len = (double)L;
            a.snr_db = prm->snr_db; a.snr_linear = (float)pow(10.0, (double)prm->snr_db / 10.0);
            a.eps = (nt == PAA_NORM_L2) ? prm->l2_size : prm->tv_epsilon;
            const int ga = std::min(cdiv(n, 256), 1024);
            if (nt == PAA_NORM_L2) hipLaunchKernelGGL(k_apply_scale<PAA_NORM_L2>, dim3(ga), dim3(RED_NT), 0, st, d_in, d_p, n, a);
            else if (nt == PAA_NORM_SNR) hipLaunchKernelGGL(k_apply_scale<PAA_NORM_SNR>, dim3(ga), dim3(RED_NT), 0, st, d_in, d_p, n, a);
            else hipLaunchKernelGGL(k_apply_scale<PAA_NORM_TV>, dim3(ga), dim3(RED_NT), 0, st, d_in, d_p, n, a);
            PAA_LAUNCH_CHECK();
            return PAA_OK;
        }
        default:
            PAA_FAIL(PAA_ERR_BAD_NORM, "Unknown norm_type: %d", nt);
    }
}

extern "C" paa_status paa_project(paa_proj* h, const paa_params* prm, float* d_p, int rows_p, const float* d_clean,
                                  int B, int L, void* stream) {
    return project_impl(h, prm, d_p, rows_p, d_clean, B, L, nullptr, 0.0, stream);
}

extern "C" paa_status paa_project_to(paa_proj* h, const paa_params* prm, const float* d_src, float* d_dst, int rows_p,
                                     const float* d_clean, int B, int L, void* stream) {
    if (!d_src || !d_dst) PAA_FAIL(PAA_ERR_ARG, "paa_project_to: null argument");
    const float* lo = d_src < d_dst ? d_src : d_dst;
    const float* hi = d_src < d_dst ? d_dst : d_src;
    if (rows_p > 0 && L > 0 && lo + (int64_t)rows_p * L > hi) PAA_FAIL(PAA_ERR_ARG, "paa_project_to: source and destination overlap");
    return project_impl(h, prm, d_dst, rows_p, d_clean, B, L, nullptr, 0.0, stream, d_src);
}

// core/projections.py:68-159 called directly on a spectrum (the reference's project_min_max_freqs / project_fm_norm /
// project_phon_level take the complex (B, F, T) STFT tensor): d_S_in, d_S_out (B, T, F) complex64 frame-major
// (in place allowed); prm->norm_type selects the op.  Works for any frame geometry (pure per-bin arithmetic).
extern "C" paa_status paa_spectrum_project(paa_proj* h, const paa_params* prm, const float* d_S_in, float* d_S_out, int B, int T,
                                           void* stream) {
    if (!h || !prm || !d_S_in || !d_S_out) PAA_FAIL(PAA_ERR_ARG, "paa_spectrum_project: null argument");
    if (B < 1 || T < 1) PAA_FAIL(PAA_ERR_SIZE, "paa_spectrum_project: B=%d T=%d", B, T);
    if (h->F != 513) PAA_FAIL(PAA_ERR_ARG, "paa_spectrum_project: n_fft=%d (only 1024 is built)", h->n_fft);
    const int nt = prm->norm_type;
    const int op = spec_op_of(nt);
    if (op == SOP_NONE) PAA_FAIL(PAA_ERR_BAD_NORM, "paa_spectrum_project: norm_type %d is not a frequency-domain projection", nt);
    hipStream_t st = (hipStream_t)stream;
    SpecArgs a = spec_args(h, 0, T, 0);
    a.part = h->d_part;
    a.min_f = prm->min_freq_attack; a.max_f = prm->max_freq_attack; a.phon_ref = prm->phon_reference_db;
    a.S_in = d_S_in;
    if (op != SOP_FM) {
        a.S_out = d_S_out;
        return spec_apply(a, op, B, nullptr, nullptr, st);
    }
    int npart = 0;
    a.S_out = nullptr;                                       // pass 1: weighted power only
    PAA_TRY(spec_apply(a, SOP_FM, B, nullptr, &npart, st));
    hipLaunchKernelGGL(k_fm_finalize, dim3(1), dim3(RED_NT), 0, st, (const double*)a.part, npart, prm->fm_epsilon, h->d_scal);
    PAA_LAUNCH_CHECK();
    a.S_out = d_S_out; a.part = nullptr;                     // pass 2: S * predicated scale
    return spec_apply(a, SOP_NONE, B, h->d_scal, nullptr, st);
}

// core/projections.py:83-113 compute_fm_weighted_norm_interp: sqrt(sum |S|^2 w(10 log10(|S|^2 + 1e-10), f)) -> d_out[0]
extern "C" paa_status paa_fm_weighted_norm(paa_proj* h, const float* d_S, int B, int T, float* d_out, void* stream) {
    if (!h || !d_S || !d_out) PAA_FAIL(PAA_ERR_ARG, "paa_fm_weighted_norm: null argument");
    if (B < 1 || T < 1) PAA_FAIL(PAA_ERR_SIZE, "paa_fm_weighted_norm: B=%d T=%d", B, T);
    if (h->F != 513) PAA_FAIL(PAA_ERR_ARG, "paa_fm_weighted_norm: n_fft=%d (only 1024 is built)", h->n_fft);
    hipStream_t st = (hipStream_t)stream;
    SpecArgs a = spec_args(h, 0, T, 0);
    a.part = h->d_part;
    a.S_in = d_S; a.S_out = nullptr;
    int npart = 0;
    PAA_TRY(spec_apply(a, SOP_FM, B, nullptr, &npart, st));
    hipLaunchKernelGGL(k_fm_finalize, dim3(1), dim3(RED_NT), 0, st, (const double*)a.part, npart, 0.f, h->d_scal);
    PAA_LAUNCH_CHECK();
    PAA_HIP(hipMemcpyAsync(d_out, h->d_scal + 1, sizeof(float), hipMemcpyDeviceToDevice, st));
    return PAA_OK;
}

extern "C" paa_status paa_project_ext(paa_proj* h, const paa_params* prm, float* d_p, int rows_p, const float* d_clean_stats,
                                      const float* d_clip_count, double clean_numel, int L, void* stream) {
    if (!d_clean_stats) PAA_FAIL(PAA_ERR_NEED_CLEAN, "paa_project_ext: clean statistics are required");
    if (!d_clip_count && !(clean_numel > 0.0)) PAA_FAIL(PAA_ERR_ARG, "paa_project_ext: neither a device clip count nor a positive clean_numel");
    return project_impl(h, prm, d_p, rows_p, nullptr, 0, L, d_clean_stats, clean_numel, stream, nullptr, d_clip_count);
}

namespace paa {
__global__ __launch_bounds__(RED_NT) void k_sum_parts(const double* __restrict__ part, int g1, int g2, float* __restrict__ out,
                                                      float* __restrict__ count_out, float count) {
    __shared__ double red[RED_NT / 64];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < g1; i += RED_NT) s1 += part[i];
    for (int i = threadIdx.x; i < g2; i += RED_NT) s2 += part[g1 + i];
    s1 = block_sum<double, RED_NT>(s1, red);
    s2 = block_sum<double, RED_NT>(s2, red);
    if (threadIdx.x == 0) { out[0] = (float)s1; out[1] = (float)s2; if (count_out) count_out[0] = count; }
}
}  // namespace paa

extern "C" paa_status paa_batch_stats(paa_proj* h, const float* d_clean, int B, int L, float* d_out2, float* d_clip_count,
                                      void* stream) {
    if (!h || !d_clean || !d_out2) PAA_FAIL(PAA_ERR_ARG, "paa_batch_stats: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nc = (int64_t)B * L;
    const int g = std::min(cdiv(nc, (int64_t)RED_NT * 16), 2048);
    hipLaunchKernelGGL(k_reduce2<RED_SQ>, dim3(g), dim3(RED_NT), 0, st, d_clean, nc, L, g, (const float*)nullptr, (int64_t)0, L, 0,
                       h->d_part);
    PAA_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce2<RED_TV>, dim3(g), dim3(RED_NT), 0, st, d_clean, nc, L, g, (const float*)nullptr, (int64_t)0, L, 0,
                       h->d_part + g);
    PAA_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(RED_NT), 0, st, (const double*)h->d_part, g, g, d_out2, d_clip_count, (float)B);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

extern "C" paa_status paa_sign_step(float* d_p, const float* d_grad, float lr, int L, void* stream) {
    if (!d_p || !d_grad) PAA_FAIL(PAA_ERR_ARG, "paa_sign_step: null argument");
    hipLaunchKernelGGL(k_sign_step, dim3(cdiv(L, 256)), dim3(256), 0, (hipStream_t)stream, d_p, d_grad, lr, L);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

// core/projections.py:37-39 project_linf(p, min_val, max_val) with an arbitrary box (the dispatcher passes +-linf_size).
extern "C" paa_status paa_clamp(float* d_p, int64_t n, float lo, float hi, void* stream) {
    if (!d_p) PAA_FAIL(PAA_ERR_ARG, "paa_clamp: null argument");
    if (n < 1) PAA_FAIL(PAA_ERR_SIZE, "paa_clamp: n=%lld", (long long)n);
    // lo > hi: every element becomes hi, as torch.clamp documents (min(max(x, lo), hi))
    hipLaunchKernelGGL(k_clamp, dim3(std::min(cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, (const float*)d_p, d_p, n, lo, hi);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

extern "C" paa_status paa_compose_clamp(const float* d_clean, const float* d_p, float* d_out, int B, int L, void* stream) {
    if (!d_clean || !d_p || !d_out) PAA_FAIL(PAA_ERR_ARG, "paa_compose_clamp: null argument");
    hipLaunchKernelGGL(k_compose_clamp, dim3(std::min(cdiv((int64_t)B * L, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       d_clean, d_p, d_out, B, L);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}
