// Fused frequency-domain projection for gfx950, default frame geometry (n_fft = win = 1024, hop = 256).
//
// Replaces, in ONE launch, the reference's  compute_stft -> project_{min_max_freqs,fm_norm,phon_level} -> compute_istft ->
// _align_to  chain (training_utils/train.py:38-66, core/fourier_transforms.py:20-41, core/projections.py:68-159):
//
//  * ONE WAVE PER FRAME.  A real 1024-point frame is a 512-point complex FFT of z[m] = x[2m] + i x[2m+1] plus a split
//    post-pass; 512 = 8 x 8 x 8, so the FFT is three radix-8 passes with 8 complex values per lane in registers.  The arithmetic
//    is written as single packed-f32 instructions with operand modifiers (spec_pk.h: a butterfly is 26 instructions, a complex
//    product 2, no moves, and every kernel rounds a frame the same way).  The exchange between the first two passes is an 8 x 8
//    register / lane transpose in registers (v_permlane32_swap, v_permlane16_swap, masked DPP moves); the second goes through a
//    5 KB per-wave LDS buffer with a conflict-free layout (rows of 10 complex) — no workgroup barrier inside a frame, twiddles
//    and window in registers.
//  * A lane owns bins lane + 64 j (its own registers after the forward transform) and their mirrors 512 - k, which sit in lane
//    64 - lane, register 7 - j: the split post-pass and the inverse pre-pass pull them through the LDS crossbar (ds_bpermute),
//    not through memory.  Lane 0 carries (DC, Nyquist) as its pair 0 and bin 256 as a ninth bin.  The per-bin projection runs on
//    those registers and feeds the inverse pre-pass directly, then three inverse radix-8 passes, window, and the windowed frame
//    stays in LDS.
//  * Single rows / small batches (k_spec_fused): a workgroup of 8 waves handles 8 consecutive frames of one row and overlap-adds
//    them in LDS: output hop-block j is the sum of frames j-3..j, so 8 frames give 5 complete blocks (the 3-frame halo is
//    recomputed by the neighbouring workgroup).  Batches (k_spec_run): a workgroup of 12 waves WALKS a run of consecutive blocks,
//    12 frames per iteration; the three frames the next blocks reach back to stay in LDS, twiddles are loaded once per run, the
//    next frame's samples are requested before the iteration's LDS-only barrier, and wave w overlap-adds the block whose newest
//    frame is its own.  Envelope division, the FM scale's partial sum, trimming of the reflect padding and the zero tail of
//    _align_to happen on the way out.  Windowed frames never touch HBM: algorithmic traffic is read p + write p (measured
//    1.04 x that on the batch).
//  * FM norm (projections.py:83-133): the same launch leaves sqrt-free partial sums (one double per workgroup, only
//    frames the workgroup OWNS are counted); the scale is a predicated factor applied by the copy-back / scale kernel.
#include <stdlib.h>

#include <algorithm>

#include "spec_kernels.h"
#include "spec_pk.h"

namespace paa {

namespace {

constexpr int XB = 640;                 // complex slots of one wave's exchange buffer (5120 B)
constexpr int N = 1024, N2 = 512, HOP = 256, F = 513;

// per-lane constants of the wave FFT (lane = a = 8 n1 + n0 in the first pass)
struct LaneTw {
    v2f a[7];       // W512^(lane k0), k0 = 1..7
    v2f b[7];       // W64^((lane & 7) k1), k1 = 1..7
    v2f pm[4];      // (-i / 2) e^{-2 pi i k / 1024}, k = lane + 64 j: the split post-pass's factor; the inverse pre-pass uses 2 conj(pm) = i conj(p)
    v2f w[8];       // window at samples 2m, 2m + 1, m = lane + 64 n2
};
__device__ __forceinline__ void lane_tw(LaneTw& t, const float2* __restrict__ tw, const float* __restrict__ win, int lane) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        t.a[k - 1] = pk_v(tw[(2 * lane * k) & 1023]);
        t.b[k - 1] = pk_v(tw[(16 * (lane & 7) * k) & 1023]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 p = tw[lane + 64 * j];
        t.pm[j] = v2f{0.5f * p.y, -0.5f * p.x};
    }
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) t.w[n2] = pk_v(*reinterpret_cast<const float2*>(win + 2 * (lane + 64 * n2)));
}

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// Workgroup barrier that orders LDS only: __syncthreads() also drains vmcnt, i.e. it would wait for the NEXT frame's samples
// (requested just before it) and for the overlap-add's global stores — a full memory round trip per iteration on the critical path.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 512-point complex FFT of one wave.  In: x[n2] = z[64 n2 + lane].  Out: x[k2] = Z[lane + 64 k2].
// S = -1: forward; S = +1: unnormalised inverse.  xb: this wave's exchange buffer (second exchange only).  3 x 26 + 14 x 2 packed
// instructions + the register transpose.
template <int S>
__device__ __forceinline__ void wave_fft512(v2f (&x)[8], v2f* xb, const LaneTw& t, int lane) {
    pk_dft8<S>(x);
#pragma unroll
    for (int k = 1; k < 8; ++k) x[k] = S < 0 ? pk_cmul(x[k], t.a[k - 1]) : pk_cmul_conj(x[k], t.a[k - 1]);
    // E1[k0][a = 8 n1 + n0] -> lane (k0, n0), register n1: register index against lane bits 5:3, in registers (spec_pk.h)
    pk_transpose_hi(x);
    const int k0 = lane >> 3, n0 = lane & 7;
    pk_dft8<S>(x);
#pragma unroll
    for (int k = 1; k < 8; ++k) x[k] = S < 0 ? pk_cmul(x[k], t.b[k - 1]) : pk_cmul_conj(x[k], t.b[k - 1]);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) xb[(k0 + 8 * k1) * 10 + n0] = x[k1];             // E2[t = k0 + 8 k1][n0], rows of 10
    wave_fence();
    {
        const float4* r = reinterpret_cast<const float4*>(xb + lane * 10);           // 80-byte rows: conflict-free ds_read_b128
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = r[q];
            x[2 * q] = v2f{v.x, v.y};
            x[2 * q + 1] = v2f{v.z, v.w};
        }
    }
    wave_fence();
    pk_dft8<S>(x);
}

struct BinCtx {
    const float* fm; const float* thr; float thr_off;      // thr_off = phon_ref - max(thr)
    float bin_hz, min_f, max_f;
    int fs = F;             // row stride of the FM table (the run kernel's LDS copy pads its rows to FM_LS)
    unsigned inb = 0;       // run kernel, FM: bit `slot` set = the lane's bin of that slot lies inside the interpolator (fm[k] >= 0), read once per run
    bool have_inb = false;
};
constexpr int FM_LS = 576;  // 9 x 64: the two table rows of a lerp are then ONE ds_read2st64_b32

// projections.py:68-159 on one bin.  FM leaves the bin untouched and returns |S|^2 w in `wsum`.
template <int OP>
__device__ __forceinline__ float2 bin_op(float2 v, int k, const BinCtx& c, float& wsum, int slot = 0) {
    if (OP == SOP_MINMAX) {                      // projections.py:68-80: keep bins OUTSIDE [min, max]
        const float f = (float)k * c.bin_hz;
        const float m = ((f < c.min_f) || (f > c.max_f)) ? 1.f : 0.f;
        return make_float2(v.x * m, v.y * m);
    } else if (OP == SOP_PHON) {                 // projections.py:138-159
        // log10 / exp10 through the hardware log2 / exp2 (arguments are normal and far from overflow here): the absolute
        // error of 20 log10(.) stays at the rounding of its own result (~1e-6 dB), as with libm's log10f.  |S| and 1 / |S| both
        // come from ONE v_rsq_f32 of |S|^2 (1 ulp; the correctly rounded sqrtf + division they replace were ~25 issue slots per bin).
        const float pw = v.x * v.x + v.y * v.y;
        const bool nz = pw > 1e-30f;             // below: |S| < 1e-15 acts as 0 (the reference would keep its phase on a 1e-8 magnitude)
        const float rs = __builtin_amdgcn_rsqf(nz ? pw : 1.f);
        const float mag = nz ? pw * rs : 0.f;
        const float mag_db = 6.02059991327962390f * __builtin_amdgcn_logf(mag + 1e-8f);            // 20 log10(2) log2(x)
        const float thr = c.thr[k] + c.thr_off;
        const float db = (mag_db > thr) ? thr : mag_db;
        const float mc = __builtin_amdgcn_exp2f(db * 0.166096404744368118f);                          // 10^(db / 20)
        // mc e^{i angle(S)}: S / |S| is that unit phasor; angle(0) = 0
        const float sc = mc * rs;
        return nz ? make_float2(v.x * sc, v.y * sc) : make_float2(mc, 0.f);
    } else if (OP == SOP_FM) {                   // projections.py:83-113: bilinear iso-grid weight at (10 log10(|S|^2 + 1e-10), f_bin)
        const float pw = v.x * v.x + v.y * v.y;  // abs() ** 2 without the round trip through the square root
        const float s = 3.01029995663981195f * __builtin_amdgcn_logf(pw + 1e-10f);                   // 10 log10(x) via log2
        float w = 1.f;
        const bool inside = c.have_inb ? ((c.inb >> slot) & 1u) != 0 : c.fm[k] >= 0.f;
        if (inside && s >= 0.f && s <= 90.f) {
            int i = (int)floorf(s * 0.1f);
            i = i > 8 ? 8 : i;
            if (s <= 10.f * (float)i && i > 0) i -= 1;       // searchsorted(side='left') - 1
            const float ys = (s - 10.f * (float)i) * 0.1f;
            const float* tb = c.fm + i * c.fs + k;
            w = tb[0] * (1.f - ys) + tb[c.fs] * ys;
        }
        wsum += pw * w;
        return v;
    }
    return v;
}

// Raw samples of frame t of one row: raw[n2] = (x[2m], x[2m + 1]), m = lane + 64 n2 (center=True: reflect pad 512).
__device__ __forceinline__ void frame_load(const SpecArgs& a, const float* __restrict__ xr, int t, int lane, float2 (&raw)[8]) {
    const int s0 = t * HOP - N2;                              // first sample of the frame
    const bool inside = s0 >= 0 && s0 + N <= a.L && ((a.L & 1) == 0);
    if (inside) {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) raw[n2] = *reinterpret_cast<const float2*>(xr + s0 + 2 * (lane + 64 * n2));
    } else {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            int i0 = s0 + 2 * (lane + 64 * n2), i1 = i0 + 1;
            if (i0 < 0) i0 = -i0;
            if (i0 >= a.L) i0 = 2 * (a.L - 1) - i0;
            if (i1 < 0) i1 = -i1;
            if (i1 >= a.L) i1 = 2 * (a.L - 1) - i1;
            raw[n2] = make_float2(xr[i0], xr[i1]);
        }
    }
}

// One frame, start to end, by one wave.
//   SRC_SPEC: the spectrum comes from S_in (iSTFT) instead of the waveform;  DST_SPEC: stop after the forward transform and
//   write the spectrum (STFT).  Otherwise the windowed inverse frame is left in xb as 1024 floats.
//   raw: the frame's samples as frame_load leaves them (unused with SRC_SPEC).
// Returns this lane's share of sum |S|^2 w (FM).
template <int OP, bool SRC_SPEC, bool DST_SPEC>
__device__ __forceinline__ float wave_frame(const SpecArgs& a, const BinCtx& c, const LaneTw& tw, float2* xbf, int row, int t, int lane,
                                            const float2 (&raw)[8]) {
    v2f* xb = reinterpret_cast<v2f*>(xbf);
    v2f x[8];
    v2f Xk[4], Xm[4];                                         // bins k = lane + 64 j and their mirrors 512 - k (lane 0, j = 0: DC and Nyquist)
    float wsum = 0.f;
    const v2f half = {0.5f, 0.5f}, two = {2.f, 2.f};
    // Bin ownership: a lane owns bins k_j = lane + 64 j (j < 4) — its OWN registers x[j] after the forward transform — and their mirrors
    // 512 - k_j, which sit in lane 64 - lane, register 7 - j: one ds_bpermute pair per mirror (a pull through the LDS crossbar, no
    // memory), instead of writing Z in natural order and reading the pairs back.  Lane 0 is special three times: its mirrors are its own
    // registers 8 - j (j = 1..3), its pair j = 0 is (DC, Nyquist), and its register 4 is bin 256, its own mirror.
    const bool l0 = lane == 0;
    const int mirror = ((64 - lane) & 63) << 2;              // ds_bpermute address (bytes) of the partner lane
    v2f X256 = {0.f, 0.f};
    if (!SRC_SPEC) {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) x[n2] = pk_mul(pk_v(raw[n2]), tw.w[n2]);
        wave_fft512<-1>(x, xb, tw, lane);
        // split post-pass: X[k] = E + T, X[512 - k] = conj(E - T), E = (Z[k] + conj Z[512-k]) / 2, T = w_k (-i/2)(Z[k] - conj Z[512-k])
        v2f m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = pk_bpermute(mirror, x[7 - j]);
        m[1] = l0 ? x[7] : m[1]; m[2] = l0 ? x[6] : m[2]; m[3] = l0 ? x[5] : m[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v2f E2 = pk_add_conj(x[j], m[j]), dz = pk_sub_conj(x[j], m[j]);
            const v2f T = pk_cmul(tw.pm[j], dz);
            Xk[j] = pk_fma(E2, half, T);
            Xm[j] = pk_fma_cnjm(E2, half, T);
        }
        Xk[0] = l0 ? v2f{x[0].x + x[0].y, 0.f} : Xk[0];       // DC
        Xm[0] = l0 ? v2f{x[0].x - x[0].y, 0.f} : Xm[0];       // Nyquist
        X256 = v2f{x[4].x, -x[4].y};                          // lane 0: X[256] = conj Z[256]
    } else {
        const float2* S = reinterpret_cast<const float2*>(a.S_in) + ((size_t)row * a.T + t) * F;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            Xk[j] = pk_v(S[k]);
            Xm[j] = pk_v(S[N2 - k]);
        }
        X256 = pk_v(S[N2 / 2]);
    }
    if (DST_SPEC) {
        float2* S = reinterpret_cast<float2*>(a.S_out) + ((size_t)row * a.T + t) * F;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            S[k] = pk_f(Xk[j]);
            S[N2 - k] = pk_f(Xm[j]);
        }
        if (l0) S[N2 / 2] = pk_f(X256);
        return 0.f;
    }
    // ---- per-bin projection: 8 bins per lane + bin 256 (evaluated by every lane, kept and counted by lane 0) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        Xk[j] = pk_v(bin_op<OP>(pk_f(Xk[j]), k, c, wsum, j));
        Xm[j] = pk_v(bin_op<OP>(pk_f(Xm[j]), N2 - k, c, wsum, 4 + j));
    }
    {
        float w256 = 0.f;
        X256 = pk_v(bin_op<OP>(pk_f(X256), N2 / 2, c, w256, 8));
        if (l0) wsum += w256;
    }
    // ---- inverse pre-pass: Z'[k] = Ee + i Oo, Z'[512 - k] = conj(Ee - i Oo), Ee = X[k] + conj X[512-k], i Oo = 2 (X[k] - conj X[512-k]) conj(pm_k)
    {
        v2f y[4], r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v2f Ee = pk_add_conj(Xk[j], Xm[j]), d = pk_sub_conj(Xk[j], Xm[j]);
            const v2f W = pk_cmul_conj(d, tw.pm[j]);
            x[j] = pk_fma(W, two, Ee);                        // Z'[lane + 64 j]: already where the inverse transform wants it
            y[j] = pk_fma_cnjn(W, two, Ee);                   // Z'[512 - k_j]: lane 64 - lane, register 7 - j
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = pk_bpermute(mirror, y[j]);
        // lane 0: Z'[0] from (DC, Nyquist) — irfft ignores their imaginary parts —, Z'[64 (8 - j)] = its own y[j], Z'[256] = 2 conj X[256]
        x[0] = l0 ? v2f{Xk[0].x + Xm[0].x, Xk[0].x - Xm[0].x} : x[0];
        x[7] = l0 ? y[1] : r[0];
        x[6] = l0 ? y[2] : r[1];
        x[5] = l0 ? y[3] : r[2];
        x[4] = l0 ? v2f{2.f * X256.x, -2.f * X256.y} : r[3];
    }
    wave_fft512<+1>(x, xb, tw, lane);
    // x[k2] = 1024 z[lane + 64 k2]: samples 2m, 2m + 1 of the frame; window again (istft), keep in LDS
    const v2f inv = {1.f / (float)N, 1.f / (float)N};
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) xb[lane + 64 * k2] = pk_mul(pk_mul(x[k2], inv), tw.w[k2]);
    return wsum;
}

// NW * FPW frames of one row per workgroup -> NW * FPW - 3 output hop-blocks.  grid: (groups, rows).
// FPW = 2 (batched shapes, round 3): every wave computes TWO frames, f = wave and f = NW + wave, one after the other — its twiddle /
// window registers serve both, and the 3-frame halo a workgroup recomputes is 3 of 24 frames instead of 3 of 16.  The first frame's
// windowed result (1024 floats) is parked in a result area behind the exchange buffers before the second frame reuses the wave's
// exchange buffer: 12 x (5 + 4) KB = 108 KB of LDS (+ 20 KB for the FM table), one workgroup per CU as before.
template <int OP, bool SRC_SPEC, int NW, int FPW = 1>
__global__ __launch_bounds__(NW * 64) void k_spec_fused(SpecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 xbuf[];                   // [NW][XB] exchange | [NW][512] parked first frames (FPW = 2)
    constexpr int NFR = NW * FPW, NOUT = NFR - 3;
    constexpr int TAIL = NW * XB + (FPW == 2 ? NW * (N / 2) : 0);                    // float2 slots in front of the small arrays
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, row = blockIdx.y;
    const int c0 = 2 + g * NOUT;                              // first output hop-block (padded-signal block index)
    float2* xb = xbuf + wave * XB;
    float* park = reinterpret_cast<float*>(xbuf + NW * XB);   // [NW][1024] (FPW = 2)
    BinCtx c;
    c.fm = a.fm; c.thr = a.thr; c.thr_off = a.phon_ref - (OP == SOP_PHON ? a.thr_max[0] : 0.f);
    c.bin_hz = a.bin_hz; c.min_f = a.min_f; c.max_f = a.max_f;
    if (OP == SOP_FM && NFR >= 16) {
        // batched shape: the 10 x 513 weight table goes to LDS once per workgroup: the per-bin lookups (row chosen by the bin's own level)
        // were three dependent global loads per bin on the frame's critical path
        float* fml = reinterpret_cast<float*>(xbuf + TAIL) + 64;
        for (int i = tid; i < 10 * F; i += NW * 64) fml[i] = a.fm[i];
        c.fm = fml;
        __syncthreads();
    }
    float wsum = 0.f;
    {
        LaneTw tw;
        lane_tw(tw, a.tw, a.win, lane);                       // once per wave: serves both of its frames
#pragma unroll 1
        for (int q = 0; q < FPW; ++q) {
            const int f = q * NW + wave;                      // frame slot inside the workgroup
            const int t = c0 - 3 + f;                         // frame index of the row
            if (t >= 0 && t < a.T) {
                float2 raw[8];
                if (!SRC_SPEC) frame_load(a, a.x + (size_t)row * a.L, t, lane, raw);
                const float ws = wave_frame<OP, SRC_SPEC, false>(a, c, tw, xb, row, t, lane, raw);
                if (f >= 3 || g == 0) wsum += ws;             // halo frames belong to the previous workgroup's sum
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) xb[lane + 64 * k2] = make_float2(0.f, 0.f);
            }
            if (FPW == 2 && q == 0) {                         // park the first frame: 1024 floats, 4 x 16 bytes per lane
                wave_fence();
                const float4* src = reinterpret_cast<const float4*>(xb);
                float4* dst = reinterpret_cast<float4*>(park + wave * N);
                const float4 v0 = src[lane], v1 = src[lane + 64], v2 = src[lane + 128], v3 = src[lane + 192];
                dst[lane] = v0; dst[lane + 64] = v1; dst[lane + 128] = v2; dst[lane + 192] = v3;
                wave_fence();
            }
        }
    }
    double dsum = 0.0;
    if (OP == SOP_FM) dsum = wave_sum((double)wsum);
    __syncthreads();                                          // every frame of the workgroup is in LDS
    if (OP == SOP_FM) {
        double* red = reinterpret_cast<double*>(xbuf + TAIL);
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += red[w];
            a.part[(size_t)row * gridDim.x + g] = s;
        }
    }
    // overlap-add: block j = c0 + jj <- frames j-3..j = frame slots jj..jj+3 at offsets 768, 512, 256, 0 (+ r)
    const float* fb = reinterpret_cast<const float*>(xbuf);
    auto frame_ptr = [&](int f) -> const float* {            // windowed samples of frame slot f
        if (FPW == 2) return f < NW ? park + f * N : fb + (f - NW) * (2 * XB);
        return fb + f * (2 * XB);
    };
    const int valid_len = HOP * (a.T - 1);
    float* outr = a.out + (size_t)row * a.out_len;
    for (int i = tid; i < NOUT * HOP; i += NW * 64) {
        const int jj = i >> 8, r = i & 255;
        const int j = c0 + jj;
        const int m = HOP * (j - 2) + r;                      // output sample (reflect padding trimmed)
        if (m >= valid_len) continue;
        float sum = 0.f, env = 0.f;
        if (j >= 3 && j < a.T) {
            // interior: all four frames exist and the periodic Hann window's squared overlap-add is exactly 3/2
            // (sum over the four quarter-period shifts of (1/2 - 1/2 cos)^2: the cos and cos 2x terms cancel): one multiply, no division
#pragma unroll
            for (int q = 0; q < 4; ++q) sum += frame_ptr(jj + q)[HOP * (3 - q) + r];
            outr[m] = sum * 0.666666686534881591796875f;
            continue;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tq = j - 3 + q;
                if (tq >= 0 && tq < a.T) {
                    const int n = HOP * (3 - q) + r;
                    sum += frame_ptr(jj + q)[n];
                    const float w = a.win[n];
                    env += w * w;
                }
            }
        }
        outr[m] = sum / env;
    }
    // _align_to (train.py:27-35): samples past the iSTFT length are zero; the last workgroup of the row writes them
    if (g == (int)gridDim.x - 1)
        for (int m = valid_len + tid; m < a.out_len; m += NW * 64) outr[m] = 0.f;
}

// Batched shapes: a workgroup of NW waves walks a RUN of consecutive output hop-blocks [J0, J1) of one row, NW frames per
// iteration (one per wave), instead of one (NW x FPW)-frame slab per launch slot:
//  * the three windowed frames a run's next blocks still need stay in LDS (carry) — the 3-frame halo is recomputed once per
//    RUN (3 of ~80 frames at (32, 160000): 8 runs per row), not once per 24 frames;
//  * twiddles / window / the FM table are loaded once per run, and the next frame's samples are requested before the
//    barrier that ends the current iteration, so their latency hides under the overlap-add and the barrier skew;
//  * wave w's frame t = J0 - 3 + it NW + w is the NEWEST frame of output block j = t, so after the barrier each wave
//    overlap-adds "its" block from its own result and the three slots before it: four ds_read_b128 and one 16-byte store per lane.
// grid: (runs per row, rows), bpr = blocks per run.
template <int OP, int NW>
__global__ __launch_bounds__(NW * 64) void k_spec_run(SpecArgs a, int bpr) {
    extern __shared__ __attribute__((aligned(16))) float2 xbuf[];                   // [NW][XB] exchange / result | [3][512] carry | small arrays
    constexpr int TAIL = NW * XB + 3 * N2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, row = blockIdx.y;
    const int J0 = 2 + g * bpr, JE = 1 + a.T;                 // output blocks of the row: 2 .. T (padded-signal block index)
    const int J1 = J0 + bpr < JE ? J0 + bpr : JE;
    if (J0 >= J1) {                                           // an empty run at the end of the row (uniform for the workgroup)
        if (OP == SOP_FM && tid == 0) a.part[(size_t)row * gridDim.x + g] = 0.0;
        return;
    }
    float2* const xb = xbuf + wave * XB;
    float* carry = reinterpret_cast<float*>(xbuf + NW * XB);  // [3][1024]: frame slots NW-3 .. NW-1 of the previous iteration
    BinCtx c;
    c.fm = a.fm; c.thr = a.thr; c.thr_off = a.phon_ref - (OP == SOP_PHON ? a.thr_max[0] : 0.f);
    c.bin_hz = a.bin_hz; c.min_f = a.min_f; c.max_f = a.max_f;
    if (OP == SOP_FM) {
        float* fml = reinterpret_cast<float*>(xbuf + TAIL) + 64;
        for (int i = tid; i < 10 * F; i += NW * 64) fml[(i / F) * FM_LS + i % F] = a.fm[i];
        // which of the lane's nine bins lie inside the interpolator does not change from frame to frame
        unsigned inb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            inb |= (a.fm[lane + 64 * j] >= 0.f ? 1u : 0u) << j;
            inb |= (a.fm[N2 - (lane + 64 * j)] >= 0.f ? 1u : 0u) << (4 + j);
        }
        inb |= (a.fm[N2 / 2] >= 0.f ? 1u : 0u) << 8;
        c.fm = fml; c.fs = FM_LS; c.inb = inb; c.have_inb = true;
        __syncthreads();
    }
    const int niter = (J1 - J0 + 3 + NW - 1) / NW;
    const int t_end = J1 < a.T ? J1 : a.T;                    // frames this run needs: [J0 - 3, t_end) (block j needs frames j-3 .. j)
    const float* xr = a.x + (size_t)row * a.L;
    float* outr = a.out + (size_t)row * a.out_len;
    const bool vec_ok = ((a.out_len & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
    LaneTw tw;
    lane_tw(tw, a.tw, a.win, lane);
    float2 raw[8];
    int t = J0 - 3 + wave;
    if (t >= 0 && t < t_end) frame_load(a, xr, t, lane, raw);
    float wsum = 0.f;
#if defined(PAA_SPEC_STAMP) && !defined(PAA_EXPERIMENTS)
#error "PAA_SPEC_STAMP is a diagnostic build: add -DPAA_EXPERIMENTS"
#endif
#ifdef PAA_SPEC_STAMP
    // workgroup (0, 0): [wave][iteration][5] = loop top, frame done, past barrier A, overlap-add done, past barrier B
    long long* stamp = reinterpret_cast<long long*>(a.part) + 64;
#define SPEC_STAMP(slot) do { if (OP == SOP_MINMAX && g == 0 && row == 0 && lane == 0 && it < 8) stamp[(wave * 8 + it) * 5 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPEC_STAMP(slot) do { } while (0)
#endif
#pragma unroll 1
    for (int it = 0; it < niter; ++it, t += NW) {
        SPEC_STAMP(0);
        if (t >= 0 && t < t_end) {
            const float ws = wave_frame<OP, false, false>(a, c, tw, xb, row, t, lane, raw);
            if (t >= J0 || g == 0) wsum += ws;                // halo frames belong to the previous run's sum
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) xb[lane + 64 * k2] = make_float2(0.f, 0.f);
        }
        SPEC_STAMP(1);
        if (it + 1 < niter && t + NW >= 0 && t + NW < t_end) frame_load(a, xr, t + NW, lane, raw);
        lds_barrier();                                        // this iteration's frames are in LDS
        SPEC_STAMP(2);
        if (t >= J0 && t < J1) {                              // block j = t <- frames t-3 .. t = slots wave-3 .. wave at offsets 768, 512, 256, 0
            const int j = t;
            const float* fq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sl = wave - 3 + q;
                fq[q] = (sl >= 0 ? reinterpret_cast<const float*>(xbuf + sl * XB) : carry + (3 + sl) * N) + HOP * (3 - q);
            }
            float* o = outr + (size_t)HOP * (j - 2);
            if (j >= 3 && j < a.T) {
                // interior: all four frames exist; the periodic Hann window's squared overlap-add is exactly 3/2
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(fq[q] + 4 * lane);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                const float s = 0.666666686534881591796875f;
                acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s;
                if (vec_ok) *reinterpret_cast<float4*>(o + 4 * lane) = acc;
                else { o[4 * lane] = acc.x; o[4 * lane + 1] = acc.y; o[4 * lane + 2] = acc.z; o[4 * lane + 3] = acc.w; }
            } else {
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * lane + e;
                    float sum = 0.f, env = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int tq = j - 3 + q;
                        if (tq >= 0 && tq < a.T) {
                            sum += fq[q][r];
                            const float w = a.win[HOP * (3 - q) + r];
                            env += w * w;
                        }
                    }
                    o[r] = sum / env;
                }
            }
        }
        SPEC_STAMP(3);
        lds_barrier();                                        // every block of the iteration is read: results may be overwritten
        SPEC_STAMP(4);
        if (wave >= NW - 3 && it + 1 < niter) {               // the next iteration's blocks reach back three frames
            const float4* src = reinterpret_cast<const float4*>(xb);
            float4* dst = reinterpret_cast<float4*>(carry + (wave - (NW - 3)) * N);
            const float4 v0 = src[lane], v1 = src[lane + 64], v2 = src[lane + 128], v3 = src[lane + 192];
            dst[lane] = v0; dst[lane + 64] = v1; dst[lane + 128] = v2; dst[lane + 192] = v3;
        }
    }
    if (OP == SOP_FM) {
        const double dsum = wave_sum((double)wsum);
        double* red = reinterpret_cast<double*>(xbuf + TAIL);
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += red[w];
            a.part[(size_t)row * gridDim.x + g] = s;
        }
    }
    // _align_to (train.py:27-35): samples past the iSTFT length are zero; the run that ends the row writes them
    if (J1 == JE) {
        const int valid_len = HOP * (a.T - 1);
        for (int m = valid_len + tid; m < a.out_len; m += NW * 64) outr[m] = 0.f;
    }
}

// STFT only: one wave per frame, 4 frames per workgroup.  grid: (ceil(T / 4), rows)
__global__ __launch_bounds__(256) void k_spec_stft(SpecArgs a) {
    __shared__ __attribute__((aligned(16))) float2 xbuf[4 * XB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + wave, row = blockIdx.y;
    if (t >= a.T) return;
    LaneTw tw;
    lane_tw(tw, a.tw, a.win, lane);
    BinCtx c{};
    float2 raw[8];
    frame_load(a, a.x + (size_t)row * a.L, t, lane, raw);
    wave_frame<SOP_NONE, false, true>(a, c, tw, xbuf + wave * XB, row, t, lane, raw);
}

// per-bin op on a spectrum in memory (frame-major (rows, T, F) complex64), optional uniform scale
template <int OP>
__global__ __launch_bounds__(256) void k_spec_apply(SpecArgs a, int64_t n, const float* __restrict__ scale) {
    __shared__ double red[4];
    BinCtx c;
    c.fm = a.fm; c.thr = a.thr; c.thr_off = a.phon_ref - (OP == SOP_PHON ? a.thr_max[0] : 0.f);
    c.bin_hz = a.bin_hz; c.min_f = a.min_f; c.max_f = a.max_f;
    const float2* S = reinterpret_cast<const float2*>(a.S_in);
    float2* O = reinterpret_cast<float2*>(a.S_out);
    const float sc = scale ? scale[0] : 1.f;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % F);
        float w = 0.f;
        float2 v = bin_op<OP>(S[i], k, c, w);
        acc += (double)w;
        if (O) O[i] = make_float2(v.x * sc, v.y * sc);
    }
    if (OP == SOP_FM && a.part) {
        acc = block_sum<double, 256>(acc, red);
        if (threadIdx.x == 0) a.part[blockIdx.x] = acc;
    }
}

// Launch geometry.  Single rows / small batches: 8 waves x 1 frame per workgroup (5 / 8 of the FFTs useful, two workgroups per CU,
// one round — latency is what counts there).  Batches: runs of consecutive blocks walked by 12-wave workgroups (k_spec_run; round 3
// ran 12 x 2-frame slabs there, 21 / 24 of the FFTs useful and every slab paying its own launch slot, twiddle loads and exposed
// sample loads).  Measured and not kept: 4 x 1 at (1, 160000): 10.2 vs 9.1 us; 12 x 1 at two workgroups per CU on the batch: 84 vs 79 us.
constexpr int RUN_NW = 12;                  // 3 waves per SIMD at <= 170 registers; 16 would need <= 128 and spills
struct SpecGeom { int nw, fpw, runs; };     // runs > 0: k_spec_run<OP, RUN_NW> with that many runs per row
static SpecGeom spec_geom(int T, int rows, int op, bool src_spec) {
#ifdef PAA_EXPERIMENTS
    static const int force = [] { const char* e = getenv("PAA_SPEC_NW"); return e ? atoi(e) : 0; }();
    static const int force_fpw = [] { const char* e = getenv("PAA_SPEC_FPW"); return e ? atoi(e) : 1; }();
    if (force == 4 || force == 8 || force == 12 || force == 16)
        return SpecGeom{force, (force_fpw == 2 && !src_spec && (force == 12 || force == 16)) ? 2 : 1, 0};
#endif
    (void)op;
    const int nblk = T - 1;
    if (rows * cdiv(nblk, 13) < 256) return SpecGeom{8, 1, 0};
    if (src_spec) return SpecGeom{16, 1, 0};
    // runs per row: one workgroup per CU and round; a run of b blocks costs ceil((b + 3) / NW) iterations plus its start-up
    // (twiddles, first samples: about half an iteration).  Pick the count with the least (rounds x run time).
    const int cus = device_cus();
    int best = 1;
    double best_cost = 1e30;
    for (int r = 1; r <= cdiv(nblk, RUN_NW - 3); ++r) {
        const int b = cdiv(nblk, r);
        const double cost = (double)cdiv((int64_t)rows * r, cus) * (cdiv(b + 3, RUN_NW) + 0.5);
        if (cost < best_cost) { best_cost = cost; best = r; }
    }
    return SpecGeom{RUN_NW, 1, best};
}

template <int OP, bool SRC_SPEC, int NW, int FPW>
paa_status launch_fused_nw(const SpecArgs& a, int rows, hipStream_t st) {
    const int nblk = a.T - 1;                                 // output hop-blocks per row
    const size_t lds = sizeof(float2) * NW * XB + (FPW == 2 ? sizeof(float) * NW * N : 0) + 256 +
                       ((OP == SOP_FM && NW * FPW >= 16) ? sizeof(float) * 10 * F : 0);
    if (lds > 64 * 1024) {
        static bool attr = false;
        if (!attr) { PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spec_fused<OP, SRC_SPEC, NW, FPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    }
    hipLaunchKernelGGL((k_spec_fused<OP, SRC_SPEC, NW, FPW>), dim3(cdiv(nblk, NW * FPW - 3), rows), dim3(NW * 64), lds, st, a);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

template <int OP>
paa_status launch_run(const SpecArgs& a, int rows, int runs, hipStream_t st) {
    constexpr int NW = RUN_NW;
    const size_t lds = sizeof(float2) * (NW * XB + 3 * N2) + 256 + (OP == SOP_FM ? sizeof(float) * 10 * FM_LS : 0);
    static bool attr = false;                                 // > 64 KB of dynamic LDS needs the attribute (per code object: set once)
    if (!attr) { PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spec_run<OP, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL((k_spec_run<OP, NW>), dim3(runs, rows), dim3(NW * 64), lds, st, a, cdiv(a.T - 1, runs));
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

template <int OP, bool SRC_SPEC>
paa_status launch_fused(const SpecArgs& a, int rows, hipStream_t st) {
    const SpecGeom gm = spec_geom(a.T, rows, OP, SRC_SPEC);
    if constexpr (!SRC_SPEC && OP != SOP_NONE) {
        if (gm.runs > 0) return launch_run<OP>(a, rows, gm.runs, st);
#ifdef PAA_EXPERIMENTS
        if (gm.fpw == 2) {
            if (gm.nw == 16 && OP != SOP_FM) return launch_fused_nw<OP, SRC_SPEC, 16, 2>(a, rows, st);
            return launch_fused_nw<OP, SRC_SPEC, 12, 2>(a, rows, st);
        }
#endif
    }
    switch (gm.nw) {
        case 16: return launch_fused_nw<OP, SRC_SPEC, 16, 1>(a, rows, st);
#ifdef PAA_EXPERIMENTS
        case 12: return launch_fused_nw<OP, SRC_SPEC, 12, 1>(a, rows, st);
        case 4: return launch_fused_nw<OP, SRC_SPEC, 4, 1>(a, rows, st);
#endif
        default: return launch_fused_nw<OP, SRC_SPEC, 8, 1>(a, rows, st);
    }
}

}  // namespace

int spec_groups(int T, int rows, int op, bool src_spec) {
    const SpecGeom gm = spec_geom(T, rows, op, src_spec);
    return gm.runs > 0 ? gm.runs : cdiv(T - 1, gm.nw * gm.fpw - 3);
}

paa_status spec_project(const SpecArgs& a, int op, int rows, int* n_part, hipStream_t st) {
    if (a.T < 2) PAA_FAIL(PAA_ERR_SIZE, "spec_project: T=%d", a.T);
    if (n_part) *n_part = rows * spec_groups(a.T, rows, op, false);
    switch (op) {
        case SOP_MINMAX: return launch_fused<SOP_MINMAX, false>(a, rows, st);
        case SOP_PHON: return launch_fused<SOP_PHON, false>(a, rows, st);
        case SOP_FM: return launch_fused<SOP_FM, false>(a, rows, st);
        default: PAA_FAIL(PAA_ERR_BAD_NORM, "spec_project: op %d", op);
    }
}

paa_status spec_stft(const SpecArgs& a, int rows, hipStream_t st) {
    hipLaunchKernelGGL(k_spec_stft, dim3(cdiv(a.T, 4), rows), dim3(256), 0, st, a);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

paa_status spec_istft(const SpecArgs& a, int rows, hipStream_t st) {
    if (a.T < 2) PAA_FAIL(PAA_ERR_SIZE, "spec_istft: T=%d", a.T);
    return launch_fused<SOP_NONE, true>(a, rows, st);
}

paa_status spec_apply(const SpecArgs& a, int op, int rows, const float* scale, int* n_part, hipStream_t st) {
    const int64_t n = (int64_t)rows * a.T * F;
    const int grid = (int)std::min<int64_t>(cdiv(n, 256 * 4), 1024);
    if (n_part) *n_part = grid;
    switch (op) {
        case SOP_NONE: hipLaunchKernelGGL(k_spec_apply<SOP_NONE>, dim3(grid), dim3(256), 0, st, a, n, scale); break;
        case SOP_MINMAX: hipLaunchKernelGGL(k_spec_apply<SOP_MINMAX>, dim3(grid), dim3(256), 0, st, a, n, scale); break;
        case SOP_PHON: hipLaunchKernelGGL(k_spec_apply<SOP_PHON>, dim3(grid), dim3(256), 0, st, a, n, scale); break;
        case SOP_FM: hipLaunchKernelGGL(k_spec_apply<SOP_FM>, dim3(grid), dim3(256), 0, st, a, n, scale); break;
        default: PAA_FAIL(PAA_ERR_BAD_NORM, "spec_apply: op %d", op);
    }
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

}  // namespace paa
