// Non-GEMM kernels of the Wav2Vec2 forward/backward for gfx950: layer norm, attention softmax,
// the C_in = 1 first convolution with its GroupNorm / LayerNorm, CTC (alpha/beta in LDS), and the
// reduction of the input gradient over the batch into the universal perturbation's gradient.
//
// Third-party algorithm restated (HuggingFace transformers modeling_wav2vec2.py, torch ATen):
//   Wav2Vec2GroupNormConvLayer / Wav2Vec2LayerNormConvLayer (:275-323), nn.LayerNorm, softmax,
//   F.ctc_loss(reduction='sum', zero_infinity=False) and its backward.
#include <cstdlib>
#include <type_traits>

#include "model_kernels.h"

namespace paa {

// ================================================================================ LayerNorm ===
// One wave per row; rows are short (<= 4 KB) and stay in L1 across the passes.
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ g,
                                              const float* __restrict__ b, float* __restrict__ y,
                                              float* __restrict__ stats, int rows, int cols, float eps, Bf yb, Bf actb,
                                              float* __restrict__ yact) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * cols;
    // the first 1024 columns of the row are read ONCE and kept in registers (4 float4 per lane); wider rows re-read the rest
    float4 xr4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane * 4 + 256 * u;
        xr4[u] = c < cols ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (xr4[u].x + xr4[u].y) + (xr4[u].z + xr4[u].w);       // columns past the row are zeros
    for (int c = 1024 + lane * 4; c < cols; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
    auto sq = [&](const float4& v) {
        const float a = v.x - mean, bb = v.y - mean, cc = v.z - mean, dd = v.w - mean;
        q += (a * a + bb * bb) + (cc * cc + dd * dd);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane * 4 + 256 * u < cols) sq(xr4[u]);
    for (int c = 1024 + lane * 4; c < cols; c += 256) sq(*reinterpret_cast<const float4*>(xr + c));
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0 && stats) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
    auto emit = [&](int c, const float4& v) {
        const float4 gg = *reinterpret_cast<const float4*>(g + c);
        const float4 bv = *reinterpret_cast<const float4*>(b + c);
        float o[4];
        o[0] = (v.x - mean) * rstd * gg.x + bv.x;
        o[1] = (v.y - mean) * rstd * gg.y + bv.y;
        o[2] = (v.z - mean) * rstd * gg.z + bv.z;
        o[3] = (v.w - mean) * rstd * gg.w + bv.w;
        const size_t i0 = (size_t)row * cols + c;
        if (y) *reinterpret_cast<float4*>(y + i0) = make_float4(o[0], o[1], o[2], o[3]);
        store_bf16x4(yb, i0, o);
        if (actb.hi || yact) {
            float ga[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ga[j] = gelu_f(o[j]);
            store_bf16x4(actb, i0, ga);
            if (yact) *reinterpret_cast<float4*>(yact + i0) = make_float4(ga[0], ga[1], ga[2], ga[3]);
        }
    };
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane * 4 + 256 * u < cols) emit(lane * 4 + 256 * u, xr4[u]);
    for (int c = 1024 + lane * 4; c < cols; c += 256) emit(c, *reinterpret_cast<const float4*>(xr + c));
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) [+ add];  if gelu_pre: dy *= gelu'(gelu_pre) first
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                              const float* __restrict__ g, const float* __restrict__ stats,
                                              const float* __restrict__ add, const float* __restrict__ gelu_pre,
                                              float* __restrict__ dx, int rows, int cols, Bf dxb) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t o = (size_t)row * cols;
    const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
    // dy (with the optional gelu' factor applied) and x of the first 1024 columns stay in registers between the two passes
    float dr[4][4], xh[4][4];
    float s1 = 0.f, s2 = 0.f;
    auto chunk = [&](int c, float (&d)[4], float (&xv)[4]) {          // loads one chunk: d = dy [* gelu'], xv = xhat
        const float4 d4 = *reinterpret_cast<const float4*>(dy + o + c);
        const float4 x4 = *reinterpret_cast<const float4*>(x + o + c);
        d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
        if (gelu_pre) {
            const float4 p4 = *reinterpret_cast<const float4*>(gelu_pre + o + c);
            d[0] *= gelu_grad_f(p4.x); d[1] *= gelu_grad_f(p4.y); d[2] *= gelu_grad_f(p4.z); d[3] *= gelu_grad_f(p4.w);
        }
        xv[0] = (x4.x - mean) * rstd; xv[1] = (x4.y - mean) * rstd; xv[2] = (x4.z - mean) * rstd; xv[3] = (x4.w - mean) * rstd;
    };
    auto sums = [&](int c, const float (&d)[4], const float (&xv)[4]) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + c);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gd = gv[j] * d[j];
            s1 += gd;
            s2 += gd * xv[j];
        }
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane * 4 + 256 * u;
        if (c < cols) { chunk(c, dr[u], xh[u]); sums(c, dr[u], xh[u]); }
    }
    for (int c = 1024 + lane * 4; c < cols; c += 256) {
        float d[4], xv[4];
        chunk(c, d, xv);
        sums(c, d, xv);
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    auto emit = [&](int c, const float (&d)[4], const float (&xv)[4]) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + c);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rstd * (gv[j] * d[j] - s1 - xv[j] * s2);
        if (add) {
            const float4 a4 = *reinterpret_cast<const float4*>(add + o + c);
            v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
        }
        if (dx) *reinterpret_cast<float4*>(dx + o + c) = make_float4(v[0], v[1], v[2], v[3]);
        store_bf16x4(dxb, o + c, v);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane * 4 + 256 * u;
        if (c < cols) emit(c, dr[u], xh[u]);
    }
    for (int c = 1024 + lane * 4; c < cols; c += 256) {
        float d[4], xv[4];
        chunk(c, d, xv);
        emit(c, d, xv);
    }
}

paa_status layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* stats, int rows, int cols,
                         float eps, Bf yb, Bf actb, float* yact, hipStream_t st) {
    if (cols & 3) PAA_FAIL(PAA_ERR_ARG, "layernorm: cols=%d must be a multiple of 4", cols);
    hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, g, b, y, stats, rows, cols, eps, yb, actb, yact);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

paa_status layernorm_bwd(const float* dy, const float* x, const float* g, const float* stats, const float* add,
                         const float* gelu_pre, float* dx, Bf dxb, int rows, int cols, hipStream_t st) {
    if (cols & 3) PAA_FAIL(PAA_ERR_ARG, "layernorm backward: cols=%d must be a multiple of 4", cols);
    hipLaunchKernelGGL(k_ln_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, st, dy, x, g, stats, add, gelu_pre, dx, rows, cols, dxb);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

// ================================================================================= softmax ===
// In place on rows of `cols` valid entries with leading dimension ld; pad columns are zeroed.
__global__ __launch_bounds__(256) void k_softmax_fwd(float* __restrict__ s, int rows, int cols, int ld, float scale,
                                                   int rows_per_mat, int mat_rows_ld) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float* p = s + ((size_t)(r / rows_per_mat) * mat_rows_ld + (r % rows_per_mat)) * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(p[c] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < ld; c += 64) p[c] = (c < cols) ? __expf(p[c] * scale - mx) * inv : 0.f;
}

// dS = scale * P * (dP - sum_j dP_j P_j), in place on dP.
__global__ __launch_bounds__(256) void k_softmax_bwd(float* __restrict__ dp, const float* __restrict__ pm, int rows,
                                                   int cols, int ld, float scale, int rows_per_mat, int mat_rows_ld) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const size_t o = ((size_t)(r / rows_per_mat) * mat_rows_ld + (r % rows_per_mat)) * ld;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += dp[o + c] * pm[o + c];
    dot = wave_sum(dot);
    for (int c = lane; c < ld; c += 64) dp[o + c] = (c < cols) ? scale * pm[o + c] * (dp[o + c] - dot) : 0.f;
}

paa_status softmax_fwd(float* s, int n_mat, int rows_per_mat, int mat_rows_ld, int cols, int ld, float scale,
                       hipStream_t st) {
    const int rows = n_mat * rows_per_mat;
    hipLaunchKernelGGL(k_softmax_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, st, s, rows, cols, ld, scale, rows_per_mat,
                       mat_rows_ld);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

paa_status softmax_bwd(float* dp, const float* p, int n_mat, int rows_per_mat, int mat_rows_ld, int cols, int ld,
                       float scale, hipStream_t st) {
    const int rows = n_mat * rows_per_mat;
    hipLaunchKernelGGL(k_softmax_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, st, dp, p, rows, cols, ld, scale, rows_per_mat,
                       mat_rows_ld);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

// ============================================================================ conv0 (C_in=1) ===
// in_sample(a, b, i): input sample i of clip b (model_kernels.h)

constexpr int C0_TCH = 128;  // frames per workgroup in the channel-per-thread kernels

// Channel-per-thread mapping (thread owns channels tid, tid+256, ...; loops over a chunk of frames).
// MODE 1: apply GroupNorm + GELU                                    -> pre[b][t][c], act[b][t][c]
// MODE 2: backward statistics partials (sum dy, sum dy*xhat)        -> part[b][chunk][c][2]
template <int MODE>
__global__ __launch_bounds__(256) void k_conv0_gn(Conv0Args a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * C0_TCH;
    const int nt = min(C0_TCH, a.T - t0);
    const int span = (nt - 1) * a.stride + a.k;
    const int span_alloc = (C0_TCH - 1) * a.stride + 10 + 8;
    for (int i = threadIdx.x; i < span_alloc; i += 256) xs[i] = (i < span) ? in_sample(a, b, t0 * a.stride + i) : 0.f;
    __syncthreads();
    // wav2vec2's conv0 (k = 10, stride 5, even C): a thread owns a channel PAIR (4-byte bf16x2 accesses instead of
    // 2-byte ones) and walks the frames 4 at a time — the 4 windows come from 7 broadcast ds_read_b128 and, in the
    // backward pass, the 4 gradient loads are issued together so that enough bytes are in flight to cover HBM latency.
    if (a.stride == 5 && a.k == 10 && (a.C & 1) == 0) {
        for (int c = 2 * threadIdx.x; c < a.C; c += 512) {
            // the channel pair rides in 2-wide vectors: v_pk_fma_f32 does both channels' multiply-adds at once
            f32x2 w2[10], mean2, rstd2, gam2, bet2;
            f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 10; ++j) w2[j] = f32x2{a.w[c * 10 + j], a.w[(c + 1) * 10 + j]};
            mean2 = f32x2{a.gn_stats[((size_t)b * a.C + c) * 2], a.gn_stats[((size_t)b * a.C + c + 1) * 2]};
            rstd2 = f32x2{a.gn_stats[((size_t)b * a.C + c) * 2 + 1], a.gn_stats[((size_t)b * a.C + c + 1) * 2 + 1]};
            gam2 = f32x2{a.gamma[c], a.gamma[c + 1]};
            bet2 = f32x2{a.beta[c], a.beta[c + 1]};
            for (int t = 0; t < nt; t += 4) {
                unsigned dh[4] = {0u, 0u, 0u, 0u}, dl4[4] = {0u, 0u, 0u, 0u};
                if (MODE == 2) {
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        if (t + f < nt) {
                            const size_t o = ((size_t)b * a.P + t0 + t + f) * a.C + c;
                            dh[f] = *reinterpret_cast<const unsigned*>(a.dpreb.hi + o);
                            if (a.dpreb.lo) dl4[f] = *reinterpret_cast<const unsigned*>(a.dpreb.lo + o);
                        }
                }
                float xw[28];
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const float4 x4 = *reinterpret_cast<const float4*>(xs + t * 5 + 4 * q);
                    xw[4 * q] = x4.x; xw[4 * q + 1] = x4.y; xw[4 * q + 2] = x4.z; xw[4 * q + 3] = x4.w;
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (t + f >= nt) continue;
                    const size_t o = ((size_t)b * a.P + t0 + t + f) * a.C + c;
                    f32x2 v = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 10; ++j) v += w2[j] * xw[5 * f + j];
                    if (MODE == 1) {
                        f32x2 y = (v - mean2) * rstd2 * gam2 + bet2, g;
                        if (a.pre16) {                       // bf16 mode: keep gelu'(y) as bf16
                            f32x2 dg;
                            g = gelu_both_fast2(y, dg);
                            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(a.pre) + o) = bf16_bits(dg.x) | ((unsigned)bf16_bits(dg.y) << 16);
                        } else if (a.gate) {                 // fp32-parity mode: keep gelu'(y) as f32; the branch-free erf (|error| <= 1.5e-7)
                            f32x2 dg;                        // of every other GELU of this mode (gemm_dev.h, epilogue_vec): with libm's erff for
                            g = gelu_both_fast2(y, dg);      // both the GELU and its derivative this HBM-sized kernel turns VALU-bound (+1 % of the step)
                            *reinterpret_cast<float2*>(a.pre + o) = make_float2(dg.x, dg.y);
                        } else {
                            g = f32x2{gelu_f(y.x), gelu_f(y.y)};
                            *reinterpret_cast<float2*>(a.pre + o) = make_float2(y.x, y.y);
                        }
                        const unsigned short h0 = bf16_bits(g.x), h1 = bf16_bits(g.y);
                        if (a.actb.il) {     // (o is even: the channel pair stays inside one 32-element group, paa_common.h Bf::il)
                            unsigned short* ph = a.actb.hi + il_index(o);
                            *reinterpret_cast<unsigned*>(ph) = h0 | ((unsigned)h1 << 16);
                            *reinterpret_cast<unsigned*>(ph + 32) = bf16_bits(g.x - bf16_to_f32(h0)) | ((unsigned)bf16_bits(g.y - bf16_to_f32(h1)) << 16);
                        } else {
                            *reinterpret_cast<unsigned*>(a.actb.hi + o) = h0 | ((unsigned)h1 << 16);
                            if (a.actb.lo)
                                *reinterpret_cast<unsigned*>(a.actb.lo + o) = bf16_bits(g.x - bf16_to_f32(h0)) | ((unsigned)bf16_bits(g.y - bf16_to_f32(h1)) << 16);
                        }
                    } else {
                        const f32x2 dy = {__uint_as_float(dh[f] << 16) + __uint_as_float(dl4[f] << 16),
                                          __uint_as_float(dh[f] & 0xFFFF0000u) + __uint_as_float(dl4[f] & 0xFFFF0000u)};
                        s1 += dy;
                        s2 += dy * ((v - mean2) * rstd2);
                    }
                }
            }
            if (MODE != 1) {
                const size_t o = (((size_t)b * gridDim.x + chunk) * a.C + c) * 2;
                *reinterpret_cast<float4*>(a.part + o) = make_float4(s1.x, s2.x, s1.y, s2.y);
            }
        }
    } else {
        for (int c = threadIdx.x; c < a.C; c += 256) {
            float w[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) w[j] = (j < a.k) ? a.w[c * a.k + j] : 0.f;
            const float mean = a.gn_stats[((size_t)b * a.C + c) * 2], rstd = a.gn_stats[((size_t)b * a.C + c) * 2 + 1];
            const float gam = a.gamma[c], bet = a.beta[c];
            float s1 = 0.f, s2 = 0.f;
            for (int t = 0; t < nt; ++t) {
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 10; ++j) v += w[j] * xs[t * a.stride + j];
                const size_t o = ((size_t)b * a.P + t0 + t) * a.C + c;
                if (MODE == 1) {
                    const float y = (v - mean) * rstd * gam + bet;
                    if (a.pre16) {
                        float dg;
                        const float gy = gelu_both_fast(y, dg);
                        reinterpret_cast<unsigned short*>(a.pre)[o] = bf16_bits(dg);
                        store_bf16(a.actb, o, gy);
                    } else if (a.gate) {
                        float dg;
                        const float gy = gelu_both_fast(y, dg);
                        a.pre[o] = dg;
                        store_bf16(a.actb, o, gy);
                    } else {
                        a.pre[o] = y;
                        store_bf16(a.actb, o, gelu_f(y));
                    }
                } else {
                    float dy = bf16_to_f32(a.dpreb.hi[o]);
                    if (a.dpreb.lo) dy += bf16_to_f32(a.dpreb.lo[o]);
                    s1 += dy; s2 += dy * ((v - mean) * rstd);
                }
            }
            if (MODE != 1) {
                const size_t o = (((size_t)b * gridDim.x + chunk) * a.C + c) * 2;
                a.part[o] = s1; a.part[o + 1] = s2;
            }
        }
    }
    // pad rows [T, P) of this clip are zero
    if (MODE == 1 && chunk == gridDim.x - 1)
        for (int t = a.T; t < a.P; ++t)
            for (int c = threadIdx.x; c < a.C; c += 256) {
                const size_t o = ((size_t)b * a.P + t) * a.C + c;
                if (a.pre16) reinterpret_cast<unsigned short*>(a.pre)[o] = 0; else a.pre[o] = 0.f;
                store_bf16(a.actb, o, 0.f);
            }
}

// GroupNorm statistics of conv0 WITHOUT a pass over the (T, C) output.  With one input channel the output of channel
// c at frame t is w_c . x_t (x_t = the k input samples of the frame), so over a clip
//   sum_t v = w_c . Sx,   sum_t v^2 = w_c^T Gx w_c,   Sx = sum_t x_t,  Gx = sum_t x_t x_t^T   (k x k, per clip)
// k_conv0_gram accumulates the 55 + 10 distinct entries per chunk of frames (f32 products of <= 8 frames per thread,
// summed in f64, fixed order => reproducible); k_conv0_gn_stats reduces the chunks and evaluates the two quadratic
// forms per channel in f64.
constexpr int C0_GCH = 2048;       // frames per workgroup of the Gram kernel
constexpr int C0_GQ = 66;          // 55 lower-triangle products + 10 sums (+1 pad)
__global__ __launch_bounds__(256) void k_conv0_gram(Conv0Args a, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ double red[256 / 64][C0_GQ];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * C0_GCH;
    const int nt = min(C0_GCH, a.T - t0);
    const int span = (nt - 1) * a.stride + a.k;
    for (int i = threadIdx.x; i < span; i += 256) xs[i] = in_sample(a, b, t0 * a.stride + i);
    __syncthreads();
    float acc[65];
#pragma unroll
    for (int q = 0; q < 65; ++q) acc[q] = 0.f;
    for (int t = threadIdx.x; t < nt; t += 256) {
        float xw[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) xw[j] = j < a.k ? xs[t * a.stride + j] : 0.f;
        int idx = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int q = 0; q <= j; ++q) acc[idx++] += xw[j] * xw[q];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[55 + j] += xw[j];
    }
    // 65 block sums: the wave sums first, ONE exchange through LDS, then the four waves' values in block_sum's own order (65 block_sum
    // calls were 130 barriers in a row: 51 us for a kernel that moves 20 MB)
    double* out = part + ((size_t)b * gridDim.x + chunk) * C0_GQ;
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 65; ++q) {
        const double t = wave_sum((double)acc[q]);
        if ((threadIdx.x & 63) == 0) red[wv][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < 65) {
        double r = 0;
#pragma unroll
        for (int i = 0; i < 256 / 64; ++i) r += red[i][threadIdx.x];
        out[threadIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void k_conv0_gn_stats(Conv0Args a, const double* __restrict__ part, int nchunk) {
    __shared__ double G[C0_GQ];
    const int b = blockIdx.x;
    if (threadIdx.x < 65) {
        double t = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) t += part[((size_t)b * nchunk + ch) * C0_GQ + threadIdx.x];
        G[threadIdx.x] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += 256) {
        double w[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) w[j] = j < a.k ? (double)a.w[c * a.k + j] : 0.0;
        double s1 = 0.0, s2 = 0.0;
        int idx = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
#pragma unroll
            for (int q = 0; q <= j; ++q) { s2 += (q == j ? 1.0 : 2.0) * w[j] * w[q] * G[idx]; ++idx; }
            s1 += w[j] * G[55 + j];
        }
        const double mean = s1 / a.T;
        double var = s2 / a.T - mean * mean;
        if (var < 0.0) var = 0.0;
        a.gn_stats[((size_t)b * a.C + c) * 2] = (float)mean;
        a.gn_stats[((size_t)b * a.C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

// Reduce the per-chunk partials of the GroupNorm backward sums in f64: -> (s1 / n, s2 / n) per (clip, channel)
__global__ __launch_bounds__(256) void k_conv0_gn_finalize(const float* __restrict__ part, float* __restrict__ out,
                                                          int B, int C, int nchunk, int n) {
    // 64 (b, c) pairs per block, 4 chunk-slices each (fixed summation order => reproducible)
    __shared__ double sh1[4][64], sh2[4][64];
    const int pair = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + pair;
    double s1 = 0.0, s2 = 0.0;
    if (i < B * C) {
        const int b = i / C, c = i % C;
        for (int k = slice; k < nchunk; k += 4) {
            const size_t o = (((size_t)b * nchunk + k) * C + c) * 2;
            s1 += (double)part[o]; s2 += (double)part[o + 1];
        }
    }
    sh1[slice][pair] = s1; sh2[slice][pair] = s2;
    __syncthreads();
    if (slice != 0 || i >= B * C) return;
    s1 = (sh1[0][pair] + sh1[1][pair]) + (sh1[2][pair] + sh1[3][pair]);
    s2 = (sh2[0][pair] + sh2[1][pair]) + (sh2[2][pair] + sh2[3][pair]);
    out[2 * (size_t)i] = (float)(s1 / n);
    out[2 * (size_t)i + 1] = (float)(s2 / n);
}

// Wave-per-frame mapping: lane owns channels lane, lane+64, ... (C <= 512).
// FWD_LN  : conv + bias -> LayerNorm over channels -> pre, GELU -> act, row stats (mean, rstd)
// BWD     : dv (through GroupNorm with the precomputed batch sums, or through LayerNorm with wave sums),
//           then G[b][t][j] = sum_c w[c][j] * dv[c]   (the transposed C_in = 1 convolution, per tap)
enum { C0_FWD_LN = 0, C0_BWD_GN = 1, C0_BWD_LN = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void k_conv0_rows(Conv0Args a) {
    constexpr int MAXC = 8;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const int b = blockIdx.y;
    const int nc = (a.C + 63) / 64;
    float w[MAXC][10];
    float gam[MAXC], bet[MAXC], bias[MAXC], mean_c[MAXC], rstd_c[MAXC], bs1[MAXC], bs2[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        const bool ok = (i < nc) && (c < a.C);
#pragma unroll
        for (int j = 0; j < 10; ++j) w[i][j] = (ok && j < a.k) ? a.w[c * a.k + j] : 0.f;
        gam[i] = ok ? a.gamma[c] : 0.f;
        bet[i] = ok ? a.beta[c] : 0.f;
        bias[i] = (ok && a.bias && MODE != C0_BWD_GN) ? a.bias[c] : 0.f;   // GroupNorm cancels a per-channel bias
        if (MODE == C0_BWD_GN) {
            mean_c[i] = ok ? a.gn_stats[((size_t)b * a.C + c) * 2] : 0.f;
            rstd_c[i] = ok ? a.gn_stats[((size_t)b * a.C + c) * 2 + 1] : 0.f;
            bs1[i] = ok ? a.gn_bsums[((size_t)b * a.C + c) * 2] : 0.f;
            bs2[i] = ok ? a.gn_bsums[((size_t)b * a.C + c) * 2 + 1] : 0.f;
        }
    }
    // LayerNorm backward (large-lv60's feature extractor): a row is a dependent chain — samples and dy from memory, convolution, two
    // wave sums, ten more for the taps — and a wave walked its rows one by one: the kernel ran at a fifth of its instruction rate,
    // waiting on HBM once per row.  The next row's samples, dy and statistics are requested before the current row is worked on.
    float xin_n[10], dp_n[MAXC], st_n[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 10; ++j) xin_n[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) dp_n[i] = 0.f;
    auto prefetch = [&](int t) {
        if (MODE != C0_BWD_LN || t >= a.T) return;      // (the forward pass, bound by its stores, measured 2.2 instead of 1.64 ms with it)
        const size_t row = (size_t)b * a.P + t;
#pragma unroll
        for (int j = 0; j < 10; ++j) xin_n[j] = (j < a.k) ? in_sample(a, b, t * a.stride + j) : 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            dp_n[i] = (i < nc && c < a.C) ? a.dpre[row * a.C + c] : 0.f;
        }
        st_n[0] = a.row_stats[2 * row]; st_n[1] = a.row_stats[2 * row + 1];
    };
    prefetch(wave_global);
    for (int t = wave_global; t < a.P; t += n_waves) {
        const size_t row = (size_t)b * a.P + t;
        float xin_c[10], dp_c[MAXC], st_c[2] = {st_n[0], st_n[1]};
#pragma unroll
        for (int j = 0; j < 10; ++j) xin_c[j] = xin_n[j];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) dp_c[i] = dp_n[i];
        if (MODE == C0_BWD_LN) prefetch(t + n_waves);
        if (t >= a.T) {                       // pad rows
            if (MODE == C0_FWD_LN) {
                for (int i = 0; i < nc; ++i) { const int c = lane + 64 * i; if (c < a.C) { a.pre[row * a.C + c] = 0.f; store_bf16(a.actb, row * a.C + c, 0.f); } }
                if (lane == 0) { a.row_stats[2 * row] = 0.f; a.row_stats[2 * row + 1] = 0.f; }
            } else if (lane < a.k) a.G[row * a.k + lane] = 0.f;
            continue;
        }
        float xin[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) xin[j] = MODE == C0_BWD_LN ? xin_c[j] : ((j < a.k) ? in_sample(a, b, t * a.stride + j) : 0.f);
        float v[MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            float acc = bias[i];
#pragma unroll
            for (int j = 0; j < 10; ++j) acc += w[i][j] * xin[j];
            v[i] = acc;
            if (i < nc && lane + 64 * i < a.C) s += acc;
        }
        float dv[MAXC];
        if (MODE == C0_FWD_LN || MODE == C0_BWD_LN) {
            float mean, rstd;
            if (MODE == C0_FWD_LN) {
                mean = wave_sum(s) / (float)a.C;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < MAXC; ++i) if (i < nc && lane + 64 * i < a.C) q += (v[i] - mean) * (v[i] - mean);
                rstd = rsqrtf(wave_sum(q) / (float)a.C + a.eps);
                if (lane == 0) { a.row_stats[2 * row] = mean; a.row_stats[2 * row + 1] = rstd; }
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = lane + 64 * i;
                    if (i < nc && c < a.C) {
                        const float y = (v[i] - mean) * rstd * gam[i] + bet[i];
                        a.pre[row * a.C + c] = y;
                        store_bf16(a.actb, row * a.C + c, gelu_f(y));
                    }
                }
                continue;
            }
            mean = st_c[0]; rstd = st_c[1];
            float s1 = 0.f, s2 = 0.f;
            float gd[MAXC], xh[MAXC];
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                const bool ok = i < nc && c < a.C;
                gd[i] = ok ? gam[i] * dp_c[i] : 0.f;
                xh[i] = ok ? (v[i] - mean) * rstd : 0.f;
                s1 += gd[i]; s2 += gd[i] * xh[i];
            }
            s1 = wave_sum(s1) / (float)a.C;
            s2 = wave_sum(s2) / (float)a.C;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) dv[i] = (i < nc && lane + 64 * i < a.C) ? rstd * (gd[i] - s1 - xh[i] * s2) : 0.f;
        } else {   // C0_BWD_GN: dv = gamma * rstd * (dy - mean_t(dy) - xhat * mean_t(dy*xhat))
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                const bool ok = i < nc && c < a.C;
                const float dy = ok ? a.dpre[row * a.C + c] : 0.f;
                const float xh = (v[i] - mean_c[i]) * rstd_c[i];
                dv[i] = ok ? gam[i] * rstd_c[i] * (dy - bs1[i] - xh * bs2[i]) : 0.f;
            }
        }
        float gj[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) acc += w[i][j] * dv[i];
            gj[j] = wave_sum(acc);
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) if (lane == j && j < a.k) a.G[row * a.k + j] = gj[j];
    }
}

// grad[l] = sum_b mask_b[l] * sum_{(t, j): t*stride + j = l} G[b][t][j]    (fixed summation order over b)
__global__ void k_input_grad(Conv0Args a, float* __restrict__ grad) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.L) return;
    float total = 0.f;
    const float pv = a.p ? a.p[l] : 0.f;
    for (int b = 0; b < a.B; ++b) {
        float gsum = 0.f;
        // taps j = l - t*stride in [0, k)  =>  t in [ceil((l-k+1)/stride), floor(l/stride)]
        int t_hi = l / a.stride;
        int t_lo = (l - a.k + 1 + a.stride - 1);
        t_lo = t_lo <= 0 ? 0 : t_lo / a.stride;
        if (t_hi > a.T - 1) t_hi = a.T - 1;
        for (int t = t_lo; t <= t_hi; ++t) gsum += a.G[((size_t)b * a.P + t) * a.k + (l - t * a.stride)];
        if (a.p && a.clamp) {
            const float u = a.clean[(size_t)b * a.L + l] + pv;
            if (!(u >= -1.f && u <= 1.f)) gsum = 0.f;           // clamp backward: pass-through inside [-1, 1]
        }
        total += gsum;
    }
    grad[l] = total;
}

// GroupNorm backward of conv0 without touching the (B, T, C) gradient row by row:
//   G[b,t,j] = sum_c w[c,j] dv[b,t,c],  dv = gamma rstd (dy - s1 - xhat s2),  xhat = (conv(x)[t,c] - mean) rstd
//            = (dy[b,t,:] . W1_b[:,j]) + kc_b[j] - sum_j' Mx_b[j,j'] x[b, t*stride + j']
// with W1_b[c,j] = w[c,j] gamma_c rstd_bc.  The first term is a GEMM over the bf16 planes of dy (per-clip B
// operand); this kernel builds W1_b (bf16 planes, [16][C] per clip), Mx_b and kc_b.  One workgroup per clip.
// mode 0: everything; 1: W1_b only (needs the forward statistics alone); 2: Mx_b and kc_b only (needs gn_bsums).
__global__ __launch_bounds__(256) void k_conv0_bwd_prep(Conv0Args a, int mode) {
    __shared__ double red[256 / 64][110];
    const int b = blockIdx.x;
    float acc[10][11];
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int q = 0; q < 11; ++q) acc[j][q] = 0.f;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        const float mean = a.gn_stats[((size_t)b * a.C + c) * 2], rstd = a.gn_stats[((size_t)b * a.C + c) * 2 + 1];
        float s1 = 0.f, s2 = 0.f;
        if (mode != 1) { s1 = a.gn_bsums[((size_t)b * a.C + c) * 2]; s2 = a.gn_bsums[((size_t)b * a.C + c) * 2 + 1]; }
        const float gr = a.gamma[c] * rstd;
        float w[10], w1[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            w[j] = j < a.k ? a.w[c * a.k + j] : 0.f;
            const float v = w[j] * gr;
            const size_t o = ((size_t)b * 16 + j) * a.C + c;
            const unsigned short h = bf16_bits(v);
            if (mode != 2) a.w1b.hi[o] = h;
            float vr = bf16_to_f32(h);
            if (a.w1b.lo) { const unsigned short l = bf16_bits(v - vr); if (mode != 2) a.w1b.lo[o] = l; vr += bf16_to_f32(l); }
            w1[j] = vr;                                   // the value the GEMM will actually multiply with
        }
        if (mode != 2)
            for (int j = 10; j < 16; ++j) { const size_t o = ((size_t)b * 16 + j) * a.C + c; a.w1b.hi[o] = 0; if (a.w1b.lo) a.w1b.lo[o] = 0; }
        const float sr = s2 * rstd;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
#pragma unroll
            for (int q = 0; q < 10; ++q) acc[j][q] += w1[j] * sr * w[q];
            acc[j][10] += w1[j] * (sr * mean - s1);
        }
    }
    if (mode == 1) return;
    // 110 sums over the block: wave sums in f64, one LDS exchange, fixed order over the four waves
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const double t = wave_sum((double)acc[j][q]);
            if ((threadIdx.x & 63) == 0) red[wv][j * 11 + q] = t;
        }
    __syncthreads();
    if (threadIdx.x < 110) {
        const int j = threadIdx.x / 11, q = threadIdx.x - 11 * j;
        const double t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (j < a.k) {
            if (q < 10) { if (q < a.k) a.Mx[((size_t)b * a.k + j) * a.k + q] = (float)t; }
            else a.kc[(size_t)b * 16 + j] = (float)t;
        }
    }
}

// grad[l] = sum_b mask_b[l] * sum_{(t, j): t*stride + j = l} G[b][t][j], G from the GEMM result G1 (see above).
// A workgroup owns 64 consecutive samples; its four waves walk the clips b = wave, wave + 4, ... (each step is two
// dependent global round trips — input window / Mx / kc into LDS, then G1 — so four clips in flight per workgroup and
// four times as many workgroups hide what a single walk over all clips exposed).  Per clip the input window, Mx_b and
// kc_b are staged in LDS once instead of being recomputed / re-read per (frame, tap).  Fixed summation order.
constexpr int IG_S = 64;
__global__ __launch_bounds__(256) void k_input_grad_gn(Conv0Args a, float* __restrict__ grad) {
    __shared__ float xs[4][IG_S + 2 * 10 + 4];
    __shared__ float smx[4][10 * 10], skc[4][16];
    __shared__ float part[4][IG_S];
    const int ls = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int l0 = blockIdx.x * IG_S, l = l0 + ls;
    const int lc = l < a.L ? l : a.L - 1;
    int t_hi = lc / a.stride;
    int t_lo = (lc - a.k + 1 + a.stride - 1);
    t_lo = t_lo <= 0 ? 0 : t_lo / a.stride;
    if (t_hi > a.T - 1) t_hi = a.T - 1;
    const int w0 = l0 - a.k;                                  // first sample of the staged window
    float total = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += 4) {
        const int b = b0 + cg;
        const bool on = b < a.B;
        __syncthreads();                                      // everyone is done with the previous clips' tiles
        if (on) {
            for (int i = ls; i < IG_S + 2 * a.k; i += 64) {
                const int sidx = w0 + i;
                xs[cg][i] = (sidx >= 0 && sidx < a.L) ? in_sample(a, b, sidx) : 0.f;
            }
            for (int i = ls; i < a.k * a.k; i += 64) smx[cg][i] = a.Mx[(size_t)b * a.k * a.k + i];
            if (ls < 16) skc[cg][ls] = a.kc[(size_t)b * 16 + ls];
        }
        __syncthreads();
        if (!on) continue;
        float gsum = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int j = lc - t * a.stride;
            float g = a.G1[((size_t)b * a.P + t) * 16 + j] + skc[cg][j];
            const float* mx = smx[cg] + j * a.k;
            const float* xw = xs[cg] + (t * a.stride - w0);
            for (int q = 0; q < a.k; ++q) g -= mx[q] * xw[q];
            gsum += g;
        }
        if (a.p && a.clamp) {
            const float u = a.clean[(size_t)b * a.L + lc] + a.p[lc];
            if (!(u >= -1.f && u <= 1.f)) gsum = 0.f;           // clamp backward: pass-through inside [-1, 1]
        }
        total += gsum;
    }
    part[cg][ls] = total;
    __syncthreads();
    if (cg == 0 && l < a.L) grad[l] = (part[0][ls] + part[1][ls]) + (part[2][ls] + part[3][ls]);
}

paa_status conv0_gn_forward(const Conv0Args& a, float* part, hipStream_t st) {
    if (a.k > 10 || a.C > 4096) PAA_FAIL(PAA_ERR_ARG, "conv0: kernel %d / channels %d unsupported", a.k, a.C);
    const int nchunk = cdiv(a.T, C0_TCH);
    const size_t lds = sizeof(float) * ((C0_TCH - 1) * a.stride + 10 + 8);
    Conv0Args b = a;
    b.part = part;
    {   // statistics from the k x k input Gram matrix of each clip (no pass over the conv output)
        const int ng = cdiv(a.T, C0_GCH);
        double* gp = reinterpret_cast<double*>(part);
        hipLaunchKernelGGL(k_conv0_gram, dim3(ng, a.B), dim3(256), sizeof(float) * ((C0_GCH - 1) * a.stride + 10), st, b, gp);
        PAA_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_conv0_gn_stats, dim3(a.B), dim3(256), 0, st, b, (const double*)gp, ng);
        PAA_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_conv0_gn<1>, dim3(nchunk, a.B), dim3(256), lds, st, b);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

paa_status conv0_ln_forward(const Conv0Args& a, hipStream_t st) {
    if (a.k > 10 || a.C > 512) PAA_FAIL(PAA_ERR_ARG, "conv0(layer): kernel %d / channels %d unsupported", a.k, a.C);
    hipLaunchKernelGGL(k_conv0_rows<C0_FWD_LN>, dim3(std::min(cdiv(a.P, 4), 512), a.B), dim3(256), 0, st, a);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

static bool g_conv0_two_pass = false;
void set_conv0_two_pass(bool on) { g_conv0_two_pass = on; }

paa_status conv0_backward(const Conv0Args& a, int layer_norm, int precision, float* part, float* grad, hipStream_t st) {
    if (a.k > 10 || a.C > 512) PAA_FAIL(PAA_ERR_ARG, "conv0 backward: kernel %d / channels %d unsupported", a.k, a.C);
    if (layer_norm) {
        hipLaunchKernelGGL(k_conv0_rows<C0_BWD_LN>, dim3(std::min(cdiv(a.P, 4), 512), a.B), dim3(256), 0, st, a);
        PAA_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_input_grad, dim3(cdiv(a.L, 256)), dim3(256), 0, st, a, grad);
        PAA_LAUNCH_CHECK();
        return PAA_OK;
    }
    // g_conv0_two_pass (paa_test_option(0, 1), tests only): take the statistics pass + GEMM pass — the path of the shapes the fused
    // kernel does not cover — on every shape, so that the two can be compared on the same operands
    if (conv0_dgrad_supported(a) && !g_conv0_two_pass) {
        // W1_b first (forward statistics only), then ONE pass over dy for G1 and the GroupNorm sums (both precision modes)
        hipLaunchKernelGGL(k_conv0_bwd_prep, dim3(a.B), dim3(256), 0, st, a, 1);
        PAA_LAUNCH_CHECK();
        PAA_TRY(conv0_dgrad_fused(a, part, st));
        hipLaunchKernelGGL(k_conv0_bwd_prep, dim3(a.B), dim3(256), 0, st, a, 2);
        PAA_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_input_grad_gn, dim3(cdiv(a.L, IG_S)), dim3(256), 0, st, a, grad);
        PAA_LAUNCH_CHECK();
        return PAA_OK;
    }
    const int nchunk = cdiv(a.T, C0_TCH);
    const size_t lds = sizeof(float) * ((C0_TCH - 1) * a.stride + 10 + 8);
    Conv0Args b = a;
    b.part = part;
    hipLaunchKernelGGL(k_conv0_gn<2>, dim3(nchunk, a.B), dim3(256), lds, st, b);     // s1 = mean_t dy, s2 = mean_t dy*xhat
    PAA_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_conv0_gn_finalize, dim3(cdiv(a.B * a.C, 64)), dim3(256), 0, st, (const float*)part,
                       (float*)a.gn_bsums, a.B, a.C, nchunk, a.T);
    PAA_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_conv0_bwd_prep, dim3(a.B), dim3(256), 0, st, a, 0);
    PAA_LAUNCH_CHECK();
    paa_gemm_desc d{};                    // G1[b] (P x 16) = dy[b] (P x C) . W1_b^T
    d.operand_bf16 = 1; d.precision = precision;
    d.A = reinterpret_cast<const float*>(a.dpreb.hi); d.A_lo = a.dpreb.lo;
    d.B = reinterpret_cast<const float*>(a.w1b.hi); d.B_lo = a.w1b.lo;
    d.C = a.G1;
    d.M = a.P; d.N = 16; d.K = a.C; d.lda = a.C; d.ldb = a.C; d.ldc = 16;
    d.a_kcontig = 1; d.b_kcontig = 1; d.alpha = 1.f;
    d.batch = a.B; d.batch2 = 1; d.a_s1 = (int64_t)a.P * a.C; d.b_s1 = (int64_t)16 * a.C; d.c_s1 = (int64_t)a.P * 16;
    PAA_TRY(gemm(d, st));
    hipLaunchKernelGGL(k_input_grad_gn, dim3(cdiv(a.L, IG_S)), dim3(256), 0, st, a, grad);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

// ===================================================================================== CTC ===
// One workgroup per clip.  Phase 1: log-softmax of every frame (32 lanes per frame) -> lp[T][V].
// Phase 2: alpha over the extended label sequence l' (blank, l1, blank, ..., blank), previous row in
// LDS, every row stored for phase 3.  Phase 3: beta backwards; the posterior occupancy
// gamma_t(s) = exp(alpha + beta + nll - lp) is summed per class with fixed-point LDS atomics (order
// independent => bitwise reproducible) and dlogits[t][c] = scale * (softmax[t][c] - occ[c]).
// The recursions ACCUMULATE in float64: alpha + beta + nll cancels numbers of magnitude ~1e3, which in float32
// (as torch's CPU kernel computes it) leaves ~1e-3 relative noise in the gradient.  Only the bounded correction
// log(sum exp(x_i - max)) in [0, log 3] is evaluated with the hardware float32 exp / log (absolute error ~1e-7 per
// step), so the recursion costs a few instructions per element instead of software float64 transcendentals.
__device__ __forceinline__ double lse2(double a, double b) {
    const double m = fmax(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + (double)log1pf(__expf((float)(-fabs(a - b))));
}
__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + (double)__logf(__expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m)));
}

// 512 threads: waves 0-3 run the alpha recursion forwards while waves 4-7 run the beta recursion backwards (one
// workgroup barrier per time step serves both); every row of both is stored.  The gradient phase then needs no
// workgroup barrier at all: each wave takes one frame, sums the posterior occupancy per class with fixed-point
// LDS atomics into its own counters and writes that frame's gradient row.
__global__ __launch_bounds__(512) void k_ctc(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                           int T, int Tpad, int V, int S_max, int blank, float gscale,
                                           float* __restrict__ nll_out, float* __restrict__ dlogits,
                                           float* __restrict__ work, int64_t work_per_clip, Bf dlb) {
    extern __shared__ __attribute__((aligned(16))) double smd[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int SPmax = 2 * S_max + 1;
    double* bufA0 = smd;                    // alpha rows (double buffered)
    double* bufA1 = smd + SPmax;
    double* bufB0 = smd + 2 * SPmax;        // beta rows
    double* bufB1 = smd + 3 * SPmax;
    int* lab = reinterpret_cast<int*>(smd + 4 * SPmax);          // [SPmax] extended labels
    unsigned* occ = reinterpret_cast<unsigned*>(lab + SPmax);    // [8][V]
    __shared__ int s_len;
    __shared__ double s_nll;

    const float* lg = logits + (size_t)b * Tpad * V;
    double* wk = reinterpret_cast<double*>(work) + (size_t)b * work_per_clip;
    double* lp = wk;                                         // [T][V]
    double* alpha = lp + (size_t)T * V;                      // [T][SPmax]
    double* beta = alpha + (size_t)T * SPmax;                // [T][SPmax]

    if (tid == 0) {
        int n = 0;
        for (int s = 0; s < S_max; ++s) if (labels[(size_t)b * S_max + s] >= 0) ++n;
        s_len = n;
    }
    // log-softmax, 32 lanes per frame
    for (int t = tid >> 5; t < T; t += 16) {
        const int c = tid & 31;
        float mx = -INFINITY;
        for (int cc = c; cc < V; cc += 32) mx = fmaxf(mx, lg[(size_t)t * V + cc]);
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
        float se = 0.f;
        for (int cc = c; cc < V; cc += 32) se += __expf(lg[(size_t)t * V + cc] - mx);
        for (int o = 16; o > 0; o >>= 1) se += __shfl_xor(se, o, 32);
        const double lz = (double)mx + (double)__logf(se);
        for (int cc = c; cc < V; cc += 32) lp[(size_t)t * V + cc] = (double)lg[(size_t)t * V + cc] - lz;
    }
    __syncthreads();
    const int S = s_len;
    const int SP = 2 * S + 1;
    // extended labels: valid labels are the non-negative entries, in order (HF masked_select)
    if (tid == 0) {
        int k = 0;
        lab[0] = blank;
        for (int s = 0; s < S_max; ++s) {
            const int v = labels[(size_t)b * S_max + s];
            if (v >= 0) { lab[2 * k + 1] = v; lab[2 * k + 2] = blank; ++k; }
        }
    }
    __syncthreads();
    const bool fwd = tid < 256;
    const int ht = tid & 255;
    double* prev = fwd ? bufA0 : bufB0;
    double* cur = fwd ? bufA1 : bufB1;
    // t = 0 (alpha) / t = T-1 (beta)
    for (int s = ht; s < SP; s += 256) {
        double v = -INFINITY;
        if (fwd) {
            if (s == 0) v = lp[blank];
            else if (s == 1) v = lp[lab[1]];
            alpha[s] = v;
        } else {
            if (s == SP - 1 || s == SP - 2) v = lp[(size_t)(T - 1) * V + lab[s]];
            beta[(size_t)(T - 1) * SPmax + s] = v;
        }
        prev[s] = v;
    }
    __syncthreads();
    // the emission log-probability of the NEXT step is fetched while this step computes (keeps the ~L2 latency
    // of the lp read out of the serial alpha / beta chain); slots cover (2S+1) <= 1024
    double em[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = ht + 256 * q;
        em[q] = (s < SP && T > 1) ? lp[(size_t)(fwd ? 1 : T - 2) * V + lab[s]] : 0.0;
    }
    for (int i = 1; i < T; ++i) {
        const int t = fwd ? i : T - 1 - i;
        double emn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = ht + 256 * q;
            const int tn = fwd ? i + 1 : T - 2 - i;
            emn[q] = (s < SP && i + 1 < T) ? lp[(size_t)tn * V + lab[s]] : 0.0;
        }
        for (int s = ht; s < SP; s += 256) {
            const int l = lab[s];
            const int q = (s - ht) >> 8;
            const double e = q == 0 ? em[0] : q == 1 ? em[1] : q == 2 ? em[2] : q == 3 ? em[3] : lp[(size_t)t * V + l];
            double v;
            if (fwd) {
                const double a0 = prev[s];
                const double a1 = s >= 1 ? prev[s - 1] : -INFINITY;
                const double a2 = (s >= 2 && l != blank && l != lab[s - 2]) ? prev[s - 2] : -INFINITY;
                v = lse3(a0, a1, a2) + e;
                alpha[(size_t)t * SPmax + s] = v;
            } else {
                const double b0 = prev[s];
                const double b1 = s + 1 < SP ? prev[s + 1] : -INFINITY;
                const double b2 = (s + 2 < SP && lab[s + 2] != blank && lab[s + 2] != l) ? prev[s + 2] : -INFINITY;
                v = lse3(b0, b1, b2) + e;
                beta[(size_t)t * SPmax + s] = v;
            }
            cur[s] = v;
        }
        __syncthreads();
        double* tmp = prev; prev = cur; cur = tmp;
#pragma unroll
        for (int q = 0; q < 4; ++q) em[q] = emn[q];
    }
    if (tid == 0) {        // alpha_{T-1} is in the forward group's `prev` (thread 0 belongs to it)
        const double l1 = prev[SP - 1];
        const double l2 = SP >= 2 ? prev[SP - 2] : -INFINITY;
        const double nll = -lse2(l1, l2);
        s_nll = nll;
        nll_out[b] = (float)nll;
    }
    __syncthreads();
    if (!dlogits) return;
    const double nll = s_nll;
    float* dl = dlogits + (size_t)b * Tpad * V;
    const size_t dlo = (size_t)b * Tpad * V;
    for (int i = tid; i < (Tpad - T) * V; i += 512) { dl[(size_t)T * V + i] = 0.f; store_bf16(dlb, dlo + (size_t)T * V + i, 0.f); }   // pad frames
    if (!(nll < INFINITY)) {      // infeasible alignment: zero_infinity=False propagates non-finite gradients
        for (int i = tid; i < T * V; i += 512) { dl[i] = NAN; store_bf16(dlb, dlo + i, NAN); }
        return;
    }
    // gradient: one wave per frame, no workgroup barriers
    const int wave = tid >> 6, lane = tid & 63;
    unsigned* wocc = occ + wave * V;
    for (int t = wave; t < T; t += 8) {
        for (int c = lane; c < V; c += 64) wocc[c] = 0u;
        for (int s = lane; s < SP; s += 64) {
            const int l = lab[s];
            const float g = __expf((float)(alpha[(size_t)t * SPmax + s] + beta[(size_t)t * SPmax + s] + nll - lp[(size_t)t * V + l]));
            const unsigned q = (unsigned)(fminf(g, 2.f) * 1073741824.f + 0.5f);
            if (q) atomicAdd(&wocc[l], q);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // this wave's LDS atomics before its reads below
        for (int c = lane; c < V; c += 64) {
            const float gv = gscale * (__expf((float)lp[(size_t)t * V + c]) - (float)wocc[c] * (1.f / 1073741824.f));
            dl[(size_t)t * V + c] = gv;
            store_bf16(dlb, dlo + (size_t)t * V + c, gv);
        }
    }
}

// ---- wave-synchronous CTC (label capacity 2 S_max + 1 <= 1024) ----------------------------------------------------------
// One wave runs alpha forwards, a second wave runs beta backwards, each lane owning NS consecutive states of the extended
// label sequence in registers: no workgroup barrier in the time loop, the two boundary states of the neighbouring lane
// arrive by DPP wave shifts, the emission log-probabilities of the next frame are fetched one step ahead.  The recursion
// is in the LOG domain and therefore total: any target the frame count admits gets a finite loss whatever the dynamic
// range of the posteriors (a probability-domain recursion rescaled by the row maximum — the previous version of this
// kernel — flushes states more than 2^-1074 below the best state of a frame, which on peaked logits is where the whole
// target path lives).  States are float64; only the bounded correction log(sum_i exp(x_i - max)) in
// [0, log 3] goes through the float32 hardware exp2 / log2 (absolute error ~1e-7 per step).  Unreachable states hold
// the finite sentinel CTC_NEG = -1e30 instead of -inf: it absorbs every finite addend exactly, so the step is
// branch-free (skip transitions and states past the end of the sequence enter as "+ 0 or + CTC_NEG").
//   nll = -log(exp(alpha_{T-1}(S'-1)) + exp(alpha_{T-1}(S'-2)))
//   gamma_t(s) = exp(alpha_t(s) + beta_t(s) + nll - lp_t(l_s))                               (posterior occupancy)
// The gradient phase needs no reduction: one wave per frame, fixed-point LDS atomics per class (order independent =>
// bitwise reproducible), dlogits[t][c] = scale * (softmax_t(c) - sum_{s: l_s = c} gamma_t(s)).
constexpr double CTC_NEG = -1.0e30;

// Per-clip work layout (floats): lp [T][V] f64 | alpha [T][64 NS] f64 | beta [T][64 NS] f64 | nll f64 |
// lab [64 NS] i32 (lab[j * 64 + lane] = label of state lane * NS + j, -1 past the end)
struct CtcWork {
    double* lp; double* arow; double* brow; double* nll; int* lab;
};
__device__ __forceinline__ CtcWork ctc_work(float* wk, int T, int V, int row) {
    CtcWork w;
    w.lp = reinterpret_cast<double*>(wk);
    w.arow = w.lp + (size_t)T * V;
    w.brow = w.arow + (size_t)T * row;
    w.nll = w.brow + (size_t)T * row;
    w.lab = reinterpret_cast<int*>(w.nll + 1);
    return w;
}

// log-softmax of every frame of every clip in float64 (from float32 max / sum-exp pieces): 32 lanes per frame
__global__ __launch_bounds__(256) void k_ctc_logsoftmax(const float* __restrict__ logits, int B, int T, int Tpad, int V,
                                                        float* __restrict__ work, int64_t wpc) {
    const int f = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (f >= B * T) return;
    const int b = f / T, t = f - b * T, c = threadIdx.x & 31;
    const float* lg = logits + ((size_t)b * Tpad + t) * V;
    double* lp = reinterpret_cast<double*>(work + (size_t)b * wpc) + (size_t)t * V;
    float mx = -INFINITY;
    for (int cc = c; cc < V; cc += 32) mx = fmaxf(mx, lg[cc]);
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
    float se = 0.f;
    for (int cc = c; cc < V; cc += 32) se += __expf(lg[cc] - mx);
    for (int o = 16; o > 0; o >>= 1) se += __shfl_xor(se, o, 32);
    const double lz = (double)mx + (double)__logf(se);
    // a -inf logit (log-probability -inf) enters the recursion as the finite sentinel, like every other dead state: with a
    // true -inf, max(x0, x1, x2) = -inf in ctc_lse3 makes x - m = NaN for the whole alpha / beta table
    for (int cc = c; cc < V; cc += 32) lp[cc] = fmax((double)lg[cc] - lz, CTC_NEG);
}

// whole-wave shift by one lane through DPP (gfx9 wave_shr / wave_shl): lane i receives lane i-1 (i+1); the first (last)
// lane keeps `old` = CTC_NEG (bound_ctrl off).  A few cycles instead of the ~120-cycle LDS round trip of ds_bpermute.
__device__ __forceinline__ double wave_shr1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(CTC_NEG), __double2loint(v), 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(CTC_NEG), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(CTC_NEG), __double2loint(v), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(CTC_NEG), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// log(exp(x0) + exp(x1) + exp(x2)), float64 in / out, float32 hardware transcendentals on the bounded part
__device__ __forceinline__ double ctc_lse3(double x0, double x1, double x2) {
    const double m = fmax(x0, fmax(x1, x2));
    const float s = __builtin_amdgcn_exp2f((float)(x0 - m) * 1.44269504088896341f) +
                    __builtin_amdgcn_exp2f((float)(x1 - m) * 1.44269504088896341f) +
                    __builtin_amdgcn_exp2f((float)(x2 - m) * 1.44269504088896341f);
    return m + (double)(__builtin_amdgcn_logf(s) * 0.69314718055994531f);
}

// the two recursions of one clip: wave 0 alpha, wave 1 beta; LDS_TAB: the clip's lp table is first copied to LDS
template <int NS, bool LDS_TAB>
__global__ __launch_bounds__(128) void k_ctc_rec(const int32_t* __restrict__ labels, int T, int V, int S_max, int blank,
                                                 float* __restrict__ nll_out, float* __restrict__ work, int64_t wpc) {
    extern __shared__ __attribute__((aligned(16))) double smd[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SPmax = 2 * S_max + 1;
    constexpr int ROW = 64 * NS;
    const CtcWork w = ctc_work(work + (size_t)b * wpc, T, V, ROW);
    int* lab = reinterpret_cast<int*>(smd + (LDS_TAB ? (size_t)T * V : 0));      // [SPmax] extended labels
    __shared__ int s_len;
    // extended labels: valid labels are the non-negative entries, in order (HF masked_select).  The label row is first
    // copied to LDS by all threads (one global round trip instead of S_max dependent ones), then compacted by thread 0.
    int* raw = lab + SPmax;                                       // [S_max]
    for (int s = tid; s < S_max; s += 128) raw[s] = labels[(size_t)b * S_max + s];
    if (LDS_TAB) {          // lp table -> LDS, 8 independent 16-byte loads in flight per thread
        const int n2 = T * V / 2;
        const double2* src = reinterpret_cast<const double2*>(w.lp);
        double2* dst = reinterpret_cast<double2*>(smd);
        const int nfull = n2 / (128 * 8) * (128 * 8);       // whole rounds carry no bounds checks (keeps v[] in registers)
        for (int i0 = 0; i0 < nfull; i0 += 128 * 8) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i0 + u * 128 + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[i0 + u * 128 + tid] = v[u];
        }
        for (int i = nfull + tid; i < n2; i += 128) dst[i] = src[i];
        if (((T * V) & 1) && tid == 0) smd[T * V - 1] = w.lp[T * V - 1];
    }
    __syncthreads();
    if (tid == 0) {
        int k = 0;
        lab[0] = blank;
        for (int s = 0; s < S_max; ++s) {
            const int v = raw[s];
            if (v >= 0) { lab[2 * k + 1] = v; lab[2 * k + 2] = blank; ++k; }
        }
        s_len = k;
    }
    const double* tab = LDS_TAB ? smd : w.lp;
    __syncthreads();
    const int S = s_len, SP = 2 * S + 1;
    int l[NS];
    unsigned skf = 0, skb = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int s = lane * NS + j;
        l[j] = s < SP ? lab[s] : blank;
        if (s < SP && s >= 2 && l[j] != blank && l[j] != lab[s - 2]) skf |= 1u << j;
        if (s + 2 < SP && lab[s + 2] != blank && lab[s + 2] != l[j]) skb |= 1u << j;
        if (wave == 0) w.lab[j * 64 + lane] = s < SP ? l[j] : -1;
    }
    // the direction is a compile-time constant inside the recursion (one instantiation per wave)
    auto recursion = [&](auto dir) {
        constexpr bool fwd = decltype(dir)::value;
        double* rows = fwd ? w.arow : w.brow;
        double a[NS], em[NS], emn[NS];
        double skn[NS], dead[NS];                  // additive masks: 0 or CTC_NEG
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            dead[j] = (lane * NS + j < SP) ? 0.0 : CTC_NEG;
            skn[j] = ((fwd ? skf : skb) & (1u << j)) ? 0.0 : CTC_NEG;
        }
        auto emis = [&](int t, double (&e)[NS]) {
#pragma unroll
            for (int j = 0; j < NS; ++j) e[j] = tab[(size_t)t * V + l[j]] + dead[j];
        };
        auto store_row = [&](int t) {
#pragma unroll
            for (int j = 0; j < NS; ++j) rows[(size_t)t * ROW + j * 64 + lane] = a[j];
        };
        const int t0 = fwd ? 0 : T - 1;
        emis(t0, em);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int s = lane * NS + j;
            const bool on = fwd ? (s == 0 || s == 1) : (s == SP - 1 || s == SP - 2);
            a[j] = (on && s < SP) ? em[j] : CTC_NEG;
        }
        store_row(t0);
        if (T > 1) emis(fwd ? 1 : T - 2, em);
        for (int i = 1; i < T; ++i) {
            const int t = fwd ? i : T - 1 - i;
            if (i + 1 < T) emis(fwd ? i + 1 : T - 2 - i, emn);      // next step's emissions fly under this step
            double n1, n2;                                          // the neighbour lane's two boundary states
            if (fwd) {
                n1 = wave_shr1(a[NS - 1]);
                n2 = NS >= 2 ? wave_shr1(a[NS >= 2 ? NS - 2 : 0]) : wave_shr1(n1);
            } else {
                n1 = wave_shl1(a[0]);
                n2 = NS >= 2 ? wave_shl1(a[NS >= 2 ? 1 : 0]) : wave_shl1(n1);
            }
            double nw[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                double x1, x2;
                if (fwd) {
                    x1 = j >= 1 ? a[j >= 1 ? j - 1 : 0] : n1;
                    x2 = j >= 2 ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? n1 : n2);
                } else {
                    x1 = j + 1 < NS ? a[j + 1 < NS ? j + 1 : 0] : n1;
                    x2 = j + 2 < NS ? a[j + 2 < NS ? j + 2 : 0] : (j + 2 == NS ? n1 : n2);
                }
                nw[j] = ctc_lse3(a[j], x1, x2 + skn[j]) + em[j];
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) { a[j] = nw[j]; em[j] = emn[j]; }
            store_row(t);
        }
        if (fwd) {           // log P = lse(alpha_{T-1}(S'-1), alpha_{T-1}(S'-2)); both live in one lane or in two neighbours
            const double prev = wave_shr1(a[NS - 1]);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int s = lane * NS + j;
                if (s == SP - 1) {
                    const double l2 = SP >= 2 ? (j >= 1 ? a[j >= 1 ? j - 1 : 0] : prev) : CTC_NEG;
                    const double nll = -ctc_lse3(a[j], l2, CTC_NEG);
                    *w.nll = nll;
                    nll_out[b] = nll < 1e29 ? (float)nll : INFINITY;
                }
            }
        }
    };
    if (wave == 0) recursion(std::true_type{}); else recursion(std::false_type{});
}

// The same two recursions with ONE WAVE PER STATE SLOT (round 3): 2 NS waves per clip — waves [0, NS) alpha, [NS, 2 NS) beta — wave j
// of a direction owning state lane * NS + j of every lane.  k_ctc_rec above evaluates its NS states per lane one after the other
// (six lse3 of ~25 dependent-latency instructions each per time step at S = 150: ~0.5 us per step, 240 us for T = 499, with the rest
// of the chip idle between the forward and the backward pass); here a step is ONE lse3 per wave plus the exchange of the two
// neighbour states through a double-buffered LDS array and one workgroup barrier (step t reads buffer t & 1 and writes the other:
// every wave has finished reading a buffer before any wave passes the next barrier and overwrites it).  Same arithmetic per
// state, same row layout in the work buffer (k_ctc_grad reads it unchanged): results are bit-identical to k_ctc_rec.
template <int NS, bool LDS_TAB>
__global__ __launch_bounds__(128 * NS) void k_ctc_rec_mw(const int32_t* __restrict__ labels, int T, int V, int S_max, int blank,
                                                          float* __restrict__ nll_out, float* __restrict__ work, int64_t wpc) {
    static_assert(NS >= 2 && NS <= 8, "one wave per state slot: 4 .. 16 waves");
    extern __shared__ __attribute__((aligned(16))) double smd[];
    constexpr int NT = 128 * NS, ROW = 64 * NS;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fwd = wave < NS;
    const int j = fwd ? wave : wave - NS;                       // state slot of this wave
    const int SPmax = 2 * S_max + 1;
    const CtcWork w = ctc_work(work + (size_t)b * wpc, T, V, ROW);
    double* xch = smd + (LDS_TAB ? (size_t)T * V : 0);          // [2 directions][2 buffers][NS][64]
    int* lab = reinterpret_cast<int*>(xch + 4 * ROW);           // [SPmax] extended labels
    __shared__ int s_len;
    int* raw = lab + SPmax;                                     // [S_max]
    for (int s = tid; s < S_max; s += NT) raw[s] = labels[(size_t)b * S_max + s];
    if (LDS_TAB) {          // lp table -> LDS
        const int n2 = T * V / 2;
        const double2* src = reinterpret_cast<const double2*>(w.lp);
        double2* dst = reinterpret_cast<double2*>(smd);
        for (int i = tid; i < n2; i += NT) dst[i] = src[i];
        if (((T * V) & 1) && tid == 0) smd[T * V - 1] = w.lp[T * V - 1];
    }
    __syncthreads();
    if (tid == 0) {
        int k = 0;
        lab[0] = blank;
        for (int s = 0; s < S_max; ++s) {
            const int v = raw[s];
            if (v >= 0) { lab[2 * k + 1] = v; lab[2 * k + 2] = blank; ++k; }
        }
        s_len = k;
    }
    const double* tab = LDS_TAB ? smd : w.lp;
    __syncthreads();
    const int S = s_len, SP = 2 * S + 1;
    const int s = lane * NS + j;                                // this lane's state
    const int l = s < SP ? lab[s] : blank;
    bool skip;                                                  // the skip transition (s -+ 2) is open
    if (fwd) skip = s < SP && s >= 2 && l != blank && l != lab[s - 2];
    else skip = s + 2 < SP && lab[s + 2] != blank && lab[s + 2] != l;
    if (fwd) w.lab[j * 64 + lane] = s < SP ? l : -1;
    const double skn = skip ? 0.0 : CTC_NEG, dead = s < SP ? 0.0 : CTC_NEG;
    double* rows = fwd ? w.arow : w.brow;
    double* xd = xch + (fwd ? 0 : 2 * ROW);                     // this direction's two buffers
    // neighbours: fwd s - 1, s - 2; bwd s + 1, s + 2 — (slot, lane shift) of each, CTC_NEG past the ends of the wave
    const int j1 = fwd ? (j >= 1 ? j - 1 : NS - 1) : (j + 1 < NS ? j + 1 : 0);
    const int j2 = fwd ? (j >= 2 ? j - 2 : NS + j - 2) : (j + 2 < NS ? j + 2 : j + 2 - NS);
    const int d1 = fwd ? (j >= 1 ? 0 : -1) : (j + 1 < NS ? 0 : 1);
    const int d2 = fwd ? (j >= 2 ? 0 : -1) : (j + 2 < NS ? 0 : 1);
    const int l1 = lane + d1, l2 = lane + d2;
    const bool ok1 = l1 >= 0 && l1 < 64, ok2 = l2 >= 0 && l2 < 64;
    const int o1 = j1 * 64 + (ok1 ? l1 : 0), o2 = j2 * 64 + (ok2 ? l2 : 0), o0 = j * 64 + lane;

    const int t0 = fwd ? 0 : T - 1;
    double em = tab[(size_t)t0 * V + l] + dead;
    const bool on = fwd ? (s == 0 || s == 1) : (s == SP - 1 || s == SP - 2);
    double a = (on && s < SP) ? em : CTC_NEG;
    rows[(size_t)t0 * ROW + o0] = a;
    xd[ROW + o0] = a;                                           // step 1 reads buffer 1
    if (T > 1) em = tab[(size_t)(fwd ? 1 : T - 2) * V + l] + dead;
    for (int i = 1; i < T; ++i) {
        const int t = fwd ? i : T - 1 - i;
        double emn = 0.0;
        if (i + 1 < T) emn = tab[(size_t)(fwd ? i + 1 : T - 2 - i) * V + l] + dead;       // next step's emission flies under this step
        __syncthreads();                                        // every wave's value of the previous step is in buffer i & 1
        const double* in = xd + (i & 1) * ROW;
        const double x1 = ok1 ? in[o1] : CTC_NEG;
        const double x2 = ok2 ? in[o2] : CTC_NEG;
        a = ctc_lse3(a, x1, x2 + skn) + em;
        xd[((i + 1) & 1) * ROW + o0] = a;
        rows[(size_t)t * ROW + o0] = a;
        em = emn;
    }
    __syncthreads();                                            // the last alpha row is in buffer T & 1
    if (tid == 0) {   // log P = lse(alpha_{T-1}(S'-1), alpha_{T-1}(S'-2))
        const double* fin = xch + (T & 1) * ROW;
        const int sa = SP - 1, sb = SP - 2;
        const double va = fin[(sa % NS) * 64 + sa / NS];
        const double vb = SP >= 2 ? fin[(sb % NS) * 64 + sb / NS] : CTC_NEG;
        const double nll = -ctc_lse3(va, vb, CTC_NEG);
        *w.nll = nll;
        nll_out[b] = nll < 1e29 ? (float)nll : INFINITY;
    }
}

// The same with K states per lane and wave (long label sequences: 2 S + 1 up to 1024 = 16 states per lane needs 32 one-state waves,
// more than a workgroup holds): 2 NS / K waves per clip, wave w of a direction owning the K consecutive slots w K .. w K + K - 1 of
// every lane.  A step is K lse3 per wave; neighbours inside the wave's own slots come from its registers (the previous step's
// values), the two beyond its range from the exchange buffer.  Same arithmetic per state and same row layout as k_ctc_rec /
// k_ctc_rec_mw: bit-identical results.  At S = 450, T = 1499 (30 s clips) the one-wave-per-direction kernel evaluated 16 states per
// lane one after the other: ~2 ms of a 73 ms step.
template <int NS, int K, bool LDS_TAB>
__global__ __launch_bounds__(128 * NS / K) void k_ctc_rec_mwk(const int32_t* __restrict__ labels, int T, int V, int S_max, int blank,
                                                               float* __restrict__ nll_out, float* __restrict__ work, int64_t wpc) {
    static_assert(K >= 2 && NS % K == 0 && NS / K >= 2 && 2 * NS / K <= 16, "K states per lane and wave, 4 .. 16 waves");
    extern __shared__ __attribute__((aligned(16))) double smd[];
    constexpr int NWD = NS / K, NT = 128 * NWD, ROW = 64 * NS;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fwd = wave < NWD;
    const int jw = fwd ? wave : wave - NWD;                     // slot group of this wave
    const int SPmax = 2 * S_max + 1;
    const CtcWork w = ctc_work(work + (size_t)b * wpc, T, V, ROW);
    double* xch = smd + (LDS_TAB ? (size_t)T * V : 0);          // [2 directions][2 buffers][NS][64]
    int* lab = reinterpret_cast<int*>(xch + 4 * ROW);           // [SPmax] extended labels
    __shared__ int s_len;
    int* raw = lab + SPmax;                                     // [S_max]
    for (int s = tid; s < S_max; s += NT) raw[s] = labels[(size_t)b * S_max + s];
    if (LDS_TAB) {
        const int n2 = T * V / 2;
        const double2* src = reinterpret_cast<const double2*>(w.lp);
        double2* dst = reinterpret_cast<double2*>(smd);
        for (int i = tid; i < n2; i += NT) dst[i] = src[i];
        if (((T * V) & 1) && tid == 0) smd[T * V - 1] = w.lp[T * V - 1];
    }
    __syncthreads();
    if (tid == 0) {
        int k = 0;
        lab[0] = blank;
        for (int s = 0; s < S_max; ++s) {
            const int v = raw[s];
            if (v >= 0) { lab[2 * k + 1] = v; lab[2 * k + 2] = blank; ++k; }
        }
        s_len = k;
    }
    const double* tab = LDS_TAB ? smd : w.lp;
    __syncthreads();
    const int S = s_len, SP = 2 * S + 1;
    int l[K], o0[K];
    double skn[K], dead[K];
    bool on[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int slot = jw * K + k, s = lane * NS + slot;
        l[k] = s < SP ? lab[s] : blank;
        bool skip;
        if (fwd) skip = s < SP && s >= 2 && l[k] != blank && l[k] != lab[s - 2];
        else skip = s + 2 < SP && lab[s + 2] != blank && lab[s + 2] != l[k];
        skn[k] = skip ? 0.0 : CTC_NEG;
        dead[k] = s < SP ? 0.0 : CTC_NEG;
        on[k] = (fwd ? (s == 0 || s == 1) : (s == SP - 1 || s == SP - 2)) && s < SP;
        o0[k] = slot * 64 + lane;
        if (fwd) w.lab[o0[k]] = s < SP ? l[k] : -1;
    }
    double* rows = fwd ? w.arow : w.brow;
    double* xd = xch + (fwd ? 0 : 2 * ROW);                     // this direction's two buffers
    // the two neighbours beyond the wave's own slots: fwd slots jw K - 1, jw K - 2 (of lane - 1 below slot 0); bwd slots (jw + 1) K,
    // (jw + 1) K + 1 (of lane + 1 past slot NS - 1)
    int on_[3];
    bool okn[3];
#pragma unroll
    for (int m = 1; m <= 2; ++m) {
        int slot = fwd ? jw * K - m : (jw + 1) * K + m - 1, ln = lane;
        if (slot < 0) { slot += NS; ln -= 1; }
        if (slot >= NS) { slot -= NS; ln += 1; }
        okn[m] = ln >= 0 && ln < 64;
        on_[m] = slot * 64 + (okn[m] ? ln : 0);
    }
    const int t0 = fwd ? 0 : T - 1;
    double em[K], a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        em[k] = tab[(size_t)t0 * V + l[k]] + dead[k];
        a[k] = on[k] ? em[k] : CTC_NEG;
        rows[(size_t)t0 * ROW + o0[k]] = a[k];
        xd[ROW + o0[k]] = a[k];                                 // step 1 reads buffer 1
    }
    if (T > 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) em[k] = tab[(size_t)(fwd ? 1 : T - 2) * V + l[k]] + dead[k];
    }
    for (int i = 1; i < T; ++i) {
        const int t = fwd ? i : T - 1 - i;
        double emn[K];
#pragma unroll
        for (int k = 0; k < K; ++k) emn[k] = 0.0;
        if (i + 1 < T) {
#pragma unroll
            for (int k = 0; k < K; ++k) emn[k] = tab[(size_t)(fwd ? i + 1 : T - 2 - i) * V + l[k]] + dead[k];       // next step's emissions fly under this step
        }
        __syncthreads();                                        // every wave's values of the previous step are in buffer i & 1
        const double* in = xd + (i & 1) * ROW;
        double nb[3];
        nb[1] = okn[1] ? in[on_[1]] : CTC_NEG;
        nb[2] = okn[2] ? in[on_[2]] : CTC_NEG;
        double nw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double x1, x2;
            if (fwd) {
                x1 = k >= 1 ? a[k >= 1 ? k - 1 : 0] : nb[1];
                x2 = k >= 2 ? a[k >= 2 ? k - 2 : 0] : nb[2 - k];
            } else {
                x1 = k + 1 < K ? a[k + 1 < K ? k + 1 : 0] : nb[1];
                x2 = k + 2 < K ? a[k + 2 < K ? k + 2 : 0] : nb[k + 3 - K];
            }
            nw[k] = ctc_lse3(a[k], x1, x2 + skn[k]) + em[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            a[k] = nw[k];
            xd[((i + 1) & 1) * ROW + o0[k]] = a[k];
            rows[(size_t)t * ROW + o0[k]] = a[k];
            em[k] = emn[k];
        }
    }
    __syncthreads();                                            // the last alpha row is in buffer T & 1
    if (tid == 0) {   // log P = lse(alpha_{T-1}(S'-1), alpha_{T-1}(S'-2))
        const double* fin = xch + (T & 1) * ROW;
        const int sa = SP - 1, sb = SP - 2;
        const double va = fin[(sa % NS) * 64 + sa / NS];
        const double vb = SP >= 2 ? fin[(sb % NS) * 64 + sb / NS] : CTC_NEG;
        const double nll = -ctc_lse3(va, vb, CTC_NEG);
        *w.nll = nll;
        nll_out[b] = nll < 1e29 ? (float)nll : INFINITY;
    }
}

// gradient rows: one wave per frame (grid: frames / 4 x clips)
template <int NS>
__global__ __launch_bounds__(256) void k_ctc_grad(int T, int Tpad, int V, float gscale, float* __restrict__ dlogits, Bf dlb,
                                                  float* __restrict__ work, int64_t wpc) {
    extern __shared__ __attribute__((aligned(16))) unsigned occs[];     // [4][V]
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave;
    constexpr int ROW = 64 * NS;
    const CtcWork w = ctc_work(work + (size_t)b * wpc, T, V, ROW);
    float* dl = dlogits + (size_t)b * Tpad * V;
    const size_t dlo = (size_t)b * Tpad * V;
    if (t >= T) {                         // pad frames [T, Tpad): zero gradient
        if (t < Tpad) for (int c = lane; c < V; c += 64) { dl[(size_t)t * V + c] = 0.f; store_bf16(dlb, dlo + (size_t)t * V + c, 0.f); }
        return;
    }
    const double nll = *w.nll;
    if (!(nll < 1e29)) {                  // infeasible alignment: zero_infinity=False propagates non-finite gradients
        for (int c = lane; c < V; c += 64) { dl[(size_t)t * V + c] = NAN; store_bf16(dlb, dlo + (size_t)t * V + c, NAN); }
        return;
    }
    unsigned* wocc = occs + wave * V;
    for (int c = lane; c < V; c += 64) wocc[c] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const double* lp = w.lp + (size_t)t * V;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int lj = w.lab[j * 64 + lane];
        if (lj >= 0) {
            const double x = w.arow[(size_t)t * ROW + j * 64 + lane] + w.brow[(size_t)t * ROW + j * 64 + lane] + nll - lp[lj];
            const float g = __expf((float)x);
            const unsigned q = (unsigned)(fminf(g, 2.f) * 1073741824.f + 0.5f);
            if (q) atomicAdd(&wocc[lj], q);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // this wave's LDS atomics before its reads below
    for (int c = lane; c < V; c += 64) {
        const float gv = gscale * (__expf((float)lp[c]) - (float)wocc[c] * (1.f / 1073741824.f));
        dl[(size_t)t * V + c] = gv;
        store_bf16(dlb, dlo + (size_t)t * V + c, gv);
    }
}

template <int NS>
static paa_status launch_ctc_ws(const float* logits, const int32_t* labels, int B, int T, int Tpad, int V, int S_max, int blank,
                                float gscale, float* nll, float* dlogits, Bf dlb, float* work, int64_t wpc, hipStream_t st) {
    const int SPmax = 2 * S_max + 1;
    hipLaunchKernelGGL(k_ctc_logsoftmax, dim3(cdiv((int64_t)B * T, 8)), dim3(256), 0, st, logits, B, T, Tpad, V, work, wpc);
    PAA_LAUNCH_CHECK();
    const size_t tab = sizeof(double) * (size_t)T * V;
    const size_t small = sizeof(int) * ((size_t)SPmax + S_max) + 64;
    if constexpr (NS >= 2 && NS <= 8) {       // one wave per state slot (k_ctc_rec_mw)
        const size_t xch = sizeof(double) * 4 * 64 * NS;
        const bool ltab = tab + xch + small <= 150 * 1024;
        const size_t ldsm = xch + small + (ltab ? tab : 0);
        static bool attr_mw = false;
        if (!attr_mw) {
            PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctc_rec_mw<NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            attr_mw = true;
        }
        if (ltab) hipLaunchKernelGGL((k_ctc_rec_mw<NS, true>), dim3(B), dim3(128 * NS), ldsm, st, labels, T, V, S_max, blank, nll, work, wpc);
        else hipLaunchKernelGGL((k_ctc_rec_mw<NS, false>), dim3(B), dim3(128 * NS), ldsm, st, labels, T, V, S_max, blank, nll, work, wpc);
        PAA_LAUNCH_CHECK();
        if (dlogits) {
            hipLaunchKernelGGL((k_ctc_grad<NS>), dim3(cdiv(Tpad, 4), B), dim3(256), sizeof(unsigned) * 4 * V, st, T, Tpad, V, gscale, dlogits, dlb, work, wpc);
            PAA_LAUNCH_CHECK();
        }
        return PAA_OK;
    }
    if constexpr (NS == 16) {                 // two states per lane and wave, sixteen waves (k_ctc_rec_mwk)
        const size_t xch = sizeof(double) * 4 * 64 * NS;
        const bool ltab = tab + xch + small <= 150 * 1024;
        const size_t ldsm = xch + small + (ltab ? tab : 0);
        static bool attr_mwk = false;
        if (!attr_mwk) {
            PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctc_rec_mwk<NS, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            attr_mwk = true;
        }
        if (ltab) hipLaunchKernelGGL((k_ctc_rec_mwk<NS, 2, true>), dim3(B), dim3(128 * NS / 2), ldsm, st, labels, T, V, S_max, blank, nll, work, wpc);
        else hipLaunchKernelGGL((k_ctc_rec_mwk<NS, 2, false>), dim3(B), dim3(128 * NS / 2), ldsm, st, labels, T, V, S_max, blank, nll, work, wpc);
        PAA_LAUNCH_CHECK();
        if (dlogits) {
            hipLaunchKernelGGL((k_ctc_grad<NS>), dim3(cdiv(Tpad, 4), B), dim3(256), sizeof(unsigned) * 4 * V, st, T, Tpad, V, gscale, dlogits, dlb, work, wpc);
            PAA_LAUNCH_CHECK();
        }
        return PAA_OK;
    }
    const bool lds_tab = tab + small <= 150 * 1024;
    const size_t lds = small + (lds_tab ? tab : 0);
    if (lds_tab) {
        static bool attr = false;
        if (!attr) { PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctc_rec<NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr = true; }
        hipLaunchKernelGGL((k_ctc_rec<NS, true>), dim3(B), dim3(128), lds, st, labels, T, V, S_max, blank, nll, work, wpc);
    } else {
        hipLaunchKernelGGL((k_ctc_rec<NS, false>), dim3(B), dim3(128), lds, st, labels, T, V, S_max, blank, nll, work, wpc);
    }
    PAA_LAUNCH_CHECK();
    if (dlogits) {
        hipLaunchKernelGGL((k_ctc_grad<NS>), dim3(cdiv(Tpad, 4), B), dim3(256), sizeof(unsigned) * 4 * V, st, T, Tpad, V, gscale, dlogits, dlb, work, wpc);
        PAA_LAUNCH_CHECK();
    }
    return PAA_OK;
}

int conv0_chunks(int T) { return cdiv(T, C0_TCH); }
// floats of the conv0 partial-sum workspace: per-chunk channel partials (backward) or Gram partials in f64 (forward)
int64_t conv0_part_floats(int B, int T, int C) {
    return std::max(std::max((int64_t)B * cdiv(T, C0_TCH) * C * 2, (int64_t)B * cdiv(T, C0_GCH) * C0_GQ * 2),
                    conv0_dgrad_part_floats(B, T, C)) + 64;
}

// states per lane of the wave-synchronous kernel for a label capacity (0: use the log-domain kernel)
static int ctc_ws_ns(int SPmax) {
    const int ns = cdiv(SPmax, 64);
    return ns <= 1 ? 1 : ns <= 2 ? 2 : ns <= 4 ? 4 : ns <= 6 ? 6 : ns <= 8 ? 8 : ns <= 16 ? 16 : 0;
}
// work size in floats per clip — block-level kernel: lp [T][V] + alpha, beta [T][2S+1] doubles;
// wave-synchronous kernels: see CtcWork
int64_t ctc_work_floats_per_clip(int T, int V, int S_max) {
    const int64_t SPmax = 2 * (int64_t)S_max + 1;
    const int ns = ctc_ws_ns((int)SPmax);
    if (ns == 0) return 2 * ((int64_t)T * V + 2 * (int64_t)T * SPmax);
    return 2 * ((int64_t)T * V + 2 * (int64_t)T * 64 * ns + 1) + 64 * ns + 16;
}

paa_status ctc(const float* logits, const int32_t* labels, int B, int T, int Tpad, int V, int S_max, int blank,
               float grad_scale, float* nll, float* dlogits, Bf dlb, float* work, hipStream_t st) {
    if (S_max < 1 || S_max > 4000) PAA_FAIL(PAA_ERR_SIZE, "ctc: S_max=%d out of range", S_max);
    if (V > 256) PAA_FAIL(PAA_ERR_SIZE, "ctc: vocab %d > 256", V);
    if ((uintptr_t)work & 7) PAA_FAIL(PAA_ERR_ARG, "ctc: work buffer must be 8-byte aligned");
    const int SPmax = 2 * S_max + 1;
    const int64_t wpc = ctc_work_floats_per_clip(T, V, S_max);
    const int ns = ctc_ws_ns(SPmax);
    if (ns) {                         // wave-synchronous kernel: ns states per lane
#define PAA_CTC_WS(N) case N: return launch_ctc_ws<N>(logits, labels, B, T, Tpad, V, S_max, blank, grad_scale, nll, dlogits, dlb, work, wpc, st)
        switch (ns) { PAA_CTC_WS(1); PAA_CTC_WS(2); PAA_CTC_WS(4); PAA_CTC_WS(6); PAA_CTC_WS(8); PAA_CTC_WS(16); }
#undef PAA_CTC_WS
    }
    const size_t lds = sizeof(double) * 4 * (size_t)SPmax + sizeof(int) * ((size_t)SPmax + 8 * V);
    if (lds > 160 * 1024) PAA_FAIL(PAA_ERR_SIZE, "ctc: label capacity %d needs %zu bytes of LDS", S_max, lds);
    hipLaunchKernelGGL(k_ctc, dim3(B), dim3(512), lds, st, logits, labels, T, Tpad, V, S_max, blank, grad_scale, nll,
                       dlogits, work, wpc / 2, dlb);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

__global__ void k_sum_small(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ double red[256 / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i];
    s = block_sum<double, 256>(s, red);
    if (threadIdx.x == 0) out[0] = (float)s;
}

paa_status sum_small(const float* x, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(256), 0, st, x, n, out);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

}  // namespace paa

using namespace paa;

extern "C" paa_status paa_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* stats,
                                        int rows, int cols, float eps, void* stream) {
    return layernorm_fwd(x, g, b, y, stats, rows, cols, eps, Bf{nullptr, nullptr}, Bf{nullptr, nullptr}, nullptr, (hipStream_t)stream);
}
extern "C" paa_status paa_layernorm_bwd(const float* dy, const float* x, const float* g, const float* stats, float* dx,
                                        int rows, int cols, void* stream) {
    return layernorm_bwd(dy, x, g, stats, nullptr, nullptr, dx, Bf{nullptr, nullptr}, rows, cols, (hipStream_t)stream);
}
extern "C" paa_status paa_softmax_fwd(float* s, int rows, int cols, int ld, float scale, void* stream) {
    return softmax_fwd(s, 1, rows, rows, cols, ld, scale, (hipStream_t)stream);
}
extern "C" paa_status paa_softmax_bwd(float* dp, const float* p, int rows, int cols, int ld, float scale, void* stream) {
    return softmax_bwd(dp, p, 1, rows, rows, cols, ld, scale, (hipStream_t)stream);
}
extern "C" paa_status paa_test_option(int option, int value) {
    if (option == 0) { paa::set_conv0_two_pass(value != 0); return PAA_OK; }
    PAA_FAIL(PAA_ERR_ARG, "paa_test_option: unknown option %d", option);
}
extern "C" int64_t paa_ctc_work_floats(int B, int T, int V, int S_max) { return (int64_t)B * ctc_work_floats_per_clip(T, V, S_max); }
extern "C" paa_status paa_ctc(const float* logits, const int32_t* labels, int B, int T, int V, int S_max, int blank,
                              float grad_scale, float* nll, float* dlogits, float* work, void* stream) {
    return ctc(logits, labels, B, T, T, V, S_max, blank, grad_scale, nll, dlogits, Bf{nullptr, nullptr}, work, (hipStream_t)stream);
}
