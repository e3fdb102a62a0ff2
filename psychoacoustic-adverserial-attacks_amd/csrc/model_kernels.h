// Declarations of the non-GEMM kernels (model_kernels.hip) and of the GEMM entry (gemm.hip).
#pragma once
#include <algorithm>

#include "gemm.h"
#include "paa_common.h"

namespace paa {

paa_status gemm(const paa_gemm_desc& d, hipStream_t st);

// y (f32, optional), yb = bf16 planes of y (optional), actb / yact = gelu(y) as bf16 planes / f32 (optional)
paa_status layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* stats, int rows, int cols,
                         float eps, Bf yb, Bf actb, float* yact, hipStream_t st);
// dx (f32, optional; may alias dy), dxb = bf16 planes of dx (optional)
paa_status layernorm_bwd(const float* dy, const float* x, const float* g, const float* stats, const float* add,
                         const float* gelu_pre, float* dx, Bf dxb, int rows, int cols, hipStream_t st);
// n_mat matrices of rows_per_mat valid rows (mat_rows_ld allocated rows each), `cols` valid columns, row stride ld
paa_status softmax_fwd(float* s, int n_mat, int rows_per_mat, int mat_rows_ld, int cols, int ld, float scale,
                       hipStream_t st);
paa_status softmax_bwd(float* dp, const float* p, int n_mat, int rows_per_mat, int mat_rows_ld, int cols, int ld,
                       float scale, hipStream_t st);

// First feature-encoder layer (C_in = 1).  Activations are channel-last with P >= T rows per clip.
struct Conv0Args {
    const float* clean;      // (B, L)
    const float* p;          // (L) or null
    int clamp;               // clamp(clean + p, -1, 1)
    int B, L, T, P, C, k, stride;
    const float* w;          // [C][k]
    const float* bias;       // [C] or null
    const float* gamma;      // norm affine
    const float* beta;
    float eps;
    float* pre;              // (B, P, C) norm output (pre-GELU); when pre16 (group-norm forward only): bf16 storage of gelu'(.)
    int pre16;
    int gate;                // group-norm forward: pre keeps gelu'(.) (f32 unless pre16)
    Bf actb;                 // (B, P, C) GELU(pre) as bf16 planes (the next conv's GEMM operand)
    float* gn_stats;         // group: (B, C, 2) mean, rstd over time
    float* gn_bsums;         // group backward: (B, C, 2) mean_t(dy), mean_t(dy * xhat)
    float* row_stats;        // layer: (B, P, 2) mean, rstd over channels
    const float* dpre;       // backward, layer-norm variant: (B, P, C) f32 gradient wrt `pre`
    Bf dpreb;                // backward, group-norm variant: the same gradient as bf16 planes (a GEMM operand)
    float* G;                // backward, layer-norm variant: (B, P, k) per-frame, per-tap input gradient
    float* G1;               // backward, group-norm variant: (B, P, 16) = dpre x W1_b (GEMM result)
    Bf w1b;                  //   W1_b[j][c] = w[c][j] * gamma_c * rstd_bc as bf16 planes, (B, 16, C)
    float* Mx;               //   (B, k, k)  sum_c W1_b[c][j] * s2_bc * rstd_bc * w[c][j']
    float* kc;               //   (B, 16)    sum_c W1_b[c][j] * (s2_bc * rstd_bc * mean_bc - s1_bc)
    float* part;             // scratch partials
};
// Input sample i of clip b: clamp(clean[b][i] + p[i], -1, 1) (train.py:136) — or clean + p unclamped
// (evaluation.py:16) — or clean alone when p is null.  Never materialised.
__device__ __forceinline__ float in_sample(const Conv0Args& a, int b, int i) {
    float v = a.clean[(size_t)b * a.L + i];
    if (a.p) {
        v += a.p[i];
        if (a.clamp) v = fminf(fmaxf(v, -1.f), 1.f);
    }
    return v;
}
paa_status conv0_gn_forward(const Conv0Args& a, float* part, hipStream_t st);
// conv0_dgrad.hip: single-pass G1 + GroupNorm backward sums (bf16 mode)
bool conv0_dgrad_supported(const Conv0Args& a);
int64_t conv0_dgrad_part_floats(int B, int T, int C);
paa_status conv0_dgrad_fused(const Conv0Args& a, float* part, hipStream_t st);
paa_status conv0_ln_forward(const Conv0Args& a, hipStream_t st);
paa_status conv0_backward(const Conv0Args& a, int layer_norm, int precision, float* part, float* grad, hipStream_t st);
int conv0_chunks(int T);
void set_conv0_two_pass(bool on);     // paa_test_option(0, .)

int64_t ctc_work_floats_per_clip(int T, int V, int S_max);
int64_t conv0_part_floats(int B, int T, int C);
paa_status ctc(const float* logits, const int32_t* labels, int B, int T, int Tpad, int V, int S_max, int blank,
               float grad_scale, float* nll, float* dlogits, Bf dlb, float* work, hipStream_t st);
paa_status sum_small(const float* x, int n, float* out, hipStream_t st);

// Fused attention (attention.hip): bf16 hi planes only (bf16 mode), head_dim 64.
struct AttnArgs {
    const unsigned short* qkv;      // (M, 3H): Q | K | V
    unsigned short* ctx;            // (M, H)   forward output O            (read by the backward for delta)
    float* lse;                     // (B*nh, Tp) log2-domain log-sum-exp
    const unsigned short* dctx;     // (M, H)   dO
    float* delta;                   // (B*nh, Tp) rowsum(dO * O)
    unsigned short* dqkv;           // (M, 3H): dQ | dK | dV
    const unsigned short* qkv_lo;   // lo planes of the same tensors (split-bf16 mode; all null in bf16 mode)
    unsigned short* ctx_lo;
    const unsigned short* dctx_lo;
    unsigned short* dqkv_lo;
    int T, P, Tp, H, nh;
    int nbh;                        // B * nh (set by the launcher)
    float scale;                    // head_dim^-0.5
};
paa_status attn_fwd(const AttnArgs& a, int B, int head_dim, hipStream_t st);
paa_status attn_bwd(const AttnArgs& a, int B, int head_dim, hipStream_t st);

}  // namespace paa
