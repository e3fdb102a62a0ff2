// Fused (flash-style) multi-head self-attention for gfx950, head_dim 64, bf16 operands, no mask
// (the reference passes no attention_mask: core/loss_helpers.py:21; HF modeling_wav2vec2.py:466-548).
//
//   forward :  O = softmax(Q K^T * scale) V, plus the per-row log-sum-exp (base 2) for the backward
//   backward:  dQ, dK, dV from dO, recomputing P = exp2(S*c - lse) tile by tile; no T x T matrix in HBM.
//
// Mapping (all three kernels share it).  A workgroup is 4 waves; a wave owns 32 "own" positions that sit on
// the 32 MFMA columns (lane & 31), and sweeps the "other" positions in tiles of 32 that sit on the
// accumulator rows: with v_mfma_f32_32x32x16_bf16 the score tile X[other][own] has its rows in the 16
// accumulator registers (row = (e&3) + 8(e>>2) + 4(lane>>5)) and its columns on the lanes.  Products that
// then sum over `other` take bf16(X) straight from the accumulator registers as their B operand (k order
// inside a 16-deep step: 16s + 8(j>>2) + 4h + (j&3); the matching A operand — 4 consecutive `other` rows of one d
// column — comes out of the ROW-MAJOR LDS tile through gfx950's transposing read ds_read_b64_tr_b16, two per step,
// so no transposed copy of any tile is ever staged).  Every statistic of an own position (running max / sum, lse,
// delta) lives in its two lanes (lane, lane^32): no LDS reductions.
//   forward, dQ : own = queries, other = keys          dK/dV : own = keys, other = queries
// Exponentials are raw v_exp_f32 (arguments are <= 8 and flush to 0 far below: no denormal-range fix-up needed).
// Q/K/V/dO are rows of the (M, 3H) / (M, H) bf16 planes the GEMM epilogues write; outputs go back as bf16.
// PREC = 1 (fp32-parity mode): every operand is a hi + lo pair of planes and every product three MFMA passes.
#include "model_kernels.h"

namespace paa {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AT_D = 64;            // head dim
constexpr int AT_RM = 72;           // row-major tile row stride (bf16): 144 B, conflict-free ds_read_b128

// ROWS x 64 tile (rows r0.., valid while < rlim, zero beyond) of a bf16 matrix with row stride ld: global -> registers
// (256 threads, ROWS / 32 16-byte chunks each), and registers -> row-major LDS tile [ROWS][AT_RM].  Split so the next
// tile's global loads fly under the current tile's MFMAs.
template <int ROWS>
__device__ __forceinline__ void tile_load(const unsigned short* __restrict__ src, int64_t ld, int r0, int rlim,
                                          uint4 (&v)[ROWS / 32]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int row = (tid >> 3) + 32 * i;
        v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + row < rlim) v[i] = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + row) * ld + (tid & 7) * 8);
    }
}
template <int ROWS>
__device__ __forceinline__ void tile_store(unsigned short* rm, const uint4 (&v)[ROWS / 32]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) *reinterpret_cast<uint4*>(rm + ((tid >> 3) + 32 * i) * AT_RM + (tid & 7) * 8) = v[i];
}

// A fragment of k-step s from a row-major tile: lane (lr, lh) -> row lr, elements 16s + 8lh .. +7
__device__ __forceinline__ bf16x8 frag_rm(const unsigned short* rm, int lr, int lh, int s) {
    return *reinterpret_cast<const bf16x8*>(rm + lr * AT_RM + 16 * s + 8 * lh);
}
// A fragment of the TRANSPOSED tile for k-step s2, in the accumulator's k order, read from the row-major tile:
// lane (lr, lh) gets column d = dt*32 + lr of rows 16 s2 + 4 lh + (0..3) [elements 0-3] and +8 [elements 4-7].
// ds_read_b64_tr_b16: within a group of 16 lanes, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block, and lane i receives column i of the 4 rows (probed on the device in round 1; the probe is in the git history under tools/scratch/).
__device__ __forceinline__ bf16x8 frag_tr(const unsigned short* rm, int lr, int lh, int dt, int s2) {
    const int i = lr & 15;
    const unsigned short* p = rm + (16 * s2 + 4 * lh + (i >> 2)) * AT_RM + dt * 32 + (lr & 16) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) bf16x4* lds4;
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + 8 * AT_RM));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
// bf16 pack of accumulator registers 8 s2 .. 8 s2 + 7 (the B operand of the follow-up product)
// (pairs go through one v_cvt_pk_bf16_f32 each: bf16_pack2, paa_common.h)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 pack8(const float (&v)[16], int s2) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = bf16_pack2(v[8 * s2 + 2 * j], v[8 * s2 + 2 * j + 1]);
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 pack8v(const f32x16& v, int s2) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = bf16_pack2(v[8 * s2 + 2 * j], v[8 * s2 + 2 * j + 1]);
    return __builtin_bit_cast(bf16x8, r);
}
// own-position operand fragments (B operand of X = other x own): 16 B at row `own`, k = 16s + 8lh
__device__ __forceinline__ void load_own(const unsigned short* __restrict__ src, int64_t ld, int row, int lh, bf16x8 (&f)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(src + (int64_t)row * ld + 16 * s + 8 * lh);
}
// write the transposed accumulators acc[dt][e] (d = dt*32 + (e&3) + 8(e>>2) + 4lh, own = lr) as bf16 rows
__device__ __forceinline__ void store_own(unsigned short* __restrict__ dst, int64_t ld, int row, int lh,
                                          const f32x16 (&acc)[2], float mul) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * lh;
            const unsigned a = bf16_pack2(acc[dt][4 * g4] * mul, acc[dt][4 * g4 + 1] * mul);
            const unsigned b = bf16_pack2(acc[dt][4 * g4 + 2] * mul, acc[dt][4 * g4 + 3] * mul);
            *reinterpret_cast<uint2*>(dst + (int64_t)row * ld + d) = make_uint2(a, b);
        }
}

// Workgroup -> (head bh, 128-position block ob).  The nb blocks of one head re-read the same K / V (or Q / dO) rows;
// consecutive workgroup ids go round-robin over the 8 XCDs, so ids are decoded such that all blocks of a head carry the
// same id % 8 (one XCD's L2 serves the re-reads) and are dispatched close together: 8 heads x nb blocks per group.
__device__ __forceinline__ void attn_block(int nb, int nheads, int& bh, int& ob) {
    const int id = blockIdx.x;
    const int full = (nheads / 8) * 8 * nb;               // ids covered by complete groups of 8 heads
    if (id < full) {
        const int grp = id / (8 * nb), r = id - grp * 8 * nb;
        ob = r >> 3;
        bh = grp * 8 + (r & 7);
    } else {                                              // tail (fewer than 8 heads left): plain order
        const int r = id - full;
        bh = (nheads / 8) * 8 + r / nb;
        ob = r - (r / nb) * nb;
    }
}

// split-bf16 helpers (PREC = 1, the fp32-parity mode): every operand is a hi + lo pair of bf16 planes and a product is
// three MFMA passes, lo*hi + hi*lo + hi*hi — the same scheme as the GEMM kernels
template <int PREC>
__device__ __forceinline__ f32x16 mma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 acc) {
    if constexpr (PREC) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}
// bf16 hi (and lo = bf16(v - hi)) packs of accumulator registers 8 s2 .. 8 s2 + 7
template <int PREC, typename V>
__device__ __forceinline__ void pack8s(const V& v, int s2, bf16x8& hi, bf16x8& lo) {
    u32x4 h, l = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[8 * s2 + 2 * j], b = v[8 * s2 + 2 * j + 1];
        h[j] = bf16_pack2(a, b);
        if constexpr (PREC) l[j] = bf16_pack2(a - __uint_as_float(h[j] << 16), b - __uint_as_float(h[j] & 0xFFFF0000u));
    }
    hi = __builtin_bit_cast(bf16x8, h);
    if constexpr (PREC) lo = __builtin_bit_cast(bf16x8, l);
}
// write the transposed accumulators acc[dt][e] * mul as bf16 rows into the hi (and lo) plane
template <int PREC>
__device__ __forceinline__ void store_own_s(unsigned short* __restrict__ dh, unsigned short* __restrict__ dl, int64_t ld, int row,
                                            int lh, const f32x16 (&acc)[2], float mul) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * lh;
            const float v0 = acc[dt][4 * g4] * mul, v1 = acc[dt][4 * g4 + 1] * mul, v2 = acc[dt][4 * g4 + 2] * mul, v3 = acc[dt][4 * g4 + 3] * mul;
            const unsigned h01 = bf16_pack2(v0, v1), h23 = bf16_pack2(v2, v3);
            *reinterpret_cast<uint2*>(dh + (int64_t)row * ld + d) = make_uint2(h01, h23);
            if constexpr (PREC)
                *reinterpret_cast<uint2*>(dl + (int64_t)row * ld + d) =
                    make_uint2(bf16_pack2(v0 - __uint_as_float(h01 << 16), v1 - __uint_as_float(h01 & 0xFFFF0000u)),
                               bf16_pack2(v2 - __uint_as_float(h23 << 16), v3 - __uint_as_float(h23 & 0xFFFF0000u)));
        }
}

// ------------------------------------------------------------------------------------------ forward
// 64 keys per iteration (two 32-key score tiles), K / V tiles double-buffered in LDS and prefetched through registers:
// one barrier per 64 keys, 16 (x3 in split mode) MFMAs per wave between barriers.
template <int PREC>
__global__ __launch_bounds__(256, 2) void k_attn_fwd(AttnArgs a) {
    constexpr int KT = 64, NPL = PREC ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short sK[NPL][2][KT * AT_RM];
    __shared__ __attribute__((aligned(16))) unsigned short sV[NPL][2][KT * AT_RM];
    int bh, oblk;
    attn_block((a.T + 127) / 128, a.nbh, bh, oblk);
    const int b = bh / a.nh, h = bh - b * a.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int q = oblk * 128 + wave * 32 + lr;
    const int qc = q < a.T ? q : a.T - 1;
    const int64_t ld = 3 * (int64_t)a.H;
    const int64_t hoff = (int64_t)b * a.P * ld + h * AT_D;
    const unsigned short* base[2] = {a.qkv + hoff, PREC ? a.qkv_lo + hoff : nullptr};
    bf16x8 qf[NPL][4];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) load_own(base[pl], ld, qc, lh, qf[pl]);
    const float c = a.scale * 1.44269504088896341f;
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; }
    const int nt = (a.T + KT - 1) / KT;
    uint4 rk[NPL][KT / 32], rv[NPL][KT / 32];
    auto tload = [&](int kt) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            tile_load<KT>(base[pl] + a.H, ld, kt * KT, a.T, rk[pl]);
            tile_load<KT>(base[pl] + 2 * a.H, ld, kt * KT, a.T, rv[pl]);
        }
    };
    auto tstore = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { tile_store<KT>(sK[pl][buf], rk[pl]); tile_store<KT>(sV[pl][buf], rv[pl]); }
    };
    tload(0);
    tstore(0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cb = kt & 1;
        if (kt + 1 < nt) tload(kt + 1);
        f32x16 s[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[u][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kh = frag_rm(sK[0][cb] + u * 32 * AT_RM, lr, lh, ks);
                bf16x8 kl = kh;
                if constexpr (PREC) kl = frag_rm(sK[NPL - 1][cb] + u * 32 * AT_RM, lr, lh, ks);
                s[u] = mma3<PREC>(kh, kl, qf[0][ks], qf[NPL - 1][ks], s[u]);
            }
        }
        // Online softmax on the raw scores (scale folded into the exponent's FMA).  The running maximum is only
        // raised — and l, O rescaled — when some row of the wave exceeds it by more than 2^8: exp2 of a bounded
        // positive excess is harmless in f32 / bf16 and the per-tile rescale of the 32 O registers mostly disappears.
        float mx = -INFINITY;
        if (kt + 1 == nt) {                                  // the only tile that can hold keys >= T (zero-filled rows)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * KT + u * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (key >= a.T) s[u][e] = -INFINITY;
                }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[u][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
        if (__any(mx > m + 8.f)) {
            const float mn = mx > m + 8.f ? mx : m;
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
        }
        float rs = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[u][e] = __builtin_amdgcn_exp2f(fmaf(s[u][e], c, -m)); rs += s[u][e]; }      // s becomes P
        rs += __shfl_xor(rs, 32, 64);
        l += rs;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl_;
                pack8s<PREC>(s[u], s2, ph, pl_);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 vh = frag_tr(sV[0][cb] + u * 32 * AT_RM, lr, lh, dt, s2);
                    bf16x8 vl = vh;
                    if constexpr (PREC) vl = frag_tr(sV[NPL - 1][cb] + u * 32 * AT_RM, lr, lh, dt, s2);
                    o[dt] = mma3<PREC>(vh, vl, ph, pl_, o[dt]);
                }
            }
        if (kt + 1 < nt) tstore((kt + 1) & 1);
        __syncthreads();
    }
    if (q < a.T) {
        const int64_t co = (int64_t)b * a.P * a.H + h * AT_D;
        store_own_s<PREC>(a.ctx + co, PREC ? a.ctx_lo + co : nullptr, a.H, q, lh, o, 1.f / l);
        if (lh == 0) a.lse[(int64_t)bh * a.Tp + q] = m + log2f(l);
    }
}

// ------------------------------------------------------------------------------------- backward: dQ
// own = queries.  Also computes delta = rowsum(dO * O) and stores it for the dK/dV kernel.
template <int PREC>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dq(AttnArgs a) {
    constexpr int NPL = PREC ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short sKb[NPL][2][32 * AT_RM];
    __shared__ __attribute__((aligned(16))) unsigned short sVb[NPL][2][32 * AT_RM];
    int bh, oblk;
    attn_block((a.T + 127) / 128, a.nbh, bh, oblk);
    const int b = bh / a.nh, h = bh - b * a.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int q = oblk * 128 + wave * 32 + lr;
    const int qc = q < a.T ? q : a.T - 1;
    const int64_t ld = 3 * (int64_t)a.H;
    const int64_t hoff = (int64_t)b * a.P * ld + h * AT_D, coff = (int64_t)b * a.P * a.H + h * AT_D;
    const unsigned short* base[2] = {a.qkv + hoff, PREC ? a.qkv_lo + hoff : nullptr};
    const unsigned short* dob[2] = {a.dctx + coff, PREC ? a.dctx_lo + coff : nullptr};
    const unsigned short* ob[2] = {a.ctx + coff, PREC ? a.ctx_lo + coff : nullptr};
    bf16x8 qf[NPL][4], dof[NPL][4];
    float delta = 0.f;
    {
        bf16x8 of[NPL][4];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            load_own(base[pl], ld, qc, lh, qf[pl]);
            load_own(dob[pl], a.H, qc, lh, dof[pl]);
            load_own(ob[pl], a.H, qc, lh, of[pl]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float dv = bf16_to_f32((unsigned short)dof[0][s][j]), ov = bf16_to_f32((unsigned short)of[0][s][j]);
                if constexpr (PREC) { dv += bf16_to_f32((unsigned short)dof[NPL - 1][s][j]); ov += bf16_to_f32((unsigned short)of[NPL - 1][s][j]); }
                delta += dv * ov;
            }
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lse = a.lse[(int64_t)bh * a.Tp + qc];
    if (q < a.T && lh == 0) a.delta[(int64_t)bh * a.Tp + q] = delta;
    const float c = a.scale * 1.44269504088896341f;
    f32x16 dq[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dq[0][e] = 0.f; dq[1][e] = 0.f; }
    const int nt = (a.T + 31) / 32;
    uint4 rk[NPL][1], rv[NPL][1];
    auto tload = [&](int kt) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            tile_load<32>(base[pl] + a.H, ld, kt * 32, a.T, rk[pl]);
            tile_load<32>(base[pl] + 2 * a.H, ld, kt * 32, a.T, rv[pl]);
        }
    };
    auto tstore = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { tile_store<32>(sKb[pl][buf], rk[pl]); tile_store<32>(sVb[pl][buf], rv[pl]); }
    };
    tload(0);
    tstore(0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cb = kt & 1;
        if (kt + 1 < nt) tload(kt + 1);
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kh = frag_rm(sKb[0][cb], lr, lh, ks), vh = frag_rm(sVb[0][cb], lr, lh, ks);
            bf16x8 kl = kh, vl = vh;
            if constexpr (PREC) { kl = frag_rm(sKb[NPL - 1][cb], lr, lh, ks); vl = frag_rm(sVb[NPL - 1][cb], lr, lh, ks); }
            s = mma3<PREC>(kh, kl, qf[0][ks], qf[NPL - 1][ks], s);
            dp = mma3<PREC>(vh, vl, dof[0][ks], dof[NPL - 1][ks], dp);
        }
        float ds[16];                                        // dS / scale (the scale multiplies dQ once, at the end)
#pragma unroll
        for (int e = 0; e < 16; ++e) ds[e] = __builtin_amdgcn_exp2f(fmaf(s[e], c, -lse)) * (dp[e] - delta);
        if (kt + 1 == nt) {                                  // keys >= T: zero-filled K rows would give p = exp2(-lse)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh >= a.T) ds[e] = 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 dh, dl;
            pack8s<PREC>(ds, s2, dh, dl);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 kh = frag_tr(sKb[0][cb], lr, lh, dt, s2);
                bf16x8 kl = kh;
                if constexpr (PREC) kl = frag_tr(sKb[NPL - 1][cb], lr, lh, dt, s2);
                dq[dt] = mma3<PREC>(kh, kl, dh, dl, dq[dt]);
            }
        }
        if (kt + 1 < nt) tstore((kt + 1) & 1);
        __syncthreads();
    }
    if (q < a.T) store_own_s<PREC>(a.dqkv + hoff, PREC ? a.dqkv_lo + hoff : nullptr, ld, q, lh, dq, a.scale);
}

// ---------------------------------------------------------------------------------- backward: dK, dV
// own = keys (lanes), other = queries (accumulator rows).
template <int PREC>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv(AttnArgs a) {
    constexpr int NPL = PREC ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short sQb[NPL][2][32 * AT_RM];
    __shared__ __attribute__((aligned(16))) unsigned short sdOb[NPL][2][32 * AT_RM];
    __shared__ float sLseb[2][32], sDelb[2][32];
    int bh, oblk;
    attn_block((a.T + 127) / 128, a.nbh, bh, oblk);
    const int b = bh / a.nh, h = bh - b * a.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int key = oblk * 128 + wave * 32 + lr;
    const int kc = key < a.T ? key : a.T - 1;
    const int64_t ld = 3 * (int64_t)a.H;
    const int64_t hoff = (int64_t)b * a.P * ld + h * AT_D, coff = (int64_t)b * a.P * a.H + h * AT_D;
    const unsigned short* base[2] = {a.qkv + hoff, PREC ? a.qkv_lo + hoff : nullptr};
    const unsigned short* dob[2] = {a.dctx + coff, PREC ? a.dctx_lo + coff : nullptr};
    bf16x8 kf[NPL][4], vf[NPL][4];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        load_own(base[pl] + a.H, ld, kc, lh, kf[pl]);
        load_own(base[pl] + 2 * a.H, ld, kc, lh, vf[pl]);
    }
    const float c = a.scale * 1.44269504088896341f;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[0][e] = 0.f; dk[1][e] = 0.f; dv[0][e] = 0.f; dv[1][e] = 0.f; }
    const int nt = (a.T + 31) / 32;
    uint4 rq[NPL][1], rdo[NPL][1];
    float rl = INFINITY, rd = 0.f;               // lse / delta of query threadIdx.x of the tile (threads 0..31)
    auto tload = [&](int qt) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            tile_load<32>(base[pl], ld, qt * 32, a.T, rq[pl]);
            tile_load<32>(dob[pl], a.H, qt * 32, a.T, rdo[pl]);
        }
        if (threadIdx.x < 32) {
            const int qq = qt * 32 + threadIdx.x;
            rl = qq < a.T ? a.lse[(int64_t)bh * a.Tp + qq] : INFINITY;     // exp2(-inf) = 0 for pad queries
            rd = qq < a.T ? a.delta[(int64_t)bh * a.Tp + qq] : 0.f;
        }
    };
    auto tstore = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { tile_store<32>(sQb[pl][buf], rq[pl]); tile_store<32>(sdOb[pl][buf], rdo[pl]); }
        if (threadIdx.x < 32) { sLseb[buf][threadIdx.x] = rl; sDelb[buf][threadIdx.x] = rd; }
    };
    tload(0);
    tstore(0);
    __syncthreads();
    for (int qt = 0; qt < nt; ++qt) {
        const int cb = qt & 1;
        const float* sLse = sLseb[cb];
        const float* sDel = sDelb[cb];
        if (qt + 1 < nt) tload(qt + 1);
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 qh = frag_rm(sQb[0][cb], lr, lh, ks), oh = frag_rm(sdOb[0][cb], lr, lh, ks);
            bf16x8 ql = qh, ol = oh;
            if constexpr (PREC) { ql = frag_rm(sQb[NPL - 1][cb], lr, lh, ks); ol = frag_rm(sdOb[NPL - 1][cb], lr, lh, ks); }
            s = mma3<PREC>(qh, ql, kf[0][ks], kf[NPL - 1][ks], s);
            dp = mma3<PREC>(oh, ol, vf[0][ks], vf[NPL - 1][ks], dp);
        }
        float p[16], ds[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
            p[e] = __builtin_amdgcn_exp2f(fmaf(s[e], c, -sLse[r]));
            ds[e] = p[e] * (dp[e] - sDel[r]);                // dS / scale (applied to dK at the end)
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl_, dh, dl;
            pack8s<PREC>(p, s2, ph, pl_);
            pack8s<PREC>(ds, s2, dh, dl);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 oh = frag_tr(sdOb[0][cb], lr, lh, dt, s2), qh = frag_tr(sQb[0][cb], lr, lh, dt, s2);
                bf16x8 ol = oh, ql = qh;
                if constexpr (PREC) { ol = frag_tr(sdOb[NPL - 1][cb], lr, lh, dt, s2); ql = frag_tr(sQb[NPL - 1][cb], lr, lh, dt, s2); }
                dv[dt] = mma3<PREC>(oh, ol, ph, pl_, dv[dt]);
                dk[dt] = mma3<PREC>(qh, ql, dh, dl, dk[dt]);
            }
        }
        if (qt + 1 < nt) tstore((qt + 1) & 1);
        __syncthreads();
    }
    if (key < a.T) {
        store_own_s<PREC>(a.dqkv + hoff + a.H, PREC ? a.dqkv_lo + hoff + a.H : nullptr, ld, key, lh, dk, a.scale);
        store_own_s<PREC>(a.dqkv + hoff + 2 * a.H, PREC ? a.dqkv_lo + hoff + 2 * a.H : nullptr, ld, key, lh, dv, 1.f);
    }
}

static paa_status attn_check(const AttnArgs& a, int B, int head_dim) {
    if (head_dim != AT_D) PAA_FAIL(PAA_ERR_ARG, "fused attention supports head_dim 64 only (got %d)", head_dim);
    if (a.T < 1 || B < 1 || (a.H & 7)) PAA_FAIL(PAA_ERR_ARG, "fused attention: bad shape");
    return PAA_OK;
}

paa_status attn_fwd(const AttnArgs& a, int B, int head_dim, hipStream_t st) {
    PAA_TRY(attn_check(a, B, head_dim));
    AttnArgs f = a; f.nbh = B * a.nh;
    if (a.qkv_lo) hipLaunchKernelGGL(k_attn_fwd<1>, dim3(cdiv(a.T, 128) * B * a.nh), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(k_attn_fwd<0>, dim3(cdiv(a.T, 128) * B * a.nh), dim3(256), 0, st, f);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

paa_status attn_bwd(const AttnArgs& a, int B, int head_dim, hipStream_t st) {
    PAA_TRY(attn_check(a, B, head_dim));
    AttnArgs f = a; f.nbh = B * a.nh;
    if (a.qkv_lo) hipLaunchKernelGGL(k_attn_bwd_dq<1>, dim3(cdiv(a.T, 128) * B * a.nh), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(k_attn_bwd_dq<0>, dim3(cdiv(a.T, 128) * B * a.nh), dim3(256), 0, st, f);
    PAA_LAUNCH_CHECK();
    if (a.qkv_lo) hipLaunchKernelGGL(k_attn_bwd_dkv<1>, dim3(cdiv(a.T, 128) * B * a.nh), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(k_attn_bwd_dkv<0>, dim3(cdiv(a.T, 128) * B * a.nh), dim3(256), 0, st, f);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

}  // namespace paa

// Test entries: bf16 planes as uint16; qkv (B*P, 3H), ctx / dctx (B*P, H), lse / delta (B*nh, Tp) f32.
extern "C" paa_status paa_attn_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int P, int Tp, int H, int nh,
                                   void* stream) {
    paa::AttnArgs a{};
    a.qkv = (const unsigned short*)qkv; a.ctx = (unsigned short*)ctx; a.lse = lse;
    a.T = T; a.P = P; a.Tp = Tp; a.H = H; a.nh = nh; a.scale = 1.0f / sqrtf((float)(H / nh));
    return paa::attn_fwd(a, B, H / nh, (hipStream_t)stream);
}
extern "C" paa_status paa_attn_bwd(const void* qkv, const void* ctx, const float* lse, const void* dctx, float* delta,
                                   void* dqkv, int B, int T, int P, int Tp, int H, int nh, void* stream) {
    paa::AttnArgs a{};
    a.qkv = (const unsigned short*)qkv; a.ctx = (unsigned short*)ctx; a.lse = (float*)lse;
    a.dctx = (const unsigned short*)dctx; a.delta = delta; a.dqkv = (unsigned short*)dqkv;
    a.T = T; a.P = P; a.Tp = Tp; a.H = H; a.nh = nh; a.scale = 1.0f / sqrtf((float)(H / nh));
    return paa::attn_bwd(a, B, H / nh, (hipStream_t)stream);
}
// Split-bf16 (hi + lo planes) forms of the two test entries.
extern "C" paa_status paa_attn_fwd_split(const void* qkv_hi, const void* qkv_lo, void* ctx_hi, void* ctx_lo, float* lse, int B,
                                         int T, int P, int Tp, int H, int nh, void* stream) {
    paa::AttnArgs a{};
    a.qkv = (const unsigned short*)qkv_hi; a.qkv_lo = (const unsigned short*)qkv_lo;
    a.ctx = (unsigned short*)ctx_hi; a.ctx_lo = (unsigned short*)ctx_lo; a.lse = lse;
    a.T = T; a.P = P; a.Tp = Tp; a.H = H; a.nh = nh; a.scale = 1.0f / sqrtf((float)(H / nh));
    if (!qkv_lo || !ctx_lo) { paa::set_error("paa_attn_fwd_split: lo planes required"); return PAA_ERR_ARG; }
    return paa::attn_fwd(a, B, H / nh, (hipStream_t)stream);
}
extern "C" paa_status paa_attn_bwd_split(const void* qkv_hi, const void* qkv_lo, const void* ctx_hi, const void* ctx_lo,
                                         const float* lse, const void* dctx_hi, const void* dctx_lo, float* delta,
                                         void* dqkv_hi, void* dqkv_lo, int B, int T, int P, int Tp, int H, int nh, void* stream) {
    paa::AttnArgs a{};
    a.qkv = (const unsigned short*)qkv_hi; a.qkv_lo = (const unsigned short*)qkv_lo;
    a.ctx = (unsigned short*)ctx_hi; a.ctx_lo = (unsigned short*)ctx_lo; a.lse = (float*)lse;
    a.dctx = (const unsigned short*)dctx_hi; a.dctx_lo = (const unsigned short*)dctx_lo; a.delta = delta;
    a.dqkv = (unsigned short*)dqkv_hi; a.dqkv_lo = (unsigned short*)dqkv_lo;
    a.T = T; a.P = P; a.Tp = Tp; a.H = H; a.nh = nh; a.scale = 1.0f / sqrtf((float)(H / nh));
    if (!qkv_lo || !ctx_lo || !dctx_lo || !dqkv_lo) { paa::set_error("paa_attn_bwd_split: lo planes required"); return PAA_ERR_ARG; }
    return paa::attn_bwd(a, B, H / nh, (hipStream_t)stream);
}
