// Device code shared by the GEMM translation units (gemm.hip, gemm_ring.hip): fragment types, the launch argument
// block and the two epilogues of gemm.h's contract.
#pragma once
#include "gemm.h"
#include "paa_common.h"

namespace paa {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128, G_BK = 32, G_LD = 40 /* bf16 per LDS row */, G_NT = 256;


struct GemmArgs {
    paa_gemm_desc d;
    int tiles_m, tiles_n;
};

// ---- persistent-grid sizing (host) ----------------------------------------------------------------
// (device_cus(): paa_common.h)
// Workgroups of `kernel` one CU holds: a property of the code object (registers, LDS), equal on every gfx950 device, so a
// per-kernel static may cache it; the grid is this times device_cus() of the device in use.
template <typename K>
inline int blocks_per_cu(K kernel, int threads) {
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, threads, 0) != hipSuccess) return 0;
    return per;
}

// ---- epilogue shared by both main loops ---------------------------------------------------------
// Lane (lr, lh) of a wave holds column n = ... + lr and rows (e&3) + 8(e>>2) + 4 lh of each 32 x 32 accumulator.
// All per-element offsets are 32-bit and relative to per-wave base pointers (tile-local row * ld + column).
template <int NJ, int MI>
__device__ __forceinline__ void epilogue(const paa_gemm_desc& d, f32x16 (&acc)[MI][NJ], int mw, int nw, int z1, int z2) {
    // mw / nw: first row / column this lane owns
    const int64_t coff = z1 * d.c_s1 + z2 * d.c_s2 + (int64_t)mw * d.ldc + nw;
    float* __restrict__ C = d.C ? d.C + coff : nullptr;
    const bool x16 = d.aux_bf16 != 0;                   // C_pre / aux stored as bf16
    float* __restrict__ Cp = (d.C_pre && !x16) ? d.C_pre + coff : nullptr;
    unsigned short* __restrict__ Cp16 = (d.C_pre && x16) ? reinterpret_cast<unsigned short*>(d.C_pre) + coff : nullptr;
    unsigned short* __restrict__ Cb = d.Cb ? reinterpret_cast<unsigned short*>(d.Cb) + coff : nullptr;
    unsigned short* __restrict__ Cbl = d.Cb_lo ? reinterpret_cast<unsigned short*>(d.Cb_lo) + coff : nullptr;
    // interleaved bf16 result (gemm.h, Cb_il): element (m, n) at 2 * batch offset + m * 2 ldc + (n / 32) * 64 + n % 32, lo 32 further
    unsigned short* __restrict__ Cil = d.Cb_il ? reinterpret_cast<unsigned short*>(d.Cb_il) + 2 * (z1 * d.c_s1 + z2 * d.c_s2) : nullptr;
    const int64_t xoff = z1 * d.aux_s1 + z2 * d.aux_s2 + (int64_t)mw * d.ld_aux + nw;
    const float* __restrict__ aux = (d.aux && !x16) ? d.aux + xoff : nullptr;
    const unsigned short* __restrict__ aux16 = (d.aux && x16) ? reinterpret_cast<const unsigned short*>(d.aux) + xoff : nullptr;
    const float* __restrict__ res = d.residual ? d.residual + z1 * d.res_s1 + z2 * d.res_s2 + (int64_t)mw * d.ld_res + nw : nullptr;
    const float* __restrict__ bias = d.bias ? d.bias + z2 * d.bias_s2 + nw : nullptr;
    const int ldc = (int)d.ldc, ld_aux = (int)d.ld_aux, ld_res = (int)d.ld_res;
    const int period = d.row_period;
    const int mrem0 = period > 0 ? mw % period : 0;
    const int act = d.act;
    const float alpha = d.alpha;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        // per-accumulator-row-block base pointers: only the 16 in-block row offsets remain per-element scalars
        const int64_t ro = (int64_t)(i * 32) * ldc;
        float* __restrict__ Ci = C ? C + ro : nullptr;
        float* __restrict__ Cpi = Cp ? Cp + ro : nullptr;
        unsigned short* __restrict__ Cpi16 = Cp16 ? Cp16 + ro : nullptr;
        const unsigned short* __restrict__ auxi16 = aux16 ? aux16 + (int64_t)(i * 32) * ld_aux : nullptr;
        unsigned short* __restrict__ Cbi = Cb ? Cb + ro : nullptr;
        unsigned short* __restrict__ Cbli = Cbl ? Cbl + ro : nullptr;
        const float* __restrict__ auxi = aux ? aux + (int64_t)(i * 32) * ld_aux : nullptr;
        const float* __restrict__ resi = res ? res + (int64_t)(i * 32) * ld_res : nullptr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int dn = j * 32;
            if (nw + dn >= d.N) continue;
            const float bv = bias ? bias[dn] : 0.f;
            float ax[16];
            if (act == PAA_ACT_GELU_GRAD) {          // issue the aux loads of this accumulator together
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int dm = (e & 3) + 8 * (e >> 2);
                    ax[e] = (mw + i * 32 + dm < d.M) ? (auxi16 ? bf16_to_f32(auxi16[dm * ld_aux + dn]) : auxi[dm * ld_aux + dn]) : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dm = (e & 3) + 8 * (e >> 2);
                if (mw + i * 32 + dm >= d.M) continue;
                float v = acc[i][j][e] * alpha + bv;
                const int ci = dm * ldc + dn;
                bool dead = false;
                if (period > 0) {
                    int rem = mrem0 + i * 32 + dm;
                    if (rem >= period) rem = (period >= 256) ? rem - period : rem % period;
                    dead = rem >= d.row_valid;
                }
                if (act == PAA_ACT_GELU) {
                    const float keep = d.aux_gate ? gelu_grad_f(v) : v;
                    if (Cpi) Cpi[ci] = dead ? 0.f : keep;
                    if (Cpi16) Cpi16[ci] = dead ? (unsigned short)0 : bf16_bits(keep);
                    v = gelu_f(v);
                } else if (act == PAA_ACT_GELU_GRAD) {
                    v *= d.aux_gate ? ax[e] : gelu_grad_f(ax[e]);
                }
                if (resi) {
                    float rv = resi[dm * ld_res + dn];
                    if (d.res_ln_stats) {          // residual = LayerNorm(stored input), gemm.h
                        const int gm = mw + i * 32 + dm, gn = nw + dn;
                        rv = (rv - d.res_ln_stats[2 * (size_t)gm]) * d.res_ln_stats[2 * (size_t)gm + 1] * d.res_ln_g[gn] + d.res_ln_b[gn];
                    }
                    v += rv;
                }
                if (dead) v = 0.f;
                if (Ci) {
                    if (d.accumulate) v += Ci[ci];
                    Ci[ci] = v;
                }
                if (Cbi) {
                    const unsigned short h = bf16_bits(v);
                    Cbi[ci] = h;
                    if (Cbli) Cbli[ci] = bf16_bits(v - bf16_to_f32(h));
                }
                if (Cil) {
                    const int n = nw + dn;
                    const int64_t o = (int64_t)(mw + i * 32 + dm) * (2 * d.ldc) + ((n >> 5) << 6) + (n & 31);
                    const unsigned short h = bf16_bits(v);
                    Cil[o] = h;
                    Cil[o + 32] = bf16_bits(v - bf16_to_f32(h));
                }
            }
        }
    }
}

// ---- vector epilogue of the bf16-operand kernel ------------------------------------------------------
// An accumulator leaves the MFMA with one COLUMN per lane (4 consecutive rows in 4 registers).  A 4 x 4 transpose
// inside each lane quad (two DPP quad_perm rounds, no LDS) turns that into one ROW per lane with 4 consecutive
// columns in 4 registers; together with the column interleave of the B tile (see store_bf) a lane then owns 8
// consecutive columns of its row: results leave as 16-byte vectors (2 x float4 f32, 1 x uint4 bf16: a wave store
// covers 8 rows x 256 B / 128 B, whole cache lines), and aux / residual / bias arrive the same way, loaded one
// row group ahead of their use.  Needs N, ldc, ld_aux, ld_res and the batch strides to be multiples of 8.
__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
}
// in: lane b of the quad holds M[k][b] in x_k; out: lane b holds M[b][k] in x_k
__device__ __forceinline__ void quad_transpose(float& x0, float& x1, float& x2, float& x3, bool o1, bool o2) {
    float s = dpp_quad_xor1(o1 ? x0 : x1);
    float t = dpp_quad_xor1(o1 ? x2 : x3);
    if (o1) { x0 = s; x2 = t; } else { x1 = s; x3 = t; }
    s = dpp_quad_xor2(o2 ? x0 : x2);
    t = dpp_quad_xor2(o2 ? x1 : x3);
    if (o2) { x0 = s; x1 = t; } else { x2 = s; x3 = t; }
}

// FAST: GELU / GELU' through the branch-free Abramowitz-Stegun erf (|error| <= 1.5e-7 absolute, paa_common.h) instead of
// libm's erff.  Since round 2 BOTH precision modes take it in the vector epilogue: the split-bf16 products themselves
// carry ~2^-16 = 1.5e-5 relative error, a hundred times the erf approximation's, and exact erff made the epilogue of the
// short-K GELU products (FFN up-projection: 96 outputs per lane x ~50 instructions) as long as their MFMA stream.
// `ex` is the one extra f32 operand stream of the epilogue: aux (GELU_GRAD) or the residual — the host takes the
// scalar epilogue when a product asks for both.  All per-lane addresses are 32-bit element offsets from uniform bases.
template <int MI, bool FAST>
__device__ __forceinline__ void epilogue_vec(const paa_gemm_desc& d, f32x16 (&acc)[MI][2], int m0w, int n0w, int z1, int z2, int lane) {
    const int lr = lane & 31, lh = lane >> 5, qa = lr >> 2, qb = lr & 3;
    const bool o1 = qb & 1, o2 = qb & 2;
    const int col = n0w + 8 * qa;                       // this lane's 8 columns
    const int row0 = m0w + 4 * lh + qb;                 // + 32 i + 8 g
    const bool col_ok = col < d.N;
    const int act = d.act;
    const bool gg = act == PAA_ACT_GELU_GRAD;
    const int64_t cbase = z1 * d.c_s1 + z2 * d.c_s2;
    float* __restrict__ C = d.C ? d.C + cbase : nullptr;
    const bool x16 = d.aux_bf16 != 0;                   // C_pre / aux stored as bf16
    const bool gate = d.aux_gate != 0;                  // C_pre / aux hold gelu'(v)
    float* __restrict__ Cp = (d.C_pre && !x16) ? d.C_pre + cbase : nullptr;
    unsigned short* __restrict__ Cp16 = (d.C_pre && x16) ? reinterpret_cast<unsigned short*>(d.C_pre) + cbase : nullptr;
    unsigned short* __restrict__ Cb = d.Cb ? reinterpret_cast<unsigned short*>(d.Cb) + cbase : nullptr;
    unsigned short* __restrict__ Cbl = d.Cb_lo ? reinterpret_cast<unsigned short*>(d.Cb_lo) + cbase : nullptr;
    // interleaved result planes (gemm.h, Cb_il): this lane's 8 columns sit inside one 32-column group, so the hi and the lo vector of
    // a row are two 16-byte stores 64 bytes apart; Cb then points at the array and Cbl at the same array + 32 elements
    const bool cil = d.Cb_il != nullptr;
    if (cil) { Cb = reinterpret_cast<unsigned short*>(d.Cb_il) + 2 * cbase; Cbl = Cb + 32; }
    const bool ex16 = gg && x16;                        // the extra stream is bf16: 8 columns = one 16-byte vector
    const float* __restrict__ ex = gg ? (ex16 ? nullptr : d.aux + z1 * d.aux_s1 + z2 * d.aux_s2)
                                      : (d.residual ? d.residual + z1 * d.res_s1 + z2 * d.res_s2 : nullptr);
    const unsigned short* __restrict__ exh = ex16 ? reinterpret_cast<const unsigned short*>(d.aux) + z1 * d.aux_s1 + z2 * d.aux_s2 : nullptr;
    const unsigned ldc = (unsigned)d.ldc, ldx = (unsigned)(gg ? d.ld_aux : d.ld_res);
    const unsigned co = (unsigned)row0 * ldc + (unsigned)col;
    const unsigned cob = cil ? (unsigned)row0 * 2u * ldc + (((unsigned)col >> 5) << 6) + ((unsigned)col & 31u) : co;      // bf16 result offset
    const unsigned ldcb = cil ? 2u * ldc : ldc;
    const unsigned xo = (unsigned)row0 * ldx + (unsigned)col;
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = 0.f;
    if (d.bias && col_ok) {
        const float* bp = d.bias + z2 * d.bias_s2;
        const float4 b0 = *reinterpret_cast<const float4*>(bp + (unsigned)col);
        const float4 b1 = *reinterpret_cast<const float4*>(bp + (unsigned)col + 4u);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    // residual = LayerNorm(stored input) (gemm.h, res_ln_stats): gain / bias of this lane's 8 columns
    const bool rln = d.res_ln_stats != nullptr && !gg && ex != nullptr;
    // 256-row tiles (MI = 4: 128 accumulator registers per wave at a 256-register budget) cannot afford 16 registers for them across
    // the row groups — held there, the compiler spilled and reloaded other epilogue state around every store burst, each reload behind
    // an s_waitcnt vmcnt(0) that also waits for the stores.  They re-read the 8 columns' gain / bias per row group instead (cache hits;
    // the post-LN residual products have N = 768 and run 192-row tiles, so this path is cold).
    constexpr bool RLN_HELD = MI < 4;
    float lg[8], lb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { lg[k] = 1.f; lb[k] = 0.f; }
    if (RLN_HELD && rln && col_ok) {
        const float4 g0 = *reinterpret_cast<const float4*>(d.res_ln_g + (unsigned)col), g1 = *reinterpret_cast<const float4*>(d.res_ln_g + (unsigned)col + 4u);
        const float4 b0 = *reinterpret_cast<const float4*>(d.res_ln_b + (unsigned)col), b1 = *reinterpret_cast<const float4*>(d.res_ln_b + (unsigned)col + 4u);
        lg[0] = g0.x; lg[1] = g0.y; lg[2] = g0.z; lg[3] = g0.w; lg[4] = g1.x; lg[5] = g1.y; lg[6] = g1.z; lg[7] = g1.w;
        lb[0] = b0.x; lb[1] = b0.y; lb[2] = b0.z; lb[3] = b0.w; lb[4] = b1.x; lb[5] = b1.y; lb[6] = b1.z; lb[7] = b1.w;
    }
    const int period = d.row_period;
    const int mrem0 = period > 0 ? row0 % period : 0;
    const float alpha = d.alpha;
    const bool accum = d.accumulate != 0;
    constexpr int NG = 4 * MI;                           // row groups of 8 rows: (i, g)
    uint4 xv[2][2];                                       // raw bits: 2 x float4, or one uint4 of 8 bf16 in [0]
#pragma unroll
    for (int u = 0; u < 2; ++u) { xv[u][0] = make_uint4(0u, 0u, 0u, 0u); xv[u][1] = xv[u][0]; }
    auto fetch = [&](int grp, uint4 (&v2)[2]) {
        const int dm = (grp >> 2) * 32 + (grp & 3) * 8;
        if (col_ok && row0 + dm < d.M) {
            const unsigned o = xo + (unsigned)dm * ldx;
            if (ex) {
                v2[0] = *reinterpret_cast<const uint4*>(ex + o);
                v2[1] = *reinterpret_cast<const uint4*>(ex + o + 4u);
            } else if (exh) {
                v2[0] = *reinterpret_cast<const uint4*>(exh + o);
            }
        }
    };
    fetch(0, xv[0]);
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        const int i = grp >> 2, gq = grp & 3, u = grp & 1;
        if (grp + 1 < NG) fetch(grp + 1, xv[u ^ 1]);
        const int dm = i * 32 + gq * 8;
        const bool live = col_ok && row0 + dm < d.M;
        bool dead = false;
        if (period > 0) {
            int rem = mrem0 + dm;
            if (rem >= period) rem = (period >= 256) ? rem - period : rem % period;
            dead = rem >= d.row_valid;
        }
        const unsigned ci = co + (unsigned)dm * ldc;
        float2 rst = make_float2(0.f, 1.f);                  // (mean, rstd) of this lane's row of the group (res_ln_stats)
        if (rln && live) rst = *reinterpret_cast<const float2*>(d.res_ln_stats + 2 * (size_t)(row0 + dm));
        unsigned hp[4], lp[4], pp[4];                        // packed bf16 pairs of the 8 columns (result hi / lo, C_pre)
        // the two 4-column halves go through the math one after the other: with the tile's 64 accumulator registers
        // still live there is no room for eight interleaved GELU chains
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float x[4] = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
            quad_transpose(x[0], x[1], x[2], x[3], o1, o2);   // every lane takes part, whatever its bounds
            float e4[4];
            if (ex16) {
                const unsigned w0 = j ? xv[u][0].z : xv[u][0].x, w1 = j ? xv[u][0].w : xv[u][0].y;
                e4[0] = __uint_as_float(w0 << 16); e4[1] = __uint_as_float(w0 & 0xFFFF0000u);
                e4[2] = __uint_as_float(w1 << 16); e4[3] = __uint_as_float(w1 & 0xFFFF0000u);
            } else {
                e4[0] = __uint_as_float(xv[u][j].x); e4[1] = __uint_as_float(xv[u][j].y);
                e4[2] = __uint_as_float(xv[u][j].z); e4[3] = __uint_as_float(xv[u][j].w);
            }
            if (rln) {
                if constexpr (RLN_HELD) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) e4[k] = (e4[k] - rst.x) * rst.y * lg[4 * j + k] + lb[4 * j + k];
                } else {
                    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (col_ok) {
                        g4 = *reinterpret_cast<const float4*>(d.res_ln_g + (unsigned)col + 4u * j);
                        b4 = *reinterpret_cast<const float4*>(d.res_ln_b + (unsigned)col + 4u * j);
                    }
                    e4[0] = (e4[0] - rst.x) * rst.y * g4.x + b4.x; e4[1] = (e4[1] - rst.x) * rst.y * g4.y + b4.y;
                    e4[2] = (e4[2] - rst.x) * rst.y * g4.z + b4.z; e4[3] = (e4[3] - rst.x) * rst.y * g4.w + b4.w;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = x[k] * alpha + bv[4 * j + k];
            if (act == PAA_ACT_GELU) {
                float kp[4];                                  // what the backward pass will want: v, or gelu'(v) (aux_gate)
                if (FAST) {                                   // two values per packed-math GELU
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        f32x2 dg;
                        const f32x2 gv = gelu_both_fast2(f32x2{x[k], x[k + 1]}, dg);
                        kp[k] = gate ? dg.x : x[k]; kp[k + 1] = gate ? dg.y : x[k + 1];
                        x[k] = gv.x; x[k + 1] = gv.y;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        kp[k] = gate ? gelu_grad_f(x[k]) : x[k];
                        x[k] = gelu_f(x[k]);
                    }
                }
                if (Cp && live) *reinterpret_cast<float4*>(Cp + ci + 4u * j) = dead ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(kp[0], kp[1], kp[2], kp[3]);
                if (Cp16) {
                    pp[2 * j] = dead ? 0u : bf16_pack2(kp[0], kp[1]);
                    pp[2 * j + 1] = dead ? 0u : bf16_pack2(kp[2], kp[3]);
                }
                if (ex) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] += e4[k];
                }
            } else if (gg) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] *= gate ? e4[k] : (FAST ? gelu_grad_fast(e4[k]) : gelu_grad_f(e4[k]));
            } else if (ex) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] += e4[k];
            }
            if (dead) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = 0.f;
            }
            if (C && live) {
                if (accum) {
                    const float4 c0 = *reinterpret_cast<const float4*>(C + ci + 4u * j);
                    x[0] += c0.x; x[1] += c0.y; x[2] += c0.z; x[3] += c0.w;
                }
                *reinterpret_cast<float4*>(C + ci + 4u * j) = make_float4(x[0], x[1], x[2], x[3]);
            }
            if (Cb) {                                         // packed conversions: one v_cvt_pk_bf16_f32 per pair and plane
                const unsigned h01 = bf16_pack2(x[0], x[1]), h23 = bf16_pack2(x[2], x[3]);
                hp[2 * j] = h01; hp[2 * j + 1] = h23;
                if (Cbl) {
                    lp[2 * j] = bf16_pack2(x[0] - __uint_as_float(h01 << 16), x[1] - __uint_as_float(h01 & 0xFFFF0000u));
                    lp[2 * j + 1] = bf16_pack2(x[2] - __uint_as_float(h23 << 16), x[3] - __uint_as_float(h23 & 0xFFFF0000u));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (Cp16 && live && act == PAA_ACT_GELU) *reinterpret_cast<uint4*>(Cp16 + ci) = make_uint4(pp[0], pp[1], pp[2], pp[3]);
        if (Cb && live) {
            const unsigned cb = cob + (unsigned)dm * ldcb;
            *reinterpret_cast<uint4*>(Cb + cb) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
            if (Cbl) *reinterpret_cast<uint4*>(Cbl + cb) = make_uint4(lp[0], lp[1], lp[2], lp[3]);
        }
    }
}

}  // namespace paa
