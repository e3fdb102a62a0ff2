// MFMA GEMM for gfx950 (see gemm.h for the contract).
//
// Tile 128 x BN (BN = 128 | 64), BK = 32, 256 threads = 4 waves as 2 (M) x 2 (N); each wave owns a
// 64 x BN/2 sub-tile = 2 x (BN/64) accumulators of v_mfma_f32_32x32x16_bf16 (16 f32 regs each).
// Operands are f32 in HBM: a thread loads float4 vectors, converts to bf16 (and, in split mode, the
// bf16 of the residual) and writes K-contiguous rows of 32 bf16 + 8 pad (80-byte row stride, which
// makes the 16-byte fragment reads of all four ds_read_b128 lane groups conflict-free).  The next
// tile's global loads are issued before the current tile's MFMAs, so HBM/L2 latency hides under them.
// The loaders cover four operand shapes: K-contiguous rows with an arbitrary (even overlapping) row
// stride — which is how the strided 1-D convolutions become plain GEMMs on a channel-last layout —
// segmented K with a zero-filled time window (the grouped positional convolution), and the
// M-/N-contiguous (transposed) operands of the attention backward products.
#include <algorithm>
#include <vector>

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

// ---- epilogue shared by both main loops ---------------------------------------------------------
// Lane (lr, lh) of a wave holds column n = ... + lr and rows (e&3) + 8(e>>2) + 4 lh of each 32 x 32 accumulator.
// All per-element offsets are 32-bit and relative to per-wave base pointers (tile-local row * ld + column).
template <int NJ, int MI>
__device__ __forceinline__ void epilogue(const paa_gemm_desc& d, f32x16 (&acc)[MI][NJ], int mw, int nw, int z1, int z2) {
    // mw / nw: first row / column this lane owns
    const int64_t coff = z1 * d.c_s1 + z2 * d.c_s2 + (int64_t)mw * d.ldc + nw;
    float* __restrict__ C = d.C ? d.C + coff : nullptr;
    float* __restrict__ Cp = d.C_pre ? d.C_pre + coff : nullptr;
    unsigned short* __restrict__ Cb = d.Cb ? reinterpret_cast<unsigned short*>(d.Cb) + coff : nullptr;
    unsigned short* __restrict__ Cbl = d.Cb_lo ? reinterpret_cast<unsigned short*>(d.Cb_lo) + coff : nullptr;
    const float* __restrict__ aux = d.aux ? d.aux + z1 * d.aux_s1 + z2 * d.aux_s2 + (int64_t)mw * d.ld_aux + nw : nullptr;
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
                    ax[e] = (mw + i * 32 + dm < d.M) ? auxi[dm * ld_aux + dn] : 0.f;
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
                    if (Cpi) Cpi[ci] = dead ? 0.f : v;
                    v = gelu_f(v);
                } else if (act == PAA_ACT_GELU_GRAD) {
                    v *= gelu_grad_f(ax[e]);
                }
                if (resi) v += resi[dm * ld_res + dn];
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
            }
        }
    }
}

// ---- global -> register tile loads --------------------------------------------------------------
// K-contiguous operand: ROWS x 32 tile, thread (r = tid>>3 [+32 i], kv = tid&7) loads float4 at k0 + 4 kv.
template <int ROWS, bool IS_A>
__device__ __forceinline__ void load_kc(const paa_gemm_desc& d, const float* __restrict__ base, int64_t ld, int r0,
                                        int k0, int rlim, float4 (&v)[ROWS / 32]) {
    const int tid = threadIdx.x;
    const int k = k0 + ((tid & 7) << 2);
    int64_t koff = k;
    int js = 0, kc = k;
    if (IS_A && d.a_kseg > 0) {
        js = k / d.a_kseg;
        kc = k - js * d.a_kseg;
        koff = (int64_t)js * d.a_kseg_stride + kc;
    }
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int r = r0 + (tid >> 3) + 32 * i;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = (r < rlim) && (k < d.K);
        const float* p;
        if (IS_A && d.a_window) {
            const int tr = r + js - d.a_pad;
            ok = ok && (tr >= 0) && (tr < d.a_rows_valid);
            p = base + (int64_t)tr * ld + kc;
        } else {
            p = base + (int64_t)r * ld + koff;
        }
        if (ok) {
            x = *reinterpret_cast<const float4*>(p);
            if (k + 3 >= d.K) {                       // K tail: the vector may straddle K
                if (k + 1 >= d.K) x.y = 0.f;
                if (k + 2 >= d.K) x.z = 0.f;
                x.w = 0.f;
            }
        }
        v[i] = x;
    }
}

// Row-contiguous ("transposed") operand: element (r, k) at base[k * ld + r]; tile 32 (k) x ROWS.
template <int ROWS>
__device__ __forceinline__ void load_rc(const paa_gemm_desc& d, const float* __restrict__ base, int64_t ld, int r0,
                                        int k0, int rlim, float4 (&v)[ROWS / 32]) {
    constexpr int VPK = ROWS / 4;          // float4 per k-row
    constexpr int KSTEP = G_NT / VPK;      // k-rows per pass
    const int tid = threadIdx.x;
    const int r = r0 + ((tid % VPK) << 2);
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int k = k0 + tid / VPK + KSTEP * i;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < d.K && r < rlim) {
            x = *reinterpret_cast<const float4*>(base + (int64_t)k * ld + r);
            if (r + 3 >= rlim) {
                if (r + 1 >= rlim) x.y = 0.f;
                if (r + 2 >= rlim) x.z = 0.f;
                x.w = 0.f;
            }
        }
        v[i] = x;
    }
}

// ---- register -> LDS (convert to bf16 hi [+ lo]) ------------------------------------------------
template <int ROWS, int PREC>
__device__ __forceinline__ void store_kc(unsigned short* hi, unsigned short* lo, const float4 (&v)[ROWS / 32]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int off = ((tid >> 3) + 32 * i) * G_LD + ((tid & 7) << 2);
        const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        unsigned short h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = bf16_bits(f[j]);
            if (PREC) l[j] = bf16_bits(f[j] - bf16_to_f32(h[j]));
        }
        *reinterpret_cast<uint2*>(hi + off) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        if (PREC)
            *reinterpret_cast<uint2*>(lo + off) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
}

template <int ROWS, int PREC>
__device__ __forceinline__ void store_rc(unsigned short* hi, unsigned short* lo, const float4 (&v)[ROWS / 32]) {
    constexpr int VPK = ROWS / 4;
    constexpr int KSTEP = G_NT / VPK;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int k = tid / VPK + KSTEP * i;
        const int r = (tid % VPK) << 2;
        const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned short h = bf16_bits(f[j]);
            hi[(r + j) * G_LD + k] = h;
            if (PREC) lo[(r + j) * G_LD + k] = bf16_bits(f[j] - bf16_to_f32(h));
        }
    }
}

template <int BN, int PREC, bool AKC, bool BKC>
__global__ __launch_bounds__(G_NT) void k_gemm(GemmArgs g) {
    constexpr int NPL = PREC ? 2 : 1;
    constexpr int NJ = BN / 64;                          // 32-wide accumulator columns per wave
    __shared__ __attribute__((aligned(16))) unsigned short smem[NPL * (G_BM + BN) * G_LD];
    unsigned short* sAh = smem;
    unsigned short* sAl = smem + (PREC ? G_BM * G_LD : 0);
    unsigned short* sBh = smem + NPL * G_BM * G_LD;
    unsigned short* sBl = sBh + (PREC ? BN * G_LD : 0);

    const paa_gemm_desc& d = g.d;
    // XCD-aware tile order: blocks that share (id % 8) — i.e. an XCD's L2 — get a contiguous run of tiles,
    // and tiles that share an A row-panel are adjacent in that run (bijective for any tile count).
    const int nwg = g.tiles_m * g.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * G_BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int z1 = z / d.batch2, z2 = z - z1 * d.batch2;
    const float* A = d.A + z1 * d.a_s1 + z2 * d.a_s2;
    const float* B = d.B + z1 * d.b_s1 + z2 * d.b_s2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[G_BM / 32], rb[BN / 32];
    const int nk = (d.K + G_BK - 1) / G_BK;
    if (AKC) load_kc<G_BM, true>(d, A, d.lda, m0, 0, d.M, ra); else load_rc<G_BM>(d, A, d.lda, m0, 0, d.M, ra);
    if (BKC) load_kc<BN, false>(d, B, d.ldb, n0, 0, d.N, rb); else load_rc<BN>(d, B, d.ldb, n0, 0, d.N, rb);

    for (int kt = 0; kt < nk; ++kt) {
        if (AKC) store_kc<G_BM, PREC>(sAh, sAl, ra); else store_rc<G_BM, PREC>(sAh, sAl, ra);
        if (BKC) store_kc<BN, PREC>(sBh, sBl, rb); else store_rc<BN, PREC>(sBh, sBl, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            const int k0 = (kt + 1) * G_BK;
            if (AKC) load_kc<G_BM, true>(d, A, d.lda, m0, k0, d.M, ra); else load_rc<G_BM>(d, A, d.lda, m0, k0, d.M, ra);
            if (BKC) load_kc<BN, false>(d, B, d.ldb, n0, k0, d.N, rb); else load_rc<BN>(d, B, d.ldb, n0, k0, d.N, rb);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + lr) * G_LD + ks * 16 + lh * 8;
                ah[i] = *reinterpret_cast<const bf16x8*>(sAh + off);
                if (PREC) al[i] = *reinterpret_cast<const bf16x8*>(sAl + off);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int off = (wn * (BN / 2) + j * 32 + lr) * G_LD + ks * 16 + lh * 8;
                bh[j] = *reinterpret_cast<const bf16x8*>(sBh + off);
                if (PREC) bl[j] = *reinterpret_cast<const bf16x8*>(sBl + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (PREC) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    epilogue<NJ, 2>(d, acc, m0 + wm * 64 + 4 * lh, n0 + wn * (BN / 2) + lr, z1, z2);
}

// ---- bf16-operand main loop ------------------------------------------------------------------------
// Both operands K-contiguous bf16 planes.  Tile 128 x BN x 64; a thread moves 16-byte chunks (8 bf16) global ->
// registers -> LDS (ds_write_b128) with no conversion work, rows padded to 72 bf16 (144 B: the four
// ds_read_b128 lane groups hit 16 distinct 16-byte slots).  16 (BN=128) MFMAs per wave per K tile.
constexpr int H_BK = 64, H_LD = 72;

template <int ROWS, bool IS_A, int NT>
__device__ __forceinline__ void load_bf(const paa_gemm_desc& d, const unsigned short* __restrict__ base, int64_t ld,
                                        int r0, int k0, int rlim, uint4 (&v)[ROWS * 8 / NT]) {
    const int tid = threadIdx.x;
    const int k = k0 + ((tid & 7) << 3);
    int64_t koff = k;
    int js = 0, kc = k;
    if (IS_A && d.a_kseg > 0) {
        js = k / d.a_kseg;
        kc = k - js * d.a_kseg;
        koff = (int64_t)js * d.a_kseg_stride + kc;
    }
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; ++i) {
        const int r = r0 + (tid >> 3) + (NT / 8) * i;
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        bool ok = (r < rlim) && (k < d.K);
        const unsigned short* p;
        if (IS_A && d.a_window) {
            const int tr = r + js - d.a_pad;
            ok = ok && (tr >= 0) && (tr < d.a_rows_valid);
            p = base + (int64_t)tr * ld + kc;
        } else {
            p = base + (int64_t)r * ld + koff;
        }
        if (ok) x = *reinterpret_cast<const uint4*>(p);
        v[i] = x;
    }
}

template <int ROWS, int NT>
__device__ __forceinline__ void store_bf(unsigned short* lds, const uint4 (&v)[ROWS * 8 / NT]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; ++i)
        *reinterpret_cast<uint4*>(lds + ((tid >> 3) + (NT / 8) * i) * H_LD + ((tid & 7) << 3)) = v[i];
}

// WM x 2 waves; each wave owns a (BM / WM) x (BN / 2) sub-tile = MI x NJ accumulators of 32 x 32.
template <int BM, int BN, int PREC, int WM>
__global__ __launch_bounds__(WM * 128, (WM == 4 && PREC == 0) ? 4 : 2) void k_gemm_bf(GemmArgs g) {
    constexpr int NT = WM * 128;
    constexpr int NPL = PREC ? 2 : 1;
    constexpr int NJ = BN / 64;
    constexpr int MI = BM / (32 * WM);
    __shared__ __attribute__((aligned(16))) unsigned short smem[NPL * (BM + BN) * H_LD];
    unsigned short* sAh = smem;
    unsigned short* sAl = smem + (PREC ? BM * H_LD : 0);
    unsigned short* sBh = smem + NPL * BM * H_LD;
    unsigned short* sBl = sBh + (PREC ? BN * H_LD : 0);

    const paa_gemm_desc& d = g.d;
    const int nwg = g.tiles_m * g.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int z1 = z / d.batch2, z2 = z - z1 * d.batch2;
    const int64_t aoff = z1 * d.a_s1 + z2 * d.a_s2, boff = z1 * d.b_s1 + z2 * d.b_s2;
    const unsigned short* Ah = reinterpret_cast<const unsigned short*>(d.A) + aoff;
    const unsigned short* Bh = reinterpret_cast<const unsigned short*>(d.B) + boff;
    const unsigned short* Al = PREC ? reinterpret_cast<const unsigned short*>(d.A_lo) + aoff : nullptr;
    const unsigned short* Bl = PREC ? reinterpret_cast<const unsigned short*>(d.B_lo) + boff : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 rah[BM * 8 / NT], rbh[BN * 8 / NT], ral[PREC ? BM * 8 / NT : 1], rbl[PREC ? BN * 8 / NT : 1];
    const int nk = (d.K + H_BK - 1) / H_BK;
    load_bf<BM, true, NT>(d, Ah, d.lda, m0, 0, d.M, rah);
    load_bf<BN, false, NT>(d, Bh, d.ldb, n0, 0, d.N, rbh);
    if constexpr (PREC) {
        load_bf<BM, true, NT>(d, Al, d.lda, m0, 0, d.M, ral);
        load_bf<BN, false, NT>(d, Bl, d.ldb, n0, 0, d.N, rbl);
    }
    for (int kt = 0; kt < nk; ++kt) {
        store_bf<BM, NT>(sAh, rah);
        store_bf<BN, NT>(sBh, rbh);
        if constexpr (PREC) { store_bf<BM, NT>(sAl, ral); store_bf<BN, NT>(sBl, rbl); }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k0 = (kt + 1) * H_BK;
            load_bf<BM, true, NT>(d, Ah, d.lda, m0, k0, d.M, rah);
            load_bf<BN, false, NT>(d, Bh, d.ldb, n0, k0, d.N, rbh);
            if constexpr (PREC) {
                load_bf<BM, true, NT>(d, Al, d.lda, m0, k0, d.M, ral);
                load_bf<BN, false, NT>(d, Bl, d.ldb, n0, k0, d.N, rbl);
            }
        }
#pragma unroll
        for (int ks = 0; ks < H_BK / 16; ++ks) {
            bf16x8 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int off = (wm * (32 * MI) + i * 32 + lr) * H_LD + ks * 16 + lh * 8;
                ah[i] = *reinterpret_cast<const bf16x8*>(sAh + off);
                if (PREC) al[i] = *reinterpret_cast<const bf16x8*>(sAl + off);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int off = (wn * (BN / 2) + j * 32 + lr) * H_LD + ks * 16 + lh * 8;
                bh[j] = *reinterpret_cast<const bf16x8*>(sBh + off);
                if (PREC) bl[j] = *reinterpret_cast<const bf16x8*>(sBl + off);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (PREC) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    epilogue<NJ, MI>(d, acc, m0 + wm * (32 * MI) + 4 * lh, n0 + wn * (BN / 2) + lr, z1, z2);
}

template <int BN, int PREC>
static void launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t st) {
    const bool akc = g.d.a_kcontig != 0, bkc = g.d.b_kcontig != 0;
    if (akc && bkc) hipLaunchKernelGGL((k_gemm<BN, PREC, true, true>), grid, dim3(G_NT), 0, st, g);
    else if (akc && !bkc) hipLaunchKernelGGL((k_gemm<BN, PREC, true, false>), grid, dim3(G_NT), 0, st, g);
    else if (!akc && bkc) hipLaunchKernelGGL((k_gemm<BN, PREC, false, true>), grid, dim3(G_NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm<BN, PREC, false, false>), grid, dim3(G_NT), 0, st, g);
}

// ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream ----------
struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<double> flops;
    std::vector<int> variant;
    size_t n = 0, cap = 0;
};
static GemmProf g_prof;

paa_status gemm(const paa_gemm_desc& d, hipStream_t st) {
    if (!d.A || !d.B || (!d.C && !d.Cb)) PAA_FAIL(PAA_ERR_ARG, "gemm: null operand");
    if (d.accumulate && !d.C) PAA_FAIL(PAA_ERR_ARG, "gemm: accumulate needs the f32 result");
    if (d.Cb_lo && !d.Cb) PAA_FAIL(PAA_ERR_ARG, "gemm: Cb_lo without Cb");
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.batch <= 0 || d.batch2 <= 0) PAA_FAIL(PAA_ERR_SIZE, "gemm: bad dims %d %d %d", d.M, d.N, d.K);
    const int al = d.operand_bf16 ? 7 : 3;     // elements per 16-byte vector - 1
    if ((d.lda & al) || (d.ldb & al) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15))
        PAA_FAIL(PAA_ERR_ARG, "gemm: operands must be 16-byte aligned with leading dimensions that are multiples of %d", al + 1);
    if ((d.a_s1 & al) || (d.a_s2 & al) || (d.b_s1 & al) || (d.b_s2 & al)) PAA_FAIL(PAA_ERR_ARG, "gemm: batch strides must be multiples of %d", al + 1);
    if (d.a_kseg > 0 && ((d.a_kseg & al) || (d.a_kseg_stride & al) || !d.a_kcontig)) PAA_FAIL(PAA_ERR_ARG, "gemm: bad K segmentation");
    if (d.operand_bf16) {
        if (!d.a_kcontig || !d.b_kcontig || (d.K & 7)) PAA_FAIL(PAA_ERR_ARG, "gemm: bf16 operands must be K-contiguous with K %% 8 == 0 (K=%d)", d.K);
        if (d.precision && (!d.A_lo || !d.B_lo || ((uintptr_t)d.A_lo & 15) || ((uintptr_t)d.B_lo & 15)))
            PAA_FAIL(PAA_ERR_ARG, "gemm: split precision needs aligned lo planes");
    }
    if (d.a_window && d.a_kseg <= 0) PAA_FAIL(PAA_ERR_ARG, "gemm: a_window needs a_kseg");
    if (d.act == PAA_ACT_GELU_GRAD && !d.aux) PAA_FAIL(PAA_ERR_ARG, "gemm: GELU_GRAD needs aux");
    GemmArgs g;
    g.d = d;
    const bool narrow = d.N <= 64;
    const int bn = narrow ? 64 : 128;
    // 256 x 128 tile (8 waves, 2 workgroups per CU) for every large-M product; 128 x 128 (4 waves) was 4-8 % faster on
    // the isolated M = 16000, N <= 2304 shapes but made no difference inside the step (A/B on one device)
    const bool tall = d.operand_bf16 && !narrow && d.M >= 2048;
    g.tiles_m = cdiv(d.M, tall ? 256 : G_BM);
    g.tiles_n = cdiv(d.N, bn);
    dim3 grid(g.tiles_m * g.tiles_n, d.batch);
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    if (prof) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
        g_prof.flops[g_prof.n] = 2.0 * d.M * d.N * (double)d.K * d.batch;
        g_prof.variant[g_prof.n] = (tall ? 32 : 0) + (d.operand_bf16 ? 16 : 0) + (narrow ? 8 : 0) + (d.precision ? 4 : 0) + (d.a_kcontig ? 2 : 0) + (d.b_kcontig ? 1 : 0);
    }
    if (d.operand_bf16) {
        if (narrow) { if (d.precision) hipLaunchKernelGGL((k_gemm_bf<128, 64, 1, 2>), grid, dim3(256), 0, st, g); else hipLaunchKernelGGL((k_gemm_bf<128, 64, 0, 2>), grid, dim3(256), 0, st, g); }
        else if (tall) { if (d.precision) hipLaunchKernelGGL((k_gemm_bf<256, 128, 1, 4>), grid, dim3(512), 0, st, g); else hipLaunchKernelGGL((k_gemm_bf<256, 128, 0, 4>), grid, dim3(512), 0, st, g); }
        else { if (d.precision) hipLaunchKernelGGL((k_gemm_bf<128, 128, 1, 2>), grid, dim3(256), 0, st, g); else hipLaunchKernelGGL((k_gemm_bf<128, 128, 0, 2>), grid, dim3(256), 0, st, g); }
    } else if (narrow) { if (d.precision) launch_gemm<64, 1>(g, grid, st); else launch_gemm<64, 0>(g, grid, st); }
    else { if (d.precision) launch_gemm<128, 1>(g, grid, st); else launch_gemm<128, 0>(g, grid, st); }
    if (prof) { (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st); ++g_prof.n; }
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

}  // namespace paa

// Enable (max_launches > 0) or disable (0) event timing of every GEMM launch.  Not capturable.
extern "C" paa_status paa_prof_enable(int max_launches) {
    using namespace paa;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear(); g_prof.n = 0; g_prof.cap = 0; g_prof.on = false;
    if (max_launches <= 0) return PAA_OK;
    g_prof.ev.resize(2 * (size_t)max_launches);
    for (auto& e : g_prof.ev) PAA_HIP(hipEventCreate(&e));
    g_prof.flops.assign(max_launches, 0.0);
    g_prof.variant.assign(max_launches, 0);
    g_prof.cap = max_launches; g_prof.on = true;
    return PAA_OK;
}

// out[64][3] = per kernel variant (tileM256*32 + bf16_operands*16 + narrow*8 + split*4 + a_kcontig*2 + b_kcontig).
// Synchronises on the last recorded event.  Resets the counters.
extern "C" paa_status paa_prof_read(double* out96) {
    using namespace paa;
    if (!out96) PAA_FAIL(PAA_ERR_ARG, "paa_prof_read: null");
    for (int i = 0; i < 192; ++i) out96[i] = 0.0;
    if (g_prof.n == 0) return PAA_OK;
    PAA_HIP(hipEventSynchronize(g_prof.ev[2 * g_prof.n - 1]));
    for (size_t i = 0; i < g_prof.n; ++i) {
        float ms = 0.f;
        PAA_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        const int v = g_prof.variant[i];
        out96[3 * v] += 1.0; out96[3 * v + 1] += ms; out96[3 * v + 2] += g_prof.flops[i];
    }
    g_prof.n = 0;
    return PAA_OK;
}

extern "C" paa_status paa_gemm(const struct paa_gemm_desc* d, void* stream) {
    if (!d) { paa::set_error("paa_gemm: null descriptor"); return PAA_ERR_ARG; }
    return paa::gemm(*d, (hipStream_t)stream);
}
