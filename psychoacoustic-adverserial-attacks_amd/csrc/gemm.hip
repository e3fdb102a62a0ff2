// MFMA GEMMs for gfx950 (see gemm.h for the contract).  One descriptor, four kernels:
//
//  * k_gemm_bf  — bf16-plane operands, both K-contiguous: every conv / linear product of the step.  Tiles 256|192 x 128 x 64
//    with 4 waves of (128|96) x 64 (bf16 mode) or 256 x 128 with 8 waves (split-bf16 mode), 128 x 128 and 128 x 64 for small
//    / irregular shapes.  Persistent tile loop with cross-tile prefetch, buffer-resource loads, register -> LDS staging
//    (ds_write_b128, 144-byte rows), fragment reads one MFMA group ahead, DPP-transposed vector epilogue.
//  * k_gemm_win — the grouped positional convolution (windowed A): input slab staged once in LDS.
//  * k_gemm     — legacy f32-operand kernel (operands converted to bf16 on their way into LDS; transposed-operand
//    loaders): only the materialised attention products of the fp32-parity mode use it.  Tile 128 x BN (BN = 128 | 64),
//    BK = 32, 4 waves as 2 x 2, 80-byte LDS rows.
//
// All of them multiply with v_mfma_f32_32x32x16_bf16 (f32 accumulators: lane (lr, lh) of a wave holds column lr and rows
// (e & 3) + 8 (e >> 2) + 4 lh of a 32 x 32 block) and share the epilogue contract of gemm.h.  The K-contiguous loaders
// take an arbitrary (even overlapping) row stride — which is how the strided 1-D convolutions become plain GEMMs on a
// channel-last layout.  What was tried and dropped on the main loop is recorded in DESIGN.md section 9.
#include <algorithm>
#include <vector>

#include "gemm_dev.h"

namespace paa {

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
// Both operands K-contiguous bf16 planes.  Tile BM x BN x 64; a thread moves 16-byte chunks (8 bf16) global ->
// registers -> LDS (ds_write_b128) with no conversion work, rows padded to 72 bf16 (144 B: the four
// ds_read_b128 lane groups hit 16 distinct 16-byte slots).
// The kernel is persistent: the grid is one wave of resident workgroups and each walks tiles t, t + grid, ...;
// the first K tile of the NEXT output tile is loaded under the last MFMA block of the current one and stored to
// LDS before the epilogue, so neither the load latency at the head of a tile nor the epilogue's memory traffic
// leaves the matrix pipes idle on the short-K (K = 768) products of the encoder.
constexpr int H_BK = 64, H_LD = 72;

// General loader (SEG kernels): segmented K and the zero-filled time window of the grouped positional convolution.
// il (A operand only, gemm.h A_il): 0 = planar; 1 / 2 = the hi / lo part of an interleaved array (`base` = the array, `ld` = 2 lda)
template <int ROWS, bool IS_A, int NT>
__device__ __forceinline__ void load_bf(const paa_gemm_desc& d, const unsigned short* __restrict__ base, int64_t ld,
                                        int r0, int k0, int rlim, uint4 (&v)[ROWS * 8 / NT], int tid, int il = 0) {
    const int k = k0 + ((tid & 7) << 3);
    int64_t koff = il ? ((k >> 5) << 6) + (k & 31) + (il == 2 ? 32 : 0) : k;
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

// Plain K-contiguous rows through a buffer resource that starts at the tile's first row and ends with the operand:
// one 32-bit per-lane byte offset for the whole product, everything else scalar; rows past the operand's end
// (last M / N tile) read zeros from the hardware bounds check instead of branching (the check covers the SGPR
// offset too: probed in round 1, git history: tools/scratch/buf_test.hip).  K must be a multiple of the 64-wide K tile.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const unsigned short* base, int64_t elems) {
    const int64_t bytes = elems > 0 ? 2 * elems : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0,
                                             (int)(unsigned)(bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : bytes), 0x00020000);
}
template <int ROWS, int NT>
__device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned k0_bytes, unsigned step_bytes,
                                         uint4 (&v)[ROWS * 8 / NT]) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; ++i) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k0_bytes + i * step_bytes, 0);
        v[i] = make_uint4(x.x, x.y, x.z, x.w);
    }
}

// PERM (B tile of the vector-epilogue kernels): tile row n of each 64-row group goes to LDS row
// ((n >> 2) & 1) * 32 + (n >> 3) * 4 + (n & 3), so that lane lr of accumulator j multiplies column
// 8 (lr >> 2) + 4 j + (lr & 3): after the quad transpose a lane owns 8 consecutive columns.
// SWZ: unpadded 128-byte rows with the 16-byte chunk index XORed by (row >> 1) & 7 (every ds_read_b128 lane group and
// every 8-lane ds_write_b128 group then covers distinct banks) instead of rows padded to 144 bytes: a 192 x 128 split tile
// is then exactly 80 KB and two workgroups fit a CU.
template <int ROWS, int NT, bool PERM, bool SWZ = false>
__device__ __forceinline__ void store_bf(unsigned short* lds, const uint4 (&v)[ROWS * 8 / NT], int tid) {
    constexpr int LD = SWZ ? 64 : H_LD;
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; ++i) {
        int r = (tid >> 3) + (NT / 8) * i;
        if (PERM) r = (r & ~63) | ((((r >> 2) & 1) << 5) + (((r & 63) >> 3) << 2) + (r & 3));
        const int ch = SWZ ? ((tid & 7) ^ ((r >> 1) & 7)) : (tid & 7);
        *reinterpret_cast<uint4*>(lds + r * LD + (ch << 3)) = v[i];
    }
}

// WM x 2 waves; each wave owns a (BM / WM) x (BN / 2) sub-tile = MI x NJ accumulators of 32 x 32.
template <int BM, int BN, int PREC, int WM, bool VEC, bool SEG, bool SWZ = false>
__global__ __launch_bounds__(WM * 128, (WM == 4 && PREC == 0) ? 4 : 2) void k_gemm_bf(GemmArgs g) {
    constexpr int NT = WM * 128;
    constexpr int LD = SWZ ? 64 : H_LD;                   // LDS row stride in bf16 elements
    constexpr int NPL = PREC ? 2 : 1;
    constexpr int NJ = BN / 64;
    constexpr int MI = BM / (32 * WM);
    static_assert(!VEC || NJ == 2, "vector epilogue needs 64 columns per wave");
    __shared__ __attribute__((aligned(16))) unsigned short smem[NPL * (BM + BN) * LD];
    unsigned short* sAh = smem;
    unsigned short* sAl = smem + (PREC ? BM * LD : 0);
    unsigned short* sBh = smem + NPL * BM * LD;
    unsigned short* sBl = sBh + (PREC ? BN * LD : 0);

    const paa_gemm_desc& d = g.d;
    const int nwg = g.tiles_m * g.tiles_n;
    const int total = nwg * d.batch;
    const int nk = (d.K + H_BK - 1) / H_BK;
    // k_group order of the K tiles (gemm.h): kg_spt slabs per tap, kg_taps taps; 0 = plain K order
    const int kg_spt = (d.k_group > 0 && d.k_group % H_BK == 0 && d.K % d.k_group == 0 && d.K > d.k_group) ? d.k_group / H_BK : 0;
    const int kg_taps = kg_spt ? d.K / d.k_group : 1;

    // tile t -> (batch z, row m0, column n0).  XCD-aware order inside a batch entry: workgroups that share
    // (id % 8) — one XCD's L2 — walk a contiguous run of tiles, A row-panel major (bijective for any tile count).
    struct Tile { int m0, n0, z1, z2; };
    auto decode = [&](int t) {
        Tile c;
        const int z = t / nwg, orig = t - z * nwg;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
        const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
        c.m0 = tm * BM; c.n0 = tn * BN;
        c.z1 = z / d.batch2; c.z2 = z - c.z1 * d.batch2;
        return c;
    };
    uint4 rah[BM * 8 / NT], rbh[BN * 8 / NT], ral[PREC ? BM * 8 / NT : 1], rbl[PREC ? BN * 8 / NT : 1];
    auto load_tile = [&](const Tile& c, int k0, int tid) {
        const int64_t aoff = c.z1 * d.a_s1 + c.z2 * d.a_s2, boff = c.z1 * d.b_s1 + c.z2 * d.b_s2;
        if constexpr (SEG) {
            const bool ail = PREC && d.A_il != nullptr;
            if (ail) load_bf<BM, true, NT>(d, reinterpret_cast<const unsigned short*>(d.A_il) + 2 * aoff, 2 * d.lda, c.m0, k0, d.M, rah, tid, 1);
            else load_bf<BM, true, NT>(d, reinterpret_cast<const unsigned short*>(d.A) + aoff, d.lda, c.m0, k0, d.M, rah, tid);
            load_bf<BN, false, NT>(d, reinterpret_cast<const unsigned short*>(d.B) + boff, d.ldb, c.n0, k0, d.N, rbh, tid);
            if constexpr (PREC) {
                if (ail) load_bf<BM, true, NT>(d, reinterpret_cast<const unsigned short*>(d.A_il) + 2 * aoff, 2 * d.lda, c.m0, k0, d.M, ral, tid, 2);
                else load_bf<BM, true, NT>(d, reinterpret_cast<const unsigned short*>(d.A_lo) + aoff, d.lda, c.m0, k0, d.M, ral, tid);
                load_bf<BN, false, NT>(d, reinterpret_cast<const unsigned short*>(d.B_lo) + boff, d.ldb, c.n0, k0, d.N, rbl, tid);
            }
        } else if (PREC && d.A_il != nullptr) {
            // interleaved A (gemm.h): the 64-wide K tile at k0 is 128 consecutive array elements at 2 k0; chunk kc of it sits at
            // (kc / 32) * 64 + kc % 32, its lo part 32 elements further
            const int64_t a0 = 2 * aoff + (int64_t)c.m0 * (2 * d.lda), b0 = boff + (int64_t)c.n0 * d.ldb;
            const int64_t ae = (int64_t)(d.M - 1 - c.m0) * (2 * d.lda) + 2 * (int64_t)d.K, be = (int64_t)(d.N - 1 - c.n0) * d.ldb + d.K;
            const unsigned kc = (unsigned)((tid & 7) << 3);
            const unsigned va = 2u * ((unsigned)(tid >> 3) * 2u * (unsigned)d.lda + ((kc >> 5) << 6) + (kc & 31u)), vb = 2u * ((unsigned)(tid >> 3) * (unsigned)d.ldb + kc);
            const unsigned sa = 2u * (NT / 8) * 2u * (unsigned)d.lda, sb = 2u * (NT / 8) * (unsigned)d.ldb;
            const __amdgpu_buffer_rsrc_t ra = tile_rsrc(reinterpret_cast<const unsigned short*>(d.A_il) + a0, ae);
            load_buf<BM, NT>(ra, va, 4u * k0, sa, rah);
            load_buf<BN, NT>(tile_rsrc(reinterpret_cast<const unsigned short*>(d.B) + b0, be), vb, 2u * k0, sb, rbh);
            if constexpr (PREC) {
                load_buf<BM, NT>(ra, va + 64u, 4u * k0, sa, ral);
                load_buf<BN, NT>(tile_rsrc(reinterpret_cast<const unsigned short*>(d.B_lo) + b0, be), vb, 2u * k0, sb, rbl);
            }
        } else {
            const int64_t a0 = aoff + (int64_t)c.m0 * d.lda, b0 = boff + (int64_t)c.n0 * d.ldb;
            const int64_t ae = (int64_t)(d.M - 1 - c.m0) * d.lda + d.K, be = (int64_t)(d.N - 1 - c.n0) * d.ldb + d.K;
            const unsigned kc = (unsigned)((tid & 7) << 3);
            const unsigned va = 2u * ((unsigned)(tid >> 3) * (unsigned)d.lda + kc), vb = 2u * ((unsigned)(tid >> 3) * (unsigned)d.ldb + kc);
            const unsigned sa = 2u * (NT / 8) * (unsigned)d.lda, sb = 2u * (NT / 8) * (unsigned)d.ldb;
            load_buf<BM, NT>(tile_rsrc(reinterpret_cast<const unsigned short*>(d.A) + a0, ae), va, 2u * k0, sa, rah);
            load_buf<BN, NT>(tile_rsrc(reinterpret_cast<const unsigned short*>(d.B) + b0, be), vb, 2u * k0, sb, rbh);
            if constexpr (PREC) {
                load_buf<BM, NT>(tile_rsrc(reinterpret_cast<const unsigned short*>(d.A_lo) + a0, ae), va, 2u * k0, sa, ral);
                load_buf<BN, NT>(tile_rsrc(reinterpret_cast<const unsigned short*>(d.B_lo) + b0, be), vb, 2u * k0, sb, rbl);
            }
        }
    };
    auto store_tile = [&](int tid) {
        store_bf<BM, NT, false, SWZ>(sAh, rah, tid);
        store_bf<BN, NT, VEC, SWZ>(sBh, rbh, tid);
        if constexpr (PREC) { store_bf<BM, NT, false, SWZ>(sAl, ral, tid); store_bf<BN, NT, VEC, SWZ>(sBl, rbl, tid); }
    };

    int t = blockIdx.x;
    if (t >= total) return;
    Tile cur = decode(t);
    load_tile(cur, 0, threadIdx.x);
    store_tile(threadIdx.x);
    for (;;) {
        // The thread index is made opaque once per tile: everything derived from it (global / LDS addresses, lane
        // coordinates) is then recomputed per tile instead of being kept alive — and spilled — across the epilogue.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave >> 1, wn = wave & 1;
        const int lr = lane & 31, lh = lane >> 5;
        const int tnx = t + gridDim.x;
        const bool more = tnx < total;
        const Tile nxt = more ? decode(tnx) : cur;
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        int ktap = 0, kc = 0;                                  // (tap, channel slab) of K tile kt in the k_group order
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                                   // K tile kt is in LDS
            const bool last = kt + 1 == nk;
            int knext = kt + 1;
            if (kg_spt) {
                if (++ktap == kg_taps) { ktap = 0; ++kc; }
                knext = ktap * kg_spt + kc;
            }
            if (!last) load_tile(cur, knext * H_BK, tid);
            else if (more) load_tile(nxt, 0, tid);
            // Fragment reads run one MFMA group ahead of their use: while the MFMAs of (ks, i) issue, the A fragment
            // of the next (ks, i) — and at the end of a ks step the B fragments of the next one — are already on their
            // way from LDS.  The sched_group_barrier chain pins that interleaving (1 read group : 1 MFMA group).
            {
                constexpr int KS = H_BK / 16;
                constexpr int NM = NJ * (PREC ? 3 : 1);                  // MFMAs per (ks, i)
                // k slice ks of the fragment row: 16-byte chunk 2 ks + lh, XOR-swizzled by the row in the unpadded layout
                int offk[KS];
#pragma unroll
                for (int q = 0; q < KS; ++q) offk[q] = SWZ ? (((2 * q + lh) ^ ((lr >> 1) & 7)) << 3) : (lh * 8 + q * 16);
                const unsigned short* pa = sAh + (wm * (32 * MI) + lr) * LD;
                const unsigned short* pb = sBh + (wn * (BN / 2) + lr) * LD;
                constexpr int LO_A = BM * LD, LO_B = BN * LD;            // hi -> lo plane distance (PREC only)
                bf16x8 bh[NJ], bl[NJ], bhn[NJ], bln[NJ], ah, al, ahn, aln;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * LD + offk[0]);
                    if (PREC) bl[j] = *reinterpret_cast<const bf16x8*>(pb + LO_B + j * 32 * LD + offk[0]);
                    bhn[j] = bh[j]; bln[j] = bl[j];
                }
                ah = *reinterpret_cast<const bf16x8*>(pa + offk[0]);
                if (PREC) al = *reinterpret_cast<const bf16x8*>(pa + LO_A + offk[0]);
                ahn = ah; aln = al;
                __builtin_amdgcn_sched_group_barrier(0x100, (NJ + 1) * NPL, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int ni = (i + 1 < MI) ? i + 1 : 0, nks = (i + 1 < MI) ? ks : ks + 1;
                        if (nks < KS) {
                            ahn = *reinterpret_cast<const bf16x8*>(pa + ni * 32 * LD + offk[nks < KS ? nks : 0]);
                            if (PREC) aln = *reinterpret_cast<const bf16x8*>(pa + LO_A + ni * 32 * LD + offk[nks < KS ? nks : 0]);
                            if (ni == 0) {
#pragma unroll
                                for (int j = 0; j < NJ; ++j) {
                                    bhn[j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * LD + offk[nks < KS ? nks : 0]);
                                    if (PREC) bln[j] = *reinterpret_cast<const bf16x8*>(pb + LO_B + j * 32 * LD + offk[nks < KS ? nks : 0]);
                                }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if (PREC) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                            }
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                        }
                        if (nks < KS) {
                            if (ni == 0) __builtin_amdgcn_sched_group_barrier(0x100, (NJ + 1) * NPL, 0);
                            else __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
                        ah = ahn; al = aln;
                        if (ni == 0) {
#pragma unroll
                            for (int j = 0; j < NJ; ++j) { bh[j] = bhn[j]; bl[j] = bln[j]; }
                        }
                    }
            }
            __syncthreads();                                   // every wave is done with K tile kt
            if (!last || more) store_tile(tid);
        }
        {
            int te = tid;
            asm volatile("" : "+v"(te));
            const int el = te & 63, ew = te >> 6;
            if constexpr (VEC) epilogue_vec<MI, true>(d, acc, cur.m0 + (ew >> 1) * (32 * MI), cur.n0 + (ew & 1) * (BN / 2), cur.z1, cur.z2, el);
            else epilogue<NJ, MI>(d, acc, cur.m0 + (ew >> 1) * (32 * MI) + 4 * (el >> 5), cur.n0 + (ew & 1) * (BN / 2) + (el & 31), cur.z1, cur.z2);
        }
        if (!more) break;
        cur = nxt;
        t = tnx;
    }
}

// ---- grouped positional convolution: windowed A, slab in LDS ------------------------------------------------------------
// C[z][m][n] = sum_{tap} sum_{ci < KS} A_z[m + tap - pad][ci] * B_z2[n][tap * KS + ci], rows outside [0, a_rows_valid) zero
// (a_window products with a_kseg = KS in {48, 64}, N <= 64: wav2vec2's 128-tap grouped conv and its dgrad).  Through
// the general loader every K tile re-reads its shifted A rows from L2 — ~100 times per output tile; here the
// 128 + taps - 1 input rows a 128-row output tile touches are loaded ONCE into an LDS slab and every tap reads its
// shifted 32-row window of it (row stride KS + 8 bf16: conflict-free ds_read_b128).  The weights stream through a
// register-prefetched LDS stage of TB = 4 taps.  4 waves, 32 MI output rows x 64 (N padded) columns each: with MI = 2
// a wave issues 4 MFMAs per 4 fragment reads (MI = 1: 2 per 3) and the slab's halo of taps - 1 rows is amortised over
// 256 output rows instead of 128 (+11 % on the kernel); fragment reads run one k slice ahead of the MFMAs (+3 %).
// Tried without effect: weight stages requested two ahead through a second register set (spills in split mode), weight
// rows padded off the 4 KB stride (L2 channel spread).  The kernel sits at ~0.6 PFLOP/s (0.8 counting the 16 padded columns).
// WV waves of 32 MI rows each: the split-mode tile (two planes of slab and weights: 137 KB of LDS) leaves room for ONE workgroup per CU,
// so it runs as 8 waves of 32 rows (two instruction streams per SIMD) instead of 4 waves of 64 rows.
template <int PREC, int KS, int MI, int WV = 4>
__global__ __launch_bounds__(WV * 64, WV == 4 ? 2 : 1) void k_gemm_win(GemmArgs g, int taps) {
    constexpr int NT = WV * 64;
    constexpr int BM = 32 * MI * WV, TB = 4, NPL = PREC ? 2 : 1;
    constexpr int ALD = KS + 8;                            // slab row stride (bf16)
    constexpr int BLD = TB * KS + 8;                       // weight stage row stride (bf16)
    constexpr int CPR = KS / 8;                            // 16-byte chunks per slab row
    constexpr int BCH = TB * KS / 8;                       // chunks per weight row per stage
    constexpr int NB = (64 * BCH + NT - 1) / NT;           // weight chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned short smw[];
    const paa_gemm_desc& d = g.d;
    const int srows = BM + taps - 1;
    unsigned short* sA = smw;                              // [NPL][srows][ALD]
    unsigned short* sB = smw + NPL * srows * ALD;          // [NPL][64][BLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int z = blockIdx.y, z1 = z / d.batch2, z2 = z - z1 * d.batch2;
    const int64_t aoff = z1 * d.a_s1 + z2 * d.a_s2, boff = z1 * d.b_s1 + z2 * d.b_s2;
    const unsigned short* Ap[2] = {reinterpret_cast<const unsigned short*>(d.A) + aoff,
                                   PREC ? reinterpret_cast<const unsigned short*>(d.A_lo) + aoff : nullptr};
    const unsigned short* Bp[2] = {reinterpret_cast<const unsigned short*>(d.B) + boff,
                                   PREC ? reinterpret_cast<const unsigned short*>(d.B_lo) + boff : nullptr};
    // slab: global row m0 - pad + sr, zero outside the clip
    for (int i = tid; i < srows * CPR; i += NT) {
        const int sr = i / CPR, ch = i - sr * CPR;
        const int tr = m0 - d.a_pad + sr;
        const bool ok = tr >= 0 && tr < d.a_rows_valid;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if (ok) x = *reinterpret_cast<const uint4*>(Ap[pl] + (int64_t)tr * d.lda + ch * 8);
            *reinterpret_cast<uint4*>(sA + (pl * srows + sr) * ALD + ch * 8) = x;
        }
    }
    // weight rows >= N stay zero for the whole kernel
    for (int i = tid; i < NPL * 64 * BLD / 8; i += NT) reinterpret_cast<uint4*>(sB)[i] = make_uint4(0u, 0u, 0u, 0u);
    uint4 rb[NPL][NB];
    auto bload = [&](int st) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + NT * u;
            const int n = i / BCH, ch = i - n * BCH;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                rb[pl][u] = make_uint4(0u, 0u, 0u, 0u);
                if (i < 64 * BCH && n < d.N) rb[pl][u] = *reinterpret_cast<const uint4*>(Bp[pl] + (int64_t)n * d.ldb + st * (TB * KS) + ch * 8);
            }
        }
    };
    auto bstore = [&]() {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + NT * u;
            const int n = i / BCH, ch = i - n * BCH;
            if (i < 64 * BCH && n < d.N) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(sB + (pl * 64 + n) * BLD + ch * 8) = rb[pl][u];
            }
        }
    };
    __syncthreads();                                       // zero fill before the first stage lands on top of it
    bload(0);
    bstore();
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nst = taps / TB;
    for (int st = 0; st < nst; ++st) {
        __syncthreads();                                   // stage st (and, first time, the slab) is in LDS
        if (st + 1 < nst) bload(st + 1);
        // fragments of step q + 1 (a step = one 16-deep k slice of one tap) are read under the MFMAs of step q
        constexpr int NSTEP = TB * (KS / 16);
        const unsigned short* pa0 = sA + (wave * (32 * MI) + lr + st * TB) * ALD + lh * 8;
        const unsigned short* pb0 = sB + lr * BLD + lh * 8;
        bf16x8 ah[2][MI], al[2][MI], bh[2][2], bl[2][2];
        auto fload = [&](int q, int buf) {
            const int tp = q / (KS / 16), ks = q - tp * (KS / 16);
            const unsigned short* pa = pa0 + tp * ALD + ks * 16;
            const unsigned short* pb = pb0 + tp * KS + ks * 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[buf][i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * ALD);
                if (PREC) al[buf][i] = *reinterpret_cast<const bf16x8*>(pa + (srows + i * 32) * ALD);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[buf][j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * BLD);
                if (PREC) bl[buf][j] = *reinterpret_cast<const bf16x8*>(pb + (64 + j * 32) * BLD);
            }
        };
        fload(0, 0);
#pragma unroll
        for (int q = 0; q < NSTEP; ++q) {
            const int cb = q & 1;
            if (q + 1 < NSTEP) fload(q + 1, cb ^ 1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (PREC) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb][i], bh[cb][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][i], bl[cb][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][i], bh[cb][j], acc[i][j], 0, 0, 0);
                }
            if constexpr (PREC == 0) {                     // one fragment read of the next step behind each MFMA of this one
#pragma unroll
                for (int r = 0; r < (MI + 2 > 2 * MI ? MI + 2 : 2 * MI); ++r) {
                    if (r < 2 * MI) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (r < MI + 2 && q + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __syncthreads();                                   // every wave is done with stage st
        if (st + 1 < nst) bstore();
    }
    epilogue<2, MI>(d, acc, m0 + wave * (32 * MI) + 4 * lh, lr, z1, z2);
}

template <int BM, int BN, int PREC, int WM, bool VEC, bool SEG, bool SWZ = false>
static void launch_bf(const GemmArgs& g, hipStream_t st) {
    static const int per_cu = blocks_per_cu(k_gemm_bf<BM, BN, PREC, WM, VEC, SEG, SWZ>, WM * 128);
    const int resident = per_cu * device_cus();
    const int total = g.tiles_m * g.tiles_n * g.d.batch;
    const int blocks = resident > 0 ? std::min(total, resident) : total;
    hipLaunchKernelGGL((k_gemm_bf<BM, BN, PREC, WM, VEC, SEG, SWZ>), dim3(blocks), dim3(WM * 128), 0, st, g);
}

// gemm_ring.hip: LDS-DMA ring kernels (configuration ids: see launch_ring_cfg)
void launch_ring_cfg(int cfg, const GemmArgs& g, hipStream_t st);
void launch_ring2_cfg(int cfg, const GemmArgs& g, hipStream_t st);      // gemm_ring2.hip: configurations 20..23
int ring_tile_rows(int cfg);
int ring_tile_cols(int cfg);
bool ring_cfg_ok(int cfg, const paa_gemm_desc& d);

// Ring-kernel selection (paa_gemm_config): 0 = automatic, 1 = never (the register-staged kernels of round 1),
// 2.. = force one ring configuration where its shape constraints hold (kernel A/B runs in tools/gemm_bench.py).
static int g_ring_mode = 0;
// A/B switches of the selection exist only in -DPAA_EXPERIMENTS builds (tools/step_ab.py), read from the environment at the first
// product and again by every paa_gemm_config call: PAA_NO_SQ=1 no 256-column rings, PAA_NO_R2=1 their two-stage form instead of
// the separate operand rings, PAA_K_GROUP=0 plain K order in the strided convs (gemm_env_kgroup, used by model.hip), PAA_NO_BIL=1
// planar weight planes even where the interleaved copy (B_il) is given.  The shipped library reads no environment variable here.
struct GemmEnv { bool init = false, no_sq = false, no_r2 = false, kgroup = true, no_bil = false; };
static GemmEnv g_env;
static void gemm_env_refresh() {
#ifdef PAA_EXPERIMENTS
    auto on = [](const char* name, char v) { const char* e = getenv(name); return e && e[0] == v; };
    g_env.no_sq = on("PAA_NO_SQ", '1');
    g_env.no_r2 = on("PAA_NO_R2", '1');
    g_env.kgroup = !on("PAA_K_GROUP", '0');
    g_env.no_bil = on("PAA_NO_BIL", '1');
#endif
    g_env.init = true;
}
bool gemm_env_no_bil() {
    if (!g_env.init) gemm_env_refresh();
    return g_env.no_bil;
}
bool gemm_env_kgroup() {
    if (!g_env.init) gemm_env_refresh();
    return g_env.kgroup;
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
    std::vector<double> flops, bytes;
    std::vector<int> variant;
    size_t n = 0, cap = 0;
};
static GemmProf g_prof;

paa_status gemm(const paa_gemm_desc& d, hipStream_t st) {
    if ((!d.A && !d.A_il) || !d.B || (!d.C && !d.Cb && !d.Cb_il)) PAA_FAIL(PAA_ERR_ARG, "gemm: null operand");
    if (d.A_il) {
        if (!d.precision || !d.operand_bf16 || d.a_kseg > 0 || d.a_window || (d.lda & 31) || (d.K & 31) || (d.a_s1 & 31) || (d.a_s2 & 31) || ((uintptr_t)d.A_il & 15))
            PAA_FAIL(PAA_ERR_ARG, "gemm: A_il needs split precision, plain K-contiguous bf16 rows and lda / K / batch strides that are multiples of 32");
    }
    if (d.Cb_il) {
        if (!d.precision || d.Cb || d.Cb_lo || (d.ldc & 31) || (d.c_s1 & 31) || (d.c_s2 & 31) || ((uintptr_t)d.Cb_il & 15))
            PAA_FAIL(PAA_ERR_ARG, "gemm: Cb_il needs split precision, no Cb / Cb_lo, and ldc / batch strides that are multiples of 32");
    }
    if (d.accumulate && !d.C) PAA_FAIL(PAA_ERR_ARG, "gemm: accumulate needs the f32 result");
    if (d.Cb_lo && !d.Cb) PAA_FAIL(PAA_ERR_ARG, "gemm: Cb_lo without Cb");
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.batch <= 0 || d.batch2 <= 0) PAA_FAIL(PAA_ERR_SIZE, "gemm: bad dims %d %d %d", d.M, d.N, d.K);
    const int al = d.operand_bf16 ? 7 : 3;     // elements per 16-byte vector - 1
    if ((d.lda & al) || (d.ldb & al) || (!d.A_il && ((uintptr_t)d.A & 15)) || ((uintptr_t)d.B & 15))
        PAA_FAIL(PAA_ERR_ARG, "gemm: operands must be 16-byte aligned with leading dimensions that are multiples of %d", al + 1);
    if ((d.a_s1 & al) || (d.a_s2 & al) || (d.b_s1 & al) || (d.b_s2 & al)) PAA_FAIL(PAA_ERR_ARG, "gemm: batch strides must be multiples of %d", al + 1);
    if (d.a_kseg > 0 && ((d.a_kseg & al) || (d.a_kseg_stride & al) || !d.a_kcontig)) PAA_FAIL(PAA_ERR_ARG, "gemm: bad K segmentation");
    if (d.operand_bf16) {
        if (!d.a_kcontig || !d.b_kcontig || (d.K & 7)) PAA_FAIL(PAA_ERR_ARG, "gemm: bf16 operands must be K-contiguous with K %% 8 == 0 (K=%d)", d.K);
        if (d.precision && ((!d.A_il && (!d.A_lo || ((uintptr_t)d.A_lo & 15))) || !d.B_lo || ((uintptr_t)d.B_lo & 15)))
            PAA_FAIL(PAA_ERR_ARG, "gemm: split precision needs aligned lo planes");
    }
    if (d.a_window && d.a_kseg <= 0) PAA_FAIL(PAA_ERR_ARG, "gemm: a_window needs a_kseg");
    if (d.act == PAA_ACT_GELU_GRAD && !d.aux) PAA_FAIL(PAA_ERR_ARG, "gemm: GELU_GRAD needs aux");
    if (d.res_ln_stats && (!d.residual || !d.res_ln_g || !d.res_ln_b || d.batch != 1 || d.act == PAA_ACT_GELU_GRAD ||
                           ((uintptr_t)d.res_ln_stats & 7) || ((uintptr_t)d.res_ln_g & 15) || ((uintptr_t)d.res_ln_b & 15)))
        PAA_FAIL(PAA_ERR_ARG, "gemm: res_ln_stats needs a residual, gain and bias (16-byte aligned), batch 1 and no GELU_GRAD");
    GemmArgs g;
    g.d = d;
    const bool narrow = d.N <= 64;
    const int bn = narrow ? 64 : 128;
    // 256 x 128 tile (8 waves, 2 workgroups per CU) for every large-M product; 128 x 128 (4 waves) was 4-8 % faster on
    // the isolated M = 16000, N <= 2304 shapes but made no difference inside the step (A/B on one device)
    // vector epilogue: every result / aux / residual row must split into aligned 8-column groups
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec = d.operand_bf16 && !narrow && (d.N & 7) == 0 && (d.ldc & 7) == 0 && (d.c_s1 & 7) == 0 && (d.c_s2 & 7) == 0 &&
                     al16(d.C) && al16(d.C_pre) && al16(d.Cb) && al16(d.Cb_lo) && al16(d.Cb_il) &&
                     (!d.aux || (al16(d.aux) && (d.ld_aux & (d.aux_bf16 ? 7 : 3)) == 0 && (d.aux_s1 & 7) == 0 && (d.aux_s2 & 7) == 0)) &&
                     (!d.residual || (al16(d.residual) && (d.ld_res & 3) == 0 && (d.res_s1 & 3) == 0 && (d.res_s2 & 3) == 0)) &&
                     (!d.bias || (al16(d.bias) && (d.bias_s2 & 3) == 0)) &&
                     !(d.act == PAA_ACT_GELU_GRAD && d.residual) &&              // one extra operand stream, 32-bit offsets
                     63 * d.lda < (1ll << 30) && 63 * d.ldb < (1ll << 30) && (int64_t)d.M * d.ldc < (1ll << 30) && (int64_t)d.M * std::max(d.ld_aux, d.ld_res) < (1ll << 30);
    const bool tall = vec && d.a_kseg <= 0 && (d.K & 63) == 0 && d.M >= 2048;
    // bf16 mode, large M: 256- or 192-row tiles, whichever wastes fewer workgroup slots in the last round of the
    // persistent grid (M = 16000, N = 768: 378 tiles of 256 rows leave a quarter of the 512 slots idle, 504 of 192 fill them)
    bool bm192 = false;
    if (tall && !d.precision) {
        static const int per_cu = std::max(1, blocks_per_cu(k_gemm_bf<256, 128, 0, 2, true, false>, 256));
        const int slots = per_cu * device_cus();
        const int64_t tn = cdiv(d.N, 128);
        const int64_t t256 = (int64_t)cdiv(d.M, 256) * tn * d.batch, t192 = (int64_t)cdiv(d.M, 192) * tn * d.batch;
        const double c256 = (double)((t256 + slots - 1) / slots) * 256.0, c192 = (double)((t192 + slots - 1) / slots) * 192.0 * 1.05;
        bm192 = c192 < c256;
    }
    // ring kernel: 0 = off, else the configuration id (see the dispatch below)
    int ring = 0;
    if (tall && g_ring_mode != 1 && d.N >= 128) {
        const int64_t t256 = (int64_t)cdiv(d.M, 256) * cdiv(d.N, 256) * d.batch, t128 = (int64_t)cdiv(d.M, 256) * cdiv(d.N, 128) * d.batch;
        // Measured on the step's shapes (tools/gemm_ring_bench.py, profiles/r2_gemm_ab.txt): the 192 x 128 ring kernels at
        // two workgroups per CU win wherever 256-row tiles leave the persistent grid's last round part-empty (N = 768:
        // 378 tiles on 256 slots, +15..22 % split, +5..10 % bf16; N = 2304: +3..10 %); everywhere else the ring kernels
        // only tie the register-staged ones (within 3 %), which stay the default.
        if (!g_env.init) gemm_env_refresh();
        const bool no_sq = g_env.no_sq;                  // A/B measurements: automatic selection without the 256-column rings
        if (g_ring_mode >= 2) ring = g_ring_mode;
        else if (d.precision) {
            const int64_t slots = device_cus();              // one 8-wave workgroup per CU
            const int64_t r256 = (t128 + slots - 1) / slots;                       // rounds of 256 x 128 tiles, one workgroup per CU
            const int64_t t192 = (int64_t)cdiv(d.M, 192) * cdiv(d.N, 128) * d.batch;
            const int64_t r192 = (t192 + 2 * slots - 1) / (2 * slots);             // rounds of 192 x 128 tiles, two per CU
            const int64_t rsq = (t256 + slots - 1) / slots;                        // rounds of 256 x 256 tiles, one per CU
            // cost = rounds x per-CU work of a round (in 128-column units), divided by the per-flop efficiency each
            // kernel measured against the register-staged 256 x 128 one (tools/gemm_ring_bench.py, profiles/r2_gemm_ab_sq.txt):
            // the 256 x 256 ring (20: eight waves, three A + two B slots) is 8..10 % faster per flop — conv stack, N = 3072 — and
            // fetches each A panel for 2 column tiles instead of 4; the 192 x 128 ring (7) wins where its tiles fill the
            // last round (N = 768 / 2304 at M = 16000).
            // (13 = the register-staged 192 x 128 split tile with swizzled, unpadded LDS, two workgroups per CU — whole
            // 128-byte rows per request — measures within +-3 % of ring 7 product by product and identically on the whole
            // step; kept as a selectable configuration.)
            const double c256 = (double)r256 * 256.0, c192 = (double)r192 * 192.0 * 2.0 / 0.95, csq = (double)rsq * 512.0 / 1.08;
            ring = 0;
            double best = c256;
            if (c192 < best && !d.A_il) { best = c192; ring = 7; }       // (the 192 x 128 ring reads planar A planes only)
            if (!no_sq && d.N % 256 == 0 && ring_cfg_ok(20, d) && csq < best) { best = csq; ring = 20; }
            // 192 x 256 ring (22: eight waves, one per CU): M = 16000, N = 768 is 252 tiles on 256 CUs; +3..11 % over
            // ring 7 on the N = 768 / 2304 products
            const int64_t r18 = ((int64_t)cdiv(d.M, 192) * cdiv(d.N, 256) * d.batch + slots - 1) / slots;
            const double c18 = (double)r18 * 384.0 / 1.05;
            if (!no_sq && d.N % 256 == 0 && ring_cfg_ok(22, d) && c18 < best) { best = c18; ring = 22; }
        } else if (bm192) {
            ring = 8;             // bf16: wherever 192-row tiles fill the grid better (N = 768 / 2304: +3..10 %)
            // one 192 x 256 tile per CU (23) instead of two 192 x 128 (8) where it needs no more rounds: +1..3 % (N = 768)
            const int64_t cus = device_cus();
            const int64_t r8 = ((int64_t)cdiv(d.M, 192) * cdiv(d.N, 128) * d.batch + 2 * cus - 1) / (2 * cus), r19 = ((int64_t)cdiv(d.M, 192) * cdiv(d.N, 256) * d.batch + cus - 1) / cus;
            if (!no_sq && d.N % 256 == 0 && ring_cfg_ok(23, d) && r19 <= r8) ring = 23;
        } else if (!no_sq && d.N % 256 == 0 && ring_cfg_ok(21, d)) {
            // bf16, 256 x 256 ring (21): ties the register-staged 256 x 128 kernel on time (within 3 %) where its tiles
            // fill the grid equally well, and halves the A-panel fetches from the fabric (profiles/r2_kgroup_pmc.txt)
            static const int per_cu2 = std::max(1, blocks_per_cu(k_gemm_bf<256, 128, 0, 2, true, false>, 256));
            const int64_t cus = device_cus(), slots2 = per_cu2 * cus;
            const int64_t r2 = (t128 + slots2 - 1) / slots2, rsq = (t256 + cus - 1) / cus;
            if (rsq <= r2) ring = 21;     // a round is one 256 x 256 tile or two co-resident 256 x 128 tiles per CU: equal work
        }
        // 20..23 are the separate-operand-ring kernels (gemm_ring2.hip: three A slots, two B slots): +0..5 % per product over the
        // two-stage rings 17 / 2 / 18 / 19 of gemm_ring.hip, which only -DPAA_EXPERIMENTS builds hold (PAA_NO_R2=1 selects them)
#ifdef PAA_EXPERIMENTS
        if (g_ring_mode < 2 && g_env.no_r2 && ring >= 20) {
            const int r1 = ring == 20 ? 17 : ring == 21 ? 2 : ring == 22 ? 18 : 19;
            if (ring_cfg_ok(r1, d)) ring = r1;
        }
#endif
#ifdef PAA_EXPERIMENTS
        if (ring == 13) { if (!d.precision) ring = 0; }          // 13: register-staged 192 x 128 split tile, swizzled LDS, two workgroups per CU
        else
#endif
        if (ring && !ring_cfg_ok(ring, d)) ring = 0;
        if (ring && ring < 20 && d.A_il) ring = 0;               // interleaved A: gemm_ring2.hip and the register-staged kernels
    }
    const int ring_bn = ring == 13 ? 128 : ring ? ring_tile_cols(ring) : 0, ring_bm = ring == 13 ? 192 : ring ? ring_tile_rows(ring) : 0;
    g.tiles_m = cdiv(d.M, ring ? ring_bm : tall ? (bm192 ? 192 : 256) : G_BM);
    g.tiles_n = cdiv(d.N, ring ? ring_bn : bn);
    dim3 grid(g.tiles_m * g.tiles_n, d.batch);
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    if (prof) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
        g_prof.flops[g_prof.n] = 2.0 * d.M * d.N * (double)d.K * d.batch;
        {   // algorithmic HBM bytes of this product: every operand / result byte once (overlapping conv rows counted once)
            const double es = d.operand_bf16 ? 2.0 * (d.precision ? 2 : 1) : 4.0;
            const double a_el = (d.a_kcontig && d.lda > 0 && d.lda < d.K && !d.a_window && d.a_kseg <= 0) ? ((double)(d.M - 1) * d.lda + d.K) : (double)d.M * d.K;
            const double mn = (double)d.M * d.N;
            double by = a_el * es + (double)d.N * d.K * es;
            if (d.a_window) by = (double)d.a_rows_valid * d.a_kseg * es + (double)d.N * d.K * es;
            if (d.C) by += mn * 4.0 * (d.accumulate ? 2 : 1);
            if (d.Cb) by += mn * 2.0 * (d.Cb_lo ? 2 : 1);
            if (d.Cb_il) by += mn * 4.0;
            if (d.C_pre) by += mn * (d.aux_bf16 ? 2.0 : 4.0);
            if (d.aux) by += mn * (d.aux_bf16 ? 2.0 : 4.0);
            if (d.residual) by += mn * 4.0;
            g_prof.bytes[g_prof.n] = by * d.batch;
        }
        g_prof.variant[g_prof.n] = (tall ? 32 : 0) + (d.operand_bf16 ? 16 : 0) + ((narrow || bm192) ? 8 : 0) + (d.precision ? 4 : 0) + (d.a_kcontig ? 2 : 0) + (d.b_kcontig ? 1 : 0);
    }
    // grouped positional convolution (and its dgrad): slab kernel
    if (d.operand_bf16 && d.a_window && (d.a_kseg == 48 || d.a_kseg == 64) && d.N <= 64 && d.K % (4 * d.a_kseg) == 0 &&
        d.ldb >= d.K && d.a_kseg_stride == d.lda) {
        const int taps = d.K / d.a_kseg;
        const int npl = d.precision ? 2 : 1;
        // 256-row tiles (64 rows per wave) for wav2vec2-base's 48-channel groups when the clip has more than 128 rows
        const int mi = (d.a_kseg == 48 && d.M > 128) ? 2 : 1;          // 256-row tiles: 4 waves of 64 rows (bf16) / 8 waves of 32 rows (split)
        const size_t lds = 2 * (size_t)npl * ((size_t)(128 * mi + taps - 1) * (d.a_kseg + 8) + 64 * (size_t)(4 * d.a_kseg + 8));
        if (lds <= 150 * 1024) {
            dim3 wgrid(cdiv(d.M, 128 * mi), d.batch);
#define PAA_WIN(P, KS_, MI_, WV_)                                                                                            \
            {                                                                                                                \
                static bool attr = false;                                                                                    \
                if (!attr) { PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_win<P, KS_, MI_, WV_>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr = true; } \
                hipLaunchKernelGGL((k_gemm_win<P, KS_, MI_, WV_>), wgrid, dim3(WV_ * 64), lds, st, g, taps);                 \
            }
            if (prof) { g_prof.variant[g_prof.n] = 40 + (d.precision ? 4 : 0); }
            if (d.a_kseg == 48) {
                if (mi == 2) { if (d.precision) PAA_WIN(1, 48, 1, 8) else PAA_WIN(0, 48, 2, 4) }
                else { if (d.precision) PAA_WIN(1, 48, 1, 4) else PAA_WIN(0, 48, 1, 4) }
            }
            else { if (d.precision) PAA_WIN(1, 64, 1, 4) else PAA_WIN(0, 64, 1, 4) }
#undef PAA_WIN
            if (prof) { (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st); ++g_prof.n; }
            PAA_LAUNCH_CHECK();
            return PAA_OK;
        }
    }
    if (ring) {
        // ring variants by tile: 192 x 128 -> 60 / 61 (bf16 / split), 256 x 256 -> 56 / 57, 192 x 256 -> 52 / 53, 256 x 128 -> 48 / 49
        if (prof) g_prof.variant[g_prof.n] = (ring_bm == 192 ? (ring_bn == 128 ? 60 : 52) : (ring_bn == 256 ? 56 : 48)) + (d.precision ? 1 : 0);
#ifdef PAA_EXPERIMENTS
        if (ring == 13) launch_bf<192, 128, 1, 2, true, false, true>(g, st);
        else
#endif
        if (ring >= 20) launch_ring2_cfg(ring, g, st);
        else launch_ring_cfg(ring, g, st);
    } else if (d.operand_bf16) {
        const bool seg = d.a_kseg > 0 || (d.K & 63);          // segmented / windowed A or a K tail: general loader
        if (narrow) {
            if (seg) { if (d.precision) launch_bf<128, 64, 1, 2, false, true>(g, st); else launch_bf<128, 64, 0, 2, false, true>(g, st); }
            else { if (d.precision) launch_bf<128, 64, 1, 2, false, false>(g, st); else launch_bf<128, 64, 0, 2, false, false>(g, st); }
        }
        else if (seg && vec && !d.precision) launch_bf<128, 128, 0, 2, true, true>(g, st);      // lm-head dgrad (K = vocab)
        else if (seg) { if (d.precision) launch_bf<128, 128, 1, 2, false, true>(g, st); else launch_bf<128, 128, 0, 2, false, true>(g, st); }
        else if (tall) {
            if (d.precision) launch_bf<256, 128, 1, 4, true, false>(g, st);
            else if (bm192) launch_bf<192, 128, 0, 2, true, false>(g, st);
            else launch_bf<256, 128, 0, 2, true, false>(g, st);
        }
        else if (vec) { if (d.precision) launch_bf<128, 128, 1, 2, true, false>(g, st); else launch_bf<128, 128, 0, 2, true, false>(g, st); }
        else { if (d.precision) launch_bf<128, 128, 1, 2, false, false>(g, st); else launch_bf<128, 128, 0, 2, false, false>(g, st); }
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
    g_prof.bytes.assign(max_launches, 0.0);
    g_prof.variant.assign(max_launches, 0);
    g_prof.cap = max_launches; g_prof.on = true;
    return PAA_OK;
}

// out[64][4] = per kernel variant {launches, total ms, total algorithmic FLOP, total algorithmic HBM bytes}
// (variant = tall*32 + bf16_operands*16 + (narrow | tall-with-192-rows)*8 + split*4 + a_kcontig*2 + b_kcontig; 40 / 44: the slab
// kernel of the grouped positional convolution, bf16 / split; the LDS-DMA ring kernels, bf16 / split, by tile: 60 / 61 = 192 x 128,
// 56 / 57 = 256 x 256, 52 / 53 = 192 x 256, 48 / 49 = 256 x 128 — ids the bit formula never produces).
// Synchronises on the last recorded event.  Resets the counters.
extern "C" paa_status paa_prof_read(double* out256) {
    using namespace paa;
    if (!out256) PAA_FAIL(PAA_ERR_ARG, "paa_prof_read: null");
    for (int i = 0; i < 256; ++i) out256[i] = 0.0;
    if (g_prof.n == 0) return PAA_OK;
    PAA_HIP(hipEventSynchronize(g_prof.ev[2 * g_prof.n - 1]));
    for (size_t i = 0; i < g_prof.n; ++i) {
        float ms = 0.f;
        PAA_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        const int v = g_prof.variant[i];
        out256[4 * v] += 1.0; out256[4 * v + 1] += ms; out256[4 * v + 2] += g_prof.flops[i]; out256[4 * v + 3] += g_prof.bytes[i];
    }
    g_prof.n = 0;
    return PAA_OK;
}

extern "C" paa_status paa_prof_pause(int paused) {
    paa::g_prof.on = !paused && paa::g_prof.cap > 0;
    return PAA_OK;
}

extern "C" void paa_gemm_config(int ring_mode) { paa::g_ring_mode = ring_mode; paa::gemm_env_refresh(); }

extern "C" paa_status paa_gemm(const struct paa_gemm_desc* d, void* stream) {
    if (!d) { paa::set_error("paa_gemm: null descriptor"); return PAA_ERR_ARG; }
    return paa::gemm(*d, (hipStream_t)stream);
}
