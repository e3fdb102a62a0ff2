// conv0 GroupNorm backward: ONE pass over the gradient dy (B, P, C) for both things the backward needs (both precision
// modes since round 2: PREC = 1 carries dy and W1 as hi + lo bf16 planes and every product as three MFMA passes).
//
// The gradient wrt the conv0 GroupNorm output is the largest tensor of the backward pass (32 x 31999 x 512 bf16 = 1 GB
// at the benchmark size) and it used to be read twice: by the channel statistics kernel (s1 = mean_t dy,
// s2 = mean_t dy * xhat) and by the GEMM G1 = dy . W1_b^T.  Both reads were HBM bound.  With one input channel
//     xhat[t, c] = rstd_c (sum_j w[c, j] x[t s + j] - mean_c)
// so both sums follow from  S[c, j] = sum_t dy[t, c] X[t, j],  X[t, :] = (x[t s], ..., x[t s + 9], 1):
//     s1_c = S[c, 10] / T,     s2_c = rstd_c (sum_j w[c, j] S[c, j] - mean_c S[c, 10]) / T.
// S = dy^T X is a (C x 11) product with K = T; G1 = dy W1^T is (T x 10) with K = C.  A workgroup stages a 32-frame
// tile of dy in LDS once and feeds it to v_mfma_f32_32x32x16_bf16 twice: row-major fragments for G1, transposing LDS
// reads (ds_read_b64_tr_b16) for S.  X rides as hi + lo bf16 planes (the waveform keeps ~16 mantissa bits).
// Tiles are prefetched through registers; S stays in accumulators over the workgroup's whole frame range and leaves as
// one partial per workgroup (summed in a fixed order by k_conv0_s_finalize => reproducible).
//
// Reference: the autograd of Wav2Vec2GroupNormConvLayer (HF modeling_wav2vec2.py, conv -> GroupNorm(C, C) -> GELU) as
// reached from core/train.py:137-140 (loss.backward() to the perturbation).
#include <stdlib.h>

#include "model_kernels.h"

namespace paa {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DG_T = 32;                 // frames per tile
constexpr int DG_CMAX = 512;             // channels (multiple of 32)
constexpr int DG_RS = DG_CMAX + 8;       // LDS row stride of the tile in bf16 (16-byte aligned rows, 4-bank skew)

int conv0_dgrad_blocks(int B, int T) {
    const int ntiles = cdiv(T, DG_T);
    return std::max(1, std::min(std::min(512 / std::max(B, 1), 64), ntiles));
}

template <int PREC>
__global__ __launch_bounds__(256, 2) void k_conv0_dgrad(Conv0Args a) {
    constexpr int NPL = PREC ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short tile[NPL * DG_T * DG_RS];          // hi rows, then lo rows
    __shared__ __attribute__((aligned(16))) float red[4][DG_T * 16];
    __shared__ float xs[(DG_T - 1) * 5 + 10 + 1];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, nblk = gridDim.x;
    const int C = a.C, nks = C >> 4, ncb = C >> 5, cpr = C >> 3;       // k-steps of G1, 32-channel blocks, 16-byte chunks per row
    const int nchunks = DG_T * cpr;
    const int ntiles = (a.T + DG_T - 1) / DG_T;
    const unsigned short* __restrict__ dy = a.dpreb.hi + (size_t)b * a.P * C;
    const unsigned short* __restrict__ dyl = PREC ? a.dpreb.lo + (size_t)b * a.P * C : nullptr;
    constexpr int LO = DG_T * DG_RS;           // hi -> lo tile distance in LDS

    // W1_b fragments of this wave's k-steps (k-step ks = w + 4 i): column j = lr, elements 16 ks + 8 lh ..
    // (PREC: both planes' fragments are re-read per tile — 16 L1-resident 16-byte loads — rather than held: holding them
    // next to the 64 accumulators and the two register-staged tiles does not fit 256 VGPRs)
    bf16x8 wf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ks = w + 4 * i;
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!PREC && ks < nks && lr < 16) z = *reinterpret_cast<const bf16x8*>(a.w1b.hi + ((size_t)b * 16 + lr) * C + 16 * ks + 8 * lh);
        wf[i] = z;
    }
    f32x16 sacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[i][e] = 0.f;

    // (PREC: the hi plane of the next tile is requested before the G1 phase, the lo plane after it — both planes next to
    // the G1 phase's fragments do not fit 256 VGPRs; the lo loads have the S phase to land)
    uint4 st[8], stl[PREC ? 8 : 1];
    auto fetch = [&](int t0) {            // tile rows -> registers (rows >= T read as zero: pad rows may hold anything)
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int idx = tid + 256 * v;
            const int row = idx / cpr, col = idx - row * cpr;
            uint4 z = make_uint4(0u, 0u, 0u, 0u);
            if (idx < nchunks && t0 + row < a.T) z = *reinterpret_cast<const uint4*>(dy + (size_t)(t0 + row) * C + 8 * col);
            st[v] = z;
        }
    };
    auto fetch_lo = [&](int t0) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int idx = tid + 256 * v;
            const int row = idx / cpr, col = idx - row * cpr;
            uint4 z = make_uint4(0u, 0u, 0u, 0u);
            if (idx < nchunks && t0 + row < a.T) z = *reinterpret_cast<const uint4*>(dyl + (size_t)(t0 + row) * C + 8 * col);
            stl[v] = z;
        }
    };
    auto stage = [&](int t0) {            // registers -> LDS tile, input window -> xs
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int idx = tid + 256 * v;
            const int row = idx / cpr, col = idx - row * cpr;
            if (idx < nchunks) {
                *reinterpret_cast<uint4*>(tile + row * DG_RS + 8 * col) = st[v];
                if (PREC) *reinterpret_cast<uint4*>(tile + LO + row * DG_RS + 8 * col) = stl[v];
            }
        }
        if (tid < (DG_T - 1) * 5 + 10) {
            const int i = t0 * 5 + tid;
            xs[tid] = i < a.L ? in_sample(a, b, i) : 0.f;
        }
    };

    int t0 = blockIdx.x * DG_T;
    if (blockIdx.x < ntiles) { fetch(t0); if (PREC) fetch_lo(t0); stage(t0); }
    __syncthreads();
    for (int tl = blockIdx.x; tl < ntiles; tl += nblk) {
        t0 = tl * DG_T;
        const int tn = (tl + nblk) * DG_T;
        const bool more = tl + nblk < ntiles;
        if (more) fetch(tn);
        // ---- G1 partial of this wave's k-steps: rows = frames, columns = taps
        f32x16 g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
        const unsigned short* wlo = PREC ? a.w1b.lo + ((size_t)b * 16 + (lr & 15)) * C + 8 * lh : nullptr;
        const unsigned short* whi = PREC ? a.w1b.hi + ((size_t)b * 16 + (lr & 15)) * C + 8 * lh : nullptr;
        if (PREC) { asm volatile("" : "+v"(wlo)); asm volatile("" : "+v"(whi)); }    // opaque per tile: keeps these loads inside the loop
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ks = w + 4 * i;
            if (ks < nks) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(tile + lr * DG_RS + 16 * ks + 8 * lh);
                if (PREC) {
                    const bf16x8 afl = *reinterpret_cast<const bf16x8*>(tile + LO + lr * DG_RS + 16 * ks + 8 * lh);
                    bf16x8 wl = *reinterpret_cast<const bf16x8*>(wlo + 16 * ks), wh = *reinterpret_cast<const bf16x8*>(whi + 16 * ks);
                    if (lr >= 16) { wl = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; wh = wl; }
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl, wh, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wl, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wh, g, 0, 0, 0);
                } else {
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[i], g, 0, 0, 0);
                }
            }
        }
        if (PREC) {
            __builtin_amdgcn_sched_barrier(0);        // keep the lo requests (32 registers) below the G1 phase
            if (more) fetch_lo(tn);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- X fragments (column j = lr; k order of the transposing read: frames 16 s + 4 lh + (0..3), then + 8)
        bf16x8 xh[2], xl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int t = 16 * s + 4 * lh + (e & 3) + 8 * (e >> 2);
                float v = 0.f;
                if (t0 + t < a.T) v = lr < 10 ? xs[t * 5 + (lr < 10 ? lr : 0)] : (lr == 10 ? 1.f : 0.f);
                const unsigned short h = bf16_bits(v);
                xh[s][e] = (short)h;
                xl[s][e] = (short)bf16_bits(v - bf16_to_f32(h));
            }
        // ---- S += dy^T X on this wave's channel blocks (block cb = w + 4 i)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cb = w + 4 * i;
            if (cb < ncb) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int q = lr & 15;
                    const unsigned short* p = tile + (16 * s + 4 * lh + (q >> 2)) * DG_RS + cb * 32 + (lr & 16) + 4 * (q & 3);
                    typedef __attribute__((address_space(3))) bf16x4* lds4;
                    const bf16x4 u0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p));
                    const bf16x4 u1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + 8 * DG_RS));
                    bf16x8 af;
                    af[0] = u0[0]; af[1] = u0[1]; af[2] = u0[2]; af[3] = u0[3]; af[4] = u1[0]; af[5] = u1[1]; af[6] = u1[2]; af[7] = u1[3];
                    if (PREC) {
                        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + LO));
                        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + LO + 8 * DG_RS));
                        bf16x8 afl;
                        afl[0] = v0[0]; afl[1] = v0[1]; afl[2] = v0[2]; afl[3] = v0[3]; afl[4] = v1[0]; afl[5] = v1[1]; afl[6] = v1[2]; afl[7] = v1[3];
                        sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl, xh[s], sacc[i], 0, 0, 0);
                    }
                    sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xh[s], sacc[i], 0, 0, 0);
                    sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xl[s], sacc[i], 0, 0, 0);
                }
            }
        }
        if (lr < 16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[w][((e & 3) + 8 * (e >> 2) + 4 * lh) * 16 + lr] = g[e];
        }
        __syncthreads();
        {   // G1 rows of the tile = sum of the four waves' partials (fixed order)
            const int idx = 2 * tid, t = idx >> 4;
            const float2 r0 = *reinterpret_cast<const float2*>(&red[0][idx]), r1 = *reinterpret_cast<const float2*>(&red[1][idx]);
            const float2 r2 = *reinterpret_cast<const float2*>(&red[2][idx]), r3 = *reinterpret_cast<const float2*>(&red[3][idx]);
            if (t0 + t < a.P)
                *reinterpret_cast<float2*>(a.G1 + ((size_t)b * a.P + t0) * 16 + idx) =
                    make_float2((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y));
        }
        if (more) stage(tn);
        __syncthreads();
    }
    // S partial of this workgroup: part[b][blk][c][16]
    float* __restrict__ sp = a.part + ((size_t)b * nblk + blockIdx.x) * C * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cb = w + 4 * i;
        if (cb < ncb && lr < 16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sp[(size_t)(cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 16 + lr] = sacc[i][e];
        }
    }
}

// ---- the same pass with the tile fetched by LDS-DMA (C = 512: one 1 KB row per wave-instruction) ---------------------------------
// The register-staged form above holds the next tile in 32 (bf16) / 64 (split) staging registers per lane: next to the 64 S
// accumulators that pins it to two workgroups per CU with ONE tile of prefetch each, and the split form spills.  It reads dy at
// 3.9 / 2.8 TB/s (bf16 / split) — the only HBM-bound kernel of the step well below what the part sustains.  Here the rows go global ->
// LDS by global_load_lds_dwordx4 into a double-buffered tile (2 x 65 KB in split mode: one workgroup per CU), the staging registers
// are gone (no spills; the W1 fragments of both planes stay in registers instead of being re-read per tile) and the next tile is
// in flight during the whole compute phase of the current one.  Same products as k_conv0_dgrad; the f32 partial sums are grouped
// differently (eight waves share the k-steps, half as many workgroups per clip), so results agree to rounding, not bit for bit
// (tests/test_gpu_model.py::test_conv0_backward_single_pass_equals_two_pass checks it against the two-pass path: ~1e-7).
typedef __attribute__((address_space(1))) const void* dg_gas;
typedef __attribute__((address_space(3))) void* dg_las;

// NWV waves share the tile: k-steps of G1 and channel blocks of S are dealt round-robin (NWV = 8: two instruction streams per SIMD).
template <int PREC, int NWV>
__global__ __launch_bounds__(NWV * 64, 1) void k_conv0_dgrad_dma(Conv0Args a) {
    constexpr int NPL = PREC ? 2 : 1, NT = NWV * 64, NKI = 32 / NWV, NCI = 16 / NWV;      // k-steps / channel blocks per wave at C = 512
    constexpr int C = DG_CMAX, nks = C >> 4, ncb = C >> 5;
    constexpr int LO = DG_T * DG_RS;                       // hi -> lo rows inside a tile buffer (bf16 elements)
    constexpr int TSZ = NPL * DG_T * DG_RS;                // one tile buffer
    constexpr int NXS = (DG_T - 1) * 5 + 10 + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];
    unsigned short* tiles = dsm;                                                   // [2][TSZ]
    float* red = reinterpret_cast<float*>(dsm + 2 * TSZ);                          // [NWV][DG_T * 16]
    float* xsb = red + NWV * DG_T * 16;                                              // [2][NXS]
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, nblk = gridDim.x;
    const int ntiles = (a.T + DG_T - 1) / DG_T;
    const unsigned short* __restrict__ dy = a.dpreb.hi + (size_t)b * a.P * C;
    const unsigned short* __restrict__ dyl = PREC ? a.dpreb.lo + (size_t)b * a.P * C : nullptr;

    // W1_b fragments of this wave's k-steps (k-step ks = w + NWV i): column j = lr (taps 0..9, rows 10..15 of W1_b are zero), both planes
    bf16x8 wf[NKI], wfl[PREC ? NKI : 1];
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
        const int ks = w + NWV * i;
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}, zl = z;
        if (lr < 16) {
            z = *reinterpret_cast<const bf16x8*>(a.w1b.hi + ((size_t)b * 16 + lr) * C + 16 * ks + 8 * lh);
            if (PREC) zl = *reinterpret_cast<const bf16x8*>(a.w1b.lo + ((size_t)b * 16 + lr) * C + 16 * ks + 8 * lh);
        }
        wf[i] = z;
        if (PREC) wfl[i] = zl;
    }
    f32x16 sacc[NCI];
#pragma unroll
    for (int i = 0; i < NCI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[i][e] = 0.f;

    auto issue = [&](int t0, int buf) {       // rows of the tile -> LDS (rows past T re-read row T - 1 and are zeroed after landing)
#pragma unroll
        for (int i = 0; i < NPL * DG_T / NWV; ++i) {
            const int r = w + NWV * i;
            const int pl = r / DG_T, row = r - pl * DG_T;
            const int tr = min(t0 + row, a.T - 1);
            const unsigned short* src = (pl ? dyl : dy) + (size_t)tr * C + lane * 8;
            __builtin_amdgcn_global_load_lds((dg_gas)src, (dg_las)(tiles + buf * TSZ + pl * LO + row * DG_RS), 16, 0, 0);
        }
        if (tid < NXS - 1) {
            const int i = t0 * 5 + tid;
            xsb[buf * NXS + tid] = i < a.L ? in_sample(a, b, i) : 0.f;
        }
    };

    int cur = 0;
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x * DG_T, 0);
    for (int tl = blockIdx.x; tl < ntiles; tl += nblk) {
        const int t0 = tl * DG_T;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // tile `cur` (and its input window) has landed everywhere
        unsigned short* tile = tiles + cur * TSZ;
        const float* xs = xsb + cur * NXS;
        if (t0 + DG_T > a.T) {                             // last tile of the clip: rows past T are zero
            const int r0 = a.T - t0;
            for (int i = tid; i < (DG_T - r0) * (C / 8) * NPL; i += NT) {
                const int pl = i / ((DG_T - r0) * (C / 8)), j = i - pl * (DG_T - r0) * (C / 8);
                const int row = r0 + j / (C / 8), col = j % (C / 8);
                *reinterpret_cast<uint4*>(tile + pl * LO + row * DG_RS + 8 * col) = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
        }
        if (tl + nblk < ntiles) issue((tl + nblk) * DG_T, cur ^ 1);     // the other buffer was last read before the barrier above
        // ---- G1 partial of this wave's k-steps: rows = frames, columns = taps
        f32x16 g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NKI; ++i) {
            const int ks = w + NWV * i;
            if (ks < nks) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(tile + lr * DG_RS + 16 * ks + 8 * lh);
                if (PREC) {
                    const bf16x8 afl = *reinterpret_cast<const bf16x8*>(tile + LO + lr * DG_RS + 16 * ks + 8 * lh);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl, wf[i], g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wfl[i], g, 0, 0, 0);
                }
                g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[i], g, 0, 0, 0);
            }
        }
        // ---- X fragments (column j = lr; k order of the transposing read: frames 16 s + 4 lh + (0..3), then + 8)
        bf16x8 xh[2], xl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int t = 16 * s + 4 * lh + (e & 3) + 8 * (e >> 2);
                float v = 0.f;
                if (t0 + t < a.T) v = lr < 10 ? xs[t * 5 + (lr < 10 ? lr : 0)] : (lr == 10 ? 1.f : 0.f);
                const unsigned short h = bf16_bits(v);
                xh[s][e] = (short)h;
                xl[s][e] = (short)bf16_bits(v - bf16_to_f32(h));
            }
        // ---- S += dy^T X on this wave's channel blocks (block cb = w + 4 i)
#pragma unroll
        for (int i = 0; i < NCI; ++i) {
            const int cb = w + NWV * i;
            if (cb < ncb) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int q = lr & 15;
                    const unsigned short* p = tile + (16 * s + 4 * lh + (q >> 2)) * DG_RS + cb * 32 + (lr & 16) + 4 * (q & 3);
                    typedef __attribute__((address_space(3))) bf16x4* lds4;
                    const bf16x4 u0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p));
                    const bf16x4 u1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + 8 * DG_RS));
                    bf16x8 af;
                    af[0] = u0[0]; af[1] = u0[1]; af[2] = u0[2]; af[3] = u0[3]; af[4] = u1[0]; af[5] = u1[1]; af[6] = u1[2]; af[7] = u1[3];
                    if (PREC) {
                        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + LO));
                        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + LO + 8 * DG_RS));
                        bf16x8 afl;
                        afl[0] = v0[0]; afl[1] = v0[1]; afl[2] = v0[2]; afl[3] = v0[3]; afl[4] = v1[0]; afl[5] = v1[1]; afl[6] = v1[2]; afl[7] = v1[3];
                        sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl, xh[s], sacc[i], 0, 0, 0);
                    }
                    sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xh[s], sacc[i], 0, 0, 0);
                    sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xl[s], sacc[i], 0, 0, 0);
                }
            }
        }
        if (lr < 16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[w * (DG_T * 16) + ((e & 3) + 8 * (e >> 2) + 4 * lh) * 16 + lr] = g[e];
        }
        __syncthreads();
        if (tid < 256) {   // G1 rows of the tile = sum of the waves' partials in a fixed order
            const int idx = 2 * tid, t = idx >> 4;
            float2 acc2 = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < NWV; q += 4) {       // k-steps were dealt round-robin: wave q holds steps q, q + NWV, ...
                const float2 r0 = *reinterpret_cast<const float2*>(red + (q + 0) * DG_T * 16 + idx), r1 = *reinterpret_cast<const float2*>(red + (q + 1) * DG_T * 16 + idx);
                const float2 r2 = *reinterpret_cast<const float2*>(red + (q + 2) * DG_T * 16 + idx), r3 = *reinterpret_cast<const float2*>(red + (q + 3) * DG_T * 16 + idx);
                acc2.x += (r0.x + r1.x) + (r2.x + r3.x);
                acc2.y += (r0.y + r1.y) + (r2.y + r3.y);
            }
            if (t0 + t < a.P) *reinterpret_cast<float2*>(a.G1 + ((size_t)b * a.P + t0) * 16 + idx) = acc2;
        }
        cur ^= 1;            // (the next iteration's first barrier also orders these reads of `red` before its next writes)
    }
    // S partial of this workgroup: part[b][blk][c][16]
    float* __restrict__ sp = a.part + ((size_t)b * nblk + blockIdx.x) * C * 16;
#pragma unroll
    for (int i = 0; i < NCI; ++i) {
        const int cb = w + NWV * i;
        if (cb < ncb && lr < 16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sp[(size_t)(cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 16 + lr] = sacc[i][e];
        }
    }
}

// (s1 / n, s2 / n) per (clip, channel) from the workgroups' S partials, summed in f64 in a fixed order
__global__ __launch_bounds__(256) void k_conv0_s_finalize(Conv0Args a, int nblk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.B * a.C) return;
    const int b = i / a.C, c = i - b * a.C;
    double s[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) s[j] = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const float4* p = reinterpret_cast<const float4*>(a.part + (((size_t)b * nblk + k) * a.C + c) * 16);
        const float4 p0 = p[0], p1 = p[1], p2 = p[2];
        s[0] += p0.x; s[1] += p0.y; s[2] += p0.z; s[3] += p0.w;
        s[4] += p1.x; s[5] += p1.y; s[6] += p1.z; s[7] += p1.w;
        s[8] += p2.x; s[9] += p2.y; s[10] += p2.z;
    }
    const double mean = a.gn_stats[2 * (size_t)i], rstd = a.gn_stats[2 * (size_t)i + 1];
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j) dot += (double)a.w[c * 10 + j] * s[j];
    a.gn_bsums[2 * (size_t)i] = (float)(s[10] / a.T);
    a.gn_bsums[2 * (size_t)i + 1] = (float)(rstd * (dot - mean * s[10]) / a.T);
}

bool conv0_dgrad_supported(const Conv0Args& a) {
    return (a.dpreb.lo != nullptr) == (a.w1b.lo != nullptr) && a.stride == 5 && a.k == 10 && (a.C & 31) == 0 && a.C <= DG_CMAX && a.C >= 32;
}

int64_t conv0_dgrad_part_floats(int B, int T, int C) { return (int64_t)B * conv0_dgrad_blocks(B, T) * C * 16; }

// G1 (B, P, 16) and gn_bsums from dpreb / w1b / gn_stats; `part` holds conv0_dgrad_part_floats floats
paa_status conv0_dgrad_fused(const Conv0Args& a, float* part, hipStream_t st) {
    if (!conv0_dgrad_supported(a)) PAA_FAIL(PAA_ERR_ARG, "conv0 dgrad: unsupported shape (C %d, k %d, stride %d)", a.C, a.k, a.stride);
    Conv0Args b = a;
    b.part = part;
    int nblk = conv0_dgrad_blocks(a.B, a.T);
    // (split mode only: with one plane the register-staged form keeps two workgroups per CU without spilling and measures 272 us
    // against 294 for the DMA form)
    bool dma = a.C == DG_CMAX && a.dpreb.lo != nullptr;
#ifdef PAA_EXPERIMENTS      // tools/model_ab.py: the register-staged form
    { const char* e = getenv("PAA_NO_C0DMA"); if (e && e[0] == '1') dma = false; }
#endif
    if (dma) {
        // LDS-DMA form: one workgroup per CU (double-buffered tile), so half as many workgroups per clip cover the chip
        const int ntiles = cdiv(a.T, DG_T);
        nblk = std::max(1, std::min(std::min(device_cus() / std::max(a.B, 1), 64), ntiles));
        const int npl = a.dpreb.lo ? 2 : 1;
        constexpr int NWV = 8;
        const size_t lds = 2 * (size_t)npl * DG_T * DG_RS * 2 + NWV * DG_T * 16 * 4 + 2 * ((DG_T - 1) * 5 + 10 + 1) * 4;
        static bool attr = false;
        if (!attr) {
            PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv0_dgrad_dma<1, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            PAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv0_dgrad_dma<0, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            attr = true;
        }
        if (a.dpreb.lo) hipLaunchKernelGGL((k_conv0_dgrad_dma<1, NWV>), dim3(nblk, a.B), dim3(NWV * 64), lds, st, b);
        else hipLaunchKernelGGL((k_conv0_dgrad_dma<0, NWV>), dim3(nblk, a.B), dim3(NWV * 64), lds, st, b);
    } else if (a.dpreb.lo) {
        hipLaunchKernelGGL(k_conv0_dgrad<1>, dim3(nblk, a.B), dim3(256), 0, st, b);
    } else {
        hipLaunchKernelGGL(k_conv0_dgrad<0>, dim3(nblk, a.B), dim3(256), 0, st, b);
    }
    PAA_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_conv0_s_finalize, dim3(cdiv((int64_t)a.B * a.C, 256)), dim3(256), 0, st, b, nblk);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

}  // namespace paa
