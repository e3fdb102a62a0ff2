// Descriptor of the one MFMA GEMM every conv / linear / attention product of the PGD step uses.
//   C[z][m][n] = epilogue( sum_k A[z][m][k] * B[z][k][n] )        m < M, n < N, k < K, z < batch
// MFMA operands are bf16 (precision 0) or split bf16 hi + lo (precision 1, three MFMA passes:
// hi*hi + hi*lo + lo*hi, fp32-parity), accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Two operand
// formats: operand_bf16 = 1 — A and B are bf16 planes in HBM (hi, plus lo planes in split mode), both
// K-contiguous: the fast path every conv / linear product takes (its producers write the bf16 planes
// in their epilogues); operand_bf16 = 0 — f32 operands converted on their way into LDS, with the
// transposed layouts the materialised-attention products need.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PAA_ACT_NONE = 0, PAA_ACT_GELU = 1, PAA_ACT_GELU_GRAD = 2 };

typedef struct paa_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    int32_t M, N, K;
    // A element (m, k):  a_kcontig ? A[m * lda + koff(k)] : A[k * lda + m]
    //   koff(k) = (k / a_kseg) * a_kseg_stride + k % a_kseg      (a_kseg = 0 means one segment: koff(k) = k)
    //   a_window != 0 (pos-conv): segment js = k / a_kseg reads row (m + js - a_pad) of a clip of a_rows_valid
    //   rows, zero outside [0, a_rows_valid):  A[(m + js - a_pad) * lda + k % a_kseg]
    int64_t lda;
    int32_t a_kcontig, a_kseg;
    int64_t a_kseg_stride;
    int32_t a_window, a_pad, a_rows_valid;
    // B element (k, n):  b_kcontig ? B[n * ldb + k] : B[k * ldb + n]
    int64_t ldb;
    int32_t b_kcontig;
    int64_t ldc;
    // batch index z -> (z1, z2) = (z / batch2, z % batch2); operand offset = z1 * s1 + z2 * s2 (elements)
    int32_t batch, batch2;
    int64_t a_s1, a_s2, b_s1, b_s2, c_s1, c_s2;
    // epilogue, in this order: v = acc * alpha; v += bias[z2 * bias_s2 + n];
    //   act GELU: C_pre[m,n] = v (if C_pre), v = gelu(v);  act GELU_GRAD: v *= gelu'(aux[m,n]);
    //   v += residual[m,n]; rows with (m % row_period) >= row_valid are forced to 0 (row_period > 0);
    //   C = accumulate ? C + v : v
    float alpha;
    const float* bias;
    int64_t bias_s2;
    int32_t act;
    float* C_pre;
    const float* aux;
    int64_t ld_aux, aux_s1, aux_s2;
    const float* residual;
    int64_t ld_res, res_s1, res_s2;
    int32_t row_period, row_valid;
    int32_t accumulate;
    int32_t precision;   // 0 = bf16, 1 = split bf16 (fp32-parity)
    // bf16 operand / result planes (uint16 bit patterns).  operand_bf16: A, B (and A_lo, B_lo when precision = 1)
    // point to bf16; strides stay in elements; lda, ldb, a_kseg, batch strides must be multiples of 8, K of 8.
    int32_t operand_bf16;
    const void* A_lo;
    const void* B_lo;
    // optional bf16 copies of the result (same ldc / batch strides as C): Cb = bf16(v), Cb_lo = bf16(v - Cb).
    // C itself may be null when only the bf16 result is wanted.
    void* Cb;
    void* Cb_lo;
    // aux_bf16 != 0: C_pre and aux are bf16 arrays (uint16 bit patterns, same leading dimensions / strides in
    // elements): the pre-activation a GELU product keeps for its backward pass is stored at half the bytes (bf16 mode).
    int32_t aux_bf16;
    // aux_gate != 0: the array a GELU product keeps for its backward pass holds gelu'(v) instead of v — act GELU writes
    // C_pre[m,n] = gelu'(v), act GELU_GRAD multiplies by aux[m,n] as it stands.  The derivative is evaluated once, next
    // to the GELU that shares its exp, instead of again in every backward epilogue (bf16 mode, together with aux_bf16).
    int32_t aux_gate;
    // k_group > 0 (operand_bf16 products whose A rows are OVERLAPPING windows: strided convolutions, K = taps * k_group,
    // lda < K): the K slabs are walked channel-slab-major, tap-minor — slab (c, tap) = elements tap * k_group + c * BK ..
    // — instead of 0 .. K in order.  Consecutive output rows share input frames between taps (row r's tap 2 is row
    // r + 1's tap 0 at stride 2); in K order those two uses of one cache line are k_group / BK slabs apart and the line has
    // left L2 by then, tap-minor they are adjacent slabs.  Same products, different f32 summation order; ignored when
    // k_group is not a multiple of the kernel's K slab.
    int32_t k_group;
    // B_il != NULL (precision 1 only, optional): the SAME weights as B / B_lo with the two planes interleaved per 32-element K
    // group — row n is [hi k 0..31 | lo k 0..31 | hi k 32..63 | lo k 32..63 | ...], 2 * ldb elements per row — so that a K slab of a
    // row is one full 128-byte line instead of one 64-byte segment in each plane (half the L2 requests).  Kernels that do not
    // take this layout read B / B_lo as before; both must be given.
    const void* B_il;
    // A_il != NULL (precision 1, operand_bf16, plain K-contiguous rows: no a_kseg / a_window): the A operand's two planes live in ONE
    // array interleaved per 32-element K group — element (m, k): hi at A_il[m * 2 lda + (k / 32) * 64 + k % 32], lo 32 elements
    // further; batch offsets are 2 (z1 a_s1 + z2 a_s2).  lda, a_s1, a_s2 and K must be multiples of 32 (lda stays the row stride in
    // ELEMENTS of the logical matrix; overlapping conv windows work as before, a window of k taps being 2 k C consecutive array
    // elements).  A and A_lo are ignored.  A K slab of a row is then one full 128-byte line instead of a 64-byte segment in each of two
    // planes: half the requests on the operand that streams from HBM.  The producer of such a tensor writes it with Cb_il (or
    // paa::Bf::il in the element-wise kernels).
    const void* A_il;
    // Cb_il != NULL (precision 1): the bf16 result goes to ONE interleaved array instead of Cb / Cb_lo (which must be NULL):
    // hi of (m, n) at Cb_il[2 (z1 c_s1 + z2 c_s2) + m * 2 ldc + (n / 32) * 64 + n % 32], lo 32 further; ldc, c_s1, c_s2 multiples of 32.
    void* Cb_il;
    // res_ln_stats != NULL (batch = 1 products with a residual): the residual is the LayerNorm of the array `residual` points at,
    // evaluated on the fly — r[m, n] = (residual[m, n] - mean_m) * rstd_m * res_ln_g[n] + res_ln_b[n] with (mean_m, rstd_m) =
    // res_ln_stats[2 m], [2 m + 1] as k_ln_fwd left them.  The post-LN encoder adds LN(x) to its attention / FFN outputs; with
    // this mode the LayerNorm kernel writes only the bf16 planes the next GEMM reads, not an f32 copy for the residual (-4 of the 12
    // bytes per element it moves).  Same arithmetic as the LayerNorm kernel.
    const float* res_ln_stats;
    const float* res_ln_g;
    const float* res_ln_b;
} paa_gemm_desc;

#ifdef __cplusplus
}
#endif
