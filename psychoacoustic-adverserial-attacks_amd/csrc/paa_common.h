// Shared host/device helpers for libpaa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/paa_hip.h"

namespace paa {

void set_error(const std::string& msg);

#define PAA_FAIL(code, ...)                                  \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof(_b), __VA_ARGS__);               \
        ::paa::set_error(_b);                                \
        return (code);                                       \
    } while (0)

#define PAA_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) PAA_FAIL(PAA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define PAA_LAUNCH_CHECK()                                                                            \
    do {                                                                                              \
        hipError_t _e = hipGetLastError();                                                            \
        if (_e != hipSuccess) PAA_FAIL(PAA_ERR_HIP, "kernel launch (%s:%d): %s", __FILE__, __LINE__, \
                                       hipGetErrorString(_e));                                        \
    } while (0)

#define PAA_TRY(expr)                      \
    do {                                   \
        paa_status _s = (expr);            \
        if (_s != PAA_OK) return _s;       \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Compute units of the CURRENT device, cached per device id (a process may drive several devices, and a partitioned part
// reports fewer CUs than the 256 of a whole MI355X).
inline int device_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cache[dev]) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cache[dev] = c;
    }
    return cache[dev];
}

// ---- wave (64 lanes) / block reductions --------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
// `sm` must hold NT/64 elements of T; the caller must not reuse it before the next barrier.
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    T r = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sm[i];
    __syncthreads();
    return r;
}

// bf16 planes: hi = bf16(v) and, in split (fp32-parity) mode, lo = bf16(v - hi).  Either pointer may be null.
// il (split mode only): ONE array of 2 n elements at `hi` holds both planes interleaved per 32-element group,
// [32 hi | 32 lo | 32 hi | ...] (gemm.h, A_il / Cb_il): element i lives at il_index(i), its lo part 32 further.  Rows of these
// tensors are multiples of 32 elements, so the map needs no row length.  `lo` is not used then.
struct Bf {
    unsigned short* hi;
    unsigned short* lo;
    bool il = false;
};
__host__ __device__ __forceinline__ size_t il_index(size_t i) { return ((i >> 5) << 6) + (i & 31); }
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
// two values -> one dword of two bf16 (a in the low half): ONE v_cvt_pk_bf16_f32 on gfx950 (round to nearest even, as bf16_bits)
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{a, b}, b2_t));
}
__device__ __forceinline__ void store_bf16(const Bf& o, size_t i, float v) {
    if (o.hi) {
        const unsigned short h = bf16_bits(v);
        if (o.il) {
            const size_t j = il_index(i);
            o.hi[j] = h;
            o.hi[j + 32] = bf16_bits(v - bf16_to_f32(h));
            return;
        }
        o.hi[i] = h;
        if (o.lo) o.lo[i] = bf16_bits(v - bf16_to_f32(h));
    }
}

// four consecutive values -> bf16 planes, one 8-byte store per plane (i multiple of 4)
__device__ __forceinline__ void store_bf16x4(const Bf& o, size_t i, const float (&v)[4]) {
    if (!o.hi) return;
    unsigned h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = bf16_bits(v[j]);
    unsigned short* ph = o.il ? o.hi + il_index(i) : o.hi + i;
    unsigned short* pl = o.il ? ph + 32 : (o.lo ? o.lo + i : nullptr);
    *reinterpret_cast<uint2*>(ph) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    if (pl) {
        unsigned l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) l[j] = bf16_bits(v[j] - __uint_as_float(h[j] << 16));
        *reinterpret_cast<uint2*>(pl) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
}

// exact (erf) GELU and its derivative, as torch.nn.functional.gelu(approximate='none')
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Branch-free GELU for the bf16-mode GEMM epilogues: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// bf16 rounding of the operands it feeds), one v_exp_f32 + one v_rcp_f32, ~20 instructions instead of libm's erff.
// e = exp(-x^2 / 2) is shared between erf(x / sqrt 2) and the normal pdf of the derivative.
__device__ __forceinline__ float gelu_cdf_fast(float x, float& e) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);     // raw v_exp_f32: the argument is <= 0, underflow -> 0
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    return 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_fast(float x) { float e; return x * gelu_cdf_fast(x, e); }
// gelu(x) and gelu'(x) together (one exp, one rcp)
__device__ __forceinline__ float gelu_both_fast(float x, float& dg) {
    float e;
    const float cdf = gelu_cdf_fast(x, e);
    dg = cdf + x * 0.39894228040143268f * e;
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float e;
    const float cdf = gelu_cdf_fast(x, e);
    return cdf + x * 0.39894228040143268f * e;
}

// two values at once (v_pk_fma_f32 / v_pk_mul_f32 carry the polynomial; exp and rcp stay scalar): same arithmetic as
// gelu_both_fast per component
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_both_fast2(f32x2 x, f32x2& dg) {
    f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    ax = ax * 0.70710678118654752f;
    const f32x2 q = -1.44269504088896341f * ax * ax;
    const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const f32x2 den = 1.0f + 0.3275911f * ax;
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x2 ea = 1.0f - poly * e;
    const f32x2 er = {copysignf(ea.x, x.x), copysignf(ea.y, x.y)};
    const f32x2 cdf = 0.5f * (1.0f + er);
    dg = cdf + x * 0.39894228040143268f * e;
    return x * cdf;
}

}  // namespace paa
