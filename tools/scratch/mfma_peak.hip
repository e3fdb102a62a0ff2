// Probe: sustained v_mfma_f32_32x32x16_bf16 rate on the whole chip (a) alone, (b) with ds_read_b128 fragment reads at the
// GEMM kernel's ratio, (c) plus one workgroup barrier per 24 MFMAs.  Cited by DESIGN.md (GEMM roofline discussion).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[320 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 320 * 64 / 8; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f32x16 acc[3][2];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a[3], b[2];
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (short)(lane + i + e);
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (short)(lane * 3 + j + e);
    const unsigned short* pa = lds + ((wave >> 1) * 96 + (lane & 31)) * 64 + ((lane >> 5) ^ ((lane >> 1) & 7)) * 8;
    const unsigned short* pb = lds + (192 + (wave & 1) * 64 + (lane & 31)) * 64 + ((lane >> 5) ^ ((lane >> 1) & 7)) * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * 64 + (ks ^ (it & 3)) * 16 % 48);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * 64 + (ks ^ (it & 3)) * 16 % 48);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 4000, blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 24 * 32768.0;
    printf("%-40s %8.3f ms  %8.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, ms, flop / ms / 1e9, ms * 1e6 / (iters * 24 * 2.0));
}
int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    run<0>("MFMA only", out);
    run<1>("MFMA + 20 ds_read_b128 per 24", out);
    run<2>("MFMA + reads + barrier per 24", out);
    run<0>("MFMA only (again)", out);
    return 0;
}
