// Probe: sustained v_mfma_f32_32x32x16_bf16 rate on the whole chip as the pieces of a GEMM main loop are added one by one:
// fragment reads (ds_read_b128), one workgroup barrier per K tile, the operand staging (global loads + ds_write_b128, or
// LDS-DMA).  Data stays in L2; no epilogue.  Cited by DESIGN.md section 9.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// MODE bits: 1 fragment reads, 2 barrier per K tile, 4 ds_write_b128 of the staged chunks, 8 global loads of them, 16 LDS-DMA
// NW waves per workgroup, MI x 2 accumulators per wave (wave tile 32 MI x 64), CH staged 16-byte chunks per thread per K tile
template <int MODE, int NW, int MI, int CH, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void k(float* out, int iters, const uint4* __restrict__ gsrc) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned short* wdst = lds;      // staged data lands in the tile region itself (values are irrelevant here)
    for (int i = tid; i < 512 * 72 / 8; i += NT) reinterpret_cast<uint4*>(lds)[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f32x16 acc[MI][2];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a[MI], b[2];
    for (int i = 0; i < MI; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (short)(lane + i + e);
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (short)(lane * 3 + j + e);
    const unsigned short* pa = lds + ((wave & 1) * 128 + (lane & 31)) * 72 + (lane >> 5) * 8;
    const unsigned short* pb = lds + (256 + (wave >> 1) * 64 + (lane & 31)) * 72 + (lane >> 5) * 8;
    uint4 gv[CH];
    for (int u = 0; u < CH; ++u) gv[u] = make_uint4(tid, u, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE & 8) {
#pragma unroll
            for (int u = 0; u < CH; ++u) gv[u] = gsrc[((size_t)(blockIdx.x & 63) * 8192 + ((it * CH + u) & 15) * NT + tid)];
        }
        if (MODE & 16) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
                __builtin_amdgcn_global_load_lds((gas_ptr)(gsrc + ((size_t)(blockIdx.x & 63) * 8192 + ((it * CH + u) & 15) * NT + tid)),
                                                 (las_ptr)(wdst + (u * NW + wave) * 512), 16, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (MODE & 1) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * 72 + ks * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * 72 + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (MODE & 4) {
#pragma unroll
            for (int u = 0; u < CH; ++u) *reinterpret_cast<uint4*>(wdst + (u * NT + tid) * 8) = gv[u];
        }
        if (MODE & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (MODE & 8) for (int u = 0; u < CH; ++u) s += (float)gv[u].x;
    out[blockIdx.x * NT + tid] = s;
}

static const uint4* gsrc;
template <int MODE, int NW, int MI, int CH, int WPE>
void run(const char* name, float* out, int blocks) {
    const int iters = 3000;
    const size_t lds = 512 * 72 * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NW, MI, CH, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NW, MI, CH, WPE>), dim3(blocks), dim3(NW * 64), lds, 0, out, 100, gsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NW, MI, CH, WPE>), dim3(blocks), dim3(NW * 64), lds, 0, out, iters, gsrc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * NW * iters * (MI * 2 * 4) * 32768.0;
    printf("%-64s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}
int main() {
    float* out;
    hipMalloc(&out, 512 * 512 * 4);
    uint4* gs;
    hipMalloc(&gs, 64 * 8192 * 16);             // 8 MB: stays in L2 / Infinity Cache
    hipMemset(gs, 1, 64 * 8192 * 16);
    gsrc = gs;
    printf("192 x 128 tile, 4 waves of 96 x 64, 2 workgroups per CU (10 chunks per thread per K tile)\n");
    run<0, 4, 3, 10, 2>("  MFMA only", out, 512);
    run<1, 4, 3, 10, 2>("  + fragment reads", out, 512);
    run<3, 4, 3, 10, 2>("  + barrier per K tile", out, 512);
    run<7, 4, 3, 10, 2>("  + ds_write_b128 of the staged chunks (no global loads)", out, 512);
    run<15, 4, 3, 10, 2>("  + global loads (L2-resident) + ds_write_b128", out, 512);
    run<19, 4, 3, 10, 2>("  + LDS-DMA instead of loads + ds_write", out, 512);
    printf("256 x 128 tile, 4 waves of 128 x 64, 2 workgroups per CU (12 chunks)\n");
    run<3, 4, 4, 12, 2>("  reads + barrier", out, 512);
    run<15, 4, 4, 12, 2>("  + global loads + ds_write_b128", out, 512);
    run<19, 4, 4, 12, 2>("  + LDS-DMA instead", out, 512);
    printf("256 x 256 tile, 8 waves of 128 x 64, 1 workgroup per CU (8 chunks)\n");
    run<3, 8, 4, 8, 2>("  reads + barrier", out, 256);
    run<15, 8, 4, 8, 2>("  + global loads + ds_write_b128", out, 256);
    run<19, 8, 4, 8, 2>("  + LDS-DMA instead", out, 256);
    run<0, 4, 3, 10, 2>("MFMA only (again)", out, 512);
    return 0;
}
