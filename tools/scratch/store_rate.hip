// Per-CU global-store throughput of 16-byte-per-lane stores for several address patterns (one 512-thread workgroup per CU,
// every CU or ONE CU streaming): cycles per wave store instruction.   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
// pattern 0: lane l writes 16 B at l*16            (1 KB contiguous per instruction)
// pattern 1: 16 B pieces at a 32-byte stride       (the f32 gelu' stores of the GEMM epilogue: two instructions fill a 2 KB span)
// pattern 2: 64 B runs (4 lanes) at a 128-byte stride (the interleaved hi / lo plane stores)
// pattern 3: 8 rows x 128 B (8 lanes per row), rows 2 KB apart (a row group of the epilogue: 8 output rows)
template <int PAT>
__global__ __launch_bounds__(512) void k(float4* out, long long* cyc, int iters, size_t per_wg_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * per_wg_bytes + (size_t)wave * (per_wg_bytes / 8);
    size_t off;
    size_t step;
    if (PAT == 0) { off = lane * 16; step = 1024; }
    else if (PAT == 1) { off = lane * 32; step = 2048; }
    else if (PAT == 2) { off = (lane >> 2) * 128 + (lane & 3) * 16; step = 2048; }
    else { off = (size_t)(lane >> 3) * 2048 + (lane & 7) * 16; step = 128; }
    const float4 v = make_float4(lane, wave, 1.f, 2.f);
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        size_t o = off + (size_t)i * step;
        if (PAT == 3) o = off + (size_t)(i & 15) * 128 + (size_t)(i >> 4) * 16384;
        *reinterpret_cast<float4*>(base + o) = v;
    }
    __builtin_amdgcn_s_waitcnt(0);           // stores issued AND retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PAT>
void run(const char* name, float4* buf, long long* c, int ncu) {
    const int iters = 256;                    // 256 KB per wave, 2 MB per workgroup (pattern 1/2: 4 MB span)
    const size_t per_wg = (size_t)8 << 20;
    for (int grid : {1, ncu}) {
        hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(512), 0, 0, buf, c, iters, per_wg);
        hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(512), 0, 0, buf, c, iters, per_wg);
        long long h[1024];
        hipMemcpy(h, c, 8 * grid, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
        s /= grid;
        printf("%-52s grid %3d: %8.0f cycles for %d store instructions per wave x 8 waves: %.1f cycles per wave-instruction per CU, %.1f B/cycle/CU\n",
               name, grid, s, iters, s / (iters * 8), iters * 8 * 1024.0 / s);
    }
}
int main() {
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    float4* buf; long long* c;
    hipMalloc(&buf, (size_t)ncu * (8 << 20)); hipMalloc(&c, 8 * 1024);
    hipMemset(buf, 0, (size_t)ncu * (8 << 20));
    run<0>("1 KB contiguous per instruction", buf, c, ncu);
    run<1>("16 B pieces at a 32 B stride", buf, c, ncu);
    run<2>("64 B runs at a 128 B stride", buf, c, ncu);
    run<3>("8 rows x 128 B, rows 2 KB apart", buf, c, ncu);
    return 0;
}
