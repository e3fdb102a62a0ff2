// Issue cost of packed vs scalar f32 VALU on one CU: cycles per wave-instruction per SIMD at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
    v2f a[8]; float s[16];
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.f - i};
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.5f + i;
    v2f c = {1.0001f, 0.9999f};
    float cs = 1.0001f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(cs));
        } else if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(cs));
        } else if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        }
    }
    const long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int per_iter) {
    float* o; long long* c; hipMalloc(&o, 4 * 1024 * 256); hipMalloc(&c, 8 * 256);
    for (int wps = 1; wps <= 4; ++wps) {
        const int iters = 4096;
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256 * wps), 0, 0, o, c, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256 * wps), 0, 0, o, c, iters);
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        // clock64 = s_memtime (shader clock ticks?) report ticks per wave-instruction per SIMD
        printf("%-28s waves/SIMD %d: %.2f ticks per instruction per SIMD (%.2f per wave)\n", name, wps, (double)h / ((double)iters * per_iter * wps), (double)h / ((double)iters * per_iter));
    }
}
int main() {
    run<3>("v_fma_f32", 16); run<4>("v_add_f32", 16); run<0>("v_pk_fma_f32", 8); run<5>("v_pk_add_f32", 8); run<1>("v_pk_add_f32 op_sel/neg", 8); run<2>("v_pk_mul_f32", 8);
    return 0;
}
