// Probe of ds_read_b64_tr_b16 lane semantics (run on the GPU box): tile[row][col] = row * 100 + col, row stride 72.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ __attribute__((aligned(16))) short t[32 * 72];
    for (int i = threadIdx.x; i < 32 * 72; i += 64) t[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // group g reads the 4 x 16 block at rows 4g .. 4g+3, cols 0..15
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(t + (4 * g + q) * 72 + 4 * p));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short h[32 * 72], o[256];
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 72; ++c) h[r * 72 + c] = r * 100 + c;
    short *di, *dout;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    return 0;
}
