#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* p, uint4* out, int n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p), 0, n, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, blockIdx.x * 1024, 0);
    out[threadIdx.x] = make_uint4(v.x, v.y, v.z, v.w);
}
int main() { return 0; }
