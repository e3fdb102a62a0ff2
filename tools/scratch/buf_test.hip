// Probe: does the raw-buffer bounds check on gfx950 include the SGPR offset?  (cited by csrc/gemm.hip load_buf)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* p, unsigned* out, int records, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, records, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    const int n = 4096;   // dwords
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r(256);
    // records = 2048 bytes (512 dwords). lanes read 16 B at voffset 16*lane (0..1008) + soffset
    for (int soff : {0, 1024, 1536, 2048, 4096}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 2048, soff);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int first_zero = -1;
        for (int l = 0; l < 64; ++l) if (r[l * 4] == 0 && first_zero < 0) first_zero = l;
        printf("soffset %5d: lane0 = %#x (expect %#x if in range), first zero lane = %d (expect %d if soffset is range-checked)\n",
               soff, r[0], 0x1000 + soff / 4, first_zero, soff >= 2048 ? 0 : (2048 - soff) / 16 >= 64 ? -1 : (2048 - soff) / 16);
    }
    return 0;
}
