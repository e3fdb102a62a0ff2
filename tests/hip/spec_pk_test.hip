// Unit check of csrc/spec_pk.h on the device against host arithmetic (built and run by tests/test_gpu_kernels.py::test_spec_pk_primitives):
//   hipcc --offload-arch=gfx950 -O3 -o spec_pk_test tests/hip/spec_pk_test.hip && ./spec_pk_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "../../psychoacoustic-adverserial-attacks_amd/csrc/spec_pk.h"   // tests/hip -> repo root
using namespace paa;
constexpr int NP = 16;
__global__ void k(const float2* in, float2* out) {
    const int i = threadIdx.x;
    v2f a = pk_v(in[3 * i]), b = pk_v(in[3 * i + 1]), c = pk_v(in[3 * i + 2]);
    float2* o = out + (NP + 16) * i;
    o[0] = pk_f(pk_add(a, b)); o[1] = pk_f(pk_sub(a, b)); o[2] = pk_f(pk_mul(a, b)); o[3] = pk_f(pk_add_conj(a, b)); o[4] = pk_f(pk_sub_conj(a, b));
    o[5] = pk_f(pk_add_mi(a, b)); o[6] = pk_f(pk_add_pi(a, b)); o[7] = pk_f(pk_cnj_add_mi(a, b)); o[8] = pk_f(pk_fma(a, b, c)); o[9] = pk_f(pk_fnma(a, b, c));
    o[10] = pk_f(pk_fma_cnjm(a, b, c)); o[11] = pk_f(pk_fma_cnjn(a, b, c)); o[12] = pk_f(pk_rot_m(a)); o[13] = pk_f(pk_rot_p(a)); o[14] = pk_f(pk_cmul(a, b));
    o[15] = pk_f(pk_cmul_conj(a, b));
    v2f x[8];
    for (int j = 0; j < 8; ++j) x[j] = pk_v(in[(3 * i + j) % 192]);
    v2f y[8];
    for (int j = 0; j < 8; ++j) y[j] = x[j];
    pk_dft8<-1>(y);
    for (int j = 0; j < 8; ++j) o[NP + j] = pk_f(y[j]);
    pk_dft8<+1>(x);
    for (int j = 0; j < 8; ++j) o[NP + 8 + j] = pk_f(x[j]);
}
__global__ void kt(unsigned* out) {
    const int l = threadIdx.x;
    v2f x[8];
    for (int r = 0; r < 8; ++r) x[r] = v2f{(float)(r * 64 + l), (float)(1000 + r * 64 + l)};
    pk_transpose_hi(x);
    for (int r = 0; r < 8; ++r) { out[(r * 64 + l) * 2] = (unsigned)x[r].x; out[(r * 64 + l) * 2 + 1] = (unsigned)x[r].y; }
    // mirror pull of the split post / pre passes: lane l reads lane (64 - l) & 63
    const v2f m = pk_bpermute(((64 - l) & 63) << 2, v2f{(float)l, (float)(100 + l)});
    out[1024 + 2 * l] = (unsigned)m.x; out[1024 + 2 * l + 1] = (unsigned)m.y;
}
int main() {
    {
        unsigned* o; static unsigned h[1024 + 128];
        hipMalloc(&o, sizeof(h));
        hipLaunchKernelGGL(kt, dim3(1), dim3(64), 0, 0, o);
        hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 8; ++r)
            for (int l = 0; l < 64; ++l) {
                const int hh = l >> 3, n0 = l & 7;
                const unsigned want = hh * 64 + (r * 8 + n0);      // old x[h] of lane (r, n0)
                if (h[(r * 64 + l) * 2] != want || h[(r * 64 + l) * 2 + 1] != 1000 + want) { if (bad < 8) printf("transpose r=%d l=%d got %u %u want %u\n", r, l, h[(r * 64 + l) * 2], h[(r * 64 + l) * 2 + 1], want); ++bad; }
            }
        printf("transpose_hi: %d wrong\n", bad);
        for (int l = 0; l < 64; ++l) {
            const unsigned want = (64 - l) & 63;
            if (h[1024 + 2 * l] != want || h[1024 + 2 * l + 1] != 100 + want) { printf("bpermute l=%d got %u %u want %u\n", l, h[1024 + 2 * l], h[1024 + 2 * l + 1], want); ++bad; }
        }
        if (bad) { printf("FAIL\n"); return 1; }
    }
    float2 h[192], *d, *o;
    for (int i = 0; i < 192; ++i) h[i] = make_float2(sinf(1.3f * i + 0.2f), cosf(0.7f * i * i + 0.1f));
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(float2) * 64 * (NP + 16));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    static float2 r[64 * (NP + 16)];
    if (hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 1; }
    double worst[NP + 2] = {0};
    for (int i = 0; i < 64; ++i) {
        const float2 a = h[3 * i], b = h[3 * i + 1], c = h[3 * i + 2];
        double e[NP][2] = {
            {a.x + b.x, a.y + b.y}, {a.x - b.x, a.y - b.y}, {a.x * b.x, a.y * b.y}, {a.x + b.x, a.y - b.y}, {a.x - b.x, a.y + b.y},
            {a.x + b.y, a.y - b.x}, {a.x - b.y, a.y + b.x}, {a.x + b.y, -a.y + b.x}, {(double)a.x * b.x + c.x, (double)a.y * b.y + c.y},
            {-(double)a.x * b.x + c.x, -(double)a.y * b.y + c.y}, {(double)a.x * b.x - c.x, -(double)a.y * b.y + c.y},
            {-(double)a.x * b.x + c.x, (double)a.y * b.y - c.y}, {a.x + a.y, a.y - a.x}, {a.x - a.y, a.y + a.x},
            {(double)a.x * b.x - (double)a.y * b.y, (double)a.x * b.y + (double)a.y * b.x},
            {(double)a.x * b.x + (double)a.y * b.y, -(double)a.x * b.y + (double)a.y * b.x}};
        for (int p = 0; p < NP; ++p) {
            const float2 g = r[(NP + 16) * i + p];
            worst[p] = fmax(worst[p], fmax(fabs(g.x - e[p][0]), fabs(g.y - e[p][1])));
        }
        for (int s = 0; s < 2; ++s)
            for (int kk = 0; kk < 8; ++kk) {
                double re = 0, im = 0;
                for (int n = 0; n < 8; ++n) {
                    const float2 v = h[(3 * i + n) % 192];
                    const double ang = (s ? 1 : -1) * 2 * M_PI * n * kk / 8;
                    re += v.x * cos(ang) - v.y * sin(ang); im += v.x * sin(ang) + v.y * cos(ang);
                }
                const float2 g = r[(NP + 16) * i + NP + 8 * s + kk];
                worst[NP + s] = fmax(worst[NP + s], fmax(fabs(g.x - re), fabs(g.y - im)));
            }
    }
    int bad = 0;
    for (int p = 0; p < NP + 2; ++p) { printf("prim %2d worst |err| %.3e\n", p, worst[p]); bad += worst[p] > 2e-6; }
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
