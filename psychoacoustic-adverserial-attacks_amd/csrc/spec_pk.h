// Complex arithmetic of the wave FFT as single packed-f32 instructions (gfx950 VOP3P: v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 on an aligned VGPR pair = one complex number, re in the low dword).
//
// hipcc forms packed operations from float2 code by itself, but every "multiply by +-i", conjugate or broadcast of one half
// costs it v_mov / v_pk_mov shuffles (a quarter of the instructions of a frame in the round-3 kernels), and whether a product
// and a sum contract into an fma depends on the basic blocks they land in — two kernels inlining the same source rounded
// frames differently.  The operand modifiers do all of that for free: op_sel / op_sel_hi pick which half of each source feeds
// the low / high result, neg_lo / neg_hi negate a source per half.  Written out here, a radix-8 butterfly is 26 instructions
// and no moves, a complex product is 2, and every kernel that uses them rounds identically.
//
// result.lo = +-src0[op_sel[0]] (*|+) +-src1[op_sel[1]] ...,  result.hi = the same with op_sel_hi (default op_sel = 0, op_sel_hi = 1).
#pragma once
#include <hip/hip_runtime.h>

namespace paa {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_v(float2 a) { v2f r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 pk_f(v2f a) { return make_float2(a.x, a.y); }

#define PAA_PK2(name, text)                                                                    \
    __device__ __forceinline__ v2f name(v2f a, v2f b) {                                        \
        v2f r;                                                                                 \
        asm(text : "=v"(r) : "v"(a), "v"(b));                                                  \
        return r;                                                                              \
    }
#define PAA_PK3(name, text)                                                                    \
    __device__ __forceinline__ v2f name(v2f a, v2f b, v2f c) {                                 \
        v2f r;                                                                                 \
        asm(text : "=v"(r) : "v"(a), "v"(b), "v"(c));                                          \
        return r;                                                                              \
    }

PAA_PK2(pk_add, "v_pk_add_f32 %0, %1, %2")                                                     // a + b
PAA_PK2(pk_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                           // a - b
PAA_PK2(pk_mul, "v_pk_mul_f32 %0, %1, %2")                                                     // (a.x b.x, a.y b.y)
PAA_PK2(pk_add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                   // a + conj(b) = (a.x + b.x, a.y - b.y)
PAA_PK2(pk_sub_conj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                   // a - conj(b) = (a.x - b.x, a.y + b.y)
PAA_PK2(pk_add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")        // a - i b = (a.x + b.y, a.y - b.x)
PAA_PK2(pk_add_pi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")        // a + i b = (a.x - b.y, a.y + b.x)
PAA_PK2(pk_cnj_add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]")    // conj(a - i b) = (a.x + b.y, -a.y + b.x)
PAA_PK3(pk_fma, "v_pk_fma_f32 %0, %1, %2, %3")                                                 // a b + c (per half)
PAA_PK3(pk_fnma, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]")                  // -a b + c
PAA_PK3(pk_fma_cnjm, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[1,0,0]")              // conj(a b - c) = (a.x b.x - c.x, -a.y b.y + c.y)
PAA_PK3(pk_fma_cnjn, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[0,0,1]")              // conj(c - a b) = (-a.x b.x + c.x, a.y b.y - c.y)

// (a.x + a.y, a.y - a.x) = a (1 - i)  and  (a.x - a.y, a.y + a.x) = a (1 + i)
__device__ __forceinline__ v2f pk_rot_m(v2f a) { return pk_add_mi(a, a); }
__device__ __forceinline__ v2f pk_rot_p(v2f a) { return pk_add_pi(a, a); }

// a b (complex): (a.x b.x, a.x b.y) then (-a.y b.y + ., a.y b.x + .)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a conj(b): (a.x b.x, -a.x b.y) then (a.y b.y + ., a.y b.x + .)
__device__ __forceinline__ v2f pk_cmul_conj(v2f a, v2f b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}

#undef PAA_PK2
#undef PAA_PK3

// 8-point DFT in registers, e^{S 2 pi i n k / 8}; S = -1 forward, +1 inverse.  26 packed instructions.
template <int S>
__device__ __forceinline__ void pk_dft8(v2f (&x)[8]) {
    const v2f rr = {0.70710678118654752f, 0.70710678118654752f};
    const v2f a0 = pk_add(x[0], x[4]), a1 = pk_sub(x[0], x[4]), a2 = pk_add(x[2], x[6]), a3 = pk_sub(x[2], x[6]);
    const v2f a4 = pk_add(x[1], x[5]), a5 = pk_sub(x[1], x[5]), a6 = pk_add(x[3], x[7]), a7 = pk_sub(x[3], x[7]);
    // b1 = a1 + S i a3, b3 = a1 - S i a3 (and the same for a5, a7)
    const v2f b0 = pk_add(a0, a2), b2 = pk_sub(a0, a2);
    const v2f b1 = S < 0 ? pk_add_mi(a1, a3) : pk_add_pi(a1, a3), b3 = S < 0 ? pk_add_pi(a1, a3) : pk_add_mi(a1, a3);
    const v2f b4 = pk_add(a4, a6), b6 = pk_sub(a4, a6);
    const v2f b5 = S < 0 ? pk_add_mi(a5, a7) : pk_add_pi(a5, a7), b7 = S < 0 ? pk_add_pi(a5, a7) : pk_add_mi(a5, a7);
    x[0] = pk_add(b0, b4); x[4] = pk_sub(b0, b4);
    x[2] = S < 0 ? pk_add_mi(b2, b6) : pk_add_pi(b2, b6);
    x[6] = S < 0 ? pk_add_pi(b2, b6) : pk_add_mi(b2, b6);
    // x1, x5 = b1 +- b5 (1 + S i) / sqrt 2;  x3, x7 = b3 +- b7 (-1 + S i) / sqrt 2 = b3 -+ b7 (1 - S i) / sqrt 2
    const v2f u = S < 0 ? pk_rot_m(b5) : pk_rot_p(b5);
    const v2f q = S < 0 ? pk_rot_p(b7) : pk_rot_m(b7);
    x[1] = pk_fma(u, rr, b1); x[5] = pk_fnma(u, rr, b1);
    x[3] = pk_fnma(q, rr, b3); x[7] = pk_fma(q, rr, b3);
}

// ---- 8 x 8 transpose between the register index and lane bits 5:3, without LDS --------------------------------------
// In: x[r] of lane (h, n0) (h = lane >> 3, n0 = lane & 7).  Out: x[r] of lane (h, n0) = the old x[h] of lane (r, n0) — the
// exchange between the first and the second radix-8 pass of the wave FFT.  Three butterfly stages, register bit s against lane
// bit 3 + s: v_permlane32_swap / v_permlane16_swap (gfx950) swap half-waves / odd-even rows of two registers in ONE instruction;
// lane bit 3 (halves of a 16-lane row) takes two masked DPP moves.  ~40 VALU instructions for 8 complex values instead of
// 8 ds_write_b64 + 8 ds_read_b64 (8 KB through the LDS per wave and exchange — the LDS pipe is what bounds the fused kernel).
template <int LANE_BIT>
__device__ __forceinline__ void pk_swap_dword(unsigned& a, unsigned& b) {
    // lanes with LANE_BIT clear: b := partner's a;  lanes with LANE_BIT set: a := partner's b  (partner = lane ^ (1 << LANE_BIT))
    if (LANE_BIT == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0]; b = r[1];
    } else if (LANE_BIT == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0]; b = r[1];
    } else {
        const unsigned na = __builtin_amdgcn_update_dpp(a, b, 0x118, 0xF, 0xC, false);      // row_shr:8 into lanes 8..15 of each row
        const unsigned nb = __builtin_amdgcn_update_dpp(b, a, 0x108, 0xF, 0x3, false);      // row_shl:8 into lanes 0..7
        a = na; b = nb;
    }
}
template <int LANE_BIT>
__device__ __forceinline__ void pk_swap(v2f& a, v2f& b) {
    // (through float temporaries: __builtin_bit_cast applied to the element lvalue a.y reads element 0 with this compiler)
    const float fax = a.x, fay = a.y, fbx = b.x, fby = b.y;
    unsigned ax = __float_as_uint(fax), ay = __float_as_uint(fay), bx = __float_as_uint(fbx), by = __float_as_uint(fby);
    pk_swap_dword<LANE_BIT>(ax, bx);
    pk_swap_dword<LANE_BIT>(ay, by);
    a = v2f{__uint_as_float(ax), __uint_as_float(ay)};
    b = v2f{__uint_as_float(bx), __uint_as_float(by)};
}
// the value lane (addr / 4) holds: a pull through the LDS crossbar (ds_bpermute_b32), no LDS memory involved
__device__ __forceinline__ v2f pk_bpermute(int addr, v2f v) {
    const float fx = v.x, fy = v.y;
    const int rx = __builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(fx));
    const int ry = __builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(fy));
    return v2f{__uint_as_float((unsigned)rx), __uint_as_float((unsigned)ry)};
}
__device__ __forceinline__ void pk_transpose_hi(v2f (&x)[8]) {
    pk_swap<5>(x[0], x[4]); pk_swap<5>(x[1], x[5]); pk_swap<5>(x[2], x[6]); pk_swap<5>(x[3], x[7]);
    pk_swap<4>(x[0], x[2]); pk_swap<4>(x[1], x[3]); pk_swap<4>(x[4], x[6]); pk_swap<4>(x[5], x[7]);
    pk_swap<3>(x[0], x[1]); pk_swap<3>(x[2], x[3]); pk_swap<3>(x[4], x[5]); pk_swap<3>(x[6], x[7]);
}

}  // namespace paa
