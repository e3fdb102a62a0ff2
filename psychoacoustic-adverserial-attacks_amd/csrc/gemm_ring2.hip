// LDS-DMA ring GEMM with SEPARATE rings for the two operands (gfx950; contract in gemm.h, epilogues in gemm_dev.h).
//
// gemm_ring.hip keeps A and B slabs of one K step in one stage, so a 256 x 256 tile (64 KB per stage) has ONE stage of
// lookahead in 160 KB of LDS: 0.5 us (bf16) / 0.7 us (split) of MFMA work to cover a fetch from HBM.  The ablation builds of
// that kernel (tools/gemm_ablate.sh, profiles/r2_gemm_ablation.txt) price the wait for operands from real memory at 20 % of
// the conv products.  Here the A operand — the activations, streamed from HBM — gets THREE slots (two stages of lookahead)
// and the B operand — the weights, re-read from L2 by every workgroup — two (one stage): 3 x 32 + 2 x 32 KB = all of the LDS
// for a 256 x 256 tile.  Per K slab s, in this order:
//     issue B(s + 1), A(s + 2)   |   MFMAs on A(s), B(s)   |   s_waitcnt vmcnt(A(s + 2) in flight)   |   barrier   |   epilogue if last
// vmcnt retires in issue order on gfx9, so with B issued BEFORE A inside a step the counted wait lets exactly the
// far-ahead A slab stay in flight while B(s + 1) and A(s + 1) are certified for the next step; the one barrier per step also
// frees the slots the next step refills.  The epilogue's stores sit behind the barrier: they drain under the next step's
// MFMAs instead of in front of its first wait.  Same fragment layout, swizzle, K order and epilogue as gemm_ring.hip, so
// results are bit-identical to it.  Measured (profiles/r2_gemm_ab_sq.txt): +0..5 % per product in isolation, -3.6 % on the
// whole fp32-parity step.  Tried on top without effect: starting the workgroups of an XCD in four phases a quarter tile
// apart, so that their epilogues' store bursts do not coincide (every kernel just got slower by the added delay); epilogues
// specialised at compile time for the hot descriptor classes (GELU forward, GELU' backward, f32 result + residual): the first two
// tie the run-time epilogue, the third is 8..16 % SLOWER per product — the epilogue's cost is not its instruction count.
#include <stdlib.h>

#include <algorithm>

#include "gemm_dev.h"

namespace paa {

// -DPAA_EXPERIMENTS builds, PAA_NO_BIL=1 (A/B measurements, cached by gemm.hip): planar weights even where B_il is given
bool gemm_env_no_bil();

namespace {

// 0 (diagnostic builds, -DPAA_EXPERIMENTS): every step reads its first fragments at its top instead of under the previous step's
// last MFMA group
#if defined(PAA_R2_DEFER) && !defined(PAA_EXPERIMENTS)
#error "PAA_R2_DEFER is a diagnostic build: add -DPAA_EXPERIMENTS"
#endif
#ifndef PAA_R2_DEFER
#define PAA_R2_DEFER 1
#endif
// cache-policy bits of the A / B operand DMA (diagnostic builds: 1 = sc0, 2 = nt, 16 = sc1); 0 = plain loads
#if (defined(PAA_R2_A_AUX) || defined(PAA_R2_B_AUX)) && !defined(PAA_EXPERIMENTS)
#error "PAA_R2_A_AUX / PAA_R2_B_AUX are diagnostic builds: add -DPAA_EXPERIMENTS"
#endif
#ifndef PAA_R2_A_AUX
#define PAA_R2_A_AUX 0
#endif
#ifndef PAA_R2_PRIO_BALANCE
#define PAA_R2_PRIO_BALANCE 1
#endif
#ifndef PAA_R2_B_AUX
#define PAA_R2_B_AUX 0
#endif
typedef __attribute__((address_space(1))) const void* gas_ptr2;
typedef __attribute__((address_space(3))) void* las_ptr2;
template <int N>
__device__ __forceinline__ void wait_vmcnt2() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Diagnostic build -DPAA_R2_STAMP (tools/gemm_stamps.py): where a K slab's cycles go, per wave of workgroup 0 — s_memtime around the
// counted vmcnt wait and around the barrier of every slab, summed per tile.
#if defined(PAA_R2_STAMP) && !defined(PAA_EXPERIMENTS)
#error "PAA_R2_STAMP is a diagnostic build: add -DPAA_EXPERIMENTS"
#endif
#ifdef PAA_R2_STAMP
__device__ long long g_r2_stamp[8 * 16 * 6];          // [wave][tile < 16][slabs, loop cycles, vmcnt wait, barrier wait, epilogue, first slab start]
#define R2_NOW() ((long long)__builtin_amdgcn_s_memtime())
#endif

// BIL (split mode): the B operand comes from paa_gemm_desc.B_il — hi and lo planes interleaved per 32-element K group — so a K slab
// of a weight row is ONE 128-byte line; its slot has 128-byte rows (chunks 0-3 hi, 4-7 lo) with the XOR swizzle of the bf16 kernels.
// AIL (split mode only): likewise the A operand from paa_gemm_desc.A_il — the activations' planes interleaved per 32-element K group by
// their producer (Cb_il of the GEMM epilogues, Bf::il of the element-wise kernels) — so that a K slab of an A row, the operand that
// streams from HBM, is one 128-byte line too; its slots take the same 128-byte-row layout.
template <int BM, int BN, int BK, int PREC, int WR, int WC, bool BIL = false, bool AIL = false>
__global__ __launch_bounds__(WR * WC * 64, 2) void k_gemm_ring2(GemmArgs g) {
    constexpr int NW = WR * WC;
    constexpr int NPL = PREC ? 2 : 1;
    constexpr int RB = BK * 2;                 // bytes per LDS row (one plane)
    constexpr int CPR = RB / 16;               // 16-byte chunks per row
    constexpr int RPI = 1024 / RB;             // rows per DMA wave-instruction
    constexpr int RBR = 256 / RB;              // rows per 256-byte bank row
    constexpr int NSTA = 3, NSTB = 2;
    static_assert(!BIL || (PREC == 1 && BK == 32), "interleaved planes: split mode, one 32-element K group per slab");
    constexpr int RBB = BIL ? 128 : RB, CPRB = RBB / 16, RPIB = 1024 / RBB, RBRB = 256 / RBB;      // the B slot's row geometry
    static_assert(!AIL || (PREC == 1 && BK == 32 && BM % (8 * WR * WC) == 0), "interleaved A planes: split mode, one 32-element K group per slab");
    constexpr int RBA = AIL ? 128 : RB, CPRA = RBA / 16, RPIA = 1024 / RBA, RBRA = 256 / RBA;      // the A slot's row geometry
    constexpr int ASZ = NPL * BM * RB, BSZ = NPL * BN * RB;        // one slot of each ring: hi rows [| lo rows]  (BIL: BN rows of 128 bytes)
    constexpr int GA = ASZ / 1024 / NW, GB = BSZ / 1024 / NW;       // DMA wave-instructions per wave per slab
    constexpr int MI = BM / WR / 32, NJ = BN / WC / 32, KS = BK / 16;
    // split mode only: the last MFMA group of a K slab runs AFTER the certifying barrier, over the reads of the next slab's first
    // fragments (+0..4 % per product; in bf16 mode a group is two MFMAs — too short to cover an LDS round trip — and the form
    // measured -3..+1 %)
    constexpr bool DEFER = PAA_R2_DEFER && PREC;
    static_assert(NJ == 2, "vector epilogue: 64 columns per wave");
    static_assert((NPL * BM) % (RPI * NW) == 0 && (NPL * BN) % (RPI * NW) == 0 && BM % RPI == 0 && BN % RPI == 0 && BSZ % (1024 * NW) == 0, "a DMA wave-instruction must not straddle planes");
    static_assert(NSTA * ASZ + NSTB * BSZ <= 160 * 1024 && GA + GB < 32, "rings do not fit");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTA * ASZ + NSTB * BSZ];
    unsigned char* const smA = smem;
    unsigned char* const smB = smem + NSTA * ASZ;

    const paa_gemm_desc& d = g.d;
    const int nwg = g.tiles_m * g.tiles_n;
    const int total = nwg * d.batch;
    const int nk = d.K / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int lr = lane & 31, lh = lane >> 5;
    if ((int)blockIdx.x >= total) return;

    struct Tile { int m0, n0, z1, z2; };
    auto decode = [&](int t) {
        Tile c;
        const int z = t / nwg, orig = t - z * nwg;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
        const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
        c.m0 = tm * BM; c.n0 = tn * BN;
        c.z1 = z / d.batch2; c.z2 = z - c.z1 * d.batch2;
        return c;
    };
    // k_group order of the K slabs (gemm.h): kg_spt slabs per tap, kg_taps taps; 0 = plain K order
    const int kg_spt = (d.k_group > 0 && d.k_group % BK == 0 && d.K % d.k_group == 0 && d.K > d.k_group) ? d.k_group / BK : 0;
    const int kg_taps = kg_spt ? d.K / d.k_group : 1;

    // ---- load cursors: the next (tile, K slab) each operand requests ----------------------------------------------------
    struct Cursor { int t, k, tap, c, slab, slot; };
    auto advance = [&](Cursor& u, int nslots) {        // returns true when the cursor moved on to another tile
        u.slot = u.slot + 1 == nslots ? 0 : u.slot + 1;
        if (kg_spt) {
            if (++u.tap == kg_taps) { u.tap = 0; ++u.c; }
            u.slab = u.tap * kg_spt + u.c;
        } else {
            ++u.slab;
        }
        if (++u.k == nk) {
            u.k = 0; u.tap = 0; u.c = 0; u.slab = 0;
            u.t += gridDim.x;
            return true;
        }
        return false;
    };
    Cursor ca{(int)blockIdx.x, 0, 0, 0, 0, 0}, cb = ca;
    const unsigned short* srcA[GA];
    const unsigned short* srcB[GB];
    auto set_srcA = [&](int t) {
        const Tile c = decode(t);
        const int64_t aoff = c.z1 * d.a_s1 + c.z2 * d.a_s2;
        // The per-lane row / chunk indices below are loop-invariant; left visible, the compiler keeps all GA of them live across the
        // K loop, runs out of registers and RELOADS them from scratch right here — with an s_waitcnt vmcnt(0) that drains the whole
        // operand pipeline once per tile.  Recomputing them from an opaque copy of the lane id costs a dozen ALU instructions per tile.
        // (the lane id itself comes from v_mbcnt inside the statement: an opaque copy of `lane` would just move the spill to `lane`)
        int lane_o;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            if constexpr (AIL) {
                const int rr = (i * NW + wave) * RPIA + lane_o / CPRA;    // row of the slot (128-byte rows: chunks 0-3 hi, 4-7 lo)
                const int ch = (lane_o % CPRA) ^ ((rr / RBRA) & (CPRA - 1));
                srcA[i] = reinterpret_cast<const unsigned short*>(d.A_il) + 2 * aoff + (int64_t)min(c.m0 + rr, d.M - 1) * (2 * d.lda) + ch * 8;
            } else {
                const int r = (i * NW + wave) * RPI + lane_o / CPR;       // row of the slot: hi rows, then lo rows
                const int pl = r / BM, rr = r - pl * BM;
                const int ch = (lane_o % CPR) ^ ((rr / RBR) & (CPR - 1)); // global chunk that lands in slot lane % CPR
                const unsigned short* A = reinterpret_cast<const unsigned short*>(pl ? d.A_lo : (const void*)d.A) + aoff;
                srcA[i] = A + (int64_t)min(c.m0 + rr, d.M - 1) * d.lda + ch * 8;
            }
        }
    };
    auto set_srcB = [&](int t) {
        const Tile c = decode(t);
        const int64_t boff = c.z1 * d.b_s1 + c.z2 * d.b_s2;
        int lane_o;                                            // as in set_srcA
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            const int r = (i * NW + wave) * RPIB + lane_o / CPRB;
            const int pl = BIL ? 0 : r / BN, p = r - pl * BN;
            const int ch = (lane_o % CPRB) ^ ((p / RBRB) & (CPRB - 1));
            const int nl = (p & ~63) + 8 * ((p & 31) >> 2) + 4 * ((p >> 5) & 1) + (p & 3);      // row permutation of the vector epilogue
            if constexpr (BIL) {
                srcB[i] = reinterpret_cast<const unsigned short*>(d.B_il) + 2 * boff + (int64_t)min(c.n0 + nl, d.N - 1) * (2 * d.ldb) + ch * 8;
            } else {
                const unsigned short* B = reinterpret_cast<const unsigned short*>(pl ? d.B_lo : (const void*)d.B) + boff;
                srcB[i] = B + (int64_t)min(c.n0 + nl, d.N - 1) * d.ldb + ch * 8;
            }
        }
    };
    auto issueA = [&]() {                      // -> true if a slab was requested
        if (ca.t >= total) return false;
        unsigned char* st = smA + ca.slot * ASZ;
#pragma unroll
        for (int i = 0; i < GA; ++i)
            __builtin_amdgcn_global_load_lds((gas_ptr2)(srcA[i] + (int64_t)ca.slab * (AIL ? 2 * BK : BK)), (las_ptr2)(st + (i * NW + wave) * 1024), 16, 0, PAA_R2_A_AUX);
        if (advance(ca, NSTA) && ca.t < total) set_srcA(ca.t);
        return true;
    };
    auto issueB = [&]() {
        if (cb.t >= total) return;
        unsigned char* st = smB + cb.slot * BSZ;
#pragma unroll
        for (int i = 0; i < GB; ++i)
            __builtin_amdgcn_global_load_lds((gas_ptr2)(srcB[i] + (int64_t)cb.slab * (BIL ? 2 * BK : BK)), (las_ptr2)(st + (i * NW + wave) * 1024), 16, 0, PAA_R2_B_AUX);
        if (advance(cb, NSTB) && cb.t < total) set_srcB(cb.t);
    };
    set_srcA(ca.t);
    set_srcB(cb.t);
    issueB();                                  // B(0)
    issueA();                                  // A(0)
    bool ahead = issueA();                     // A(1): the one slab that may stay in flight across the certifying wait
    if (ahead) wait_vmcnt2<GA>(); else wait_vmcnt2<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();              // slab 0 of both operands has landed everywhere
    __builtin_amdgcn_sched_barrier(0);

    // fragment addressing: row base + swizzled 16-byte chunk of the k slice
    const int sw = (lr / RBR) & (CPR - 1);
    int offk[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) offk[ks] = ((2 * ks + lh) ^ sw) << 4;
    const int arow = (wr * (BM / WR) + lr) * RBA, brow = (wc * (BN / WC) + lr) * RBB;
    constexpr int ALO = BM * RB, BLO = BN * RB;     // hi -> lo plane distance inside a slot
    // A fragment offsets of k slice ks: hi / lo plane (AIL: chunks 2 ks + lh and 4 + 2 ks + lh of the 128-byte row)
    const int swa = (lr / RBRA) & (CPRA - 1);
    int offah[KS], offal[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        offah[ks] = AIL ? (((2 * ks + lh) ^ swa) << 4) : offk[ks];
        offal[ks] = AIL ? (((4 + 2 * ks + lh) ^ swa) << 4) : ALO + offk[ks];
    }
    // B fragment offsets of k slice ks: hi / lo plane (BIL: chunks 2 ks + lh and 4 + 2 ks + lh of the 128-byte row)
    const int swb = (lr / RBRB) & (CPRB - 1);
    int offbh[KS], offbl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        offbh[ks] = BIL ? (((2 * ks + lh) ^ swb) << 4) : offk[ks];
        offbl[ks] = BIL ? (((4 + 2 * ks + lh) ^ swb) << 4) : BLO + offk[ks];
    }

    int sa_slot = 0, sb_slot = 0;              // slots the MFMAs consume next
    constexpr int NM = NJ * (PREC ? 3 : 1);
    bf16x8 bh[NJ], bl[NJ], bhn[NJ], bln[NJ], ah, al, ahn, aln;
    bool have_first = false;                   // the coming slab's first fragments are already in the "next" registers
#ifdef PAA_R2_STAMP
    int st_tile = 0;
#endif
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const Tile cur = decode(t);
#ifdef PAA_R2_STAMP
        long long st_vm = 0, st_bar = 0;
        const long long st_t0 = R2_NOW();
#endif
        f32x16 acc[MI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            // the slots of slab s - 1 are free since the barrier that ended the previous step
            issueB();
            ahead = issueA();
            const unsigned char* sa = smA + sa_slot * ASZ + arow;
            const unsigned char* sb = smB + sb_slot * BSZ + brow;
            sa_slot = sa_slot + 1 == NSTA ? 0 : sa_slot + 1;
            sb_slot = sb_slot + 1 == NSTB ? 0 : sb_slot + 1;
            const bool last = kt + 1 == nk;
            if (!(DEFER && have_first)) {          // first fragments of this slab (else: read under the previous slab's last MFMAs)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bhn[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * RBB + offbh[0]);
                    if (PREC) bln[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * RBB + offbl[0]);
                }
                ahn = *reinterpret_cast<const bf16x8*>(sa + offah[0]);
                if (PREC) aln = *reinterpret_cast<const bf16x8*>(sa + offal[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, (NJ + 1) * NPL, 0);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { bh[j] = bhn[j]; bl[j] = bln[j]; }
            ah = ahn; al = aln;
            have_first = false;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int ni = (i + 1 < MI) ? i + 1 : 0, nks = (i + 1 < MI) ? ks : ks + 1;
                    if (nks < KS) {
                        ahn = *reinterpret_cast<const bf16x8*>(sa + ni * 32 * RBA + offah[nks < KS ? nks : 0]);
                        if (PREC) aln = *reinterpret_cast<const bf16x8*>(sa + ni * 32 * RBA + offal[nks < KS ? nks : 0]);
                        if (ni == 0) {
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                bhn[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * RBB + offbh[nks < KS ? nks : 0]);
                                if (PREC) bln[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * RBB + offbl[nks < KS ? nks : 0]);
                            }
                        }
                    } else if (DEFER && !last) {
                        // last MFMA group of the slab (its fragments are in registers): certify the next slab FIRST and read its
                        // first fragments under these MFMAs, instead of paying an LDS round trip with all eight waves at the top
                        // of the next step.  (Not across a tile boundary: the fragments would stay live over the epilogue.)
#ifdef PAA_R2_STAMP
                        const long long s0 = R2_NOW();
#endif
                        if (ahead) wait_vmcnt2<GA>(); else wait_vmcnt2<0>();
#ifdef PAA_R2_STAMP
                        const long long s1 = R2_NOW();
#endif
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my last fragment reads of this slab have returned: its slots may be refilled
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
#ifdef PAA_R2_STAMP
                        const long long s2 = R2_NOW();
                        st_vm += s1 - s0; st_bar += s2 - s1;
#endif
                        const unsigned char* san = smA + sa_slot * ASZ + arow;
                        const unsigned char* sbn = smB + sb_slot * BSZ + brow;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            bhn[j] = *reinterpret_cast<const bf16x8*>(sbn + j * 32 * RBB + offbh[0]);
                            if (PREC) bln[j] = *reinterpret_cast<const bf16x8*>(sbn + j * 32 * RBB + offbl[0]);
                        }
                        ahn = *reinterpret_cast<const bf16x8*>(san + offah[0]);
                        if (PREC) aln = *reinterpret_cast<const bf16x8*>(san + offal[0]);
                        have_first = true;
                    }
#if PAA_R2_PRIO_BALANCE
                    // The two waves of a SIMD share its matrix pipe; at equal priority the older one wins every tie, runs ahead, finishes its
                    // slab ~1700 cycles early and parks at the barrier while the younger one drags its last third alone at a third of the
                    // issue rate (stamps, tools/gemm_stamps.py).  Priority that FALLS with progress through the slab lets whichever wave is
                    // behind win the pipe: 3, 3, 2, 2, 1, 1, 0, 0 over the 8 MFMA groups of a 256-row slab.
                    switch (3 - ((ks * MI + i) * 4) / (KS * MI)) {
                        case 3: __builtin_amdgcn_s_setprio(3); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        default: __builtin_amdgcn_s_setprio(0); break;
                    }
#else
                    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (PREC) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    if (nks < KS) {
                        if (ni == 0) __builtin_amdgcn_sched_group_barrier(0x100, (NJ + 1) * NPL, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
                    } else if (DEFER && !last) {
                        __builtin_amdgcn_sched_group_barrier(0x100, (NJ + 1) * NPL, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
                    if (nks < KS) {
                        ah = ahn; al = aln;
                        if (ni == 0) {
#pragma unroll
                            for (int j = 0; j < NJ; ++j) { bh[j] = bhn[j]; bl[j] = bln[j]; }
                        }
                    }
                }
            if (!DEFER || last) {
                // certify the next slab: everything but the far-ahead A slab has landed; everyone is done reading this one
                if (ahead) wait_vmcnt2<GA>(); else wait_vmcnt2<0>();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef PAA_R2_STAMP
        const long long st_t1 = R2_NOW();
#endif
        epilogue_vec<MI, true>(d, acc, cur.m0 + wr * (BM / WR), cur.n0 + wc * (BN / WC), cur.z1, cur.z2, lane);
#ifdef PAA_R2_STAMP
        if (blockIdx.x == 0 && lane == 0 && st_tile < 16) {
            long long* o = g_r2_stamp + (wave * 16 + st_tile) * 6;
            o[0] = nk; o[1] = st_t1 - st_t0; o[2] = st_vm; o[3] = st_bar; o[4] = R2_NOW() - st_t1; o[5] = st_t0;
        }
        ++st_tile;
#endif
    }
}

template <int BM, int BN, int BK, int PREC, int WR, int WC, bool BIL, bool AIL>
void launch_ring2_il(const GemmArgs& g, hipStream_t st) {
    static const int per_cu = blocks_per_cu(k_gemm_ring2<BM, BN, BK, PREC, WR, WC, BIL, AIL>, WR * WC * 64);
    const int resident = per_cu * device_cus();
    const int total = g.tiles_m * g.tiles_n * g.d.batch;
    const int blocks = resident > 0 ? std::min(total, resident) : total;
    hipLaunchKernelGGL((k_gemm_ring2<BM, BN, BK, PREC, WR, WC, BIL, AIL>), dim3(blocks), dim3(WR * WC * 64), 0, st, g);
}
template <int BM, int BN, int BK, int PREC, int WR, int WC>
void launch_ring2(const GemmArgs& g, hipStream_t st) {
    if constexpr (PREC == 1) {
        const bool bil = g.d.B_il && g.d.b_s1 == 0 && g.d.b_s2 == 0 && !gemm_env_no_bil();
        if (g.d.A_il) {         // interleaved activations (gemm.h, A_il)
            if (bil) launch_ring2_il<BM, BN, BK, PREC, WR, WC, true, true>(g, st);
            else launch_ring2_il<BM, BN, BK, PREC, WR, WC, false, true>(g, st);
            return;
        }
        if (bil) { launch_ring2_il<BM, BN, BK, PREC, WR, WC, true, false>(g, st); return; }
    }
    launch_ring2_il<BM, BN, BK, PREC, WR, WC, false, false>(g, st);
}

}  // namespace

#ifdef PAA_R2_STAMP
}  // namespace paa
extern "C" int paa_debug_r2_stamps(long long* host) {      // diagnostic builds: the stamps of the LAST ring2 launch's workgroup 0
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(paa::g_r2_stamp), sizeof(long long) * 8 * 16 * 6) != hipSuccess) return 1;
    static long long zeros[8 * 16 * 6];                   // next shape starts from a clean table
    return hipMemcpyToSymbol(HIP_SYMBOL(paa::g_r2_stamp), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
}
namespace paa {
#endif

// configuration ids continue gemm_ring.hip's: 20 / 21 = 256 x 256 split / bf16, 22 / 23 = 192 x 256 split / bf16
void launch_ring2_cfg(int cfg, const GemmArgs& g, hipStream_t st) {
    switch (cfg) {
        case 20: launch_ring2<256, 256, 32, 1, 2, 4>(g, st); break;
        case 21: launch_ring2<256, 256, 64, 0, 2, 4>(g, st); break;
        case 22: launch_ring2<192, 256, 32, 1, 2, 4>(g, st); break;
        case 23: launch_ring2<192, 256, 64, 0, 2, 4>(g, st); break;
        default: break;
    }
}

}  // namespace paa
