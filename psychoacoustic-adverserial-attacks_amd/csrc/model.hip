// Wav2Vec2ForCTC forward + CTC + backward-to-waveform, orchestrated over the gfx950 kernels.
//
// Replaces, for the PGD step (training_utils/train.py:136-158 of the reference):
//   core/loss_helpers.py:21  model(input_values=perturbed, labels=labels)   [HF modeling_wav2vec2.py:1667-1736]
//   train.py:158             (direction * loss).backward()                   [input gradient only]
//
// Layout.  Every activation is channel-last.  The feature encoder keeps P_i >= T_i rows per clip with
// P_{i-1} = stride_i * P_i, so layer i's im2col matrix over the WHOLE batch is a plain strided view of
// layer i-1's output (row m starts at row stride_i*m, K = k_i*C contiguous elements, lda = stride_i*C):
// each strided 1-D convolution is ONE GEMM with overlapping A rows, and its input gradient is one GEMM
// per residue class of the stride.  Pad rows are kept at zero.  The transformer runs on the same padded
// row space (M = B * P_last); attention and the grouped positional convolution address clips
// individually through the GEMM's batch strides / time window.
//
// Precision.  Every tensor that is a conv / linear GEMM operand exists in HBM as bf16 planes written by
// its producer's epilogue ("H" buffers: hi, plus lo = bf16(v - hi) in fp32-parity mode); tensors read by
// element-wise kernels (norm inputs, residual streams, pre-activations) stay f32 ("F" buffers).  Weights
// are packed to bf16 planes once on the host.  No weight gradients are computed (the reference computes
// and discards them).
#include <string.h>

#include <map>
#include <stdlib.h>
#include <string>
#include <vector>

#include "model_kernels.h"

using namespace paa;

namespace {

struct CBf {                       // read-only bf16 planes
    const unsigned short* hi = nullptr;
    const unsigned short* lo = nullptr;
    const unsigned short* il = nullptr;      // weights, fp32-parity mode: the two planes interleaved per 32-element K group (gemm.h, B_il)
    bool ail = false;                        // activations: `hi` IS an interleaved array (paa_common.h Bf::il; gemm.h, A_il), `lo` unused
    CBf off(int64_t e) const { return ail ? CBf{hi + 2 * e, nullptr, nullptr, true} : CBf{hi + e, lo ? lo + e : nullptr, nullptr, false}; }
};
static inline CBf ro(const Bf& b) { return CBf{b.hi, b.lo, nullptr, b.il}; }
static inline Bf boff(const Bf& b, int64_t e) { return b.il ? Bf{b.hi + 2 * e, nullptr, true} : Bf{b.hi + e, b.lo ? b.lo + e : nullptr, false}; }
static const Bf NOBF{nullptr, nullptr};
// bf16 result planes of a product: planar (Cb, Cb_lo) or one interleaved array (Cb_il)
static inline void set_cb(paa_gemm_desc& d, const Bf& b) {
    if (b.il) { d.Cb = nullptr; d.Cb_lo = nullptr; d.Cb_il = b.hi; }
    else { d.Cb = b.hi; d.Cb_lo = b.lo; d.Cb_il = nullptr; }
}

struct ConvL {
    int cin, cout, k, s, T, P;
    const float *w0 = nullptr, *b = nullptr, *g = nullptr, *beta = nullptr;   // w0: conv0 weights (f32)
    CBf w;                             // [cout][k*cin]
    std::vector<CBf> wd;               // per residue class of the stride: [cin][(Q+1)*cout]
    std::vector<int> wdQ;
    float *pre = nullptr, *act_f = nullptr, *cv = nullptr, *row_stats = nullptr;
    bool pre16 = false;                // pre holds bf16 (see paa_model::pre16)
    bool gate = false;                 // pre holds gelu'(v) instead of v (both modes; see paa_model::pre16)
    Bf actb{nullptr, nullptr};
};

struct EncL {
    CBf wqkv, wqkv_t, wo, wo_t, w1, w1_t, w2, w2_t;
    const float *bqkv, *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b;
    float *qkv = nullptr, *P = nullptr, *ln1_in, *st1, *fpre, *ln2_in, *st2;   // qkv, P: materialised attention only
                                     // fpre holds bf16 when the model's pre16 flag is set
    Bf qkvH{nullptr, nullptr}, ctxH{nullptr, nullptr};                          // fused attention: bf16 Q|K|V and O
    float* lse = nullptr;
};

__global__ void k_copy_logits(const float* __restrict__ src, float* __restrict__ dst, int B, int T, int P, int V) {
    const int64_t n = (int64_t)B * T * V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % V);
        const int64_t r = i / V;
        const int t = (int)(r % T), b = (int)(r / T);
        dst[i] = src[((int64_t)b * P + t) * V + v];
    }
}

// greedy CTC ids (loss_helpers.py:26 / :61, torch.argmax(logits, -1)): one thread per frame, first maximum wins, a NaN
// counts as the maximum (torch semantics)
__global__ void k_argmax_ids(const float* __restrict__ x, int64_t rows, int V, int16_t* __restrict__ ids) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* row = x + r * V;
    float best = row[0];
    int bi = 0;
    for (int c = 1; c < V; ++c) {
        const float v = row[c];
        if (best != best) break;
        if (v > best || v != v) { best = v; bi = c; }
    }
    ids[r] = (int16_t)bi;
}

// out = dy * gelu'(pre)   (f32 and / or bf16 planes)
__global__ void k_mul_gelu_grad(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ out,
                                Bf outb, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = dy[i] * gelu_grad_f(pre[i]);
        if (out) out[i] = v;
        store_bf16(outb, i, v);
    }
}

}  // namespace

struct paa_model {
    paa_arch a;
    int Bmax, L, prec;
    int T, P, Tp, M;                 // encoder frames, padded frames per clip, score-matrix ld, Bmax * P
    bool rln;                        // post-LN encoder: the residual LN(x) of the attention / FFN output products is evaluated in their
                                     // epilogues from x and the LayerNorm statistics (gemm.h, res_ln_stats); the LayerNorm kernels then
                                     // write bf16 planes only.  false only in -DPAA_EXPERIMENTS builds under PAA_NO_RLN=1
    bool ail;                        // fp32-parity mode: the activation planes that only GEMMs read (conv stack outputs and gradients, the
                                     // LayerNorm outputs that feed QKV / FFN products, the FFN hidden activation and its gradient) are kept
                                     // interleaved per 32-element group (paa_common.h Bf::il; gemm.h A_il / Cb_il)
    bool fused;                      // flash-style attention kernels (head_dim 64); else materialised scores
    bool gate;                       // see pre16; false only in -DPAA_EXPERIMENTS builds under PAA_NO_GATE32=1 (fp32-parity A/B)
    bool pre16;                      // What a GELU keeps for its backward pass (L{l}.fpre, and conv{i}.pre, i < last, of the
                                     // group-norm extractor) is the derivative gelu'(v) itself in BOTH modes (paa_gemm_desc.aux_gate;
                                     // ConvL::gate): it only ever multiplies a gradient, and evaluating it next to the GELU that
                                     // shares its exp takes ~20 instructions per element out of every backward epilogue — same
                                     // function of the same input, evaluated in the forward pass.  bf16 mode (pre16) also stores
                                     // it as bf16 (paa_gemm_desc.aux_bf16)
    std::vector<ConvL> conv;
    std::vector<EncL> enc;
    std::map<std::string, std::pair<const float*, int64_t>> tensors;
    // weights
    const float *fp_ln_g, *fp_ln_b, *fp_b, *pc_b, *enc_ln_g, *enc_ln_b, *lm_b;
    CBf fp_w, fp_wt, pc_w, pc_wd, lm_w, lm_wt;
    // workspace
    float* arena = nullptr;
    int64_t arena_floats = 0;
    float *gn_stats, *gn_bsums, *c0_part, *G = nullptr, *G1 = nullptr, *c0_Mx = nullptr, *c0_kc = nullptr;
    Bf c0_w1H{nullptr, nullptr};
    float* gF[2];                    // f32 conv-stack gradients (conv0's input; every layer under the layer-norm variant)
    Bf gH[2];                        // bf16 conv-stack gradients (dgrad GEMM operands), with zero guard rows in front
    Bf g0H{nullptr, nullptr};        // the gradient wrt conv0's output (planar: k_conv0_dgrad / k_conv0_gn<2> read it); gH[0] when !ail
    float* gz;
    Bf fnH, h0H, xaH, xbH, ctxH, factH, xfinalH;
    float *fp_stats, *h0, *pos_pre, *hsum, *enc_stats, *xa, *xb, *final_in;
    float *logits, *dlogits, *nll, *ctc_work;
    Bf dlogitsH, dxaH, dxbH, dqkvH, dfpreH, dposH, dh0H;
    float *dxa, *dxb, *dctx = nullptr, *dP = nullptr, *dh0, *dfn, *delta = nullptr;
    Bf dctxH{nullptr, nullptr};
    int S_cap;
};

static const float* find(paa_model* m, const std::string& n, int64_t numel, paa_status* st) {
    auto it = m->tensors.find(n);
    if (it == m->tensors.end()) { set_error("missing weight tensor '" + n + "'"); *st = PAA_ERR_MISSING; return nullptr; }
    if (it->second.second != numel) {
        set_error("weight tensor '" + n + "' has " + std::to_string(it->second.second) + " elements, expected " + std::to_string(numel));
        *st = PAA_ERR_SIZE;
        return nullptr;
    }
    return it->second.first;
}
#define NEED(dst, name, numel) do { paa_status _st = PAA_OK; dst = find(m, name, numel, &_st); if (_st != PAA_OK) { paa_model_destroy(m); return _st; } } while (0)
// bf16 weight planes: "<name>" (hi) and, in fp32-parity mode, "<name>.lo"
#define NEEDB(dst, name, numel) do { const float* _h; const float* _l = nullptr; NEED(_h, name, numel); \
    if (m->prec) NEED(_l, std::string(name) + ".lo", numel); \
    dst = CBf{reinterpret_cast<const unsigned short*>(_h), reinterpret_cast<const unsigned short*>(_l), nullptr}; \
    if (m->prec) { auto _it = m->tensors.find(std::string(name) + ".il"); \
                   if (_it != m->tensors.end() && _it->second.second == 2 * (int64_t)(numel)) dst.il = reinterpret_cast<const unsigned short*>(_it->second.first); } } while (0)

extern "C" void paa_model_destroy(paa_model* m) {
    if (!m) return;
    if (m->arena) (void)hipFree(m->arena);
    delete m;
}
extern "C" int64_t paa_model_workspace_bytes(const paa_model* m) { return m ? m->arena_floats * 4 : 0; }
extern "C" int paa_model_frames(const paa_model* m) { return m ? m->T : 0; }

extern "C" paa_status paa_model_create(paa_model** out, const paa_arch* arch, const paa_tensor* tensors, int n_tensors,
                                       int max_batch, int length, int precision) {
    if (!out || !arch || !tensors) PAA_FAIL(PAA_ERR_ARG, "paa_model_create: null argument");
    const paa_arch& a = *arch;
    if (a.n_conv < 2 || a.n_conv > 8) PAA_FAIL(PAA_ERR_ARG, "n_conv=%d unsupported", a.n_conv);
    if (a.hidden % a.heads || (a.hidden / a.heads) % 8 || a.hidden % 8 || a.ffn % 8 || a.vocab % 8)
        PAA_FAIL(PAA_ERR_ARG, "hidden/heads/ffn/vocab must give 16-byte aligned bf16 rows");
    if (a.hidden % a.pos_groups || (a.hidden / a.pos_groups) % 8) PAA_FAIL(PAA_ERR_ARG, "pos-conv group width must be a multiple of 8");
    for (int i = 0; i < a.n_conv; ++i) if (a.conv_dim[i] % 8) PAA_FAIL(PAA_ERR_ARG, "conv_dim must be multiples of 8");
    if (max_batch < 1 || length < 1) PAA_FAIL(PAA_ERR_SIZE, "max_batch/length");
    paa_model* m = new paa_model();
    m->a = a; m->Bmax = max_batch; m->L = length; m->prec = precision ? 1 : 0;
    m->fused = a.hidden / a.heads == 64;      // flash-style kernels; split-bf16 (hi + lo planes) in fp32-parity mode
    m->pre16 = m->prec == 0;
    m->gate = true;
#ifdef PAA_EXPERIMENTS      // tools/gate_ab.py: raw pre-activations in fp32-parity mode
    { const char* e = getenv("PAA_NO_GATE32"); m->gate = m->pre16 || !(e && e[0] == '1'); }
#endif
    m->rln = true;
#ifdef PAA_EXPERIMENTS      // tools/model_ab.py: f32 LayerNorm outputs read back as residuals
    { const char* e = getenv("PAA_NO_RLN"); if (e && e[0] == '1') m->rln = false; }
#endif
    m->ail = m->prec == 1 && a.hidden % 32 == 0 && a.ffn % 32 == 0;
    for (int i = 0; i < a.n_conv; ++i) m->ail = m->ail && a.conv_dim[i] % 32 == 0;
#ifdef PAA_EXPERIMENTS      // tools/ail_ab.py: planar activation planes everywhere
    { const char* e = getenv("PAA_NO_AIL"); if (e && e[0] == '1') m->ail = false; }
#endif
    for (int i = 0; i < n_tensors; ++i) m->tensors[tensors[i].name] = {tensors[i].d_ptr, tensors[i].numel};

    // ---- shapes: conv output lengths and the padded row counts (P_{i-1} = s_i * P_i) ----
    const int nc = a.n_conv;
    m->conv.resize(nc);
    int len = length;
    for (int i = 0; i < nc; ++i) {
        ConvL& c = m->conv[i];
        c.cin = i ? a.conv_dim[i - 1] : 1; c.cout = a.conv_dim[i]; c.k = a.conv_kernel[i]; c.s = a.conv_stride[i];
        if (len < c.k) { paa_model_destroy(m); PAA_FAIL(PAA_ERR_SIZE, "length %d too short for the feature encoder", length); }
        len = (len - c.k) / c.s + 1;
        c.T = len;
    }
    m->T = m->conv[nc - 1].T;
    for (int extra = 1;; ++extra) {
        int p = m->T + extra;
        bool ok = true;
        for (int i = nc - 1; i >= 0; --i) {
            m->conv[i].P = p;
            if (p < m->conv[i].T) ok = false;
            p *= m->conv[i].s;
        }
        if (ok) break;
        if (extra > 64) { paa_model_destroy(m); PAA_FAIL(PAA_ERR_SIZE, "cannot find a padded layout for this conv stack"); }
    }
    m->P = m->conv[nc - 1].P;
    m->Tp = (m->T + 31) / 32 * 32;
    m->M = max_batch * m->P;
    const int H = a.hidden, F = a.ffn, V = a.vocab, nh = a.heads, Hg = H / a.pos_groups, C6 = a.conv_dim[nc - 1];
    const int B = max_batch;

    // ---- weights ----
    for (int i = 0; i < nc; ++i) {
        ConvL& c = m->conv[i];
        const std::string p = "c" + std::to_string(i);
        if (i == 0) NEED(c.w0, p + ".w", (int64_t)c.cout * c.k);
        else NEEDB(c.w, p + ".w", (int64_t)c.cout * c.k * c.cin);
        if (a.conv_bias) NEED(c.b, p + ".b", c.cout);
        if ((a.feat_norm_layer == 0 && i == 0) || a.feat_norm_layer == 1) { NEED(c.g, p + ".g", c.cout); NEED(c.beta, p + ".beta", c.cout); }
        if (i > 0) {
            for (int rho = 0; rho < c.s; ++rho) {
                if (rho > c.k - 1) { c.wd.push_back(CBf{}); c.wdQ.push_back(-1); continue; }
                const int Q = (c.k - 1 - rho) / c.s;
                CBf w;
                NEEDB(w, p + ".wd" + std::to_string(rho), (int64_t)c.cin * (Q + 1) * c.cout);
                c.wd.push_back(w); c.wdQ.push_back(Q);
            }
        }
    }
    NEED(m->fp_ln_g, "fp.ln_g", C6); NEED(m->fp_ln_b, "fp.ln_b", C6); NEED(m->fp_b, "fp.b", H);
    NEEDB(m->fp_w, "fp.w", (int64_t)H * C6); NEEDB(m->fp_wt, "fp.wt", (int64_t)H * C6);
    NEEDB(m->pc_w, "pc.w", (int64_t)H * Hg * a.pos_k); NEEDB(m->pc_wd, "pc.wd", (int64_t)H * Hg * a.pos_k); NEED(m->pc_b, "pc.b", H);
    NEED(m->enc_ln_g, "enc.ln_g", H); NEED(m->enc_ln_b, "enc.ln_b", H);
    NEEDB(m->lm_w, "lm.w", (int64_t)V * H); NEEDB(m->lm_wt, "lm.wt", (int64_t)V * H); NEED(m->lm_b, "lm.b", V);
    m->enc.resize(a.layers);
    for (int l = 0; l < a.layers; ++l) {
        EncL& e = m->enc[l];
        const std::string p = "L" + std::to_string(l);
        NEEDB(e.wqkv, p + ".wqkv", (int64_t)3 * H * H); NEEDB(e.wqkv_t, p + ".wqkv_t", (int64_t)3 * H * H); NEED(e.bqkv, p + ".bqkv", 3 * H);
        NEEDB(e.wo, p + ".wo", (int64_t)H * H); NEEDB(e.wo_t, p + ".wo_t", (int64_t)H * H); NEED(e.bo, p + ".bo", H);
        NEED(e.ln1_g, p + ".ln1_g", H); NEED(e.ln1_b, p + ".ln1_b", H);
        NEEDB(e.w1, p + ".w1", (int64_t)F * H); NEEDB(e.w1_t, p + ".w1_t", (int64_t)F * H); NEED(e.b1, p + ".b1", F);
        NEEDB(e.w2, p + ".w2", (int64_t)F * H); NEEDB(e.w2_t, p + ".w2_t", (int64_t)F * H); NEED(e.b2, p + ".b2", H);
        NEED(e.ln2_g, p + ".ln2_g", H); NEED(e.ln2_b, p + ".ln2_b", H);
    }

    // ---- workspace arena (two passes: size, then carve) ----
    m->S_cap = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int64_t off = 0;
        auto take = [&](int64_t n) -> float* {
            n = (n + 63) / 64 * 64;          // keep every buffer 256-byte aligned
            float* p = pass ? m->arena + off : nullptr;
            off += n;
            return p;
        };
        auto take_bf = [&](int64_t n, bool il = false) -> Bf {     // n bf16 elements per plane; split mode: the two planes are contiguous
            Bf b;                                                    // (hi, then lo), so the same 2 n elements can hold the interleaved form
            n = (n + 127) / 128 * 128;
            b.hi = reinterpret_cast<unsigned short*>(take(m->prec ? n : n / 2));
            b.lo = m->prec ? b.hi + n : nullptr;
            b.il = il && m->prec;
            if (b.il) b.lo = nullptr;
            return b;
        };
        const int64_t GUARD = 8;             // zero rows before / after a row-matrix (dgrad look-back, im2col look-ahead)
        int maxC = 0;
        for (int i = 0; i < nc; ++i) maxC = std::max(maxC, a.conv_dim[i]);
        for (int i = 0; i < nc; ++i) {
            ConvL& c = m->conv[i];
            const int64_t n = ((int64_t)B * c.P + GUARD) * c.cout;
            c.gate = m->gate && !a.feat_norm_layer && i < nc - 1;
            c.pre16 = m->pre16 && c.gate;
            c.pre = take(c.pre16 ? (n + 1) / 2 : n);
            if (i < nc - 1) c.actb = take_bf(n, m->ail); else c.act_f = take(n);
            if (a.feat_norm_layer) { if (i) c.cv = take(n); c.row_stats = take((int64_t)B * c.P * 2); }
        }
        const ConvL& c0 = m->conv[0];
        m->gn_stats = take((int64_t)B * c0.cout * 2); m->gn_bsums = take((int64_t)B * c0.cout * 2);
        m->c0_part = take(conv0_part_floats(B, c0.T, c0.cout));
        if (a.feat_norm_layer) m->G = take((int64_t)B * c0.P * c0.k);
        else {
            m->G1 = take((int64_t)B * c0.P * 16); m->c0_Mx = take((int64_t)B * c0.k * c0.k); m->c0_kc = take((int64_t)B * 16);
            m->c0_w1H = take_bf((int64_t)B * 16 * c0.cout);
        }
        const int64_t gsz = ((int64_t)B * c0.P + 2 * GUARD) * maxC;
        for (int j = 0; j < 2; ++j) {
            float* p = a.feat_norm_layer ? take(gsz) : nullptr;
            m->gF[j] = (pass && p) ? p + GUARD * maxC : nullptr;
            Bf h = take_bf(gsz, m->ail);
            m->gH[j] = pass ? boff(h, GUARD * maxC) : NOBF;
        }
        if (m->ail && !a.feat_norm_layer) { Bf h = take_bf(gsz); m->g0H = pass ? boff(h, GUARD * maxC) : NOBF; }
        else m->g0H = m->gH[0];
        const int64_t MH = (int64_t)m->M * H, MF = (int64_t)m->M * F, MC = (int64_t)m->M * C6;
        m->gz = take(MC);
        m->fnH = take_bf(MC); m->fp_stats = take((int64_t)m->M * 2);
        m->h0 = take(MH); m->h0H = take_bf(MH); m->pos_pre = take(MH); m->hsum = take(MH); m->enc_stats = take((int64_t)m->M * 2);
        m->xaH = take_bf(MH, m->ail); m->xbH = take_bf(MH, m->ail);
        m->xa = m->rln ? nullptr : take(MH); m->xb = m->rln ? nullptr : take(MH);      // f32 LayerNorm outputs: only without the recomputed residual
        m->ctxH = take_bf(MH); m->factH = take_bf(MF, m->ail); m->xfinalH = take_bf(MH); m->final_in = take(MH);
        const int64_t PM = (int64_t)B * nh * m->Tp * m->Tp;
        const int64_t LS = (int64_t)B * nh * m->Tp;
        for (int l = 0; l < a.layers; ++l) {
            EncL& e = m->enc[l];
            if (m->fused) { e.qkvH = take_bf(3 * MH); e.ctxH = take_bf(MH); e.lse = take(LS); }
            else { e.qkv = take(3 * MH); e.P = take(PM); }
            e.ln1_in = take(MH); e.st1 = take((int64_t)m->M * 2);
            e.fpre = take(m->pre16 ? (MF + 1) / 2 : MF); e.ln2_in = take(MH); e.st2 = take((int64_t)m->M * 2);
        }
        m->logits = take((int64_t)m->M * V); m->dlogits = take((int64_t)m->M * V); m->dlogitsH = take_bf((int64_t)m->M * V);
        m->nll = take(B);
        m->S_cap = std::min(2047, std::max(64, m->T));    // labels longer than T_e are infeasible (infinite CTC loss) but legal
        m->ctc_work = take((int64_t)B * ctc_work_floats_per_clip(m->T, V, m->S_cap));
        m->dxa = take(MH); m->dxaH = take_bf(MH, m->ail); m->dxb = take(MH); m->dxbH = take_bf(MH, m->ail);
        m->dqkvH = take_bf(3 * MH);
        if (m->fused) { m->dctxH = take_bf(MH); m->delta = take(LS); }
        else { m->dctx = take(MH); m->dP = take(PM); }
        m->dfpreH = take_bf(MF, m->ail); m->dposH = take_bf(MH); m->dh0 = take(MH); m->dh0H = take_bf(MH); m->dfn = take(MC);
        if (!pass) {
            m->arena_floats = off;
            hipError_t e = hipMalloc(&m->arena, sizeof(float) * off);
            if (e != hipSuccess) {
                set_error(std::string("hipMalloc of the model workspace (") + std::to_string(off * 4 >> 20) + " MiB): " + hipGetErrorString(e));
                m->arena = nullptr; paa_model_destroy(m); return PAA_ERR_HIP;
            }
            e = hipMemset(m->arena, 0, sizeof(float) * off);
            if (e != hipSuccess) { set_error(std::string("hipMemset: ") + hipGetErrorString(e)); paa_model_destroy(m); return PAA_ERR_HIP; }
        }
    }
    *out = m;
    return PAA_OK;
}

// ---------------------------------------------------------------------------------------------------
// f32-operand descriptor (materialised attention products)
static paa_gemm_desc gd(const paa_model* m, const float* A, const float* Bm, float* C, int M, int N, int K, int64_t lda,
                        int64_t ldb, int64_t ldc) {
    paa_gemm_desc d{};
    d.A = A; d.B = Bm; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.a_kcontig = 1; d.b_kcontig = 1; d.batch = 1; d.batch2 = 1; d.alpha = 1.f; d.precision = m->prec;
    return d;
}
// -DPAA_EXPERIMENTS builds, PAA_K_GROUP=0 (A/B measurements): plain K order in the strided-conv products instead of gemm.h's
// k_group order; always true in the shipped library
namespace paa { bool gemm_env_kgroup(); }
static bool kgroup_on() { return paa::gemm_env_kgroup(); }
// bf16-operand descriptor (every conv / linear product)
static paa_gemm_desc gdb(const paa_model* m, CBf A, CBf W, float* C, Bf Cb, int M, int N, int K, int64_t lda, int64_t ldb,
                         int64_t ldc) {
    paa_gemm_desc d{};
    d.operand_bf16 = 1;
    d.A = reinterpret_cast<const float*>(A.hi); d.A_lo = A.ail ? A.hi : A.lo; d.A_il = A.ail ? A.hi : nullptr;
    d.B = reinterpret_cast<const float*>(W.hi); d.B_lo = W.lo; d.B_il = W.il;
    d.C = C; set_cb(d, Cb);
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.a_kcontig = 1; d.b_kcontig = 1; d.batch = 1; d.batch2 = 1; d.alpha = 1.f; d.precision = m->prec;
    return d;
}

// y = x W^T (+bias) (+epilogue): x (M, K) bf16 planes, W [N][K] bf16 planes
// A residual that is the LayerNorm of a stored array (gemm.h, res_ln_stats): the post-LN encoder adds LN(x) to its attention / FFN
// outputs, and the epilogue evaluates it from x and the statistics k_ln_fwd left instead of reading an f32 copy of LN(x)
struct LnRef { const float* in = nullptr; const float* stats = nullptr; const float* g = nullptr; const float* b = nullptr; };

static paa_status linear(const paa_model* m, CBf x, CBf w, const float* bias, float* y, Bf yb, int M, int N, int K,
                         hipStream_t st, const float* residual = nullptr, int act = 0, float* pre = nullptr,
                         const float* aux = nullptr, bool x16 = false, const LnRef* rln = nullptr) {
    paa_gemm_desc d = gdb(m, x, w, y, yb, M, N, K, K, K, N);
    if (rln && rln->in) { residual = rln->in; d.res_ln_stats = rln->stats; d.res_ln_g = rln->g; d.res_ln_b = rln->b; }
    d.bias = bias; d.residual = residual; d.ld_res = N; d.act = act; d.C_pre = pre; d.aux = aux; d.ld_aux = N;
    d.aux_bf16 = x16 ? 1 : 0;
    d.aux_gate = (m->gate && (act == PAA_ACT_GELU || act == PAA_ACT_GELU_GRAD)) ? 1 : 0;     // kept pre-activations hold gelu'(v) (paa_model::pre16)
    return gemm(d, st);
}

static paa_status forward(paa_model* m, const float* clean, const float* p, int clamp, int B, hipStream_t st) {
    const paa_arch& a = m->a;
    const int nc = a.n_conv, H = a.hidden, F = a.ffn, V = a.vocab, nh = a.heads, hd = H / nh, G = a.pos_groups, Hg = H / G;
    const int M = B * m->P, T = m->T, P = m->P, Tp = m->Tp;
    // ---- feature encoder ----
    {
        ConvL& c = m->conv[0];
        Conv0Args ca{};
        ca.clean = clean; ca.p = p; ca.clamp = clamp; ca.B = B; ca.L = m->L; ca.T = c.T; ca.P = c.P; ca.C = c.cout;
        ca.k = c.k; ca.stride = c.s; ca.w = c.w0; ca.bias = c.b; ca.gamma = c.g; ca.beta = c.beta; ca.eps = 1e-5f;
        ca.pre = c.pre; ca.pre16 = c.pre16; ca.gate = c.gate; ca.actb = c.actb; ca.gn_stats = m->gn_stats; ca.row_stats = c.row_stats;
        if (a.feat_norm_layer) PAA_TRY(conv0_ln_forward(ca, st)); else PAA_TRY(conv0_gn_forward(ca, m->c0_part, st));
    }
    for (int i = 1; i < nc; ++i) {
        ConvL& c = m->conv[i];
        const ConvL& pr = m->conv[i - 1];
        const int K = c.k * c.cin;
        const bool last = i == nc - 1;
        paa_gemm_desc d = gdb(m, ro(pr.actb), c.w, nullptr, NOBF, B * c.P, c.cout, K, (int64_t)c.s * c.cin, K, c.cout);
        d.k_group = kgroup_on() ? c.cin : 0;
        d.bias = c.b; d.row_period = c.P; d.row_valid = c.T;
        if (a.feat_norm_layer) {
            d.C = c.cv;
            PAA_TRY(gemm(d, st));
            PAA_TRY(layernorm_fwd(c.cv, c.g, c.beta, c.pre, c.row_stats, B * c.P, c.cout, 1e-5f, NOBF, last ? NOBF : c.actb,
                                  last ? c.act_f : nullptr, st));
        } else {
            d.C_pre = c.pre; d.aux_bf16 = c.pre16 ? 1 : 0; d.aux_gate = c.gate ? 1 : 0; d.act = PAA_ACT_GELU;
            if (last) d.C = c.act_f; else set_cb(d, c.actb);
            PAA_TRY(gemm(d, st));
        }
    }
    // ---- feature projection: LN + Linear (pad rows forced to zero) ----
    const ConvL& cl = m->conv[nc - 1];
    PAA_TRY(layernorm_fwd(cl.act_f, m->fp_ln_g, m->fp_ln_b, nullptr, m->fp_stats, M, cl.cout, a.ln_eps, m->fnH, NOBF, nullptr, st));
    {
        paa_gemm_desc d = gdb(m, ro(m->fnH), m->fp_w, m->h0, m->h0H, M, H, cl.cout, cl.cout, cl.cout, H);
        d.bias = m->fp_b; d.row_period = P; d.row_valid = T;
        PAA_TRY(gemm(d, st));
    }
    // ---- positional conv (grouped, k taps, zero padded in time per clip) + GELU + residual ----
    float* enc_in = a.stable_ln ? m->enc[0].ln1_in : m->hsum;
    {
        const int K = a.pos_k * Hg;
        paa_gemm_desc d = gdb(m, ro(m->h0H), m->pc_w, enc_in, NOBF, T, Hg, K, H, K, H);
        d.a_kseg = Hg; d.a_kseg_stride = H; d.a_window = 1; d.a_pad = a.pos_k / 2; d.a_rows_valid = T;
        d.batch = B * G; d.batch2 = G;
        d.a_s1 = (int64_t)P * H; d.a_s2 = Hg; d.b_s1 = 0; d.b_s2 = (int64_t)Hg * K; d.c_s1 = (int64_t)P * H; d.c_s2 = Hg;
        d.bias = m->pc_b; d.bias_s2 = Hg; d.act = PAA_ACT_GELU; d.C_pre = m->pos_pre;
        d.residual = m->h0; d.ld_res = H; d.res_s1 = (int64_t)P * H; d.res_s2 = Hg;
        PAA_TRY(gemm(d, st));
    }
    const float* x = enc_in;          // f32 hidden state entering the layer (pre-LN variant)
    CBf xH{};                         // its bf16 planes (post-LN variant only)
    LnRef xln;                        // post-LN variant: the hidden state entering the layer is LN(xln.in), never stored in f32
    if (!a.stable_ln) {
        PAA_TRY(layernorm_fwd(m->hsum, m->enc_ln_g, m->enc_ln_b, m->xa, m->enc_stats, M, H, a.ln_eps, m->xaH, NOBF, nullptr, st));
        if (m->rln) xln = LnRef{m->hsum, m->enc_stats, m->enc_ln_g, m->enc_ln_b};
        x = m->xa; xH = ro(m->xaH);
    }
    const float scale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < a.layers; ++l) {
        EncL& e = m->enc[l];
        const bool lastl = l == a.layers - 1;
        CBf attn_in = xH;
        if (a.stable_ln) {   // x is e.ln1_in
            PAA_TRY(layernorm_fwd(x, e.ln1_g, e.ln1_b, nullptr, e.st1, M, H, a.ln_eps, m->xbH, NOBF, nullptr, st));
            attn_in = ro(m->xbH);
        }
        CBf ctxH = ro(m->ctxH);
        if (m->fused) {
            PAA_TRY(linear(m, attn_in, e.wqkv, e.bqkv, nullptr, e.qkvH, M, 3 * H, H, st));
            AttnArgs aa{};
            aa.qkv = e.qkvH.hi; aa.ctx = e.ctxH.hi; aa.lse = e.lse; aa.qkv_lo = e.qkvH.lo; aa.ctx_lo = e.ctxH.lo;
            aa.T = T; aa.P = P; aa.Tp = Tp; aa.H = H; aa.nh = nh; aa.scale = scale;
            PAA_TRY(attn_fwd(aa, B, hd, st));
            ctxH = ro(e.ctxH);
        } else {
        PAA_TRY(linear(m, attn_in, e.wqkv, e.bqkv, e.qkv, NOBF, M, 3 * H, H, st));
        {   // S = Q K^T  per (clip, head)
            paa_gemm_desc d = gd(m, e.qkv, e.qkv + H, e.P, T, T, hd, 3 * H, 3 * H, Tp);
            d.batch = B * nh; d.batch2 = nh;
            d.a_s1 = (int64_t)P * 3 * H; d.a_s2 = hd; d.b_s1 = (int64_t)P * 3 * H; d.b_s2 = hd;
            d.c_s1 = (int64_t)nh * Tp * Tp; d.c_s2 = (int64_t)Tp * Tp;
            PAA_TRY(gemm(d, st));
        }
        PAA_TRY(softmax_fwd(e.P, B * nh, T, Tp, T, Tp, scale, st));
        {   // ctx = P V  -> bf16 planes only (operand of the output projection)
            paa_gemm_desc d = gd(m, e.P, e.qkv + 2 * H, nullptr, T, hd, T, Tp, 3 * H, H);
            d.b_kcontig = 0; d.Cb = m->ctxH.hi; d.Cb_lo = m->ctxH.lo;
            d.batch = B * nh; d.batch2 = nh;
            d.a_s1 = (int64_t)nh * Tp * Tp; d.a_s2 = (int64_t)Tp * Tp; d.b_s1 = (int64_t)P * 3 * H; d.b_s2 = hd;
            d.c_s1 = (int64_t)P * H; d.c_s2 = hd;
            PAA_TRY(gemm(d, st));
        }
        }
        if (!a.stable_ln) {
            // (m->rln: x / m->xb are null and the residual comes from the LnRef; else the f32 LayerNorm outputs are read back)
            PAA_TRY(linear(m, ctxH, e.wo, e.bo, e.ln1_in, NOBF, M, H, H, st, x, 0, nullptr, nullptr, false, &xln));       // r1 = LN(x_in) + attn
            PAA_TRY(layernorm_fwd(e.ln1_in, e.ln1_g, e.ln1_b, m->xb, e.st1, M, H, a.ln_eps, m->xbH, NOBF, nullptr, st));        // y1
            LnRef y1;
            if (m->rln) y1 = LnRef{e.ln1_in, e.st1, e.ln1_g, e.ln1_b};
            PAA_TRY(linear(m, ro(m->xbH), e.w1, e.b1, nullptr, m->factH, M, F, H, st, nullptr, PAA_ACT_GELU, e.fpre, nullptr, m->pre16));
            PAA_TRY(linear(m, ro(m->factH), e.w2, e.b2, e.ln2_in, NOBF, M, H, F, st, m->xb, 0, nullptr, nullptr, false, &y1));   // r2 = y1 + ffn
            if (lastl) PAA_TRY(layernorm_fwd(e.ln2_in, e.ln2_g, e.ln2_b, nullptr, e.st2, M, H, a.ln_eps, m->xfinalH, NOBF, nullptr, st));
            else PAA_TRY(layernorm_fwd(e.ln2_in, e.ln2_g, e.ln2_b, m->xa, e.st2, M, H, a.ln_eps, m->xaH, NOBF, nullptr, st));
            if (m->rln) xln = LnRef{e.ln2_in, e.st2, e.ln2_g, e.ln2_b};
            x = m->xa; xH = ro(m->xaH);
        } else {
            PAA_TRY(linear(m, ctxH, e.wo, e.bo, e.ln2_in, NOBF, M, H, H, st, x));                                // r1 = x + attn
            PAA_TRY(layernorm_fwd(e.ln2_in, e.ln2_g, e.ln2_b, nullptr, e.st2, M, H, a.ln_eps, m->xbH, NOBF, nullptr, st));
            PAA_TRY(linear(m, ro(m->xbH), e.w1, e.b1, nullptr, m->factH, M, F, H, st, nullptr, PAA_ACT_GELU, e.fpre, nullptr, m->pre16));
            float* xo = lastl ? m->final_in : m->enc[l + 1].ln1_in;
            PAA_TRY(linear(m, ro(m->factH), e.w2, e.b2, xo, NOBF, M, H, F, st, e.ln2_in));                      // r2 = r1 + ffn
            x = xo;
        }
    }
    if (a.stable_ln)
        PAA_TRY(layernorm_fwd(x, m->enc_ln_g, m->enc_ln_b, nullptr, m->enc_stats, M, H, a.ln_eps, m->xfinalH, NOBF, nullptr, st));
    PAA_TRY(linear(m, ro(m->xfinalH), m->lm_w, m->lm_b, m->logits, NOBF, M, V, H, st));
    return PAA_OK;
}

static paa_status backward(paa_model* m, const float* clean, const float* p, int clamp, int B, float* grad, hipStream_t st) {
    const paa_arch& a = m->a;
    const int nc = a.n_conv, H = a.hidden, F = a.ffn, V = a.vocab, nh = a.heads, hd = H / nh, G = a.pos_groups, Hg = H / G;
    const int M = B * m->P, T = m->T, P = m->P, Tp = m->Tp;
    const float scale = 1.0f / sqrtf((float)hd);
    float* dx = m->dxa;   Bf dxH = m->dxaH;      // gradient of the layer output (f32 + bf16 planes)
    float* dx2 = m->dxb;  Bf dx2H = m->dxbH;
    PAA_TRY(linear(m, ro(m->dlogitsH), m->lm_wt, nullptr, dx, NOBF, M, H, V, st));
    if (a.stable_ln)
        PAA_TRY(layernorm_bwd(dx, m->final_in, m->enc_ln_g, m->enc_stats, nullptr, nullptr, dx, dxH, M, H, st));
    for (int l = a.layers - 1; l >= 0; --l) {
        EncL& e = m->enc[l];
        if (!a.stable_ln) {
            PAA_TRY(layernorm_bwd(dx, e.ln2_in, e.ln2_g, e.st2, nullptr, nullptr, dx, dxH, M, H, st));                  // dr2
            PAA_TRY(linear(m, ro(dxH), e.w2_t, nullptr, nullptr, m->dfpreH, M, F, H, st, nullptr, PAA_ACT_GELU_GRAD, nullptr, e.fpre, m->pre16));
            PAA_TRY(linear(m, ro(m->dfpreH), e.w1_t, nullptr, dx2, NOBF, M, H, F, st, dx));                               // dy1 = dr2 + ...
            PAA_TRY(layernorm_bwd(dx2, e.ln1_in, e.ln1_g, e.st1, nullptr, nullptr, dx2, dx2H, M, H, st));               // dr1
        } else {   // dx = dr2 with its planes in dxH
            PAA_TRY(linear(m, ro(dxH), e.w2_t, nullptr, nullptr, m->dfpreH, M, F, H, st, nullptr, PAA_ACT_GELU_GRAD, nullptr, e.fpre, m->pre16));
            PAA_TRY(linear(m, ro(m->dfpreH), e.w1_t, nullptr, dx2, NOBF, M, H, F, st));                                   // dn2
            PAA_TRY(layernorm_bwd(dx2, e.ln2_in, e.ln2_g, e.st2, dx, nullptr, dx2, dx2H, M, H, st));                     // dr1 = dr2 + LN2'
        }
        if (m->fused) {
            PAA_TRY(linear(m, ro(dx2H), e.wo_t, nullptr, nullptr, m->dctxH, M, H, H, st));
            AttnArgs aa{};
            aa.qkv = e.qkvH.hi; aa.ctx = e.ctxH.hi; aa.lse = e.lse; aa.dctx = m->dctxH.hi; aa.delta = m->delta; aa.dqkv = m->dqkvH.hi;
            aa.qkv_lo = e.qkvH.lo; aa.ctx_lo = e.ctxH.lo; aa.dctx_lo = m->dctxH.lo; aa.dqkv_lo = m->dqkvH.lo;
            aa.T = T; aa.P = P; aa.Tp = Tp; aa.H = H; aa.nh = nh; aa.scale = scale;
            PAA_TRY(attn_bwd(aa, B, hd, st));
        } else {
        PAA_TRY(linear(m, ro(dx2H), e.wo_t, nullptr, m->dctx, NOBF, M, H, H, st));
        const int64_t sq = (int64_t)P * 3 * H, sp = (int64_t)nh * Tp * Tp, sp2 = (int64_t)Tp * Tp, sc = (int64_t)P * H;
        {   // dP = dctx V^T
            paa_gemm_desc d = gd(m, m->dctx, e.qkv + 2 * H, m->dP, T, T, hd, H, 3 * H, Tp);
            d.batch = B * nh; d.batch2 = nh;
            d.a_s1 = sc; d.a_s2 = hd; d.b_s1 = sq; d.b_s2 = hd; d.c_s1 = sp; d.c_s2 = sp2;
            PAA_TRY(gemm(d, st));
        }
        {   // dV = P^T dctx
            paa_gemm_desc d = gd(m, e.P, m->dctx, nullptr, T, hd, T, Tp, H, 3 * H);
            d.a_kcontig = 0; d.b_kcontig = 0;
            d.Cb = m->dqkvH.hi + 2 * H; d.Cb_lo = m->dqkvH.lo ? m->dqkvH.lo + 2 * H : nullptr;
            d.batch = B * nh; d.batch2 = nh;
            d.a_s1 = sp; d.a_s2 = sp2; d.b_s1 = sc; d.b_s2 = hd; d.c_s1 = sq; d.c_s2 = hd;
            PAA_TRY(gemm(d, st));
        }
        PAA_TRY(softmax_bwd(m->dP, e.P, B * nh, T, Tp, T, Tp, scale, st));
        {   // dQ = dS K
            paa_gemm_desc d = gd(m, m->dP, e.qkv + H, nullptr, T, hd, T, Tp, 3 * H, 3 * H);
            d.b_kcontig = 0; d.Cb = m->dqkvH.hi; d.Cb_lo = m->dqkvH.lo;
            d.batch = B * nh; d.batch2 = nh;
            d.a_s1 = sp; d.a_s2 = sp2; d.b_s1 = sq; d.b_s2 = hd; d.c_s1 = sq; d.c_s2 = hd;
            PAA_TRY(gemm(d, st));
        }
        {   // dK = dS^T Q
            paa_gemm_desc d = gd(m, m->dP, e.qkv, nullptr, T, hd, T, Tp, 3 * H, 3 * H);
            d.a_kcontig = 0; d.b_kcontig = 0;
            d.Cb = m->dqkvH.hi + H; d.Cb_lo = m->dqkvH.lo ? m->dqkvH.lo + H : nullptr;
            d.batch = B * nh; d.batch2 = nh;
            d.a_s1 = sp; d.a_s2 = sp2; d.b_s1 = sq; d.b_s2 = hd; d.c_s1 = sq; d.c_s2 = hd;
            PAA_TRY(gemm(d, st));
        }
        }
        if (!a.stable_ln) {
            PAA_TRY(linear(m, ro(m->dqkvH), e.wqkv_t, nullptr, dx, NOBF, M, H, 3 * H, st, dx2));          // dx = dr1 + dqkv Wqkv
        } else {
            PAA_TRY(linear(m, ro(m->dqkvH), e.wqkv_t, nullptr, dx, NOBF, M, H, 3 * H, st));               // dn1
            PAA_TRY(layernorm_bwd(dx, e.ln1_in, e.ln1_g, e.st1, dx2, nullptr, dx, dxH, M, H, st));        // dx = dr1 + LN1'
        }
    }
    if (!a.stable_ln)
        PAA_TRY(layernorm_bwd(dx, m->hsum, m->enc_ln_g, m->enc_stats, nullptr, nullptr, dx, NOBF, M, H, st));   // d hsum
    // ---- positional conv backward: dh0 = dhsum + convT(dhsum * gelu'(pos_pre)) ----
    hipLaunchKernelGGL(k_mul_gelu_grad, dim3(std::min(cdiv((int64_t)M * H, 256), 4096)), dim3(256), 0, st, (const float*)dx,
                       (const float*)m->pos_pre, (float*)nullptr, m->dposH, (int64_t)M * H);
    PAA_LAUNCH_CHECK();
    {
        const int K = a.pos_k * Hg;
        paa_gemm_desc d = gdb(m, ro(m->dposH), m->pc_wd, m->dh0, m->dh0H, T, Hg, K, H, K, H);
        d.a_kseg = Hg; d.a_kseg_stride = H; d.a_window = 1; d.a_pad = a.pos_k - 1 - a.pos_k / 2; d.a_rows_valid = T;
        d.batch = B * G; d.batch2 = G;
        d.a_s1 = (int64_t)P * H; d.a_s2 = Hg; d.b_s1 = 0; d.b_s2 = (int64_t)Hg * K; d.c_s1 = (int64_t)P * H; d.c_s2 = Hg;
        d.residual = dx; d.ld_res = H; d.res_s1 = (int64_t)P * H; d.res_s2 = Hg;
        PAA_TRY(gemm(d, st));
    }
    // ---- feature projection backward ----
    const ConvL& cl = m->conv[nc - 1];
    PAA_TRY(linear(m, ro(m->dh0H), m->fp_wt, nullptr, m->dfn, NOBF, M, cl.cout, H, st));
    // gradient wrt conv_{last}'s GELU output, then through the GELU -> gradient wrt its norm output
    PAA_TRY(layernorm_bwd(m->dfn, cl.act_f, m->fp_ln_g, m->fp_stats, nullptr, nullptr, m->gz, NOBF, M, cl.cout, st));
    {
        const int j = (nc - 1) & 1;
        hipLaunchKernelGGL(k_mul_gelu_grad, dim3(std::min(cdiv((int64_t)M * cl.cout, 256), 4096)), dim3(256), 0, st,
                           (const float*)m->gz, (const float*)cl.pre, a.feat_norm_layer ? m->gF[j] : (float*)nullptr,
                           a.feat_norm_layer ? NOBF : m->gH[j], (int64_t)M * cl.cout);
        PAA_LAUNCH_CHECK();
    }
    // ---- feature encoder backward ----
    for (int i = nc - 1; i >= 1; --i) {
        ConvL& c = m->conv[i];
        const ConvL& pr = m->conv[i - 1];
        const int ji = i & 1, jo = (i - 1) & 1;
        if (a.feat_norm_layer)                 // through LayerNorm_i to the raw conv output (f32 in, bf16 planes out)
            PAA_TRY(layernorm_bwd(m->gF[ji], c.cv, c.g, c.row_stats, nullptr, nullptr, nullptr, m->gH[ji], B * c.P, c.cout, st));
        const bool out_f32 = a.feat_norm_layer != 0;          // next consumer is an element-wise (LayerNorm backward) kernel
        const Bf& gout = i == 1 ? m->g0H : m->gH[jo];         // conv0's kernels read planar planes; the dgrad GEMMs further up interleaved ones
        for (int rho = 0; rho < c.s; ++rho) {
            const int Q = c.wdQ[rho];
            const int64_t ldo = (int64_t)c.s * c.cin;
            if (Q < 0) {   // no tap reaches this residue class: zero gradient rows
                if (out_f32) PAA_HIP(hipMemset2DAsync(m->gF[jo] + (int64_t)rho * c.cin, ldo * 4, 0, (size_t)c.cin * 4, (size_t)B * c.P, st));
                else if (gout.il) PAA_HIP(hipMemset2DAsync(gout.hi + 2 * (int64_t)rho * c.cin, ldo * 4, 0, (size_t)c.cin * 4, (size_t)B * c.P, st));
                else {
                    PAA_HIP(hipMemset2DAsync(gout.hi + (int64_t)rho * c.cin, ldo * 2, 0, (size_t)c.cin * 2, (size_t)B * c.P, st));
                    if (gout.lo) PAA_HIP(hipMemset2DAsync(gout.lo + (int64_t)rho * c.cin, ldo * 2, 0, (size_t)c.cin * 2, (size_t)B * c.P, st));
                }
                continue;
            }
            const int K = (Q + 1) * c.cout;
            paa_gemm_desc d = gdb(m, ro(m->gH[ji]).off(-(int64_t)Q * c.cout), c.wd[rho], nullptr, NOBF, B * c.P, c.cin, K, c.cout, K, ldo);
            d.k_group = kgroup_on() ? c.cout : 0;
            if (out_f32) d.C = m->gF[jo] + (int64_t)rho * c.cin;
            else set_cb(d, boff(gout, (int64_t)rho * c.cin));
            d.act = PAA_ACT_GELU_GRAD; d.ld_aux = ldo; d.aux_bf16 = pr.pre16 ? 1 : 0; d.aux_gate = pr.gate ? 1 : 0;
            d.aux = pr.pre16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(pr.pre) + (int64_t)rho * c.cin)
                             : pr.pre + (int64_t)rho * c.cin;
            PAA_TRY(gemm(d, st));
        }
    }
    {
        ConvL& c = m->conv[0];
        Conv0Args ca{};
        ca.clean = clean; ca.p = p; ca.clamp = clamp; ca.B = B; ca.L = m->L; ca.T = c.T; ca.P = c.P; ca.C = c.cout;
        ca.k = c.k; ca.stride = c.s; ca.w = c.w0; ca.bias = c.b; ca.gamma = c.g; ca.beta = c.beta; ca.eps = 1e-5f;
        ca.gn_stats = m->gn_stats; ca.gn_bsums = m->gn_bsums; ca.row_stats = c.row_stats; ca.dpre = m->gF[0]; ca.G = m->G;
        ca.dpreb = m->g0H; ca.G1 = m->G1; ca.w1b = m->c0_w1H; ca.Mx = m->c0_Mx; ca.kc = m->c0_kc;
        PAA_TRY(conv0_backward(ca, a.feat_norm_layer, m->prec, m->c0_part, grad, st));
    }
    return PAA_OK;
}

extern "C" paa_status paa_model_fwd_bwd(paa_model* m, const float* d_clean, const float* d_p, const int32_t* d_labels, int B,
                                        int S_max, int direction, float* d_grad, float* d_logits, float* d_stats,
                                        void* stream) {
    if (!m || !d_clean) PAA_FAIL(PAA_ERR_ARG, "paa_model_fwd_bwd: null argument");
    if (B < 1 || B > m->Bmax) PAA_FAIL(PAA_ERR_SIZE, "batch %d exceeds max_batch %d", B, m->Bmax);
    if (d_labels && (S_max < 1 || S_max > m->S_cap)) PAA_FAIL(PAA_ERR_SIZE, "S_max=%d exceeds capacity %d", S_max, m->S_cap);
    if (d_grad && !d_labels) PAA_FAIL(PAA_ERR_ARG, "gradient requested without labels");
    hipStream_t st = (hipStream_t)stream;
    const int clamp = d_p ? 1 : 0;
    PAA_TRY(forward(m, d_clean, d_p, clamp, B, st));
    const int V = m->a.vocab;
    if (d_logits) {
        hipLaunchKernelGGL(k_copy_logits, dim3(std::min(cdiv((int64_t)B * m->T * V, 256), 2048)), dim3(256), 0, st,
                           (const float*)m->logits, d_logits, B, m->T, m->P, V);
        PAA_LAUNCH_CHECK();
    }
    if (d_labels) {
        PAA_TRY(ctc(m->logits, d_labels, B, m->T, m->P, V, S_max, m->a.blank, (float)direction, m->nll,
                    d_grad ? m->dlogits : nullptr, d_grad ? m->dlogitsH : NOBF, m->ctc_work, st));
        if (d_stats) PAA_TRY(sum_small(m->nll, B, d_stats, st));
    }
    if (d_grad) PAA_TRY(backward(m, d_clean, d_p, clamp, B, d_grad, st));
    return PAA_OK;
}

// Forward + CTC loss only, with explicit control of the clamp: the reference's evaluation adds p WITHOUT clamping
// (training_utils/evaluation.py:16), its training step clamps (train.py:136).
extern "C" paa_status paa_model_forward(paa_model* m, const float* d_clean, const float* d_p, int clamp, const int32_t* d_labels,
                                        int B, int S_max, float* d_logits, float* d_stats, void* stream) {
    if (!m || !d_clean) PAA_FAIL(PAA_ERR_ARG, "paa_model_forward: null argument");
    if (B < 1 || B > m->Bmax) PAA_FAIL(PAA_ERR_SIZE, "batch %d exceeds max_batch %d", B, m->Bmax);
    if (d_labels && (S_max < 1 || S_max > m->S_cap)) PAA_FAIL(PAA_ERR_SIZE, "S_max=%d exceeds capacity %d", S_max, m->S_cap);
    hipStream_t st = (hipStream_t)stream;
    PAA_TRY(forward(m, d_clean, d_p, (d_p && clamp) ? 1 : 0, B, st));
    const int V = m->a.vocab;
    if (d_logits) {
        hipLaunchKernelGGL(k_copy_logits, dim3(std::min(cdiv((int64_t)B * m->T * V, 256), 2048)), dim3(256), 0, st,
                           (const float*)m->logits, d_logits, B, m->T, m->P, V);
        PAA_LAUNCH_CHECK();
    }
    if (d_labels) {
        PAA_TRY(ctc(m->logits, d_labels, B, m->T, m->P, V, S_max, m->a.blank, 1.f, m->nll, nullptr, NOBF, m->ctc_work, st));
        if (d_stats) PAA_TRY(sum_small(m->nll, B, d_stats, st));
    }
    return PAA_OK;
}

// torch.argmax(logits, dim=-1) of core/loss_helpers.py:26,61 on a caller-owned (rows, V) f32 tensor -> int16 ids.
extern "C" paa_status paa_argmax_ids(const float* d_logits, int64_t rows, int V, int16_t* d_ids, void* stream) {
    if (!d_logits || !d_ids) PAA_FAIL(PAA_ERR_ARG, "paa_argmax_ids: null argument");
    if (rows < 1 || V < 1 || V > 32767) PAA_FAIL(PAA_ERR_SIZE, "paa_argmax_ids: rows=%lld V=%d", (long long)rows, V);
    hipLaunchKernelGGL(k_argmax_ids, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, d_logits, rows, V, d_ids);
    PAA_LAUNCH_CHECK();
    return PAA_OK;
}

// sizeof of every struct that crosses the ABI, so the host binding can verify its layout.
extern "C" void paa_abi_sizes(int32_t* out4) {
    out4[0] = (int32_t)sizeof(paa_params);
    out4[1] = (int32_t)sizeof(paa_arch);
    out4[2] = (int32_t)sizeof(paa_tensor);
    out4[3] = (int32_t)sizeof(paa_gemm_desc);
}

// Test/diagnostic access to the internal activations (synchronous copy to host as f32; never on the step path).
// conv{i}.pre (i < last, group-norm extractor) and L{l}.fpre return gelu'(pre) in both modes (see paa_model::pre16).
// f32 buffers: conv{i}.pre, conv{last}.act, conv{i}.cv, h0, pos_pre, hsum, logits, dlogits, nll, G, gbuf0, dh0, dfn, dxa,
//              gn_stats, L{l}.qkv|P|ln1_in|fpre|ln2_in.   bf16 planes (hi + lo summed): conv{i}.act (i < last), fn, xfinal.
extern "C" int64_t paa_model_debug_read(paa_model* m, const char* name, float* host, int64_t max_floats, int B) {
    if (!m || !name) return 0;
    const std::string n(name);
    const paa_arch& a = m->a;
    const int nc = a.n_conv, H = a.hidden, F = a.ffn, V = a.vocab;
    const int64_t M = (int64_t)B * m->P;
    const float* p = nullptr;
    Bf pb = NOBF;
    int64_t cnt = 0;
    for (int i = 0; i < nc && !p && !pb.hi; ++i) {
        const ConvL& c = m->conv[i];
        const std::string b = "conv" + std::to_string(i);
        const int64_t sz = (int64_t)B * c.P * c.cout;
        if (n == b + ".pre") { if (c.pre16) pb = Bf{reinterpret_cast<unsigned short*>(c.pre), nullptr}; else p = c.pre; cnt = sz; }
        else if (n == b + ".act") { if (c.act_f) p = c.act_f; else pb = c.actb; cnt = sz; }
        else if (n == b + ".cv" && c.cv) { p = c.cv; cnt = sz; }
    }
    for (int l = 0; l < a.layers && !p && !pb.hi; ++l) {
        const EncL& e = m->enc[l];
        const std::string b = "L" + std::to_string(l);
        if (n == b + ".qkv") { if (e.qkv) p = e.qkv; else pb = e.qkvH; cnt = M * 3 * H; }
        else if (n == b + ".P" && e.P) { p = e.P; cnt = (int64_t)B * a.heads * m->Tp * m->Tp; }
        else if (n == b + ".ln1_in") { p = e.ln1_in; cnt = M * H; }
        else if (n == b + ".fpre") { if (m->pre16) pb = Bf{reinterpret_cast<unsigned short*>(e.fpre), nullptr}; else p = e.fpre; cnt = M * F; }
        else if (n == b + ".ln2_in") { p = e.ln2_in; cnt = M * H; }
    }
    if (!p && !pb.hi) {
        const ConvL& c0 = m->conv[0];
        if (n == "fn") { pb = m->fnH; cnt = M * a.conv_dim[nc - 1]; }
        else if (n == "h0") { p = m->h0; cnt = M * H; }
        else if (n == "pos_pre") { p = m->pos_pre; cnt = M * H; }
        else if (n == "hsum") { p = a.stable_ln ? m->enc[0].ln1_in : m->hsum; cnt = M * H; }
        else if (n == "xfinal") { pb = m->xfinalH; cnt = M * H; }
        else if (n == "logits") { p = m->logits; cnt = M * V; }
        else if (n == "dlogits") { p = m->dlogits; cnt = M * V; }
        else if (n == "nll") { p = m->nll; cnt = B; }
        else if (n == "G" && m->G) { p = m->G; cnt = (int64_t)B * c0.P * c0.k; }
        else if (n == "gbuf0") { if (m->gF[0]) p = m->gF[0]; else pb = m->g0H; cnt = (int64_t)B * c0.P * c0.cout; }
        else if (n == "dh0") { p = m->dh0; cnt = M * H; }
        else if (n == "dfn") { p = m->dfn; cnt = M * a.conv_dim[nc - 1]; }
        else if (n == "dxa") { p = m->dxa; cnt = M * H; }
        else if (n == "gn_stats") { p = m->gn_stats; cnt = (int64_t)B * c0.cout * 2; }
    }
    if (!p && !pb.hi) return 0;
    if (host && max_floats > 0) {
        const int64_t nn = std::min(cnt, max_floats);
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        if (p) {
            if (hipMemcpy(host, p, sizeof(float) * nn, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        } else {
            if (pb.il) {      // interleaved planes (paa_common.h Bf::il): element i at il_index(i), its lo part 32 further
                const int64_t n2 = ((nn + 31) / 32) * 64;
                std::vector<unsigned short> both(n2);
                if (hipMemcpy(both.data(), pb.hi, 2 * n2, hipMemcpyDeviceToHost) != hipSuccess) return -1;
                for (int64_t i = 0; i < nn; ++i) {
                    const size_t j = il_index((size_t)i);
                    uint32_t u = (uint32_t)both[j] << 16, u2 = (uint32_t)both[j + 32] << 16;
                    float v, v2;
                    memcpy(&v, &u, 4); memcpy(&v2, &u2, 4);
                    host[i] = v + v2;
                }
                return cnt;
            }
            std::vector<unsigned short> hi(nn), lo(pb.lo ? nn : 0);
            if (hipMemcpy(hi.data(), pb.hi, 2 * nn, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            if (pb.lo && hipMemcpy(lo.data(), pb.lo, 2 * nn, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            for (int64_t i = 0; i < nn; ++i) {
                uint32_t u = (uint32_t)hi[i] << 16;
                float v;
                memcpy(&v, &u, 4);
                if (pb.lo) { uint32_t u2 = (uint32_t)lo[i] << 16; float v2; memcpy(&v2, &u2, 4); v += v2; }
                host[i] = v;
            }
        }
    }
    return cnt;
}
// (padded rows per clip of conv layer i; i = -1: encoder frame rows P, -2: score-matrix ld Tp)
extern "C" int paa_model_layout(const paa_model* m, int i) {
    if (!m) return 0;
    if (i == -1) return m->P;
    if (i == -2) return m->Tp;
    if (i >= 0 && i < m->a.n_conv) return m->conv[i].P;
    return 0;
}
