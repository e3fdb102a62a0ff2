/* paa_hip.h — C ABI of libpaa_hip.so, the MI355X (gfx950) implementation of the PGD inner step of
 * tomer-erez/Psychoacoustic-adverserial-attacks (hot path only).
 *
 * The reference has no FFI: its boundary is a set of Python call signatures.  Each entry point
 * below names the reference function it replaces (paths are into the reference's src/).  All
 * pointers named d_* are DEVICE pointers owned by the caller (torch tensors on the host side);
 * every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no allocation,
 * no host synchronisation and is hipGraph-capturable, unless stated otherwise.  Every function
 * returns a paa_status; nothing throws or aborts across this boundary.  paa_last_error() returns
 * a thread-local message for the most recent non-zero status.
 */
#ifndef PAA_HIP_H
#define PAA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PAA_OK = 0,
    PAA_ERR_BAD_NORM = 1,    /* train.py:98   ValueError("Unknown norm_type") */
    PAA_ERR_NEED_CLEAN = 2,  /* train.py:90-95 ValueError: snr / tv without clean audio */
    PAA_ERR_SIZE = 3,        /* shape / capacity mismatch (build.py:315 ValueError on length) */
    PAA_ERR_HIP = 4,         /* a HIP runtime call failed; message carries hipGetErrorString */
    PAA_ERR_ARG = 5,         /* null pointer / invalid enum / unsupported parameter */
    PAA_ERR_MISSING = 6      /* a required weight tensor was not supplied */
} paa_status;

/* args.norm_type (training_utils/parser.py:38-40), in the parser's order of choices */
typedef enum {
    PAA_NORM_L2 = 0, PAA_NORM_LINF = 1, PAA_NORM_SNR = 2, PAA_NORM_TV = 3,
    PAA_NORM_FLETCHER_MUNSON = 4, PAA_NORM_MIN_MAX_FREQS = 5, PAA_NORM_MAX_PHON = 6
} paa_norm;

/* The fields of the reference's argparse namespace that the hot path reads (parser.py:10-66). */
typedef struct {
    int32_t norm_type;        /* paa_norm */
    float l2_size, linf_size, snr_db, tv_epsilon, fm_epsilon;
    float min_freq_attack, max_freq_attack, phon_reference_db;
    float lr;                 /* args.lr, PGD step size (train.py:161) */
    int32_t direction;        /* +1 untargeted, -1 targeted (train.py:124) */
} paa_params;

const char* paa_last_error(void);
/* 300 = this header; 301 = the same ABI built with -DPAA_EXPERIMENTS (diagnostic kernels and environment switches compiled in,
 * tools/ only).  Bindings refuse other values. */
int paa_version(void);
/* sizeof(paa_params), sizeof(paa_arch), sizeof(paa_tensor), sizeof(paa_gemm_desc): layout check for bindings */
void paa_abi_sizes(int32_t* out4);

/* ------------------------------------------------------------------ projection context ----- */
/* Holds FFT twiddles, the periodic Hann window, the Fletcher-Munson weight table, the max_phon
 * contour and the frame / partial-sum workspace.  Replaces the per-call setup in
 * core/fourier_transforms.py:20,32 and the scipy interpolator object of core/iso.py:238-266.
 *   fm_table:   host, [10][n_fft/2+1] float64 — iso weight grid already lerped along frequency to
 *               the rfft bins (value < 0 marks a bin outside [20 Hz, 20 kHz] => weight 1.0)
 *   spl_thresh: host, [n_fft/2+1] float32 — build.py:325-348 init_phon_threshold_tensor
 * Allocates device memory (not capturable). max_batch*max_len bounds later calls. */
typedef struct paa_proj paa_proj;
paa_status paa_proj_create(paa_proj** out, int n_fft, int hop_length, int win_length, int sr,
                           const double* fm_table, const float* spl_thresh, int max_batch, int max_len);
void paa_proj_destroy(paa_proj* h);
paa_status paa_proj_set_spl_thresh(paa_proj* h, const float* spl_thresh /* host, F */);

/* training_utils/train.py:69-99 perturbation_constraint (dispatch) over
 * core/projections.py:11-159.  In place on d_p (rows_p, L); rows_p is 1 for the universal
 * perturbation (build.py:301).  d_clean (B, L) may be NULL except for snr / tv; when NULL the
 * frequency-domain result keeps iSTFT length semantics by zeroing samples >= hop*(T-1)
 * exactly as _align_to (train.py:27-35) does with a length-L clean batch. */
paa_status paa_project(paa_proj* h, const paa_params* prm, float* d_p, int rows_p,
                       const float* d_clean, int B, int L, void* stream);

/* Out-of-place form of paa_project: reads d_src (rows_p, L), writes d_dst (rows_p, L); the two must not overlap.  This is
 * how the reference's functions behave (they return a new tensor, projections.py:24,33,46) and it is the fast path of the
 * frequency-domain norms: ONE fused launch (STFT -> per-bin op -> iSTFT + overlap-add; plus the scale launch for
 * fletcher_munson), whereas the in-place form goes through the workspace and a copy-back launch. */
paa_status paa_project_to(paa_proj* h, const paa_params* prm, const float* d_src, float* d_dst, int rows_p,
                          const float* d_clean, int B, int L, void* stream);

/* core/projections.py:68-80 project_min_max_freqs, :116-133 project_fm_norm, :138-159 project_phon_level applied to a
 * spectrum the caller already holds (these reference functions take the complex (B, F, T) STFT tensor, train.py:52-59):
 * d_S_in / d_S_out (B, T, F) complex64 interleaved, frame-major (the transposed view of the reference's layout);
 * in place allowed.  prm->norm_type must be one of the three frequency-domain norms (else PAA_ERR_BAD_NORM). */
paa_status paa_spectrum_project(paa_proj* h, const paa_params* prm, const float* d_S_in, float* d_S_out, int B, int T,
                                void* stream);
/* core/projections.py:83-113 compute_fm_weighted_norm_interp: d_out[0] = sqrt(sum |S|^2 * w(10 log10(|S|^2 + 1e-10), f_bin)) */
paa_status paa_fm_weighted_norm(paa_proj* h, const float* d_S, int B, int T, float* d_out, void* stream);

/* Data-parallel form: project_snr / project_tv use whole-batch statistics of the clean
 * audio (projections.py:11-35, 56-66), so every rank reduces its shard with paa_batch_stats —
 * d_out2 = [sum clean^2, TV(clean)], d_clip_count[0] = (float)B (nullable) — the caller all-reduces
 * these three numbers with the gradient, and paa_project_ext projects from the GLOBAL values:
 * d_clean_stats = device [sum clean^2, TV(clean)] over all ranks; the SNR target norm's clean.numel()
 * (projections.py:27) is d_clip_count[0] * L when d_clip_count (device, the all-reduced clip count — a
 * small integer, exact in f32) is given, else the host value clean_numel.  Nothing here depends on the
 * local batch size being equal across ranks or steps. */
paa_status paa_batch_stats(paa_proj* h, const float* d_clean, int B, int L, float* d_out2, float* d_clip_count,
                           void* stream);
paa_status paa_project_ext(paa_proj* h, const paa_params* prm, float* d_p, int rows_p,
                           const float* d_clean_stats /* device [2] */, const float* d_clip_count /* device [1] or NULL */,
                           double clean_numel /* used when d_clip_count is NULL */, int L, void* stream);

/* core/fourier_transforms.py:4-29 compute_stft: (B, L) -> d_out (B, T, F) complex64 interleaved,
 * T = 1 + L / hop; the (B, F, T) tensor the reference returns is the transpose-view of this. */
paa_status paa_stft(paa_proj* h, const float* d_x, int B, int L, float* d_out, void* stream);
/* core/fourier_transforms.py:31-41 compute_istft: d_S (B, T, F) complex64 -> d_out (B, hop*(T-1)). */
paa_status paa_istft(paa_proj* h, const float* d_S, int B, int T, float* d_out, void* stream);

/* train.py:160-161  p += lr * sign(grad). */
paa_status paa_sign_step(float* d_p, const float* d_grad, float lr, int L, void* stream);
/* core/projections.py:37-39 project_linf(p, min_val, max_val): in-place clamp of n floats to [lo, hi]. */
paa_status paa_clamp(float* d_p, int64_t n, float lo, float hi, void* stream);
/* train.py:136  out = clamp(clean + p, -1, 1), p broadcast over the batch. */
paa_status paa_compose_clamp(const float* d_clean, const float* d_p, float* d_out, int B, int L, void* stream);

/* ------------------------------------------------------------------ model context ---------- */
/* Wav2Vec2ForCTC forward + CTC loss + backward to the waveform (core/loss_helpers.py:12-23 ->
 * transformers modeling_wav2vec2.py:1667-1736; training_utils/train.py:136-158). */
typedef struct {
    int32_t n_conv;               /* 7 */
    int32_t conv_dim[8], conv_kernel[8], conv_stride[8];
    int32_t conv_bias;            /* 0/1 */
    int32_t feat_norm_layer;      /* 0 = "group" (GroupNorm after conv0), 1 = "layer" (LN after every conv) */
    int32_t hidden, layers, heads, ffn;
    int32_t pos_k, pos_groups;
    int32_t stable_ln;            /* do_stable_layer_norm */
    int32_t vocab, blank;
    float ln_eps;
} paa_arch;

typedef struct {
    const char* name;             /* packed-tensor name, see paa_amd/model.py */
    const float* d_ptr;           /* device pointer, stays owned by the caller and must outlive the model */
    int64_t numel;
} paa_tensor;

/* precision: 0 = bf16 MFMA operands / f32 accumulate; 1 = split-bf16 (hi+lo, 3 MFMA passes),
 * fp32-parity mode.  Activations are stored in f32 in both modes.
 * GEMM weights arrive as bf16 bit patterns: "<name>" (hi plane) and, for precision 1, "<name>.lo" (lo plane, bf16(w - hi));
 * optionally "<name>.il" (precision 1, 2 * numel elements): the same two planes interleaved per 32-element K group of each
 * row, [32 hi | 32 lo | 32 hi | ...] — used by the large products where present (csrc/gemm.h, B_il), never required. */
typedef struct paa_model paa_model;
paa_status paa_model_create(paa_model** out, const paa_arch* arch, const paa_tensor* tensors, int n_tensors,
                            int max_batch, int length, int precision);
void paa_model_destroy(paa_model* m);
int64_t paa_model_workspace_bytes(const paa_model* m);
int paa_model_frames(const paa_model* m);     /* T_e for the configured length */

/* One forward + backward.  d_clean (B, L), d_p (1, L) [may be NULL: no perturbation, no clamp —
 * evaluation.py:16 semantics], d_labels (B, S_max) int32 with negatives as padding.
 *   d_grad  (L)  out: sum_b mask_b * dLoss/dperturbed_b, times `direction`   (train.py:158) — NULL => forward only
 *   d_logits (B, T_e, V) out (may be NULL)
 *   d_stats  (8) out: [0] loss (sum over the batch, HF ctc_loss_reduction='sum').  Slots [1..7] are NOT written by
 *            this call: they belong to the caller's data-parallel bookkeeping (paa_amd/training_utils/pgd.py packs
 *            [1] sum clean^2 and [2] TV(clean) from paa_batch_stats, [3] WER word errors, [4] WER reference words,
 *            [5] the local clip count B (paa_batch_stats), behind the gradient, so that ONE all-reduce carries everything
 *            and the global clean.numel() = L * sum_r B_r needs no collective of its own).
 */
paa_status paa_model_fwd_bwd(paa_model* m, const float* d_clean, const float* d_p, const int32_t* d_labels,
                             int B, int S_max, int direction, float* d_grad, float* d_logits, float* d_stats,
                             void* stream);

/* Forward + CTC loss only (core/loss_helpers.py:46-57 get_loss, training_utils/evaluation.py:5-31): clamp = 0 adds p
 * without clamping as the reference's evaluation does (evaluation.py:16); d_p may be NULL (clean evaluation). */
paa_status paa_model_forward(paa_model* m, const float* d_clean, const float* d_p, int clamp, const int32_t* d_labels,
                             int B, int S_max, float* d_logits, float* d_stats, void* stream);

/* core/loss_helpers.py:26,61  pred_ids = torch.argmax(logits, dim=-1): d_logits (rows, V) f32 -> d_ids (rows) int16
 * (first maximum wins; a NaN counts as the maximum, as in torch).  Feeds the host-side greedy CTC decode / WER. */
paa_status paa_argmax_ids(const float* d_logits, int64_t rows, int V, int16_t* d_ids, void* stream);

/* Diagnostics for tests: synchronous copy of a named internal activation to the host (see csrc/model.hip);
 * returns the number of floats the buffer holds for batch B (0 = unknown name, <0 = HIP error). */
int64_t paa_model_debug_read(paa_model* m, const char* name, float* host, int64_t max_floats, int B);
int paa_model_layout(const paa_model* m, int i);  /* padded rows of conv layer i; -1: frame rows P; -2: score ld */

/* ------------------------------------------------------------------ kernel-level test entries */
/* C[M,N] = epilogue(A[M,K] * B) — the MFMA GEMM all conv / linear / attention products go through.
 * See csrc/gemm.h for the descriptor; exported so tests can check each variant against the oracle. */
struct paa_gemm_desc;
paa_status paa_gemm(const struct paa_gemm_desc* d, void* stream);
/* Test / measurement aid: kernel selection of paa_gemm for the large regular products.  0 = automatic (default),
 * 1 = register-staged kernels only, else force ONE LDS-DMA ring configuration wherever its shape constraints hold: 7 / 8 =
 * 192 x 128 split / bf16 (csrc/gemm_ring.hip), 20 / 21 = 256 x 256 split / bf16, 22 / 23 = 192 x 256 split / bf16
 * (csrc/gemm_ring2.hip).  Results are bit-identical across all of them (same K order); tests assert that.  Any other value
 * selects a configuration only a -DPAA_EXPERIMENTS build holds (measured and rejected variants, timing probes); the shipped
 * library falls back to the register-staged kernels for it. */
void paa_gemm_config(int ring_mode);
/* Test aid.  option 0: value != 0 makes the GroupNorm backward of conv0 take its statistics-pass + GEMM-pass path (the fallback of
 * the shapes the fused single-pass kernel does not cover) on every shape, so that tests can compare the two on the same operands. */
paa_status paa_test_option(int option, int value);
/* Measurement aid (bench.py roofline leg): HIP-event timing of every GEMM launch on its own stream.
 * paa_prof_enable(n>0) starts recording up to n launches (0 stops); paa_prof_read fills out[64][4] =
 * {launches, total ms, total algorithmic FLOP (2*M*N*K*batch), total algorithmic HBM bytes (every operand and result
 * byte of the descriptor once)} per kernel variant
 * (tall*32 + bf16_operands*16 + (narrow | 192-row tall tile)*8 + split*4 + a_kcontig*2 + b_kcontig; 40 / 44 = slab kernel
 * of the grouped positional convolution; 60 / 61 = LDS-DMA ring kernels, bf16 / split) and resets. */
paa_status paa_prof_enable(int max_launches);
paa_status paa_prof_read(double* out256);
/* paa_prof_pause(1) stops recording without releasing the events, paa_prof_pause(0) resumes: bench.py creates the
 * events before its timed region and records on the LAST timed step only (the first timing event recorded on a HIP
 * stream switches its queue to profiled dispatch for good, which costs ~4 % on every later launch). */
paa_status paa_prof_pause(int paused);
/* Fused attention (head_dim 64, bf16 planes as uint16): qkv (B*P, 3H) = Q|K|V, ctx/dctx (B*P, H), lse/delta (B*nh, Tp) */
paa_status paa_attn_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int P, int Tp, int H, int nh, void* stream);
paa_status paa_attn_bwd(const void* qkv, const void* ctx, const float* lse, const void* dctx, float* delta, void* dqkv,
                        int B, int T, int P, int Tp, int H, int nh, void* stream);
/* the same with every operand as a hi + lo pair of bf16 planes (split-bf16, the fp32-parity mode) */
paa_status paa_attn_fwd_split(const void* qkv_hi, const void* qkv_lo, void* ctx_hi, void* ctx_lo, float* lse, int B, int T, int P,
                              int Tp, int H, int nh, void* stream);
paa_status paa_attn_bwd_split(const void* qkv_hi, const void* qkv_lo, const void* ctx_hi, const void* ctx_lo, const float* lse,
                              const void* dctx_hi, const void* dctx_lo, float* delta, void* dqkv_hi, void* dqkv_lo,
                              int B, int T, int P, int Tp, int H, int nh, void* stream);
paa_status paa_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* stats,
                             int rows, int cols, float eps, void* stream);
paa_status paa_layernorm_bwd(const float* dy, const float* x, const float* g, const float* stats, float* dx,
                             int rows, int cols, void* stream);
paa_status paa_softmax_fwd(float* s, int rows, int cols, int ld, float scale, void* stream);
paa_status paa_softmax_bwd(float* dp, const float* p, int rows, int cols, int ld, float scale, void* stream);
paa_status paa_ctc(const float* logits, const int32_t* labels, int B, int T, int V, int S_max, int blank,
                   float grad_scale, float* nll /* B */, float* dlogits /* may be NULL */, float* work, void* stream);
int64_t paa_ctc_work_floats(int B, int T, int V, int S_max);

#ifdef __cplusplus
}
#endif
#endif
