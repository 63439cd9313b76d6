/*
 * sea_hip.h — C ABI of libsea_hip.so, the MI355X (gfx950) kernels under the SEA temporal-rollout path.
 *
 * The reference (ParsaEsmati/SEA) has no FFI: its boundary is the Python class surface of
 * models/temporal.py + models/base_blocks.py, whose forward()s dispatch eager ATen ops.  Each entry point
 * below replaces the ATen work of the reference lines it cites; sea_amd/ (the host-side mirror of that class
 * surface) binds them with ctypes.  All paths are relative to the reference repository root.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller; the library never
 *     allocates, frees or retains device memory;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, graph-capturable;
 *   - return 0 on success, a negative SEA_E* code on failure (nothing was launched); the message is available
 *     from sea_last_error() (thread-local); the library never throws and never exits;
 *   - `dtype` selects the ACTIVATION/WEIGHT operand type of the matrix contractions: SEA_F32 uses the exact-f32
 *     MFMA (v_mfma_f32_16x16x4_f32), SEA_BF16 the bf16 MFMA (v_mfma_f32_16x16x32_bf16); accumulation, softmax,
 *     normalisation statistics, residual stream and optimizer state are always fp32;
 *   - "act" tensors have element type `dtype`; "f32" tensors are float.
 */
#ifndef SEA_HIP_H
#define SEA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEA_ABI_VERSION 8

enum { SEA_F32 = 0, SEA_BF16 = 1 };

enum {
    SEA_OK = 0,
    SEA_EINVAL = -1,   /* bad argument (null, misaligned, unsupported shape) */
    SEA_ELAUNCH = -2,  /* HIP reported a launch error */
    SEA_EUNSUPPORTED = -3
};

int sea_abi_version(void);
const char* sea_last_error(void);
/* sizeof of every ABI struct in declaration order (SeaGemmGroup, SeaQkvGroup, SeaQkvCommon, SeaAttnProblem,
 * SeaAttnParams, SeaNormGroup, SeaSiluGroup, SeaIbParams, ..., SeaLaunchRec, SeaGemmNormGroup, SeaExchangeTail, SeaMlpGroup, SeaMlp2Group, SeaKvNorm, SeaKvField, SeaKvPair, SeaKvLayer, SeaKvGlobal, SeaStepPatch, SeaRowChain, SeaAdalnGroup, SeaAdalnQkv, SeaSplitkGroup last): lets a binding verify its layout.  Host only. */
int sea_struct_sizes(int* out, int cap);
/* Number of compute units / name of device 0's architecture as HIP reports them (diagnostics for bench.py). */
int sea_device_info(int* cu_count, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------------------
 * Grouped GEMM with fused epilogue:  for each group g
 *     acc = sum_{s < n_seg} A_s[M,K] . W[N,K]^T                      (W in nn.Linear layout [out, in])
 *     v   = acc + bias_scale * bias                                    (bias may be NULL)
 *     act = 1 (forward GELU):   Z = (act dtype) v if Z != NULL (pre-activation kept for the backward); v = gelu_erf(v)
 *     act = 2 (backward GELU):  v = v * gelu_erf'(Z)                   (Z = the saved pre-activation, required)
 *     v   = v + R                                                      (R fp32 residual, may be NULL; R == C32 accumulates in place)
 *     C32 = v (fp32, may be NULL);  Cact = (act dtype) v (may be NULL)
 * Replaces nn.Linear / `+` / nn.GELU call sites: attention output projection + residual
 * (models/base_blocks.py:201,293; models/temporal.py:136), cross_down / cross_up (+GELU, + sum over j, + residual;
 * models/temporal.py:177-178,185,189-191), MLP Linear layers (models/base_blocks.py:22,25; models/temporal.py:145),
 * proj (models/temporal.py:146), AdaLN cond_mlp.2 (models/base_blocks.py:339,344).
 * n_seg > 1 sums several A operands against the same W (sum_j cross_up(GELU(a_ij)) = cross_up applied to the
 * sum, done inside the MFMA accumulation): segment s is at A + s * a_seg_stride elements.
 * The same entry point runs the data-gradient GEMMs of the backward pass (dX = dY . W with W^T kept as a shadow copy).
 * Requirements: K % 8 == 0; N % 4 == 0; lda, ldw multiples of 8; ldr, ldc32, ldcact, ldz multiples of 4; all pointers
 * 16-byte aligned.
 */
/* Dropout (training only).  thr = round(256 p) in [0, 255], 0 = off.  Element (row, col) of random stream `stream` is KEPT iff
 * byte (col & 3) of the 32-bit word  mix(seed, stream, row, col >> 2)  is >= thr; kept values are scaled by 256 / (256 - thr)
 * (so the effective rate is thr/256 and the expectation is exact).  The same counter-based definition is evaluated by the
 * forward and the backward kernels, so no mask is ever stored.  (The reference draws torch.nn.Dropout masks — models/base_blocks.py:47,
 * 194, 286 — which cannot be reproduced bit for bit; parity is distributional, checked with sea_dropout_mask.) */
typedef struct {
    uint32_t seed;
    uint32_t stream;
    int32_t thr;
    int32_t mode; /* GEMM epilogue: 1 = on (acc + bias) before the residual, 2 = on the act-dtype output only (backward), 3 = on the complete value (acc + bias + R),
                   * both outputs — the position-encoded rows of exchange_mode 'pool' (dropout(x + pe), models/base_blocks.py:370-372) and their gradient */
} SeaDropout;
/* Writes keep * scale (0 or 256/(256-thr)) for a [rows, cols] element grid of one stream: test / debugging aid. */
int sea_dropout_mask(float* out, int64_t rows, int64_t cols, uint32_t seed, uint32_t stream, int32_t thr, void* stream_handle);

#define SEA_MAX_GROUPS 16
typedef struct {
    const void* A;      /* act [M, K], row stride lda */
    const void* W;      /* act [N, K], row stride ldw */
    const float* bias;  /* f32 [N] or NULL */
    const float* R;     /* f32 [M, N] row stride ldr, or NULL */
    float* C32;         /* f32 [M, N] row stride ldc32, or NULL */
    void* Cact;         /* act [M, N] row stride ldcact, or NULL */
    void* Z;            /* act [M, N] row stride ldz: pre-activation out (act = 1, optional) / in (act = 2) */
    int64_t a_seg_stride;
    int32_t lda, ldw, ldr, ldc32, ldcact, ldz;
    int32_t M, N, K, n_seg;
    int32_t act;        /* 0 none, 1 GELU(erf) forward, 2 multiply by GELU'(Z) */
    float bias_scale;
    SeaDropout drop;    /* element grid = (output row, output column) */
    /* generated A operand (all groups of a launch or none): A[m, k] = silu(silu_w1[k] * silu_c[m] + silu_b1[k]) — AdaLN's cond_mlp.0 + SiLU
     * (models/base_blocks.py:337-338, 344) evaluated inside the GEMM of cond_mlp.2; A is ignored, n_seg = 1, act = 0, K <= 1024 */
    const float* silu_c;   /* f32 [M] or NULL */
    const float* silu_w1;  /* f32 [K] */
    const float* silu_b1;  /* f32 [K] */
} SeaGemmGroup;

int sea_gemm_grouped(const SeaGemmGroup* groups, int n_groups, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Q/K/V projection + bias + rotary embedding, written in the layouts the attention kernel consumes.
 * Replaces models/base_blocks.py:179-188 (self) and :271-280 (cross): three nn.Linear, view to heads,
 * apply_rotary_emb (:314-324) on q and k in fp32, transposes.
 * The virtual output row is [q | k | v] (3 * H*hd columns); a group computes columns [col0, col0 + N) of it from
 * its own A and W (self-attention: one group, col0 = 0, N = 3*H*hd, W = [Wq;Wk;Wv]; cross-attention: a q group
 * from x_i with col0 = 0, N = H*hd and a k,v group from x_j with col0 = H*hd, N = 2*H*hd).
 * Row m of A is (trajectory b = m / T, step t = m % T) at absolute position pos0 + t.
 *   Qout  : act [B, H, T,   hd]   (scaled by q_scale after rotation.  The attention kernels below score in LOG2 units: the caller passes
 *                                  q_scale = hd^-1/2 * log2(e) — reference scale models/base_blocks.py:191 times log2 e — so that
 *                                  softmax(q k^T hd^-1/2) = 2^(Q K^T - max) / sum: one v_exp_f32 per probability)
 *   Kout  : act [B, H, cap, hd]   written at row pos0 + t
 *   Vtout : act [B, H, hd, cap]   written at column pos0 + t  (V transposed: keys contiguous)
 *   Vout  : act [B, H, cap, hd]   optional row-major copy of V (training: the attention backward reads it)
 * rope: f32 [>= pos0 + T, hd/2, 2] (cos, sin) — the reference's freqs_cis buffer viewed as real.
 */
typedef struct {
    const void* A;
    const void* W;
    const float* bias; /* f32 [N] */
    void* Qout;
    void* Kout;
    void* Vtout;
    void* Vout;
    int32_t lda, ldw;
    int32_t M, N, K;
    int32_t col0;
} SeaQkvGroup;

typedef struct {
    const float* rope;
    int32_t H, hd, T, pos0, cap;
    float q_scale;
} SeaQkvCommon;

int sea_qkv_rope_grouped(const SeaQkvGroup* groups, int n_groups, const SeaQkvCommon* common, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Causal flash attention forward, several independent problems (fields / field pairs) per launch.
 * Replaces models/base_blocks.py:191-198 and :283-290: (q k^T) * hd^-1/2 (scale already folded into Q),
 * masked_fill(tril(diagonal=src_len) == 0, -inf), softmax, @ v, merge heads — without materialising [T, T].
 * Scores are taken in LOG2 units: P = 2^(Q K^T - rowmax) / rowsum, i.e. Q must carry hd^-1/2 * log2(e) (SeaQkvCommon.q_scale).
 * Query row i (absolute position q_pos0 + i) attends keys j <= q_pos0 + i + src_len, j < Tk.
 *   O   : act [B, Tq, H*hd] row stride ldo
 *   LSE : f32 [B, H, Tq] BASE-2 log-sum-exp of the scores, log2(sum_j 2^(S_ij)) (for the backward pass), may be NULL
 */
#define SEA_MAX_ATTN_PROBLEMS 8
typedef struct {
    const void* Q;
    const void* K;
    const void* Vt;
    void* O;
    float* LSE;
} SeaAttnProblem;

typedef struct {
    SeaAttnProblem p[SEA_MAX_ATTN_PROBLEMS];
    int32_t n_problems;
    int32_t B, H, hd, Tq, Tk, cap, q_pos0, src_len, ldo;
    SeaDropout drop;    /* on the attention probabilities; element grid = (query, key) of stream (drop.stream + problem) * B*H + b*H + h */
} SeaAttnParams;

int sea_attention_fwd(const SeaAttnParams* params, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Row normalisation with optional adaptive modulation and GELU, several groups per launch:
 *     xhat = (x - mean) / sqrt(var_biased + eps)
 *     y    = xhat * (gamma + (mod ? 1 + mod[:, :d] : 0)) + (beta + (mod ? mod[:, d:] : 0));  y = gelu ? gelu_erf(y) : y
 * Replaces AdaLN.forward's normalisation + modulation (models/base_blocks.py:345-350; `mod` is the cond_mlp
 * output), the custom LayerNorm (models/base_blocks.py:87-88: beta NULL, mod NULL) and nn.LayerNorm + nn.GELU inside
 * MLP (models/base_blocks.py:23-24).
 * x is f32 (x_is_act = 0) or act (x_is_act = 1); y is written as f32 (Y32) and/or act (Yact).
 * mean/rstd (f32 [M]) are saved for the backward pass when non-NULL.
 */
typedef struct {
    const void* X;
    const void* mod;    /* act [M, 2d] row stride ldmod, or NULL */
    const float* gamma; /* f32 [d] */
    const float* beta;  /* f32 [d] or NULL */
    float* Y32;
    void* Yact;
    float* mean;
    float* rstd;
    int32_t ldx, ldmod, ldy32, ldyact;
    /* optional addend (x f32, d <= 2048): x' = x + addend is what gets normalised; Xout <- x' when non-NULL (may be X itself) — the
     * info-bottleneck add of models/temporal.py:140-142 folded into the AdaLN_2 pass that follows it */
    const float* addend;  /* f32 [M, d] row stride ldadd, or NULL */
    float* Xout;          /* f32 [M, d] row stride ldxout, or NULL */
    int32_t ldadd, ldxout;
} SeaNormGroup;

int sea_rownorm(const SeaNormGroup* groups, int n_groups, int M, int d, int x_is_act, int gelu, float eps,
                int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * The Linear layers of a KV-cache rollout STEP (utils/train_utils.py:202-209 with per-layer K/V caches: one row per trajectory and field) at the shipped
 * widths (configs/cylinder_flow.py embed_dim 1024, configs/multiphase_flow.py 2048), where a step's time is its launch count:
 * sea_gemm_grouped / sea_qkv_rope_grouped for M <= 4 rows per group with the row norm in front of the layer folded into the launch.
 *   groups  as sea_gemm_grouped (n_seg = 1, act 0 / 1, no dropout, no generated operand) / sea_qkv_rope_grouped; bf16; every group of a launch has the
 *           same K, one of 512, 1024, 2048 (4096, 8192, 16384 too for sea_gemm_fewrows); n_groups <= 8
 *   pre     NULL, or [n_groups]: pre[g].X != NULL makes the A operand of group g  sea_rownorm(pre[g])  of fp32 rows [M, K] (gamma, beta, mod, addend as
 *           SeaNormGroup; K <= 2048), evaluated by every workgroup for itself — groups[g].A is ignored.  Xout (x + addend, fp32) is written once and must
 *           NOT alias X; Y32 / Yact / mean / rstd must be NULL.  (LayerNorm / AdaLN in front of q/k/v, cross-attention and the MLP: models/temporal.py:126-131,
 *           139-145, 170-186; models/base_blocks.py:345-350.)
 *           sea_gemm_fewrows, pre_x_is_act = 1: pre[g].X are rows in the activation dtype instead and the prologue is sea_rownorm(x_is_act, pre_gelu) with
 *           gamma / beta only — nn.LayerNorm + GELU of the hidden rows in front of MLP.layers.3 (models/base_blocks.py:23-25); K <= 8192.
 * Arithmetic: products of bf16 operands accumulated in fp32 (v_dot2c_f32_bf16), norms in fp32 two-pass form exactly as sea_rownorm. */
int sea_gemm_fewrows(const SeaGemmGroup* groups, const SeaNormGroup* pre, int n_groups, int pre_x_is_act, int pre_gelu, float eps, int dtype, void* stream);
int sea_qkv_rope_fewrows(const SeaQkvGroup* groups, const SeaNormGroup* pre, int n_groups, const SeaQkvCommon* common, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Linear layer whose output rows are normalised in the same launch (a workgroup tile spans the whole output row, N <= 256):
 *     v    = sum_s A_s[M,K] . W[N,K]^T + bias_scale * bias (+ R)       (as sea_gemm_grouped);  Cact <- (act dtype) v when non-NULL
 *     v    = v + ib(c)  when ib_c != NULL (info-bottleneck addend of sea_ib_add, no dropout);  C32 <- v when non-NULL
 *     y    = LayerNorm_N(v) with gamma / beta / mod exactly as SeaNormGroup (two-pass fp32 statistics, biased variance, eps)
 *     Yact <- (act dtype) y ;  Y32 <- y ;  mean / rstd saved when non-NULL
 * Replaces the pairs cross_down[i] + ln_cross[i] (models/temporal.py:177-181), proj[i] + the model's final per-field norm
 * (models/temporal.py:146, 412-415), and the triple cross_up[i] (+ sum over j, + residual) / _add_info / ln.exp[i][2]
 * (models/temporal.py:178,189-191 / 140-142 / 145) — one launch and one HBM round trip of the [M, N] intermediate less than
 * sea_gemm_grouped followed by sea_rownorm, with identical arithmetic.
 * Requirements: N % 16 == 0, N <= 256; K % 8 == 0; lda, ldw multiples of 8; ldr, ldc32, ldy32, ldyact, ldmod multiples of 4;
 * all pointers 16-byte aligned.
 */
#define SEA_MAX_GEMM_NORM_GROUPS 8
typedef struct {
    const void* A;      /* act [M, K], row stride lda */
    const void* W;      /* act [N, K], row stride ldw */
    const float* bias;  /* f32 [N] or NULL */
    const float* R;     /* f32 [M, N] row stride ldr, or NULL */
    float* C32;         /* f32 [M, N] pre-normalisation output, row stride ldc32, or NULL */
    const void* mod;    /* act [M, 2N] row stride ldmod, or NULL */
    const float* gamma; /* f32 [N] */
    const float* beta;  /* f32 [N] or NULL */
    float* Y32;         /* f32 [M, N] row stride ldy32, or NULL */
    void* Yact;         /* act [M, N] row stride ldyact, or NULL */
    float* mean;        /* f32 [M] or NULL */
    float* rstd;        /* f32 [M] or NULL */
    int32_t lda, ldw, ldr, ldc32, ldmod, ldy32, ldyact;
    int32_t M, N, K;
    /* extensions (all-zero = off, except n_seg >= 1 and bias_scale): */
    int32_t n_seg;          /* sum of n_seg A operands against the same W, segment s at A + s * a_seg_stride (as SeaGemmGroup) */
    int64_t a_seg_stride;
    float bias_scale;
    int32_t ldcact;
    void* Cact;             /* act [M, N]: v BEFORE the info-bottleneck addend (the copy cross_down reads), or NULL */
    const float* ib_c;      /* f32 [M] condition: non-NULL adds ib = W2 . gelu(LN_h(w1 c + b1)) + b2 (SeaIbParams) to v before C32 and the norm */
    const float* ib_w1;
    const float* ib_b1;
    const float* ib_lnw;
    const float* ib_lnb;
    const float* ib_w2;     /* f32 [N, h] */
    const float* ib_b2;     /* f32 [N] */
    int32_t ib_h;           /* 4 or 8 */
    int32_t pad_;
} SeaGemmNormGroup;

int sea_gemm_rownorm(const SeaGemmNormGroup* groups, int n_groups, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Tail of one field's state-exchange stage in ONE launch (bf16 compute; a workgroup owns 16 rows through all three Linear layers):
 *     g_s   = gelu_erf(att_s[M,D] . Wp_s[D,D]^T)                      s < n_seg      (cross_attn[i][j].projection + GELU, models/temporal.py:183,185)
 *     x     = X + sum_s g_s . Wup[E,D]^T + bias_scale * bup            X updated in place; Xact <- (act dtype) x when non-NULL
 *                                                                      (cross_up[i], the sum over j and the residual, models/temporal.py:185,189-191)
 *     y     = LayerNorm_D(x . Wdown[D,E]^T + bdown) with gamma / beta / mod as SeaNormGroup;  down.Yact / down.Y32 <- y      when has_down
 *                                                                      (cross_down[i] + ln_cross[i] of the UPDATED field: what later fields attend to)
 * The weights of the middle layer go L2 -> LDS in one global_load_lds burst while the first layer runs from registers; the intermediates
 * (g, the new x) never leave the CU.  Replaces three launches (sea_gemm_grouped x 2, sea_gemm_rownorm) on the serial Gauss-Seidel chain.
 * `down`: W, ldw, bias, gamma, beta, mod, ldmod, Yact / Y32 (+ strides), mean, rstd are read; its A, M, N, K and extensions are ignored.
 * Requirements: dtype SEA_BF16; (D, E) in {(128, 256), (64, 128)}; n_seg * D <= 256; all row strides multiples of 8 (act) / 4 (f32);
 * pointers 16-byte aligned.  Returns SEA_EUNSUPPORTED for other shapes / dtypes (callers keep the three-launch form).
 */
#define SEA_XTAIL_MAX_SEG 4
typedef struct {
    const void* att[SEA_XTAIL_MAX_SEG];   /* act [M, D], row stride ldatt */
    const void* Wp[SEA_XTAIL_MAX_SEG];    /* act [D, D], row stride ldwp */
    const void* Wup;                      /* act [E, D], row stride ldwup */
    const float* bup;                     /* f32 [E] or NULL */
    float* X;                             /* f32 [M, E], row stride ldx: residual stream, updated in place */
    void* Xact;                           /* act [M, E], row stride ldxact, or NULL */
    int32_t n_seg, ldatt, ldwp, ldwup, ldx, ldxact;
    int32_t M, D, E, has_down;
    float bias_scale;
    SeaGemmNormGroup down;
} SeaExchangeTail;

/* params: n_groups <= SEA_XTAIL_MAX_GROUPS problems of one shape and mode (e.g. the F fields), one grid row each */
#define SEA_XTAIL_MAX_GROUPS 4
int sea_exchange_tail(const SeaExchangeTail* params, int n_groups, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * First half of the field MLP in one launch (bf16):  Hg = gelu_erf(LayerNorm_S(A[M,E] . W1[S,E]^T + b1) * lnw + lnb)
 * Replaces models/base_blocks.py:22-24 (Linear(E, S), nn.LayerNorm(S), GELU) = sea_gemm_grouped + sea_rownorm(gelu): a workgroup owns 32 complete
 * rows of the hidden matrix, the pre-activation matrix never reaches HBM.  (E, S) in {(256, 2048), (128, 1024)}; row strides multiples of 8;
 * pointers 16-byte aligned.  Returns SEA_EUNSUPPORTED for other shapes / dtypes (callers keep the two-launch form).
 */
#define SEA_MAX_MLP_GROUPS 8
typedef struct {
    const void* A;      /* act [M, E], row stride lda */
    const void* W1;     /* act [S, E], row stride ldw */
    const float* b1;    /* f32 [S] */
    const float* lnw;   /* f32 [S] */
    const float* lnb;   /* f32 [S] */
    void* Hg;           /* act [M, S], row stride ldh */
    int32_t lda, ldw, ldh;
    int32_t M, E, S;
    /* Optional prologue (X32 != NULL; A is then ignored): the operand rows are produced inside the launch from the fp32 residual stream —
     *     x = X32 (+ addend);  Xout <- x when non-NULL (may be X32);  A row = (x - mean) * rstd * (gamma [+ 1 + mod[0:E]]) + (beta [+ mod[E:2E]])
     * i.e. the info-bottleneck add and AdaLN_2 / LayerNorm in front of the MLP (models/temporal.py:139-145; SeaNormGroup's semantics, two-pass fp32
     * statistics): one launch less on the chain, the normalised rows never reach HBM. */
    const float* X32;    /* f32 [M, E], row stride ldx32 */
    const float* addend; /* f32 [M, E], row stride ldadd, or NULL */
    float* Xout;         /* f32 [M, E], row stride ldxout, or NULL */
    const void* mod;     /* act [M, 2E] (scale | shift), row stride ldmod, or NULL */
    const float* gamma;  /* f32 [E] */
    const float* beta;   /* f32 [E] or NULL */
    int32_t ldx32, ldadd, ldxout, ldmod;
    float norm_eps;
    int32_t pad_;
} SeaMlpGroup;

int sea_mlp_fc1_ln_gelu(const SeaMlpGroup* groups, int n_groups, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Second half of the field MLP, proj and the final norm in one launch (bf16; a workgroup owns 32 complete rows through both Linear layers):
 *     x3  = Hg[M,S] . W2[E,S]^T + b2 + R            (models/base_blocks.py:25; the residual of models/temporal.py:145)
 *     y   = x3 . Wproj[E,E]^T + bproj               (models/temporal.py:146)
 *     out = norm(y) with gamma / beta / mod as SeaNormGroup when gamma != NULL (the model's final per-field norm, models/temporal.py:412-415), else y
 * Replaces sea_gemm_grouped x 2 + sea_rownorm on the tail of the layer; x3 and y never reach HBM.  Same shapes and requirements as sea_mlp_fc1_ln_gelu.
 */
typedef struct {
    const void* Hg;      /* act [M, S], row stride ldh: the activated hidden rows */
    const void* W2;      /* act [E, S], row stride ldw2 */
    const float* b2;     /* f32 [E] */
    const float* R;      /* f32 [M, E], row stride ldr: the residual stream */
    const void* Wproj;   /* act [E, E], row stride ldwp */
    const float* bproj;  /* f32 [E] */
    const float* gamma;  /* f32 [E] or NULL (no norm) */
    const float* beta;   /* f32 [E] or NULL */
    const void* mod;     /* act [M, 2E] (scale | shift), row stride ldmod, or NULL */
    float* Y32;          /* f32 [M, E], row stride ldy32, or NULL */
    void* Yact;          /* act [M, E], row stride ldyact, or NULL */
    int32_t ldh, ldw2, ldr, ldwp, ldmod, ldy32, ldyact;
    int32_t M, E, S;
} SeaMlp2Group;

int sea_mlp_fc2_proj_norm(const SeaMlp2Group* groups, int n_groups, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * The whole field MLP, proj and the final norm in ONE launch (bf16): sea_mlp_fc1_ln_gelu(g1[i]) followed by sea_mlp_fc2_proj_norm(g2[i]) for every group i,
 * with the activated hidden rows kept in the owning workgroup's registers (models/base_blocks.py:22-25, models/temporal.py:139-146, 412-415): g1[i].Hg and
 * g2[i].Hg are ignored (may be NULL), g1[i].Xout must be NULL, and g2[i].R may be NULL when g1[i] carries the norm prologue — the residual is then
 * X32 (+ addend), formed by the same fp32 add as the prologue's.  g2[i].Y32 / Yact may alias g1[i].X32 (a workgroup writes its rows after its last read of
 * them).  Same shapes and requirements as the two entry points; g1[i].M == g2[i].M.  Returns SEA_EUNSUPPORTED for other shapes / dtypes.
 */
int sea_mlp_block(const SeaMlpGroup* g1, const SeaMlp2Group* g2, int n_groups, float eps, int dtype, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Hidden layer of the AdaLN condition MLP for a scalar condition: Hid[m, k] = silu(w1[k] * c[m] + b1[k]).
 * Replaces cond_mlp.0 (Linear(1, 2d)) + nn.SiLU (models/base_blocks.py:337-338, 344); cond_mlp.2 is a
 * sea_gemm_grouped group.  c: f32 [M]; w1, b1: f32 [K2]; Hid: act [M, K2] row stride ld.
 */
typedef struct {
    const float* w1;
    const float* b1;
    void* Hid;
    int32_t K2, ld;
} SeaSiluGroup;

int sea_silu_outer(const SeaSiluGroup* groups, int n_groups, const float* c, int M, int dtype, void* stream);
/* The same launch with up to SEA_MAX_SILU_IB extra row passes that EVALUATE the information-bottleneck MLP of a block (SeaIbParams below;
 * both depend on the condition only): ibs[k].X[0][m, :] = W2 . gelu(LN_h(w1 c[m] + b1)) + b2 is STORED (f32 [M, E], row stride ldx;
 * n_fields, c and drop of ibs[k] are ignored).  A later sea_rownorm adds it through SeaNormGroup.addend: no launch of its own for the add. */
#define SEA_MAX_SILU_IB 4
struct SeaIbParams_;
int sea_silu_outer_ib(const SeaSiluGroup* groups, int n_groups, const float* c, int M, int dtype, const struct SeaIbParams_* ibs, int n_ib, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Information-bottleneck add: ib = W2 . gelu(LN_h(w1 * c + b1)) + b2, then X_f += ib for every field f.
 * Replaces BaseBlockTemporal._add_info with ib_scale_mode='mlp', ib_addition_mode='add'
 * (models/temporal.py:111-116,140-142) = MLP(1 -> h -> E) of models/base_blocks.py:22-26 with h = scale_ratio.
 * X: n_fields f32 [M, E] matrices (row stride ldx); all parameters f32; h <= 64.
 */
typedef struct SeaIbParams_ {
    float* X[8];
    int32_t n_fields, ldx;
    const float* c;   /* [M] */
    const float* w1;  /* [h] (Linear(1,h).weight) */
    const float* b1;  /* [h] */
    const float* lnw; /* [h] */
    const float* lnb; /* [h] */
    const float* w2;  /* [E, h] */
    const float* b2;  /* [E] */
    int32_t M, E, h;
    SeaDropout drop;  /* on the MLP output, an independent stream (drop.stream + field) per field; element grid = (row, column) */
    int32_t mode;     /* which ib layer (models/temporal.py:103-109): 0 = 'mlp' (the fields above); 1 = 'linear': ib = w1[E] * c + b1[E] (nn.Linear(1, E));
                       * 2 = 'fourier': ib = [sin(2 pi c W), cos(2 pi c W)] with w1 = W[E/2] (GaussianFourierProjection, models/base_blocks.py:143-151).
                       * Modes 1 and 2 read only c, w1 (and b1), ignore h / lnw / lnb / w2 / b2 and take no dropout. */
    int32_t pad_;
} SeaIbParams;

int sea_ib_add(const SeaIbParams* params, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Strided 2-D conversions between fp32 and the activation dtype (weights shadow copy, model input split).
 * dst[r, c] = (dst type) src[r, c] for r < rows, c < cols (cols % 4 == 0).
 */
int sea_convert_f32_to_act(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype,
                           void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Mean-squared-error loss and its gradient in one pass: loss[0] = mean((out - tgt)^2),
 * dout = 2 * grad_scale / n * (out - tgt) (dout may be NULL for evaluation).  Replaces nn.MSELoss forward + backward
 * (train/train_temporal.py:221,256-257).  `partial` is an f32 workspace of n_partial_cap >= 1 elements (<= 1024 used); the
 * reduction order is fixed, so the loss is bitwise reproducible.  n % 4 == 0.
 */
int sea_mse_fwd_bwd(const float* out, const float* tgt, float* dout, float* loss, float* partial, int n_partial_cap,
                    int64_t n, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * relativeMSE (utils/train_utils.py:112-116) over the last dimension:
 * y[row] = sum_d (pred - truth)^2 / (sum_d truth^2 + 1e-8), rows = product of the leading dimensions, d % 4 == 0.
 */
int sea_relative_mse(const float* pred, const float* truth, float* y, int64_t rows, int d, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * One AdamW step over a flat fp32 parameter buffer (torch.optim.AdamW as configured by initialize_optimizer,
 * utils/train_utils.py:33-34: decoupled weight decay, bias-corrected moments; `step` counts from 1):
 *     g' = grad_scale * g;  p *= 1 - lr*wd;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 *     p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * and, in the same pass, the refreshed activation-dtype shadow of the parameters (shadow may be NULL).
 * grad_scale = 1/world_size folds the data-parallel mean into the step.  n % 4 == 0.
 */
int sea_adamw_flat(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ============================================================================================================
 * Backward pass.  The reference gets it from autograd (loss.backward(), train/train_temporal.py:257); here every
 * operator has a hand-written backward.  Data gradients of the Linear layers reuse sea_gemm_grouped with the W^T shadow
 * (sea_transpose_weights); parameter gradients are ACCUMULATED (fp32 atomics) into buffers the caller zeroes each step.
 * ============================================================================================================ */

/* Weight gradient of y = x W^T + b for several layers per launch: dW[N,K] += dY[M,N]^T . X[M,K]; db[N] += sum_m dY[m,:]
 * (db may be NULL).  The contraction over M is split across workgroups (unless M is short).  N % 8 == 0, K % 8 == 0.  At most 32 groups per call (weight gradients
 * have no consumers inside the backward: a caller may collect the small ones of a layer and run them as one call). */
typedef struct {
    const void* dY; /* act [M, N] row stride lddy */
    const void* X;  /* act [M, K] row stride ldx */
    float* dW;      /* f32 [N, K] row stride lddw, accumulated */
    float* db;      /* f32 [N] accumulated, or NULL */
    int32_t lddy, ldx, lddw;
    int32_t M, N, K;
    int32_t overwrite;   /* non-zero: dW holds zeros and this launch is its only contribution in the step — the kernel may STORE instead of adding
                          * (taken when the contraction is not split across workgroups; db is always accumulated) */
    int32_t pad_;
} SeaWgradGroup;
int sea_wgrad_grouped(const SeaWgradGroup* groups, int n_groups, int dtype, void* stream);

/* Activation-dtype TRANSPOSED shadow of weight matrices: for each descriptor {src_off, dst_off, rows, cols} (int64 x 4, in
 * elements, DEVICE memory) dst[dst_off + c*rows + r] = src[src_off + r*cols + c].  tile_start (int32 [n_desc+1], DEVICE) is the
 * prefix sum of ceil(rows/32)*ceil(cols/32). */
int sea_transpose_weights(const float* src, void* dst, int dst_dtype, const int64_t* desc, const int32_t* tile_start, int n_desc,
                          int total_tiles, void* stream);

/* Backward of sea_rownorm.  Given dY, the forward input X, the saved mean/rstd and the modulation:
 *   dX   = rstd * (dxhat - mean_d(dxhat) - xhat * mean_d(dxhat * xhat)),  dxhat = dY' * (gamma (+ 1 + mod_w)),
 *          dY' = dY * gelu'(y_pre) when gelu (beta is needed only then); written to dX32 (added to it when accumulate) and/or dXact;
 *   dmod = [dY' * xhat | dY'] (act, optional);  dgamma += sum_m dY' * xhat;  dbeta += sum_m dY' (dbeta may be NULL). */
typedef struct {
    const void* dY;     /* f32 or act [M, d] */
    const void* X;      /* f32 or act [M, d] */
    const void* mod;    /* act [M, 2d] or NULL */
    const float* gamma; /* [d] */
    const float* beta;  /* [d] or NULL */
    const float* mean;  /* [M] */
    const float* rstd;  /* [M] */
    float* dX32;
    void* dXact;
    void* dmod;
    float* dgamma;
    float* dbeta;
    int32_t lddy, ldx, ldmod, lddx32, lddxact, lddmod;
} SeaNormBwdGroup;
/* ws (optional, f32, >= n_groups * min(ceil(M/4), 512) * 2 * d floats): when given, the column sums are written as per-workgroup
 * partials and reduced by a second small launch instead of hundreds of workgroups atomically adding to the same addresses. */
int sea_rownorm_bwd(const SeaNormBwdGroup* groups, int n_groups, int M, int d, int dy_is_act, int x_is_act, int gelu,
                    int accumulate, int dtype, float* ws, int64_t ws_floats, void* stream);

/* Backward of sea_attention_fwd + the rotary embedding of sea_qkv_rope_grouped (flash-style: P is recomputed from Q, K and the
 * forward's LSE).  For each problem, from dO (gradient of the attention output, [B, Tq, H*hd] row stride lddo):
 *   dQ [B*Tq, H*hd] (row stride lddq), dK, dV [B*Tk, H*hd] (lddk, lddv) = gradients with respect to the OUTPUTS of the q / k / v
 *   Linear layers (RoPE and the q scale are undone in the epilogue), i.e. the dY operands of sea_wgrad_grouped / the dgrad GEMM.
 * V is the row-major copy written by sea_qkv_rope_grouped (Vout); delta (f32 [B, H, Tq]) is workspace.  Deterministic.
 * Q, LSE and q_scale are the forward's: Q in log2 units (q_scale = hd^-1/2 * log2 e is the factor sea_qkv_rope_grouped put on q), LSE base 2; the
 * kernels carry the ln 2 of d(2^s)/ds themselves. */
typedef struct {
    const void* Q;   /* act [B, H, Tq, hd]  (rotated, scaled) */
    const void* K;   /* act [B, H, cap, hd] (rotated) */
    const void* V;   /* act [B, H, cap, hd] */
    const void* O;   /* act [B, Tq, H*hd] row stride ldo */
    const void* dO;  /* act [B, Tq, H*hd] row stride lddo */
    const float* LSE;
    float* delta;
    void* dQ;
    void* dK;
    void* dV;
} SeaAttnBwdProblem;

typedef struct {
    SeaAttnBwdProblem p[SEA_MAX_ATTN_PROBLEMS];
    const float* rope; /* f32 [>= max(q_pos0 + Tq, Tk), hd/2, 2] */
    int32_t n_problems;
    int32_t B, H, hd, Tq, Tk, cap, q_pos0, src_len;
    int32_t ldo, lddo, lddq, lddk, lddv;
    float q_scale;
    SeaDropout drop;   /* must equal the forward's */
} SeaAttnBwdParams;
int sea_attention_bwd(const SeaAttnBwdParams* params, int dtype, void* stream);

/* Backward of sea_silu_outer: dpre = dHid * silu'(w1 c + b1); dw1[k] += sum_m dpre * c[m]; db1[k] += sum_m dpre. */
typedef struct {
    const void* dHid; /* act [M, K2] row stride ld */
    const float* w1;
    const float* b1;
    float* dw1;
    float* db1;
    int32_t K2, ld;
} SeaSiluBwdGroup;
int sea_silu_outer_bwd(const SeaSiluBwdGroup* groups, int n_groups, const float* c, int M, int dtype, float* ws, int64_t ws_floats,
                       void* stream);  /* ws: as for sea_rownorm_bwd, >= n_groups * min(ceil(M/4), 256) * 2 * max K2 floats */

/* Parameter gradients of the information-bottleneck MLP (sea_ib_add) from dib = sum_f dX_f; dX itself passes through. */
typedef struct {
    const float* dX[8];
    int32_t n_fields, ldx;
    const float* c;
    const float* w1;
    const float* b1;
    const float* lnw;
    const float* lnb;
    const float* w2;
    float* dw1;
    float* db1;
    float* dlnw;
    float* dlnb;
    float* dw2;
    float* db2;
    int32_t M, E, h;
    SeaDropout drop;
    int32_t mode;   /* 0: the MLP form above; 1: ib = nn.Linear(1, E) (ib_scale_mode 'linear', models/temporal.py:106-107): dw1 [E] += sum_m c[m] sum_f dX_f[m, :],
                     * db1 [E] += sum_m sum_f dX_f[m, :]; the other pointers are ignored */
    int32_t pad_;
    /* workspaces of the column-block form (mode 0, h <= 8; both optional — without them the one-wave-per-row kernel runs):
     *   ws   f32 [>= rs * E * (1 + h)]   per-row-split partial sums of db2 | dW2, rs = ws_floats / (E * (1 + h)) row splits at most
     *   dhid f32 [M, 8]                   d(loss)/d(hidden) per row; must be ZERO on entry when E > 256 (column blocks add into it) and is left zero */
    float* ws;
    int64_t ws_floats;
    float* dhid;
} SeaIbBwdParams;
int sea_ib_bwd(const SeaIbBwdParams* params, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * AdaLN whose modulation never reaches memory (bf16 compute): the normalisation is the epilogue of cond_mlp.2's GEMM.
 *     [w | b] = A[M, K] . W[2d, K]^T + bias            A = silu(cond_mlp.0(c)) rows (sea_silu_outer), W = cond_mlp.2.weight (scale rows 0..d-1, shift rows d..2d-1)
 *     y       = xhat * (gamma + 1 + w) + (beta + b),   xhat = (X - mean) / sqrt(var_biased + eps) over the d columns of a row of X
 * (models/base_blocks.py:337-350: AdaLN.forward) — instead of storing [w | b] as an [M, 2 d] matrix that a row-norm launch reads back.  A group WITHOUT X is
 * the plain GEMM: Yact [M, 2 d] receives [w | b] (modulations whose normalisation is another launch's epilogue: ln_cross); one launch carries both kinds.
 * mean / rstd (f32 [M]) are saved when non-NULL.  Requirements: dtype SEA_BF16; 16 <= d <= 1024, d % 8 == 0; K % 8 == 0; strides multiples of 8 (act) /
 * 4 (f32); pointers 16-byte aligned.  Returns SEA_EUNSUPPORTED for fp32.
 */
#define SEA_MAX_ADALN_GROUPS 16
typedef struct {
    const void* A;        /* act [M, K], row stride lda */
    const void* W;        /* act [2d, K], row stride ldw */
    const float* bias;    /* f32 [2d] or NULL */
    const float* X;       /* f32 [M, d], row stride ldx: the rows to normalise; NULL = plain group */
    const float* gamma;   /* f32 [d] */
    const float* beta;    /* f32 [d] or NULL */
    void* Yact;           /* act [M, d] (normalising group) or act [M, 2d] (plain group), row stride ldyact */
    float* Y32;           /* f32 [M, d], row stride ldy32, or NULL */
    float* mean;          /* f32 [M] or NULL */
    float* rstd;          /* f32 [M] or NULL */
    int32_t lda, ldw, ldx, ldyact, ldy32;
    int32_t M, d, K;
} SeaAdalnGroup;
int sea_gemm_adaln(const SeaAdalnGroup* groups, int n_groups, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Split-K finish: out = sum_s P[s] + bias * bias_scale + R, as fp32 (C32) and / or in the activation dtype (Cact) — the pass behind a sea_gemm_grouped
 * launch whose groups are K-slices of ONE Linear layer writing fp32 partial matrices (a few row tiles against a contraction of 16384: the MLP of
 * configs/multiphase_flow.py:112-141 at M = B T = 796).  The partial sums are added in split order.  N % 4 == 0; strides multiples of 4.
 */
typedef struct {
    const float* P;      /* f32 [S][M, N]: the partial matrices, p_stride elements apart, row stride ldp */
    const float* bias;   /* f32 [N] or NULL */
    const float* R;      /* f32 [M, N] row stride ldr, or NULL */
    float* C32;          /* f32 [M, N] row stride ldc32, or NULL */
    void* Cact;          /* act [M, N] row stride ldcact, or NULL */
    int64_t p_stride;
    int32_t S, M, N, ldp, ldr, ldc32, ldcact;
    float bias_scale;
} SeaSplitkGroup;
int sea_splitk_finish(const SeaSplitkGroup* groups, int n_groups, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * The front of a block in one launch (bf16, E = 256): the AdaLN condition MLP of AdaLN_0 with its hidden rows generated in place, AdaLN_0, and the
 * self-attention's q / k / v projections with the rotary epilogue —
 *     h = silu(w1 * cond[m] + b1)  [2E];   [w | b] = h . W2c^T + b2c;   y = xhat(X) * (gamma + 1 + w) + (beta + b)              (models/base_blocks.py:337-350)
 *     [q | k | v] = y . Wqkv^T + bqkv -> rotary embedding on q / k, q scale, Q [B,H,T,hd] / K [B,H,cap,hd] / Vt [B,H,hd,cap] as sea_qkv_rope_grouped (col0 = 0, N = 3E)
 * — a workgroup owns 32 rows from the caller's rows to the attention operands: replaces sea_silu_outer + sea_gemm_adaln + sea_qkv_rope_grouped for these modules
 * (no hidden matrix, no modulation matrix, no normalised rows in memory).  `riders`: plain bf16 groups (A, W, bias, Cact only; whole 64-wide K-tiles) of later
 * launches, run as 128 x 128 tiles by extra workgroups (sea_row_chain_riders' contract).  M = B * T rows per group; head dim 16 or 32, H * hd = E; cap % 8 == 0.
 * Returns SEA_EUNSUPPORTED for other shapes / dtypes.
 */
#define SEA_MAX_AQKV_GROUPS 4
typedef struct {
    const float* X;      /* f32 [M, E], row stride ldx */
    const float* cond;   /* f32 [M]: the condition scalar of a row */
    const float* w1;     /* f32 [2E]: cond_mlp.0.weight */
    const float* b1;     /* f32 [2E]: cond_mlp.0.bias */
    const void* W2c;     /* act [2E, 2E], row stride ldw2c: cond_mlp.2.weight (rows 0..E-1 scale, E..2E-1 shift) */
    const float* b2c;    /* f32 [2E] or NULL */
    const float* gamma;  /* f32 [E] */
    const float* beta;   /* f32 [E] or NULL */
    const void* Wqkv;    /* act [3E, E], row stride ldw: [Wq; Wk; Wv] */
    const float* bqkv;   /* f32 [3E] or NULL */
    void* Q;             /* act [B, H, T, hd] */
    void* K;             /* act [B, H, cap, hd], rows pos0 + t */
    void* Vt;            /* act [B, H, hd, cap], columns pos0 + t */
    int32_t ldx, ldw2c, ldw, M, E, N3;
    /* optional third layer (W3 != NULL; N3 = 256): mod3 = silu(w13 * cond[m] + b13) . W3^T + b3 — the modulation of another AdaLN module of the same rows
     * (ln_cross of the field, 2 D = 256 columns), stored for a later launch */
    const float* w13;    /* f32 [N3] */
    const float* b13;    /* f32 [N3] */
    const void* W3;      /* act [N3, N3], row stride ldw3 */
    const float* b3;     /* f32 [N3] or NULL */
    void* mod3;          /* act [M, N3], row stride ldmod3 */
    int32_t ldw3, ldmod3;
} SeaAdalnQkv;
/* silu / silu_c / silu_M, ib (both optional): row riders — hidden rows silu(w1 * silu_c[m] + b1) of up to 8 other modules (sea_silu_outer's groups) and the
 * information-bottleneck rows (sea_silu_outer_ib's SeaIbParams), evaluated by extra workgroups for launches further down the step. */
int sea_adaln_qkv(const SeaAdalnQkv* groups, int n_groups, const SeaQkvCommon* common, const SeaGemmGroup* riders, int n_riders, const SeaSiluGroup* silu, int n_silu,
                  const float* silu_c, int silu_M, const SeaIbParams* ib, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * A ROW-LOCAL CHAIN of the state-exchange block in one launch (bf16 compute; a workgroup owns 16 or 32 rows from an attention launch's output to the
 * next attention launch's operands; sea_amd/csrc/chain.hip).  Per group (field), in order:
 *   form A (n_seg = 0):   x = Xin + a2[M,E] . W2[E,E]^T (+ b2)                          self-attention out-projection + residual
 *                                                                                       (models/base_blocks.py:201, models/temporal.py:136)
 *   form B (n_seg >= 1):  g = sum_s gelu_erf(att_s[M,D] . Wp_s[D,D]^T);  x = Xin + g . W2[E,D]^T + bias_scale * b2
 *                                                                                       (cross_attn[i][j].projection + GELU, cross_up[i] of the sum, the residual:
 *                                                                                       models/temporal.py:183-191 — exactly sea_exchange_tail's first two layers)
 *   X <- x (f32; Xin and X may be the same rows, or Xin the caller's strided [B,T,F,E] tensor);  Xact <- (act) x when non-NULL
 *   has_down:             y = LayerNorm_D(x . Wd[D,E]^T + bd) with gamma / beta / mod as SeaNormGroup -> down.Yact / down.Y32 when non-NULL
 *                                                                                       (cross_down[i] + ln_cross[i], models/temporal.py:177-181)
 *   n_proj entries:       columns [col0, col0 + N) of the virtual [q | k | v] row of  y . W[N,D]^T + bias, rotary embedding on q / k, q scale, written in the
 *                         attention layouts exactly as sea_qkv_rope_grouped (proj[e].A / lda / M are ignored; K = D; an entry is a q projection — N = D, col0 = 0 — or
 *                         a k | v projection — N = 2 D, col0 = D):
 *                         every q of the field and the k / v of the pairs that read this field's rows at this point of the Gauss-Seidel sweep
 *                         (models/base_blocks.py:271-280; models/temporal.py:187-192)
 * `down` is read as in SeaExchangeTail.  Requirements: dtype SEA_BF16; (D, E) in {(128, 256), (64, 128)}; n_seg * D <= E; H * hd == D for the projections;
 * sum of the projections' N <= 1024; head dim 16 or 32 (16 for launches of more than 256 16-row tiles), col0 a multiple of 16; strides multiples of 8 (act) / 4 (f32); pointers
 * 16-byte aligned.  Returns SEA_EUNSUPPORTED for other shapes / dtypes (callers keep the separate launches).
 */
#define SEA_CHAIN_MAX_PROJ 6
#define SEA_CHAIN_MAX_GROUPS 3
typedef struct {
    const void* att[SEA_XTAIL_MAX_SEG];   /* act [M, D], row stride ldatt (form B) */
    const void* Wp[SEA_XTAIL_MAX_SEG];    /* act [D, D], row stride ldwp */
    const void* a2;                       /* act [M, E], row stride lda2 (form A) */
    const void* W2;                       /* act [E, K2], row stride ldw2; K2 = D (form B) or E (form A) */
    const float* b2;                      /* f32 [E] or NULL */
    const float* Xin;                     /* f32 [M, E], row stride ldxin: the residual rows */
    float* X;                             /* f32 [M, E], row stride ldx: the updated rows (may alias Xin) */
    void* Xact;                           /* act [M, E], row stride ldxact, or NULL */
    int32_t n_seg, ldatt, ldwp, lda2, ldw2, ldxin, ldx, ldxact;
    int32_t M, D, E, has_down, n_proj;
    float bias_scale;
    SeaGemmNormGroup down;
    SeaQkvGroup proj[SEA_CHAIN_MAX_PROJ];
} SeaRowChain;

/* params: n_groups <= SEA_CHAIN_MAX_GROUPS problems of one (D, E) (e.g. the F fields), one grid row each; common: rotary table, heads, positions of the projections
 * (may be NULL when no group has projections) */
int sea_row_chain(const SeaRowChain* params, int n_groups, const SeaQkvCommon* common, float eps, int dtype, void* stream);
/* The same launch carrying RIDERS: work of later launches that depends on nothing this one computes, run by extra workgroups on the CUs the chain leaves idle
 * (at one trajectory a chain launch is 127-192 workgroups on 256 CUs):
 *   riders   n_riders <= 8 plain groups as sea_gemm_grouped takes them (A, W, bias, Cact; bf16, K a multiple of 64: AdaLN's cond_mlp.2 of the modules the field
 *            MLP and the final norm read, models/base_blocks.py:339,344), cut into 128 x 128 tiles numbered group by group; THIS launch runs tiles
 *            [tile0, tile0 + n_tiles) — the caller spreads a GEMM over several chain launches;
 *   ib       NULL, or the information-bottleneck MLP whose rows ib->X[0][m, :] are STORED as sea_silu_outer_ib stores them (models/temporal.py:111-116).
 * Results are complete when the launch is; nothing in the same launch may read them. */
struct SeaIbParams_;
int sea_row_chain_riders(const SeaRowChain* params, int n_groups, const SeaQkvCommon* common, const SeaGemmGroup* riders, int n_riders, int tile0, int n_tiles,
                         const struct SeaIbParams_* ib, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Launch list: a whole plan (every launch of TemporalModel.forward, models/temporal.py:405-416, in order) replayed by ONE call, so the host
 * side of a replay is a C loop over prepared argument structs instead of one interpreter round trip per launch (KV-cache rollout steps
 * and plain, un-captured forwards are host-bound otherwise).  `op` selects the entry point, the other fields are its arguments:
 *     SEA_OP_GEMM   p0 = SeaGemmGroup[n]                          SEA_OP_QKV    p0 = SeaQkvGroup[n], p1 = SeaQkvCommon
 *     SEA_OP_ATTN   p0 = SeaAttnParams                            SEA_OP_NORM   p0 = SeaNormGroup[n], i0 = M, i1 = d, i2 = x_is_act, i3 = gelu, f0 = eps
 *     SEA_OP_SILU   p0 = SeaSiluGroup[n], p1 = c, i0 = M, l0 = (intptr) SeaIbParams[l1] or 0, l1 = n_ib      SEA_OP_IB     p0 = SeaIbParams
 *     SEA_OP_CONVERT p0 = src, p1 = dst, l0 = lds, l1 = ldd, l2 = rows, l3 = cols
 *     SEA_OP_GEMM_NORM p0 = SeaGemmNormGroup[n], f0 = eps          SEA_OP_XTAIL  p0 = SeaExchangeTail[n], f0 = eps
 *     SEA_OP_MLP1   p0 = SeaMlpGroup[n], f0 = eps
 *     SEA_OP_MLP2   p0 = SeaMlp2Group[n], f0 = eps
 *     SEA_OP_GEMM_FEW p0 = SeaGemmGroup[n], p1 = pre or NULL, i0 = pre_x_is_act, i1 = pre_gelu, f0 = eps
 *     SEA_OP_QKV_FEW  p0 = SeaQkvGroup[n], p1 = SeaQkvCommon, l0 = (intptr) pre or 0, f0 = eps
 *     SEA_OP_ADALN    p0 = SeaAdalnGroup[n], f0 = eps
 *     SEA_OP_MLPB     p0 = SeaMlpGroup[n], p1 = SeaMlp2Group[n], f0 = eps
 *     SEA_OP_SPLITK   p0 = SeaSplitkGroup[n]
 *     SEA_OP_AQKV     p0 = SeaAdalnQkv[n], p1 = SeaQkvCommon, l0 = (intptr) SeaGemmGroup[i0] riders or 0, l1 = (intptr) SeaSiluGroup[i1] or 0, l2 = (intptr) silu_c, i2 = silu_M,
 *                     l3 = (intptr) SeaIbParams or 0, f0 = eps
 *     SEA_OP_CHAIN    p0 = SeaRowChain[n], p1 = SeaQkvCommon or NULL, f0 = eps; riders (sea_row_chain_riders): l0 = (intptr) SeaGemmGroup[i0] or 0, i1 = tile0, i2 = n_tiles,
 *                     l1 = (intptr) SeaIbParams or 0
 * Returns 0, or the failing entry's error code with sea_last_error() set (entries before it have been launched).
 */
enum { SEA_OP_GEMM = 1, SEA_OP_QKV = 2, SEA_OP_ATTN = 3, SEA_OP_NORM = 4, SEA_OP_SILU = 5, SEA_OP_IB = 6, SEA_OP_CONVERT = 8, SEA_OP_GEMM_NORM = 9, SEA_OP_XTAIL = 10, SEA_OP_MLP1 = 11, SEA_OP_MLP2 = 13, SEA_OP_GEMM_FEW = 14, SEA_OP_QKV_FEW = 15, SEA_OP_CHAIN = 16, SEA_OP_ADALN = 17, SEA_OP_MLPB = 18, SEA_OP_AQKV = 19, SEA_OP_SPLITK = 20 };   /* 7 and 12 were round-1 entry points (sea_rowchain, sea_cond_mlp), removed in ABI v2 */
typedef struct {
    int32_t op, n, dtype, i0, i1, i2, i3;
    float f0;
    const void* p0;
    const void* p1;
    int64_t l0, l1, l2, l3;
} SeaLaunchRec;
int sea_run_list(const SeaLaunchRec* recs, int n_recs, void* stream);

/* The same list replayed for n_steps consecutive steps of a KV-cache rollout (the loop of utils/train_utils.py:202-209) without returning to the caller's
 * language between steps: before step s (s = step0 .. step0 + n_steps - 1) every patch writes  base + s * stride  into the HOST argument struct it points
 * into — the position of a SeaQkvCommon / SeaAttnParams (kind 0: int32), the row pointers of the step's input / condition / output (kind 1: 64-bit) —, then
 * the list is launched (kernel arguments are copied at launch, so the structs may change under launches still in flight). */
typedef struct {
    void* addr;          /* host address of the field (inside a struct a SeaLaunchRec points to, or inside a SeaLaunchRec of `recs`) */
    int32_t kind;        /* 0: int32 field, 1: 64-bit field (a device pointer) */
    int32_t pad_;
    int64_t base, stride;
} SeaStepPatch;
int sea_run_list_steps(const SeaLaunchRec* recs, int n_recs, const SeaStepPatch* patches, int n_patches, int step0, int n_steps, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Un-patchify + inverse MinMax scaling of decoded fields: the scatter of DataPartitioner2D.inverse_partition
 * (utils/data_processors.py:93-111) fused with MinMaxScaler.inverse_transform (:258-273) and with the [B,P,F,C] -> [B,P,C,F]
 * permute of the evaluation loop (utils/train_utils.py:223-226):
 *     out[b, index_map[p, c], f] = in[b, p, f, c] * scale[f] + shift[f]        for index_map[p, c] >= 0 (pad_id = -1 skipped)
 * `in` is addressed with element strides (sb, sp, sf, sc) so that both the decoder's [B,P,F,C] output and the reference's
 * [T,P,C,F] argument layout are read in place; out f32 [B, n_points, F] contiguous; index_map int32 [P, C]; scale, shift f32 [F].
 * Every mesh point belongs to exactly one (p, c), so the scatter needs no atomics and writes every output element once.
 * `point_slot` (optional, int32 [n_points], point_slot[index_map[p, c]] = p * C + c) selects the GATHER form: one thread per output
 * point, fully coalesced writes, reads served from L2 (one snapshot's decoded cells are ~1 MB) — 3x faster on a 30000-point mesh.
 */
int sea_unpatchify(const float* in, int64_t sb, int64_t sp, int64_t sf, int64_t sc, const int32_t* index_map, const int32_t* point_slot,
                   const float* scale, const float* shift, float* out, int B, int P, int F, int C, int n_points, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Patchify + MinMax scaling of mesh fields (forward direction): MeshProcessor._scale_fields
 * (utils/data_processors.py:528-536: MinMaxScaler.transform per field group, :241-247) followed by DataPartitioner2D.create_partitions /
 * pad_partitions (:60-92) and the stack / permute the encoder's input needs (:519-525):
 *     out[b, p, f, c] = in[b, index_map[p, c], f] * scale[f] + shift[f]     for c < C_map and index_map[p, c] >= 0
 *                     = pad_value                                           otherwise (empty slots; the row padding C_map .. C_out - 1)
 * `in` f32 [B, n_points, F] contiguous; index_map int32 [P, C_map] (pad_id = -1); `out` is addressed with element strides (sb, sp, sf, sc), so
 * both the encoder's [B, P, F, C_out] layout and the reference's [T, P, C, F] are written in place.  The pad value is NOT scaled (the reference
 * scales first and pads with pad_field_value afterwards).
 */
int sea_patchify(const float* in, const int32_t* index_map, const float* scale, const float* shift, float* out, int64_t sb, int64_t sp, int64_t sf,
                 int64_t sc, int B, int P, int F, int C_map, int C_out, int n_points, float pad_value, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Exact KV-cache rollout, the whole step loop in one call (utils/train_utils.py:202-209 around models/temporal.py:120-200, 398-417; one row per
 * call instead of the growing prefix).  Step k (position pos0 + k) reads traj[pos0 + k] ([B, F, E] f32), appends this position's keys / values to the
 * caches, attends over positions <= pos0 + k and writes traj[pos0 + k + 1].  Seven launches per layer and step (sea_amd/csrc/kvstep.hip): every
 * Linear is a GEMV of fp32 activation vectors against act-dtype weights; the q/k/v projections of a head ride in its attention workgroup; the
 * Gauss-Seidel tails of all fields are one launch whose workgroups hand the updated field over through tagged 8-byte granules.
 * Supported: exchange_mode 'sea' (exchange = 1, F >= 2) or 'simple' (exchange = 0), LN_type adaln / ln, any info-bottleneck mode (the caller
 * evaluates the term for all steps), src_len = 0, E <= 512, head dims 8 / 16 / 32 / 64, widths whose 16-byte chunk count is a power of two <= 64
 * or 128 / 256 / 512.
 * What depends on the condition only is evaluated by the CALLER for all steps before the call: SeaKvNorm.mod = act [n_rows, 2 d] (row (pos) * B + b:
 * AdaLN's cond_mlp output, models/base_blocks.py:337-344; NULL for LN_type 'ln'), SeaKvLayer.ib = f32 [n_rows, E] (the info-bottleneck term,
 * models/temporal.py:103-114; NULL for ib_addition_mode 'none').
 * Caches are act [B, H, cap, hd] for keys AND values (row-major values: an append is one row).  Workspaces are f32, sized in elements:
 * att_e, xr, xq, x3, xl[0], xl[1]: B F E; hbuf: B F S; nd_old: B F D; oc, qc: F (F-1) B D; ml: F (F-1) B H 2; handoff: B F D 8-byte words, zeroed once
 * by the caller; err: one int32, zeroed by the caller — non-zero after the call's work has completed means a hand-off wait gave up (results invalid).
 * tag0: the caller's running launch counter (each (step, layer) consumes one value; values must not repeat while `handoff` lives; 0 is never used).
 */
#define SEA_KV_MAX_FIELDS 4
typedef struct {
    const float* gamma;   /* f32 [d] */
    const float* beta;    /* f32 [d] or NULL */
    const void* mod;      /* act [n_rows, ldmod] (scale | shift) or NULL */
    int32_t ldmod, pad_;
} SeaKvNorm;
typedef struct {
    SeaKvNorm ln0, ln_cross, ln2;                  /* ln.exp.i.0, ln_cross.i, ln.exp.i.2 */
    const void* Wqkv; const float* bqkv;           /* act [3E, E] (q | k | v), f32 [3E] */
    const void* Wo;                                /* act [E, E], no bias */
    const void* Wdown; const float* bdown;         /* act [D, E] */
    const void* Wup; const float* bup;             /* act [E, D] */
    const void* W1; const float* b1;               /* act [S, E] */
    const float* lnw; const float* lnb;            /* f32 [S]: nn.LayerNorm(S) */
    const void* W2; const float* b2;               /* act [E, S] */
    const void* Wproj; const float* bproj;         /* act [E, E] */
    void* Ks; void* Vs;                            /* act [B, H, cap, E / H] */
} SeaKvField;
typedef struct {
    const void* Wq; const float* bq;               /* act [D, D] */
    const void* Wkv; const float* bkv;             /* act [2D, D] (k | v) */
    const void* Wp;                                /* act [D, D], no bias */
    void* Kc; void* Vc;                            /* act [B, H, cap, D / H] */
} SeaKvPair;
typedef struct {
    SeaKvField f[SEA_KV_MAX_FIELDS];
    SeaKvPair p[SEA_KV_MAX_FIELDS][SEA_KV_MAX_FIELDS];   /* p[i][j]: field i attends to field j (j != i) */
    const float* ib;
} SeaKvLayer;
typedef struct {
    int32_t F, E, D, S, H, B, L, cap, exchange, ib_after_cross;
    SeaKvNorm final_ln[SEA_KV_MAX_FIELDS];
    const float* rope_self;    /* f32 [max_len, E / H / 2, 2] (cos, sin) */
    const float* rope_cross;   /* f32 [max_len, D / H / 2, 2] */
    float* traj;               /* f32 [n_positions + 1, B, F, E] */
    float* xl[2];
    float* att_e; float* xr; float* xq; float* x3; float* hbuf; float* nd_old; float* oc; float* qc; float* ml;
    unsigned long long* handoff;
    int32_t* err;
    int64_t handoff_words;     /* size of `handoff` in 8-byte words: >= B F D; with >= sea_kv_arena_words() words, B = 1 and L = 1 the whole rollout runs as ONE
                                * persistent launch (every workgroup keeps the weights of its role in registers across steps) */
} SeaKvGlobal;

int sea_kv_rollout(const SeaKvGlobal* G, const SeaKvLayer* layers, int pos0, int n_steps, uint32_t tag0, int dtype, void* stream);
/* Words of `handoff` the persistent form of sea_kv_rollout needs for these sizes (host only). */
int64_t sea_kv_arena_words(const SeaKvGlobal* G);
/* Tuning aid: register a device buffer of n_steps * 64 8-byte words that the persistent form fills with 100 MHz clock stamps of its hand-offs
 * (tools/kv_persist_timeline.py); NULL switches it off. */
void sea_kv_debug_stamps(unsigned long long* buf);

/* ------------------------------------------------------------------------------------------------------------
 * Self-test of the MFMA fragment maps this library relies on (16x16x32 bf16 and 16x16x4 f32, A/B/C lane maps):
 * multiplies exact small-integer matrices on the device and checks every element on the host.  Synchronous.
 * Returns 0 when both maps are as documented in cdna_hip_programming.md §3.
 */
int sea_selftest_mfma(void);

#ifdef __cplusplus
}
#endif
#endif /* SEA_HIP_H */
