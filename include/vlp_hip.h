/* libvlp_hip.so -- C ABI of the MI355X (gfx950) hot path of LuoweiZhou/VLP.
 *
 * The reference has no FFI/plugin layer: its seam is the Python class API of
 * pytorch_pretrained_bert/modeling.py + optimization*.py, underneath which apex / cuBLAS / torch
 * CUDA kernels run.  This header is the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md) to replace those third-party kernels.  Each entry point cites the reference call
 * site(s) it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C linkage, POD argument structs, raw DEVICE pointers, no torch types;
 *   - all floating tensors are IEEE fp16 ("half") unless a field says f32; accumulation is fp32;
 *   - the caller owns every buffer (the library never allocates or frees device memory); scratch is
 *     passed in and sized by the matching *_workspace_bytes() query;
 *   - `stream` is a hipStream_t; launches are asynchronous, no entry point synchronises the device;
 *   - re-entrant and thread-safe (backward runs on autograd worker threads);
 *   - return value: VLP_OK (0) or a negative vlp_status; vlp_last_error_string() returns the
 *     thread-local message of the last failure.  Nothing throws or aborts across the ABI.
 *   - dropout masks are a pure function of (seed, rng_stream, element index), so backward entry
 *     points regenerate them from the same triple instead of reading a stored mask.
 */
#ifndef VLP_HIP_H
#define VLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLP_ABI_VERSION 5      /* 5 (round 6): + vlp_dec_gemm, vlp_dec_reduce_ln, vlp_argmax_rows2 (no existing struct changed) */

typedef enum {
    VLP_OK = 0,
    VLP_ERR_BAD_ARG = -1,     /* shape / alignment / null pointer / unsupported configuration */
    VLP_ERR_HIP = -2,         /* a HIP runtime call or kernel launch failed */
    VLP_ERR_WORKSPACE = -3    /* workspace too small */
} vlp_status;

enum { VLP_ACT_NONE = 0, VLP_ACT_GELU = 1, VLP_ACT_RELU = 2, VLP_ACT_TANH = 3,
       VLP_ACT_GELU_SAVE_GRAD = 4 /* y = gelu(z), and `preact` receives gelu'(z) instead of z (z = fp16-rounded pre-activation) */ };
enum { VLP_MUL_NONE = 0, VLP_MUL_GELU_GRAD = 1, VLP_MUL_RELU_MASK = 2, VLP_MUL_PLAIN = 3 /* y *= mul_src (a stored derivative) */ };

int vlp_version(void);
const char* vlp_last_error_string(void);
/* 1 when the library was compiled with -DVLP_LAB_BUILD (tools/build_variant_lib.sh): the investigation variants of vlp_gemm_nt (phased
 * kernels, k32 ring, further wave-pipelined configurations), the two-kernel / exchange-tile attention backward and the stream-K grouped
 * wgrad are then present; the product library (0) launches its selected kernels and the fallback rings only. */
int vlp_lab_build(void);
/* Test hook: owner lookups of device pointers by the calling thread and how many of them asked the driver (the entry guard caches the
 * owner per allocation). */
void vlp_debug_device_lookup_stats(unsigned long long* lookups, unsigned long long* driver_queries);

/* ------------------------------------------------------------------------------------------------
 * Y[M,N] = epilogue( alpha * X[M,K] . W[N,K]^T )                      (fp16 in/out, fp32 accumulate)
 * epilogue order:  + bias[n] -> (store preact) -> act -> * mul(mul_src) -> dropout -> + residual
 * Replaces every nn.Linear forward on the path (modeling.py:270-272, 314, 341 + gelu :62-67, 354,
 * 415, 432, 481, 1003-1005, 1016, 1027-1029) with its bias/activation/dropout/residual glue
 * (:315-316, :355-356), and -- called with a transposed weight shadow as W -- the dgrad GEMMs of
 * autograd's LinearBackward, with gelu'/relu' fused through mul_mode.
 * Requirements: K % 64 == 0; ld* % 8 == 0; ldy (ldr, ldp, ldm) >= roundup8(N); 16-byte aligned bases.
 * Columns in [N, roundup8(N)) of Y / preact are written as 0.
 */
typedef struct {
    const void* X; int64_t ldx;          /* [M,K] */
    const void* W; int64_t ldw;          /* [N,K] */
    void* Y; int64_t ldy;                /* [M,N] */
    const void* bias;                    /* [N] or NULL */
    const void* residual; int64_t ldr;   /* [M,N] or NULL */
    void* preact; int64_t ldp;           /* [M,N] or NULL: value after bias, before act (fp16) */
    const void* mul_src; int64_t ldm;    /* [M,N] operand of mul_mode or NULL */
    int32_t M, N, K;
    int32_t act;                         /* VLP_ACT_* */
    int32_t mul_mode;                    /* VLP_MUL_*: GELU_GRAD multiplies by gelu'(mul_src), RELU_MASK by (mul_src > 0), PLAIN by mul_src */
    float alpha;
    float dropout_p; uint64_t seed; uint32_t rng_stream;
    int32_t variant;                     /* tile shape / staging of the launch; every variant computes the same bits (ascending-k fp32 chains:
                                            profiles/r04_nt_variant_identity.json).  Product library:
                                              value       tile      staging                                   used for
                                              0           128x128   register-staged, 2 stages                 default / tiny problems
                                              1, 2        128x128   LDS-DMA double buffer / single buffer     M <= 1024 (decoder, heads)
                                              3, 4        256x128   LDS-DMA double / single buffer            A/B
                                              5           256x256   LDS-DMA double buffer (16 waves)          A/B
                                              17, 19      128x128 / 256x128  4- / 3-stage LDS-DMA ring        skinny M / N <= 1024 fallback
                                              21          256x256   2-stage ring, one raw barrier per k tile  wide outputs (N > 1024)
                                              64+1, 64+5  256x256 / 256x128  wave-pipelined (gemm_nt_wp.hip)  N <= 1024: 64 + 5
                                              256         256x128   persistent k stream (gemm_nt_ps.hip)      QKV forward
                                            + 8 on any value = XCD-aware tile (run) order: the table uses 27 = 19 + 8, 29 = 21 + 8, 77, 264.
                                            A wave-pipelined / persistent variant whose epilogue or shape is not carried (erf / tanh epilogues;
                                            N % 128, K <= 512, 32-bit offsets) runs on the rings 27 / 29 instead: vlp_gemm_nt_resolved_variant().
                                            Investigation variants (6 / 7 phased, 53 k32 ring, other wave-pipelined configurations) exist in
                                            -DVLP_LAB_BUILD libraries only (vlp_lab_build()); the product library maps them to 27 / 29. */
    const int32_t* row_map;              /* ABI 4: [M] or NULL.  Padding-free (packed) runs: the dropout element of output (m, n) is
                                            (row_map[m], n) -- the row's LOGICAL index b*L + l -- so a packed run draws the masks of the
                                            dense run bit for bit.  NULL: row m itself. */
} vlp_gemm_nt_args;
int vlp_gemm_nt(const vlp_gemm_nt_args* a, void* stream);
/* The variant the calling thread's last vlp_gemm_nt launched after its fallbacks (a wave-pipelined variant without the requested
 * epilogue, or with operands beyond 32-bit offsets, and a persistent variant outside its epilogues / shapes, run on the rings 27 / 29);
 * -1 before the first call. */
int vlp_gemm_nt_resolved_variant(void);
/* Split-K form for skinny problems (incremental decoding, M = 128..640 rows: only N/128 output tiles): the k range is cut into `splits`
 * slices computed by separate workgroups into an fp32 workspace of vlp_gemm_nt_splitk_workspace_bytes(M, N, splits) bytes; a second
 * kernel sums the slices in a fixed order (deterministic) and applies the same fused epilogue.  `variant` is ignored. */
int64_t vlp_gemm_nt_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splits);
int vlp_gemm_nt_splitk(const vlp_gemm_nt_args* a, int32_t splits, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * dW[N,K] (+)= A[M,N]^T . B[M,K]          (wgrad: A = dY, B = layer input; contraction over rows M)
 * Replaces autograd's weight-gradient GEMMs for every nn.Linear above.
 * beta = 0 overwrites C, beta = 1 accumulates into it (gradient accumulation,
 * run_img2txt_dist.py:566-576).  Requirements: N % 8 == 0 is NOT required (rows n >= N are skipped)
 * but lda >= roundup8(N); K % 8 == 0; ld* % 8 == 0; 16-byte aligned bases.
 * workspace: vlp_gemm_tn_workspace_bytes(M,N,K) bytes of scratch (fp32 split-M partial slabs).
 */
typedef struct {
    const void* A; int64_t lda;          /* [M,N] */
    const void* B; int64_t ldb;          /* [M,K] */
    void* C; int64_t ldc;                /* [N,K] */
    int32_t M, N, K;
    int32_t beta;                        /* 0 or 1 */
    void* workspace; int64_t workspace_bytes;
    int32_t variant;                     /* 0 = ds_read_u16 fragment gathers, 1 = ds_read_b64_tr_b16 (register-staged), 2 = ds_read_b64_tr_b16 +
                                            LDS-DMA staging with a transpose-read swizzle, 128x128 tile; 3 / 4 = the same with
                                            256x128 / 128x256 (n x k) tiles (5 = 256x256 was removed in round 2: rejected); +8 = XCD-aware tile order,
                                            +16 = split-major tile order (with +8: one XCD per row range) */
    int32_t splits;                      /* 0 = choose automatically */
    void* bias_out;                      /* optional [N] fp16: (+)= column sums of A, i.e. the bias gradient of the same
                                            Linear, fused into the k-tile-0 workgroups (replaces a separate vlp_colsum) */
} vlp_gemm_tn_args;
int64_t vlp_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K);
int vlp_gemm_tn(const vlp_gemm_tn_args* a, void* stream);
/* Several weight gradients in ONE launch (1..8 problems; the four Linears of a BertLayer: modeling.py:270-272, 314, 341, 354).
 * Every 128x128 output tile of every problem is one workgroup that walks that problem's WHOLE contraction: no split-M slabs, no
 * reduce launches (`splits`, `variant` of the entries are ignored); `beta` and `bias_out` as in vlp_gemm_tn.
 * Stream-K form (VLP_TN_GROUP_MODE=5): with list[0].workspace / workspace_bytes >= vlp_gemm_tn_grouped_workspace_bytes(tiles) (tiles =
 * sum over the problems of ceil(N/128) * ceil(K/128); must be a multiple of 6, one M for all problems) the tiles are taken six at a
 * time and their contraction is dealt to seven workgroups in equal runs: every tile is cut once, its two fp32 partial chains meet
 * through the workspace (flags reset by a hipMemsetAsync ahead of the launch); without a workspace the launch runs in the plain form. */
int vlp_gemm_tn_grouped(const vlp_gemm_tn_args* list, int32_t count, void* stream);
int64_t vlp_gemm_tn_grouped_workspace_bytes(int32_t tiles);

/* out[n] (+)= sum_m A[m,n]  -- bias gradients (autograd SumBackward of the broadcast bias add). */
typedef struct {
    const void* A; int64_t lda; int32_t M, N;
    void* out;                           /* [N] fp16 */
    int32_t beta;
    void* workspace; int64_t workspace_bytes;   /* vlp_colsum_workspace_bytes(M,N) */
} vlp_colsum_args;
int64_t vlp_colsum_workspace_bytes(int32_t M, int32_t N);
int vlp_colsum(const vlp_colsum_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused masked-softmax attention.  qkv is the packed projection [B, L, 3*H] (q | k | v, head h at
 * column h*64 of each third); mask is the byte mask [B, L, Lp] produced by vlp_mask_pack
 * (1 = attend, 0 = masked -> additive -10000 exactly like modeling.py:807-833, 2 = padding); Lp = roundup32(L).
 * Forward replaces modeling.py:279-302 (transpose_for_scores, QK^T/sqrt(d), +mask, softmax, dropout,
 * PV, permute+contiguous); backward replaces the autograd of those ops.  head_dim must be 64,
 * L <= 256.  lse [B, heads, L] f32 is the per-row log-sum-exp saved for backward.
 */
typedef struct {
    const void* qkv; int64_t ld_qkv;     /* [B*L, 3H] */
    const uint8_t* mask;                 /* [B, L, Lp] */
    void* ctx; int64_t ld_ctx;           /* [B*L, H] out */
    float* lse;                          /* [B, heads, L] out */
    int32_t B, L, heads;
    float scale;                         /* 1/sqrt(head_dim) */
    float dropout_p; uint64_t seed; uint32_t rng_stream;
    const int32_t* row_off;              /* ABI 4: [B+1] or NULL.  Padding-free layout: sample b owns rows [row_off[b], row_off[b+1]) of
                                            qkv / ctx (its first n_b = row_off[b+1] - row_off[b] <= L positions; the dropped positions
                                            must be inert: attended by no kept query -- vlp_amd.engine derives n_b from the mask).  mask,
                                            lse and the dropout element (b, head, q, key) stay LOGICAL ([B, L, ..]).  NULL: row b*L + l. */
} vlp_attn_fwd_args;
int vlp_attn_fwd(const vlp_attn_fwd_args* a, void* stream);

/* Inference form of the same kernel for incremental decoding (modeling.py:1189-1253 with BertSelfAttention's history path
 * :273-277): Lq new query rows per sequence attend to Lk cached + new key/value rows that live in a separate K/V cache
 * (a TRUE K/V cache: the reference re-projects K and V over the whole history every step).  Row (b, i) of q is at
 * q + (b*q_rows_per_batch + i)*ld_q (+ 64*head); rows of k / v likewise with kv_rows_per_batch and ld_kv.
 * mask: bytes [B, Lq, roundup32(Lk)] as produced by vlp_mask_pack on the [B, Lq, Lk] slice of the attention mask.
 * ctx [B*Lq, heads*64].  No dropout, no statistics. */
typedef struct {
    const void* q; int64_t ld_q; int64_t q_rows_per_batch;
    const void* k; const void* v; int64_t ld_kv; int64_t kv_rows_per_batch;
    const uint8_t* mask;
    void* ctx; int64_t ld_ctx;
    int32_t B, Lq, Lk, heads;
    float scale;
    /* beam search: key/value rows [0, n_prefix) of sequence b are read from a cache SHARED by the `beams` sequences of one sample
     * (row b / beams of k_prefix / v_prefix, kv-style [*, prefix_rows_per_batch, ld_kv] layout), rows >= n_prefix from k / v.
     * The prefix (regions + [SEP]) is identical for all beams (select_beam_items only permutes generated positions), so it is
     * stored and streamed once per sample.  n_prefix = 0: everything comes from k / v. */
    const void* k_prefix; const void* v_prefix;
    int64_t prefix_rows_per_batch;
    int32_t n_prefix, beams;
} vlp_attn_decode_args;
int vlp_attn_decode(const vlp_attn_decode_args* a, void* stream);

typedef struct {
    const void* qkv; int64_t ld_qkv;
    const uint8_t* mask;
    const uint8_t* mask_t;               /* [B, Lp, Lp] key-major copy of the byte mask (vlp_mask_pack's second output) */
    const void* ctx; int64_t ld_ctx;     /* forward output */
    const void* dctx; int64_t ld_dctx;   /* [B*L, H] gradient of ctx */
    const float* lse;
    void* dqkv; int64_t ld_dqkv;         /* [B*L, 3H] out: dq | dk | dv */
    float* delta;                        /* [B, heads, L] f32 scratch */
    int32_t B, L, heads;
    float scale;
    float dropout_p; uint64_t seed; uint32_t rng_stream;
    const int32_t* row_off;              /* ABI 4: packed rows of qkv / ctx / dctx / dqkv, see vlp_attn_fwd_args; delta, lse, masks stay logical */
} vlp_attn_bwd_args;
int vlp_attn_bwd(const vlp_attn_bwd_args* a, void* stream);

/* int64 [B,L,L] 0/1 (seq2seq_loader.py:292-304) -> uint8 [B,L,Lp]: 1 attend, 0 masked, 2 for the padding
 * columns >= L (excluded from the softmax). */
/* out_t (optional, [B,Lp,Lp]): the key-major copy used by vlp_attn_bwd -- row `key` holds the flags of all queries, 2 wherever key >= L
 * or q >= L.  ABI 3: inside a row the byte of query q = 16 t + 4 g + e sits at g * (Lp / 4) + 4 t + e (the order in which a lane of the
 * backward kernel consumes them: its words are consecutive); treat the buffer as opaque, produced here and by vlp_mask_build only. */
int vlp_mask_pack(const int64_t* mask, uint8_t* out, uint8_t* out_t, int32_t B, int32_t L, int32_t Lp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm, TF style (eps inside sqrt, biased variance, fp32 statistics): modeling.py:174-192 /
 * apex FusedLayerNorm.  y = dropout( gamma * (x - mean) * rstd + beta ).  H % 8 == 0, H <= 4096.
 */
typedef struct {
    const void* x; int64_t ldx;          /* [M,H] */
    const void* gamma; const void* beta; /* [H] */
    void* y; int64_t ldy;                /* [M,H] */
    float* mean; float* rstd;            /* [M] out (may be NULL for inference) */
    int32_t M, H; float eps;
    float dropout_p; uint64_t seed; uint32_t rng_stream;
    const int32_t* row_map;              /* ABI 4: [M] or NULL: dropout element of (m, c) is (row_map[m], c): packed rows, see vlp_gemm_nt_args */
} vlp_layernorm_fwd_args;
int vlp_layernorm_fwd(const vlp_layernorm_fwd_args* a, void* stream);

/* dx = LN'(x)·(drop(dy)); dgamma/dbeta (+)= column sums.  If dy_drop_p > 0 the incoming dy is first
 * multiplied by the forward's output-dropout mask.  If out_drop_p > 0 a second output
 * dx_drop = dx * mask(out_seed, out_stream) is written (the gradient that flows into the dropout
 * that FED the pre-LN sum: modeling.py:315-316, 355-356).
 */
typedef struct {
    const void* dy; int64_t lddy;
    const void* x; int64_t ldx;
    const void* gamma;
    const float* mean; const float* rstd;
    void* dx; int64_t lddx;
    void* dx_drop; int64_t lddxd;        /* optional */
    void* dgamma; void* dbeta;           /* [H] fp16 */
    int32_t M, H; int32_t beta;          /* beta: accumulate into dgamma/dbeta */
    float dy_drop_p; uint64_t dy_seed; uint32_t dy_stream;
    float out_drop_p; uint64_t out_seed; uint32_t out_stream;
    void* workspace; int64_t workspace_bytes;   /* vlp_layernorm_bwd_workspace_bytes(H) */
    int32_t defer_reduce;                /* 1: leave the per-block dgamma/dbeta partials in `workspace` and do NOT touch dgamma/dbeta;
                                            the caller reduces many LayerNorms at once with vlp_layernorm_bwd_reduce_batched */
    const int32_t* row_map;              /* ABI 4: [M] or NULL: both dropout masks are drawn at (row_map[m], c) (packed rows) */
} vlp_layernorm_bwd_args;
int64_t vlp_layernorm_bwd_workspace_bytes(int32_t H);
int vlp_layernorm_bwd(const vlp_layernorm_bwd_args* a, void* stream);
/* Second stage of `count` deferred vlp_layernorm_bwd calls in ONE launch (the 25 LayerNorms of a 12-layer step otherwise cost 25
 * tiny reduce launches): slot i of `parts` (stride = vlp_layernorm_bwd_workspace_bytes(H) / 4 floats) holds the partials of
 * LayerNorm i (all with the same M, H); dst[2*i] / dst[2*i+1] (device array of pointers) are its dgamma / dbeta ([H] fp16). */
int vlp_layernorm_bwd_reduce_batched(const float* parts, const void* const* dst, int32_t count, int32_t M, int32_t H, int32_t beta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * mask_image_regions / vis_pretext_loss (modeling.py:1049-1056, 1113-1131; loader: seq2seq_loader.py:267-269, 303-304).
 * vlp_region_mask_build: vis_masked_pos [B, Pm] (int64, values 1..Nv) -> out [B*Nv] bytes, 1 on the masked region rows.
 * vlp_pretext_fwd: per sample b, with r_i = vis_masked_pos[b,i] - 1:  A_i = vispe_h[b, r_i] + pooled[b]  (rounded to fp16 like the
 *   reference's in-place add :1124),  V_j = vis_h[b, r_j],  sim = A . V^T ([Pm, Pm], rounded to fp16 like the half matmul :1126),
 *   probs = softmax(sim) rows (fp32, kept for backward),  sample_loss[b] = -mean_i log probs[i,i];  loss[0] = mean_b sample_loss[b]
 *   (:1127-1131; summed in a fixed order: bitwise reproducible).  Pm <= 64.
 * vlp_pretext_bwd: gscale[0] = upstream gradient of the pretext loss (x loss scale).  dsim = gscale / (B Pm) (probs - I);
 *   dA = dsim . V, dV = dsim^T . A.  Writes ONLY the masked rows of d_vis_h / d_vispe_h (through ReLU + dropout exactly as
 *   vlp_embed_bwd does for the other rows: y > 0 mask and the dropout multiplier of element (region row, col) of streams
 *   vis_stream / vispe_stream), and d_pooled_pre[b] = (sum_i dA_i) * (1 - pooled[b]^2) (backward of the pooler's tanh, :416).
 */
int vlp_region_mask_build(const int64_t* vis_masked_pos, int32_t B, int32_t Pm, int32_t Nv, uint8_t* out, void* stream);
typedef struct {
    const void* vis_h; const void* vispe_h;                 /* f16 [B*Nv, H] (post ReLU + dropout) */
    const void* pooled;                                     /* f16 [B, H] */
    const int64_t* vis_masked_pos;                          /* [B, Pm] */
    float* probs;                                           /* f32 [B, Pm, Pm] out */
    float* sample_loss;                                     /* f32 [B] out */
    float* loss;                                            /* f32 [1] out */
    int32_t B, Nv, Pm, H;
} vlp_pretext_fwd_args;
int vlp_pretext_fwd(const vlp_pretext_fwd_args* a, void* stream);
typedef struct {
    const void* vis_h; const void* vispe_h; const void* pooled;
    const int64_t* vis_masked_pos;
    const float* probs;                                     /* from vlp_pretext_fwd */
    const float* gscale;                                    /* device f32 [1] */
    void* d_vis_h; void* d_vispe_h;                         /* f16 [B*Nv, H]: masked rows written */
    void* d_pooled_pre;                                     /* f16 [B, H] out */
    int32_t B, Nv, Pm, H;
    float drop_p; uint64_t seed; uint32_t vis_stream; uint32_t vispe_stream;
} vlp_pretext_bwd_args;
int vlp_pretext_bwd(const vlp_pretext_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding splice (modeling.py:217-236): pre[b,l,:] = word(l) + pos(l) + type[seg[b,l]] where for
 * l in [1, Nv] word(l) = vis_h[b,l-1] and pos(l) = vispe_h[b,l-1] (projected region features / box
 * encodings), else word = word_emb[input_ids[b,l]], pos = pos_emb[l].  (LayerNorm + dropout :239-240
 * are done by vlp_layernorm_fwd on `pre`.)
 */
typedef struct {
    const int64_t* input_ids; const int64_t* segment_ids;   /* [B,L] */
    const void* word_emb; const void* pos_emb; const void* type_emb;   /* [V,H] [P,H] [T,H] */
    const void* vis_h; const void* vispe_h;                 /* [B*Nv, H] */
    void* pre;                                              /* [B*L, H] out */
    int32_t B, L, Nv, H, vocab, type_vocab;
    const int64_t* position_ids;                            /* [B,L] or NULL (= 0..L-1); incremental decoding passes them (:856-865) */
    int32_t max_pos;                                        /* rows of pos_emb */
    const uint8_t* region_mask;                             /* ABI 3: [B*Nv] or NULL; 1 = this region row enters as zeros (word and position
                                                               stream; mask_image_regions, modeling.py:1049-1056), vlp_region_mask_build */
    const int32_t* row_map; int32_t rows;                   /* ABI 4: packed output: `pre` has `rows` rows, row p holds logical row row_map[p] = b*L + l
                                                               (input_ids / segment_ids / position_ids stay [B,L]).  NULL: rows = B*L, p = b*L + l */
} vlp_embed_fwd_args;
int vlp_embed_fwd(const vlp_embed_fwd_args* a, void* stream);

/* Backward of the splice: adds dpre into d_word_emb rows (no atomics: one owner per token id sums its rows in a fixed order in fp32, so
 * the result is bitwise reproducible), d_pos_emb, d_type_emb, and writes the region-row gradients.
 * d_vis_h = dpre * relu'(vis_h) * dropmask(vis), d_vispe_h = dpre * relu'(vispe_h) * dropmask(vispe) (backward of ReLU+Dropout,
 * modeling.py:1006-1007, 1017-1018).  acc32 is f32 scratch of vlp_embed_bwd_workspace_floats(B, L, Nv, H) floats; type_vocab <= 8; H <= 2048.
 */
int64_t vlp_embed_bwd_workspace_floats(int32_t B, int32_t L, int32_t Nv, int32_t H);
typedef struct {
    const void* dpre;                                       /* [B*L, H] */
    const int64_t* input_ids; const int64_t* segment_ids;
    const void* vis_h; const void* vispe_h;                 /* forward outputs (post ReLU+dropout) */
    void* d_word_emb; void* d_pos_emb; void* d_type_emb;    /* accumulated into (+=) */
    void* d_vis_h; void* d_vispe_h;                         /* [B*Nv, H] out */
    float* acc32;
    int32_t B, L, Nv, H, vocab, type_vocab;
    float drop_p; uint64_t seed; uint32_t vis_stream; uint32_t vispe_stream;
    const uint8_t* region_mask;                             /* ABI 3: [B*Nv] or NULL; rows with 1 got no signal from the encoder: their d_vis_h /
                                                               d_vispe_h rows are NOT written here (vlp_pretext_bwd owns them) */
    int32_t parts;                                          /* ABI 4: 0 = everything; 1 = region rows only (d_vis_h / d_vispe_h: what the region-
                                                               projection dgrad chain waits for); 2 = the three embedding tables only.  The two
                                                               halves read dpre and write disjoint outputs: a caller may run them on two streams. */
} vlp_embed_bwd_args;
int vlp_embed_bwd(const vlp_embed_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * small data-movement helpers
 */
/* dst[r, 0:cols_dst] = (beta ? dst : 0) + cast_f16(src[r, 0:cols_src]) ; columns >= cols_src are 0.
 * src_f32 != 0 reads fp32 input (features arrive as fp32 then .half(): run_img2txt_dist.py:466-468). */
int vlp_copy2d(const void* src, int64_t lds, int32_t src_f32, void* dst, int64_t ldd, int32_t rows,
               int32_t cols_src, int32_t cols_dst, int32_t beta, void* stream);
/* dst[c, r] = src[r, c] for r < rows, c < cols; dst columns in [rows, ldd) of rows c < cols_pad are 0
 * (weight shadows W^T for the dgrad GEMMs; zero padding keeps K % 64 == 0). */
int vlp_transpose(const void* src, int64_t lds, void* dst, int64_t ldd, int32_t rows, int32_t cols,
                  int32_t rows_pad, void* stream);
/* Batched form: n independent transposes in one launch.  descs / tile_start live in DEVICE memory; tile_start[i] is the
 * index of matrix i's first 64x64 tile in the concatenated list (tiles = ceil(rows_pad/64) * ceil(cols/64) per matrix),
 * total_tiles their sum.  Requirements per matrix: lds, ldd multiples of 8, 16-byte aligned bases. */
typedef struct {
    const void* src; void* dst;
    int64_t lds, ldd;
    int32_t rows, cols, rows_pad, reserved;
} vlp_transpose_desc;
int vlp_transpose_batched(const vlp_transpose_desc* descs_dev, const int32_t* tile_start_dev, int32_t n, int32_t total_tiles, void* stream);
/* out[i,:] = src[base(i / P) + pos[i], :]   (gather_seq_out_by_pos, modeling.py:1068-1069); base(b) = b * L, or row_off[b] when
 * row_off ([B+1], ABI 4) is given (packed rows; pos must then lie inside sample b's kept rows: clamped to them) */
int vlp_gather_rows(const void* src, int64_t lds, const int64_t* pos, void* out, int64_t ldo,
                    int32_t B, int32_t P, int32_t L, int32_t H, const int32_t* row_off, void* stream);
/* dst[base(i / P) + pos[i], :] += src[i, :]   (backward of the gather; fp16 packed atomics) */
int vlp_scatter_add_rows(const void* src, int64_t lds, const int64_t* pos, void* dst, int64_t ldd,
                         int32_t B, int32_t P, int32_t L, int32_t H, const int32_t* row_off, void* stream);
/* Padding-free (packed) row layout, ABI 4.  row_off [B+1] (int32, device): sample b keeps its first n_b = row_off[b+1] - row_off[b]
 * positions and owns the packed rows [row_off[b], row_off[b+1]).
 *   vlp_rowmap_build: row_map[row_off[b] + l] = b*L + l for l < n_b  (the logical index of every packed row);
 *   vlp_rows_unpack: dst[row_map[p], 0:H] = src[p, 0:H] for p < rows (dst rows that no packed row maps to are NOT touched: clear dst first);
 *   vlp_rows_pack:   dst[p, 0:H] = src[row_map[p], 0:H]. */
int vlp_rowmap_build(const int32_t* row_off, int32_t B, int32_t L, int32_t* row_map, void* stream);
int vlp_rows_unpack(const void* src, int64_t lds, const int32_t* row_map, int32_t rows, void* dst, int64_t ldd, int32_t H, void* stream);
int vlp_rows_pack(const void* src, int64_t lds, const int32_t* row_map, int32_t rows, void* dst, int64_t ldd, int32_t H, void* stream);
/* Incremental decoding helpers (modeling.py:1189-1253):
 *   vlp_mask_pack_rect: [B, Lq, Lk] slice of the int64 attention mask (element strides given) -> bytes [B, Lq, roundup32(Lk)];
 *   vlp_kv_append: cache[b, start + i, 0:2H] = qkv_new[b*T + i, H:3H]  (K | V of the new tokens into the K/V cache [B, Lcap, 2H]);
 *   vlp_argmax_rows: ids[r*ids_stride] = argmax_v logits[r, v] (first maximum), vals[r*vals_stride] = max  (greedy token choice :1228). */
int vlp_mask_pack_rect(const int64_t* mask, int64_t batch_stride, int64_t row_stride, uint8_t* out, int32_t B, int32_t Lq, int32_t Lk,
                       int32_t Lkp, void* stream);
int vlp_kv_append(const void* qkv_new, int64_t ld, void* cache, int32_t Lcap, int32_t B, int32_t T, int32_t start, int32_t H, void* stream);
int vlp_argmax_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int64_t* ids, int64_t ids_stride, float* vals, int64_t vals_stride,
                    void* stream);
/* Token-step kernels of the incremental decoder (round 6; csrc/decode.hip).  One token step of BertForSeq2SeqDecoder.forward
 * (modeling.py:1189-1253) runs BertLayer (:360-372) on M = sequences x T rows (T = 1-2 new positions): launch- and latency-bound, so a
 * Linear is ONE load burst per workgroup instead of a k loop, and the split-K reduce IS the LayerNorm launch.
 *   vlp_dec_gemm: Y[M, N] = act(X[M, K] . W[N, K]^T + bias)   (the Linears of :270-272, :314, :341, :354 on the new rows), M any, N any,
 *       K % (64 * splits) == 0, K / splits <= 768.
 *       splits == 1, slab == NULL: fp16 result.  If kv_cache != NULL, output columns >= kv_col0 (the K | V part of a packed QKV projection,
 *       kv_col0 % 32 == 0) are written to kv_cache[(m / kv_T) * kv_Lcap + kv_start + m % kv_T][n - kv_col0] (row pitch kv_ld) instead of Y:
 *       the K/V append of vlp_kv_append fused into the projection (a TRUE K/V cache in place of the hidden-state history of :273-277, 386-394).
 *       splits > 1 (or slab != NULL): raw fp32 partial sums of k slice s to slab[s][m][n] (pitch ldslab, ldslab % 8 == 0); bias / act / Y unused.
 *   vlp_dec_reduce_ln: Y[m, :] = LayerNorm( fp16( sum_s slab[s][m, :] + bias + residual[m, :] ) ) * gamma + beta  (:315-316, :355-356: dense
 *       output + bias, dropout p = 0, residual add, BertLayerNorm :188-192 with fp32 statistics); slabs summed in index order (deterministic).
 *       H in {256, 512, 768, 1024}. */
typedef struct {
    const void* X; int64_t ldx;          /* f16 [M, K] */
    const void* W; int64_t ldw;          /* f16 [N, K] */
    const void* bias;                    /* f16 [N] or NULL (splits == 1 only) */
    void* Y; int64_t ldy;                /* f16 [M, N] (splits == 1) */
    float* slab; int64_t ldslab;         /* f32 [splits][M][ldslab] (split form) or NULL */
    void* kv_cache; int64_t kv_ld;       /* f16 [sequences][kv_Lcap][kv_ld] or NULL */
    int32_t kv_col0, kv_Lcap, kv_T, kv_start;
    int32_t M, N, K, splits;
    int32_t act;                         /* VLP_ACT_NONE | VLP_ACT_GELU (splits == 1 only) */
    const void* residual; int64_t ldr;   /* f16 [M, N] or NULL (splits == 1): Y = act(X W^T + bias + residual), one rounding (modeling.py:315-316) */
    /* LayerNorm prologue (splits == 1, K <= 768): X holds PRE-LayerNorm rows; the kernel computes X' = LayerNorm(X) * gamma + beta (fp32 statistics,
     * one rounding, modeling.py:188-192) in LDS and multiplies X'.  ln_out (optional): X' [M, K] is also written there (the next residual). */
    const void* ln_gamma; const void* ln_beta; float ln_eps;
    void* ln_out; int64_t ld_ln_out;
} vlp_dec_gemm_args;
int vlp_dec_gemm(const vlp_dec_gemm_args* a, void* stream);
typedef struct {
    const float* slab; int64_t ldslab; int32_t splits;
    const void* bias;                    /* f16 [H] or NULL */
    const void* residual; int64_t ldr;   /* f16 [M, H] or NULL */
    const void* gamma; const void* beta; /* f16 [H] */
    float eps;
    void* Y; int64_t ldy;                /* f16 [M, H] */
    int32_t M, H;
} vlp_dec_reduce_ln_args;
int vlp_dec_reduce_ln(const vlp_dec_reduce_ln_args* a, void* stream);
/* vlp_argmax_rows2: vlp_argmax_rows with the index written to TWO destinations (the output column and the next step's input token,
 * :1228, 1248-1250; ids_b may be NULL) by one launch: 1024 threads per row, 16-byte loads (ld % 8 == 0, logits 16-byte aligned). */
int vlp_argmax_rows2(const void* logits, int64_t ld, int32_t rows, int32_t V, int64_t* ids_a, int64_t ids_a_stride, int64_t* ids_b, int64_t ids_b_stride,
                     float* vals, int64_t vals_stride, void* stream);
/* On-device input preparation (replaces per-sample CPU work of vlp/seq2seq_loader.py):
 *   vlp_mask_build: the packed attention masks (format of vlp_mask_pack, incl. the optional key-major copy) straight from the per-sample
 *       lengths, seq2seq_loader.py:292-301:  second_st = len(tokens_a)+2, second_end = len(tokens_a)+len(tokens_b)+3;
 *       s2s: attend(q,k) = k < st || (st <= q < en && st <= k <= q);  bi: attend(q,k) = k < en.  No int64 [B,L,L] tensor is shipped or read.
 *   vlp_vis_pe_prep: raw boxes [B,Nv,6] (x1,y1,x2,y2,-,confidence) + class probabilities [B*Nv, ld_cls] (f16 as stored on disk, or f32)
 *       -> the 6+n_cls encoding of seq2seq_loader.py:338-351 (corners / largest corner of the image, clamped relative area, layer norm of the 6
 *       box numbers and of the class probabilities, eps inside the sqrt) in f16, zero padded to pad_to columns = the K-padded operand of the
 *       vis_pe_embed GEMM (modeling.py:1016). */
/* ABI 3: region_mask ([B*Nv] bytes from vlp_region_mask_build, or NULL) additionally blocks the key columns 1..Nv of masked regions
 * for every query (mask_image_regions, seq2seq_loader.py:303-304). */
int vlp_mask_build(const int32_t* second_st, const int32_t* second_end, const int32_t* is_s2s, uint8_t* out, uint8_t* out_t, int32_t B, int32_t L,
                   int32_t Lp, const uint8_t* region_mask, int32_t Nv, void* stream);
typedef struct {
    const float* bbox;       /* [B, Nv, 6] f32 */
    const void* cls;         /* [B*Nv, ld_cls] f16 or f32 */
    int64_t ld_cls;
    void* out;               /* [B*Nv, ld_out] f16 */
    int64_t ld_out;
    int32_t B, Nv, n_cls, pad_to, cls_is_f32;
    float eps;               /* 1e-5 (F.layer_norm default) */
} vlp_vis_pe_prep_args;
int vlp_vis_pe_prep(const vlp_vis_pe_prep_args* a, void* stream);
/* sample_mode == 'sample' (modeling.py:1229-1235): ids[r*ids_stride] ~ Categorical(softmax(logits[r, :V])) drawn by the Gumbel-max trick
 * on the library's counter-based hash (seed, rng_stream, row, column) -- torch.multinomial's stream cannot be reproduced, the
 * distribution is the same; logp[r*logp_stride] = log_softmax(logits[r])[ids[r]]. */
int vlp_sample_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, uint64_t seed, uint32_t rng_stream, int64_t* ids,
                    int64_t ids_stride, float* logp, int64_t logp_stride, void* stream);
/* Beam search (modeling.py:1255-1494):
 *   vlp_logsoftmax_topk: per row log_softmax over V (fp32 from fp16 logits), -10000 added on forbidden words (uint8 [rows, V], may be
 *       NULL; :1298-1299), the eos column forced to -10000 while the minimum length is not reached (:1300-1301), then the K best
 *       (value descending, index ascending on ties; :1302) -> out_scores / out_ids [rows, K];
 *   vlp_beam_select: per sample the K best of the K*K continuations  kk + last_eos * -10000 + last_total  (:1308-1316), their back
 *       pointers, ids and eos flags; src_rows[b*K+k] = the cache row the beam continues (b in the first step, b*K+ptr later);
 *       next_ids receives the chosen ids with the given element stride (the decoder's next input column);
 *   vlp_kv_gather: dst[r, pos, :] = src[idx[r], pos, :] for pos in [lo, hi) -- first_expand / select_beam_items (:1325-1349) applied
 *       to the K/V caches instead of the reference's hidden-state history. */
int vlp_logsoftmax_topk(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const uint8_t* forbid, int32_t eos_id,
                        int32_t block_eos, float* out_scores, int64_t* out_ids, void* stream);
typedef struct {
    const float* kk_scores;   /* [rows, K]  rows = B (first) or B*K */
    const int64_t* kk_ids;    /* [rows, K] */
    const float* last_total;  /* [B, K] cumulative scores of the previous frame (NULL when first) */
    const float* last_eos;    /* [B, K] 1.0 where the previous frame's word was eos (NULL when first) */
    float* out_scores;        /* [B, K] */
    int64_t* out_ids;         /* [B, K] */
    int64_t* out_ptrs;        /* [B, K] */
    float* out_eos;           /* [B, K] */
    int64_t* src_rows;        /* [B*K] */
    int64_t* next_ids;        /* B*K elements, stride next_ids_stride */
    int64_t next_ids_stride;
    int32_t B, K, first;
    int64_t eos_id;
} vlp_beam_select_args;
int vlp_beam_select(const vlp_beam_select_args* a, void* stream);
int vlp_kv_gather(const void* src, int64_t src_rows_per_batch, void* dst, int64_t dst_rows_per_batch, const int64_t* idx, int32_t R, int32_t lo,
                  int32_t hi, int32_t row_elems, void* stream);
/* VQA fusion (modeling.py:1044,1138): out[b,:] = h[b,0,:] * h[b,Nv+1,:]; backward adds into dh rows.  row_off ([B+1] or NULL, ABI 4):
 * packed rows, sample b starts at row row_off[b] instead of b*L. */
int vlp_vqa_mul_fwd(const void* h, void* out, int32_t B, int32_t L, int32_t Nv, int32_t H, const int32_t* row_off, void* stream);
int vlp_vqa_mul_bwd(const void* h, const void* dout, void* dh, int32_t B, int32_t L, int32_t Nv, int32_t H, const int32_t* row_off, void* stream);
/* dz = dy * dropmask * (y > 0): backward of Linear->ReLU->Dropout given the layer OUTPUT y */
int vlp_relu_dropout_bwd(const void* dy, const void* y, void* dz, int64_t n, int64_t ncols, float drop_p,
                         uint64_t seed, uint32_t rng_stream, void* stream);

/* dz = dy * gelu'(z)   (backward of the head transform's gelu, modeling.py:433; n % 8 == 0) */
int vlp_gelu_bwd(const void* dy, const void* z, void* dz, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses
 * Masked-LM loss (modeling.py:1083-1111): per-row CE in fp32 over V logits, * masked_weights,
 * per-sample sum, keep the int(B*(1-ratio)) smallest samples, / (sum of kept weights + 1e-5), sum.
 * vlp_mlm_loss_fwd writes loss[0] (f32), per-row lse [B*P] and the per-row gradient coefficient
 * coef[B*P] = keep * weight / denominator; vlp_mlm_loss_bwd writes
 * dlogits[r, v] = grad_scale * coef[r] * (softmax(logits[r])[v] - [v == label[r]]) for v < V and 0 for
 * V <= v < ld_dlogits.
 */
typedef struct {
    const void* logits; int64_t ld_logits;    /* [B*P, V] fp16 */
    const int64_t* labels;                    /* [B*P] */
    const int64_t* weights;                   /* [B*P] masked_weights */
    float* loss;                              /* [1] out */
    float* lse; float* coef;                  /* [B*P] out */
    float* row_loss;                          /* [B*P] scratch */
    int32_t B, P, V;
    float drop_worst_ratio;
} vlp_mlm_loss_fwd_args;
int vlp_mlm_loss_fwd(const vlp_mlm_loss_fwd_args* a, void* stream);
typedef struct {
    const void* logits; int64_t ld_logits;
    const int64_t* labels;
    const float* lse; const float* coef;
    const float* grad_scale;                  /* [1] device scalar (upstream gradient x loss scale) */
    void* dlogits; int64_t ld_dlogits;        /* [B*P, ld] fp16 out */
    int32_t rows, V;
} vlp_mlm_loss_bwd_args;
int vlp_mlm_loss_bwd(const vlp_mlm_loss_bwd_args* a, void* stream);

/* VQA loss (modeling.py:1030,1140): BCEWithLogits(mean) * num_answers. fwd -> loss[0] (loss must hold
 * 257 floats: loss[1..257) is scratch);
 * bwd -> dlogits = grad_scale * (sigmoid(x) - y) / B, zero for columns >= N. */
int vlp_bce_loss_fwd(const void* logits, int64_t ld, const void* labels_f32, int64_t ldl, int32_t B, int32_t N,
                     float* loss, void* stream);
int vlp_bce_loss_bwd(const void* logits, int64_t ld, const void* labels_f32, int64_t ldl, int32_t B, int32_t N,
                     const float* grad_scale, void* dlogits, int64_t ldd, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizers
 * vlp_sumsq: out[0] = sum(g^2) in f32, out[1] = 1.0 if any element is inf/nan else 0 (the overflow
 * check + grad norm of apex FP16_Optimizer._compute_grad_norm).  partial = f32 scratch [2048].
 */
int vlp_sumsq(const void* g_f16, int64_t n, float* out2, float* partial, void* stream);
/* ABI 3: the same over another range, ADDED to out2 (sum += , flag = max): the sharded optimizer step (vlp_amd/distributed.py ShardPlan)
 * sums over the chunks a rank owns; launches of one stream run in order, so the total is reproducible. */
int vlp_sumsq_acc(const void* g_f16, int64_t n, float* out2, float* partial, void* stream);

/* apex fused_adam_cuda.adam as called by FusedAdam.step (run_img2txt_dist.py:411-420):
 *   g = g16 / (*combined_scale); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   denom = eps_inside_sqrt ? sqrt(v + eps) : sqrt(v) + eps;
 *   p32 -= step_size * (m / denom + decay * p32); p16 = half(p32)
 * hyper points to 3 device floats {combined_scale, step_size, skip}; when skip != 0 (overflow) the
 * kernel leaves all state untouched -- so the loss-scale logic never forces a host sync.
 */
typedef struct {
    float* p32; float* m; float* v;      /* master weights and moments [n] */
    const void* g16;                     /* [n] fp16 gradients (still multiplied by the loss scale) */
    void* p16;                           /* [n] fp16 model weights out */
    int64_t n;
    float b1, b2, eps, decay;
    int32_t eps_inside_sqrt;
    const float* hyper;                  /* device {combined_scale, step_size, skip} */
} vlp_fused_adam_args;
int vlp_fused_adam(const vlp_fused_adam_args* a, void* stream);
/* device-side scalar logic of FP16_Optimizer.step + FusedAdam.step for one param group:
 *   scale = scale_state[0]; norm = sqrt(sumsq[0]); overflow = max(sumsq[1], any_overflow[0]);
 *   clip = (norm/scale + 1e-6)/max_grad_norm; combined = scale * max(clip, 1); hyper = {combined, step_size, overflow} */
int vlp_adam_hyper(const float* sumsq2, const float* any_overflow, const float* scale_state, float max_grad_norm,
                   float step_size, float* hyper3, void* stream);
/* apex FP16_Optimizer._update_scale on the device, so the loss scale never forces a host sync.
 * scale_state (8 device floats) = {cur_scale, cur_iter, last_overflow_iter, scale_factor, scale_window, dynamic,
 * skipped_steps, reserved}: on overflow cur_scale = max(cur_scale / factor, 1), last_overflow_iter = cur_iter; otherwise
 * cur_scale *= factor whenever (cur_iter - last_overflow_iter) % window == 0; then cur_iter += 1. */
int vlp_loss_scale_update(float* scale_state, const float* overflow, void* stream);

/* BertAdam (optimization.py:112-182) over a flat fp32 master buffer made of `ntensors` tensors:
 * per-tensor L2 clip to max_grad_norm (:146-147), no bias correction, decoupled decay, lr already
 * scheduled by the host (warmup_linear :45-48).  seg_off[ntensors+1] (int64, device) are the tensor
 * boundaries inside the flat buffers.  g may be fp16 or fp32.
 * norms: f32 scratch of vlp_bert_adam_norms_floats(n, ntensors) floats (ABI 2; it was [ntensors] in ABI 1): [0, ntensors) receive the
 * per-tensor squared gradient norms, the rest holds one partial sum per (4096-element chunk, tensor) pair, which a second kernel adds
 * up in a fixed order -- no atomics: the clip factors, hence the update, are bitwise reproducible.
 */
int64_t vlp_bert_adam_norms_floats(int64_t n, int32_t ntensors);
typedef struct {
    float* p32; float* m; float* v;
    const void* g; int32_t g_is_f32;
    void* p16;                           /* optional fp16 copy out (NULL for pure-fp32 use) */
    const int64_t* seg_off; int32_t ntensors; int64_t n;
    float* norms; int64_t norms_floats;  /* ABI 3: size of `norms` in floats, checked against vlp_bert_adam_norms_floats() */
    float lr, b1, b2, eps, decay, max_grad_norm;
    float grad_scale;                    /* gradients are divided by this (loss scale), 1 for fp32 */
    const int32_t* active;               /* [ntensors] device flags or NULL: tensors with 0 are skipped entirely
                                            (the reference skips parameters whose .grad is None, optimization.py:125-126) */
} vlp_bert_adam_args;
int vlp_bert_adam(const vlp_bert_adam_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLP_HIP_H */
