/* medtsllm_hip.h — C-ABI of libmedtsllm_hip.so (MI355X / gfx950 kernels for the MedTsLLM hot path).
 *
 * The reference (flixpar/med-ts-llm @ 2024_10_08) has NO native/FFI layer: its hot path is PyTorch ATen
 * ops + HuggingFace `transformers` modules called from Python. Each entry point below
 * therefore names the reference Python call site (R: = /root/reference, HF: = transformers 5.15.0) whose
 * device arithmetic it replaces. Host side binds these with ctypes (med-ts-llm_amd/hip/_native.py) from
 * torch.autograd.Function.forward/backward — see INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*.
 *   - no allocation, no ownership transfer, no host synchronisation inside; the caller passes
 *     PyTorch-allocated outputs/workspaces and the stream (hipStream_t as void*).
 *   - bf16 tensors are raw uint16 storage; row-major; `ld*` are row strides in ELEMENTS.
 *   - return 0 on success, negative MTL_ERR_* otherwise (mtl_strerror gives the text).
 *   - re-entrant / thread-compatible: no mutable globals except the opt-in, mutex-guarded launch profiler, and NO environment variable is read
 *     by the product build (no getenv in the shipped .so: tests/test_host_logic.py checks the sources and the binary). Per-call A/B knobs are
 *     fields of the argument structs (mtl_gemm_args.tune_*, mtl_attn_fwd_args.tune). DIAGNOSTIC builds (tools/build_variant.sh ... -DMTL_DIAG;
 *     never shipped, reported by mtl_build_flags() and refused by hip/_native.py unless MTL_ALLOW_DIAG_LIB=1) additionally read, each once,
 *     the dispatch switches MTL_ATTN_WIDE, MTL_ATTN_WIDE_X, MTL_ATTN_WIDE_MIN, MTL_ATTN_XMAP, MTL_ATTN_W32, MTL_ATTN_W32_NW, MTL_ATTN_MERGED,
 *     MTL_ATTN_D128 (csrc/mtl_attention.hip), MTL_GEMM_RULES_OFF, MTL_GEMM_G, MTL_GEMM_FORCE, MTL_W4_ROT, MTL_W4_LINSRC (csrc/mtl_gemm.hip), MTL_ROPE_FUSE (csrc/mtl_backbone.hip).
 */
#ifndef MEDTSLLM_HIP_H
#define MEDTSLLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTL_ABI_VERSION 15

enum { MTL_OK = 0, MTL_ERR_ARG = -1, MTL_ERR_ALIGN = -2, MTL_ERR_UNSUPPORTED = -3, MTL_ERR_LAUNCH = -4,
       MTL_ERR_WORKSPACE = -5 };
enum { MTL_F32 = 0, MTL_BF16 = 1 };

int mtl_abi_version(void);
/* how this library was built: 0 = the product. Bit 0: -DMTL_DIAG (reads the diagnostic environment switches listed above); bit 1: a build that
 * computes WRONG results on purpose for timing ablations (MTL_DIAG_NOHASH, MTL_DIAG_ATTN_NODROP, MTL_DIAG_W4VAR ...). */
enum { MTL_BUILD_DIAG_ENV = 1, MTL_BUILD_DIAG_WRONG = 2 };
int mtl_build_flags(void);
const char* mtl_strerror(int code);

/* ------------------------------------------------------------------ patch tokeniser (a1-a4)
 * Replaces RevIN "norm" (R:models/layers/RevIN.py:37-56), permute (R:models/medtsllm.py:272),
 * ReplicationPad1d + unfold + reshape (R:models/layers/embed.py:160-163,188-191), TokenEmbedding Conv1d
 * k=3 circular, no bias (R:models/layers/embed.py:44-46) and the concat relayout (R:models/medtsllm.py:276-279).
 *   x        f32 [B, L, C]
 *   conv_w   f32 [d_patch, patch_len, 3]
 *   out      bf16; concat==0: [B*C, P, ld_out] (cols >= d_patch zero-filled up to ld_out)
 *                  concat==1: [B, P, ld_out]   (col = c*d_patch + o; cols >= C*d_patch zero-filled)
 *   mean, stdev  f32 [B, C]  (stdev = sqrt(biased var + eps), as RevIN stores them for the de-norm)
 * P = (L + stride - patch_len)/stride + 1 (== R:models/medtsllm.py:52 for the supported L).
 * drop_p > 0: PatchEmbedding's train-mode dropout (R:models/layers/embed.py:197) on the conv output, with the library's counter
 * mask of (drop_seed, row of `out`, column of `out`); the backward takes the same pair and regenerates it. */
int mtl_patch_tokenize_fwd(const float* x, const float* conv_w, void* out, float* mean, float* stdev,
                           int64_t B, int64_t L, int64_t C, int64_t patch_len, int64_t stride, int64_t d_patch,
                           int64_t ld_out, int concat, float eps, float drop_p, uint32_t drop_seed, void* stream);
/* dW of the token conv (x_enc needs no gradient; RevIN statistics are detached in the reference).
 *   dout  bf16, same layout as `out`;  partial f32 [B*C, d_patch*patch_len*3] workspace;  dw f32 [d_patch, patch_len, 3] */
int mtl_patch_tokenize_bwd(const float* x, const float* mean, const float* stdev, const void* dout, float* partial,
                           float* dw, int64_t B, int64_t L, int64_t C, int64_t patch_len, int64_t stride,
                           int64_t d_patch, int64_t ld_out, int concat, float drop_p, uint32_t drop_seed, void* stream);
/* int32 [P, patch_len] source-index map idx[p][j] = min(p*stride + j, L-1), produced by the SAME device
 * function the tokeniser uses (bit-exact parity target). */
int mtl_patch_index_map(int32_t* idx, int64_t L, int64_t patch_len, int64_t stride, void* stream);
/* RevIN "denorm" (R:models/layers/RevIN.py:58-69): out[b,t,c] = y[b,t,c]*stdev[b,c] + mean[b,c]; y/out f32 [B,T,C].
 * mean == NULL multiplies by stdev only (its backward: statistics are detached constants). */
int mtl_revin_denorm(const void* y, int y_dtype, const float* mean, const float* stdev, void* out, int out_dtype, int64_t B, int64_t T,
                     int64_t C, void* stream);   /* y, out: MTL_F32 or MTL_BF16 (the model's bf16 head output in, the bf16 gradient out: no cast passes) */

/* ------------------------------------------------------------------ bf16 MFMA GEMM (NT)
 * C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T + bias[N]),  fp32 accumulation on v_mfma_f32_16x16x32_bf16.
 * Replaces every nn.Linear / HF Conv1D / einsum-free matmul on the path: mapping_layer (R:models/medtsllm.py:281),
 * ReprogrammingLayer projections (R:models/medtsllm.py:571-573,579), embedding_downsample_layer (:358),
 * FlattenHead.linear (:550), GPT-2 c_attn/c_proj/c_fc (HF:models/gpt2/modeling_gpt2.py:185,238-243),
 * Llama q/k/v/o/gate/up/down (HF:models/llama/modeling_llama.py:174-176,254-256,280) and their dX / dW.
 * Requirements: K % 64 == 0, lda % 8 == 0, ldb % 8 == 0, A/B 16-byte aligned (callers zero-pad K). */
enum { MTL_EPI_STORE = 0,   /* C = v                                  (C bf16 or f32)                       */
       MTL_EPI_GELU = 1,    /* aux_out(bf16) = v ; C(bf16) = gelu_new(v)   (HF:activations.py:65-66)         */
       MTL_EPI_RESID = 2,   /* C(f32) = aux_in(f32) + v               (residual stream update)              */
       MTL_EPI_DGELU = 3,   /* C(bf16) = v * gelu_new'(aux_in(bf16))  (backward through the activation)     */
       MTL_EPI_ACCUM = 4,   /* C(f32) += v                            (gradient accumulation)               */
       MTL_EPI_SWIGLU = 5,  /* C(bf16) = v with columns INTERLEAVED (2j = gate_j, 2j+1 = up_j: B rows in that order);
                             * aux_out(bf16)[m, j] = silu(gate_j) * up_j   (HF:models/llama/modeling_llama.py:174-176) */
       MTL_EPI_DSWIGLU = 6 };/* its backward: v = d(act)[m, n]; aux_in(bf16)[m, 2n..2n+1] = saved (gate_n, up_n);
                             * C(bf16) has 2N columns: C[m, 2n] = d gate_n, C[m, 2n+1] = d up_n (ldc >= 2N, ldc % 8 == 0) */
typedef struct {
    const void* A; int64_t lda;      /* bf16 [M, K]                                                          */
    const void* B; int64_t ldb;      /* bf16 [N, K]                                                          */
    void* C; int64_t ldc; int c_dtype;
    int64_t M, N, K;
    /* optional A-row gather: logical row m reads physical row (m / a_group_rows)*a_group_stride + a_row_offset
     * + (m % a_group_rows); a_group_rows == 0 -> identity. Used to slice "the last n_patches tokens of every
     * sample" (R:models/medtsllm.py:353) without a copy. Same for C rows (c_group_*).                       */
    int64_t a_group_rows, a_group_stride, a_row_offset;
    int64_t c_group_rows, c_group_stride, c_row_offset;
    const float* bias;               /* f32 [N] or NULL                                                      */
    int epilogue;
    const void* aux_in; int64_t ld_aux_in;
    void* aux_out; int64_t ld_aux_out;
    float alpha;
    int split_k;                     /* >1: fp32 partial slabs in `workspace`, reduced by a second kernel    */
    void* workspace; size_t workspace_bytes;
    /* MTL_EPI_RESID only: C = aux_in + dropout(v) (GPT-2 resid_pdrop, HF:models/gpt2/modeling_gpt2.py:243,330,397):
     * keep mask = mtl counter hash of (drop_seed, physical C row, column) >= drop_p * 2^32, kept values / (1 - drop_p).
     * drop_p == 0 -> off. The backward regenerates the mask from the same triple (mtl_norm_bwd). */
    float drop_p; uint32_t drop_seed;
    /* MTL_EPI_GELU / MTL_EPI_SWIGLU only: the output that only a backward pass reads (GELU: aux_out, the saved pre-activation;
     * SWIGLU: C, the saved gate|up pre-activations) is written for row m only when (m % bwd_group_rows) >= bwd_first_row.
     * A pruned backward (mtl_backbone_bwd's n_grad) never reads the other rows; inference passes bwd_first_row = bwd_group_rows
     * and writes none. bwd_group_rows == 0 -> every row. */
    int64_t bwd_group_rows, bwd_first_row;
    /* Tile configuration of THIS call (A/B runs and tests; production passes zeros = the library's own choice). No process state:
     * tune_mode 0 automatic | 1 one output tile per workgroup | 2 persistent flat-K kernel; tune_bm tile rows (128, 256), tune_bn tile
     * columns (64, 96, 128, 192, 256), tune_stages LDS ring depth (2, 3), tune_waves waves per workgroup (4, 8, 16); each 0 = automatic.
     * Instantiated combinations: 128x64/4w/{2,3}, 128x96/4w/{2,3}, 128x96/8w/2 (two k-groups; needs at most one tile per CU and an even
     * number of 64-wide k-tiles >= 4), 128x128/{4,8}w/2, 128x128/8w/3, 128x192/8w/2, 256x96/8w/{2,3}, 256x128/16w/{2,3}, 256x192/8w/2,
     * 256x256/8w/2; anything else makes mtl_gemm_nt return MTL_ERR_UNSUPPORTED. Results agree in every configuration up to the fp32
     * summation order (two k-groups and their per-XCD k rotation change it). */
    int tune_mode, tune_bm, tune_bn, tune_stages, tune_waves;
} mtl_gemm_args;
size_t mtl_gemm_workspace_bytes(int64_t M, int64_t N, int split_k);
/* split_k the library would pick for a problem: > 1 only for few output tiles with a long K (the flatten head, small weight
 * gradients), where it spreads the K range over the idle CUs; the caller allocates the workspace and passes it in mtl_gemm_args. */
int mtl_gemm_auto_split_k(int64_t M, int64_t N, int64_t K, int epilogue);
int mtl_gemm_nt(const mtl_gemm_args* args, void* stream);
/* Transposed-operand GEMM: C[M, N] = alpha * sum_k A(m, k) B(n, k) with either operand stored K-MAJOR. The backward of the trainable
 * Linear layers (R:models/medtsllm.py:358,550,571-573,579) runs on what the forward already holds, without transposed copies:
 *     dW[n, k] = sum_m dY[m, n] X[m, k]      a_trans = b_trans = 1 (A = dY [rows, n], B = X [rows, k]); a_colsum = the bias gradient
 *     dX[m, k] = sum_n dY[m, n] W[n, k]      b_trans = 1 (B = W [n, k])
 * a_trans = 0: A is bf16 [M, K] (lda = row stride, K % 8 == 0); 1: bf16 [K, M] (lda >= M rounded up to 8). Same for B / N.
 * lda, ldb % 8 == 0, A / B 16-byte aligned. C f32 or bf16 [M, N]. At least one operand must be K-major (else: mtl_gemm_nt).
 * split_k > 1: the contraction is cut into split_k ranges, fp32 partial slabs in `workspace` (mtl_gemm_xt_workspace_bytes), summed
 * in a fixed order by a second kernel — deterministic. */
typedef struct {
    const void* A; int64_t lda; int a_trans;
    const void* B; int64_t ldb; int b_trans;
    void* C; int64_t ldc; int c_dtype;
    int64_t M, N, K;
    float alpha;
    float* a_colsum;                 /* optional f32 [M]: sum_k A(m, k); a_trans = 1 only                    */
    int split_k; void* workspace; size_t workspace_bytes;
} mtl_gemm_xt_args;
size_t mtl_gemm_xt_workspace_bytes(int64_t M, int64_t N, int split_k);
int mtl_gemm_xt_auto_split_k(int64_t M, int64_t N, int64_t K);
int mtl_gemm_xt(const mtl_gemm_xt_args* args, void* stream);
/* Measurement aid (bench.py's roofline legs; off by default, no effect on results): while enabled, every GEMM, attention, norm and
 * optimiser launch carries its own start / stop event pair (hipExtLaunchKernelGGL), whose elapsed time is the kernel's begin -> end
 * on the device — what rocprofv3 --kernel-trace reports for the same dispatch. mtl_prof_read aggregates per kernel instance:
 * `name` as rocprofv3 prints it (without the anonymous-namespace prefix and the parameter list), launches, total / min / max
 * duration, and the launches' ALGORITHMIC work: FLOPs (kind 0: 2*M*N*K for a GEMM, full-rectangle 4*T*T*d convention for
 * attention) or HBM bytes (kind 1). Enabling clears the records. Not to be read concurrently with launches. */
typedef struct mtl_prof_row {
    char name[128];
    int32_t kind;            /* 0: total_work in FLOPs (MFMA family); 1: in bytes (HBM family) */
    int64_t launches;
    double total_ms, min_ms, max_ms, total_work;
} mtl_prof_row;
int mtl_prof_enable(int on);
int mtl_prof_read(mtl_prof_row* rows, int cap);
/* mtl_prof_enable(2): as 1, and the GEMM rows are kept per problem size (" [MxNxK]" appended to the kernel name). */
/* Host-only (no device call): the tile order the persistent GEMM would use for a grid of tiles_m x tiles_n tiles of bm x bn with
 * per_cu resident workgroups per CU: bits 0-7 rows of a tile group, bit 8 per-XCD k rotation, bit 9 per-XCD column rotation
 * (negative: error code). Lets the CPU test suite check that every order visits every tile exactly once. */
int mtl_gemm_tile_order(int tiles_m, int tiles_n, int bm, int bn, int per_cu, int64_t K, int one_tile_per_wg);

/* ------------------------------------------------------------------ prompt input statistics (a6 / f2)
 * Replaces the five reductions + rFFT autocorrelation of `build_input_stats_prompt` / `calcute_lags` (R:models/medtsllm.py:476-481,
 * 530-538) and their five `.tolist()` syncs: per sample and selected channel the minimum, maximum, LOWER median (torch.median) and
 * trend (1 if the summed first differences are > 0), and per sample the top `n_lags` lags of the channel-mean circular
 * autocorrelation, evaluated directly (exactly symmetric; ties towards the smaller lag — the reference's order inside a twin pair
 * lag / L - lag is FFT round-off noise).
 *   x          f32 [B, L, C]
 *   channel    >= 0: that channel only (n_channels = 1);  -1: all C channels (statistics per channel, lags of the channel mean)
 *   out_stats  f32 [B, n_channels, 4] = (min, max, median, trend)
 *   out_lags   f32 [B, n_lags]  (integers, exact in fp32)
 * out_stats and out_lags may be the two halves of one buffer, so that ONE device-to-host copy fetches everything. */
size_t mtl_input_stats_workspace_bytes(int64_t B, int64_t L, int64_t n_channels);
int mtl_input_stats(const float* x, float* out_stats, float* out_lags, void* workspace, size_t workspace_bytes, int64_t B, int64_t L,
                    int64_t C, int64_t channel, int64_t n_lags, void* stream);

/* ------------------------------------------------------------------ layout / cast helpers
 * f32 [R, Cc] (ld_src) -> bf16 [R, ld_dst] zero-padding cols >= Cc; optionally also the transpose
 * dst_t bf16 [Cc, ld_dst_t] (zero-padded cols >= R). Used once per step on trainable fp32 master weights
 * (autocast's weight cast, tasks/forecasting.py:22) and on gradients. */
int mtl_cast_pad_f32_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t,
                          int64_t R, int64_t Cc, void* stream);
/* bf16 [R, Cc] (ld_src) -> bf16 [Cc, ld_dst] transpose, zero-padding cols >= R up to ld_dst. */
int mtl_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t R, int64_t Cc, void* stream);
/* the same, and colsum f32 [Cc] = column sums of src (zeroed inside): a Linear's dY^T for the dW GEMM and its bias gradient in one pass */
int mtl_transpose_colsum_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, float* colsum, int64_t R, int64_t Cc, void* stream);
/* elementwise casts of contiguous buffers */
int mtl_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int mtl_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* column sums of a bf16 [R, Cc] matrix into f32 [Cc] (bias gradients) */
int mtl_colsum_bf16(const void* src, int64_t ld_src, float* dst, int64_t R, int64_t Cc, void* stream);
/* Channel mixing of the non-concat covariate modes: y[b, r, o] = bias[o] + sum_{j, c} W[o, j*C + c] * x[b, c, r, j], x bf16 [B, C, R, J],
 * y bf16 or f32 [B, R, O]; W == NULL: the plain mean over the channels (J = O = 1). Replaces float().mean(dim = 1) of "add" /
 * "independent" (R:models/medtsllm.py:286,371) and the feature_weighting Linear on the channel-last view of "weighted-average" /
 * "merge-end" (:288-291,373-375). fp32 arithmetic, one rounding at the output. Backward: dx bf16 (NULL: skip), dW f32 [O, J*C], dbias
 * f32 [O] (NULL: skip) through `workspace` (mtl_channel_mix_workspace_bytes), summed in a fixed order. */
int mtl_channel_mix_fwd(const void* x, const float* W, const float* bias, void* y, int y_dtype, int64_t B, int64_t C, int64_t R, int64_t J,
                        int64_t O, void* stream);
size_t mtl_channel_mix_workspace_bytes(int64_t C, int64_t J, int64_t O);
int mtl_channel_mix_bwd(const void* x, const float* W, const void* dy, int dy_dtype, void* dx, float* dW, float* dbias, void* workspace,
                        int64_t B, int64_t C, int64_t R, int64_t J, int64_t O, void* stream);
/* dst[r] = sum_c src[r, c] (bf16 in, fp32 out): the mapping layer's bias gradient (row sums of d source, R:models/medtsllm.py:281) */
int mtl_rowsum_bf16(const void* src, int64_t ld_src, float* dst, int64_t R, int64_t Cc, void* stream);

/* ------------------------------------------------------------------ optimiser step
 * torch.optim.Adam / AdamW (R:tasks/base.py:97,99; stepped at R:tasks/forecasting.py:27 and the 4 other task loops)
 * for every trainable tensor in one launch per MTL_ADAM_MAX_TENSORS tensors: fp32 p/g/m/v, bias-corrected with
 * `step` (1-based, after increment, as torch does). weight_decay: decoupled = 0 -> Adam's L2 (g += wd*p),
 * 1 -> AdamW (p *= 1 - lr*wd). `shadow` (optional) receives the bf16 copy of the UPDATED p viewed as
 * [n / cols, cols] with row stride ld_shadow (the autocast weight copy the next forward needs). */
#define MTL_ADAM_MAX_TENSORS 24   /* (the table travels as a kernel argument: 24 x 72 B) */
typedef struct mtl_adam_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    void* shadow;        /* bf16 or NULL */
    int64_t cols;        /* used with shadow */
    int64_t ld_shadow;
    int64_t param_dtype; /* MTL_F32 (default) | MTL_BF16: p and g are bf16 storage (setup.dtype = "bf16", R:tasks/base.py:261-262 casts the
                          * whole model); m and v stay fp32, the update is formed in fp32 and p is rounded once (RNE) */
} mtl_adam_tensor;
int mtl_adam_step(const mtl_adam_tensor* tensors, int count, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int decoupled, int64_t step, void* stream);

/* ------------------------------------------------------------------ attention (flash-style, never materialises scores)
 * O = softmax(scale * Q K^T [+ causal mask]) V per (batch, head); fp32 softmax statistics, bf16 I/O.
 * Replaces (a) the reprogramming cross-attention einsum/softmax/einsum (R:models/medtsllm.py:586-589;
 * no mask, K/V shared by every batch element -> k_bs = v_bs = 0) and (b) the backbone's eager causal
 * attention (HF:models/gpt2/modeling_gpt2.py:54-72, HF:models/llama/modeling_llama.py:191-213 incl. GQA repeat_kv).
 * Strides are in elements: *_bs batch, *_ts token, *_hs head. head_dim D in {32, 64, 128}.
 * lse f32 [B, Hq, Tq] (natural-log sum-exp of the scaled scores), saved for the backward. */
typedef struct {
    const void* q; int64_t q_bs, q_ts, q_hs;
    const void* k; int64_t k_bs, k_ts, k_hs;
    const void* v; int64_t v_bs, v_ts, v_hs;
    void* o; int64_t o_bs, o_ts, o_hs;
    float* lse;
    int64_t B, Hq, Hkv, Tq, Tk, D;
    float scale;
    int causal;
    int64_t causal_off;   /* causal: key k is visible to query q iff k <= q + causal_off (0 for Tq == Tk; Tk - Tq aligns bottom-right) */
    int64_t stat_stride;  /* elements between consecutive (batch, head) rows of lse / delta; 0 -> Tq                          */
    /* attention dropout A = dropout(softmax(.)) of the reprogramming layer (R:models/medtsllm.py:588, p = training.dropout);
     * non-causal only. keep(b*Hq+h, q, key) is a counter-based hash of dropout_seed, regenerated identically in the backward. */
    float dropout_p; uint32_t dropout_seed;
    /* optional fp32 copy of the output (same element strides as o), non-causal kernels only. The backward then takes
     * delta_q = dO_q . O_q from it: with the bf16-rounded O the error of delta is coherent over the keys when the probabilities
     * are near-uniform (reprogramming attention); with the fp32 O it is the rounding of the individual P V products,
     * sqrt(Tk) times smaller, and the extra pass over K / V that the bf16 route needs to rebuild delta is not run. */
    float* o_f32;
    /* per-call kernel selection for A/B runs and tests (0 = the library's own choice; no process state): bit 0 = never the K/V-resident
     * kernels (whole head in LDS, no barrier in the key loop) — the chunked ones; bit 1 = the resident backward of hd-64 MHA heads as two
     * launches (dQ, then dK / dV) instead of the merged one. Results agree to rounding (bit 1: bit-identical). */
    int tune;
} mtl_attn_fwd_args;
int mtl_attention_fwd(const mtl_attn_fwd_args* a, void* stream);
typedef struct {
    mtl_attn_fwd_args f;            /* same tensors as the forward (o = forward output, lse = saved)        */
    const void* dout; int64_t do_bs, do_ts, do_hs;
    void* dq; int64_t dq_bs, dq_ts, dq_hs;
    void* dk; int64_t dk_bs, dk_ts, dk_hs;   /* with k_bs == 0 the batch is reduced into dk/dv              */
    void* dv; int64_t dv_bs, dv_ts, dv_hs;
    float* delta;                   /* f32 [B, Hq, Tq] workspace: rowsum(dO * O)                            */
    int64_t kv_row0;                /* dK/dV are produced only for keys >= kv_row0 (keys whose gradient is dead are skipped) */
    float* dkv_ws; int64_t kv_splits; /* batch-shared K/V only: fp32 [kv_splits, 2, Tk, Hkv, D] partial slabs; the batch is
                                       split into kv_splits chunks summed by a second kernel (NULL / <=1: one pass)  */
    /* optional INVERSE rotary embedding of the query / key gradients (the backward of the forward's RoPE on q and k,
     * HF:models/llama/modeling_llama.py:151-153) in the store epilogues: dq row r is position r + f.causal_off, dk row k position k;
     * rope_cos / rope_sin f32 [positions, D]. = mtl_rope_inplace(inverse) on dq and dk, bit for bit. Not with batch-shared K/V. */
    const float* rope_cos; const float* rope_sin;
} mtl_attn_bwd_args;
int mtl_attention_bwd(const mtl_attn_bwd_args* a, void* stream);


/* ------------------------------------------------------------------ norms (fp32 statistics)
 * LayerNorm eps 1e-5 (HF:models/gpt2/modeling_gpt2.py:252,254,497) and LlamaRMSNorm
 * (HF:models/llama/modeling_llama.py:62-67). x f32 [M, d] (the fp32 residual stream of dtype="mixed"),
 * y bf16 [M, d] (ld_y). rms != 0 -> RMSNorm (beta ignored). stats f32 [M, 2] = (mean, rstd).
 * Row gather like the GEMM: logical row m -> (m / group_rows)*group_stride + row_offset + m % group_rows.
 * y rows are compact (logical); stats rows are physical (gathered) when stats_physical != 0, logical otherwise. */
int mtl_norm_fwd(const float* x, const float* gamma, const float* beta, void* y, int64_t ld_y, float* stats,
                 int64_t M, int64_t d, float eps, int rms, int64_t group_rows, int64_t group_stride,
                 int64_t row_offset, int stats_physical, void* stream);
/* the same with the residual stream's dtype as an argument: x_dtype MTL_F32 (above) or MTL_BF16 — the reference's setup.dtype = "bf16"
 * (R:tasks/base.py:261-262: the whole model in bf16, i.e. a bf16 residual stream through the HF stack). bf16 stream: statistics still fp32, and
 * RMSNorm rounds the normalised row to bf16 before the weight multiplies it, as HF:models/llama/modeling_llama.py:64-69 does for bf16 inputs. */
int mtl_norm_fwd_t(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int64_t ld_y, float* stats,
                   int64_t M, int64_t d, float eps, int rms, int64_t group_rows, int64_t group_stride,
                   int64_t row_offset, int stats_physical, void* stream);
/* dX only (gamma/beta are frozen). dres_out[m] = (dres_in ? dres_in[m] : 0) + LN'(dy[m]); optionally also
 * a bf16 copy of dres_out (A operand of the next dX GEMM). dres_in may alias dres_out. dy rows are compact (logical);
 * x / dres rows are physical (gathered); stats rows are physical when stats_physical != 0 (statistics saved by a
 * full-size forward), logical otherwise (statistics saved by a gathered forward). */
int mtl_norm_bwd(const void* dy, int64_t ld_dy, const float* x, const float* gamma, const float* stats,
                 const float* dres_in, float* dres_out, void* dres_out_bf16, int64_t M, int64_t d, int rms,
                 int64_t group_rows, int64_t group_stride, int64_t row_offset, int stats_physical,
                 float bf16_drop_p, uint32_t bf16_drop_seed, void* stream);
/* stream_dtype = dtype of x, dres_in and dres_out (MTL_F32: above; MTL_BF16: the bf16 residual stream, whose gradient stream is bf16 too —
 * dres_out_bf16 may then be NULL when no dropout mask applies: dres_out itself is the next GEMM's operand) */
int mtl_norm_bwd_t(const void* dy, int64_t ld_dy, const void* x, int stream_dtype, const float* gamma, const float* stats,
                   const void* dres_in, void* dres_out, void* dres_out_bf16, int64_t M, int64_t d, int rms,
                   int64_t group_rows, int64_t group_stride, int64_t row_offset, int stats_physical,
                   float bf16_drop_p, uint32_t bf16_drop_seed, void* stream);
/* (bf16_drop_p > 0: the bf16 copy is dropout'(dres_out) with the keep mask of (seed, physical row, column) — the gradient
 *  that flows into the residual branch whose forward output was dropped with that mask; dres_out itself stays unmasked.)
 * Plain dropout of an f32 [M, d] matrix with the same counter hash (GPT-2 embd_pdrop on inputs_embeds + wpe,
 * HF:models/gpt2/modeling_gpt2.py:579; its own backward: apply it to the gradient). x may alias y. */
int mtl_dropout_f32(const float* x, float* y, int64_t M, int64_t d, float p, uint32_t seed, void* stream);

/* ------------------------------------------------------------------ Llama elementwise
 * RoPE, half-split rotate_half form (HF:models/llama/modeling_llama.py:130-160), in place on the q and k
 * heads of a fused qkv buffer bf16 [M, ld] laid out [q heads | k heads | v heads]; cos/sin f32 [T, D].
 * inverse != 0 applies the transpose rotation (backward). Row m has position m % T. */
int mtl_rope_inplace(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, int64_t M, int64_t T,
                     int64_t n_rot_heads, int64_t D, int inverse, void* stream);
/* SwiGLU (HF:models/llama/modeling_llama.py:174-176): gu bf16 [M, 2F] = [gate | up] -> h bf16 [M, F] = silu(gate)*up.
 * bwd: dgu[M, 2F] from dh[M, F] and the saved gu. */
int mtl_swiglu_fwd(const void* gu, void* h, int64_t M, int64_t F, void* stream);
int mtl_swiglu_bwd(const void* gu, const void* dh, void* dgu, int64_t M, int64_t F, void* stream);
/* row-gathered variants used by the pruned backbone backward: logical row m of the M processed rows is physical row
 * (m / group_rows)*group_stride + row_offset + m % group_rows of qkv / gu (group_rows == 0: identity); dh, dgu compact. */
int mtl_rope_inplace_rows(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, int64_t M, int64_t T,
                          int64_t n_rot_heads, int64_t D, int inverse, int64_t group_rows, int64_t group_stride,
                          int64_t row_offset, void* stream);
int mtl_swiglu_bwd_rows(const void* gu, const void* dh, void* dgu, int64_t M, int64_t F, int64_t group_rows,
                        int64_t group_stride, int64_t row_offset, int interleaved, void* stream);
/* (interleaved != 0: gu and dgu hold (gate_j, up_j) in columns (2j, 2j+1), the layout MTL_EPI_SWIGLU writes, instead of
 *  [gate | up] halves.) */

/* ------------------------------------------------------------------ LLM input assembly (a6 tail, K10/K11)
 * h0[b, t, :] = (t < n_tok ? embed[ids[b, t]] : x_tok[b, t - n_tok, :]) + (wpe ? wpe[t] : 0)   -> f32 [B, T, d]
 * Replaces the embedding gather, left padding (ids are already left-padded with pad_token_id host-side,
 * identical to padding with the pad embedding, R:models/medtsllm.py:304-311) and torch.cat (:349), plus
 * GPT-2's position add (HF:models/gpt2/modeling_gpt2.py:576-577). ids int32 [ids_B, n_tok] with ids_B == B or 1
 * (1 = the same constant prompt for every sample). x_tok bf16 [B, P, d]. embed f32 [V, d].
 * drop_p > 0: GPT-2's embd_pdrop (HF:models/gpt2/modeling_gpt2.py:579) applied to the assembled rows in the same pass, mask of
 * (drop_seed, row b * T + t, column) — the one mtl_dropout_f32 applies to h0 viewed as [B * T, d].
 * mtl_assemble_bwd: dx_tok bf16 [B, P, d] = mask * dh0[:, n_tok:, :] (the prompt rows have no trainable ancestor). */
int mtl_assemble_llm_input(const int32_t* ids, int64_t ids_B, const float* embed, const void* x_tok, const float* wpe,
                           float* h0, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p, uint32_t drop_seed, void* stream);
int mtl_assemble_bwd(const float* dh0, void* dx_tok, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p, uint32_t drop_seed,
                     void* stream);
/* the same for a residual stream of dtype h0_dtype / dh0_dtype (MTL_F32: above; MTL_BF16: the reference's setup.dtype = "bf16", where the embedding
 * tables, their sum and the dropped result are bf16 tensors: each is rounded) */
int mtl_assemble_llm_input_t(const int32_t* ids, int64_t ids_B, const float* embed, const void* x_tok, const float* wpe,
                             void* h0, int h0_dtype, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p, uint32_t drop_seed, void* stream);
int mtl_assemble_bwd_t(const void* dh0, int dh0_dtype, void* dx_tok, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p,
                       uint32_t drop_seed, void* stream);

/* ------------------------------------------------------------------ frozen backbone stack (a7)
 * Whole GPT-2 / Llama decoder stack forward and activation-gradient-only backward as ONE host call each
 * (the kernels above chained on `stream`, no host work in between). Replaces
 * self.llm(inputs_embeds=enc).last_hidden_state (R:models/medtsllm.py:350) = HF GPT2Model.forward
 * (HF:models/gpt2/modeling_gpt2.py:514-628) / LlamaModel.forward (HF:models/llama/modeling_llama.py:367-417),
 * eager attention semantics (R:models/medtsllm.py:159-160 always selects "eager"). GPT-2's train-mode dropouts
 * (attn_pdrop on the attention probabilities, resid_pdrop on both residual branches) are applied when a
 * mtl_backbone_dropout is passed; embd_pdrop is the caller's (mtl_dropout_f32 on h0).
 * Weights: bf16 [out, in] row-major plus the transposed copy [in, out] (frozen, prepared once);
 * per-layer pointer arrays live in HOST memory. */
enum { MTL_ARCH_GPT2 = 0, MTL_ARCH_LLAMA = 1 };
typedef struct {
    int arch, n_layers;
    int64_t d, n_heads, n_kv_heads, head_dim, ffn;
    float eps;
    const void* const* w_qkv;   const void* const* w_qkv_t;  const float* const* b_qkv;   /* [(Hq+2Hkv)*hd, d]        */
    const void* const* w_o;     const void* const* w_o_t;    const float* const* b_o;     /* [d, Hq*hd]               */
    const void* const* w_fc;    const void* const* w_fc_t;   const float* const* b_fc;    /* gpt2 [ffn,d]; llama [2ffn,d], rows interleaved: 2j = gate_j, 2j+1 = up_j */
    const void* const* w_proj;  const void* const* w_proj_t; const float* const* b_proj;  /* [d, ffn]                 */
    const float* const* ln1_w;  const float* const* ln1_b;
    const float* const* ln2_w;  const float* const* ln2_b;
    const float* lnf_w; const float* lnf_b;
    const float* rope_cos; const float* rope_sin;                                         /* llama: f32 [T, hd]       */
    /* dtype of the residual stream: MTL_F32 — the reference's setup.dtype = "mixed" (autocast keeps the stream fp32) and "fp32" — or MTL_BF16 — its
     * setup.dtype = "bf16" (R:tasks/base.py:261-262,205-208: model and inputs cast to bf16, no autocast): h0, the saved residual states, the
     * gradient stream dh0 are bf16, every residual add is rounded to bf16, half the stream's HBM bytes. The norm / bias parameters stay f32 arrays
     * (the caller rounds them to bf16 values for this mode, as the reference's .to(bfloat16) does). */
    int stream_dtype;
} mtl_backbone_weights;
typedef struct { float attn_p, resid_p; uint32_t seed; } mtl_backbone_dropout;   /* per-layer seeds are derived from `seed` */
/* bytes of the `saved` buffer (activations kept for the backward) and of the scratch `work` buffer */
size_t mtl_backbone_saved_bytes(const mtl_backbone_weights* w, int64_t B, int64_t T);
/* byte offset inside `saved` of the fp32 residual stream [B*T, d] AFTER decoder layer `layer` (1 .. n_layers; the HF hidden_states[layer] before the final
 * norm, R:models/medtsllm.py:350) — parity tests read intermediate depths there; (size_t)-1 on bad arguments. */
size_t mtl_backbone_saved_hidden_offset(const mtl_backbone_weights* w, int64_t B, int64_t T, int layer);
size_t mtl_backbone_work_bytes(const mtl_backbone_weights* w, int64_t B, int64_t T);
/* h0 f32 [B, T, d] (input embeddings, wpe already added for GPT-2). out bf16 [B, n_last, d]: final norm applied
 * to the last n_last tokens of every sample (only those are consumed downstream, R:models/medtsllm.py:353). */
/* n_save (0 <= n_save <= T): the MLP pre-activations that only the backward reads are stored for the last n_save tokens of every
 * sample: pass the n_grad the matching mtl_backbone_bwd will use (T when unknown), 0 for inference. */
/* Prompt-row forward cache (prefix_kv != NULL, 0 < n_prefix <= T - n_last; SURVEY.md 7 "legal shortcut i"): the first n_prefix tokens of
 * EVERY sample are the same constant text prompt (R:models/medtsllm.py:328-339 with a dataset / task prompt only) and the stack is
 * deterministic (no dropout struct): attention is causal, so every hidden state of those rows is the same in every sample and every
 * step. The forward then runs norms / GEMMs / MLP / attention queries on the last T - n_prefix tokens of every sample only and takes the
 * per-layer keys and values of the prompt rows from `prefix_kv` (mtl_backbone_prefix_build). h0's first n_prefix rows are not read.
 * Results for the computed rows equal the full forward's up to the summation order of differently tiled GEMMs; the matching
 * mtl_backbone_bwd needs n_grad <= T - n_prefix. Reported as executed work only: no roofline denominator takes the discount. */
int mtl_backbone_fwd(const mtl_backbone_weights* w, const void* h0 /* [B, T, d] of w->stream_dtype */, void* out, void* saved, void* work,
                     int64_t B, int64_t T, int64_t n_last, int64_t n_save, const mtl_backbone_dropout* drop /* NULL: off */,
                     const void* prefix_kv, int64_t n_prefix, void* stream);
/* bytes of the prompt-row cache: bf16 [n_layers, n_prefix, 2 * n_kv_heads * head_dim] (keys after RoPE | values) */
size_t mtl_backbone_prefix_bytes(const mtl_backbone_weights* w, int64_t n_prefix);
/* fills `prefix_kv` from the prompt rows h0_prefix f32 [1, n_prefix, d] (wpe already added for GPT-2): one forward of that single
 * sequence; `saved` / `work` sized by mtl_backbone_saved_bytes / _work_bytes(w, 1, n_prefix). w->rope_cos / rope_sin need >= n_prefix rows. */
int mtl_backbone_prefix_build(const mtl_backbone_weights* w, const void* h0_prefix /* w->stream_dtype */, void* prefix_kv, void* saved, void* work,
                              int64_t n_prefix, void* stream);
/* dout bf16 [B, n_last, d] -> dh0 [B, T, d] (f32, or bf16 on the bf16 stream). `saved` from the matching forward.
 * n_grad (n_last <= n_grad <= the forward's n_save): only the LAST n_grad tokens of every sample receive a gradient; rows before that are
 * left zero. The leading tokens are the text prompt: causal attention never lets them see a patch token, so they are
 * independent of every trainable parameter and their gradient is never consumed (SURVEY.md §7 "legal shortcut ii").
 * All backward GEMMs / norms / attention then run on B*n_grad rows. n_grad = T computes the full dh0. */
int mtl_backbone_bwd(const mtl_backbone_weights* w, const void* h0, const void* dout, void* dh0 /* both of w->stream_dtype */, void* saved,
                     void* work, int64_t B, int64_t T, int64_t n_last, int64_t n_grad,
                     const mtl_backbone_dropout* drop /* the forward's */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDTSLLM_HIP_H */
