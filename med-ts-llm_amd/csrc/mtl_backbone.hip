// mtl_backbone.hip — the frozen GPT-2 / Llama decoder stack as ONE host call per direction.
//
// Pure launch sequencing over the kernels in mtl_gemm/mtl_attention/mtl_norm/mtl_elementwise on the caller's
// stream: no host synchronisation, no allocation (the caller owns `saved` and `work`). The backward is
// activation-gradient only — the backbone is frozen (R:models/medtsllm.py:231-233), so no dW is ever formed and
// LN/attention-projection INPUTS need not be saved; what is saved per layer is exactly
//   residual stream before each norm (fp32), norm statistics, fused qkv (bf16), attention output (bf16), LSE,
//   MLP pre-activation (GPT-2) / gate|up (Llama).
#include "mtl_common.h"

#include <cstdlib>

namespace {

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Dims {
    int64_t B, T, M, d, Hq, Hkv, hd, ffn, Nqkv, No, Nfc;
    int L;
    bool llama;
    int sdt;           // dtype of the residual stream: MTL_F32 (the reference's dtype "mixed" / "fp32") or MTL_BF16 (its dtype "bf16")
    size_t es;         // bytes per stream element
};

Dims dims_of(const mtl_backbone_weights* w, int64_t B, int64_t T) {
    Dims D;
    D.B = B; D.T = T; D.M = B * T; D.d = w->d; D.Hq = w->n_heads; D.Hkv = w->n_kv_heads; D.hd = w->head_dim; D.ffn = w->ffn;
    D.Nqkv = (D.Hq + 2 * D.Hkv) * D.hd; D.No = D.Hq * D.hd; D.L = w->n_layers; D.llama = (w->arch == MTL_ARCH_LLAMA);
    D.Nfc = D.llama ? 2 * D.ffn : D.ffn;
    D.sdt = w->stream_dtype == MTL_BF16 ? MTL_BF16 : MTL_F32;
    D.es = D.sdt == MTL_BF16 ? 2 : 4;
    return D;
}

// ---- `saved` layout
struct SavedLayout {
    size_t h, stats, qkv, attn, lse, fc, stats_f, total;  // base offsets; per-layer strides below
    size_t h_stride, stats_stride, qkv_stride, attn_stride, lse_stride, fc_stride;
};
SavedLayout saved_layout(const Dims& D, int64_t n_last) {
    SavedLayout s;
    s.h_stride = align_up((size_t)D.M * D.d * D.es);
    s.stats_stride = align_up((size_t)D.M * 2 * 4);
    s.qkv_stride = align_up((size_t)D.M * D.Nqkv * 2);
    s.attn_stride = align_up((size_t)D.M * D.No * 2);
    s.lse_stride = align_up((size_t)D.B * D.Hq * D.T * 4);
    s.fc_stride = align_up((size_t)D.M * D.Nfc * 2);
    size_t off = 0;
    s.h = off; off += s.h_stride * 2 * D.L;          // H[1 .. 2L]
    s.stats = off; off += s.stats_stride * 2 * D.L;  // (ln1, ln2) per layer
    s.qkv = off; off += s.qkv_stride * D.L;
    s.attn = off; off += s.attn_stride * D.L;
    s.lse = off; off += s.lse_stride * D.L;
    s.fc = off; off += s.fc_stride * D.L;
    s.stats_f = off; off += align_up((size_t)D.B * (n_last > 0 ? n_last : D.T) * 2 * 4);
    s.total = off;
    return s;
}

// ---- `work` layout (scratch shared by fwd and bwd)
struct WorkLayout {
    size_t xln, act, dres_b, dact, dx, dqkv, dO, delta, total;
};
WorkLayout work_layout(const Dims& D) {
    WorkLayout w;
    size_t off = 0;
    w.xln = off; off += align_up((size_t)D.M * D.d * 2);
    w.act = off; off += align_up((size_t)D.M * D.ffn * 2);
    w.dres_b = off; off += align_up((size_t)D.M * D.d * 2);
    w.dact = off; off += align_up((size_t)D.M * D.Nfc * 2);
    w.dx = off; off += align_up((size_t)D.M * D.d * 2);
    w.dqkv = off; off += align_up((size_t)D.M * D.Nqkv * 2);
    w.dO = off; off += align_up((size_t)D.M * D.No * 2);
    w.delta = off; off += align_up((size_t)D.B * D.Hq * D.T * 4);
    w.total = off;
    return w;
}

struct RowMap { int64_t rows, stride, offset; };   // logical row m -> (m / rows) * stride + offset + m % rows; rows == 0: identity
const RowMap kIdentity = {0, 0, 0};

int gemm(const void* A, int64_t lda, const void* Bm, int64_t ldb, void* C, int64_t ldc, int cdt, int64_t M, int64_t N, int64_t K,
         const float* bias, int epi, const void* aux_in, int64_t ld_aux_in, void* aux_out, int64_t ld_aux_out, void* stream,
         RowMap am = kIdentity, RowMap cm = kIdentity, float drop_p = 0.f, uint32_t drop_seed = 0u, int64_t bwd_group = 0,
         int64_t bwd_first = 0) {
    mtl_gemm_args g = {};
    g.drop_p = drop_p; g.drop_seed = drop_seed;
    g.bwd_group_rows = bwd_group; g.bwd_first_row = bwd_first;
    g.a_group_rows = am.rows; g.a_group_stride = am.stride; g.a_row_offset = am.offset;
    g.c_group_rows = cm.rows; g.c_group_stride = cm.stride; g.c_row_offset = cm.offset;
    g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.c_dtype = cdt;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.epilogue = epi;
    g.aux_in = aux_in; g.ld_aux_in = ld_aux_in; g.aux_out = aux_out; g.ld_aux_out = ld_aux_out;
    g.alpha = 1.0f; g.split_k = 1;
    return mtl_gemm_nt(&g, stream);
}

#define MTL_TRY(expr)                 \
    do {                              \
        const int rc__ = (expr);      \
        if (rc__ != MTL_OK) return rc__; \
    } while (0)

void attn_args(const Dims& D, char* qkv, char* attn, float* lse, mtl_attn_fwd_args* f) {
    bf16_t* q = reinterpret_cast<bf16_t*>(qkv);
    f->q = q; f->q_bs = D.T * D.Nqkv; f->q_ts = D.Nqkv; f->q_hs = D.hd;
    f->k = q + D.Hq * D.hd; f->k_bs = D.T * D.Nqkv; f->k_ts = D.Nqkv; f->k_hs = D.hd;
    f->v = q + (D.Hq + D.Hkv) * D.hd; f->v_bs = D.T * D.Nqkv; f->v_ts = D.Nqkv; f->v_hs = D.hd;
    f->o = attn; f->o_bs = D.T * D.No; f->o_ts = D.No; f->o_hs = D.hd;
    f->lse = lse;
    f->B = D.B; f->Hq = D.Hq; f->Hkv = D.Hkv; f->Tq = D.T; f->Tk = D.T; f->D = D.hd;
    f->scale = 1.0f / sqrtf((float)D.hd);
    f->causal = 1;
    f->causal_off = 0;
    f->stat_stride = 0;
}

int check_weights(const mtl_backbone_weights* w) {
    if (!w) return MTL_ERR_ARG;
    if (w->arch != MTL_ARCH_GPT2 && w->arch != MTL_ARCH_LLAMA) return MTL_ERR_UNSUPPORTED;
    if (w->n_layers <= 0 || w->d <= 0 || w->n_heads <= 0 || w->n_kv_heads <= 0 || w->head_dim <= 0 || w->ffn <= 0) return MTL_ERR_ARG;
    if (!w->w_qkv || !w->w_qkv_t || !w->w_o || !w->w_o_t || !w->w_fc || !w->w_fc_t || !w->w_proj || !w->w_proj_t) return MTL_ERR_ARG;
    if (!w->ln1_w || !w->ln2_w || !w->lnf_w) return MTL_ERR_ARG;
    if (w->arch == MTL_ARCH_GPT2 && (!w->ln1_b || !w->ln2_b || !w->lnf_b)) return MTL_ERR_ARG;
    if (w->arch == MTL_ARCH_LLAMA && (!w->rope_cos || !w->rope_sin)) return MTL_ERR_ARG;
    if (w->stream_dtype != MTL_F32 && w->stream_dtype != MTL_BF16) return MTL_ERR_ARG;
    return MTL_OK;
}

}  // namespace

extern "C" size_t mtl_backbone_saved_bytes(const mtl_backbone_weights* w, int64_t B, int64_t T) {
    if (check_weights(w) != MTL_OK || B <= 0 || T <= 0) return 0;
    return saved_layout(dims_of(w, B, T), T).total;
}

extern "C" size_t mtl_backbone_saved_hidden_offset(const mtl_backbone_weights* w, int64_t B, int64_t T, int layer) {
    if (check_weights(w) != MTL_OK || B <= 0 || T <= 0 || layer < 1 || layer > w->n_layers) return (size_t)-1;
    const SavedLayout S = saved_layout(dims_of(w, B, T), T);
    return S.h + S.h_stride * (size_t)(2 * layer - 1);          // H[2 * layer] lives in slot 2 * layer - 1 (H[0] is the caller's h0); dtype = w->stream_dtype
}

extern "C" size_t mtl_backbone_work_bytes(const mtl_backbone_weights* w, int64_t B, int64_t T) {
    if (check_weights(w) != MTL_OK || B <= 0 || T <= 0) return 0;
    return work_layout(dims_of(w, B, T)).total;
}

// prompt-row cache: K (after RoPE) | V of the first n_prefix tokens, per layer, the same for every sample
__global__ void prefix_kv_fill_kernel(const bf16_t* __restrict__ cache, bf16_t* __restrict__ qkv, int64_t n_prefix, int64_t T, int64_t ldq,
                                      int64_t kv_off, int64_t kv_cols, int64_t B) {
    // one 16-byte chunk of one cache row per thread, written to that row of every sample (the cache row stays in L2 / registers)
    const int64_t cpr = kv_cols / 8, total = n_prefix * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cpr, c = (idx % cpr) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(cache + r * kv_cols + c);
        for (int64_t b = blockIdx.y; b < B; b += gridDim.y) *reinterpret_cast<u32x4*>(qkv + (b * T + r) * ldq + kv_off + c) = v;
    }
}

namespace {

// the forward over the last n_fwd = T - n_prefix tokens of every sample (n_prefix == 0: all of them, identity row maps)
int backbone_fwd_impl(const mtl_backbone_weights* w, const void* h0, void* out, void* saved, void* work, int64_t B, int64_t T, int64_t n_last,
                      int64_t n_save, const mtl_backbone_dropout* drop, const void* prefix_kv, int64_t n_prefix, void* stream) {
    MTL_TRY(check_weights(w));
    if (n_save < 0 || n_save > T) return MTL_ERR_ARG;
    // GPT-2 train-mode dropouts (HF:models/gpt2/modeling_gpt2.py:65,243,397): attention probabilities + both residual branches
    const float attn_p = drop ? drop->attn_p : 0.f, resid_p = drop ? drop->resid_p : 0.f;
    const uint32_t dseed = drop ? drop->seed : 0u;
    if (attn_p < 0.f || attn_p >= 1.f || resid_p < 0.f || resid_p >= 1.f) return MTL_ERR_ARG;
    if (!h0 || !out || !saved || !work || B <= 0 || T <= 0 || n_last <= 0 || n_last > T) return MTL_ERR_ARG;
    if (n_prefix < 0 || (n_prefix > 0 && (!prefix_kv || n_prefix > T - n_last || attn_p > 0.f || resid_p > 0.f))) return MTL_ERR_ARG;
    const Dims D = dims_of(w, B, T);
    const SavedLayout S = saved_layout(D, T);
    const WorkLayout W = work_layout(D);
    char* sv = reinterpret_cast<char*>(saved);
    char* wk = reinterpret_cast<char*>(work);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rms = D.llama ? 1 : 0;
    auto H = [&](int idx) -> char* {  // residual stream H[0] = h0, H[1..2L] in saved (fp32 or bf16: D.sdt)
        return idx == 0 ? reinterpret_cast<char*>(const_cast<void*>(h0)) : sv + S.h + S.h_stride * (size_t)(idx - 1);
    };
    // computed rows: the last n_fwd tokens of every sample. Buffers addressed by (b, t) keep their full-size physical layout (the
    // backward reads them there) and are touched through the row map; the norm output xln is compact.
    const int64_t r0 = n_prefix, n_fwd = D.T - n_prefix, Mf = D.B * n_fwd;
    const RowMap rm = (n_prefix == 0) ? kIdentity : RowMap{n_fwd, D.T, r0};
    if (n_save > n_fwd) n_save = n_fwd;
    const int64_t save_group = 0, save_first = 0;     // (per layer below: the last layer may compute fewer rows)
    const int64_t kv_cols = 2 * D.Hkv * D.hd;
    // LAST layer, after its keys / values exist: nothing downstream reads the hidden states of the leading tokens — the final norm takes the
    // last n_last tokens, the backward the last n_save — so its attention queries, output projection and whole MLP run on the last
    // max(n_last, n_save) tokens of every sample only (exact: those rows' results do not change; the others are dead stores).
    const int64_t n_tail = (n_last > n_save ? n_last : n_save) < n_fwd ? (n_last > n_save ? n_last : n_save) : n_fwd;
    for (int i = 0; i < D.L; ++i) {
        float* st1 = reinterpret_cast<float*>(sv + S.stats + S.stats_stride * (size_t)(2 * i));
        float* st2 = reinterpret_cast<float*>(sv + S.stats + S.stats_stride * (size_t)(2 * i + 1));
        char* qkv = sv + S.qkv + S.qkv_stride * (size_t)i;
        char* attn = sv + S.attn + S.attn_stride * (size_t)i;
        float* lse = reinterpret_cast<float*>(sv + S.lse + S.lse_stride * (size_t)i);
        char* fc = sv + S.fc + S.fc_stride * (size_t)i;
        const float* bq = w->b_qkv ? w->b_qkv[i] : nullptr;
        const float* bo = w->b_o ? w->b_o[i] : nullptr;
        const float* bf = w->b_fc ? w->b_fc[i] : nullptr;
        const float* bp = w->b_proj ? w->b_proj[i] : nullptr;
        // --- attention block
        MTL_TRY(mtl_norm_fwd_t(H(2 * i), D.sdt, w->ln1_w[i], w->ln1_b ? w->ln1_b[i] : nullptr, wk + W.xln, D.d, st1, Mf, D.d, w->eps, rms, rm.rows, rm.stride,
                               rm.offset, 1, stream));
        MTL_TRY(gemm(wk + W.xln, D.d, w->w_qkv[i], D.d, qkv, D.Nqkv, MTL_BF16, Mf, D.Nqkv, D.d, bq, MTL_EPI_STORE, nullptr, 0, nullptr, 0, stream,
                     kIdentity, rm));
        if (D.llama) MTL_TRY(mtl_rope_inplace_rows(qkv, D.Nqkv, w->rope_cos, w->rope_sin, Mf, D.T, D.Hq + D.Hkv, D.hd, 0, rm.rows, rm.stride, rm.offset, stream));
        if (n_prefix > 0) {      // keys / values of the prompt rows: from the cache into every sample's rows [0, n_prefix)
            const bf16_t* cache = reinterpret_cast<const bf16_t*>(prefix_kv) + (size_t)i * n_prefix * kv_cols;
            const int64_t chunks = n_prefix * (kv_cols / 8);
            const unsigned gx = (unsigned)((chunks + 255) / 256), gy = (unsigned)(D.B < 8 ? D.B : 8);
            hipLaunchKernelGGL(prefix_kv_fill_kernel, dim3(gx, gy), dim3(256), 0, st, cache, reinterpret_cast<bf16_t*>(qkv), n_prefix, D.T, D.Nqkv,
                               D.Hq * D.hd, kv_cols, D.B);
            MTL_CHECK_LAUNCH();
        }
        // rows of the rest of this layer: the computed rows, or (last layer) only the tail somebody still reads
        const bool tail_only = (i == D.L - 1) && n_tail < n_fwd;
        const int64_t n_rows = tail_only ? n_tail : n_fwd, q0 = D.T - n_rows, Mr = D.B * n_rows;
        const RowMap rr = (n_rows == D.T) ? kIdentity : RowMap{n_rows, D.T, q0};
        const int64_t sv_group = n_save < n_rows ? n_rows : 0, sv_first = n_rows - n_save;
        (void)save_group; (void)save_first;
        mtl_attn_fwd_args fa = {};
        attn_args(D, qkv, attn, lse, &fa);
        fa.dropout_p = attn_p; fa.dropout_seed = drop_site_seed(dseed, i, 0);
        if (n_rows < D.T) {      // queries = the rows computed from here on, keys = all rows
            fa.q = reinterpret_cast<const bf16_t*>(fa.q) + q0 * fa.q_ts;
            fa.o = reinterpret_cast<bf16_t*>(fa.o) + q0 * fa.o_ts;
            fa.lse = lse + q0;
            fa.Tq = n_rows; fa.causal_off = q0; fa.stat_stride = D.T;
        }
        MTL_TRY(mtl_attention_fwd(&fa, stream));
        MTL_TRY(gemm(attn, D.No, w->w_o[i], D.No, H(2 * i + 1), D.d, D.sdt, Mr, D.d, D.No, bo, MTL_EPI_RESID, H(2 * i), D.d, nullptr, 0, stream,
                     rr, rr, resid_p, drop_site_seed(dseed, i, 1)));
        // --- MLP block
        MTL_TRY(mtl_norm_fwd_t(H(2 * i + 1), D.sdt, w->ln2_w[i], w->ln2_b ? w->ln2_b[i] : nullptr, wk + W.xln, D.d, st2, Mr, D.d, w->eps, rms, rr.rows, rr.stride,
                               rr.offset, 1, stream));
        if (D.llama) {
            // gate|up GEMM with the SwiGLU fused into its epilogue (weights row-interleaved: columns 2j / 2j+1 = gate_j / up_j)
            // (the saved pre-activations are only read by the backward: stored for the last n_save tokens of every sample)
            MTL_TRY(gemm(wk + W.xln, D.d, w->w_fc[i], D.d, fc, D.Nfc, MTL_BF16, Mr, D.Nfc, D.d, bf, MTL_EPI_SWIGLU, nullptr, 0, wk + W.act, D.ffn, stream,
                         kIdentity, rr, 0.f, 0u, sv_group, sv_first));
        } else {
            MTL_TRY(gemm(wk + W.xln, D.d, w->w_fc[i], D.d, wk + W.act, D.ffn, MTL_BF16, Mr, D.ffn, D.d, bf, MTL_EPI_GELU, nullptr, 0, fc, D.ffn, stream,
                         kIdentity, rr, 0.f, 0u, sv_group, sv_first));
        }
        MTL_TRY(gemm(wk + W.act, D.ffn, w->w_proj[i], D.ffn, H(2 * i + 2), D.d, D.sdt, Mr, D.d, D.ffn, bp, MTL_EPI_RESID, H(2 * i + 1), D.d, nullptr, 0, stream,
                     rr, rr, resid_p, drop_site_seed(dseed, i, 2)));
    }
    float* stf = reinterpret_cast<float*>(sv + S.stats_f);
    return mtl_norm_fwd_t(H(2 * D.L), D.sdt, w->lnf_w, w->lnf_b, out, D.d, stf, D.B * n_last, D.d, w->eps, rms, n_last, D.T, D.T - n_last, 0, stream);
}

}  // namespace

extern "C" int mtl_backbone_fwd(const mtl_backbone_weights* w, const void* h0, void* out, void* saved, void* work, int64_t B,
                                int64_t T, int64_t n_last, int64_t n_save, const mtl_backbone_dropout* drop, const void* prefix_kv,
                                int64_t n_prefix, void* stream) {
    return backbone_fwd_impl(w, h0, out, saved, work, B, T, n_last, n_save, drop, prefix_kv, prefix_kv ? n_prefix : 0, stream);
}

extern "C" size_t mtl_backbone_prefix_bytes(const mtl_backbone_weights* w, int64_t n_prefix) {
    if (check_weights(w) != MTL_OK || n_prefix <= 0) return 0;
    return (size_t)w->n_layers * (size_t)n_prefix * (size_t)(2 * w->n_kv_heads * w->head_dim) * 2;
}

extern "C" int mtl_backbone_prefix_build(const mtl_backbone_weights* w, const void* h0_prefix, void* prefix_kv, void* saved, void* work,
                                         int64_t n_prefix, void* stream) {
    MTL_TRY(check_weights(w));
    if (!h0_prefix || !prefix_kv || !saved || !work || n_prefix <= 0) return MTL_ERR_ARG;
    const Dims D = dims_of(w, 1, n_prefix);
    const SavedLayout S = saved_layout(D, n_prefix);
    // one ordinary forward of the single prompt sequence; its final-norm output (1 row) lands in the work buffer's dx scratch
    const WorkLayout W = work_layout(D);
    MTL_TRY(backbone_fwd_impl(w, h0_prefix, reinterpret_cast<char*>(work) + W.dx, saved, work, 1, n_prefix, 1, 0, nullptr, nullptr, 0, stream));
    const size_t kv_bytes = (size_t)(2 * D.Hkv * D.hd) * 2;
    for (int i = 0; i < D.L; ++i) {
        const char* qkv = reinterpret_cast<const char*>(saved) + S.qkv + S.qkv_stride * (size_t)i + (size_t)D.Hq * D.hd * 2;
        char* dst = reinterpret_cast<char*>(prefix_kv) + (size_t)i * n_prefix * kv_bytes;
        if (hipMemcpy2DAsync(dst, kv_bytes, qkv, (size_t)D.Nqkv * 2, kv_bytes, (size_t)n_prefix, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
            return MTL_ERR_LAUNCH;
    }
    return MTL_OK;
}

extern "C" int mtl_backbone_bwd(const mtl_backbone_weights* w, const void* h0, const void* dout, void* dh0, void* saved, void* work,
                                int64_t B, int64_t T, int64_t n_last, int64_t n_grad, const mtl_backbone_dropout* drop, void* stream) {
    MTL_TRY(check_weights(w));
    const float attn_p = drop ? drop->attn_p : 0.f, resid_p = drop ? drop->resid_p : 0.f;
    const uint32_t dseed = drop ? drop->seed : 0u;
    if (attn_p < 0.f || attn_p >= 1.f || resid_p < 0.f || resid_p >= 1.f) return MTL_ERR_ARG;
    if (!h0 || !dout || !dh0 || !saved || !work || B <= 0 || T <= 0 || n_last <= 0 || n_last > T) return MTL_ERR_ARG;
    if (n_grad < n_last || n_grad > T) return MTL_ERR_ARG;
    const Dims D = dims_of(w, B, T);
    const SavedLayout S = saved_layout(D, T);
    const WorkLayout W = work_layout(D);
    char* sv = reinterpret_cast<char*>(saved);
    char* wk = reinterpret_cast<char*>(work);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rms = D.llama ? 1 : 0;
    auto H = [&](int idx) -> char* {
        return idx == 0 ? reinterpret_cast<char*>(const_cast<void*>(h0)) : sv + S.h + S.h_stride * (size_t)(idx - 1);
    };
    // bf16 stream without resid_pdrop: the gradient stream dh0 IS the bf16 operand of the next branch's first GEMM (same [B * T, d] layout): no
    // second copy is written. With the dropout the branch sees the masked copy (dres_b), as on the fp32 stream.
    const bool share = D.sdt == MTL_BF16 && resid_p == 0.f;
    char* const dA = share ? reinterpret_cast<char*>(dh0) : wk + W.dres_b;          // A operand of the branch-input GEMMs
    void* const dB = share ? nullptr : (void*)(wk + W.dres_b);                      // masked bf16 copy written by the norm backward
    // rows that receive a gradient: the last n_grad tokens of every sample (see header). Buffers addressed by (b, t)
    // keep their full-size physical layout and are touched through the row map; dx / dact(llama) scratch is compact.
    const int64_t r0 = D.T - n_grad;
    const int64_t Mg = D.B * n_grad;
    const RowMap rm = (n_grad == D.T) ? kIdentity : RowMap{n_grad, D.T, r0};
    if (n_last < T) {   // rows without incoming gradient start at zero (dh0: all of them, it is the returned gradient; the bf16 scratch copy:
                        // only rows [r0, T - n_last) — the kernels never read it below r0)
        if (hipMemsetAsync(dh0, 0, (size_t)D.M * D.d * D.es, st) != hipSuccess) return MTL_ERR_LAUNCH;
        if (!share && n_last < n_grad &&
            hipMemset2DAsync(wk + W.dres_b + (size_t)r0 * D.d * 2, (size_t)D.T * D.d * 2, 0, (size_t)(n_grad - n_last) * D.d * 2, (size_t)D.B, st) != hipSuccess)
            return MTL_ERR_LAUNCH;
    }
    const float* stf = reinterpret_cast<const float*>(sv + S.stats_f);
    // the bf16 copy of the residual gradient is the A operand of the next branch's first GEMM: it carries that branch's
    // resid_pdrop mask (the fp32 stream dh0 is the identity path and stays unmasked)
    MTL_TRY(mtl_norm_bwd_t(dout, D.d, H(2 * D.L), D.sdt, w->lnf_w, stf, nullptr, dh0, dB, D.B * n_last, D.d, rms, n_last, D.T, D.T - n_last, 0,
                           resid_p, drop_site_seed(dseed, D.L - 1, 2), stream));
    for (int i = D.L - 1; i >= 0; --i) {
        const float* st1 = reinterpret_cast<const float*>(sv + S.stats + S.stats_stride * (size_t)(2 * i));
        const float* st2 = reinterpret_cast<const float*>(sv + S.stats + S.stats_stride * (size_t)(2 * i + 1));
        char* qkv = sv + S.qkv + S.qkv_stride * (size_t)i;
        char* attn = sv + S.attn + S.attn_stride * (size_t)i;
        float* lse = reinterpret_cast<float*>(sv + S.lse + S.lse_stride * (size_t)i);
        char* fc = sv + S.fc + S.fc_stride * (size_t)i;
        // --- MLP block backward: h_out = h_mid + proj(act(fc(norm2(h_mid))))
        if (D.llama) {
            // d(act) GEMM with the SwiGLU backward fused into its epilogue: reads the saved (gate, up) pairs, writes d(gate|up)
            // interleaved, at physical rows like the GPT-2 path (the next GEMM gathers them)
            MTL_TRY(gemm(dA, D.d, w->w_proj_t[i], D.d, wk + W.dact, D.Nfc, MTL_BF16, Mg, D.ffn, D.d, nullptr, MTL_EPI_DSWIGLU, fc, D.Nfc, nullptr, 0,
                         stream, rm, rm));
        } else {
            // C (and the saved pre-activation read by the dgelu epilogue) keep physical rows; the next GEMM gathers them
            MTL_TRY(gemm(dA, D.d, w->w_proj_t[i], D.d, wk + W.dact, D.ffn, MTL_BF16, Mg, D.ffn, D.d, nullptr, MTL_EPI_DGELU, fc, D.ffn, nullptr, 0, stream, rm, rm));
        }
        MTL_TRY(gemm(wk + W.dact, D.Nfc, w->w_fc_t[i], D.Nfc, wk + W.dx, D.d, MTL_BF16, Mg, D.d, D.Nfc, nullptr, MTL_EPI_STORE, nullptr, 0, nullptr, 0, stream,
                     rm));
        MTL_TRY(mtl_norm_bwd_t(wk + W.dx, D.d, H(2 * i + 1), D.sdt, w->ln2_w[i], st2, dh0, dh0, dB, Mg, D.d, rms, rm.rows, rm.stride, rm.offset, 1,
                               resid_p, drop_site_seed(dseed, i, 1), stream));
        // --- attention block backward: h_mid = h_in + o_proj(attn(qkv(norm1(h_in))))
        MTL_TRY(gemm(dA, D.d, w->w_o_t[i], D.d, wk + W.dO, D.No, MTL_BF16, Mg, D.No, D.d, nullptr, MTL_EPI_STORE, nullptr, 0, nullptr, 0, stream, rm, rm));
        mtl_attn_bwd_args ba = {};
        attn_args(D, qkv, attn, lse, &ba.f);
        ba.f.dropout_p = attn_p; ba.f.dropout_seed = drop_site_seed(dseed, i, 0);
        bf16_t* dq = reinterpret_cast<bf16_t*>(wk + W.dqkv);
        // queries = the last n_grad rows (they see every key); dK/dV only for keys >= r0, fed by those queries only
        ba.f.q = reinterpret_cast<const bf16_t*>(ba.f.q) + r0 * ba.f.q_ts;
        ba.f.o = reinterpret_cast<bf16_t*>(ba.f.o) + r0 * ba.f.o_ts;
        ba.f.lse = lse + r0;
        ba.f.Tq = n_grad; ba.f.causal_off = r0; ba.f.stat_stride = D.T;
        ba.kv_row0 = r0;
        ba.dout = reinterpret_cast<bf16_t*>(wk + W.dO) + r0 * D.No; ba.do_bs = D.T * D.No; ba.do_ts = D.No; ba.do_hs = D.hd;
        ba.dq = dq + r0 * D.Nqkv; ba.dq_bs = D.T * D.Nqkv; ba.dq_ts = D.Nqkv; ba.dq_hs = D.hd;
        ba.dk = dq + D.Hq * D.hd; ba.dk_bs = D.T * D.Nqkv; ba.dk_ts = D.Nqkv; ba.dk_hs = D.hd;
        ba.dv = dq + (D.Hq + D.Hkv) * D.hd; ba.dv_bs = D.T * D.Nqkv; ba.dv_ts = D.Nqkv; ba.dv_hs = D.hd;
        ba.delta = reinterpret_cast<float*>(wk + W.delta) + r0;
        // Llama: the inverse rotary embedding of dq / dk rides in the attention kernels' store epilogues (diagnostic builds, MTL_ROPE_FUSE=0: a pass over dqkv)
        static const bool rope_fuse = mtl_env_int("MTL_ROPE_FUSE", 1) != 0;
        if (D.llama && rope_fuse) { ba.rope_cos = w->rope_cos; ba.rope_sin = w->rope_sin; }
        MTL_TRY(mtl_attention_bwd(&ba, stream));
        if (D.llama && !rope_fuse)
            MTL_TRY(mtl_rope_inplace_rows(wk + W.dqkv, D.Nqkv, w->rope_cos, w->rope_sin, Mg, D.T, D.Hq + D.Hkv, D.hd, 1, rm.rows, rm.stride, rm.offset, stream));
        MTL_TRY(gemm(wk + W.dqkv, D.Nqkv, w->w_qkv_t[i], D.Nqkv, wk + W.dx, D.d, MTL_BF16, Mg, D.d, D.Nqkv, nullptr, MTL_EPI_STORE, nullptr, 0, nullptr, 0, stream, rm));
        MTL_TRY(mtl_norm_bwd_t(wk + W.dx, D.d, H(2 * i), D.sdt, w->ln1_w[i], st1, dh0, dh0, dB, Mg, D.d, rms, rm.rows, rm.stride, rm.offset, 1,
                               i > 0 ? resid_p : 0.f, drop_site_seed(dseed, i - 1, 2), stream));
    }
    return MTL_OK;
}
