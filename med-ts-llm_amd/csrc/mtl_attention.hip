// mtl_attention.hip — flash-style attention forward / backward on MFMA (never materialises the score matrix).
//
// One kernel family serves (a) the backbone's causal self-attention (GPT-2 MHA, Llama MHA/GQA) and (b) the
// reprogramming cross-attention (no mask, K/V = the shared vocabulary prototypes, batch stride 0).
//
// Layout trick (wave64, v_mfma_f32_16x16x32_bf16): scores are computed TRANSPOSED, S^T[key][q] = K . Q^T, so a
// lane owns one query column (q = lane & 15) and 4 keys per 16-key tile. Softmax statistics are then per-lane
// scalars (+ two cross-lane-group shuffles), and the exponentiated P values of two 16-key tiles ARE the B operand
// of the P.V MFMA without any cross-lane movement: the contraction index of that MFMA is mapped to keys as
//     k-index (g, j)  <->  key  g*4 + j (j < 4)   |   16 + g*4 + (j - 4) (j >= 4),        g = lane >> 4
// and the V^T operand is gathered from the LDS V tile with the same mapping. The backward kernels use the same
// idea (dS / P in registers feed the dQ, dK, dV MFMAs directly).
//
// v1 structure: 4 waves x 16 query rows per workgroup, 64-key K/V chunks staged through padded LDS tiles
// (row stride D+8 bf16: conflict-free 16-B fragment reads), fp32 online softmax.
#include "mtl_common.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace {

constexpr int KC = 64;  // keys (or queries, in the dK/dV kernel) per LDS chunk
// Row padding of every LDS tile, in bf16 elements. 16 (row stride = 8 mod 64 dwords): conflict-free for BOTH read shapes of a tile —
// the 16-byte fragment rows (ds_read_b128 is serviced in the four non-contiguous 16-lane groups of GUIDE MI355X_MICROARCH "LDS": with a
// stride of 8 dwords mod 64 each group's 16 slots are distinct) and the transposing gathers (ds_read_b64_tr_b16, two 32-lane groups = 8
// rows x 32 B: 8 dwords apart). The earlier 8 (stride = 4 mod 64) left the gathers 2-way conflicted (SQ_LDS_BANK_CONFLICT / LDS_ACTIVE =
// 0.40-0.45 on every attention kernel) and rows 11 / 12 of a fragment read on one slot.
// hd 64 / 32 keep 8: their stride (36 / 20 dwords) is already conflict-free for both shapes, and 16 would push the resident hd-64 dK/dV kernel
// from two workgroups per CU to one (84 KB of LDS: 35.9 -> 46.6 us at T = 256, profiles/r04_attn_longT_experiments.txt).
#ifndef MTL_ATTN_PAD
#define MTL_ATTN_PAD 16
#endif
#ifndef MTL_ATTN_PAD64
#define MTL_ATTN_PAD64 8
#endif
__host__ __device__ constexpr int attn_pad(int D) { return D >= 128 ? MTL_ATTN_PAD : MTL_ATTN_PAD64; }
// The resident hd-64 FORWARD and the one-launch (merged) BACKWARD — the two attention kernels of the metric step — pad their rows by 16: the stride of
// 40 dwords is conflict-free for both read shapes under the lane groups of GUIDE MI355X_MICROARCH "LDS" (ds_read_b128: the 16 four-dword slots
// (10 r + g) mod 16 of a group's rows {0-3, 12-15} x g and {4-11} x (g + 1) are distinct; ds_read_b64_tr_b16: 8 rows x 8 dwords at 40 r mod 64 tile the 64
// banks), whereas stride 36 collides 7 of 16 slots per b128 group (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.43 - 0.45). Both still
// fit as before (forward: 2 x 80 KB per CU; merged backward: 141 KB, one workgroup per CU either way). Time: unchanged within noise (19.7 -> 19.6 - 19.9 us,
// 28.5 -> 28.1 - 28.3 us, profiles/r05_experiments_flat.txt) — the kernels are VALU-issue-bound — so the padding is picked from the conflict counter.
// The other hd-64 kernels keep 8: the two-launch resident dK/dV kernel would lose its second workgroup per CU (profiles/r04_attn_longT_experiments.txt).
#ifndef MTL_ATTN_PAD64_RES
#define MTL_ATTN_PAD64_RES 16
#endif
__host__ __device__ constexpr int attn_pad_res(int D) { return D >= 128 ? MTL_ATTN_PAD : MTL_ATTN_PAD64_RES; }
#ifndef MTL_CONSISTENT_DELTA
#define MTL_CONSISTENT_DELTA 0     // causal self-attention: 0 = delta = dO . O with the bf16-rounded forward output (measured: the consistent
                                   // form changes nothing there — tools/diag_bias.py: the stack's input gradient is unbiased and on par
                                   // with the reference's mixed mode — and costs two more MFMA groups); the reprogramming attention
                                   // (unmasked, near-uniform probabilities over shared keys) always uses the consistent form
#endif

// reductions across the four 16-lane rows of a wave (lanes with equal lane & 15) without touching the LDS crossbar:
// v_permlane16_swap / v_permlane32_swap exchange whole rows / halves in one VALU instruction (ds_bpermute costs an LDS
// round trip and these sit on the serial softmax critical path).
__device__ __forceinline__ float rows_max(float v) {
    const uint32_t u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const uint32_t w = __float_as_uint(m);
    auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const uint32_t w = __float_as_uint(m);
    auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

union frag8 {
    bf16x8 v;
    u32x4 u;
};

__device__ __forceinline__ bf16x8 pack8(const float* a, const float* b) {
    frag8 f;
    f.u = (u32x4){pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return f.v;
}

// store epilogue of a gradient row held as NDT column tiles (lane: columns dt*16 + g*4 .. +3): scale, round to bf16 and — cos given — apply the
// INVERSE rotary embedding (partner column + D/2 = tile dt + NDT/2 of the same lane) on the rounded values before the final rounding
template <int NDT>
__device__ __forceinline__ void store_grad_row(bf16_t* dst, const f32x4 (&v)[NDT], float scale, int g, const float* cos_row, const float* sin_row) {
    if (cos_row) {
#pragma unroll
        for (int dt = 0; dt < NDT / 2; ++dt) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cos_row + dt * 16 + g * 4), s4 = *reinterpret_cast<const f32x4*>(sin_row + dt * 16 + g * 4);
            float y1[4], y2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                rope_pair(bf16_to_f32(f32_to_bf16(v[dt][e] * scale)), bf16_to_f32(f32_to_bf16(v[dt + NDT / 2][e] * scale)), c4[e], -s4[e], y1[e], y2[e]);
            *reinterpret_cast<u32x2*>(dst + dt * 16 + g * 4) = (u32x2){pack_bf16x2(y1[0], y1[1]), pack_bf16x2(y1[2], y1[3])};
            *reinterpret_cast<u32x2*>(dst + (dt + NDT / 2) * 16 + g * 4) = (u32x2){pack_bf16x2(y2[0], y2[1]), pack_bf16x2(y2[2], y2[3])};
        }
        return;
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
        *reinterpret_cast<u32x2*>(dst + dt * 16 + g * 4) = (u32x2){pack_bf16x2(v[dt][0] * scale, v[dt][1] * scale), pack_bf16x2(v[dt][2] * scale, v[dt][3] * scale)};
}

// TRANSPOSED operand from a row-major LDS tile via the gfx950 hardware transpose read (ds_read_b64_tr_b16):
// the lane with l15 = lane & 15 receives, for output column c0 + l15, the 8 contraction rows
//     j < 4: ra + j      j >= 4: rb + (j - 4)
// Semantics (verified on MI355X, tools/probes/tr16.hip): within each 16-lane group lane i receives element (i & 3) of
// the 4 contiguous bf16 addressed by lane 4j + (i >> 2), for j = 0..3. So lane L addresses row (L >> 2) of the 4-row
// block and columns c0 + (L & 3) * 4 .. +3. Two 8-byte reads replace eight 2-byte gathers per MFMA operand.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8 gather_col(const bf16_t* tile, int ldt, int ra, int rb, int c0, int l15) {
    const int off = (l15 >> 2) * ldt + c0 + (l15 & 3) * 4;
    union { bf16x8 v; s16x4 h[2]; } f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + ra * ldt + off));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + rb * ldt + off));
    return f.v;
}

// cooperative load of KC rows x D bf16 (row stride src_ts elements) into a padded LDS tile; rows >= limit are clamped.
// All global loads of a thread are issued BEFORE the first LDS store: a load->store->load loop costs one full memory
// round trip per iteration (hipcc waits vmcnt(0) in front of every ds_write), which was most of these kernels' time.
template <int D, int NT = 256>
__device__ __forceinline__ void load_tile(bf16_t* tile, const bf16_t* src, int64_t src_ts, int64_t row0, int64_t limit) {
    constexpr int LDT = D + attn_pad(D), CPR = D / 8, NIT = KC * CPR / NT;
    static_assert(KC * CPR % NT == 0 && NIT >= 1, "whole 16-byte chunks per thread");
    u32x4 v[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int s = threadIdx.x + i * NT;
        const int r = s / CPR, c = s % CPR;
        int64_t gr = row0 + r;
        if (gr > limit - 1) gr = limit - 1;
        v[i] = *reinterpret_cast<const u32x4*>(src + gr * src_ts + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int s = threadIdx.x + i * NT;
        *reinterpret_cast<u32x4*>(tile + (s / CPR) * LDT + (s % CPR) * 8) = v[i];
    }
}

// the two halves of load_tile for a software pipeline (GUIDE T14: issue early, write late): fetch_tile issues the NEXT chunk's global loads
// right after the barrier that publishes the CURRENT chunk, so their latency runs under the current chunk's MFMAs; stash_tile writes them to
// the LDS tile after the next barrier. Costs KC * D * 2 / NT bytes of registers per tile and thread (16 B .. 64 B).
template <int D, int NT>
struct TileRegs { u32x4 v[KC * (D / 8) / NT]; };
template <int D, int NT>
__device__ __forceinline__ void fetch_tile(TileRegs<D, NT>& t, const bf16_t* src, int64_t src_ts, int64_t row0, int64_t limit) {
    constexpr int CPR = D / 8, NIT = KC * CPR / NT;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int s = threadIdx.x + i * NT;
        int64_t gr = row0 + s / CPR;
        if (gr > limit - 1) gr = limit - 1;
        t.v[i] = *reinterpret_cast<const u32x4*>(src + gr * src_ts + (s % CPR) * 8);
    }
}
template <int D, int NT>
__device__ __forceinline__ void stash_tile(bf16_t* tile, const TileRegs<D, NT>& t) {
    constexpr int LDT = D + attn_pad(D), CPR = D / 8, NIT = KC * CPR / NT;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int s = threadIdx.x + i * NT;
        *reinterpret_cast<u32x4*>(tile + (s / CPR) * LDT + (s % CPR) * 8) = t.v[i];
    }
}

// XCD-aware workgroup -> (row block, head, batch) map for the long-sequence kernels. The hardware hands consecutive workgroup ids to the eight XCDs
// round robin, and each XCD has its own 4 MB L2: with the plain (x = row block, y = head, z = batch) grid the 13 row blocks of one head land on
// eight different XCDs and every one of those L2s fetches that head's whole K / V (or Q / dO) — 3 GB per forward launch at T = 1664 where 0.45 GB
// are algorithmic. Launched 1-D (8 * ceil(heads / 8) * nx workgroups): XCD x owns a CONTIGUOUS range of (batch, head) pairs (adjacent query heads
// of a GQA group share their K / V through the same L2) and walks each head's row blocks back to back, heaviest (most keys) first.
struct AttnBlock { int64_t x, h, b; bool live; };
template <bool XMAP>
__device__ __forceinline__ AttnBlock attn_block(int64_t nx, int64_t ny, int64_t nz, bool heavy_first_is_last) {
    if constexpr (!XMAP) {
        return {(int64_t)blockIdx.x, (int64_t)blockIdx.y, (int64_t)blockIdx.z, true};
    } else {
        const int64_t L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int64_t nh = ny * nz, per = (nh + 7) >> 3;
        const int64_t hh = xcd * per + slot / nx;
        const int64_t xr = slot % nx;
        const bool live = hh < nh && slot / nx < per;
        return {heavy_first_is_last ? nx - 1 - xr : xr, hh % ny, hh / ny, live};
    }
}
__host__ inline unsigned attn_xmap_grid(int64_t nx, int64_t ny, int64_t nz) { return (unsigned)(8 * ((ny * nz + 7) / 8) * nx); }

#define NEG_BIG (-1.0e30f)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// =============================================================================================== forward
// NW waves x 16 query rows per workgroup. NW = 8 for long causal sequences (T >= 512; interleave / independent covariates on a Llama backbone):
// every K / V chunk a workgroup stages serves 128 instead of 64 queries — at T = 1664 the 64-row blocks re-read each head's K / V 13 times
// through L2 (5.6 GB per layer), which is what the kernels spent most of their time on.
template <int D, bool CAUSAL, bool DROP, int NW = 4>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const mtl_attn_fwd_args a) {
    constexpr int LDT = D + attn_pad(D), NKS = D / 32, NDT = D / 16;
    __shared__ __attribute__((aligned(16))) bf16_t ktile[KC * LDT];
    __shared__ __attribute__((aligned(16))) bf16_t vtile[KC * LDT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;   // (wave index in an SGPR: tile offsets, loop bounds and the mask test become scalar)
    const int64_t b = blockIdx.z, h = blockIdx.y, hk = h / (a.Hq / a.Hkv);
    const int64_t qblk0 = (int64_t)blockIdx.x * (NW * 16);
    const int64_t q0 = qblk0 + wave * 16;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * a.q_hs;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + hk * a.k_hs;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + hk * a.v_hs;

    int64_t qrow = q0 + l15;
    const bool q_valid = qrow < a.Tq;
    if (qrow > a.Tq - 1) qrow = a.Tq - 1;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * a.q_ts + ks * 32 + g * 8);

    f32x4 o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // running max is kept in the exp2 domain: m2 = max(score) * scale * log2(e); p = exp2(s * c - m2): one FMA + v_exp_f32
    const float c = a.scale * LOG2E;
    float m_run = NEG_BIG, l_run = 0.f;
    const uint32_t drop_thr = DROP ? drop_threshold(a.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    const uint32_t dbase = DROP ? drop_base(a.dropout_seed, (uint32_t)(b * a.Hq + h)) : 0u;

    // causal: key visible to query q iff key <= q + causal_off (causal_off = Tk - Tq aligns the diagonal bottom-right)
    const int64_t coff = a.causal_off;
    const int klim = (int)((CAUSAL && qrow + coff < a.Tk - 1) ? qrow + coff : a.Tk - 1);   // last visible key of the lane's query
    int64_t k_end = a.Tk;
    if (CAUSAL) {
        const int64_t lim = (qblk0 + NW * 16 < a.Tq ? qblk0 + NW * 16 : a.Tq) + coff;  // keys <= last query of the block
        k_end = lim < a.Tk ? lim : a.Tk;
    }
    const int64_t wave_qmin = q0 + coff;
    const int64_t wave_qmax = ((q0 + 15 < a.Tq - 1) ? q0 + 15 : a.Tq - 1) + coff;

    // NW = 8 (long causal sequences, 2 workgroups per CU): software pipeline over the key chunks — 880 -> 730 us from the wider workgroup, -> 678 us
    // with the pipeline at T = 1664 (per layer, Llama-2-7B, B = 16). The 4-wave kernels keep the plain load: four workgroups per CU already hide
    // the latency, and the pipeline's registers cost them occupancy (cross-attention forward at Tq = 128: 242 -> 280 us with it).
    constexpr bool PIPE = NW == 8;
    TileRegs<D, NW * 64> rk, rv;
    if constexpr (PIPE) {
        fetch_tile<D, NW * 64>(rk, K, a.k_ts, 0, a.Tk);
        fetch_tile<D, NW * 64>(rv, V, a.v_ts, 0, a.Tk);
    }
    for (int64_t kc0 = 0; kc0 < k_end; kc0 += KC) {
        __syncthreads();
        if constexpr (PIPE) {
            stash_tile<D, NW * 64>(ktile, rk);
            stash_tile<D, NW * 64>(vtile, rv);
        } else {
            load_tile<D, NW * 64>(ktile, K, a.k_ts, kc0, a.Tk);
            load_tile<D, NW * 64>(vtile, V, a.v_ts, kc0, a.Tk);
        }
        __syncthreads();
        if (PIPE && kc0 + KC < k_end) {      // the next chunk's loads fly under this chunk's MFMAs
            fetch_tile<D, NW * 64>(rk, K, a.k_ts, kc0 + KC, a.Tk);
            fetch_tile<D, NW * 64>(rv, V, a.v_ts, kc0 + KC, a.Tk);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int64_t kb = kc0 + sub * 32;
            if (kb >= k_end) continue;
            if (CAUSAL && kb > wave_qmax) continue;  // whole 32-key slab is above the diagonal for this wave
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = ktile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], s[t], 0, 0, 0);
            }
            float p[2][4];
            // masking is needed only on slabs that touch the diagonal or the end of the key range (wave-uniform test)
            const bool need_mask = (kb + 32 > a.Tk) || (CAUSAL && kb + 31 > wave_qmin);
            if (need_mask) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[t][r] = ((int)kb + t * 16 + g * 4 + r > klim) ? NEG_BIG : s[t][r] * c;
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[t][r] = s[t][r] * c;
            }
            float mx = fmaxf(fmaxf(fmaxf(p[0][0], p[0][1]), fmaxf(p[0][2], p[0][3])), fmaxf(fmaxf(p[1][0], p[1][1]), fmaxf(p[1][2], p[1][3])));
            mx = rows_max(mx);
            const float m_new = fmaxf(m_run, mx);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[t][r] = __builtin_amdgcn_exp2f(p[t][r] - m_new);
                    psum += p[t][r];
                }
            if (__any(m_new != m_run)) {   // rescale only when some row's running max moved (wave-uniform branch)
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
                m_run = m_new;
            }
            l_run += psum;
            if (DROP) {   // a lane's keys kb + t*16 + g*4 + {0,1,2,3} are one mask quad: one hash per tile
#pragma unroll
                for (int t = 0; t < 2; ++t)
                {
                    const uint2 w = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);   // absolute query index
                    p[t][0] = (w.x & 0xffffu) >= drop_thr ? p[t][0] * drop_scale : 0.f;
                    p[t][1] = (w.x >> 16) >= drop_thr ? p[t][1] * drop_scale : 0.f;
                    p[t][2] = (w.y & 0xffffu) >= drop_thr ? p[t][2] * drop_scale : 0.f;
                    p[t][3] = (w.y >> 16) >= drop_thr ? p[t][3] * drop_scale : 0.f;
                }
            }
            const bf16x8 pf = pack8(p[0], p[1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 vt = gather_col(vtile, LDT, sub * 32 + g * 4, sub * 32 + 16 + g * 4, dt * 16, l15);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf, o[dt], 0, 0, 0);
            }
        }
    }
    l_run = rows_sum(l_run);
    if (!q_valid) return;
    const float inv_l = 1.0f / l_run;
    bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + b * a.o_bs + h * a.o_hs + qrow * a.o_ts;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        u32x2 pk = {pack_bf16x2(o[dt][0] * inv_l, o[dt][1] * inv_l), pack_bf16x2(o[dt][2] * inv_l, o[dt][3] * inv_l)};
        *reinterpret_cast<u32x2*>(O + dt * 16 + g * 4) = pk;
    }
    if (a.o_f32) {       // fp32 copy for the backward's delta (see the header)
        float* O32 = a.o_f32 + b * a.o_bs + h * a.o_hs + qrow * a.o_ts;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<f32x4*>(O32 + dt * 16 + g * 4) = o[dt] * inv_l;
    }
    // natural-log sum-exp of the scaled scores
    if (g == 0 && a.lse) a.lse[(b * a.Hq + h) * (a.stat_stride ? a.stat_stride : a.Tq) + qrow] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;
}

// =============================================================================================== backward: dQ (+ delta)
template <int D, bool CAUSAL, bool DROP, int NW = 4, bool XMAP = false>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const mtl_attn_bwd_args a) {
    constexpr int LDT = D + attn_pad(D), NKS = D / 32, NDT = D / 16;
    __shared__ __attribute__((aligned(16))) bf16_t ktile[KC * LDT];
    __shared__ __attribute__((aligned(16))) bf16_t vtile[KC * LDT];
    const mtl_attn_fwd_args& f = a.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;   // (wave index in an SGPR: tile offsets, loop bounds and the mask test become scalar)
    const AttnBlock blk = attn_block<XMAP>((f.Tq + NW * 16 - 1) / (NW * 16), f.Hq, f.B, CAUSAL);
    if (!blk.live) return;
    const int64_t b = blk.b, h = blk.h, hk = h / (f.Hq / f.Hkv);
    const int64_t qblk0 = blk.x * (NW * 16);
    const int64_t q0 = qblk0 + wave * 16;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(f.q) + b * f.q_bs + h * f.q_hs;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(f.k) + b * f.k_bs + hk * f.k_hs;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(f.v) + b * f.v_bs + hk * f.v_hs;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(f.o) + b * f.o_bs + h * f.o_hs;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_bs + h * a.do_hs;

    int64_t qrow = q0 + l15;
    const bool q_valid = qrow < f.Tq;
    if (qrow > f.Tq - 1) qrow = f.Tq - 1;
    bf16x8 qf[NKS], dof[NKS];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * f.q_ts + ks * 32 + g * 8);
        frag8 d8, o8;
        d8.v = *reinterpret_cast<const bf16x8*>(dO + qrow * a.do_ts + ks * 32 + g * 8);
        o8.v = *reinterpret_cast<const bf16x8*>(O + qrow * f.o_ts + ks * 32 + g * 8);
        dof[ks] = d8.v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dl += __uint_as_float(d8.u[e] << 16) * __uint_as_float(o8.u[e] << 16);
            dl += __uint_as_float(d8.u[e] & 0xffff0000u) * __uint_as_float(o8.u[e] & 0xffff0000u);
        }
    }
    dl = rows_sum(dl);
    const int64_t stat_idx = (b * f.Hq + h) * (f.stat_stride ? f.stat_stride : f.Tq) + qrow;
    const float c = f.scale * LOG2E;
    const float lse2 = f.lse[stat_idx] * LOG2E;   // exp(s*scale - lse) == exp2(s*c - lse2)
    const uint32_t drop_thr = DROP ? drop_threshold(f.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    const uint32_t dbase = DROP ? drop_base(f.dropout_seed, (uint32_t)(b * f.Hq + h)) : 0u;

    f32x4 dq[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t coff = f.causal_off;
    const int klim = (int)((CAUSAL && qrow + coff < f.Tk - 1) ? qrow + coff : f.Tk - 1);   // last visible key of the lane's query
    int64_t k_end = f.Tk;
    if (CAUSAL) {
        const int64_t lim = (qblk0 + NW * 16 < f.Tq ? qblk0 + NW * 16 : f.Tq) + coff;
        k_end = lim < f.Tk ? lim : f.Tk;
    }
    const int64_t wave_qmax = ((q0 + 15 < f.Tq - 1) ? q0 + 15 : f.Tq - 1) + coff;

    if (!CAUSAL && f.o_f32) {
        // delta from the fp32 forward output: no extra pass
        const float* O32 = f.o_f32 + b * f.o_bs + h * f.o_hs + qrow * f.o_ts;
        float acc = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            frag8 d8;
            d8.v = dof[ks];
            const f32x4 oa = *reinterpret_cast<const f32x4*>(O32 + ks * 32 + g * 8), ob = *reinterpret_cast<const f32x4*>(O32 + ks * 32 + g * 8 + 4);
            acc += __uint_as_float(d8.u[0] << 16) * oa[0] + __uint_as_float(d8.u[0] & 0xffff0000u) * oa[1];
            acc += __uint_as_float(d8.u[1] << 16) * oa[2] + __uint_as_float(d8.u[1] & 0xffff0000u) * oa[3];
            acc += __uint_as_float(d8.u[2] << 16) * ob[0] + __uint_as_float(d8.u[2] & 0xffff0000u) * ob[1];
            acc += __uint_as_float(d8.u[3] << 16) * ob[2] + __uint_as_float(d8.u[3] & 0xffff0000u) * ob[3];
        }
        dl = rows_sum(acc);
    } else if (!CAUSAL || MTL_CONSISTENT_DELTA) {
        // CONSISTENT delta for the (unmasked) reprogramming attention: delta_q = sum_s p_qs * dP_qs from the very p and dP the main loop
        // uses, so that sum_s dS_qs = 0 holds to fp32 round-off, as it does in an unfused softmax backward. dO . O with the bf16-rounded
        // O is the same number only to ~2^-9, and with near-uniform probabilities over the vocabulary prototypes (keys that share a
        // large common component) the resulting ~1e-3 * delta * sum_s p_s K_s is a coherent error along the mean key: measured on the
        // reference goldens it doubled the error of dQ (query-projection / patch-embedding gradients 2-3 % instead of 1 %).
        // One extra pass over K and V (S and dP MFMAs only); the causal self-attention kernels keep dO . O.
        float acc = 0.f;
        for (int64_t kc0 = 0; kc0 < k_end; kc0 += KC) {
            __syncthreads();
            load_tile<D, NW * 64>(ktile, K, f.k_ts, kc0, f.Tk);
            load_tile<D, NW * 64>(vtile, V, f.v_ts, kc0, f.Tk);
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int64_t kb = kc0 + sub * 32;
                if (kb >= k_end) continue;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    const bf16_t* kr = ktile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
                    const bf16_t* vr = vtile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(vr + ks * 32), dof[ks], dp, 0, 0, 0);
                    }
                    uint2 dw = make_uint2(0u, 0u);      // mask words of the lane's four keys
                    if (DROP) dw = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = ((int)kb + t * 16 + g * 4 + r > klim) ? 0.f : __builtin_amdgcn_exp2f(s[r] * c - lse2);
                        float dpv = dp[r];
                        if (DROP) dpv = drop_field(dw, (uint32_t)r) >= drop_thr ? dpv * drop_scale : 0.f;
                        acc += p * dpv;
                    }
                }
            }
        }
        dl = rows_sum(acc);
    }
    if (g == 0 && q_valid) a.delta[stat_idx] = dl;

    // (no software pipeline here: measured 989 -> 1106 us at T = 1664 with it, the extra registers cost a wave per SIMD)
    for (int64_t kc0 = 0; kc0 < k_end; kc0 += KC) {
        __syncthreads();
        load_tile<D, NW * 64>(ktile, K, f.k_ts, kc0, f.Tk);
        load_tile<D, NW * 64>(vtile, V, f.v_ts, kc0, f.Tk);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int64_t kb = kc0 + sub * 32;
            if (kb >= k_end) continue;
            if (CAUSAL && kb > wave_qmax) continue;
            float ds[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = ktile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
                const bf16_t* vr = vtile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(vr + ks * 32), dof[ks], dp, 0, 0, 0);
                }
                uint2 dw = make_uint2(0u, 0u);      // mask words of the lane's four keys
                if (DROP) dw = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool masked = (int)kb + t * 16 + g * 4 + r > klim;
                    const float p = masked ? 0.f : __builtin_amdgcn_exp2f(s[r] * c - lse2);
                    float dpv = dp[r];
                    if (DROP) dpv = drop_field(dw, (uint32_t)r) >= drop_thr ? dpv * drop_scale : 0.f;
                    ds[t][r] = p * (dpv - dl);
                }
            }
            const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 kt = gather_col(ktile, LDT, sub * 32 + g * 4, sub * 32 + 16 + g * 4, dt * 16, l15);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[dt], 0, 0, 0);
            }
        }
    }
    if (!q_valid) return;
    bf16_t* DQ = reinterpret_cast<bf16_t*>(a.dq) + b * a.dq_bs + h * a.dq_hs + qrow * a.dq_ts;
    store_grad_row<NDT>(DQ, dq, f.scale, g, a.rope_cos ? a.rope_cos + (qrow + coff) * D : nullptr, a.rope_cos ? a.rope_sin + (qrow + coff) * D : nullptr);
}

// =============================================================================================== backward: dK, dV
// One workgroup per (64-key tile, kv head, batch or ALL batches when K/V are batch-shared); loops over the query
// heads of the GQA group and over 64-query chunks. Lane owns key = lane & 15 of its wave's 16 keys.
template <int D, bool CAUSAL, bool DROP, int NW = 4, bool XMAP = false>
__global__ __launch_bounds__(NW * 64, (D >= 128 && NW == 4) ? 2 : 1) void attn_bwd_dkv_kernel(const mtl_attn_bwd_args a) {
    constexpr int LDT = D + attn_pad(D), NKS = D / 32, NDT = D / 16;
    __shared__ __attribute__((aligned(16))) bf16_t qtile[KC * LDT];
    __shared__ __attribute__((aligned(16))) bf16_t dotile[KC * LDT];
    __shared__ __attribute__((aligned(16))) float lse_s[KC];
    __shared__ __attribute__((aligned(16))) float delta_s[KC];
    const mtl_attn_fwd_args& f = a.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;   // (wave index in an SGPR: tile offsets, loop bounds and the mask test become scalar)
    // (XMAP: 1-D launch, per-sample K / V only — a key block's workgroups of one kv head stay on one XCD, whose L2 then holds that head's Q / dO)
    const AttnBlock blk = attn_block<XMAP>((f.Tk - a.kv_row0 + NW * 16 - 1) / (NW * 16), f.Hkv, f.B, false);
    if (!blk.live) return;
    const int64_t bx = blk.x, hk = blk.h, bz = blk.b;
    const int group = (int)(f.Hq / f.Hkv);
    const bool shared_kv = (f.k_bs == 0);
    // shared K/V (reprogramming attention): blockIdx.z is a CHUNK of the batch; partial dK/dV are accumulated into the
    // fp32 workspace with hardware float atomics and converted afterwards (dkv_convert_kernel)
    const int64_t chunk = shared_kv ? (f.B + gridDim.z - 1) / gridDim.z : 1;
    const int64_t b_begin = shared_kv ? bz * chunk : bz;
    const int64_t b_end = shared_kv ? ((b_begin + chunk < f.B) ? b_begin + chunk : f.B) : bz + 1;
    const int64_t coff = f.causal_off;
    const int64_t kblk0 = a.kv_row0 + bx * (NW * 16);   // keys below kv_row0 need no gradient (pruned)
    const int64_t k0 = kblk0 + wave * 16;
    int64_t krow = k0 + l15;
    const bool k_valid = krow < f.Tk;
    const int qhi = (int)f.Tq - 1, qlo = CAUSAL ? (int)(krow - coff) : 0;      // queries that see the lane's key
    const int64_t key_u = krow;                       // (unclamped: the mask words are exchanged inside lane quads, also by lanes past the last key)
    const bool quad_ok = (k0 & 3) == 0;               // the four keys of a lane quad share their mask quad
    if (krow > f.Tk - 1) krow = f.Tk - 1;

    const float c = f.scale * LOG2E;
    const uint32_t drop_thr = DROP ? drop_threshold(f.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    f32x4 dk[NDT], dv[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
        dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    for (int64_t b = b_begin; b < b_end; ++b) {
        const bf16_t* K = reinterpret_cast<const bf16_t*>(f.k) + b * f.k_bs + hk * f.k_hs;
        const bf16_t* V = reinterpret_cast<const bf16_t*>(f.v) + b * f.v_bs + hk * f.v_hs;
        bf16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(K + krow * f.k_ts + ks * 32 + g * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(V + krow * f.v_ts + ks * 32 + g * 8);
        }
        for (int hg = 0; hg < group; ++hg) {
            const int64_t h = hk * group + hg;
            const uint32_t dbase_h = DROP ? drop_base(f.dropout_seed, (uint32_t)(b * f.Hq + h)) : 0u;
            const bf16_t* Q = reinterpret_cast<const bf16_t*>(f.q) + b * f.q_bs + h * f.q_hs;
            const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_bs + h * a.do_hs;
            const int64_t stat0 = (b * f.Hq + h) * (f.stat_stride ? f.stat_stride : f.Tq);
            // queries q with q + coff < kblk0 never see this key tile
            const int64_t qc_begin = (CAUSAL && kblk0 > coff) ? ((kblk0 - coff) / KC) * KC : 0;
            // software pipeline over the query chunks of this head: chunk c + 1's Q / dO rows and statistics are fetched while chunk c computes
            // (this kernel runs at two waves per SIMD: nothing else hides the load round trip; 1621 -> 1523 us at T = 1664, 470 -> 455 us for the
            //  batch-shared reprogramming keys)
            TileRegs<D, NW * 64> rq, rdo;
            float r_lse = 0.f, r_delta = 0.f;
            auto fetch_chunk = [&](int64_t qc) __attribute__((always_inline)) {
                fetch_tile<D, NW * 64>(rq, Q, f.q_ts, qc, f.Tq);
                fetch_tile<D, NW * 64>(rdo, dO, a.do_ts, qc, f.Tq);
                if (threadIdx.x < KC) {
                    int64_t qq = qc + threadIdx.x;
                    if (qq > f.Tq - 1) qq = f.Tq - 1;
                    r_lse = f.lse[stat0 + qq] * LOG2E;
                    r_delta = a.delta[stat0 + qq];
                }
            };
            if (qc_begin < f.Tq) fetch_chunk(qc_begin);
            for (int64_t qc0 = qc_begin; qc0 < f.Tq; qc0 += KC) {
                __syncthreads();
                stash_tile<D, NW * 64>(qtile, rq);
                stash_tile<D, NW * 64>(dotile, rdo);
                if (threadIdx.x < KC) {
                    lse_s[threadIdx.x] = r_lse;
                    delta_s[threadIdx.x] = r_delta;
                }
                __syncthreads();
                if (qc0 + KC < f.Tq) fetch_chunk(qc0 + KC);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int64_t qb = qc0 + sub * 32;
                    if (qb >= f.Tq) continue;
                    if (CAUSAL && qb + 31 + coff < k0) continue;  // every query of the slab precedes this wave's keys
                    float p[2][4], ds[2][4];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                        const bf16_t* qr = qtile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
                        const bf16_t* dr = dotile + (sub * 32 + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) {
                            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qr + ks * 32), kf[ks], s, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(dr + ks * 32), vf[ks], dp, 0, 0, 0);
                        }
                        uint32_t fld[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
                        if (DROP) {
                            const uint32_t a0 = (uint32_t)((int)qc0 + sub * 32 + t * 16 + g * 4 + (int)coff);
                            if (quad_ok) drop_fields_shared(dbase_h, a0, (uint32_t)key_u, fld);
                            else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) fld[r] = drop_field(drop_quad(dbase_h, a0 + r, (uint32_t)krow >> 2), (uint32_t)krow);
                            }
                        }
                        // the lane's four queries' statistics in two 16-byte reads, the exponential UNCONDITIONAL and a select afterwards: with the LDS
                        // reads inside the conditional hipcc emitted a BRANCH per element (two ds_read_b32 + an lgkmcnt(0) wait each: eight serialized
                        // LDS round trips per slab between the MFMA groups)
                        const int ql0 = sub * 32 + t * 16 + g * 4;
                        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + ql0), dl4 = *reinterpret_cast<const f32x4*>(delta_s + ql0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = (int)qc0 + ql0 + r;
                            const bool masked = q > qhi || q < qlo;
                            const float e = __builtin_amdgcn_exp2f(s[r] * c - lse4[r]);
                            const float pv = masked ? 0.f : e;
                            float keep = 1.0f;
                            if (DROP) keep = fld[r] >= drop_thr ? drop_scale : 0.f;
                            p[t][r] = pv * keep;                               // feeds dV = (dropped P)^T dO
                            ds[t][r] = pv * (dp[r] * keep - dl4[r]);           // feeds dK
                        }
                    }
                    const bf16x8 pf = pack8(p[0], p[1]);
                    const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const bf16x8 dot = gather_col(dotile, LDT, sub * 32 + g * 4, sub * 32 + 16 + g * 4, dt * 16, l15);
                        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf, dv[dt], 0, 0, 0);
                        const bf16x8 qt = gather_col(qtile, LDT, sub * 32 + g * 4, sub * 32 + 16 + g * 4, dt * 16, l15);
                        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dk[dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!k_valid) return;
    if (shared_kv && gridDim.z > 1) {   // partial slab of this batch chunk: [split][2][Tk][Hkv][D] fp32, plain 16-B stores
        const int64_t n = f.Tk * f.Hkv * D;
        float* wk = a.dkv_ws + bz * 2 * n + (krow * f.Hkv + hk) * D;
        float* wv = wk + n;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            *reinterpret_cast<float4*>(wk + dt * 16 + g * 4) = make_float4(dk[dt][0] * f.scale, dk[dt][1] * f.scale, dk[dt][2] * f.scale, dk[dt][3] * f.scale);
            *reinterpret_cast<float4*>(wv + dt * 16 + g * 4) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
        }
        return;
    }
    const int64_t bo = shared_kv ? 0 : bz;
    bf16_t* DK = reinterpret_cast<bf16_t*>(a.dk) + bo * a.dk_bs + hk * a.dk_hs + krow * a.dk_ts;
    bf16_t* DV = reinterpret_cast<bf16_t*>(a.dv) + bo * a.dv_bs + hk * a.dv_hs + krow * a.dv_ts;
    store_grad_row<NDT>(DK, dk, f.scale, g, a.rope_cos ? a.rope_cos + krow * D : nullptr, a.rope_cos ? a.rope_sin + krow * D : nullptr);
    store_grad_row<NDT>(DV, dv, 1.0f, g, nullptr, nullptr);
}

// =============================================================================================== long sequences: 32 query rows per wave
// Causal self-attention at T >= 256 when K / V do not fit the LDS (interleave / independent covariates on a Llama backbone: T = 1.7 k .. 6.5 k).
// The 16-row kernels above read one LDS fragment per MFMA and reduce / rescale per 16 x 32 slab; this family follows GUIDE "Fused attention
// prefill": v_mfma_f32_32x32x16_bf16 with the scores TRANSPOSED (S^T = K Q^T: lane = one query column, 16 keys of a 32-key tile in its
// registers), 32 query rows per wave, 64-key chunks through a DOUBLE-BUFFERED LDS ring (one barrier per chunk; the next chunk travels
// global -> registers under this chunk's MFMAs and is written to the other buffer after the barrier: GUIDE T14), one max / rescale decision per
// 64 keys, and the exponentiated scores are the B operand of the P V MFMA as they stand (the contraction index of that MFMA is mapped to
// keys as the accumulator layout hands them out: register r of lane-half hi holds key (r & 3) + 8 (r >> 2) + 4 hi of the tile).
// Per MAC: half the LDS fragment bytes, half the MFMA issues, a quarter of the cross-lane reductions of the 16-row kernels.
// LDS rows: K [64][D + 8] (ds_read_b128 rows: stride = 4 * odd dwords -> the four 16-lane groups of a b128 read hit 16 distinct slots),
//           V [64][D + 32] (ds_read_b64_tr_b16 gathers: a 32-lane group covers 4 rows x 64 B; stride = 16 mod 64 dwords -> no bank twice).
typedef __attribute__((ext_vector_type(16))) float f32x16;
#ifndef MTL_W32_DIAG
#define MTL_W32_DIAG 0      // diagnostic builds only (tools/build_variant.sh): 1 no exp / sum, 2 no P V MFMAs, 4 no Q K MFMAs, 8 no global -> LDS staging in the loop, 16 no barrier
#endif

template <int D, int NT, int LD>
__device__ __forceinline__ void stash_tile_ld(bf16_t* tile, const TileRegs<D, NT>& t) {
    constexpr int CPR = D / 8, NIT = KC * CPR / NT;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int s = threadIdx.x + i * NT;
        *reinterpret_cast<u32x4*>(tile + (s / CPR) * LD + (s % CPR) * 8) = t.v[i];
    }
}

// fetch_tile with the address arithmetic hoisted out of the chunk loop: a thread's NIT source pointers are formed once (row0 = 0) and every chunk
// adds chunk * KC * stride to them; only a chunk that crosses the end of the sequence (wave-uniform test) takes the clamping form. The 64-bit
// multiplies of the general form cost ~60 VALU instructions per chunk and thread — a fifth of the loop's issue slots.
template <int D, int NT>
struct TilePtrs { const bf16_t* p[KC * (D / 8) / NT]; };
template <int D, int NT>
__device__ __forceinline__ void tile_ptrs(TilePtrs<D, NT>& tp, const bf16_t* src, int64_t src_ts) {
    constexpr int CPR = D / 8, NIT = KC * CPR / NT;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int s = threadIdx.x + i * NT;
        tp.p[i] = src + (int64_t)(s / CPR) * src_ts + (s % CPR) * 8;
    }
}
template <int D, int NT>
__device__ __forceinline__ void fetch_tile_at(TileRegs<D, NT>& t, const TilePtrs<D, NT>& tp, const bf16_t* src, int64_t src_ts, int64_t row0, int64_t limit) {
    constexpr int NIT = KC * (D / 8) / NT;
    if (row0 + KC <= limit) {
        const int64_t off = row0 * src_ts;
#pragma unroll
        for (int i = 0; i < NIT; ++i) t.v[i] = *reinterpret_cast<const u32x4*>(tp.p[i] + off);
    } else {
        fetch_tile<D, NT>(t, src, src_ts, row0, limit);
    }
}

// row maximum / sum across the two lane halves (lanes l and l + 32 hold the same query)
__device__ __forceinline__ float halves_max(float v) {
    const uint32_t w = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float halves_sum(float v) {
    const uint32_t w = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

template <int D, bool CAUSAL, int NW, bool XMAP = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_w32_kernel(const mtl_attn_fwd_args a) {
    constexpr int LDK = D + 8, LDV = D + 32, NKS = D / 16, NDB = D / 32, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* kbuf = reinterpret_cast<bf16_t*>(smem_raw);      // [2][KC * LDK]
    bf16_t* vbuf = kbuf + 2 * KC * LDK;                      // [2][KC * LDV]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, hi = lane >> 5;
    const AttnBlock blk = attn_block<XMAP>((a.Tq + NW * 32 - 1) / (NW * 32), a.Hq, a.B, CAUSAL);
    if (!blk.live) return;
    const int64_t b = blk.b, h = blk.h, hk = h / (a.Hq / a.Hkv);
    const int64_t qblk0 = blk.x * (NW * 32);
    const int64_t q0 = qblk0 + wave * 32;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * a.q_hs;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + hk * a.k_hs;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + hk * a.v_hs;

    int64_t qrow = q0 + l31;
    const bool q_valid = qrow < a.Tq;
    if (qrow > a.Tq - 1) qrow = a.Tq - 1;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * a.q_ts + ks * 16 + hi * 8);

    f32x16 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    const float c = a.scale * LOG2E;
    float m_run = NEG_BIG, l_run = 0.f;      // l_run: this lane's 16-of-32 keys per tile only; the halves are added at the end

    const int64_t coff = a.causal_off;
    const int klim = (int)((CAUSAL && qrow + coff < a.Tk - 1) ? qrow + coff : a.Tk - 1);   // last visible key of the lane's query
    int64_t k_end = a.Tk;
    if (CAUSAL) {
        const int64_t lim = (qblk0 + NW * 32 < a.Tq ? qblk0 + NW * 32 : a.Tq) + coff;  // keys <= last query of the block
        k_end = lim < a.Tk ? lim : a.Tk;
    }
    const int64_t wave_qmin = q0 + coff;
    const int64_t wave_qmax = ((q0 + 31 < a.Tq - 1) ? q0 + 31 : a.Tq - 1) + coff;
    const bool wave_live = q0 < a.Tq;

    TileRegs<D, NT> rk, rv;
    TilePtrs<D, NT> pk, pv;
    tile_ptrs<D, NT>(pk, K, a.k_ts);
    tile_ptrs<D, NT>(pv, V, a.v_ts);
    fetch_tile_at<D, NT>(rk, pk, K, a.k_ts, 0, a.Tk);
    fetch_tile_at<D, NT>(rv, pv, V, a.v_ts, 0, a.Tk);
    stash_tile_ld<D, NT, LDK>(kbuf, rk);
    stash_tile_ld<D, NT, LDV>(vbuf, rv);
    // The query fragments were requested before chunk 0 and have landed by now (the stores above waited for every load). Tell the compiler: its
    // wait counters are in-order, and with the fragments still "pending" on the loop's entry edge it guards every first use inside the loop
    // with vmcnt(7 - ks) — i.e. each Q K^T MFMA waited for one of the CURRENT iteration's prefetch loads to come back from L2 / HBM.
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
    if (KC < k_end) {
        fetch_tile_at<D, NT>(rk, pk, K, a.k_ts, KC, a.Tk);
        fetch_tile_at<D, NT>(rv, pv, V, a.v_ts, KC, a.Tk);
    }
    int cur = 0;
    for (int64_t kc0 = 0; kc0 < k_end; kc0 += KC, cur ^= 1) {
        if (!(MTL_W32_DIAG & 16)) __syncthreads();        // chunk kc0 is visible in buffer `cur`; every wave has finished reading the other buffer
        if (!(MTL_W32_DIAG & 8) && kc0 + KC < k_end) {
            stash_tile_ld<D, NT, LDK>(kbuf + (cur ^ 1) * KC * LDK, rk);
            stash_tile_ld<D, NT, LDV>(vbuf + (cur ^ 1) * KC * LDV, rv);
            if (kc0 + 2 * KC < k_end) {      // two chunks ahead: flies under this chunk's and the next chunk's MFMAs
                fetch_tile_at<D, NT>(rk, pk, K, a.k_ts, kc0 + 2 * KC, a.Tk);
                fetch_tile_at<D, NT>(rv, pv, V, a.v_ts, kc0 + 2 * KC, a.Tk);
            }
        }
        if (!wave_live || (CAUSAL && kc0 > wave_qmax)) continue;     // every key of the chunk lies above this wave's diagonal (wave-uniform)
        const bf16_t* kt = kbuf + cur * KC * LDK;
        const bf16_t* vt = vbuf + cur * KC * LDV;
        // ---- S^T = K Q^T for the two 32-key tiles of the chunk. Fragment reads run TWO MFMAs ahead of their use through a three-deep register
        // ring, and the two tiles' accumulator chains alternate (hipcc by itself emits read -> wait -> MFMA per fragment on one chain: every
        // MFMA then waits out an LDS round trip). sched_barrier pins the order; the waits that come out are counted ones (lgkmcnt(2)).
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        {
            const bf16_t* kr = kt + l31 * LDK + hi * 8;
            constexpr int RING = 4;                  // fragments requested RING - 1 units ahead of their MFMA
            bf16x8 kf[RING];
#pragma unroll
            for (int i = 0; i < RING - 1; ++i) kf[i] = *reinterpret_cast<const bf16x8*>(kr + (i & 1) * 32 * LDK + (i >> 1) * 16);
#pragma unroll
            for (int i = 0; i < 2 * NKS; ++i) {      // i = 2 ks + t
                constexpr int A = RING - 1;
                if (i + A < 2 * NKS) kf[(i + A) % RING] = *reinterpret_cast<const bf16x8*>(kr + ((i + A) & 1) * 32 * LDK + ((i + A) >> 1) * 16);
                if (!(MTL_W32_DIAG & 4)) s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % RING], qf[i >> 1], s[i & 1], 0, 0, 0);
                else s[i & 1][i & 15] += __builtin_bit_cast(float, ((u32x4)__builtin_bit_cast(u32x4, kf[i % RING]))[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- online softmax over the chunk's 64 keys (this lane: 2 x 16 of them). The maximum is taken on the raw scores (c > 0) and the
        // scale rides in the exponent's FMA: one v_max3 per pair, one v_fma + v_exp + v_add per element
        const bool need_mask = (kc0 + KC > a.Tk) || (CAUSAL && kc0 + KC - 1 > wave_qmin);
        if (need_mask) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (int)kc0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key > klim) s[t][r] = NEG_BIG;
                }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = halves_max(mx) * c;           // (NEG_BIG * c stays far below every real score)
        const float m_new = fmaxf(m_run, mx);
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!(MTL_W32_DIAG & 1)) {
                    s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c, -m_new));
                    psum += s[t][r];
                }
            }
        if (__any(m_new != m_run)) {   // rescale only when some row's running max moved (wave-uniform branch)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            m_run = m_new;
        }
        l_run += psum;
        // ---- O^T += V^T P^T: contraction block (t, cb) = keys 32 t + 16 cb + {(j & 3) + 8 (j >> 2) + 4 hi}: registers 8 cb .. 8 cb + 7 of tile t
        bf16x8 pf[4];
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            const int t = tc >> 1, cb = tc & 1;
            frag8 f;
            f.u = (u32x4){pack_bf16x2(s[t][8 * cb + 0], s[t][8 * cb + 1]), pack_bf16x2(s[t][8 * cb + 2], s[t][8 * cb + 3]),
                          pack_bf16x2(s[t][8 * cb + 4], s[t][8 * cb + 5]), pack_bf16x2(s[t][8 * cb + 6], s[t][8 * cb + 7])};
            pf[tc] = f.v;
        }
        {
            // 16-lane gather group: d half of the 32-row block (lane bit 4), key half hi (lane bit 5). Unit i = tc * NDB + db: consecutive MFMAs
            // go to DIFFERENT accumulators, their V^T fragments are gathered two units ahead
            const bf16_t* vg = vt + (4 * hi + ((lane & 15) >> 2)) * LDV + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
            auto vfrag = [&](int i) __attribute__((always_inline)) {
                const bf16_t* p = vg + (i / NDB) * 16 * LDV + (i % NDB) * 32;
                union { bf16x8 v; s16x4 h[2]; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * LDV));
                return f.v;
            };
            constexpr int RING = 4;
            bf16x8 vf[RING];
#pragma unroll
            for (int i = 0; i < RING - 1; ++i) vf[i] = vfrag(i);
#pragma unroll
            for (int i = 0; i < 4 * NDB; ++i) {
                constexpr int A = RING - 1;
                if (i + A < 4 * NDB) vf[(i + A) % RING] = vfrag(i + A);
                if (!(MTL_W32_DIAG & 2)) o[i % NDB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % RING], pf[i / NDB], o[i % NDB], 0, 0, 0);
                else o[i % NDB][i & 15] += __builtin_bit_cast(float, ((u32x4)__builtin_bit_cast(u32x4, vf[i % RING]))[0]) + __builtin_bit_cast(float, ((u32x4)__builtin_bit_cast(u32x4, pf[i / NDB]))[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    l_run = halves_sum(l_run);
    if (!q_valid) return;
    const float inv_l = 1.0f / l_run;
    bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + b * a.o_bs + h * a.o_hs + qrow * a.o_ts;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {      // registers 4 rg .. 4 rg + 3 = feature columns 32 db + 8 rg + 4 hi + {0, 1, 2, 3}
            u32x2 pk = {pack_bf16x2(o[db][4 * rg] * inv_l, o[db][4 * rg + 1] * inv_l), pack_bf16x2(o[db][4 * rg + 2] * inv_l, o[db][4 * rg + 3] * inv_l)};
            *reinterpret_cast<u32x2*>(O + db * 32 + rg * 8 + hi * 4) = pk;
        }
    if (hi == 0 && a.lse) a.lse[(b * a.Hq + h) * (a.stat_stride ? a.stat_stride : a.Tq) + qrow] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;
}

// ---- dQ of the long causal sequences in the same structure (32 query rows per wave, 32x32x16 MFMAs, double-buffered 64-key chunks).
// Per 32-key tile: S^T = K Q^T and dP^T = V dO^T (two accumulator chains, alternating), p = exp2(s c - lse2), dS = p (dP - delta), and
// dQ^T += K^T dS^T with the K^T fragments gathered from the SAME K tile (row stride D + 16: 2-way conflicts for both of its read shapes; a
// second image of K would cost the second workgroup per CU). delta = dO . O is formed from the wave's own fragments and written for the dK/dV kernel.
template <int NDB>
__device__ __forceinline__ void store_grad_row_w32(bf16_t* dst, const f32x16 (&v)[NDB], float scale, int hi, const float* cos_row, const float* sin_row) {
    // lane (query, hi): register 4 rg + e of block db = feature column 32 db + 8 rg + 4 hi + e; the rotary partner column + D/2 is block db + NDB/2
    if (cos_row) {
#pragma unroll
        for (int db = 0; db < NDB / 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int col = db * 32 + rg * 8 + hi * 4;
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(cos_row + col), s4 = *reinterpret_cast<const f32x4*>(sin_row + col);
                float y1[4], y2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    rope_pair(bf16_to_f32(f32_to_bf16(v[db][4 * rg + e] * scale)), bf16_to_f32(f32_to_bf16(v[db + NDB / 2][4 * rg + e] * scale)), c4[e], -s4[e], y1[e], y2[e]);
                *reinterpret_cast<u32x2*>(dst + col) = (u32x2){pack_bf16x2(y1[0], y1[1]), pack_bf16x2(y1[2], y1[3])};
                *reinterpret_cast<u32x2*>(dst + col + NDB * 16) = (u32x2){pack_bf16x2(y2[0], y2[1]), pack_bf16x2(y2[2], y2[3])};
            }
        return;
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            *reinterpret_cast<u32x2*>(dst + db * 32 + rg * 8 + hi * 4) =
                (u32x2){pack_bf16x2(v[db][4 * rg] * scale, v[db][4 * rg + 1] * scale), pack_bf16x2(v[db][4 * rg + 2] * scale, v[db][4 * rg + 3] * scale)};
}

template <int D, int NW, bool XMAP = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_w32_kernel(const mtl_attn_bwd_args a) {
    constexpr int LDK = D + 16, LDV = D + 8, NKS = D / 16, NDB = D / 32, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* kbuf = reinterpret_cast<bf16_t*>(smem_raw);      // [2][KC * LDK]
    bf16_t* vbuf = kbuf + 2 * KC * LDK;                      // [2][KC * LDV]
    const mtl_attn_fwd_args& f = a.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, hi = lane >> 5;
    const AttnBlock blk = attn_block<XMAP>((f.Tq + NW * 32 - 1) / (NW * 32), f.Hq, f.B, true);
    if (!blk.live) return;
    const int64_t b = blk.b, h = blk.h, hk = h / (f.Hq / f.Hkv);
    const int64_t qblk0 = blk.x * (NW * 32);
    const int64_t q0 = qblk0 + wave * 32;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(f.q) + b * f.q_bs + h * f.q_hs;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(f.k) + b * f.k_bs + hk * f.k_hs;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(f.v) + b * f.v_bs + hk * f.v_hs;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(f.o) + b * f.o_bs + h * f.o_hs;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_bs + h * a.do_hs;

    int64_t qrow = q0 + l31;
    const bool q_valid = qrow < f.Tq;
    if (qrow > f.Tq - 1) qrow = f.Tq - 1;
    bf16x8 qf[NKS], dof[NKS];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * f.q_ts + ks * 16 + hi * 8);
        frag8 d8, o8;
        d8.v = *reinterpret_cast<const bf16x8*>(dO + qrow * a.do_ts + ks * 16 + hi * 8);
        o8.v = *reinterpret_cast<const bf16x8*>(O + qrow * f.o_ts + ks * 16 + hi * 8);
        dof[ks] = d8.v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dl += __uint_as_float(d8.u[e] << 16) * __uint_as_float(o8.u[e] << 16);
            dl += __uint_as_float(d8.u[e] & 0xffff0000u) * __uint_as_float(o8.u[e] & 0xffff0000u);
        }
    }
    dl = halves_sum(dl);                                    // delta = dO . O over the whole row (the two lane halves hold disjoint columns)
    const int64_t stat_idx = (b * f.Hq + h) * (f.stat_stride ? f.stat_stride : f.Tq) + qrow;
    const float c = f.scale * LOG2E;
    const float lse2 = f.lse[stat_idx] * LOG2E;             // exp(s * scale - lse) == exp2(s * c - lse2)
    if (hi == 0 && q_valid) a.delta[stat_idx] = dl;

    f32x16 dq[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    const int64_t coff = f.causal_off;
    const int klim = (int)((qrow + coff < f.Tk - 1) ? qrow + coff : f.Tk - 1);   // last visible key of the lane's query
    int64_t k_end = f.Tk;
    {
        const int64_t lim = (qblk0 + NW * 32 < f.Tq ? qblk0 + NW * 32 : f.Tq) + coff;
        k_end = lim < f.Tk ? lim : f.Tk;
    }
    const int64_t wave_qmin = q0 + coff;
    const int64_t wave_qmax = ((q0 + 31 < f.Tq - 1) ? q0 + 31 : f.Tq - 1) + coff;
    const bool wave_live = q0 < f.Tq;

    TileRegs<D, NT> rk, rv;
    TilePtrs<D, NT> pk, pv;
    tile_ptrs<D, NT>(pk, K, f.k_ts);
    tile_ptrs<D, NT>(pv, V, f.v_ts);
    fetch_tile_at<D, NT>(rk, pk, K, f.k_ts, 0, f.Tk);
    fetch_tile_at<D, NT>(rv, pv, V, f.v_ts, 0, f.Tk);
    stash_tile_ld<D, NT, LDK>(kbuf, rk);
    stash_tile_ld<D, NT, LDV>(vbuf, rv);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));      // landed (see attn_fwd_w32_kernel)
    if (KC < k_end) {
        fetch_tile_at<D, NT>(rk, pk, K, f.k_ts, KC, f.Tk);
        fetch_tile_at<D, NT>(rv, pv, V, f.v_ts, KC, f.Tk);
    }
    int cur = 0;
    for (int64_t kc0 = 0; kc0 < k_end; kc0 += KC, cur ^= 1) {
        __syncthreads();
        if (kc0 + KC < k_end) {
            stash_tile_ld<D, NT, LDK>(kbuf + (cur ^ 1) * KC * LDK, rk);
            stash_tile_ld<D, NT, LDV>(vbuf + (cur ^ 1) * KC * LDV, rv);
            if (kc0 + 2 * KC < k_end) {
                fetch_tile_at<D, NT>(rk, pk, K, f.k_ts, kc0 + 2 * KC, f.Tk);
                fetch_tile_at<D, NT>(rv, pv, V, f.v_ts, kc0 + 2 * KC, f.Tk);
            }
        }
        if (!wave_live || kc0 > wave_qmax) continue;
        const bf16_t* kt = kbuf + cur * KC * LDK;
        const bf16_t* vt = vbuf + cur * KC * LDV;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t kb = kc0 + t * 32;
            if (kb >= k_end || kb > wave_qmax) continue;       // (wave-uniform)
            // ---- S^T (chain 0) and dP^T (chain 1) of the tile: fragments of K and V rows alternate through one ring, three units ahead
            f32x16 sd[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sd[0][r] = 0.f; sd[1][r] = 0.f; }
            {
                const bf16_t* kr = kt + (t * 32 + l31) * LDK + hi * 8;
                const bf16_t* vr = vt + (t * 32 + l31) * LDV + hi * 8;
                constexpr int RING = 4, A = RING - 1;
                bf16x8 fr[RING];
#pragma unroll
                for (int i = 0; i < A; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(((i & 1) ? vr : kr) + (i >> 1) * 16);
#pragma unroll
                for (int i = 0; i < 2 * NKS; ++i) {          // i = 2 ks + which (0: K row x Q, 1: V row x dO)
                    if (i + A < 2 * NKS) fr[(i + A) % RING] = *reinterpret_cast<const bf16x8*>((((i + A) & 1) ? vr : kr) + ((i + A) >> 1) * 16);
                    sd[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % RING], (i & 1) ? dof[i >> 1] : qf[i >> 1], sd[i & 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- dS = p (dP - delta): this lane's 16 keys of the tile, key(r) = kb + (r & 3) + 8 (r >> 2) + 4 hi
            const bool need_mask = (kb + 32 > f.Tk) || (kb + 31 > wave_qmin);
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sd[0][r], c, -lse2));
                if (need_mask && (int)kb + (r & 3) + 8 * (r >> 2) + 4 * hi > klim) p = 0.f;
                ds[r] = p * (sd[1][r] - dl);
            }
            bf16x8 dsf[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                frag8 fpk;
                fpk.u = (u32x4){pack_bf16x2(ds[8 * cb + 0], ds[8 * cb + 1]), pack_bf16x2(ds[8 * cb + 2], ds[8 * cb + 3]),
                                pack_bf16x2(ds[8 * cb + 4], ds[8 * cb + 5]), pack_bf16x2(ds[8 * cb + 6], ds[8 * cb + 7])};
                dsf[cb] = fpk.v;
            }
            // ---- dQ^T += K^T dS^T: unit i = cb * NDB + db, K^T fragments gathered three units ahead
            {
                const bf16_t* kg = kt + (t * 32 + 4 * hi + ((lane & 15) >> 2)) * LDK + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
                auto kfrag = [&](int i) __attribute__((always_inline)) {
                    const bf16_t* p = kg + (i / NDB) * 16 * LDK + (i % NDB) * 32;
                    union { bf16x8 v; s16x4 h[2]; } fu;
                    fu.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
                    fu.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * LDK));
                    return fu.v;
                };
                constexpr int RING = 4, A = RING - 1;
                bf16x8 kf[RING];
#pragma unroll
                for (int i = 0; i < A; ++i) kf[i] = kfrag(i);
#pragma unroll
                for (int i = 0; i < 2 * NDB; ++i) {
                    if (i + A < 2 * NDB) kf[(i + A) % RING] = kfrag(i + A);
                    dq[i % NDB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % RING], dsf[i / NDB], dq[i % NDB], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (!q_valid) return;
    bf16_t* DQ = reinterpret_cast<bf16_t*>(a.dq) + b * a.dq_bs + h * a.dq_hs + qrow * a.dq_ts;
    store_grad_row_w32<NDB>(DQ, dq, f.scale, hi, a.rope_cos ? a.rope_cos + (qrow + coff) * D : nullptr, a.rope_cos ? a.rope_sin + (qrow + coff) * D : nullptr);
}

// =============================================================================================== resident variants
// Causal self-attention of the backbone at T <= 256..512: the whole K and V of one (batch, head) fit in LDS, so a
// workgroup pays the global-load latency ONCE, synchronises once, and every wave then streams through the keys with no
// barrier at all (the chunked kernels above spend most of their time parked at the two barriers per 64-key chunk:
// 32 us per layer for 3.2 GFLOP). Waves take PAIRS of 16-row query tiles (i, nt-1-i): with a causal mask every pair
// costs the same, so waves and workgroups are balanced by construction.
// dynamic LDS: K tile [Tk][D+8] | V tile [Tk][D+8]
// rows [0, rows) are copied, rows [rows, ceil32(rows)) are ZERO-filled: every 32-row slab read below stays in bounds and
// the masked (p == 0) rows contribute exact zeros instead of 0 * garbage.
__device__ __forceinline__ int64_t ceil32(int64_t v) { return (v + 31) & ~(int64_t)31; }
template <int D, int NT, int BATCH = 8>
__device__ __forceinline__ void load_rows(bf16_t* tile, const bf16_t* src, int64_t src_ts, int64_t rows) {
    constexpr int LDT = D + attn_pad(D), CPR = D / 8;
    const int64_t total = ceil32(rows) * CPR;
    for (int64_t base = 0; base < total; base += (int64_t)BATCH * NT) {
        u32x4 v[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {       // 8 independent 16-B loads in flight per thread, then the stores
            const int64_t s = base + threadIdx.x + (int64_t)i * NT;
            const int64_t r = s / CPR;
            v[i] = (u32x4){0u, 0u, 0u, 0u};
            if (s < total && r < rows) v[i] = *reinterpret_cast<const u32x4*>(src + r * src_ts + (s % CPR) * 8);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int64_t s = base + threadIdx.x + (int64_t)i * NT;
            if (s < total) *reinterpret_cast<u32x4*>(tile + (s / CPR) * LDT + (s % CPR) * 8) = v[i];
        }
    }
}

// loads in flight per thread and matrix while a resident kernel fills its LDS tiles: the whole 256-row tile in ONE round trip where the registers
// allow it (hd 128, 8 waves: 8 chunks per thread and matrix; with 4 the fill took two trips per workgroup, and a workgroup is alone on its CU)
template <int D, int NW>
struct RES_BATCH { static constexpr int value = (D >= 128 && NW >= 8) ? 8 : 4; };

// two matrices (K and V, or Q and dO) in ONE round trip: every load of both is issued before the first LDS store
template <int D, int NT, int BATCH, int PAD = attn_pad(D)>
__device__ __forceinline__ void load_rows_pair(bf16_t* tile_a, const bf16_t* src_a, int64_t ts_a, bf16_t* tile_b, const bf16_t* src_b,
                                               int64_t ts_b, int64_t rows) {
    constexpr int LDT = D + PAD, CPR = D / 8;
    const int64_t total = ceil32(rows) * CPR;
    for (int64_t base = 0; base < total; base += (int64_t)BATCH * NT) {
        u32x4 va[BATCH], vb[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int64_t s = base + threadIdx.x + (int64_t)i * NT;
            const int64_t r = s / CPR;
            va[i] = (u32x4){0u, 0u, 0u, 0u};
            vb[i] = va[i];
            if (s < total && r < rows) {
                va[i] = *reinterpret_cast<const u32x4*>(src_a + r * ts_a + (s % CPR) * 8);
                vb[i] = *reinterpret_cast<const u32x4*>(src_b + r * ts_b + (s % CPR) * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int64_t s = base + threadIdx.x + (int64_t)i * NT;
            if (s < total) {
                *reinterpret_cast<u32x4*>(tile_a + (s / CPR) * LDT + (s % CPR) * 8) = va[i];
                *reinterpret_cast<u32x4*>(tile_b + (s / CPR) * LDT + (s % CPR) * 8) = vb[i];
            }
        }
    }
}

template <int D, int NW, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void attn_fwd_res_kernel(const mtl_attn_fwd_args a) {
    constexpr int LDT = D + attn_pad_res(D), NKS = D / 32, NDT = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* ktile = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* vtile = ktile + ceil32(a.Tk) * LDT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;   // (wave index in an SGPR: tile offsets, loop bounds and the mask test become scalar)
    const int64_t b = blockIdx.z, h = blockIdx.y, hk = h / (a.Hq / a.Hkv);
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * a.q_hs;
    load_rows_pair<D, NW * 64, RES_BATCH<D, NW>::value, attn_pad_res(D)>(ktile, reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + hk * a.k_hs, a.k_ts,
                                  vtile, reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + hk * a.v_hs, a.v_ts, a.Tk);
    __syncthreads();
    const uint32_t dbase = DROP ? drop_base(a.dropout_seed, (uint32_t)(b * a.Hq + h)) : 0u;
    const uint32_t drop_thr = DROP ? drop_threshold(a.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    const float c = a.scale * LOG2E;
    const int64_t coff = a.causal_off;
    const int nt = (int)((a.Tq + 15) / 16), npairs = (nt + 1) / 2;
    // few query tiles (the prompt-row cache / last layer: Tq = the patch rows, e.g. 8 tiles for 8 waves): ONE tile per wave instead of a pair per
    // wave on half of the waves, as in the dQ kernel below
    const bool single = nt <= NW && gridDim.x == 1;
    const int pi = blockIdx.x * NW + wave;
    if (pi >= (single ? nt : npairs)) return;
    for (int half = 0; half < 2; ++half) {
        const int tile = single ? pi : (half == 0 ? pi : nt - 1 - pi);
        if (half == 1 && (single || tile == pi)) break;
        const int64_t q0 = (int64_t)tile * 16;
        int64_t qrow = q0 + l15;
        const bool q_valid = qrow < a.Tq;
        if (qrow > a.Tq - 1) qrow = a.Tq - 1;
        const int klim = (int)(qrow + coff < a.Tk - 1 ? qrow + coff : a.Tk - 1);
        bf16x8 qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * a.q_ts + ks * 32 + g * 8);
        f32x4 o[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float m_run = NEG_BIG, l_run = 0.f;
        const int64_t wave_qmin = q0 + coff;
        const int64_t wave_qmax = ((q0 + 15 < a.Tq - 1) ? q0 + 15 : a.Tq - 1) + coff;
        const int64_t k_end = (wave_qmax + 1 < a.Tk) ? wave_qmax + 1 : a.Tk;
        for (int64_t kb = 0; kb < k_end; kb += 32) {
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = ktile + (kb + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], s[t], 0, 0, 0);
            }
            float p[2][4];
            const bool need_mask = (kb + 32 > a.Tk) || (kb + 31 > wave_qmin);
            if (need_mask) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[t][r] = ((int)kb + t * 16 + g * 4 + r > klim) ? NEG_BIG : s[t][r] * c;
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[t][r] = s[t][r] * c;
            }
            float mx = fmaxf(fmaxf(fmaxf(p[0][0], p[0][1]), fmaxf(p[0][2], p[0][3])), fmaxf(fmaxf(p[1][0], p[1][1]), fmaxf(p[1][2], p[1][3])));
            mx = rows_max(mx);
            const float m_new = fmaxf(m_run, mx);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[t][r] = __builtin_amdgcn_exp2f(p[t][r] - m_new);
                    psum += p[t][r];
                }
            if (__any(m_new != m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
                m_run = m_new;
            }
            l_run += psum;
            if (DROP) {   // attn_pdrop: the normaliser keeps the undropped sum, the P.V operand carries the mask (one word per key pair)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                {
                    const uint2 w = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);   // absolute query index
                    p[t][0] = (w.x & 0xffffu) >= drop_thr ? p[t][0] * drop_scale : 0.f;
                    p[t][1] = (w.x >> 16) >= drop_thr ? p[t][1] * drop_scale : 0.f;
                    p[t][2] = (w.y & 0xffffu) >= drop_thr ? p[t][2] * drop_scale : 0.f;
                    p[t][3] = (w.y >> 16) >= drop_thr ? p[t][3] * drop_scale : 0.f;
                }
            }
            const bf16x8 pf = pack8(p[0], p[1]);
            const int ra = (int)(kb + g * 4), rb = (int)(kb + 16 + g * 4);   // rows >= Tk are zero-filled and carry p == 0
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 vt = gather_col(vtile, LDT, ra, rb, dt * 16, l15);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf, o[dt], 0, 0, 0);
            }
        }
        l_run = rows_sum(l_run);
        if (q_valid) {
            const float inv_l = 1.0f / l_run;
            bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + b * a.o_bs + h * a.o_hs + qrow * a.o_ts;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2 pk = {pack_bf16x2(o[dt][0] * inv_l, o[dt][1] * inv_l), pack_bf16x2(o[dt][2] * inv_l, o[dt][3] * inv_l)};
                *reinterpret_cast<u32x2*>(O + dt * 16 + g * 4) = pk;
            }
            if (g == 0 && a.lse) a.lse[(b * a.Hq + h) * (a.stat_stride ? a.stat_stride : a.Tq) + qrow] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;
        }
    }
}

template <int D, int NW, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_res_kernel(const mtl_attn_bwd_args a) {
    constexpr int LDT = D + attn_pad(D), NKS = D / 32, NDT = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const mtl_attn_fwd_args& f = a.f;
    bf16_t* ktile = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* vtile = ktile + ceil32(f.Tk) * LDT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;   // (wave index in an SGPR: tile offsets, loop bounds and the mask test become scalar)
    const int64_t b = blockIdx.z, h = blockIdx.y, hk = h / (f.Hq / f.Hkv);
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(f.q) + b * f.q_bs + h * f.q_hs;
    const bf16_t* O = reinterpret_cast<const bf16_t*>(f.o) + b * f.o_bs + h * f.o_hs;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_bs + h * a.do_hs;
    load_rows_pair<D, NW * 64, RES_BATCH<D, NW>::value>(ktile, reinterpret_cast<const bf16_t*>(f.k) + b * f.k_bs + hk * f.k_hs, f.k_ts,
                                  vtile, reinterpret_cast<const bf16_t*>(f.v) + b * f.v_bs + hk * f.v_hs, f.v_ts, f.Tk);
    __syncthreads();
    const uint32_t dbase = DROP ? drop_base(f.dropout_seed, (uint32_t)(b * f.Hq + h)) : 0u;
    const uint32_t drop_thr = DROP ? drop_threshold(f.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    const float c = f.scale * LOG2E;
    const int64_t coff = f.causal_off;
    const int nt = (int)((f.Tq + 15) / 16), npairs = (nt + 1) / 2;
    // few query tiles (the pruned backward: Tq = n_grad, e.g. 8 tiles for 8 waves): ONE tile per wave instead of a pair per wave on half
    // of the waves — the launch then lasts as long as its heaviest tile (Tk keys) instead of a pair (~1.6 Tk)
    const bool single = nt <= NW && gridDim.x == 1;
    const int pi = blockIdx.x * NW + wave;
    if (pi >= (single ? nt : npairs)) return;
    for (int half = 0; half < 2; ++half) {
        const int tile = single ? pi : (half == 0 ? pi : nt - 1 - pi);
        if (half == 1 && (single || tile == pi)) break;
        const int64_t q0 = (int64_t)tile * 16;
        int64_t qrow = q0 + l15;
        const bool q_valid = qrow < f.Tq;
        if (qrow > f.Tq - 1) qrow = f.Tq - 1;
        const int klim = (int)(qrow + coff < f.Tk - 1 ? qrow + coff : f.Tk - 1);
        bf16x8 qf[NKS], dof[NKS];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * f.q_ts + ks * 32 + g * 8);
            frag8 d8, o8;
            d8.v = *reinterpret_cast<const bf16x8*>(dO + qrow * a.do_ts + ks * 32 + g * 8);
            o8.v = *reinterpret_cast<const bf16x8*>(O + qrow * f.o_ts + ks * 32 + g * 8);
            dof[ks] = d8.v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dl += __uint_as_float(d8.u[e] << 16) * __uint_as_float(o8.u[e] << 16);
                dl += __uint_as_float(d8.u[e] & 0xffff0000u) * __uint_as_float(o8.u[e] & 0xffff0000u);
            }
        }
        dl = rows_sum(dl);
        const int64_t stat_idx = (b * f.Hq + h) * (f.stat_stride ? f.stat_stride : f.Tq) + qrow;
        const float lse2 = f.lse[stat_idx] * LOG2E;
        f32x4 dq[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int64_t wave_qmax = ((q0 + 15 < f.Tq - 1) ? q0 + 15 : f.Tq - 1) + coff;
        const int64_t k_end = (wave_qmax + 1 < f.Tk) ? wave_qmax + 1 : f.Tk;
        if (MTL_CONSISTENT_DELTA) {
            // delta_q = sum_s p_qs * dP_qs from the very p and dP of the main loop (see attn_bwd_dq_kernel): sum_s dS_qs = 0 to fp32
            // round-off. K and V are already resident in LDS, so the extra pass costs S and dP MFMAs only.
            float acc = 0.f;
            for (int64_t kb = 0; kb < k_end; kb += 32) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    const bf16_t* kr = ktile + (kb + t * 16 + l15) * LDT + g * 8;
                    const bf16_t* vr = vtile + (kb + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(vr + ks * 32), dof[ks], dp, 0, 0, 0);
                    }
                    uint2 dw = make_uint2(0u, 0u);      // mask words of the lane's four keys
                    if (DROP) dw = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool masked = (int)kb + t * 16 + g * 4 + r > klim;
                        const float pv = masked ? 0.f : __builtin_amdgcn_exp2f(s[r] * c - lse2);
                        float dpv = dp[r];
                        if (DROP) dpv = drop_field(dw, (uint32_t)r) >= drop_thr ? dpv * drop_scale : 0.f;
                        acc += pv * dpv;
                    }
                }
            }
            dl = rows_sum(acc);
        }
        if (g == 0 && q_valid) a.delta[stat_idx] = dl;
        for (int64_t kb = 0; kb < k_end; kb += 32) {
            float ds[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = ktile + (kb + t * 16 + l15) * LDT + g * 8;
                const bf16_t* vr = vtile + (kb + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(vr + ks * 32), dof[ks], dp, 0, 0, 0);
                }
                uint2 dw = make_uint2(0u, 0u);      // mask words of the lane's four keys
                if (DROP) dw = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool masked = (int)kb + t * 16 + g * 4 + r > klim;
                    const float pv = masked ? 0.f : __builtin_amdgcn_exp2f(s[r] * c - lse2);
                    float dpv = dp[r];
                    if (DROP) dpv = drop_field(dw, (uint32_t)r) >= drop_thr ? dpv * drop_scale : 0.f;
                    ds[t][r] = pv * (dpv - dl);
                }
            }
            const bf16x8 dsf = pack8(ds[0], ds[1]);
            const int ra = (int)(kb + g * 4), rb = (int)(kb + 16 + g * 4);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 kt = gather_col(ktile, LDT, ra, rb, dt * 16, l15);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[dt], 0, 0, 0);
            }
        }
        if (q_valid) {
            bf16_t* DQ = reinterpret_cast<bf16_t*>(a.dq) + b * a.dq_bs + h * a.dq_hs + qrow * a.dq_ts;
            store_grad_row<NDT>(DQ, dq, f.scale, g, a.rope_cos ? a.rope_cos + (qrow + coff) * D : nullptr, a.rope_cos ? a.rope_sin + (qrow + coff) * D : nullptr);
        }
    }
}

// dK/dV with every query row (Q, dO, lse, delta) of the head resident in LDS; waves take pairs of 16-key tiles.
// dynamic LDS: Q tile [Tq][D+8] | dO tile [Tq][D+8] | lse2[Tq] | delta[Tq]
template <int D, int NW, bool DROP = false>
__global__ __launch_bounds__(NW * 64, (D >= 128 && !DROP) ? 2 : 1) void attn_bwd_dkv_res_kernel(const mtl_attn_bwd_args a) {
    constexpr int LDT = D + attn_pad(D), NKS = D / 32, NDT = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const mtl_attn_fwd_args& f = a.f;
    bf16_t* qtile = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* dotile = qtile + ceil32(f.Tq) * LDT;
    float* lse_s = reinterpret_cast<float*>(dotile + ceil32(f.Tq) * LDT);
    float* delta_s = lse_s + ceil32(f.Tq);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;   // (wave index in an SGPR: tile offsets, loop bounds and the mask test become scalar)
    const int64_t b = blockIdx.z, hk = blockIdx.y;
    const uint32_t drop_thr = DROP ? drop_threshold(f.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    const int group = (int)(f.Hq / f.Hkv);
    const float c = f.scale * LOG2E;
    const int64_t coff = f.causal_off;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(f.k) + b * f.k_bs + hk * f.k_hs;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(f.v) + b * f.v_bs + hk * f.v_hs;
    const int nkt = (int)((f.Tk - a.kv_row0 + 15) / 16), npairs = (nkt + 1) / 2;
    const int pi = blockIdx.x * NW + wave;
    const bool active = pi < npairs;
    // one accumulator set: the two key tiles of the pair are processed one after the other (for GQA the query heads of
    // the group are re-staged per tile; group == 1 stages Q/dO exactly once)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int tile = half == 0 ? pi : nkt - 1 - pi;
        const bool tile_on = active && !(half == 1 && tile == pi);
        const int64_t k0 = a.kv_row0 + (int64_t)(tile_on ? tile : 0) * 16;
        const int64_t krow = k0 + l15;
        const int qhi = (int)f.Tq - 1, qlo = (int)(krow - coff);      // queries that see the lane's key
        const bool quad_ok = (k0 & 3) == 0;           // the four keys of a lane quad share their mask quad
        const int64_t krc = krow > f.Tk - 1 ? f.Tk - 1 : krow;
        bf16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(K + krc * f.k_ts + ks * 32 + g * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(V + krc * f.v_ts + ks * 32 + g * 8);
        }
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) {
            dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int hg = 0; hg < group; ++hg) {
            const uint32_t dbase_h = DROP ? drop_base(f.dropout_seed, (uint32_t)(b * f.Hq + hk * group + hg)) : 0u;
            if (half == 0 || group > 1) {
                const int64_t h = hk * group + hg;
                const int64_t stat0 = (b * f.Hq + h) * (f.stat_stride ? f.stat_stride : f.Tq);
                __syncthreads();
                load_rows_pair<D, NW * 64, RES_BATCH<D, NW>::value>(qtile, reinterpret_cast<const bf16_t*>(f.q) + b * f.q_bs + h * f.q_hs, f.q_ts,
                                              dotile, reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_bs + h * a.do_hs, a.do_ts, f.Tq);
                for (int64_t i = threadIdx.x; i < ceil32(f.Tq); i += NW * 64) {
                    lse_s[i] = i < f.Tq ? f.lse[stat0 + i] * LOG2E : 0.f;
                    delta_s[i] = i < f.Tq ? a.delta[stat0 + i] : 0.f;
                }
                __syncthreads();
            }
            if (!tile_on) continue;
            // first query that can see key k0: q + coff >= k0
            const int64_t qs = k0 > coff ? ((k0 - coff) / 32) * 32 : 0;
            for (int64_t qb = qs; qb < f.Tq; qb += 32) {
                float p[2][4], ds[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    const bf16_t* qr = qtile + (qb + t * 16 + l15) * LDT + g * 8;
                    const bf16_t* dr = dotile + (qb + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qr + ks * 32), kf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(dr + ks * 32), vf[ks], dp, 0, 0, 0);
                    }
                    uint32_t fld[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
                    if (DROP) {
                        const uint32_t a0 = (uint32_t)((int)qb + t * 16 + g * 4 + (int)coff);
                        if (quad_ok) drop_fields_shared(dbase_h, a0, (uint32_t)krow, fld);
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) fld[r] = drop_field(drop_quad(dbase_h, a0 + r, (uint32_t)krow >> 2), (uint32_t)krow);
                        }
                    }
                    // (statistics as two 16-byte reads, unconditional exponential + select: see attn_bwd_dkv_kernel)
                    const int q0 = (int)qb + t * 16 + g * 4;
                    const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + q0), dl4 = *reinterpret_cast<const f32x4*>(delta_s + q0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + r;
                        const bool masked = q > qhi || q < qlo;
                        const float e = __builtin_amdgcn_exp2f(s[r] * c - lse4[r]);
                        const float pv = masked ? 0.f : e;
                        float keep = 1.0f;
                        if (DROP) keep = fld[r] >= drop_thr ? drop_scale : 0.f;
                        p[t][r] = pv * keep;                               // feeds dV = (dropped P)^T dO
                        ds[t][r] = pv * (dp[r] * keep - dl4[r]);           // feeds dK
                    }
                }
                const bf16x8 pf = pack8(p[0], p[1]);
                const bf16x8 dsf = pack8(ds[0], ds[1]);
                const int ra = (int)(qb + g * 4), rb = (int)(qb + 16 + g * 4);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const bf16x8 dot = gather_col(dotile, LDT, ra, rb, dt * 16, l15);
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf, dv[dt], 0, 0, 0);
                    const bf16x8 qt = gather_col(qtile, LDT, ra, rb, dt * 16, l15);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        if (!tile_on || krow >= f.Tk) continue;
        bf16_t* DK = reinterpret_cast<bf16_t*>(a.dk) + b * a.dk_bs + hk * a.dk_hs + krow * a.dk_ts;
        bf16_t* DV = reinterpret_cast<bf16_t*>(a.dv) + b * a.dv_bs + hk * a.dv_hs + krow * a.dv_ts;
        store_grad_row<NDT>(DK, dk, f.scale, g, a.rope_cos ? a.rope_cos + krow * D : nullptr, a.rope_cos ? a.rope_sin + krow * D : nullptr);
        store_grad_row<NDT>(DV, dv, 1.0f, g, nullptr, nullptr);
    }
}

// =============================================================================================== resident backward, ONE launch per layer
// The pruned backward of the backbone (queries = the n_grad patch rows, dK / dV for the patch keys only: <= NW 16-row tiles of each) moves about as many
// bytes as it computes on: the dQ kernel above fills K | V and gathers Q / dO / O, the dK/dV kernel fills Q | dO and gathers K / V — every operand of
// the head crosses the memory system twice, in two launches that are each a fill round trip, a few hundred MFMAs per wave and a store. Here the head's
// five operands (K, V: Tk rows; Q, dO, O: Tq rows) and the row statistics are staged ONCE, all loads in flight together; delta = rowsum(dO . O) is
// formed from the LDS tiles by all threads; then the first half of the workgroup's waves takes one query tile each for dQ while the second half takes
// one key tile each for dK / dV — 16 waves on the CU (one workgroup fits: 130 KB of LDS), four per SIMD: the same thread-level parallelism the two
// separate launches get from co-resident workgroups (with 8 waves doing the two phases one after the other the kernel was latency-bound: 31.8 us, a
// persistent software-pipelined variant of that 33.4 us, against 18.5 + 14.4 us for the two launches). Same arithmetic per element as the two kernels (same fragment layouts, MFMA order, mask words): results are bit-identical to theirs.
// dynamic LDS: K [RK][D+8] | V [RK][D+8] | Q [RQ][D+8] | dO [RQ][D+8] | O [RQ][D+8] | lse2[RQ] | delta[RQ]   (RK = ceil32(Tk), RQ = ceil32(Tq))
template <int D, int NW, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void attn_bwd_res_merged_kernel(const mtl_attn_bwd_args a) {
    constexpr int LDT = D + attn_pad_res(D), NKS = D / 32, NDT = D / 16, CPR = D / 8, NT = NW * 64, BK = 2048 / NT, BQ = 1024 / NT, NTL = NW / 2;
    static_assert(NW % 2 == 0 && 2048 % NT == 0 && 1024 % NT == 0, "two wave groups; whole staging chunks per thread");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const mtl_attn_fwd_args& f = a.f;
    const int64_t RK = ceil32(f.Tk), RQ = ceil32(f.Tq);
    bf16_t* ktile = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* vtile = ktile + RK * LDT;
    bf16_t* qtile = vtile + RK * LDT;
    bf16_t* dotile = qtile + RQ * LDT;
    bf16_t* otile = dotile + RQ * LDT;
    float* lse_s = reinterpret_cast<float*>(otile + RQ * LDT);
    float* delta_s = lse_s + RQ;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;
    const int64_t b = blockIdx.z, h = blockIdx.y;          // (MHA only: one key / value head per query head)
    const bf16_t* Kg = reinterpret_cast<const bf16_t*>(f.k) + b * f.k_bs + h * f.k_hs;
    const bf16_t* Vg = reinterpret_cast<const bf16_t*>(f.v) + b * f.v_bs + h * f.v_hs;
    const bf16_t* Qg = reinterpret_cast<const bf16_t*>(f.q) + b * f.q_bs + h * f.q_hs;
    const bf16_t* Og = reinterpret_cast<const bf16_t*>(f.o) + b * f.o_bs + h * f.o_hs;
    const bf16_t* dOg = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_bs + h * a.do_hs;
    const int64_t stat0 = (b * f.Hq + h) * (f.stat_stride ? f.stat_stride : f.Tq);
    // ---- stage everything: every global load of a round is issued before its first LDS store (rows past the end are zero-filled)
    const int64_t totk = RK * CPR, totq = RQ * CPR;
    for (int64_t bk = 0, bq = 0; bk < totk || bq < totq; bk += (int64_t)BK * NT, bq += (int64_t)BQ * NT) {
        u32x4 va[BK], vb[BK], vq[BQ], vd[BQ], vo[BQ];
#pragma unroll
        for (int i = 0; i < BK; ++i) {
            const int64_t s = bk + threadIdx.x + (int64_t)i * NT, r = s / CPR;
            va[i] = (u32x4){0u, 0u, 0u, 0u};
            vb[i] = va[i];
            if (s < totk && r < f.Tk) {
                va[i] = *reinterpret_cast<const u32x4*>(Kg + r * f.k_ts + (s % CPR) * 8);
                vb[i] = *reinterpret_cast<const u32x4*>(Vg + r * f.v_ts + (s % CPR) * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int64_t s = bq + threadIdx.x + (int64_t)i * NT, r = s / CPR;
            vq[i] = (u32x4){0u, 0u, 0u, 0u};
            vd[i] = vq[i];
            vo[i] = vq[i];
            if (s < totq && r < f.Tq) {
                vq[i] = *reinterpret_cast<const u32x4*>(Qg + r * f.q_ts + (s % CPR) * 8);
                vd[i] = *reinterpret_cast<const u32x4*>(dOg + r * a.do_ts + (s % CPR) * 8);
                vo[i] = *reinterpret_cast<const u32x4*>(Og + r * f.o_ts + (s % CPR) * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < BK; ++i) {
            const int64_t s = bk + threadIdx.x + (int64_t)i * NT;
            if (s < totk) {
                *reinterpret_cast<u32x4*>(ktile + (s / CPR) * LDT + (s % CPR) * 8) = va[i];
                *reinterpret_cast<u32x4*>(vtile + (s / CPR) * LDT + (s % CPR) * 8) = vb[i];
            }
        }
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int64_t s = bq + threadIdx.x + (int64_t)i * NT;
            if (s < totq) {
                *reinterpret_cast<u32x4*>(qtile + (s / CPR) * LDT + (s % CPR) * 8) = vq[i];
                *reinterpret_cast<u32x4*>(dotile + (s / CPR) * LDT + (s % CPR) * 8) = vd[i];
                *reinterpret_cast<u32x4*>(otile + (s / CPR) * LDT + (s % CPR) * 8) = vo[i];
            }
        }
    }
    for (int64_t i = threadIdx.x; i < RQ; i += NT) lse_s[i] = i < f.Tq ? f.lse[stat0 + i] * LOG2E : 0.f;
    __syncthreads();
    // ---- delta[q] = sum_d dO[q, d] O[q, d] (the bf16-rounded forward output, as the dQ kernel forms it): four threads per row
    static_assert(D % 32 == 0, "a quarter row is whole 16-byte chunks");
    for (int64_t q = threadIdx.x >> 2; q < RQ; q += NT / 4) {
        const int part = threadIdx.x & 3;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < D / 32; ++c) {
            frag8 d8, o8;
            // (the columns and the summation order of the dQ kernel's lanes: thread `part` = its lane group g, chunk c = its k-step)
            d8.v = *reinterpret_cast<const bf16x8*>(dotile + q * LDT + c * 32 + part * 8);
            o8.v = *reinterpret_cast<const bf16x8*>(otile + q * LDT + c * 32 + part * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += __uint_as_float(d8.u[e] << 16) * __uint_as_float(o8.u[e] << 16);
                acc += __uint_as_float(d8.u[e] & 0xffff0000u) * __uint_as_float(o8.u[e] & 0xffff0000u);
            }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) {
            delta_s[q] = acc;
            if (q < f.Tq) a.delta[stat0 + q] = acc;
        }
    }
    __syncthreads();
    const uint32_t dbase = DROP ? drop_base(f.dropout_seed, (uint32_t)(b * f.Hq + h)) : 0u;
    const uint32_t drop_thr = DROP ? drop_threshold(f.dropout_p) : 0u;
    const float drop_scale = DROP ? drop_scale_of(drop_thr) : 1.0f;
    const float c = f.scale * LOG2E;
    const int64_t coff = f.causal_off;
    const int nt = (int)((f.Tq + 15) / 16), nkt = (int)((f.Tk - a.kv_row0 + 15) / 16);
    // ---- waves [0, NTL): dQ of query tile `wave`
    if (wave < NTL && wave < nt) {
        const int64_t q0 = (int64_t)wave * 16;
        int64_t qrow = q0 + l15;
        const bool q_valid = qrow < f.Tq;
        if (qrow > f.Tq - 1) qrow = f.Tq - 1;
        const int klim = (int)(qrow + coff < f.Tk - 1 ? qrow + coff : f.Tk - 1);
        bf16x8 qf[NKS], dof[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qtile + qrow * LDT + ks * 32 + g * 8);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dotile + qrow * LDT + ks * 32 + g * 8);
        }
        const float dl = delta_s[qrow], lse2 = lse_s[qrow];
        f32x4 dq[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int64_t wave_qmax = ((q0 + 15 < f.Tq - 1) ? q0 + 15 : f.Tq - 1) + coff;
        const int64_t k_end = (wave_qmax + 1 < f.Tk) ? wave_qmax + 1 : f.Tk;
        for (int64_t kb = 0; kb < k_end; kb += 32) {
            float ds[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = ktile + (kb + t * 16 + l15) * LDT + g * 8;
                const bf16_t* vr = vtile + (kb + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + ks * 32), qf[ks], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(vr + ks * 32), dof[ks], dp, 0, 0, 0);
                }
                uint2 dw = make_uint2(0u, 0u);
                if (DROP) dw = drop_quad(dbase, (uint32_t)(qrow + coff), (uint32_t)(kb + t * 16 + g * 4) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool masked = (int)kb + t * 16 + g * 4 + r > klim;
                    const float pv = masked ? 0.f : __builtin_amdgcn_exp2f(sv[r] * c - lse2);
                    float dpv = dp[r];
                    if (DROP) dpv = drop_field(dw, (uint32_t)r) >= drop_thr ? dpv * drop_scale : 0.f;
                    ds[t][r] = pv * (dpv - dl);
                }
            }
            const bf16x8 dsf = pack8(ds[0], ds[1]);
            const int ra = (int)(kb + g * 4), rb = (int)(kb + 16 + g * 4);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 kt = gather_col(ktile, LDT, ra, rb, dt * 16, l15);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[dt], 0, 0, 0);
            }
        }
        if (q_valid) {
            bf16_t* DQ = reinterpret_cast<bf16_t*>(a.dq) + b * a.dq_bs + h * a.dq_hs + qrow * a.dq_ts;
            store_grad_row<NDT>(DQ, dq, f.scale, g, a.rope_cos ? a.rope_cos + (qrow + coff) * D : nullptr, a.rope_cos ? a.rope_sin + (qrow + coff) * D : nullptr);
        }
    }
    // ---- waves [NTL, NW): dK / dV of key tile `wave - NTL` (keys kv_row0 + 16 (wave - NTL) ...), concurrently with the dQ waves
    if (wave >= NTL && wave - NTL < nkt) {
        const int64_t k0 = a.kv_row0 + (int64_t)(wave - NTL) * 16;
        const int64_t krow = k0 + l15;
        const int qhi = (int)f.Tq - 1, qlo = (int)(krow - coff);      // queries that see the lane's key
        const bool quad_ok = (k0 & 3) == 0;
        const int64_t krc = krow > f.Tk - 1 ? f.Tk - 1 : krow;
        bf16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(ktile + krc * LDT + ks * 32 + g * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vtile + krc * LDT + ks * 32 + g * 8);
        }
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) {
            dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int64_t qs = k0 > coff ? ((k0 - coff) / 32) * 32 : 0;
        for (int64_t qb = qs; qb < f.Tq; qb += 32) {
            float p[2][4], ds[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                const bf16_t* qr = qtile + (qb + t * 16 + l15) * LDT + g * 8;
                const bf16_t* dr = dotile + (qb + t * 16 + l15) * LDT + g * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qr + ks * 32), kf[ks], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(dr + ks * 32), vf[ks], dp, 0, 0, 0);
                }
                uint32_t fld[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
                if (DROP) {
                    const uint32_t a0 = (uint32_t)((int)qb + t * 16 + g * 4 + (int)coff);
                    if (quad_ok) drop_fields_shared(dbase, a0, (uint32_t)krow, fld);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) fld[r] = drop_field(drop_quad(dbase, a0 + r, (uint32_t)krow >> 2), (uint32_t)krow);
                    }
                }
                const int q0 = (int)qb + t * 16 + g * 4;
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + q0), dl4 = *reinterpret_cast<const f32x4*>(delta_s + q0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + r;
                    const bool masked = q > qhi || q < qlo;
                    const float e = __builtin_amdgcn_exp2f(sv[r] * c - lse4[r]);
                    const float pv = masked ? 0.f : e;
                    float keep = 1.0f;
                    if (DROP) keep = fld[r] >= drop_thr ? drop_scale : 0.f;
                    p[t][r] = pv * keep;
                    ds[t][r] = pv * (dp[r] * keep - dl4[r]);
                }
            }
            const bf16x8 pf = pack8(p[0], p[1]);
            const bf16x8 dsf = pack8(ds[0], ds[1]);
            const int ra = (int)(qb + g * 4), rb = (int)(qb + 16 + g * 4);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 dot = gather_col(dotile, LDT, ra, rb, dt * 16, l15);
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf, dv[dt], 0, 0, 0);
                const bf16x8 qt = gather_col(qtile, LDT, ra, rb, dt * 16, l15);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dk[dt], 0, 0, 0);
            }
        }
        if (krow < f.Tk) {
            bf16_t* DK = reinterpret_cast<bf16_t*>(a.dk) + b * a.dk_bs + h * a.dk_hs + krow * a.dk_ts;
            bf16_t* DV = reinterpret_cast<bf16_t*>(a.dv) + b * a.dv_bs + h * a.dv_hs + krow * a.dv_ts;
            store_grad_row<NDT>(DK, dk, f.scale, g, a.rope_cos ? a.rope_cos + krow * D : nullptr, a.rope_cos ? a.rope_sin + krow * D : nullptr);
            store_grad_row<NDT>(DV, dv, 1.0f, g, nullptr, nullptr);
        }
    }
}

// fp32 partial slabs [splits][2][Tk][Hkv][D] -> summed bf16 dk / dv (strided)
__global__ void dkv_convert_kernel(const mtl_attn_bwd_args a, const int splits) {
    const mtl_attn_fwd_args& f = a.f;
    const int64_t n = f.Tk * f.Hkv * f.D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = i % f.D, hk = (i / f.D) % f.Hkv, t = i / (f.D * f.Hkv);
        float sk = 0.f, sv = 0.f;
        for (int s = 0; s < splits; ++s) {
            sk += a.dkv_ws[(int64_t)s * 2 * n + i];
            sv += a.dkv_ws[(int64_t)s * 2 * n + n + i];
        }
        reinterpret_cast<bf16_t*>(a.dk)[hk * a.dk_hs + t * a.dk_ts + d] = f32_to_bf16(sk);
        reinterpret_cast<bf16_t*>(a.dv)[hk * a.dv_hs + t * a.dv_ts + d] = f32_to_bf16(sv);
    }
}

int check_fwd(const mtl_attn_fwd_args& f) {
    if (!f.q || !f.k || !f.v || !f.o) return MTL_ERR_ARG;
    if (f.B <= 0 || f.Hq <= 0 || f.Hkv <= 0 || f.Tq <= 0 || f.Tk <= 0 || f.Hq % f.Hkv != 0) return MTL_ERR_ARG;
    if (f.D != 32 && f.D != 64 && f.D != 128) return MTL_ERR_UNSUPPORTED;
    // a query row whose FIRST key chunk is entirely masked would average masked keys into its output (the running maximum then equals the mask value):
    // with causal_off >= 0 every query row sees key 0, so every row's first chunk holds a visible key
    if (f.causal && f.causal_off < 0) return MTL_ERR_ARG;
    const int64_t st[] = {f.q_bs, f.q_ts, f.q_hs, f.k_bs, f.k_ts, f.k_hs, f.v_bs, f.v_ts, f.v_hs, f.o_bs, f.o_ts, f.o_hs};
    for (int64_t s : st) if (s % 8 != 0) return MTL_ERR_ALIGN;
    if (((uintptr_t)f.q % 16) || ((uintptr_t)f.k % 16) || ((uintptr_t)f.v % 16) || ((uintptr_t)f.o % 16)) return MTL_ERR_ALIGN;
    return MTL_OK;
}

}  // namespace

namespace {

// resident-K/V path: causal, per-sample K/V, >= 4 rows, and both tiles fit the 160 KiB LDS
constexpr size_t kLdsBudget = 156 * 1024;
// (per-call knobs: mtl_attn_fwd_args.tune bit 0 = chunked kernels only, bit 1 = resident backward as two launches; the switches
//  below are constants of the product build; -DMTL_DIAG builds read them from the environment once)
const int g_attn_wide = mtl_env_int("MTL_ATTN_WIDE", 1);   // A/B knob: 0 = 64-row workgroups for long sequences too
// rows from which the 128-row workgroups are used. Forward / dQ from 256 (PSM shape, Tq = 256 of T = 384: 131 -> 97 us, 184 -> 150 us); the
// dK/dV kernel gains nothing there (190 -> 194 us) and switches at 512.
const int g_attn_wide_x = mtl_env_int("MTL_ATTN_WIDE_X", 1);   // A/B knob: 128-row workgroups for the non-causal hd-128 (reprogramming) attention
const int g_attn_wide_min = mtl_env_int("MTL_ATTN_WIDE_MIN", 256);
const int g_attn_xmap = mtl_env_int("MTL_ATTN_XMAP", 1);   // A/B knob: 0 = plain (row block, head, batch) grids for the long-sequence kernels
const int g_attn_w32_nw = mtl_env_int("MTL_ATTN_W32_NW", 0);   // A/B knob: 4 / 8 waves per workgroup of the 32-row kernels (0 = automatic)
const int g_attn_w32 = mtl_env_int("MTL_ATTN_W32", 1);   // A/B knob: 0 = the 16-rows-per-wave kernels for long causal sequences
// hd-128 causal self-attention without dropout (the Llama stacks) where K / V WOULD fit the LDS: the 32-rows-per-wave forward and the 8-wave chunked dQ
// kernel beat their resident twins from 128 query rows on, and so does the chunked dK/dV kernel under GQA or with more than 128 key rows (round 5,
// tools/diag/attn_matrix.sh: Llama-2 cached shape forward 57.5 -> 42.0 us, dQ 66.8 -> 53.2; Llama-3 GQA dK/dV 52.9 -> 32.6). 0 = the resident kernels.
const int g_attn_d128 = mtl_env_int("MTL_ATTN_D128", 1);
const int g_attn_merged = mtl_env_int("MTL_ATTN_MERGED", 1);   // A/B knob: 0 = the resident backward as two launches (dQ, then dK / dV)

template <typename KernelT>
void set_lds(KernelT k, size_t bytes) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }

}  // namespace


namespace {

size_t pad32(int64_t v) { return (size_t)((v + 31) & ~(int64_t)31); }
// `pad`: the LDS row padding of the kernel that would run — attn_pad_res for the resident forward / the one-launch backward, attn_pad for the two-launch
// resident dQ and dK/dV kernels (each kernel gets its own fit test: with one test on the wider padding, hd-64 causal shapes with T in (480, 512] fell
// to the chunked backward although the two-launch resident kernels fit — ADVICE r05)
bool resident_ok(const mtl_attn_fwd_args& f, int64_t rows, int pad) {
    return !(f.tune & 1) && f.causal && f.k_bs != 0 && f.dropout_p < 1.f && (f.D == 64 || f.D == 128) &&
           2 * pad32(rows) * (f.D + pad) * 2 + 2 * pad32(rows) * 4 <= kLdsBudget;
}

}  // namespace

extern "C" int mtl_attention_fwd(const mtl_attn_fwd_args* a, void* stream) {
    if (!a) return MTL_ERR_ARG;
#ifdef MTL_DIAG_ATTN_NODROP      // diagnostic builds only: the causal self-attention without its dropout (what the mask costs in-step; WRONG results)
    mtl_attn_fwd_args nodrop = *a;
    if (nodrop.causal) nodrop.dropout_p = 0.f;
    a = &nodrop;
#endif
    const int rc = check_fwd(*a);
    if (rc != MTL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // algorithmic FLOPs, full-rectangle convention: S = Q K^T and O = P V, 2 * Tq * Tk * D each per head
    const double fl_fwd = 4.0 * (double)a->B * a->Hq * a->Tq * a->Tk * a->D;
    (void)fl_fwd;
    const bool d128_fwd = g_attn_d128 == 1 && a->causal && a->D == 128 && a->dropout_p == 0.f && !a->o_f32 && a->Tq >= 128 && a->k_bs != 0 && g_attn_wide == 1 && g_attn_w32 == 1;
    if (resident_ok(*a, a->Tk, attn_pad_res((int)a->D)) && !d128_fwd) {
        const size_t lds = 2 * pad32(a->Tk) * (a->D + attn_pad_res((int)a->D)) * 2;
        const int npairs = (int)(((a->Tq + 15) / 16 + 1) / 2);
        if (a->D == 64 && a->dropout_p > 0.f) {
            static std::once_flag once; std::call_once(once, [&] { set_lds(attn_fwd_res_kernel<64, 8, true>, kLdsBudget); });
            MTL_LAUNCH("attn_fwd_res_kernel<64, 8, true>", fl_fwd, 0, (attn_fwd_res_kernel<64, 8, true>), dim3((unsigned)((npairs + 7) / 8), (unsigned)a->Hq, (unsigned)a->B), dim3(512), lds, st, *a);
        } else if (a->D == 128 && a->dropout_p > 0.f) {
            static std::once_flag once; std::call_once(once, [&] { set_lds(attn_fwd_res_kernel<128, 8, true>, kLdsBudget); });
            MTL_LAUNCH("attn_fwd_res_kernel<128, 8, true>", fl_fwd, 0, (attn_fwd_res_kernel<128, 8, true>), dim3((unsigned)((npairs + 7) / 8), (unsigned)a->Hq, (unsigned)a->B), dim3(512), lds, st, *a);
        } else if (a->D == 64) {
            static std::once_flag once; std::call_once(once, [&] { set_lds(attn_fwd_res_kernel<64, 8>, kLdsBudget); });
            MTL_LAUNCH("attn_fwd_res_kernel<64, 8, false>", fl_fwd, 0, (attn_fwd_res_kernel<64, 8>), dim3((unsigned)((npairs + 7) / 8), (unsigned)a->Hq, (unsigned)a->B), dim3(512), lds, st, *a);
        } else {
            static std::once_flag once; std::call_once(once, [&] { set_lds(attn_fwd_res_kernel<128, 8>, kLdsBudget); });
            MTL_LAUNCH("attn_fwd_res_kernel<128, 8, false>", fl_fwd, 0, (attn_fwd_res_kernel<128, 8>), dim3((unsigned)((npairs + 7) / 8), (unsigned)a->Hq, (unsigned)a->B), dim3(512), lds, st, *a);
        }
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
    const dim3 grid((unsigned)((a->Tq + 63) / 64), (unsigned)a->Hq, (unsigned)a->B), block(256);
    const bool drop = a->dropout_p > 0.f;
    if (drop && a->dropout_p >= 1.f) return MTL_ERR_UNSUPPORTED;
    // the reprogramming cross-attention (R:models/medtsllm.py ReprogrammingLayer: not causal, 1000 batch-shared prototypes as keys, hd 128): every
    // workgroup streams all keys through its LDS tiles, so 128-row workgroups halve the K / V bytes that leave the L2 (forward 50.6 -> 43.1 us
    // in-step on the metric workload; MTL_ATTN_WIDE_X=0 switches back)
    if (!a->causal && a->D == 128 && a->Tq >= 128 && g_attn_wide == 1 && g_attn_wide_x == 1) {
        const dim3 grid8((unsigned)((a->Tq + 127) / 128), (unsigned)a->Hq, (unsigned)a->B), block8(512);
        if (drop) MTL_LAUNCH("attn_fwd_kernel<128, false, true, 8>", fl_fwd, 0, (attn_fwd_kernel<128, false, true, 8>), grid8, block8, 0, st, *a);
        else MTL_LAUNCH("attn_fwd_kernel<128, false, false, 8>", fl_fwd, 0, (attn_fwd_kernel<128, false, false, 8>), grid8, block8, 0, st, *a);
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
    if (a->causal && !drop && !a->o_f32 && (a->Tq >= g_attn_wide_min || d128_fwd) && (a->D == 64 || a->D == 128) && g_attn_wide == 1 && g_attn_w32 == 1) {
        // long sequences: 32 query rows per wave on 32x32x16 MFMAs, double-buffered 64-key chunks, 4 waves = 128 queries per workgroup (8 waves measured
        // slower: profiles/r04_attn_longT_experiments.txt). The kernel has no fp32 output copy: a caller asking for o_f32 gets the 16-row kernel below.
        const int nww = (g_attn_w32_nw == 4 || g_attn_w32_nw == 8) ? g_attn_w32_nw : 4;
        const int64_t nx = (a->Tq + nww * 32 - 1) / (nww * 32);
        const dim3 gridw(g_attn_xmap ? attn_xmap_grid(nx, a->Hq, a->B) : (unsigned)nx, g_attn_xmap ? 1u : (unsigned)a->Hq, g_attn_xmap ? 1u : (unsigned)a->B), blockw(nww * 64);
        const size_t lds = (size_t)2 * KC * ((a->D + 8) + (a->D + 32)) * 2;
        static std::once_flag once;
        std::call_once(once, [&] {
            set_lds(attn_fwd_w32_kernel<64, true, 4, false>, kLdsBudget); set_lds(attn_fwd_w32_kernel<128, true, 4, false>, kLdsBudget);
            set_lds(attn_fwd_w32_kernel<64, true, 4, true>, kLdsBudget); set_lds(attn_fwd_w32_kernel<128, true, 4, true>, kLdsBudget);
            set_lds(attn_fwd_w32_kernel<64, true, 8, false>, kLdsBudget); set_lds(attn_fwd_w32_kernel<128, true, 8, false>, kLdsBudget);
            set_lds(attn_fwd_w32_kernel<64, true, 8, true>, kLdsBudget); set_lds(attn_fwd_w32_kernel<128, true, 8, true>, kLdsBudget);
        });
#define MTL_W32(DD, NN, XX) MTL_LAUNCH("attn_fwd_w32_kernel<" #DD ", true, " #NN ">", fl_fwd, 0, (attn_fwd_w32_kernel<DD, true, NN, XX>), gridw, blockw, lds, st, *a)
        if (a->D == 64) {
            if (nww == 8) { if (g_attn_xmap) MTL_W32(64, 8, true); else MTL_W32(64, 8, false); }
            else { if (g_attn_xmap) MTL_W32(64, 4, true); else MTL_W32(64, 4, false); }
        } else {
            if (nww == 8) { if (g_attn_xmap) MTL_W32(128, 8, true); else MTL_W32(128, 8, false); }
            else { if (g_attn_xmap) MTL_W32(128, 4, true); else MTL_W32(128, 4, false); }
        }
#undef MTL_W32
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
    if (a->causal && !drop && a->Tq >= g_attn_wide_min && (a->D == 64 || a->D == 128) && g_attn_wide == 1) {      // long sequences: 128 queries per workgroup
        const dim3 grid8((unsigned)((a->Tq + 127) / 128), (unsigned)a->Hq, (unsigned)a->B), block8(512);
        if (a->D == 64) MTL_LAUNCH("attn_fwd_kernel<64, true, false, 8>", fl_fwd, 0, (attn_fwd_kernel<64, true, false, 8>), grid8, block8, 0, st, *a);
        else MTL_LAUNCH("attn_fwd_kernel<128, true, false, 8>", fl_fwd, 0, (attn_fwd_kernel<128, true, false, 8>), grid8, block8, 0, st, *a);
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
#define MTL_FWD(DD)                                                                                                                        \
    if (a->causal && drop) MTL_LAUNCH("attn_fwd_kernel<" #DD ", true, true>", fl_fwd, 0, (attn_fwd_kernel<DD, true, true>), grid, block, 0, st, *a);   \
    else if (a->causal) MTL_LAUNCH("attn_fwd_kernel<" #DD ", true, false>", fl_fwd, 0, (attn_fwd_kernel<DD, true, false>), grid, block, 0, st, *a);    \
    else if (drop) MTL_LAUNCH("attn_fwd_kernel<" #DD ", false, true>", fl_fwd, 0, (attn_fwd_kernel<DD, false, true>), grid, block, 0, st, *a);        \
    else MTL_LAUNCH("attn_fwd_kernel<" #DD ", false, false>", fl_fwd, 0, (attn_fwd_kernel<DD, false, false>), grid, block, 0, st, *a)
    if (a->D == 32) { MTL_FWD(32); } else if (a->D == 64) { MTL_FWD(64); } else { MTL_FWD(128); }
#undef MTL_FWD
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_attention_bwd(const mtl_attn_bwd_args* a, void* stream) {
    if (!a) return MTL_ERR_ARG;
#ifdef MTL_DIAG_ATTN_NODROP
    mtl_attn_bwd_args nodrop = *a;
    if (nodrop.f.causal) nodrop.f.dropout_p = 0.f;
    a = &nodrop;
#endif
    const mtl_attn_fwd_args& f = a->f;
    const int rc = check_fwd(f);
    if (rc != MTL_OK) return rc;
    if (!a->dout || !a->dq || !a->dk || !a->dv || !a->delta || !f.lse) return MTL_ERR_ARG;
    if ((f.k_bs == 0) != (f.v_bs == 0)) return MTL_ERR_ARG;
    if ((a->rope_cos != nullptr) != (a->rope_sin != nullptr) || (a->rope_cos && f.k_bs == 0)) return MTL_ERR_ARG;
    const int64_t sts[] = {a->do_bs, a->do_ts, a->do_hs, a->dq_bs, a->dq_ts, a->dq_hs, a->dk_bs, a->dk_ts, a->dk_hs, a->dv_bs, a->dv_ts, a->dv_hs};
    for (int64_t s : sts) if (s % 4 != 0) return MTL_ERR_ALIGN;
    if (a->do_ts % 8 != 0 || a->do_hs % 8 != 0 || a->do_bs % 8 != 0 || ((uintptr_t)a->dout % 16)) return MTL_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    // algorithmic FLOPs of the backward = 2 x forward (dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q; the recomputation of S is
    // not counted), split evenly over the dQ kernel (dP, dQ) and the dK/dV kernel (dV, dK)
    const double fl_half = 4.0 * (double)f.B * f.Hq * f.Tq * f.Tk * f.D;
    (void)fl_half;
    if (resident_ok(f, f.Tk, attn_pad((int)f.D)) && resident_ok(f, f.Tq, attn_pad((int)f.D))) {
        if (a->kv_row0 < 0 || a->kv_row0 >= f.Tk) return MTL_ERR_ARG;
        // few tiles on both sides (the backbone's pruned backward: n_grad query rows, dK / dV for the patch keys): ONE launch stages the head once
        const size_t lds_m = (2 * pad32(f.Tk) + 3 * pad32(f.Tq)) * (f.D + attn_pad_res((int)f.D)) * 2 + 2 * pad32(f.Tq) * 4;
        if (g_attn_merged == 1 && !(f.tune & 2) && f.D == 64 && f.Hq == f.Hkv && (f.Tq + 15) / 16 <= 8 && (f.Tk - a->kv_row0 + 15) / 16 <= 8 && lds_m <= kLdsBudget) {
            static std::once_flag once;
            std::call_once(once, [&] { set_lds(attn_bwd_res_merged_kernel<64, 16, true>, kLdsBudget); set_lds(attn_bwd_res_merged_kernel<64, 16, false>, kLdsBudget); });
            const dim3 gm(1, (unsigned)f.Hq, (unsigned)f.B);
            if (f.dropout_p > 0.f) MTL_LAUNCH("attn_bwd_res_merged_kernel<64, 16, true>", 2.0 * fl_half, 0, (attn_bwd_res_merged_kernel<64, 16, true>), gm, dim3(1024), lds_m, st, *a);
            else MTL_LAUNCH("attn_bwd_res_merged_kernel<64, 16, false>", 2.0 * fl_half, 0, (attn_bwd_res_merged_kernel<64, 16, false>), gm, dim3(1024), lds_m, st, *a);
            MTL_CHECK_LAUNCH();
            return MTL_OK;
        }
        const size_t lds_q = 2 * pad32(f.Tk) * (f.D + attn_pad((int)f.D)) * 2;
        const size_t lds_k = 2 * pad32(f.Tq) * (f.D + attn_pad((int)f.D)) * 2 + 2 * pad32(f.Tq) * 4;
        const int npq = (int)(((f.Tq + 15) / 16 + 1) / 2), npk = (int)(((f.Tk - a->kv_row0 + 15) / 16 + 1) / 2);
        if (f.dropout_p > 0.f && f.D == 64) {
            static std::once_flag once;
            std::call_once(once, [&] { set_lds(attn_bwd_dq_res_kernel<64, 8, true>, kLdsBudget); set_lds(attn_bwd_dkv_res_kernel<64, 4, true>, kLdsBudget); });
            MTL_LAUNCH("attn_bwd_dq_res_kernel<64, 8, true>", fl_half, 0, (attn_bwd_dq_res_kernel<64, 8, true>), dim3((unsigned)((npq + 7) / 8), (unsigned)f.Hq, (unsigned)f.B), dim3(512), lds_q, st, *a);
            MTL_LAUNCH("attn_bwd_dkv_res_kernel<64, 4, true>", fl_half, 0, (attn_bwd_dkv_res_kernel<64, 4, true>), dim3((unsigned)((npk + 3) / 4), (unsigned)f.Hkv, (unsigned)f.B), dim3(256), lds_k, st, *a);
        } else if (f.dropout_p > 0.f) {
            static std::once_flag once;
            std::call_once(once, [&] { set_lds(attn_bwd_dq_res_kernel<128, 8, true>, kLdsBudget); set_lds(attn_bwd_dkv_res_kernel<128, 4, true>, kLdsBudget); });
            MTL_LAUNCH("attn_bwd_dq_res_kernel<128, 8, true>", fl_half, 0, (attn_bwd_dq_res_kernel<128, 8, true>), dim3((unsigned)((npq + 7) / 8), (unsigned)f.Hq, (unsigned)f.B), dim3(512), lds_q, st, *a);
            MTL_LAUNCH("attn_bwd_dkv_res_kernel<128, 4, true>", fl_half, 0, (attn_bwd_dkv_res_kernel<128, 4, true>), dim3((unsigned)((npk + 3) / 4), (unsigned)f.Hkv, (unsigned)f.B), dim3(256), lds_k, st, *a);
        } else if (f.D == 64) {
            static std::once_flag once;
            std::call_once(once, [&] { set_lds(attn_bwd_dq_res_kernel<64, 8>, kLdsBudget); set_lds(attn_bwd_dkv_res_kernel<64, 4>, kLdsBudget); });
            MTL_LAUNCH("attn_bwd_dq_res_kernel<64, 8, false>", fl_half, 0, (attn_bwd_dq_res_kernel<64, 8>), dim3((unsigned)((npq + 7) / 8), (unsigned)f.Hq, (unsigned)f.B), dim3(512), lds_q, st, *a);
            MTL_LAUNCH("attn_bwd_dkv_res_kernel<64, 4, false>", fl_half, 0, (attn_bwd_dkv_res_kernel<64, 4>), dim3((unsigned)((npk + 3) / 4), (unsigned)f.Hkv, (unsigned)f.B), dim3(256), lds_k, st, *a);
        } else {
            static std::once_flag once;
            std::call_once(once, [&] { set_lds(attn_bwd_dq_res_kernel<128, 8>, kLdsBudget); set_lds(attn_bwd_dkv_res_kernel<128, 4>, kLdsBudget); });
            // hd 128 without dropout (Llama): per kernel, the faster of the resident and the chunked form (g_attn_d128 above). dQ: the 8-wave chunked
            // kernel from 128 query rows on. dK/dV: resident only for MHA with at most 128 key rows to differentiate (the pruned / cached backward);
            // GQA (a K/V head serves 4 query heads: the resident kernel's one workgroup per KV head leaves CUs idle) and the full backward go chunked.
            // Both dQ kernels write delta for whichever dK/dV kernel follows.
            const bool rules = g_attn_d128 == 1 && f.Tq >= 128 && g_attn_wide == 1 && g_attn_xmap == 1 && f.k_bs != 0;
            const int64_t nkeys = f.Tk - a->kv_row0;
            const bool dq_chunked = rules, dkv_chunked = rules && (f.Hq != f.Hkv || nkeys > 128);
            if (dq_chunked) MTL_LAUNCH("attn_bwd_dq_kernel<128, true, false, 8>", fl_half, 0, (attn_bwd_dq_kernel<128, true, false, 8, true>), dim3(attn_xmap_grid((f.Tq + 127) / 128, f.Hq, f.B)), dim3(512), 0, st, *a);
            else MTL_LAUNCH("attn_bwd_dq_res_kernel<128, 8, false>", fl_half, 0, (attn_bwd_dq_res_kernel<128, 8>), dim3((unsigned)((npq + 7) / 8), (unsigned)f.Hq, (unsigned)f.B), dim3(512), lds_q, st, *a);
            if (dkv_chunked && nkeys >= 512) MTL_LAUNCH("attn_bwd_dkv_kernel<128, true, false, 8>", fl_half, 0, (attn_bwd_dkv_kernel<128, true, false, 8, true>), dim3(attn_xmap_grid((nkeys + 127) / 128, f.Hkv, f.B)), dim3(512), 0, st, *a);
            else if (dkv_chunked) MTL_LAUNCH("attn_bwd_dkv_kernel<128, true, false>", fl_half, 0, (attn_bwd_dkv_kernel<128, true, false, 4, true>), dim3(attn_xmap_grid((nkeys + 63) / 64, f.Hkv, f.B)), dim3(256), 0, st, *a);
            else MTL_LAUNCH("attn_bwd_dkv_res_kernel<128, 4, false>", fl_half, 0, (attn_bwd_dkv_res_kernel<128, 4>), dim3((unsigned)((npk + 3) / 4), (unsigned)f.Hkv, (unsigned)f.B), dim3(256), lds_k, st, *a);
        }
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
    const dim3 block(256);
    const dim3 gq((unsigned)((f.Tq + 63) / 64), (unsigned)f.Hq, (unsigned)f.B);
    int splits = 1;
    if (f.k_bs == 0 && a->dkv_ws && a->kv_splits > 1) {
        splits = (int)(a->kv_splits < f.B ? a->kv_splits : f.B);
        // every chunk must be non-empty so that every slab is fully written
        const int64_t chunk = (f.B + splits - 1) / splits;
        splits = (int)((f.B + chunk - 1) / chunk);
    }
    if (a->kv_row0 < 0 || a->kv_row0 >= f.Tk) return MTL_ERR_ARG;
    const dim3 gk((unsigned)((f.Tk - a->kv_row0 + 63) / 64), (unsigned)f.Hkv, (unsigned)(f.k_bs == 0 ? splits : f.B));
    const bool drop = f.dropout_p > 0.f;
    if (drop && f.dropout_p >= 1.f) return MTL_ERR_UNSUPPORTED;
    if (f.causal && !drop && f.k_bs != 0 && f.Tq >= g_attn_wide_min && (f.D == 64 || f.D == 128) && g_attn_wide == 1) {   // long sequences: 128 rows per workgroup
        const dim3 gq8((unsigned)((f.Tq + 127) / 128), (unsigned)f.Hq, (unsigned)f.B), gk8((unsigned)((f.Tk - a->kv_row0 + 127) / 128), (unsigned)f.Hkv, (unsigned)f.B), block8(512);
        const bool wide_kv = f.Tk - a->kv_row0 >= 512;
        const dim3 gk4((unsigned)((f.Tk - a->kv_row0 + 63) / 64), (unsigned)f.Hkv, (unsigned)f.B);
        // dQ on the 32-rows-per-wave kernel (4 waves = 128 queries per workgroup) from 512 query rows on: -7 % at T = 1664 (843 -> 782 us, hd 128), -4 % at
        // T = 3328; at the PSM shape (Tq = 256 of T = 384) the 16-row kernel is faster (125 vs 132 us) and stays
        const bool dq_w32 = g_attn_w32 == 1 && g_attn_xmap && f.Tq >= 512;
        if (dq_w32) {
            const size_t lds = (size_t)2 * KC * ((f.D + 16) + (f.D + 8)) * 2;
            static std::once_flag once;
            std::call_once(once, [&] { set_lds(attn_bwd_dq_w32_kernel<64, 4, true>, kLdsBudget); set_lds(attn_bwd_dq_w32_kernel<128, 4, true>, kLdsBudget); });
            const dim3 xq(attn_xmap_grid((f.Tq + 127) / 128, f.Hq, f.B));
            if (f.D == 64) MTL_LAUNCH("attn_bwd_dq_w32_kernel<64, 4>", fl_half, 0, (attn_bwd_dq_w32_kernel<64, 4, true>), xq, dim3(256), lds, st, *a);
            else MTL_LAUNCH("attn_bwd_dq_w32_kernel<128, 4>", fl_half, 0, (attn_bwd_dq_w32_kernel<128, 4, true>), xq, dim3(256), lds, st, *a);
        }
        if (g_attn_xmap) {      // XCD-aware 1-D grids (attn_block): a head's row blocks share one L2
            const dim3 xq(attn_xmap_grid((f.Tq + 127) / 128, f.Hq, f.B)), xk8(attn_xmap_grid((f.Tk - a->kv_row0 + 127) / 128, f.Hkv, f.B)),
                       xk4(attn_xmap_grid((f.Tk - a->kv_row0 + 63) / 64, f.Hkv, f.B));
            if (f.D == 64) {
                if (!dq_w32) MTL_LAUNCH("attn_bwd_dq_kernel<64, true, false, 8>", fl_half, 0, (attn_bwd_dq_kernel<64, true, false, 8, true>), xq, block8, 0, st, *a);
                if (wide_kv) MTL_LAUNCH("attn_bwd_dkv_kernel<64, true, false, 8>", fl_half, 0, (attn_bwd_dkv_kernel<64, true, false, 8, true>), xk8, block8, 0, st, *a);
                else MTL_LAUNCH("attn_bwd_dkv_kernel<64, true, false>", fl_half, 0, (attn_bwd_dkv_kernel<64, true, false, 4, true>), xk4, dim3(256), 0, st, *a);
            } else {
                if (!dq_w32) MTL_LAUNCH("attn_bwd_dq_kernel<128, true, false, 8>", fl_half, 0, (attn_bwd_dq_kernel<128, true, false, 8, true>), xq, block8, 0, st, *a);
                if (wide_kv) MTL_LAUNCH("attn_bwd_dkv_kernel<128, true, false, 8>", fl_half, 0, (attn_bwd_dkv_kernel<128, true, false, 8, true>), xk8, block8, 0, st, *a);
                else MTL_LAUNCH("attn_bwd_dkv_kernel<128, true, false>", fl_half, 0, (attn_bwd_dkv_kernel<128, true, false, 4, true>), xk4, dim3(256), 0, st, *a);
            }
            MTL_CHECK_LAUNCH();
            return MTL_OK;
        }
        if (f.D == 64) {
            MTL_LAUNCH("attn_bwd_dq_kernel<64, true, false, 8>", fl_half, 0, (attn_bwd_dq_kernel<64, true, false, 8>), gq8, block8, 0, st, *a);
            if (wide_kv) MTL_LAUNCH("attn_bwd_dkv_kernel<64, true, false, 8>", fl_half, 0, (attn_bwd_dkv_kernel<64, true, false, 8>), gk8, block8, 0, st, *a);
            else MTL_LAUNCH("attn_bwd_dkv_kernel<64, true, false>", fl_half, 0, (attn_bwd_dkv_kernel<64, true, false>), gk4, dim3(256), 0, st, *a);
        } else {
            MTL_LAUNCH("attn_bwd_dq_kernel<128, true, false, 8>", fl_half, 0, (attn_bwd_dq_kernel<128, true, false, 8>), gq8, block8, 0, st, *a);
            if (wide_kv) MTL_LAUNCH("attn_bwd_dkv_kernel<128, true, false, 8>", fl_half, 0, (attn_bwd_dkv_kernel<128, true, false, 8>), gk8, block8, 0, st, *a);
            else MTL_LAUNCH("attn_bwd_dkv_kernel<128, true, false>", fl_half, 0, (attn_bwd_dkv_kernel<128, true, false>), gk4, dim3(256), 0, st, *a);
        }
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    }
    // (the reprogramming attention's dQ kernel on 128-row workgroups, as its forward runs: 51.3 -> 62.7 us in-step — it is not pipelined and loses
    //  its second workgroup per CU; measured and left on the 64-row form)
#define MTL_BWD(DD)                                                                                        \
    if (f.causal && drop) {                                                                                \
        MTL_LAUNCH("attn_bwd_dq_kernel<" #DD ", true, true>", fl_half, 0, (attn_bwd_dq_kernel<DD, true, true>), gq, block, 0, st, *a);                    \
        MTL_LAUNCH("attn_bwd_dkv_kernel<" #DD ", true, true>", fl_half, 0, (attn_bwd_dkv_kernel<DD, true, true>), gk, block, 0, st, *a);                   \
    } else if (f.causal) {                                                                                 \
        MTL_LAUNCH("attn_bwd_dq_kernel<" #DD ", true, false>", fl_half, 0, (attn_bwd_dq_kernel<DD, true, false>), gq, block, 0, st, *a);                   \
        MTL_LAUNCH("attn_bwd_dkv_kernel<" #DD ", true, false>", fl_half, 0, (attn_bwd_dkv_kernel<DD, true, false>), gk, block, 0, st, *a);                  \
    } else if (drop) {                                                                                     \
        MTL_LAUNCH("attn_bwd_dq_kernel<" #DD ", false, true>", fl_half, 0, (attn_bwd_dq_kernel<DD, false, true>), gq, block, 0, st, *a);                   \
        MTL_LAUNCH("attn_bwd_dkv_kernel<" #DD ", false, true>", fl_half, 0, (attn_bwd_dkv_kernel<DD, false, true>), gk, block, 0, st, *a);                  \
    } else {                                                                                               \
        MTL_LAUNCH("attn_bwd_dq_kernel<" #DD ", false, false>", fl_half, 0, (attn_bwd_dq_kernel<DD, false, false>), gq, block, 0, st, *a);                  \
        MTL_LAUNCH("attn_bwd_dkv_kernel<" #DD ", false, false>", fl_half, 0, (attn_bwd_dkv_kernel<DD, false, false>), gk, block, 0, st, *a);                 \
    }
    if (f.D == 32) { MTL_BWD(32) } else if (f.D == 64) { MTL_BWD(64) } else { MTL_BWD(128) }
#undef MTL_BWD
    if (splits > 1) {
        const int64_t n = f.Tk * f.Hkv * f.D;
        hipLaunchKernelGGL(dkv_convert_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st, *a, splits);
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
