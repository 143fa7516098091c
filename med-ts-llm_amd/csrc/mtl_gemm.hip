// mtl_gemm.hip — bf16 MFMA GEMM (NT): C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias)
//
// gfx950 design (GUIDE cdna_hip_programming.md §5):
//   * 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles, fp32 accumulate.
//   * A/B tiles staged HBM -> LDS with the 16-byte LDS-DMA (global_load_lds_dwordx4): no VGPR round trip.
//     The LDS image is lane-linear, so the bank-conflict swizzle is applied on the per-lane SOURCE address
//     and again on the ds_read_b128 (rule 21): physical 16-B chunk = logical chunk ^ ((row >> 1) & 7).
//   * double-buffered LDS (2 x 32 KiB), one barrier per K tile, next tile's DMA issued before the MFMAs.
//   * MFMA operands are swapped (D = Btile . Atile^T) so each lane owns 4 CONSECUTIVE output columns of one
//     row: epilogue stores are 8-byte (bf16) / 16-byte (f32) vectors and bias/residual loads are vectors too.
//   * XCD-aware bijective block remap: each of the 8 XCDs gets a contiguous run of tiles that share A panels
//     in its private L2.
//   * split-K (fp32 slabs + reduce kernel) for the batch-independent mapping GEMM (M=1024, N=d_llm, K=V).
#include "mtl_common.h"
#include <cstdarg>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

// ---------------------------------------------------------------- launch profiler state (interface: mtl_common.h)
struct ProfRec { hipEvent_t e0, e1; int slot; };
struct ProfSlot { char name[128]; int kind; int64_t launches; double work; };
struct Profiler {
    std::mutex mu;
    std::atomic<bool> on{false};
    std::atomic<bool> shapes{false};   // mtl_prof_enable(2): GEMM rows are kept per problem size
    std::vector<ProfRec> recs;
    std::vector<ProfSlot> slots;
};
Profiler& prof() { static Profiler p; return p; }

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

#ifndef MTL_GEMM_FRAG_PIPE
#define MTL_GEMM_FRAG_PIPE 1      // 0: the compiler-scheduled fragment reads (diagnostic builds)
#endif
#ifndef MTL_XT_LA
#define MTL_XT_LA 2               // k-steps of register-staged look-ahead in gemm_xt_kernel
#endif
#ifndef MTL_GEMM_FRAG_D
#define MTL_GEMM_FRAG_D 3         // prefetch distance of the fragment pipeline in units of 4 MFMAs
#endif
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ int64_t remap_row(int64_t m, int64_t group_rows, int64_t group_stride, int64_t off) {
    if (group_rows == 0) return m;
    // 32-bit divide (mtl_gemm_nt rejects M or group_rows >= 2^31): the 64-bit one is a ~100-instruction branchy
    // expansion, paid 8-12 times per lane per tile by the row-mapped (pruned backward) GEMMs
    const uint32_t mm = (uint32_t)m, gr = (uint32_t)group_rows, q = mm / gr;
    return (int64_t)q * group_stride + off + (int64_t)(mm - q * gr);
}

#ifndef MTL_PERSIST_SNAKE
#define MTL_PERSIST_SNAKE 0   // MFMA order of the 8-wave kernels' k-step: 1 = snake over the row tiles (A/B: tools/build_variant.sh psnake only=mtl_gemm -DMTL_PERSIST_SNAKE=1)
#endif
#ifndef MTL_W4_NCHW
#define MTL_W4_NCHW 4      // column blocks per epilogue chunk of the residual-type epilogues (A/B builds: -DMTL_W4_NCHW=2|4)
#endif
template <int EPI, int CDT>
__device__ __forceinline__ void epilogue4(const mtl_gemm_args& p, int64_t m, int64_t n, f32x4 v, bool vec_ok) {
    const int64_t crow = remap_row(m, p.c_group_rows, p.c_group_stride, p.c_row_offset);
    float o[4];
    const int nvalid = (p.N - n) >= 4 ? 4 : (int)(p.N - n);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * p.alpha;
    if (p.bias) {
        if (vec_ok) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
            o[0] += b4.x; o[1] += b4.y; o[2] += b4.z; o[3] += b4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nvalid) o[e] += p.bias[n + e];
        }
    }
    if constexpr (EPI == MTL_EPI_GELU) {
        bf16_t* aux = reinterpret_cast<bf16_t*>(p.aux_out) + crow * p.ld_aux_out + n;
        const bool save = p.bwd_group_rows == 0 || (uint32_t)m % (uint32_t)p.bwd_group_rows >= (uint32_t)p.bwd_first_row;
        if (!save) {
        } else if (vec_ok) {
            u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2*>(aux) = pk;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nvalid) aux[e] = f32_to_bf16(o[e]);
        }
        // the activation sees the bf16-rounded pre-activation, exactly like a bf16 Linear followed by gelu_new
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_new_f(bf16_to_f32(f32_to_bf16(o[e])));
    } else if constexpr (EPI == MTL_EPI_RESID) {
        // v is rounded to bf16 first (a bf16 Linear output) and then added to the residual stream: fp32 (CDT f32: the reference's "mixed" stream) or
        // bf16 (CDT bf16 = its dtype "bf16": the dropped branch is a bf16 tensor too, and the sum is rounded by the store)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bf16_to_f32(f32_to_bf16(o[e]));
        if (p.drop_p > 0.f) {   // resid_pdrop: same (seed, physical row, column) hash as epilogue_wave
            const uint32_t thr = drop_threshold(p.drop_p), dbase = drop_base(p.drop_seed, 0u);
            const float sc = drop_scale_of(thr);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = drop_keep(dbase, (uint32_t)crow, (uint32_t)(n + e), thr) ? o[e] * sc : 0.f;
                if constexpr (CDT == MTL_BF16) o[e] = bf16_to_f32(f32_to_bf16(o[e]));
            }
        }
        if constexpr (CDT == MTL_BF16) {
            const bf16_t* r = reinterpret_cast<const bf16_t*>(p.aux_in) + crow * p.ld_aux_in + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nvalid) o[e] += bf16_to_f32(r[e]);
        } else {
            const float* r = reinterpret_cast<const float*>(p.aux_in) + crow * p.ld_aux_in + n;
            if (vec_ok) {
                const float4 r4 = *reinterpret_cast<const float4*>(r);
                o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nvalid) o[e] += r[e];
            }
        }
    } else if constexpr (EPI == MTL_EPI_DGELU) {
        const bf16_t* h = reinterpret_cast<const bf16_t*>(p.aux_in) + crow * p.ld_aux_in + n;
        if (vec_ok) {
            const u32x2 hk = *reinterpret_cast<const u32x2*>(h);
            o[0] *= dgelu_new_f(__uint_as_float(hk[0] << 16));
            o[1] *= dgelu_new_f(__uint_as_float(hk[0] & 0xffff0000u));
            o[2] *= dgelu_new_f(__uint_as_float(hk[1] << 16));
            o[3] *= dgelu_new_f(__uint_as_float(hk[1] & 0xffff0000u));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nvalid) o[e] *= dgelu_new_f(bf16_to_f32(h[e]));
        }
    }
    if constexpr (CDT == MTL_BF16) {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + n;
        if (vec_ok) {
            u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2*>(c) = pk;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nvalid) c[e] = f32_to_bf16(o[e]);
        }
    } else {
        float* c = reinterpret_cast<float*>(p.C) + crow * p.ldc + n;
        if constexpr (EPI == MTL_EPI_ACCUM) {
            if (vec_ok) {
                float4 c4 = *reinterpret_cast<float4*>(c);
                c4.x += o[0]; c4.y += o[1]; c4.z += o[2]; c4.w += o[3];
                *reinterpret_cast<float4*>(c) = c4;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nvalid) c[e] += o[e];
            }
        } else {
            if (vec_ok) {
                *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nvalid) c[e] = o[e];
            }
        }
    }
}

// Wave-level epilogue of a 64 x (NI*16) sub-tile, vector path: ALL auxiliary loads (bias, residual, saved
// pre-activation) are issued first, then the math and the stores. Calling epilogue4 per 16x16 tile interleaves
// load -> use -> store; the compiler may not move a later tile's load above an earlier tile's store (C may alias the
// residual), so every tile paid a full memory round trip: ~14 us of fixed cost on the N = 768 residual GEMMs.
// FULL = the whole tile lies inside C: no per-lane predicate at all, so the loads, the single wait and the stores come
// out as straight-line code. (With predicates every use sits in its own branch and hipcc waits vmcnt(0) in each, i.e.
// every store waits for the previous store to complete, and the still-"pending" bias registers force a vmcnt(0) in
// front of the next k-step's first ds_read, draining the LDS-DMA pipeline.)
// PAIR: the B fragments of column tiles 2t / 2t+1 were read from LDS rows permuted so that a lane's 4 + 4 columns are the 8
// CONSECUTIVE columns n_base + 32t + 8g .. +7: tile ni < NPT covers n_base + (ni/2)*32 + 8g + (ni%2)*4 + 0..3; an odd last tile
// (ni >= NPT) keeps the plain layout n_base + 16 ni + 4g + 0..3. `g` = lane >> 4.
// The epilogue of a wave's 64 x (NI*16) sub-tile runs as a SOFTWARE PIPELINE over column chunks (<= 4 column tiles each, so that the prefetched
// auxiliary operands — residual: 4 float4 per column tile — fit the register budget of the wide wave tiles): chunk c + 1's auxiliary loads
// (bias, residual, saved pre-activations) are issued BEFORE chunk c's stores. gfx950 retires loads and stores through ONE in-order counter
// (vmcnt): with load -> wait -> store per chunk, every chunk's wait for its loads also waited for the previous chunk's stores to be
// acknowledged — one full store round trip per chunk, 3 per tile on the 256 x 256 Llama tiles (NI = 8). Issued in this order the loads
// a chunk waits for are OLDER than the previous chunk's stores, and hipcc's own counted s_waitcnt leaves those stores in flight.
struct EpiRows { int64_t crow[4]; bool mok[4]; bool bwd_ok[4]; };

template <int EPI, bool FULL, int RS = 16>      // RS: rows between the lane's 4 row tiles (16: 16x16 MFMA tiles, 32: 32x32)
__device__ __forceinline__ EpiRows epi_rows(const mtl_gemm_args& p, int64_t m_first) {
    // lane owns rows m_first + mi*16 (mi = 0..3). Edge tiles (!FULL) LOAD from clamped (always valid) addresses and predicate only the
    // stores: no load result is ever consumed inside a branch.
    EpiRows r;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m_first + mi * RS;
        r.mok[mi] = FULL || m < p.M;
        r.crow[mi] = remap_row(r.mok[mi] ? m : p.M - 1, p.c_group_rows, p.c_group_stride, p.c_row_offset);
        r.bwd_ok[mi] = true;          // rows whose backward-only output is stored (GELU: aux_out, SWIGLU: C)
    }
    if constexpr (EPI == MTL_EPI_GELU || EPI == MTL_EPI_SWIGLU) {
        if (p.bwd_group_rows > 0) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                r.bwd_ok[mi] = (uint32_t)(m_first + mi * RS) % (uint32_t)p.bwd_group_rows >= (uint32_t)p.bwd_first_row;
        }
    }
    return r;
}

// auxiliary operands of one chunk of NI column tiles
template <int EPI, int NI>
struct EpiAux {
    float4 b4[NI];
    float4 res[EPI == MTL_EPI_RESID || EPI == MTL_EPI_ACCUM ? NI : 1][4];
    u32x2 hk[EPI == MTL_EPI_DGELU || EPI == MTL_EPI_RESID ? NI : 1][4];     // (RESID: the bf16 residual stream's 4 values, CDT bf16)
    u32x4 gq[EPI == MTL_EPI_DSWIGLU ? NI : 1][4];        // saved (gate, up) pairs of the lane's 4 activation columns
};

// PAIR: the B fragments of column tiles 2t / 2t+1 were read from LDS rows permuted so that a lane's 4 + 4 columns are the 8
// CONSECUTIVE columns n_base + 32t + 8g .. +7: tile ni < NPT covers n_base + (ni/2)*32 + 8g + (ni%2)*4 + 0..3; an odd last tile
// (ni >= NPT) keeps the plain layout n_base + 16 ni + 4g + 0..3. `g` = lane >> 4.
// columns of the chunk's NI column tiles; T0 = index of its first tile inside the wave's sub-tile, NPTW = how many of the wave's tiles are paired
// LAY = 1 (gemm_nt_w4_kernel, 32x32x16 MFMA): a "column tile" is one 4-register quad of a 32x32 accumulator = 8 columns shared by the two lane
// halves (g = lane >> 5); the B fragment rows are permuted so that quads 2t / 2t+1 of a lane are the 8 consecutive columns (t >> 1)*16 + 8g .. +7.
template <int NI, bool FULL, int NPTW, int T0, int LAY = 0>
__device__ __forceinline__ void epi_cols(const mtl_gemm_args& p, int64_t n_wave, const int g, int64_t (&ncol)[NI], bool (&nok)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int t = T0 + ni;
        const int64_t n = n_wave + (LAY == 1 ? (t >> 1) * 16 + g * 8 + (t & 1) * 4 : (t < NPTW ? (t >> 1) * 32 + g * 8 + (t & 1) * 4 : t * 16 + g * 4));
        nok[ni] = FULL || n < p.N;  // vector path: N % 4 == 0, so the 4 columns are valid together
        ncol[ni] = nok[ni] ? n : p.N - 4;
    }
}

template <int EPI, int CDT, int NI, bool FULL, int NPTW, int T0, int LAY = 0>
__device__ __forceinline__ void epi_load(const mtl_gemm_args& p, const EpiRows& r, int64_t n_base, const int g, EpiAux<EPI, NI>& a) {
    int64_t ncol[NI];
    bool nok[NI];
    epi_cols<NI, FULL, NPTW, T0, LAY>(p, n_base, g, ncol, nok);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) a.b4[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {                   // uniform branch around ALL bias loads
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) a.b4[ni] = *reinterpret_cast<const float4*>(p.bias + ncol[ni]);
    }
    if constexpr (EPI == MTL_EPI_RESID || EPI == MTL_EPI_ACCUM || EPI == MTL_EPI_DGELU || EPI == MTL_EPI_DSWIGLU) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if constexpr (EPI == MTL_EPI_RESID && CDT == MTL_F32) a.res[ni][mi] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.aux_in) + r.crow[mi] * p.ld_aux_in + ncol[ni]);
                if constexpr (EPI == MTL_EPI_RESID && CDT == MTL_BF16) a.hk[ni][mi] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.aux_in) + r.crow[mi] * p.ld_aux_in + ncol[ni]);
                if constexpr (EPI == MTL_EPI_ACCUM) a.res[ni][mi] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + r.crow[mi] * p.ldc + ncol[ni]);
                if constexpr (EPI == MTL_EPI_DGELU) a.hk[ni][mi] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.aux_in) + r.crow[mi] * p.ld_aux_in + ncol[ni]);
                if constexpr (EPI == MTL_EPI_DSWIGLU) a.gq[ni][mi] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.aux_in) + r.crow[mi] * p.ld_aux_in + 2 * ncol[ni]);
            }
    }
}

// output stores of the wave-level epilogue: plain stores. (Write-through `sc1` stores — the line is then not retained in the XCD's L2, so a
// 100-360 MB output would not evict the operand panels the main loop re-reads — were measured and are WORSE everywhere: Llama residual GEMM
// +7 %, SwiGLU +4.5 %, dSwiGLU +14 %, GPT-2 residual GEMM +18 %; profiles/r03_gemm_experiments.txt.)
template <typename V>
__device__ __forceinline__ void epi_st(void* ptr, const V v) { *reinterpret_cast<V*>(ptr) = v; }

// math of one chunk, IN PLACE: the lane's results replace its accumulators as raw bits —
//   fp32 outputs (STORE / RESID / ACCUM, CDT f32): acc[ni][mi] = the 4 output values
//   bf16 outputs: [0..1] = the 4 packed C values; GELU: [0..1] = saved pre-activation, [2..3] = activation; SWIGLU: [2] = the 2 packed
//   activation columns; DSWIGLU: [0..3] = the 8 packed d(gate | up) values
// so that the auxiliary registers are dead when the NEXT chunk's loads are issued into them, before this chunk's stores (epilogue_wave).
template <int EPI, int CDT, int NI, bool FULL, int NPTW, int T0, int LAY = 0>
__device__ __forceinline__ void epi_math(const mtl_gemm_args& p, const EpiRows& r, int64_t n_base, const int g, f32x4 (*acc)[4], const EpiAux<EPI, NI>& a,
                                         const int mi0 = 0, const int mi1 = 4) {
    int64_t ncol[NI];
    bool nok[NI];
    epi_cols<NI, FULL, NPTW, T0, LAY>(p, n_base, g, ncol, nok);
#pragma unroll
    for (int mi = mi0; mi < mi1; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float o[4] = {acc[ni][mi][0] * p.alpha + a.b4[ni].x, acc[ni][mi][1] * p.alpha + a.b4[ni].y, acc[ni][mi][2] * p.alpha + a.b4[ni].z,
                          acc[ni][mi][3] * p.alpha + a.b4[ni].w};
            u32x4 out = {0u, 0u, 0u, 0u};
            if constexpr (EPI == MTL_EPI_GELU) {
                const u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                // the activation sees the bf16-rounded pre-activation, exactly like a bf16 Linear followed by gelu_new
                out = (u32x4){pk[0], pk[1],
                              pack_bf16x2(gelu_new_f(__uint_as_float(pk[0] << 16)), gelu_new_f(__uint_as_float(pk[0] & 0xffff0000u))),
                              pack_bf16x2(gelu_new_f(__uint_as_float(pk[1] << 16)), gelu_new_f(__uint_as_float(pk[1] & 0xffff0000u)))};
            } else if constexpr (EPI == MTL_EPI_RESID) {
                const u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};   // bf16 Linear output, then fp32 add
                float v[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u), __uint_as_float(pk[1] << 16),
                              __uint_as_float(pk[1] & 0xffff0000u)};
                if (p.drop_p > 0.f) {                                                      // resid_pdrop (uniform branch)
                    // the lane's 4 columns n .. n+3 (n % 4 == 0) are one mask quad: one hash
                    const uint32_t thr = drop_threshold(p.drop_p), dbase = drop_base(p.drop_seed, 0u);
                    const float sc = drop_scale_of(thr);
                    const uint2 wq = drop_quad(dbase, (uint32_t)r.crow[mi], (uint32_t)ncol[ni] >> 2);
                    const uint32_t w0 = wq.x, w1 = wq.y;
                    v[0] = (w0 & 0xffffu) >= thr ? v[0] * sc : 0.f; v[1] = (w0 >> 16) >= thr ? v[1] * sc : 0.f;
                    v[2] = (w1 & 0xffffu) >= thr ? v[2] * sc : 0.f; v[3] = (w1 >> 16) >= thr ? v[3] * sc : 0.f;
                    if constexpr (CDT == MTL_BF16) {                                       // bf16 stream: the dropped branch is a bf16 tensor
                        const u32x2 dk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        v[0] = __uint_as_float(dk[0] << 16); v[1] = __uint_as_float(dk[0] & 0xffff0000u);
                        v[2] = __uint_as_float(dk[1] << 16); v[3] = __uint_as_float(dk[1] & 0xffff0000u);
                    }
                }
                if constexpr (CDT == MTL_BF16) {       // bf16 residual stream in and out (the reference's dtype "bf16"): one rounding of the sum
                    const u32x2 hr = a.hk[ni][mi];
                    out = (u32x4){pack_bf16x2(__uint_as_float(hr[0] << 16) + v[0], __uint_as_float(hr[0] & 0xffff0000u) + v[1]),
                                  pack_bf16x2(__uint_as_float(hr[1] << 16) + v[2], __uint_as_float(hr[1] & 0xffff0000u) + v[3]), 0u, 0u};
                } else
                out = (u32x4){__float_as_uint(a.res[ni][mi].x + v[0]), __float_as_uint(a.res[ni][mi].y + v[1]), __float_as_uint(a.res[ni][mi].z + v[2]),
                              __float_as_uint(a.res[ni][mi].w + v[3])};
            } else if constexpr (EPI == MTL_EPI_DGELU) {
                o[0] *= dgelu_new_f(__uint_as_float(a.hk[ni][mi][0] << 16)); o[1] *= dgelu_new_f(__uint_as_float(a.hk[ni][mi][0] & 0xffff0000u));
                o[2] *= dgelu_new_f(__uint_as_float(a.hk[ni][mi][1] << 16)); o[3] *= dgelu_new_f(__uint_as_float(a.hk[ni][mi][1] & 0xffff0000u));
                out = (u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), 0u, 0u};
            } else if constexpr (EPI == MTL_EPI_ACCUM) {
                out = (u32x4){__float_as_uint(o[0] + a.res[ni][mi].x), __float_as_uint(o[1] + a.res[ni][mi].y), __float_as_uint(o[2] + a.res[ni][mi].z),
                              __float_as_uint(o[3] + a.res[ni][mi].w)};
            } else if constexpr (EPI == MTL_EPI_DSWIGLU) {
                // d(act) is a bf16 tensor in the unfused chain: round first. d gate = dh*up*sig*(1 + g*(1 - sig)), d up = dh*g*sig
                const u32x2 dk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                const float dh[4] = {__uint_as_float(dk[0] << 16), __uint_as_float(dk[0] & 0xffff0000u), __uint_as_float(dk[1] << 16),
                                     __uint_as_float(dk[1] & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gv = __uint_as_float(a.gq[ni][mi][e] << 16), uv = __uint_as_float(a.gq[ni][mi][e] & 0xffff0000u);
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-gv));
                    out[e] = pack_bf16x2(dh[e] * uv * sg * (1.0f + gv * (1.0f - sg)), dh[e] * gv * sg);
                }
            } else if constexpr (EPI == MTL_EPI_SWIGLU) {
                const u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                // columns (n .. n+3) = (gate_j, up_j, gate_j+1, up_j+1), j = n/2; the activation sees the bf16-rounded Linear
                // outputs and silu's own output is a bf16 tensor before the product (HF:modeling_llama.py:176)
                const float g0 = __uint_as_float(pk[0] << 16), u0 = __uint_as_float(pk[0] & 0xffff0000u);
                const float g1 = __uint_as_float(pk[1] << 16), u1 = __uint_as_float(pk[1] & 0xffff0000u);
                const float s0 = bf16_to_f32(f32_to_bf16(g0 * __builtin_amdgcn_rcpf(1.0f + __expf(-g0))));
                const float s1 = bf16_to_f32(f32_to_bf16(g1 * __builtin_amdgcn_rcpf(1.0f + __expf(-g1))));
                out = (u32x4){pk[0], pk[1], pack_bf16x2(s0 * u0, s1 * u1), 0u};     // 2 activation columns
            } else if constexpr (CDT == MTL_BF16) {
                out = (u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), 0u, 0u};
            } else {
                out = (u32x4){__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
            }
            acc[ni][mi] = __builtin_bit_cast(f32x4, out);
        }
    }
}

// stores of one chunk whose results epi_math left in the accumulators
template <int EPI, int CDT, int NI, bool FULL, int NPTW, int T0, int LAY = 0>
__device__ __forceinline__ void epi_store(const mtl_gemm_args& p, const EpiRows& r, int64_t n_base, const int g, f32x4 (*acc)[4], const bool dword_stores,
                                          const int mi0 = 0, const int mi1 = 4) {
    constexpr bool PAIR = NPTW > 0;
    constexpr int NPT = NPTW > T0 ? (NPTW - T0 < NI ? NPTW - T0 : NI) : 0;       // paired tiles of THIS chunk (they lead it)
    // outputs whose stores join a column-tile pair need both tiles of a pair in one chunk; fp32 / DSWIGLU outputs store per tile
    static_assert(NPT == 0 || EPI == MTL_EPI_DSWIGLU || (CDT == MTL_F32) || (T0 % 2 == 0 && NPT % 2 == 0), "a pair of column tiles stays inside one chunk");
    int64_t ncol[NI];
    bool nok[NI];
    epi_cols<NI, FULL, NPTW, T0, LAY>(p, n_base, g, ncol, nok);
    const int64_t (&crow)[4] = r.crow;
    // bf16 outputs of a column-tile pair leave as ONE 16-byte store per lane (8 consecutive columns) when the rows are 16-B aligned
    const bool wide_c = PAIR && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (p.ldc & 7) == 0;
    const bool wide_a = PAIR && (reinterpret_cast<uintptr_t>(p.aux_out) & 15) == 0 && (p.ld_aux_out & 7) == 0;
    auto put_bf16 = [&](bf16_t* base, int64_t ld, int mi, int ni, const u32x2& pk, bool ok, u32x2& hold, bool& hold_ok, bool wide) {
        bf16_t* dst = base + crow[mi] * ld + ncol[ni];
        if (!PAIR || ni >= NPT) {
            if (ok) epi_st<u32x2>(dst, pk);
        } else if ((ni & 1) == 0) {
            hold = pk; hold_ok = ok;
        } else {
            bf16_t* d0 = base + crow[mi] * ld + ncol[ni - 1];
            if (wide && hold_ok && ok) {
                epi_st<u32x4>(d0, (u32x4){hold[0], hold[1], pk[0], pk[1]});
            } else {
                if (hold_ok) epi_st<u32x2>(d0, hold);
                if (ok) epi_st<u32x2>(dst, pk);
            }
        }
    };
#pragma unroll
    for (int mi = mi0; mi < mi1; ++mi) {
        u32x2 hold_c = {0u, 0u}, hold_a = {0u, 0u};
        uint32_t hold_s = 0u;
        bool hold_c_ok = false, hold_a_ok = false, hold_s_ok = false;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const bool ok = r.mok[mi] && nok[ni];
            const int64_t n = ncol[ni];
            const u32x4 v = __builtin_bit_cast(u32x4, acc[ni][mi]);
            if constexpr (EPI == MTL_EPI_DSWIGLU) {
                if (ok) epi_st<u32x4>(reinterpret_cast<bf16_t*>(p.C) + crow[mi] * p.ldc + 2 * n, v);
            } else if constexpr (EPI == MTL_EPI_GELU) {
                put_bf16(reinterpret_cast<bf16_t*>(p.aux_out), p.ld_aux_out, mi, ni, (u32x2){v[0], v[1]}, ok && r.bwd_ok[mi], hold_a, hold_a_ok, wide_a);
                put_bf16(reinterpret_cast<bf16_t*>(p.C), p.ldc, mi, ni, (u32x2){v[2], v[3]}, ok, hold_c, hold_c_ok, wide_c);
            } else if constexpr (CDT == MTL_BF16) {
                put_bf16(reinterpret_cast<bf16_t*>(p.C), p.ldc, mi, ni, (u32x2){v[0], v[1]}, ok && (EPI != MTL_EPI_SWIGLU || r.bwd_ok[mi]), hold_c, hold_c_ok, wide_c);
                if constexpr (EPI == MTL_EPI_SWIGLU) {
                    const uint32_t av = v[2];      // a tile pair's 4 activation columns leave in one 8-byte store
                    bf16_t* ad = reinterpret_cast<bf16_t*>(p.aux_out) + crow[mi] * p.ld_aux_out + (n >> 1);
                    if (!PAIR || ni >= NPT) {
                        if (ok) epi_st<uint32_t>(ad, av);
                    } else if ((ni & 1) == 0) {
                        hold_s = av; hold_s_ok = ok;
                    } else {
                        bf16_t* a0 = reinterpret_cast<bf16_t*>(p.aux_out) + crow[mi] * p.ld_aux_out + (ncol[ni - 1] >> 1);
                        if (hold_s_ok && ok && (reinterpret_cast<uintptr_t>(p.aux_out) & 7) == 0 && (p.ld_aux_out & 3) == 0) {
                            epi_st<u32x2>(a0, (u32x2){hold_s, av});
                        } else {
                            if (hold_s_ok) epi_st<uint32_t>(a0, hold_s);
                            if (ok) epi_st<uint32_t>(ad, av);
                        }
                    }
                }
            } else {
                float* cp = reinterpret_cast<float*>(p.C) + crow[mi] * p.ldc + n;
                const float o[4] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
                if (EPI == MTL_EPI_STORE && dword_stores) {
                    // rows that are only 4-B aligned (N % 4 != 0, e.g. the [num_tokens, 50257] mapping weight gradient):
                    // four dword stores, still straight-line on interior tiles
                    if (FULL) {
                        // one dwordx4 store at a dword-aligned address: legal in the unaligned access mode ROCm runs
                        // compute queues in (SH_MEM_CONFIG.ALIGNMENT_MODE); 4x fewer store instructions than dwords
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        *reinterpret_cast<f4u*>(cp) = (f4u){o[0], o[1], o[2], o[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ok && n + e < p.N) cp[e] = o[e];
                    }
                } else {
                    if (ok) epi_st<f32x4>(cp, (f32x4){o[0], o[1], o[2], o[3]});
                }
            }
        }
    }
}

// Wave-level epilogue of a 64 x (NI*16) sub-tile, vector path. FULL = the whole tile lies inside C: no per-lane predicate at all, so the loads,
// the waits and the stores come out as straight-line code. (With predicates every use sits in its own branch and hipcc waits vmcnt(0) in
// each, i.e. every store waits for the previous store to complete, and the still-"pending" bias registers force a vmcnt(0) in front of
// the next k-step's first ds_read, draining the LDS-DMA pipeline.)
// Column tiles go NCH at a time (<= 4 in one piece; wider wave tiles 2 at a time: 64x96 .. 64x144 per wave); an odd count (NI = 9) ends with ONE
// unpaired column tile. Per chunk: wait for its auxiliary operands, math in place, ISSUE THE NEXT CHUNK'S LOADS (into the same registers), stores.
template <int EPI, int CDT, int NI, bool FULL, bool PAIR = false, int LAY = 0, int NCHW = 0>      // NCHW: column tiles per chunk (0 = the rule below)
__device__ __forceinline__ void epilogue_wave(const mtl_gemm_args& p, int64_t m_first, int64_t n_base, const int g, f32x4 (&acc)[NI][4],
                                              const bool dword_stores = false) {
    // PIPE (bias-only epilogues: plain store, GELU, SwiGLU): chunk c + 1's bias loads go out between chunk c's math and its stores, so the wait
    // for them does not drain chunk c's stores (in-step A/B inside one process: GELU GEMM 256x192 54.4 -> 50.7 us, plain 256x256 340.9 -> 337.5 us).
    // Residual-type epilogues (16 B of auxiliary operand per output quad) keep load -> math -> store per chunk: what they wait for is the
    // auxiliary loads' own round trip, and more chunks in flight do not fit the 256-register wide-tile kernels (measured, same A/B: two chunks
    // of prefetch spilled into the main loop, 222 -> 495 us; one-tile chunks exposed 8 load round trips per tile, dSwiGLU 376 -> 402 us).
    constexpr bool PIPE = (EPI == MTL_EPI_STORE || EPI == MTL_EPI_GELU || EPI == MTL_EPI_SWIGLU);
    constexpr int NCH = NCHW > 0 ? NCHW : (NI > 4 ? 2 : NI);
    constexpr int NFULL = NI / NCH;            // whole chunks
    constexpr int NTAIL = NI - NFULL * NCH;    // 0 or 1 column tile left over
    constexpr int NPTW = PAIR ? (NI & ~1) : 0; // paired column tiles of the wave's sub-tile
    static_assert(NFULL >= 1 && NFULL <= 4 && NTAIL <= 1, "chunk schedule");
    const EpiRows r = epi_rows<EPI, FULL, LAY == 1 ? 32 : 16>(p, m_first);
    EpiAux<EPI, NCH> a;
    EpiAux<EPI, NTAIL ? NTAIL : 1> at;
    epi_load<EPI, CDT, NCH, FULL, NPTW, 0, LAY>(p, r, n_base, g, a);
    // edge tiles: hipcc sinks the bias/residual math into the predicated store blocks, which leaves the load results "pending" at the
    // merge and costs a vmcnt(0) before every later ds_read; a compiler-visible wait per chunk settles it (edge tiles only).
#define MTL_EPI_NEXT(C)                                                                                                                  \
    if constexpr ((C) + 1 < NFULL) epi_load<EPI, CDT, NCH, FULL, NPTW, ((C) + 1) * NCH, LAY>(p, r, n_base, g, a);                              \
    else if constexpr (NTAIL > 0) epi_load<EPI, CDT, NTAIL, FULL, NPTW, NFULL * NCH, LAY>(p, r, n_base, g, at);
#define MTL_EPI_STEP(C)                                                                                                                  \
    if constexpr ((C) < NFULL) {                                                                                                          \
        if constexpr (!FULL) __builtin_amdgcn_s_waitcnt(0x0f70);   /* vmcnt(0) only (gfx9 encoding) */                                    \
        if constexpr (PIPE && ((C) + 1 < NFULL || NTAIL > 0)) {                                                                           \
            epi_math<EPI, CDT, NCH, FULL, NPTW, (C) * NCH, LAY>(p, r, n_base, g, &acc[(C) * NCH], a);                                          \
            MTL_EPI_NEXT(C)                                                                                                               \
            epi_store<EPI, CDT, NCH, FULL, NPTW, (C) * NCH, LAY>(p, r, n_base, g, &acc[(C) * NCH], dword_stores);                              \
        } else {                                                                                                                          \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {          /* row by row: the first stores leave while the rest computes */  \
                epi_math<EPI, CDT, NCH, FULL, NPTW, (C) * NCH, LAY>(p, r, n_base, g, &acc[(C) * NCH], a, mi, mi + 1);                          \
                epi_store<EPI, CDT, NCH, FULL, NPTW, (C) * NCH, LAY>(p, r, n_base, g, &acc[(C) * NCH], dword_stores, mi, mi + 1);              \
            }                                                                                                                             \
            MTL_EPI_NEXT(C)                                                                                                               \
        }                                                                                                                                 \
    }
    MTL_EPI_STEP(0)
    MTL_EPI_STEP(1)
    MTL_EPI_STEP(2)
    MTL_EPI_STEP(3)
#undef MTL_EPI_STEP
#undef MTL_EPI_NEXT
    if constexpr (NTAIL > 0) {
        if constexpr (!FULL) __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            epi_math<EPI, CDT, NTAIL, FULL, NPTW, NFULL * NCH, LAY>(p, r, n_base, g, &acc[NFULL * NCH], at, mi, mi + 1);
            epi_store<EPI, CDT, NTAIL, FULL, NPTW, NFULL * NCH, LAY>(p, r, n_base, g, &acc[NFULL * NCH], dword_stores, mi, mi + 1);
        }
    }
}

// SPLIT: raw fp32 partial sums go to workspace slab [split][M][N]; epilogue runs in splitk_reduce_kernel.
template <int EPI, int CDT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const mtl_gemm_args p, const int vec_ok_i) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][A|B]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const bool vec_ok = vec_ok_i != 0;

    // ---- XCD-aware bijective tile remap (GUIDE §5.5 T1): block b runs on XCD b % 8
    const int tiles_n = (int)((p.N + BN - 1) / BN);
    const int nwg = gridDim.x;
    int wgid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = wgid / tiles_n, tn = wgid % tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const int nkt_total = (int)(p.K / BK);
    int kt_begin = 0, kt_end = nkt_total;
    if constexpr (SPLIT) {
        const int S = gridDim.y, s = blockIdx.y;
        kt_begin = (int)((int64_t)s * nkt_total / S);
        kt_end = (int)((int64_t)(s + 1) * nkt_total / S);
    }

    // ---- per-thread staging sources: 4 x 16 B for A and for B per K tile
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = i * 256 + tid;       // 16-B slot in the lane-linear LDS image
        const int r = s >> 3, pc = s & 7;
        const int c = pc ^ ((r >> 1) & 7); // source-side swizzle
        int64_t am = m0 + r; if (am > p.M - 1) am = p.M - 1;
        int64_t bn = n0 + r; if (bn > p.N - 1) bn = p.N - 1;
        const int64_t arow = remap_row(am, p.a_group_rows, p.a_group_stride, p.a_row_offset);
        asrc[i] = reinterpret_cast<const bf16_t*>(p.A) + arow * p.lda + c * 8;
        bsrc[i] = reinterpret_cast<const bf16_t*>(p.B) + bn * p.ldb + c * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * 2 * TILE_BYTES;
        char* lb = la + TILE_BYTES;
        const int64_t koff = (int64_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[i] + koff),
                                             (lds_void_t*)(la + (i * 256 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[i] + koff),
                                             (lds_void_t*)(lb + (i * 256 + wave * 64) * 16), 16, 0, 0);
    };

    f32x4 acc[4][4];  // [ni][mi]; element e <-> column n + e
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row r -> r*128 B, physical chunk = (ks*4 + g) ^ ((r >> 1) & 7); the wave/mi parts of
    // r are multiples of 16, so ((r >> 1) & 7) == (l15 >> 1).
    const int sw = l15 >> 1;
    const int a_off = (wr * 64 + l15) * 128;
    const int b_off = (wc * 64 + l15) * 128;

    if (kt_begin < kt_end) {
        stage(0, kt_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
            const char* la = smem + cur * 2 * TILE_BYTES;
            const char* lb = la + TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pc16 = ((ks * 4 + g) ^ sw) * 16;
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    af[mi] = *reinterpret_cast<const bf16x8*>(la + a_off + mi * 16 * 128 + pc16);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    bfr[ni] = *reinterpret_cast<const bf16x8*>(lb + b_off + ni * 16 * 128 + pc16);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
            // raw barrier: __syncthreads() in a loop with LDS-DMA in flight makes hipcc drain vmcnt(0) in front of the
            // k-step's first ds_read as well (tools/asm_waits.py), i.e. no load/compute overlap at all
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }

    // ---- epilogue: lane owns row m = ... + l15, columns n .. n+3 with n = ... + g*4
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wr * 64 + mi * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int64_t n = n0 + wc * 64 + ni * 16 + g * 4;
            if (n >= p.N) continue;
            if constexpr (SPLIT) {
                float* w = reinterpret_cast<float*>(p.workspace) + ((int64_t)blockIdx.y * p.M + m) * p.N + n;
                const int nvalid = (p.N - n) >= 4 ? 4 : (int)(p.N - n);
                if (vec_ok) {
                    *reinterpret_cast<float4*>(w) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nvalid) w[e] = acc[ni][mi][e];
                }
            } else {
                epilogue4<EPI, CDT>(p, m, n, acc[ni][mi], vec_ok);
            }
        }
    }
}

// ---------------------------------------------------------------- persistent variant (short-K shapes)
// One workgroup per (CU slot) walks several output tiles; the K loop is FLAT over (tile, k-tile) pairs so the
// first LDS-DMA of the next tile is already in flight while the current tile's epilogue runs — with K = 768
// (12 k-tiles) the per-tile prologue latency and store tail are otherwise ~40 % of a tile's life time.
// Tiles are dealt per XCD in contiguous runs (blocks with equal blockIdx % 8 share an L2).
// BN_ = 128 (2 blocks/CU) or 64 (3 blocks/CU; used when N is small so that the tile count fills the chip evenly).
// swizzle of the 16-B chunk index inside a 128-B tile row: slot = (r&1)*8 + (c ^ (r>>1 & 7)) -> conflict-free
// ds_read_b128 lane groups (SQ_LDS_BANK_CONFLICT == 0 measured).
__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }
// B tiles read in column pairs: fragment lane i = 4a + b reads LDS row 32t + 8a + 4*(ni & 1) + b, and this swizzle gives that lane the
// same (row & 1, swizzle) = (i & 1, i >> 1) pair as swz64 gives lane i on row i: bank-for-bank the same conflict-free ds_read_b128
__device__ __forceinline__ int swz_pair(int row) { return (((row >> 3) & 3) << 1) | ((row >> 1) & 1); }

// grouped (GM rows at a time) tile order: consecutive linear ids form compact GM x n patches, so the workgroups that
// run concurrently on one XCD stream the SAME few A/B panels through its 4 MiB L2 (measured: the flat row-major
// order re-fetched B once per tile row -> 38x the algorithmic HBM bytes on the Llama shapes).
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
    const int gsize = GM * tiles_n;
    const int grp = t / gsize, rem = t - grp * gsize;
    const int first_m = grp * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    tm = first_m + rem % gm;
    tn = rem / gm;
}

// STAGES-deep LDS ring, ONE raw s_barrier per K tile, counted s_waitcnt vmcnt: STAGES-1 tiles of LDS-DMA stay in
// flight across the barrier (GUIDE §5 "Pipelining across barriers"); __syncthreads() would drain them (vmcnt(0)).
// BM_ x BN_ tile, NW waves = (BM_/64)(M) x rest(N); every wave owns a 64-row x (BN_/WN)-column sub-tile.
//   128x64 / 4 waves  (3 workgroups/CU)  small grids
//   128x128 / 8 waves (2 workgroups/CU)  same LDS bytes per stage feed twice the MFMA work per resident slot
//   256x128 / 16 waves, 3 stages (1 workgroup/CU): two 48-KiB stages in flight cover the loaded L2 latency with
//   4.2 MFLOP of MFMA work each and halve the L1->LDS bytes per FLOP again
// SPLIT: a work item is (k-slab s, tile): the slab's K range of that tile, raw fp32 partial sums to workspace slab s
// (splitk_reduce_kernel applies the epilogue). Slab-major item order: neighbours share A/B panels. `vec_ok_i` carries S.
// KS = 2 (grids of at most one tile per CU): the workgroup's waves form two groups that run the same pipeline over the two
// halves of K, each with its own LDS ring (twice the LDS-DMA bytes in flight and two waves per SIMD, which a lone 4-wave
// workgroup per CU lacks: M = B*n_grad backward GEMMs 46 -> ~30 us cold); group 1 hands its partial sums over through LDS.
// Host guarantees one tile per workgroup and an even number of k-tiles.
// waves per SIMD the launcher counts on (workgroups per CU by LDS x waves per workgroup / 4 SIMDs): told to the compiler so that an
// epilogue change cannot silently push a kernel over an occupancy cliff (the 128x192 / 8-wave qkv kernel went 126 -> 136 VGPRs and
// lost its second workgroup per CU: 38.5 -> 40.9 us)
// ---- fragment pipeline of the persistent kernel's k-tile (see the call site): inline-asm ds_reads + hand-counted lgkmcnt waits.
// D = prefetch distance in units of 4 MFMAs (one column tile x 4 row tiles x one 32-deep k half); the B fragments live in a ring of D + 1.
template <int KS_, int ROWB>
__device__ __forceinline__ void frag_read_a(bf16x8 (&afa)[2][4], const uint32_t (&abase)[2]) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(afa[KS_][0]) : "v"(abase[KS_]));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afa[KS_][1]) : "v"(abase[KS_]), "n"(16 * ROWB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afa[KS_][2]) : "v"(abase[KS_]), "n"(32 * ROWB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afa[KS_][3]) : "v"(abase[KS_]), "n"(48 * ROWB));
}
template <int NI, int NPT, int ROWB, int D, int U>    // unit U = ks * NI + ni -> ring slot U % (D + 1)
__device__ __forceinline__ void frag_read_b(bf16x8 (&bq)[D + 1], const uint32_t (&bpair)[2], const uint32_t (&bplain)[2]) {
    constexpr int ks = U / NI, ni = U % NI;
    if constexpr (ni < NPT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[U % (D + 1)]) : "v"(bpair[ks]), "n"(((ni >> 1) * 32 + (ni & 1) * 4) * ROWB));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[U % (D + 1)]) : "v"(bplain[ks]), "n"(ni * 16 * ROWB));
}
template <int NI, int NPT, int ROWB, int D, int U>    // B(0) .. B(D - 1): what is in flight when unit 0 starts
__device__ __forceinline__ void frag_read_first(bf16x8 (&bq)[D + 1], const uint32_t (&bpair)[2], const uint32_t (&bplain)[2]) {
    if constexpr (U < D) {
        frag_read_b<NI, NPT, ROWB, D, U>(bq, bpair, bplain);
        frag_read_first<NI, NPT, ROWB, D, U + 1>(bq, bpair, bplain);
    }
}
// program order of the reads: A(0) B(0) .. B(D - 1) | ahead of unit u: [A(1) if unit u + D opens the second k half] B(u + D)
template <int NI, int NPT, int ROWB, int D, int U>
__device__ __forceinline__ void frag_units(f32x4 (&acc)[NI][4], bf16x8 (&afa)[2][4], bf16x8 (&bq)[D + 1], const uint32_t (&abase)[2],
                                           const uint32_t (&bpair)[2], const uint32_t (&bplain)[2]) {
    constexpr int NU = 2 * NI;
    static_assert(NI >= D && D >= 1, "the second k half's A fragments are requested D units ahead of it");
    if constexpr (U < NU) {
        constexpr int ks = U / NI, ni = U % NI, slot = U % (D + 1);
        if constexpr (U + D < NU) {
            if constexpr (U + D == NI) frag_read_a<1, ROWB>(afa, abase);
            frag_read_b<NI, NPT, ROWB, D, U + D>(bq, bpair, bplain);
        }
        // reads younger than B(U): B(U+1) .. B(last), plus A(1)'s four when it was issued behind B(U)
        constexpr int last = U + D < NU ? U + D : NU - 1;
        constexpr int younger = (last - U) + ((U < NI && last >= NI) ? 4 : 0);
        if constexpr (ni == 0)
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(bq[slot]), "+v"(afa[ks][0]), "+v"(afa[ks][1]), "+v"(afa[ks][2]), "+v"(afa[ks][3]) : "n"(younger));
        else
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[slot]) : "n"(younger));
#pragma unroll
        for (int mj = 0; mj < 4; ++mj) {
            const int mi = (MTL_PERSIST_SNAKE && (ni & 1)) ? 3 - mj : mj;      // (snake order: consecutive MFMAs share a fragment — A/B builds, see gemm_nt_w4_kernel)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[slot], afa[ks][mi], acc[ni][mi], 0, 0, 0);
        }
        frag_units<NI, NPT, ROWB, D, U + 1>(acc, afa, bq, abase, bpair, bplain);
    }
}

constexpr int persist_waves_per_simd(int bm, int bn, int stages, int nw_all, int ks) {
    const int bn_lds = ((bn * 8 + (nw_all / ks) * 64 - 1) / ((nw_all / ks) * 64)) * ((nw_all / ks) * 64) / 8;
    const int lds = ks * stages * (bm + bn_lds) * 128;
    const int wgs = 160 * 1024 / lds < 1 ? 1 : 160 * 1024 / lds;
    const int w = wgs * nw_all / 4;
    return w < 1 ? 1 : (w > 4 ? 4 : w);        // (beyond 4 waves per SIMD nothing here is register-limited)
}
template <int EPI, int CDT, int BM_, int BN_, int STAGES, int NW_ALL, bool SPLIT = false, int KS = 1>
__global__ __launch_bounds__(NW_ALL * 64, persist_waves_per_simd(BM_, BN_, STAGES, NW_ALL, KS)) void gemm_nt_persist_kernel(const mtl_gemm_args p, const int vec_ok_i, const int tiles_m,
                                                                 const int tiles_n, const int gm_all) {
    constexpr int BK_ = 64;
    constexpr int NW = NW_ALL / KS;            // waves per k-group (all tile geometry below is per group)
    constexpr int NT = NW * 64;                // threads per k-group
    constexpr int WM = BM_ / 64;               // waves along M
    constexpr int WN = NW / WM;                // waves along N
    constexpr int WCOLS = BN_ / WN;            // columns per wave
    constexpr int NI = WCOLS / 16;             // 16-wide n tiles per wave
    constexpr bool PAIR = NI >= 2;             // column tiles read in pairs: a lane owns 8 consecutive output columns (epilogue_chunk)
    constexpr int NPT = PAIR ? (NI & ~1) : 0;  // an odd last tile keeps the plain layout
    constexpr int CPR = BK_ / 8;               // 16-B chunks per tile row
    constexpr int ROWB = BK_ * 2;              // bytes per tile row
    constexpr int NA = BM_ * CPR / NT;         // 16-B staging slots per thread, A tile
    constexpr int NB = (BN_ * CPR + NT - 1) / NT;   // ... B tile (256x96 / 8 waves: the B image is padded to 128 rows; the
    constexpr int BN_PAD = NB * NT / CPR;      //     extra rows re-load clamped rows and are never read)
    constexpr int NL = NA + NB;                // LDS-DMA instructions per wave per stage
    // fragment pipeline (below): for the instances that run at most two waves per SIMD — with four (128 x 192, 128 x 128 / 8 waves: capped at
    // 128 VGPRs) the other waves already cover a wave's LDS latency and the pipeline's extra registers only spill (qkv 128 x 192: 34.9 -> 37.3 us)
    constexpr bool FRAG_PIPE = MTL_GEMM_FRAG_PIPE != 0 && persist_waves_per_simd(BM_, BN_, STAGES, NW_ALL, KS) <= 2;
    constexpr int A_BYTES = BM_ * ROWB, B_BYTES = BN_PAD * ROWB;
    static_assert(BM_ * CPR % NT == 0, "A tile must be whole LDS-DMA instructions");
    constexpr int STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kgrp = KS > 1 ? wave_all / NW : 0;
    const int wave = KS > 1 ? wave_all % NW : wave_all;
    const int tid = wave * 64 + lane;          // thread index inside the k-group
    char* const smem = smem_all + (KS > 1 ? kgrp * STAGES * STAGE : 0);
    const int wr = wave / WN, wc = wave % WN;
    const int l15 = lane & 15, g = lane >> 4;
    const int n_split = SPLIT ? vec_ok_i : 1;  // (the persistent kernel is only launched when the vector epilogue applies)
    const int tiles_mn = tiles_m * tiles_n;

    const int ntiles = tiles_mn * n_split;
    const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int xblocks = (nblk - xcd + 7) >> 3;                 // blocks living on this XCD
    const int q = ntiles >> 3, r8 = ntiles & 7;
    const int t0 = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int cnt = q + (xcd < r8 ? 1 : 0);
    if (slot >= cnt) return;
    const int my_count = (cnt - slot + xblocks - 1) / xblocks;
    const int nkt = (int)(p.K / BK_) / n_split / KS;
    const int total = my_count * nkt;

    const bf16_t* asrc[NA];
    const bf16_t* bsrc[NB];
    // (Measured and dropped, profiles/r03_gemm_experiments.txt: an L2 warm-up load per lane and k-step, three k-steps ahead of the LDS-DMA that
    //  stages its line — the loads retire in order with the DMA (one vmcnt), so the miss it takes off the DMA's path lands in front of the NEXT
    //  stage's wait instead: 6-10 % slower on every shape.)
    const int gm = gm_all & 0xff;
    const int col_rot = (gm_all & (1 << 9)) ? (xcd * tiles_n) >> 3 : 0;   // per-XCD column rotation (host: chunks of whole tile rows only)
    // k-tile rotation per XCD (a tile's k-steps commute): the eight XCDs walk the SAME B panels; started at the same k they ask the
    // memory side for the same lines at the same moment. Workgroups of one XCD keep a common k, so they still share through its L2.
    const int s_off = (gm_all >> 8) ? (xcd * nkt) >> 3 : 0;
    auto set_sources = [&](int item) {
        int tm, tn;
        const int slab = SPLIT ? item / tiles_mn : 0;
        tile_coords(SPLIT ? item - slab * tiles_mn : item, tiles_m, tiles_n, gm, tm, tn);
        if (col_rot) { tn += col_rot; if (tn >= tiles_n) tn -= tiles_n; }
        const int64_t m0 = (int64_t)tm * BM_, n0 = (int64_t)tn * BN_;
        const int64_t k0 = (int64_t)(slab * KS + kgrp) * nkt * BK_;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int sl = i * NT + tid;
            const int rr = sl / CPR, pc = sl % CPR;
            const int c = pc ^ swz64(rr);
            int64_t am = m0 + rr; if (am > p.M - 1) am = p.M - 1;
            asrc[i] = reinterpret_cast<const bf16_t*>(p.A) + remap_row(am, p.a_group_rows, p.a_group_stride, p.a_row_offset) * p.lda + c * 8 + k0;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = i * NT + tid;
            const int rr = sl / CPR, pc = sl % CPR;
            const int loc = rr % WCOLS;            // row inside its wave's column block: the paired rows swizzle on it
            const int c = pc ^ (loc < NPT * 16 ? swz_pair(loc) : swz64(rr));
            int64_t bn = n0 + rr; if (bn > p.N - 1) bn = p.N - 1;
            bsrc[i] = reinterpret_cast<const bf16_t*>(p.B) + bn * p.ldb + c * 8 + k0;
        }
    };
    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * STAGE;
        char* lb = la + A_BYTES;
        int ktp = kt + s_off;
        if (ktp >= nkt) ktp -= nkt;
        const int64_t koff = (int64_t)ktp * BK_;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[i] + koff), (lds_void_t*)(la + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[i] + koff), (lds_void_t*)(lb + (i * NT + wave * 64) * 16), 16, 0, 0);
    };

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // epilogue of a finished item: the fused one, or (SPLIT) a raw fp32 store into the item's workspace slab
    auto finish = [&](int item) {
        int tm, tn;
        const int slab = SPLIT ? item / tiles_mn : 0;
        tile_coords(SPLIT ? item - slab * tiles_mn : item, tiles_m, tiles_n, gm, tm, tn);
        if (col_rot) { tn += col_rot; if (tn >= tiles_n) tn -= tiles_n; }
        const int64_t m0 = (int64_t)tm * BM_, n0 = (int64_t)tn * BN_;
        const bool full = m0 + BM_ <= p.M && n0 + BN_ <= p.N;
        if constexpr (SPLIT) {
            mtl_gemm_args q = p;
            q.C = reinterpret_cast<float*>(p.workspace) + (int64_t)slab * p.M * p.N;
            q.ldc = p.N; q.bias = nullptr; q.alpha = 1.f; q.c_group_rows = 0;
            if (full) epilogue_wave<MTL_EPI_STORE, MTL_F32, NI, true, PAIR>(q, m0 + wr * 64 + l15, n0 + wc * WCOLS, g, acc);
            else epilogue_wave<MTL_EPI_STORE, MTL_F32, NI, false, PAIR>(q, m0 + wr * 64 + l15, n0 + wc * WCOLS, g, acc);
        } else {
            const bool dw = vec_ok_i == 0;     // fp32 plain store without bias into 4-B aligned rows (host guarantees the rest)
            if (full) epilogue_wave<EPI, CDT, NI, true, PAIR>(p, m0 + wr * 64 + l15, n0 + wc * WCOLS, g, acc, dw);
            else epilogue_wave<EPI, CDT, NI, false, PAIR>(p, m0 + wr * 64 + l15, n0 + wc * WCOLS, g, acc, dw);
        }
    };
    const int sw = swz64(l15);                 // wave / mi / ni row offsets are multiples of 16: swz unchanged
    const int a_off = (wr * 64 + l15) * ROWB;
    const int b_off = (wc * WCOLS + l15) * ROWB;                                     // plain layout (an odd last column tile)
    const int b_off_pair = (wc * WCOLS + (l15 >> 2) * 8 + (l15 & 3)) * ROWB;         // paired column tiles
    // the LDS-DMA swizzles the paired rows on the row inside the wave's block: for lane i = 4a + b that is (a << 1) | (b >> 1) = i >> 1
    // = sw, so every lane keeps the (row & 1, swizzle) pair, i.e. the bank slot, it has in the plain layout (same pc16 for all tiles)
    static_assert(WCOLS % 16 == 0, "wave column blocks start on 16-row boundaries");

    int s_i = 0, s_kt = 0, s_buf = 0;   // next (tile index, k-tile, ring slot) to stage
    int c_i = 0, c_kt = 0, c_buf = 0;   // being computed
    int done_tile = -1;                 // finished tile whose epilogue is still pending
    // ---- prologue: STAGES-1 tiles in flight
    set_sources(t0 + slot);
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0) {
        if (s0 < total) {
            if (s0 > 0 && s_kt == 0) set_sources(t0 + slot + s_i * xblocks);
            stage(s_buf, s_kt);
            s_buf = (s_buf + 1 == STAGES) ? 0 : s_buf + 1;
            if (++s_kt == nkt) { s_kt = 0; ++s_i; }
        }
    }
    for (int it = 0; it < total; ++it) {
        // tile `it` landed once at most (STAGES-2) younger stages are still outstanding
        if (it + STAGES - 2 < total - 0 && STAGES > 2 && it + 1 < total) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * NL) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (done_tile >= 0) {           // epilogue of the previous tile: its stores are queued BEFORE the next DMA stage
            finish(done_tile);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
            done_tile = -1;
        }
        if (it + STAGES - 1 < total) {
            if (s_kt == 0) set_sources(t0 + slot + s_i * xblocks);
            stage(s_buf, s_kt);
            s_buf = (s_buf + 1 == STAGES) ? 0 : s_buf + 1;
            if (++s_kt == nkt) { s_kt = 0; ++s_i; }
        }
        const char* la = smem + c_buf * STAGE;
        const char* lb = la + A_BYTES;
        if constexpr (FRAG_PIPE) {
            // Fragment pipeline inside the wave. The k-tile's MFMAs go in 2 * NI units of 4 (one column tile x 4 row tiles x one 32-deep k
            // half); the B fragment of unit u + 2 (and, ahead of the second k half, its A fragments) is REQUESTED before unit u issues, so a
            // wave's ds_read latency runs under its own MFMAs. hipcc's own order is read -> s_waitcnt lgkmcnt(0) -> MFMAs per group (every
            // group waits out one LDS round trip), and with the reads merely moved up in the source it still waits lgkmcnt(0) behind every
            // second group. The reads are therefore issued from inline asm (invisible to its counter model) with hand-counted waits: LDS
            // operations return in order, so "the reads issued after this unit's fragment may stay out" is a counted lgkmcnt. A wait names
            // the registers the following MFMAs read as in/out operands, so they cannot be scheduled above it; asm statements keep their
            // order. Measured on the Llama shapes (cold pools, libraries alternated in one call): 256 x 256 plain -3.0 ... -6.7 %.
            const uint32_t la32 = (uint32_t)(uintptr_t)la, lb32 = (uint32_t)(uintptr_t)lb;       // (flat LDS address: the low half is the LDS offset)
            const uint32_t pck[2] = {((uint32_t)g ^ (uint32_t)sw) * 16u, ((4u + (uint32_t)g) ^ (uint32_t)sw) * 16u};
            const uint32_t abase[2] = {la32 + (uint32_t)a_off + pck[0], la32 + (uint32_t)a_off + pck[1]};
            const uint32_t bpair[2] = {lb32 + (uint32_t)b_off_pair + pck[0], lb32 + (uint32_t)b_off_pair + pck[1]};
            const uint32_t bplain[2] = {lb32 + (uint32_t)b_off + pck[0], lb32 + (uint32_t)b_off + pck[1]};
            constexpr int FD = NI >= MTL_GEMM_FRAG_D ? MTL_GEMM_FRAG_D : NI;
            bf16x8 afa[2][4], bq[FD + 1];
            frag_read_a<0, ROWB>(afa, abase);
            frag_read_first<NI, NPT, ROWB, FD, 0>(bq, bpair, bplain);
            frag_units<NI, NPT, ROWB, FD, 0>(acc, afa, bq, abase, bpair, bplain);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pc16 = ((ks * 4 + g) ^ sw) * 16;
            bf16x8 af[4], bfr[NI];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(la + a_off + mi * 16 * ROWB + pc16);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bfr[ni] = ni < NPT ? *reinterpret_cast<const bf16x8*>(lb + b_off_pair + ((ni >> 1) * 32 + (ni & 1) * 4) * ROWB + pc16)
                                   : *reinterpret_cast<const bf16x8*>(lb + b_off + ni * 16 * ROWB + pc16);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mj = 0; mj < 4; ++mj) {
                    const int mi = (MTL_PERSIST_SNAKE && (ni & 1)) ? 3 - mj : mj;
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
                }
        }
        }
        c_buf = (c_buf + 1 == STAGES) ? 0 : c_buf + 1;
        if (++c_kt == nkt) {
            done_tile = t0 + slot + c_i * xblocks;
            c_kt = 0;
            ++c_i;
        }
    }
    if constexpr (KS > 1) {
        // hand-over of group 1's partial sums: lane-linear 16-B slots (conflict-free), in the rings nobody reads any more
        static_assert(KS == 1 || (size_t)NW * NI * 4 * 64 * 16 <= (size_t)KS * STAGES * STAGE, "reduction scratch must fit the rings");
        f32x4* red = reinterpret_cast<f32x4*>(smem_all) + wave * (NI * 4 * 64) + lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave is past its last ds_read
        asm volatile("" ::: "memory");
        if (kgrp == 1) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) red[(ni * 4 + mi) * 64] = acc[ni][mi];
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] += red[(ni * 4 + mi) * 64];
    }
    if (done_tile >= 0) finish(done_tile);
}

// ---------------------------------------------------------------- 256 x 256 x 64, 4 waves, one wave per SIMD, hand-placed k-loop
// The Llama-class instance (round 6). What the 8-wave 256 x 256 kernel above could not be made to do — keep the matrix pipe issuing while the
// wave also reads fragments, stages the next k-tiles and meets its barriers — is a matter of per-instruction placement, so the k-loop of a tile
// is ONE asm statement (generated: tools/gen_gemm_w4_loop.py -> mtl_gemm_w4_loop.inc, which documents the time structure):
//   * 4 waves (2 x 2), each a 128 x 128 sub-tile as 8 x 8 output tiles of v_mfma_f32_16x16x32_bf16 = 256 AGPRs: tile (mi, ni) in the PHYSICAL quad
//     a[4 (8 mi + ni) .. + 3], named by the asm and handed to the C++ epilogue through 16 "={a[16 k : 16 k + 15]}" operands; 128 fragment VGPRs in
//     two k-step sets; 128 operand bytes per 32 KFLOP through the LDS instead of the 8-wave kernel's 192. (The first version used 4 x 4
//     v_mfma_f32_32x32x16_bf16: the same cycles, but the chip clocks 5 - 9 % lower under that shape — profiles/r06_gemm_w4_experiments.txt section 15.)
//   * LDS image: two 32 KiB buffers per operand, row r of a tile at r * 128 B, 16-B chunk c of it at slot c ^ swz(r) — written by the LDS-DMA
//     (buffer_load_dwordx4 ... lds: 8 rows per wave-instruction, the swizzle applied on the per-lane SOURCE offset), read back with the same XOR.
//     A image: swz = row bits 1..3 (swz64); B image: row bits 1, 3, 4 — its column blocks are read in pairs (lane row 8 (l15 >> 2) + (l15 & 3)).
//     The rows an instruction fetches are given by a wave-uniform SGPR offset per instruction (row-mapped A operands: 8-row pieces never straddle
//     a row group, the host checks), the k position by the descriptor's base address, advanced 128 B per k-tile.
//   * Swapped MFMA operands as everywhere in this file (D = Btile . Atile^T): a lane owns output row l15 of a row block and 4 consecutive columns of
//     a column block; the paired B reads make the 4 + 4 columns of blocks 2t / 2t + 1 the 8 consecutive columns 32 t + 8 g .. + 7 (epilogue_wave's
//     PAIR layout: 16-byte bf16 stores) — the epilogue is the 8-wave kernels' own, called once per 64-row half of the wave's sub-tile.
// Whole tiles only (M % 256 == 0, N % 256 == 0); everything else stays with gemm_nt_persist_kernel.
#include "mtl_gemm_w4_loop.inc"
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int EPI, int CDT, int VAR = 0>      // VAR > 0: timing ablations of diagnostic builds (-DMTL_DIAG_W4VAR; wrong results)
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(const mtl_gemm_args p, const int vec_ok_i, const int tiles_m, const int tiles_n, const int gm_all) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int ntiles = tiles_m * tiles_n;
    // split-K (gm_all bits 16..23 = S > 1): the work items are (tile, k-slab) pairs, slab-major (neighbouring items share panels at the same k); every item
    // stores its raw fp32 partial sums into workspace slab [S][M][N] and splitk_reduce_kernel applies the epilogue. Slabs may be uneven (nkt / S rounded
    // up for the first nkt % S slabs): the item's k-tile count and first k-tile travel to the asm in its packed word.
    const int S = (gm_all >> 16) & 0xff;
    const bool split = S > 1;
    const int nitems = split ? ntiles * S : ntiles;
    const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int xblocks = (nblk - xcd + 7) >> 3;                 // blocks living on this XCD
    const int q = nitems >> 3, r8 = nitems & 7;
    const int t0 = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int cnt = q + (xcd < r8 ? 1 : 0);
    const int gm = gm_all & 0xff;
    const int nkt = (int)(p.K / 64);
    // per-XCD k rotation (gm_all bit 8; one-round grids): XCD x stages its tiles' k-tiles from x * nkt / 8 on and wraps, so that the eight XCDs are
    // never on the same lines of the shared panels at the same moment (a tile's k-steps commute; the fp32 summation order then depends on the XCD)
    int rot = (gm_all & (1 << 8)) ? (xcd * nkt) >> 3 : 0;
#ifdef MTL_DIAG_W4VAR      // timing experiments: bit 11 = per-WORKGROUP k stagger (workgroups of an XCD at different k), bit 10 = linear (unswizzled) DMA source
    if (gm_all & (1 << 11)) rot = (int)(((unsigned)slot * (unsigned)nkt) >> 5) % nkt;
#endif
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_all;       // (flat LDS address: the low half is the LDS offset)
    // fragment read addresses: lane (l15, g) reads row l15 of a 16-row block, 16-byte chunk 4 ks + g of its k-tile row (k-step ks; the asm derives both
    // k-steps and adds 2 KiB per block); block bases are multiples of 16 rows, so the row's swizzle is swz64(l15)
    const uint32_t rba = lds0 + (uint32_t)((wr * 128 + l15) * 128), xa = (uint32_t)(g ^ swz64(l15));
    // B column blocks are read in PAIRS (epilogue_wave's PAIR layout: a lane's 4 + 4 columns of blocks 2t / 2t + 1 are 8 consecutive columns): lane row
    // 8 (l15 >> 2) + (l15 & 3) of the pair's 32 rows (+ 4 for the odd block, an immediate in the asm). The B image's swizzle uses row bits 1, 3, 4
    // (swzb below, the LDS-DMA writes it the same way): the 16 rows of such a read then fall on 16 distinct bank groups.
    const int browl = (l15 >> 2) * 8 + (l15 & 3);
    const uint32_t rbb = lds0 + 0x10000u + (uint32_t)((wc * 128 + browl) * 128), xb = (uint32_t)(g ^ (((browl >> 1) & 1) | (((browl >> 3) & 3) << 1)));
    // LDS-DMA: instruction i of wave w stages tile rows (4 i + w) * 8 .. + 7; lane -> row lane / 8, slot lane % 8
    const uint32_t dma = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)wave * 1024u));
    const int drow = lane >> 3, dsl = lane & 7;
    int dsw = ((wave & 1) * 4 + (lane >> 4)) & 7;                                   // swz64 of the row (tile row bases are multiples of 32)
#ifdef MTL_DIAG_W4VAR
    if (gm_all & (1 << 10)) dsw = 0;
#endif
    const uint32_t voa = (uint32_t)(drow * (int)p.lda * 2 + ((dsl ^ dsw) << 4));
    int dswb = ((lane >> 4) & 1) | (wave << 1);           // B image: swzb(row) = bit 1 | bits 3..4 << 1 of the tile row (4 i + wave) * 8 + drow
#ifdef MTL_DIAG_W4VAR
    if (gm_all & (1 << 10)) dswb = 0;
#endif
    const uint32_t vob = (uint32_t)(drow * (int)p.ldb * 2 + ((dsl ^ dswb) << 4));

    // (Measured and dropped, profiles/r06_gemm_w4_experiments.txt: starting XCD x late by x * 0.4 / 0.8 / 1.6 us so that the rounds' store bursts of the
    //  eight XCDs interleave with the other XCDs' main loops — in-step Llama-2-7B 102.29 / 102.11 / 101.95 / 102.09 ms per step, i.e. nothing.)
#ifdef MTL_DIAG_W4VAR      // phase stamps of the workgroup's FIRST tile (100 MHz realtime counter): entry, k-loop begin / end, epilogue end -> workspace[6 * block]
    uint64_t stamp[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0}, cyc[2] = {0, 0};       // cyc: the SHADER clock counter around the k-loop (cycles per k-tile, actual clock)
#endif
    // row offsets of the 8 + 8 LDS-DMA instructions of this wave for tile (m0, n0): lanes 0..7 carry A's, lanes 8..15 B's (the asm reads them out with v_readlane)
    auto row_table = [&](int64_t m0, int64_t n0) -> uint32_t {
        const int j = lane & 7;
        const int64_t row = (int64_t)((4 * j + wave) * 8);
        const int64_t arow = remap_row(m0 + row, p.a_group_rows, p.a_group_stride, p.a_row_offset);
        return (lane & 8) ? (uint32_t)((n0 + row) * p.ldb * 2) : (uint32_t)(arow * p.lda * 2);
    };
    // tile chaining (mtl_gemm_w4_loop.inc): a workgroup's tile i stages k-tiles 0 / 1 of its tile i + 1 in its two trailing iterations, so that tile
    // i + 1's k-loop starts as soon as tile i's epilogue stores are ISSUED (they drain under its first iterations) instead of after they have drained
    // plus a load round trip. Even k-tile counts only (the next tile's k-tile 0 must land in LDS buffer 0); no per-XCD rotation.
    const bool chain = !split && (nkt & 1) == 0 && rot == 0 && VAR == 0;
    int tm_next = 0, tn_next = 0;
    uint32_t tab_next = 0;
    bool have_next = false;                        // the previous iteration already worked out this tile (it staged its first k-tiles)
    for (int i = slot; i < cnt; i += xblocks) {
        int tm = tm_next, tn = tn_next;
        const int item = t0 + i, slab = split ? item / ntiles : 0;
        if (!have_next) tile_coords(split ? item - slab * ntiles : item, tiles_m, tiles_n, gm, tm, tn);
        int nkt_i = nkt, k0 = 0;
        if (split) {
            const int qk = nkt / S, rk = nkt - qk * S;
            k0 = slab * qk + (slab < rk ? slab : rk);
            nkt_i = qk + (slab < rk ? 1 : 0);
        }
        const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
        const uint32_t tab = have_next ? tab_next : row_table(m0, n0);
        uint32_t tabn = tab;
        int flags = (chain && i > slot) ? 1 : 0;
        have_next = chain && i + xblocks < cnt;
        if (have_next) {
            tile_coords(t0 + i + xblocks, tiles_m, tiles_n, gm, tm_next, tn_next);
            tabn = tab_next = row_table((int64_t)tm_next * 256, (int64_t)tn_next * 256);
            flags |= 2;
        }
        flags |= (nkt_i << 2) | (k0 << 16);           // the asm's packed word: chaining flags | k-tile count | first k-tile
        // the wave's 8 x 8 output tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16): tile (row block mi, column block ni) lives in the PHYSICAL accumulator quad
        // a[4 (8 mi + ni) .. + 3] — the asm names them directly; c_k = a[16 k .. 16 k + 15] = tiles 4 k .. 4 k + 3 tells the compiler. Write-only for the asm
        // (the first MFMA of each tile takes C = 0).
        f32x16 c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15;
#define MTL_W4_RUN(ASM)                                                                                                                                  \
    asm volatile(ASM                                                                                                                                     \
                 : "={a[0:15]}"(c0), "={a[16:31]}"(c1), "={a[32:47]}"(c2), "={a[48:63]}"(c3), "={a[64:79]}"(c4), "={a[80:95]}"(c5), "={a[96:111]}"(c6),        \
                   "={a[112:127]}"(c7), "={a[128:143]}"(c8), "={a[144:159]}"(c9), "={a[160:175]}"(c10), "={a[176:191]}"(c11), "={a[192:207]}"(c12),             \
                   "={a[208:223]}"(c13), "={a[224:239]}"(c14), "={a[240:255]}"(c15)                                                                            \
                 : [pa] "s"(p.A), [pb] "s"(p.B), [voa] "v"(voa), [vob] "v"(vob), [tab] "v"(tab), [rba] "v"(rba), [xa] "v"(xa), [rbb] "v"(rbb),           \
                   [xb] "v"(xb), [dma] "s"(dma), [rot] "s"(rot), [tabn] "v"(tabn), [flags] "v"(flags)                                                     \
                 : MTL_W4_LOOP_CLOBBERS)
#ifdef MTL_DIAG_W4VAR
        if (i == slot) { stamp[1] = __builtin_amdgcn_s_memrealtime(); cyc[0] = __builtin_amdgcn_s_memtime(); }
#endif
#ifdef MTL_W4_ASM_SELECT      // A/B builds (tools/build_variant.sh ... -DMTL_W4_ASM_SELECT=MTL_W4_LOOP_ASM_V1): one of the generator's timing ablations in the product's place
        if constexpr (VAR == 0) MTL_W4_RUN(MTL_W4_ASM_SELECT);
#else
        if constexpr (VAR == 0) MTL_W4_RUN(MTL_W4_LOOP_ASM);
#endif
#ifdef MTL_DIAG_W4VAR
        else if constexpr (VAR == 1) MTL_W4_RUN(MTL_W4_LOOP_ASM_V1);
        else if constexpr (VAR == 2) MTL_W4_RUN(MTL_W4_LOOP_ASM_V2);
        else if constexpr (VAR == 3) MTL_W4_RUN(MTL_W4_LOOP_ASM_V3);
        else if constexpr (VAR == 4) MTL_W4_RUN(MTL_W4_LOOP_ASM_V4);
        else if constexpr (VAR == 5) MTL_W4_RUN(MTL_W4_LOOP_ASM_V5);
#endif
#undef MTL_W4_RUN
#ifdef MTL_DIAG_W4VAR
        if (i == slot) { cyc[1] = __builtin_amdgcn_s_memtime(); stamp[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
        // ---- epilogue: the wave's 128 x 128 as two 64-row halves of 4 row blocks x 8 column blocks: the 16 x 16-tile layout of the 8-wave kernels'
        //      wave-level epilogue (lane: row l15 of a row block, 4 consecutive columns 4 g .. 4 g + 3 of a column block)
        const f32x16* cc[16] = {&c0, &c1, &c2, &c3, &c4, &c5, &c6, &c7, &c8, &c9, &c10, &c11, &c12, &c13, &c14, &c15};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 pc[8][4];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int t = (half * 4 + mi) * 8 + ni;          // output tile index -> c_(t / 4), quad t % 4
                    const f32x16& v = *cc[t >> 2];
                    const int qd = (t & 3) * 4;
                    pc[ni][mi] = (f32x4){v[qd], v[qd + 1], v[qd + 2], v[qd + 3]};
                }
            // residual-type epilogues (16 B of auxiliary operand per output quad) go load -> math -> store per chunk of column blocks: 4 blocks per chunk
            // (64 auxiliary VGPRs — the fragment registers are dead here) = 4 dependent round trips per wave tile instead of the default rule's 8
            constexpr int NCHW = (EPI == MTL_EPI_RESID || EPI == MTL_EPI_ACCUM || EPI == MTL_EPI_DGELU || EPI == MTL_EPI_DSWIGLU) ? MTL_W4_NCHW : 0;
            if constexpr (EPI == MTL_EPI_STORE && CDT == MTL_F32) if (split) {       // raw partial sums -> this slab of the workspace (the host launches THIS instance for split-K; 16-byte aligned slabs)
                mtl_gemm_args ps = p;
                ps.C = reinterpret_cast<char*>(p.workspace) + (size_t)slab * (size_t)p.M * (size_t)p.N * sizeof(float);
                ps.ldc = p.N; ps.c_dtype = MTL_F32; ps.alpha = 1.0f; ps.bias = nullptr; ps.c_group_rows = 0;
                epilogue_wave<MTL_EPI_STORE, MTL_F32, 8, true, true, 0, 0>(ps, m0 + wr * 128 + half * 64 + l15, n0 + wc * 128, g, pc);
                continue;
            }
            epilogue_wave<EPI, CDT, 8, true, true, 0, NCHW>(p, m0 + wr * 128 + half * 64 + l15, n0 + wc * 128, g, pc, vec_ok_i == 0);      // (fp32 plain store into 4-B aligned rows: dword stores)
        }
#ifdef MTL_DIAG_W4VAR
        if (i == slot && p.workspace && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp[3] = __builtin_amdgcn_s_memrealtime();
            uint64_t* o = reinterpret_cast<uint64_t*>(p.workspace) + 6 * blockIdx.x;
            o[0] = stamp[0]; o[1] = stamp[1]; o[2] = stamp[2]; o[3] = stamp[3]; o[4] = cyc[0]; o[5] = cyc[1];
        }
#endif
    }
}

template <int EPI, int CDT>
__global__ void splitk_reduce_kernel(const mtl_gemm_args p, const int S, const int vec_ok_i) {
    const int64_t nq = (p.N + 3) / 4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.M * nq) return;
    const int64_t m = idx / nq, n = (idx % nq) * 4;
    const float* w = reinterpret_cast<const float*>(p.workspace) + m * p.N + n;
    const int nvalid = (p.N - n) >= 4 ? 4 : (int)(p.N - n);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int64_t slab = p.M * p.N;
    int s = 0;
    if (p.N % 4 == 0 && (reinterpret_cast<uintptr_t>(p.workspace) & 15) == 0) {
        // four slabs' loads in flight per round (a serial load -> add chain costs one memory round trip per slab: 17 us for the
        // 32-slab flatten-head reduction of 37 k outputs); the summation order stays slab 0, 1, 2, ...
        for (; s + 4 <= S; s += 4) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(w + (int64_t)(s + u) * slab);
#pragma unroll
            for (int u = 0; u < 4; ++u) v += t[u];
        }
    }
    for (; s < S; ++s) {
        const float* ws = w + (int64_t)s * slab;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nvalid) v[e] += ws[e];
    }
    epilogue4<EPI, CDT>(p, m, n, v, vec_ok_i != 0);
}

// =============================================================================================== transposed-operand GEMM (mtl_gemm_xt)
// C[M, N] = alpha * sum_k A(m, k) B(n, k) where either operand may be stored K-MAJOR ([K, M] / [K, N] row-major): the weight-gradient
// and input-gradient GEMMs of the trainable Linear layers, whose operands are what the forward already holds —
//     dW[n, k] = sum_m dY[m, n] X[m, k]      both operands K-major (the contraction runs over their rows)
//     dX[m, k] = sum_n dY[m, n] W[n, k]      B = W K-major
// — so that no transposed copy of dY, X or W is ever written. A K-major tile is staged row-major ([64 k][128 + 16]) and its MFMA
// fragments come from the hardware transpose read (ds_read_b64_tr_b16: two 8-byte reads give a lane the 8 contraction values of its
// output row); a K-contiguous tile is staged [128][64 + 8] and read with one ds_read_b128, as in the NT kernels. Both paddings make
// the reads conflict-free (row strides of 72 / 36 dwords). 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each), 64-deep k-steps, the next step's
// global loads in flight during the MFMAs, two workgroups per CU. Small problems by construction (trainable projections): no persistence, split over
// the contraction (grid.y) for parallelism, partial slabs reduced by xt_reduce_kernel. a_colsum (A K-major): sum_k A(m, k) — the bias
// gradient falls out of the dY tiles the weight-gradient GEMM loads anyway.
typedef __attribute__((ext_vector_type(4))) short xt_s16x4;
typedef __attribute__((address_space(3))) xt_s16x4 xt_lds_s16x4;
__device__ __forceinline__ bf16x8 xt_frag_t(const bf16_t* tile, int ldt, int r0, int c0, int l15) {     // rows r0..r0+7 of column c0 + l15
    const int off = (l15 >> 2) * ldt + c0 + (l15 & 3) * 4;
    union { bf16x8 v; xt_s16x4 h[2]; } f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xt_lds_s16x4*)(tile + r0 * ldt + off));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xt_lds_s16x4*)(tile + (r0 + 4) * ldt + off));
    return f.v;
}

struct xt_args {
    const bf16_t* A; int64_t lda; const bf16_t* B; int64_t ldb; void* C; int64_t ldc;
    int64_t M, N, K; float alpha; float* a_colsum; float* ws; int S;
};

template <bool AT, bool BT, int CDT>
__global__ __launch_bounds__(256, 2) void gemm_xt_kernel(const xt_args p) {
    constexpr int TM = 128, KC = 64, LDN = KC + 8, LDT = TM + 16, NCH = TM * KC / 8 / 256;      // 4 chunks per operand per thread per step
    constexpr int A_EL = AT ? KC * LDT : TM * LDN, B_EL = BT ? KC * LDT : TM * LDN;
    __shared__ __attribute__((aligned(16))) bf16_t sa[2][A_EL];
    __shared__ __attribute__((aligned(16))) bf16_t sb[2][B_EL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (int)((p.N + TM - 1) / TM);
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * TM, n0 = (int64_t)(blockIdx.x % tiles_n) * TM;
    const int s = blockIdx.y;
    const int64_t per = ((p.K + (int64_t)p.S * KC - 1) / ((int64_t)p.S * KC)) * KC;
    const int64_t k_begin = s * per, k_end = k_begin + per < p.K ? k_begin + per : p.K;

    // global -> register staging, one k-step ahead of the MFMAs
    auto fetch = [&](const bf16_t* src, int64_t ld, bool trans, int64_t r0, int64_t rows, int64_t k0, u32x4 (&v)[NCH]) __attribute__((always_inline)) {
        // every load is issued unconditionally from a clamped (always valid) address and zeroed afterwards: a load inside a branch
        // ends with a full vmcnt(0) wait at the merge, which serialises the chunks (measured: 2.5 us per k-step)
        bool ok[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            int64_t r, k;
            if (trans) { k = k0 + idx / 16; r = r0 + (idx % 16) * 8; }      // [K, rows]: 8 consecutive rows-dimension elements of contraction row k
            else { r = r0 + idx / 8; k = k0 + (idx % 8) * 8; }              // [rows, K]: 8 consecutive contraction elements of one row
            ok[i] = k < k_end && r < rows;
            const int64_t kc = k < k_end ? k : 0, rc = r < rows ? r : 0;
            v[i] = *reinterpret_cast<const u32x4*>(trans ? src + kc * ld + rc : src + rc * ld + kc);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (!ok[i]) v[i] = (u32x4){0u, 0u, 0u, 0u};
    };
    auto stash = [&](bf16_t* tile, bool trans, const u32x4 (&v)[NCH]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            if (trans) *reinterpret_cast<u32x4*>(tile + (idx / 16) * LDT + (idx % 16) * 8) = v[i];
            else *reinterpret_cast<u32x4*>(tile + (idx / 8) * LDN + (idx % 8) * 8) = v[i];
        }
    };
    f32x4 acc[4][4];      // [nj][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_cs = AT && p.a_colsum != nullptr && n0 == 0;
    // LA k-steps of look-ahead (LA x 32 KB of loads in flight per workgroup): with one, a step lasts as long as a load round trip
    constexpr int LA = MTL_XT_LA;
    static_assert(LA >= 2 && LA <= 4, "look-ahead depth");
    u32x4 va[LA][NCH], vb[LA][NCH];
#pragma unroll
    for (int d = 0; d < LA; ++d) {
        fetch(p.A, p.lda, AT, m0, p.M, k_begin + d * KC, va[d]);      // (rows past k_end load zeros)
        fetch(p.B, p.ldb, BT, n0, p.N, k_begin + d * KC, vb[d]);
    }
    auto step = [&](const int64_t k0, const int buf, u32x4 (&ra)[NCH], u32x4 (&rb)[NCH]) __attribute__((always_inline)) {
        stash(sa[buf], AT, ra);
        stash(sb[buf], BT, rb);
        if (want_cs) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    csum[2 * e] += __uint_as_float(ra[i][e] << 16);
                    csum[2 * e + 1] += __uint_as_float(ra[i][e] & 0xffff0000u);
                }
        }
        __syncthreads();                              // (this buffer was last read before the previous barrier)
        fetch(p.A, p.lda, AT, m0, p.M, k0 + LA * KC, ra);
        fetch(p.B, p.ldb, BT, n0, p.N, k0 + LA * KC, rb);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (AT) af[i] = xt_frag_t(sa[buf], LDT, ks * 32 + g * 8, wr * 64 + i * 16, l15);
                else af[i] = *reinterpret_cast<const bf16x8*>(sa[buf] + (wr * 64 + i * 16 + l15) * LDN + ks * 32 + g * 8);
                if (BT) bfr[i] = xt_frag_t(sb[buf], LDT, ks * 32 + g * 8, wc * 64 + i * 16, l15);
                else bfr[i] = *reinterpret_cast<const bf16x8*>(sb[buf] + (wc * 64 + i * 16 + l15) * LDN + ks * 32 + g * 8);
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[nj][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nj], af[mi], acc[nj][mi], 0, 0, 0);
        }
    };
    // register set i % LA, LDS buffer i % 2 for step i: unrolled over lcm(LA, 2) steps so that both indices are compile-time constants
    constexpr int UN = (LA % 2 == 0) ? LA : 2 * LA;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += UN * KC) {
        static_for<UN>([&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            if (I == 0 || k0 + I * KC < k_end) step(k0 + I * KC, I & 1, va[I % LA], vb[I % LA]);
        });
    }
    // lane owns row m0 + wr*64 + mi*16 + l15 and the four columns n0 + wc*64 + nj*16 + g*4 .. +3
    const bool vec = (p.N % 4 == 0) && (p.ldc % 4 == 0);
    float* slab = p.S > 1 ? p.ws + (int64_t)s * p.M * p.N : nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wr * 64 + mi * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const int64_t n = n0 + wc * 64 + nj * 16 + g * 4;
            if (n >= p.N) continue;
            const f32x4 v = acc[nj][mi];
            if (slab) {
                if (vec) *reinterpret_cast<f32x4*>(slab + m * p.N + n) = v;
                else for (int e = 0; e < 4; ++e) if (n + e < p.N) slab[m * p.N + n + e] = v[e];
            } else if (CDT == MTL_F32) {
                float* cp = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
                if (vec && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) *reinterpret_cast<f32x4*>(cp) = v * p.alpha;
                else for (int e = 0; e < 4; ++e) if (n + e < p.N) cp[e] = v[e] * p.alpha;
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n;
                if (vec && (reinterpret_cast<uintptr_t>(p.C) & 7) == 0) *reinterpret_cast<u32x2*>(cp) = (u32x2){pack_bf16x2(v[0] * p.alpha, v[1] * p.alpha), pack_bf16x2(v[2] * p.alpha, v[3] * p.alpha)};
                else for (int e = 0; e < 4; ++e) if (n + e < p.N) cp[e] = f32_to_bf16(v[e] * p.alpha);
            }
        }
    }
    if (want_cs) {         // column chunk (tid % 16) * 8 of the tile, summed over the 16 threads that share it (fixed order)
        __syncthreads();
        float* red = reinterpret_cast<float*>(&sa[0][0]);       // [16][128] floats = 8 KB <= sizeof(sa)
        static_assert(sizeof(sa) >= 16 * 128 * sizeof(float), "column-sum scratch");
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid / 16) * 128 + (tid % 16) * 8 + e] = csum[e];
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
            if (p.S > 1) p.ws[(int64_t)p.S * p.M * p.N + (int64_t)s * p.M + m0 + tid] = t;
            else p.a_colsum[m0 + tid] = t;
        }
    }
}

// sums the split slabs (fixed order) into C and the column-sum partials into a_colsum
template <int CDT>
__global__ void xt_reduce_kernel(const xt_args p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, mn = p.M * p.N;
    if (idx < mn) {
        float v = 0.f;
        for (int s = 0; s < p.S; ++s) v += p.ws[(int64_t)s * mn + idx];
        const int64_t m = idx / p.N, n = idx - m * p.N;
        if (CDT == MTL_F32) reinterpret_cast<float*>(p.C)[m * p.ldc + n] = v * p.alpha;
        else reinterpret_cast<bf16_t*>(p.C)[m * p.ldc + n] = f32_to_bf16(v * p.alpha);
    } else if (p.a_colsum && idx - mn < p.M) {
        float v = 0.f;
        for (int s = 0; s < p.S; ++s) v += p.ws[(int64_t)p.S * mn + (int64_t)s * p.M + (idx - mn)];
        p.a_colsum[idx - mn] = v;
    }
}


// Tile order of a launch, packed into the kernel's `gm_all` argument: bits 0-7 the rows g of a tile group (tile_coords), bit 8 the
// per-XCD k rotation, bit 9 the per-XCD column rotation. Each XCD walks a contiguous chunk of ~tiles/8 ids = g rows x (chunk/g) columns of tiles.
// * Short chunks (at most ~2 rounds of the XCD's resident workgroups; every GPT-2-small GEMM): g = sqrt(chunk*BN/BM) (at least
//   chunk/tiles_n) balances the A and B rows a chunk touches — PMC, residual GEMM 256x96: 100 MB read per launch with g = 8 ->
//   80.5 MB (= algorithmic). In-step sweep of one forced g for all launches: g in {4, 8, 16} 6.64 ms per step, every g that does
//   NOT divide the chunk 6.46-6.52 ms — with aligned groups all eight XCDs start on the same B panels at the same moment.
// * Per-XCD k rotation (two-k-group launches: one tile per workgroup, all tiles in flight together): XCD x starts its tiles at
//   k-tile x*nkt/8 and wraps (a tile's k-steps commute; the workgroups of one XCD keep a common k and still share operands
//   through its L2), so the XCDs are never on the same lines at once. Per-kernel A/B inside one process (bench.py's event
//   brackets, 3 alternations): two-k-group GEMMs 24.7 -> 22.2 us (-10 %), but the kernels whose workgroups walk several tiles
//   lose 3-5 % (qkv 36.7 -> 38.7, residual 36.7 -> 38.3, GELU 62.0 -> 65.1 us), so only the former rotate. (A rotation keyed on
//   the tile row instead, which breaks the sharing inside an XCD, is 2-4 % slower per step than none.) The fp32 summation
//   order of such a tile depends on the XCD it runs on: deterministic for a launch configuration, not bit-identical across
//   configurations.
// * Long chunks / long K (Llama grids): g = 8 — other g measured 0.5-0.9 % slower per Llama-2-7B step.
// (MTL_GEMM_FORCE, in the launcher, forces one tile configuration for one epilogue and N the same way.)
// MTL_GEMM_RULES_OFF=<bitmask> switches single launch rules off for in-step A/B runs of bench.py (1: 256x192 for the GELU GEMM,
// 2: 256x96 for residual GEMMs, 4: two k-groups, 8: per-XCD k rotation, 16: balanced group height, 32: XCD-affine rows + column rotation). Diagnostic only.
int rules_off() {
    static const int v = mtl_env_int("MTL_GEMM_RULES_OFF", 0);
    return v;
}

int tile_order(int tiles_m, int tiles_n, int bm, int bn, int per_cu, int64_t K, bool one_tile_per_wg = false) {
    const int nt = tiles_m * tiles_n;
    if (nt > 8 * 2 * 32 * per_cu) return 8;
    const int rotate = (one_tile_per_wg && K / BK <= 48 && !(rules_off() & 8)) ? 1 << 8 : 0;
    if (rules_off() & 16) return 8 | rotate;
#ifdef MTL_DIAG
    {   // diagnostic builds: MTL_GEMM_G="bm,g" forces the group height of the short-chunk grids of bm-row tiles (in-step A/B)
        static const char* spec = getenv("MTL_GEMM_G");
        static int fg[2] = {0, 0};
        static const bool parsed = spec && sscanf(spec, "%d,%d", &fg[0], &fg[1]) == 2;
        if (parsed && fg[0] == bm && fg[1] > 0) return (fg[1] > tiles_m ? tiles_m : fg[1]) | rotate;
    }
#endif
    // whole tile rows per XCD (tiles_m % 8 == 0): g = tiles_m/8 makes every XCD's chunk one group (its own eighth of the rows, all
    // columns: fewest A bytes) and a per-XCD column rotation (bit 9) keeps the XCDs off the same B panels. Per-kernel A/B against
    // the sqrt rule: qkv 35.6 -> 34.2, two-k-group 21.1 -> 20.5, dGELU 33.1 -> 32.2, residual 36.0 -> 34.8, GELU 61.0 -> 59.9 us.
    // Doing it for only some of a layer's GEMMs made every kernel 3-6 % slower; giving the norm and attention kernels the same
    // row -> XCD ownership changed nothing (tools/ab build, 3 alternations), so the effect is not producer -> consumer L2 reuse.
    if (tiles_m % 8 == 0 && tiles_m / 8 <= 255 && !(rules_off() & 32)) return (tiles_m / 8) | rotate | (1 << 9);   // (rotation needs g == chunk rows)
    const double chunk = nt / 8.0;
    double g = std::sqrt(chunk * bn / bm);
    if (g < chunk / tiles_n) g = chunk / tiles_n;
    int gi = (int)(g + 0.5);
    gi = gi < 1 ? 1 : (gi > tiles_m ? tiles_m : (gi > 255 ? 255 : gi));
    return gi | rotate;
}

bool aligned(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

// may this problem run on gemm_nt_w4_kernel? (whole 256 x 256 tiles, vector epilogue — or the plain fp32 store's dword form: the caller passes vec_ok || dword_ok —, operand offsets that fit the 32-bit buffer addressing of its
// LDS-DMA, row-mapped A operands whose 8-row staging pieces stay inside one row group)
bool w4_ok(const mtl_gemm_args& p, int vec_ok) {
    if (!vec_ok || p.M % 256 != 0 || p.N % 256 != 0 || p.K < 64) return false;
    const int64_t lim = (int64_t)1 << 31;
    int64_t a_last = p.M;
    if (p.a_group_rows > 0) {
        if (p.a_group_rows % 8 != 0 || p.a_row_offset < 0 || p.a_group_stride < 0) return false;
        a_last = ((p.M - 1) / p.a_group_rows) * p.a_group_stride + p.a_row_offset + p.a_group_rows;
    }
    return a_last * p.lda * 2 < lim && p.N * p.ldb * 2 < lim && p.lda > 0 && p.ldb > 0;
}

// per-call experiment knobs (mtl_gemm_args.tune_*): mode 0 = one tile per workgroup, 1 = persistent flat-K (the default); 0 = automatic elsewhere
struct Tuning { int mode = 1; int bn = 0; int stages = 0; int waves = 0; int bm = 0; };
Tuning call_tuning(const mtl_gemm_args& p) {
    Tuning t;
    t.mode = p.tune_mode == 1 ? 0 : 1;
    t.bm = p.tune_bm; t.bn = p.tune_bn; t.stages = p.tune_stages; t.waves = p.tune_waves;
    return t;
}
int num_cus() {       // (constant of the device, cached once)
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// kernel name of a launch for the profiler; mtl_prof_enable(2) appends the problem size (per-shape rows: tools/gemm_shapes.py)
__attribute__((format(printf, 4, 5))) void kname_shape(char* buf, size_t cap, const mtl_gemm_args& p, const char* fmt, ...) {
    const bool shapes = prof().shapes.load(std::memory_order_relaxed);
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, cap, fmt, ap);
    va_end(ap);
    if (shapes && n > 0 && (size_t)n < cap) snprintf(buf + n, cap - n, " [%lldx%lldx%lld]", (long long)p.M, (long long)p.N, (long long)p.K);
}

template <int EPI, int CDT>
int launch(const mtl_gemm_args& p, int vec_ok, hipStream_t st, int fbm = 0, int fbn = 0, int fstages = 0, int fnw = 0) {
    const int tiles_m = (int)((p.M + BM - 1) / BM), tiles_n = (int)((p.N + BN - 1) / BN);
    const int S = p.split_k > 1 ? p.split_k : 1;
    const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
    char kname[128];
    // the persistent kernel has the wave-level epilogue only: 16-B aligned rows, or (plain fp32 store, no bias) dword stores
    const bool dword_ok = EPI == MTL_EPI_STORE && CDT == MTL_F32 && !p.bias && aligned(p.C, 4) && p.c_group_rows == 0;
    const Tuning tn = call_tuning(p);
    if (S == 1 && tn.mode == 1 && (vec_ok || dword_ok) && p.N >= 4) {
        const int ncu = num_cus();
        // tile choice, measured on MI355X (tools/bench_gemm.py, profiles/r01_gemm_ab*.txt)
        int bm = fbm ? fbm : tn.bm, bn = fbn ? fbn : tn.bn, stages = fstages ? fstages : tn.stages, nw = fnw ? fnw : tn.waves;
        const int t128 = tiles_m * tiles_n;            // grid size in 128x128 tiles
        // Llama-class grids: 256x128 / 16 waves / 3 stages reaches 1.04-1.12 PF/s (also the M = B*n_grad backward GEMMs with long K)
        if (bm == 0) bm = (t128 >= 8 * ncu || (t128 >= 4 * ncu && p.K >= 4096)) ? 256 : 128;
        if (bn == 0) {
            bn = (bm == 256 || t128 >= 2 * ncu) ? 128 : 64;
            // widths that divide 768-multiples (GPT-2 family: 768 / 2304 / 3072): fewer operand bytes per FLOP through the
            // global->LDS path (the binding resource, profiles/r01_gemm_diag.txt) and whole numbers of tiles per CU.
            // Measured (tools/bench_gemm.py): 128x192 wins for plain / GELU epilogues on >= 2 tiles per CU (qkv 41 -> 35 us,
            // fc 66 -> 59 us), 128x96 wins or ties wherever 128x64 was chosen (mproj 52 -> 47 us); DGELU keeps 128x128.
            // Llama-class grids, plain epilogue: 256x192 / 8 waves (each wave 64x96) moves 22 % fewer operand bytes per FLOP
            // than 256x128 (qkv 775 -> 727 us = 1.13 PF/s, gate|up 1404 -> 1336 us); residual epilogues keep 256x128
            if (bm == 256 && (EPI == MTL_EPI_STORE || EPI == MTL_EPI_SWIGLU) && (p.N % 192 == 0 || p.N >= 8192)) { bn = 192; if (nw == 0) nw = 8; if (stages == 0) stages = 2; }
            // ... and 256x256 / 8 waves (each wave 64x128, 32 B/clk/CU of operand traffic at MFMA peak) beats both wherever the
            // last column tile wastes < 6 %: qkv 768 -> 683 us (1.21 PF/s), down 695 -> 638, o-proj 296 -> 254, dX 371 -> 329 us
            // (1.25 PF/s = 50 % of peak). Plain and residual epilogues only (the others do not fit the register budget).
            if (bm == 256 && (EPI == MTL_EPI_STORE || EPI == MTL_EPI_RESID || EPI == MTL_EPI_SWIGLU || EPI == MTL_EPI_DSWIGLU) &&
                ((p.N + 255) / 256) * 256 * 100 <= p.N * 106) {
                bn = 256; if (nw == 0 || nw == 8) nw = 8; if (stages == 0 || stages == 2) stages = 2;
            }
            // GELU GEMM on 768-multiples: 256x192 / 8 waves. Per-kernel A/B inside one bench.py process (event brackets, alternating
            // runs): 71.4 us with 128x192 -> 63.5 us; the warm same-buffer microbench prefers 128x192, the L2-cold one
            // (GEMM_COLD=1) ranks them as the step does. The dGELU GEMM keeps 128x128 (33.7 us vs 36.3 us with 256x192).
            if (bm == 128 && p.N % 192 == 0 && (int64_t)((p.M + 255) / 256) * (p.N / 192) >= ncu && EPI == MTL_EPI_GELU && !(rules_off() & 1)) {
                bm = 256; bn = 192; if (nw == 0) nw = 8; if (stages == 0) stages = 2;
            }
            // residual GEMMs on 96-multiples with at least one 256x96 tile per CU: one 8-wave workgroup per CU with a 3-deep ring
            // (96 KB of operands in flight instead of 2 x 28 KB; 21 % fewer operand bytes per FLOP than 128x96): cold aproj 25.2 ->
            // 24.5 us, mproj 59.3 -> 57.8 us, in-step -1.7 % per GPT-2-small step
            if (bm == 128 && EPI == MTL_EPI_RESID && p.N % 96 == 0 && (int64_t)((p.M + 255) / 256) * (p.N / 96) >= ncu && !(rules_off() & 2)) {
                bm = 256; bn = 96; if (nw == 0) nw = 8; if (stages == 0) stages = 3;
            }
            const int64_t t192 = (int64_t)tiles_m * (p.N / 192);
            if (bm == 128 && bn == 128 && p.N % 192 == 0 && t192 >= 2 * ncu && (EPI == MTL_EPI_STORE || EPI == MTL_EPI_GELU)) bn = 192;
            else if (bm == 128 && bn == 64 && p.N % 96 == 0) bn = 96;
        }
#ifdef MTL_DIAG
        {   // diagnostic builds: MTL_GEMM_FORCE="epi,N,bm,bn,stages,waves" forces one tile configuration for the launches of that epilogue and N
            static const char* spec = getenv("MTL_GEMM_FORCE");
            static int f[6] = {-1, 0, 0, 0, 0, 0};
            static const bool parsed = spec && sscanf(spec, "%d,%d,%d,%d,%d,%d", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5]) == 6;
            if (parsed && f[0] == EPI && f[1] == p.N && tn.bm == 0) { bm = f[2]; bn = f[3]; stages = f[4]; nw = f[5]; }
        }
#endif
        // 256 x 256 grids whose last round of tiles would leave most CUs idle: the columns that fill WHOLE rounds go in this launch, the
        // remaining columns in a second one with half-width tiles (256 x 128 / 16 waves / 3 stages: twice the tiles for the same columns).
        // Llama-2 gate|up with the prompt-row cache: [4096 x 22016 x 4096] = 16 x 86 tiles = 5.375 rounds of 256 -> 5 rounds + 96 half-width
        // pairs on 192 of the 256 CUs (rule 128 of MTL_GEMM_RULES_OFF switches it off). Column-split only: the outputs are disjoint column
        // ranges of the same buffers. Not for the residual epilogue (its dropout mask is indexed by the absolute column).
        if (bm == 256 && bn == 256 && fbm == 0 && tn.bm == 0 && !(rules_off() & 128) && p.N % 256 == 0 &&
            (EPI == MTL_EPI_STORE || EPI == MTL_EPI_GELU || EPI == MTL_EPI_DGELU || EPI == MTL_EPI_SWIGLU || EPI == MTL_EPI_DSWIGLU)) {
            const int tm256 = (int)((p.M + 255) / 256), tn256 = (int)(p.N / 256);
            if (tm256 <= ncu && ncu % tm256 == 0) {
                const int per_round = ncu / tm256, rem = tn256 % per_round;
                if (tn256 > per_round && rem > 0 && 2 * rem <= per_round) {          // the tail round would be at most half full
                    const int64_t n_main = (int64_t)(tn256 - rem) * 256;
                    mtl_gemm_args a = p, b = p;
                    a.N = n_main;
                    b.N = p.N - n_main;
                    b.B = reinterpret_cast<const bf16_t*>(p.B) + n_main * p.ldb;
                    if (p.bias) b.bias = p.bias + n_main;
                    const int64_t cmul = EPI == MTL_EPI_DSWIGLU ? 2 : 1;              // dSwiGLU writes (and reads) 2 columns per GEMM column
                    b.C = reinterpret_cast<char*>(p.C) + n_main * cmul * (CDT == MTL_BF16 ? 2 : 4);
                    if (p.aux_in) b.aux_in = reinterpret_cast<const char*>(p.aux_in) + n_main * cmul * 2;       // (bf16 for every epilogue listed)
                    if (p.aux_out) b.aux_out = reinterpret_cast<char*>(p.aux_out) + (EPI == MTL_EPI_SWIGLU ? n_main / 2 : n_main) * 2;
                    const int rc = launch<EPI, CDT>(a, vec_ok, st, 256, 256, 2, w4_ok(a, vec_ok || dword_ok) ? 4 : 8);
                    if (rc != MTL_OK) return rc;
                    return launch<EPI, CDT>(b, vec_ok, st, 256, 128, 3, 16);
                }
            }
        }
        // 256 x 256 / 4 waves: the hand-placed instance (gemm_nt_w4_kernel). Whole tiles, 32-bit operand offsets, 8-row pieces inside one row group.
        // Chosen wherever the 8-wave 256 x 256 tile was (cold operands, same box, profiles/r06_gemm_w4_vs_vendor.txt: [4096 x 4096 x 22016] 580 -> 510 us,
        // [4096 x 4096 x 4096] 122 -> 112, [4096 x 4096 x 11008] 297 -> 270, [4096 x 12288 x 4096] 345 -> 332); rule 256 of MTL_GEMM_RULES_OFF (diagnostic
        // builds) keeps the 8-wave kernel for in-step A/B runs.
        if (bm == 256 && bn == 256 && nw == 8 && stages == 2 && fnw == 0 && tn.waves == 0 && w4_ok(p, vec_ok || dword_ok) && !(rules_off() & 256)) nw = 4;
        if (bm == 256 && bn == 256 && nw == 4) {
            if (!w4_ok(p, vec_ok || dword_ok)) return MTL_ERR_UNSUPPORTED;
            const int tm = (int)(p.M / 256), tn = (int)(p.N / 256), nt = tm * tn;
            const int grid = nt < ncu ? nt : ncu;
            const size_t lds = 128 * 1024;
            // group height of the XCD chunks: a one-round grid gives every XCD 32 tiles; 4 rows x 8 columns of them touch the fewest panel rows
            int order = tile_order(tm, tn, 256, 256, 1, p.K) & 0xff;
            if (nt <= ncu && tm % 4 == 0) order = 4;
            // (per-XCD k rotation: measured no gain on this kernel, 542.8 vs 533.7 us at K = 22016 — profiles/r06_gemm_w4_experiments.txt — so it is off)
            const int rot_on = mtl_env_int("MTL_W4_ROT", 0);
            if (rot_on == 1) order |= 1 << 8;
            if (rot_on == 2) order |= 1 << 11;
            if (mtl_env_int("MTL_W4_LINSRC", 0)) order |= 1 << 10;
#ifdef MTL_DIAG_W4VAR
            if constexpr (EPI == MTL_EPI_STORE && CDT == MTL_BF16) {
#define MTL_W4_VARIANT(V)                                                                                                                  \
    if (stages == 2 + V) {                                                                                                                 \
        auto kv = gemm_nt_w4_kernel<EPI, CDT, V>;                                                                                          \
        (void)hipFuncSetAttribute((const void*)kv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
        hipLaunchKernelGGL(kv, dim3(grid), dim3(256), lds, st, p, vec_ok, tm, tn, order);                                                  \
        MTL_CHECK_LAUNCH();                                                                                                                \
        return MTL_OK;                                                                                                                     \
    }
                MTL_W4_VARIANT(1) MTL_W4_VARIANT(2) MTL_W4_VARIANT(3) MTL_W4_VARIANT(4) MTL_W4_VARIANT(5)
#undef MTL_W4_VARIANT
            }
#endif
            auto kfn = gemm_nt_w4_kernel<EPI, CDT>;
            static std::once_flag once;
            std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            kname_shape(kname, sizeof kname, p, "gemm_nt_w4_kernel<%d, %d>", EPI, CDT);
            MTL_LAUNCH(kname, flops, 0, kfn, dim3(grid), dim3(256), lds, st, p, vec_ok, tm, tn, order);
            MTL_CHECK_LAUNCH();
            return MTL_OK;
        }
        const bool auto_cfg = nw == 0 && stages == 0;
        if (nw == 0) nw = bm == 256 ? 16 : (bn >= 128 ? 8 : 4);
        if (stages == 0) stages = bm == 256 ? 3 : 2;
        const int tm = (int)((p.M + bm - 1) / bm), tn = (int)((p.N + bn - 1) / bn), nt = tm * tn;
        // at most one 128x96 tile per CU: two k-groups of 4 waves (KS = 2; tune: waves = 8 on a 128x96 tile)
        const int nkt_all = (int)(p.K / BK);
        int ks = 1;
        // (in-step A/B, GPT-2-small metric step: two k-groups 7.20 ms, one group with a 3-deep ring 7.38, 128x128 / 8 waves / 3 stages 7.22+,
        //  one group with 2 stages 7.60)
        if (bm == 128 && bn == 96 && nt <= ncu && nkt_all % 2 == 0 && nkt_all >= 4 && stages == 2 && ((auto_cfg && !(rules_off() & 4)) || nw == 8)) { ks = 2; nw = 8; }
        else if (bm == 128 && bn == 96 && nw == 8) return MTL_ERR_UNSUPPORTED;
        const int bn_lds = (bm == 256 && bn == 96) ? 128 : bn;      // B image padded to whole LDS-DMA instructions (8 waves)
        const size_t lds = (size_t)ks * stages * (bm + bn_lds) * BK * 2;
        const int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
        const int grid = nt < per_cu * ncu ? nt : per_cu * ncu;
#define MTL_PERSIST(BMV, BNV, STV, NWV)                                                                                \
    do {                                                                                                               \
        auto kfn = gemm_nt_persist_kernel<EPI, CDT, BMV, BNV, STV, NWV>;                                               \
        static std::once_flag once;                                                                                    \
        std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
        kname_shape(kname, sizeof kname, p, "gemm_nt_persist_kernel<%d, %d, %d, %d, %d, %d, false, 1>", EPI, CDT, BMV, BNV, STV, NWV); \
        MTL_LAUNCH(kname, flops, 0, kfn, dim3(grid), dim3(NWV * 64), lds, st, p, vec_ok, tm, tn, tile_order(tm, tn, bm, bn, per_cu, p.K)); \
    } while (0)
        if (ks == 2) {
            auto kfn = gemm_nt_persist_kernel<EPI, CDT, 128, 96, 2, 8, false, 2>;
            static std::once_flag once;
            std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            kname_shape(kname, sizeof kname, p, "gemm_nt_persist_kernel<%d, %d, 128, 96, 2, 8, false, 2>", EPI, CDT);
            MTL_LAUNCH(kname, flops, 0, kfn, dim3(nt), dim3(512), lds, st, p, vec_ok, tm, tn, tile_order(tm, tn, bm, bn, 1, p.K, true));
        } else if (bm == 256 && bn == 128 && nw == 16 && stages == 3) MTL_PERSIST(256, 128, 3, 16);
        else if (bm == 256 && bn == 128 && nw == 16 && stages == 2) MTL_PERSIST(256, 128, 2, 16);
        else if (bm == 128 && bn == 128 && nw == 8 && stages == 2) MTL_PERSIST(128, 128, 2, 8);
        else if (bm == 128 && bn == 128 && nw == 8 && stages == 3) MTL_PERSIST(128, 128, 3, 8);
        else if (bm == 128 && bn == 128 && nw == 4 && stages == 2) MTL_PERSIST(128, 128, 2, 4);
        else if (bm == 128 && bn == 64 && nw == 4 && stages == 2) MTL_PERSIST(128, 64, 2, 4);
        else if (bm == 128 && bn == 96 && nw == 4 && stages == 2) MTL_PERSIST(128, 96, 2, 4);
        else if (bm == 128 && bn == 96 && nw == 4 && stages == 3) MTL_PERSIST(128, 96, 3, 4);
        else if (bm == 128 && bn == 192 && nw == 8 && stages == 2) MTL_PERSIST(128, 192, 2, 8);
        else if (bm == 256 && bn == 192 && nw == 8 && stages == 2) MTL_PERSIST(256, 192, 2, 8);
        else if (bm == 256 && bn == 256 && nw == 8 && stages == 2) MTL_PERSIST(256, 256, 2, 8);
        else if (bm == 128 && bn == 64 && nw == 4 && stages == 3) MTL_PERSIST(128, 64, 3, 4);
        else if (bm == 256 && bn == 96 && nw == 8 && stages == 3) MTL_PERSIST(256, 96, 3, 8);
        else if (bm == 256 && bn == 96 && nw == 8 && stages == 2) MTL_PERSIST(256, 96, 2, 8);
        else return MTL_ERR_UNSUPPORTED;
#undef MTL_PERSIST
    } else if (S == 1) {
        if (EPI == MTL_EPI_SWIGLU || EPI == MTL_EPI_DSWIGLU) return MTL_ERR_UNSUPPORTED;      // the fused activation lives in the wave-level epilogue only
        kname_shape(kname, sizeof kname, p, "gemm_nt_kernel<%d, %d, false>", EPI, CDT);
        MTL_LAUNCH(kname, flops, 0, (gemm_nt_kernel<EPI, CDT, false>), dim3(tiles_m * tiles_n, 1), dim3(256), 0, st, p, vec_ok);
    } else {
        const int ws_vec = (p.N % 4 == 0) && aligned(p.workspace, 16);
        const int nkt_total = (int)(p.K / BK);
        const bool w4_split = tn.mode == 1 && ws_vec && aligned(p.workspace, 16) && S <= 255 && nkt_total / S >= 8 && w4_ok(p, 1) && !(rules_off() & 256);
        if (w4_split && (nkt_total % S != 0 || (tn.bm == 256 && tn.bn == 256 && tn.waves == 4))) {
            // split-K on the 4-wave kernel: (tile, k-slab) items with UNEVEN slabs, raw partial sums to the workspace, the reduce kernel below applies the
            // epilogue. Taken where S does not divide the k-steps (the 8-wave split path below needs that and such calls used to fall to the one-tile
            // kernel) or when forced (tune 256 x 256 / 4 waves). Not the default for divisible S: on the mapping GEMM [1024 x 768 x 51200] it measured
            // 131 us (S = 21) / 142 (16) against the 8-wave path's 125 (S = 16) — that GEMM streams 0.3 GB of operands and partial sums, it is not MFMA-bound
            // (tools/probes/mapping_fwd_split.py, profiles/r06_gemm_w4_experiments.txt section 23)
            const int tm = (int)(p.M / 256), tnn = (int)(p.N / 256), items = tm * tnn * S;
            const int ncu = num_cus(), grid = items < ncu ? items : ncu;
            const size_t lds = 128 * 1024;
            const int order = (tile_order(tm, tnn, 256, 256, 1, p.K) & 0xff) | (S << 16);
            auto kfn = gemm_nt_w4_kernel<MTL_EPI_STORE, MTL_F32>;
            static std::once_flag once;
            std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            kname_shape(kname, sizeof kname, p, "gemm_nt_w4_kernel<0, 0> split-K x%d", S);
            MTL_LAUNCH(kname, flops, 0, kfn, dim3(grid), dim3(256), lds, st, p, 1, tm, tnn, order);
        } else if (tn.mode == 1 && ws_vec && nkt_total % S == 0) {
            // persistent split-K: S x tiles work items of K/S each through the 128x128 / 8-wave pipeline (mapping GEMM:
            // K = padded vocabulary; the one-tile-per-workgroup kernel below reached 395 TF/s on it)
            auto go = [&](auto bmv, auto bnv) {
                constexpr int BMV = decltype(bmv)::value, BNV = decltype(bnv)::value, STV = 2, NWV = 8;
                const int tm = (int)((p.M + BMV - 1) / BMV), tn = (int)((p.N + BNV - 1) / BNV), items = tm * tn * S;
                const size_t lds = (size_t)STV * (BMV + BNV) * BK * 2;
                const int ncu = num_cus(), per_cu = (int)(160 * 1024 / lds);
                const int grid = items < per_cu * ncu ? items : per_cu * ncu;
                auto kfn = gemm_nt_persist_kernel<MTL_EPI_STORE, MTL_F32, BMV, BNV, STV, NWV, true>;
                static std::once_flag once;
                std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
                kname_shape(kname, sizeof kname, p, "gemm_nt_persist_kernel<0, 0, %d, %d, %d, %d, true, 1>", BMV, BNV, STV, NWV);
                MTL_LAUNCH(kname, flops, 0, kfn, dim3(grid), dim3(NWV * 64), lds, st, p, S, tm, tn, tile_order(tm, tn, BMV, BNV, per_cu, p.K));
            };
            using I128 = std::integral_constant<int, 128>; using I192 = std::integral_constant<int, 192>; using I256 = std::integral_constant<int, 256>;
            // operands of a split GEMM stream from HBM (K is huge): the fewest operand bytes per FLOP wins (256x192 when it still
            // leaves >= 1 item per CU: mapping GEMM 154 -> ~110 us)
            if (p.N % 192 == 0 && p.M % 256 == 0 && (p.M / 256) * (p.N / 192) * S >= num_cus()) go(I256{}, I192{});
            else if (p.N % 192 == 0) go(I128{}, I192{});
            else go(I128{}, I128{});
        } else
        {
            kname_shape(kname, sizeof kname, p, "gemm_nt_kernel<%d, %d, true>", EPI, CDT);
            MTL_LAUNCH(kname, flops, 0, (gemm_nt_kernel<EPI, CDT, true>), dim3(tiles_m * tiles_n, S), dim3(256), 0, st, p, ws_vec);
        }
        const int64_t items = p.M * ((p.N + 3) / 4);
        kname_shape(kname, sizeof kname, p, "splitk_reduce_kernel<%d, %d>", EPI, CDT);
        MTL_LAUNCH(kname, (double)S * p.M * p.N * 4.0 + (double)p.M * p.N * (CDT == MTL_BF16 ? 2.0 : 4.0), 1, (splitk_reduce_kernel<EPI, CDT>),
                   dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p, S, vec_ok);
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

}  // namespace

extern "C" size_t mtl_gemm_xt_workspace_bytes(int64_t M, int64_t N, int split_k) {
    return split_k > 1 ? (size_t)split_k * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float) : 0;
}

extern "C" int mtl_gemm_xt_auto_split_k(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128), ksteps = (K + 63) / 64;
    const int ncu = num_cus();
    if (tiles * 4 > ncu) return 1;                        // (a second pass over the output costs more than the idle CUs)
    int s = 1;
    while (s * 2 <= 64 && (int64_t)s * 2 * tiles <= 2 * ncu && ksteps / (s * 2) >= 4) s *= 2;      // two workgroups fit a CU
    return s;
}

extern "C" int mtl_gemm_xt(const mtl_gemm_xt_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return MTL_ERR_ARG;
    if (a->c_dtype != MTL_F32 && a->c_dtype != MTL_BF16) return MTL_ERR_ARG;
    if (!a->a_trans && !a->b_trans) return MTL_ERR_ARG;                      // (that is mtl_gemm_nt's problem)
    // 16-byte chunks: along K for a K-contiguous operand, along M / N for a K-major one
    if (a->lda % 8 != 0 || a->ldb % 8 != 0 || !aligned(a->A, 16) || !aligned(a->B, 16)) return MTL_ERR_ALIGN;
    if ((!a->a_trans || !a->b_trans) && a->K % 8 != 0) return MTL_ERR_ALIGN;
    if ((a->a_trans && a->lda < ((a->M + 7) & ~(int64_t)7)) || (a->b_trans && a->ldb < ((a->N + 7) & ~(int64_t)7))) return MTL_ERR_ARG;
    if (a->a_colsum && !a->a_trans) return MTL_ERR_ARG;
    const int S = a->split_k > 1 ? a->split_k : 1;
    if (S > 1 && (!a->workspace || a->workspace_bytes < mtl_gemm_xt_workspace_bytes(a->M, a->N, S))) return MTL_ERR_WORKSPACE;
    if (S > (a->K + 63) / 64) return MTL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    xt_args p = {reinterpret_cast<const bf16_t*>(a->A), a->lda, reinterpret_cast<const bf16_t*>(a->B), a->ldb, a->C, a->ldc,
                 a->M, a->N, a->K, a->alpha, a->a_colsum, reinterpret_cast<float*>(a->workspace), S};
    const dim3 grid((unsigned)(((a->M + 127) / 128) * ((a->N + 127) / 128)), (unsigned)S), block(256);
    const double flops = 2.0 * (double)a->M * (double)a->N * (double)a->K;
    char kname[96];
    {
        const bool shapes = prof().shapes.load(std::memory_order_relaxed);
        const int n = snprintf(kname, sizeof kname, "gemm_xt_kernel<%d, %d, %d>", a->a_trans ? 1 : 0, a->b_trans ? 1 : 0, S > 1 ? MTL_F32 : a->c_dtype);
        if (shapes && n > 0) snprintf(kname + n, sizeof kname - n, " [%lldx%lldx%lld /%d]", (long long)a->M, (long long)a->N, (long long)a->K, S);
    }
#define MTL_XT(AT, BT)                                                                                                   \
    do {                                                                                                                 \
        if (S > 1 || a->c_dtype == MTL_F32) MTL_LAUNCH(kname, flops, 0, (gemm_xt_kernel<AT, BT, MTL_F32>), grid, block, 0, st, p);   \
        else MTL_LAUNCH(kname, flops, 0, (gemm_xt_kernel<AT, BT, MTL_BF16>), grid, block, 0, st, p);                    \
    } while (0)
    if (a->a_trans && a->b_trans) MTL_XT(true, true);
    else if (a->a_trans) MTL_XT(true, false);
    else MTL_XT(false, true);
#undef MTL_XT
    if (S > 1) {
        const int64_t items = a->M * a->N + (a->a_colsum ? a->M : 0);
        const double bytes = (double)S * a->M * a->N * 4.0 + (double)a->M * a->N * (a->c_dtype == MTL_BF16 ? 2.0 : 4.0);
        if (a->c_dtype == MTL_F32) MTL_LAUNCH("xt_reduce_kernel<0>", bytes, 1, (xt_reduce_kernel<MTL_F32>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p);
        else MTL_LAUNCH("xt_reduce_kernel<1>", bytes, 1, (xt_reduce_kernel<MTL_BF16>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p);
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_gemm_tile_order(int tiles_m, int tiles_n, int bm, int bn, int per_cu, int64_t K, int one_tile_per_wg) {
    if (tiles_m <= 0 || tiles_n <= 0 || bm <= 0 || bn <= 0 || per_cu <= 0 || K <= 0) return MTL_ERR_ARG;
    return tile_order(tiles_m, tiles_n, bm, bn, per_cu, K, one_tile_per_wg != 0);
}

namespace mtlprof {
bool enabled() { return prof().on.load(std::memory_order_relaxed); }
bool begin(hipEvent_t* e0, hipEvent_t* e1) {
    if (hipEventCreate(e0) != hipSuccess) return false;
    if (hipEventCreate(e1) != hipSuccess) { (void)hipEventDestroy(*e0); return false; }
    return true;
}
void end(hipEvent_t e0, hipEvent_t e1, const char* name, double work, int kind) {
    Profiler& pf = prof();
    std::lock_guard<std::mutex> lk(pf.mu);
    int slot = -1;
    for (size_t i = 0; i < pf.slots.size(); ++i)
        if (strncmp(pf.slots[i].name, name, sizeof pf.slots[i].name) == 0) { slot = (int)i; break; }
    if (slot < 0) {
        ProfSlot sl = {};
        strncpy(sl.name, name, sizeof sl.name - 1);
        sl.kind = kind;
        pf.slots.push_back(sl);
        slot = (int)pf.slots.size() - 1;
    }
    pf.slots[slot].launches += 1;
    pf.slots[slot].work += work;
    pf.recs.push_back(ProfRec{e0, e1, slot});
}
}  // namespace mtlprof

extern "C" int mtl_prof_enable(int on) {
    Profiler& pf = prof();
    std::lock_guard<std::mutex> lk(pf.mu);
    pf.on.store(on != 0);
    pf.shapes.store(on == 2);
    if (on) {
        for (auto& r : pf.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        pf.recs.clear();
        pf.slots.clear();
    }
    return MTL_OK;
}

// fills up to `cap` rows, one per kernel instance launched since mtl_prof_enable(1); returns the number of instances.
// Synchronises the recorded events (call after the work has been enqueued; do not call concurrently with launches).
extern "C" int mtl_prof_read(mtl_prof_row* rows, int cap) {
    if (!rows || cap <= 0) return MTL_ERR_ARG;
    Profiler& pf = prof();
    std::lock_guard<std::mutex> lk(pf.mu);
    const int n = (int)pf.slots.size() < cap ? (int)pf.slots.size() : cap;
    for (int i = 0; i < n; ++i) {
        memset(&rows[i], 0, sizeof rows[i]);
        memcpy(rows[i].name, pf.slots[i].name, sizeof rows[i].name);
        rows[i].kind = pf.slots[i].kind;
        rows[i].launches = pf.slots[i].launches;
        rows[i].total_work = pf.slots[i].work;
        rows[i].min_ms = 1e30;
    }
    for (auto& r : pf.recs) {
        if (r.slot >= n) continue;
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { rows[r.slot].launches -= 1; continue; }
        rows[r.slot].total_ms += ms;
        if (ms < rows[r.slot].min_ms) rows[r.slot].min_ms = ms;
        if (ms > rows[r.slot].max_ms) rows[r.slot].max_ms = ms;
    }
    return n;
}

extern "C" size_t mtl_gemm_workspace_bytes(int64_t M, int64_t N, int split_k) {
    return split_k > 1 ? (size_t)split_k * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// Few output tiles and a long K (the flatten head: [B, d_ff*P] x [pred*C, d_ff*P]^T = 12 tiles of 128x96 with 256 k-steps each, 131 us on
// 12 CUs; the weight gradients of the small projections): split K so that about one work item per CU exists. 1 = do not split.
extern "C" int mtl_gemm_auto_split_k(int64_t M, int64_t N, int64_t K, int epilogue) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0) return 1;
    if (epilogue == MTL_EPI_ACCUM || epilogue == MTL_EPI_SWIGLU || epilogue == MTL_EPI_DSWIGLU || N % 4 != 0) return 1;
    const int64_t nkt = K / BK, bn = N % 192 == 0 ? 192 : 128;
    const int64_t tiles = ((M + 127) / 128) * ((N + bn - 1) / bn);
    const int ncu = num_cus();
    if (tiles * 4 > ncu || nkt < 32) return 1;
    int s = 1;
    while (s * 2 <= 32 && (int64_t)s * 2 * tiles <= ncu && nkt / (s * 2) >= 8 && nkt % (s * 2) == 0) s *= 2;
    return s;
}

extern "C" int mtl_gemm_nt(const mtl_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return MTL_ERR_ARG;
    const mtl_gemm_args& p = *a;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MTL_ERR_ARG;
    if (p.K % BK != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0) return MTL_ERR_ALIGN;
    if (!aligned(p.A, 16) || !aligned(p.B, 16)) return MTL_ERR_ALIGN;
    if (p.M >= (int64_t)1 << 31 || p.a_group_rows >= (int64_t)1 << 31 || p.c_group_rows >= (int64_t)1 << 31) return MTL_ERR_ARG;
    if (p.bwd_group_rows < 0 || p.bwd_group_rows >= (int64_t)1 << 31 || p.bwd_first_row < 0 || p.bwd_first_row > p.bwd_group_rows) return MTL_ERR_ARG;
    if (p.c_dtype != MTL_F32 && p.c_dtype != MTL_BF16) return MTL_ERR_ARG;
    if (p.tune_mode < 0 || p.tune_mode > 2 || (p.tune_bm != 0 && p.tune_bm != 128 && p.tune_bm != 256) ||
        (p.tune_bn != 0 && p.tune_bn != 64 && p.tune_bn != 128 && p.tune_bn != 96 && p.tune_bn != 192 && p.tune_bn != 256) ||
        (p.tune_stages != 0 && (p.tune_stages < 2 || p.tune_stages > 7)) || (p.tune_waves != 0 && p.tune_waves != 4 && p.tune_waves != 8 && p.tune_waves != 16))
        return MTL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int S = p.split_k > 1 ? p.split_k : 1;
    if (S > 1) {
        if (!p.workspace || p.workspace_bytes < mtl_gemm_workspace_bytes(p.M, p.N, S)) return MTL_ERR_WORKSPACE;
        if (S > p.K / BK) return MTL_ERR_ARG;
    }
    const size_t ce = p.c_dtype == MTL_BF16 ? 2 : 4;
    int vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0) && aligned(p.C, 4 * ce) && (!p.bias || aligned(p.bias, 16));
    switch (p.epilogue) {
        case MTL_EPI_STORE:
            return p.c_dtype == MTL_BF16 ? launch<MTL_EPI_STORE, MTL_BF16>(p, vec_ok, st) : launch<MTL_EPI_STORE, MTL_F32>(p, vec_ok, st);
        case MTL_EPI_GELU:
            if (p.c_dtype != MTL_BF16 || !p.aux_out) return MTL_ERR_ARG;
            vec_ok = vec_ok && (p.ld_aux_out % 4 == 0) && aligned(p.aux_out, 8);
            return launch<MTL_EPI_GELU, MTL_BF16>(p, vec_ok, st);
        case MTL_EPI_RESID:      // the residual stream (aux_in) has the output's dtype: fp32 ("mixed") or bf16 (the reference's dtype "bf16")
            if (!p.aux_in) return MTL_ERR_ARG;
            if (p.c_dtype == MTL_BF16) {
                vec_ok = vec_ok && (p.ld_aux_in % 4 == 0) && aligned(p.aux_in, 8);
                return launch<MTL_EPI_RESID, MTL_BF16>(p, vec_ok, st);
            }
            vec_ok = vec_ok && (p.ld_aux_in % 4 == 0) && aligned(p.aux_in, 16);
            return launch<MTL_EPI_RESID, MTL_F32>(p, vec_ok, st);
        case MTL_EPI_DGELU:
            if (p.c_dtype != MTL_BF16 || !p.aux_in) return MTL_ERR_ARG;
            vec_ok = vec_ok && (p.ld_aux_in % 4 == 0) && aligned(p.aux_in, 8);
            return launch<MTL_EPI_DGELU, MTL_BF16>(p, vec_ok, st);
        case MTL_EPI_ACCUM:
            if (p.c_dtype != MTL_F32 || S > 1) return MTL_ERR_ARG;
            return launch<MTL_EPI_ACCUM, MTL_F32>(p, vec_ok, st);
        case MTL_EPI_DSWIGLU:
            if (p.c_dtype != MTL_BF16 || !p.aux_in || S > 1 || p.bias) return MTL_ERR_ARG;
            if (!vec_ok || p.ldc % 8 != 0 || !aligned(p.C, 16) || p.ld_aux_in % 8 != 0 || !aligned(p.aux_in, 16)) return MTL_ERR_ALIGN;
            return launch<MTL_EPI_DSWIGLU, MTL_BF16>(p, vec_ok, st);
        case MTL_EPI_SWIGLU:
            if (p.c_dtype != MTL_BF16 || !p.aux_out || S > 1) return MTL_ERR_ARG;
            if (!vec_ok || p.ld_aux_out % 2 != 0 || !aligned(p.aux_out, 4)) return MTL_ERR_ALIGN;   // wave-level epilogue only
            return launch<MTL_EPI_SWIGLU, MTL_BF16>(p, vec_ok, st);
        default:
            return MTL_ERR_UNSUPPORTED;
    }
}
