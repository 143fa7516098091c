// mtl_norm.hip — LayerNorm / RMSNorm forward and dX-only backward over the fp32 residual stream.
//
// HBM-bound: one wave64 per row, the whole row lives in registers (NV float4 per lane, 16 B/lane coalesced
// loads), statistics by wave shuffles in fp32, bf16x4 (8 B/lane) stores. Algorithmic bytes per row:
// fwd 4d (read) + 2d (write); bwd 2d (dy) + 4d (x) + 4d (dres in) + 4d (+2d) (dres out).
#include "mtl_common.h"

#include <cstdio>
#include <type_traits>

namespace {

__device__ __forceinline__ int64_t remap_row(int64_t m, int64_t group_rows, int64_t group_stride, int64_t off) {
    if (group_rows == 0) return m;
    return (m / group_rows) * group_stride + off + (m % group_rows);
}

// the residual stream is fp32 (the reference's dtype = "mixed" / "fp32") or bf16 (its dtype = "bf16": R:tasks/base.py:261-262, the whole model in
// bf16): XT = float | bf16_t. 4 consecutive elements of a row <-> float4.
template <typename XT>
__device__ __forceinline__ float4 ld4(const XT* p) {
    if constexpr (std::is_same<XT, float>::value) {
        return *reinterpret_cast<const float4*>(p);
    } else {
        const u32x2 k = *reinterpret_cast<const u32x2*>(p);
        return make_float4(__uint_as_float(k[0] << 16), __uint_as_float(k[0] & 0xffff0000u), __uint_as_float(k[1] << 16), __uint_as_float(k[1] & 0xffff0000u));
    }
}
template <typename XT>
__device__ __forceinline__ void st4(XT* p, const float4 v) {
    if constexpr (std::is_same<XT, float>::value) {
        *reinterpret_cast<float4*>(p) = v;
    } else {
        const u32x2 k = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        *reinterpret_cast<u32x2*>(p) = k;
    }
}
__device__ __forceinline__ float rbf(float v) { return bf16_to_f32(f32_to_bf16(v)); }

#ifndef MTL_NORM_RPW
#define MTL_NORM_RPW 1      // rows per wave of the forward kernel (narrow rows): all rows' loads are in flight before the first reduction
#endif
template <int NV, bool RMS, int RPW = 1, typename XT = float>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                       int64_t ld_y, float* __restrict__ stats, int64_t M, int d, float eps,
                                                       int64_t group_rows, int64_t group_stride, int64_t row_offset, int stats_physical) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    float4 v[RPW][NV];
    int64_t prow[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int64_t row = row0 + j < M ? row0 + j : M - 1;
        prow[j] = remap_row(row, group_rows, group_stride, row_offset);
        const XT* xr = x + prow[j] * (int64_t)d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            v[j][i] = c < d ? ld4<XT>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // narrow rows: the affine parameters are requested with the rows (one round trip for everything); wide rows (NV float4 of the row per lane)
    // fetch them where they are used, as before — they hit the L2 and would otherwise double the live registers
    constexpr bool PRE = NV <= 4;
    float4 gm[PRE ? NV : 1], bt[PRE ? NV : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            gm[i] = c < d ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            bt[i] = (!RMS && c < d) ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float inv_d = 1.0f / (float)d;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        if (row0 + j >= M) break;
        const int64_t row = row0 + j;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[j][i].x + v[j][i].y) + (v[j][i].z + v[j][i].w);      // (columns >= d hold zeros)
        float mean = 0.f;
        if (!RMS) mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                const float a = v[j][i].x - mean, b = v[j][i].y - mean, cc = v[j][i].z - mean, dd = v[j][i].w - mean;
                q += (a * a + b * b) + (cc * cc + dd * dd);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
        bf16_t* yr = y + row * ld_y;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                float o0, o1, o2, o3;
                const float4 g4 = PRE ? gm[PRE ? i : 0] : *reinterpret_cast<const float4*>(gamma + c);
                if (RMS) {
                    // HF LlamaRMSNorm: weight * (x * rsqrt(var + eps)).to(input_dtype): the input dtype is fp32 on the fp32 stream; on the bf16
                    // stream the normalised row is a bf16 tensor before the (bf16) weight multiplies it (HF:models/llama/modeling_llama.py:64-69)
                    float n0 = v[j][i].x * rstd, n1 = v[j][i].y * rstd, n2 = v[j][i].z * rstd, n3 = v[j][i].w * rstd;
                    if constexpr (!std::is_same<XT, float>::value) { n0 = rbf(n0); n1 = rbf(n1); n2 = rbf(n2); n3 = rbf(n3); }
                    o0 = g4.x * n0; o1 = g4.y * n1; o2 = g4.z * n2; o3 = g4.w * n3;
                } else {
                    const float4 b4 = PRE ? bt[PRE ? i : 0] : *reinterpret_cast<const float4*>(beta + c);
                    o0 = (v[j][i].x - mean) * rstd * g4.x + b4.x; o1 = (v[j][i].y - mean) * rstd * g4.y + b4.y;
                    o2 = (v[j][i].z - mean) * rstd * g4.z + b4.z; o3 = (v[j][i].w - mean) * rstd * g4.w + b4.w;
                }
                u32x2 pk = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
                *reinterpret_cast<u32x2*>(yr + c) = pk;
            }
        }
        if (lane == 0 && stats) {
            const int64_t srow = stats_physical ? prow[j] : row;
            stats[srow * 2] = mean;
            stats[srow * 2 + 1] = rstd;
        }
    }
}

// WPR waves share a row (wide rows: d >= 2048): the row's NV float4 per lane would otherwise pin ~190 registers (two waves per SIMD, 3.3 TB/s
// at d = 4096); with four waves per row a lane keeps NV / 4 of them and the two row sums cross the waves through LDS.
template <int NV, bool RMS, int WPR = 1, typename XT = float>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const bf16_t* __restrict__ dy, int64_t ld_dy, const XT* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ stats,
                                                       const XT* dres_in, XT* dres_out, bf16_t* dres_out_bf16,
                                                       int64_t M, int d, int64_t group_rows, int64_t group_stride,
                                                       int64_t row_offset, int stats_physical, float drop_p, uint32_t drop_seed) {
    static_assert(NV % WPR == 0 && 4 % WPR == 0, "waves per row");
    constexpr int NVW = NV / WPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = wave % WPR;
    const int64_t row_raw = (int64_t)blockIdx.x * (4 / WPR) + wave / WPR;
    const bool row_ok = row_raw < M;
    if (WPR == 1 && !row_ok) return;
    const int64_t row = row_ok ? row_raw : M - 1;       // (WPR > 1: every wave reaches the barrier; stores are predicated)
    const int64_t prow = remap_row(row, group_rows, group_stride, row_offset);
    const XT* xr = x + prow * (int64_t)d;
    const bf16_t* dyr = dy + row * ld_dy;
    const int64_t srow = stats_physical ? prow : row;
    const float mean = RMS ? 0.f : stats[srow * 2];
    const float rstd = stats[srow * 2 + 1];
    float4 g[NVW], xh[NVW];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = (lane + 64 * (sub + WPR * i)) * 4;
        if (c < d) {
            const u32x2 dk = *reinterpret_cast<const u32x2*>(dyr + c);
            const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
            const float4 xv = ld4<XT>(xr + c);
            g[i].x = __uint_as_float(dk[0] << 16) * gm.x; g[i].y = __uint_as_float(dk[0] & 0xffff0000u) * gm.y;
            g[i].z = __uint_as_float(dk[1] << 16) * gm.z; g[i].w = __uint_as_float(dk[1] & 0xffff0000u) * gm.w;
            xh[i].x = (xv.x - mean) * rstd; xh[i].y = (xv.y - mean) * rstd;
            xh[i].z = (xv.z - mean) * rstd; xh[i].w = (xv.w - mean) * rstd;
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        } else {
            g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[i] = g[i];
        }
    }
    const float inv_d = 1.0f / (float)d;
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if constexpr (WPR > 1) {
        __shared__ float red[4][2];
        if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
        __syncthreads();
        s1 = 0.f; s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) { s1 += red[(wave / WPR) * WPR + w][0]; s2 += red[(wave / WPR) * WPR + w][1]; }
        if (!row_ok) return;
    }
    const float c1 = RMS ? 0.f : s1 * inv_d;
    const float c2 = s2 * inv_d;
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = (lane + 64 * (sub + WPR * i)) * 4;
        if (c < d) {
            float4 o;
            o.x = rstd * (g[i].x - c1 - xh[i].x * c2); o.y = rstd * (g[i].y - c1 - xh[i].y * c2);
            o.z = rstd * (g[i].z - c1 - xh[i].z * c2); o.w = rstd * (g[i].w - c1 - xh[i].w * c2);
            if (dres_in) {
                if constexpr (!std::is_same<XT, float>::value) {       // bf16 stream: the branch gradient is a bf16 tensor before the add
                    o.x = rbf(o.x); o.y = rbf(o.y); o.z = rbf(o.z); o.w = rbf(o.w);
                }
                const float4 r = ld4<XT>(dres_in + prow * (int64_t)d + c);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            st4<XT>(dres_out + prow * (int64_t)d + c, o);
            if constexpr (!std::is_same<XT, float>::value) { o.x = rbf(o.x); o.y = rbf(o.y); o.z = rbf(o.z); o.w = rbf(o.w); }
            if (dres_out_bf16) {
                if (drop_p > 0.f) {   // gradient entering a residual branch whose forward output was dropped with this mask
                    const uint32_t thr = drop_threshold(drop_p), dbase = drop_base(drop_seed, 0u);
                    const float sc = drop_scale_of(thr);
                    const uint2 wq = drop_quad(dbase, (uint32_t)prow, (uint32_t)c >> 2);
                        const uint32_t w0 = wq.x, w1 = wq.y;   // c % 4 == 0
                    o.x = (w0 & 0xffffu) >= thr ? o.x * sc : 0.f; o.y = (w0 >> 16) >= thr ? o.y * sc : 0.f;
                    o.z = (w1 & 0xffffu) >= thr ? o.z * sc : 0.f; o.w = (w1 >> 16) >= thr ? o.w * sc : 0.f;
                }
                u32x2 pk = {pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w)};
                *reinterpret_cast<u32x2*>(dres_out_bf16 + prow * (int64_t)d + c) = pk;
            }
        }
    }
}

template <bool RMS, typename F>
int dispatch_nv(int64_t d, F&& f) {
    const int64_t nv = (d + 255) / 256;
    if (nv <= 1) return f(std::integral_constant<int, 1>{});
    if (nv <= 2) return f(std::integral_constant<int, 2>{});
    if (nv <= 3) return f(std::integral_constant<int, 3>{});
    if (nv <= 4) return f(std::integral_constant<int, 4>{});
    if (nv <= 8) return f(std::integral_constant<int, 8>{});
    if (nv <= 16) return f(std::integral_constant<int, 16>{});
    if (nv <= 32) return f(std::integral_constant<int, 32>{});
    return MTL_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int mtl_norm_fwd_t(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int64_t ld_y, float* stats,
                              int64_t M, int64_t d, float eps, int rms, int64_t group_rows, int64_t group_stride,
                              int64_t row_offset, int stats_physical, void* stream) {
    if (!x || !gamma || !y || M <= 0 || d <= 0 || (!rms && !beta)) return MTL_ERR_ARG;
    if (x_dtype != MTL_F32 && x_dtype != MTL_BF16) return MTL_ERR_ARG;
    if (d % 4 != 0 || ld_y % 4 != 0) return MTL_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 block(256);
    auto go = [&](auto nv) -> int {
        constexpr int NV = decltype(nv)::value;
        constexpr int RPW = NV <= 4 ? MTL_NORM_RPW : 1;      // (wide rows already keep NV float4 per lane in flight)
        const dim3 grid((unsigned)((M + 4 * RPW - 1) / (4 * RPW)));
        char kname[64];
        snprintf(kname, sizeof kname, "norm_fwd_kernel<%d, %s%s>", NV, rms ? "true" : "false", x_dtype == MTL_BF16 ? ", bf16" : "");
        const double bytes = (double)M * d * ((x_dtype == MTL_BF16 ? 2 : 4) + 2) + (stats ? (double)M * 8 : 0.0);      // row in, bf16 row out, 2 statistics
#define MTL_NORM_FWD(RMSV, XT)                                                                                                            \
    MTL_LAUNCH(kname, bytes, 1, (norm_fwd_kernel<NV, RMSV, RPW, XT>), grid, block, 0, st, (const XT*)x, gamma, beta, (bf16_t*)y, ld_y, stats, M, \
               (int)d, eps, group_rows, group_stride, row_offset, stats_physical)
        if (x_dtype == MTL_BF16) { if (rms) MTL_NORM_FWD(true, bf16_t); else MTL_NORM_FWD(false, bf16_t); }
        else { if (rms) MTL_NORM_FWD(true, float); else MTL_NORM_FWD(false, float); }
#undef MTL_NORM_FWD
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    };
    return dispatch_nv<false>(d, go);
}

extern "C" int mtl_norm_fwd(const float* x, const float* gamma, const float* beta, void* y, int64_t ld_y, float* stats,
                            int64_t M, int64_t d, float eps, int rms, int64_t group_rows, int64_t group_stride,
                            int64_t row_offset, int stats_physical, void* stream) {
    return mtl_norm_fwd_t(x, MTL_F32, gamma, beta, y, ld_y, stats, M, d, eps, rms, group_rows, group_stride, row_offset, stats_physical, stream);
}

extern "C" int mtl_norm_bwd_t(const void* dy, int64_t ld_dy, const void* x, int stream_dtype, const float* gamma, const float* stats,
                              const void* dres_in, void* dres_out, void* dres_out_bf16, int64_t M, int64_t d, int rms,
                              int64_t group_rows, int64_t group_stride, int64_t row_offset, int stats_physical,
                              float bf16_drop_p, uint32_t bf16_drop_seed, void* stream) {
    if (!dy || !x || !gamma || !stats || !dres_out || M <= 0 || d <= 0 || bf16_drop_p < 0.f || bf16_drop_p >= 1.f) return MTL_ERR_ARG;
    if (stream_dtype != MTL_F32 && stream_dtype != MTL_BF16) return MTL_ERR_ARG;
    if (d % 4 != 0 || ld_dy % 4 != 0) return MTL_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 block(256);
    auto go = [&](auto nv) -> int {
        constexpr int NV = decltype(nv)::value;
        constexpr int WPR = NV >= 8 ? 4 : 1;
        const dim3 grid((unsigned)((M + 4 / WPR - 1) / (4 / WPR)));
        char kname[64];
        snprintf(kname, sizeof kname, "norm_bwd_kernel<%d, %s%s>", NV, rms ? "true" : "false", stream_dtype == MTL_BF16 ? ", bf16" : "");
        // bf16 dy + the stream's x in (+ the incoming residual gradient), residual gradient out (+ its masked bf16 copy)
        const double e = stream_dtype == MTL_BF16 ? 2 : 4;
        const double bytes = (double)M * d * (2 + e + (dres_in ? e : 0) + e + (dres_out_bf16 ? 2 : 0)) + (double)M * 8;
#define MTL_NORM_BWD(RMSV, XT)                                                                                                             \
    MTL_LAUNCH(kname, bytes, 1, (norm_bwd_kernel<NV, RMSV, WPR, XT>), grid, block, 0, st, (const bf16_t*)dy, ld_dy, (const XT*)x, gamma, stats, \
               (const XT*)dres_in, (XT*)dres_out, (bf16_t*)dres_out_bf16, M, (int)d, group_rows, group_stride, row_offset, stats_physical,  \
               bf16_drop_p, bf16_drop_seed)
        if (stream_dtype == MTL_BF16) { if (rms) MTL_NORM_BWD(true, bf16_t); else MTL_NORM_BWD(false, bf16_t); }
        else { if (rms) MTL_NORM_BWD(true, float); else MTL_NORM_BWD(false, float); }
#undef MTL_NORM_BWD
        MTL_CHECK_LAUNCH();
        return MTL_OK;
    };
    return dispatch_nv<false>(d, go);
}

extern "C" int mtl_norm_bwd(const void* dy, int64_t ld_dy, const float* x, const float* gamma, const float* stats,
                            const float* dres_in, float* dres_out, void* dres_out_bf16, int64_t M, int64_t d, int rms,
                            int64_t group_rows, int64_t group_stride, int64_t row_offset, int stats_physical,
                            float bf16_drop_p, uint32_t bf16_drop_seed, void* stream) {
    return mtl_norm_bwd_t(dy, ld_dy, x, MTL_F32, gamma, stats, dres_in, dres_out, dres_out_bf16, M, d, rms, group_rows, group_stride, row_offset,
                          stats_physical, bf16_drop_p, bf16_drop_seed, stream);
}
