// mtl_stats.hip — the prompt's input statistics on the device (R:models/medtsllm.py:441-495, 530-538):
// per (sample, channel) minimum, maximum, lower median, trend (sign of the summed first differences) and the top-k lags of the
// channel-mean circular autocorrelation. The reference computes them with five ATen reductions + an rFFT round trip and five
// `.tolist()` syncs; here: two launches, one packed [B, ...] fp32 result for ONE device-to-host copy.
//
// The autocorrelation follows the reference's spectral definition, irfft(|rfft(x)|^2) with irfft's DEFAULT length n = 2 * (L/2): for even L
// that is the circular autocorrelation, for odd L it is what the reference computes (a length L - 1 sequence). The power spectrum is
// evaluated by a direct DFT against a sine table in LDS (L is at most a few thousand: O(L^2) per series is microseconds), the inverse
// for lags 0 .. n/2 only and mirrored: the result is symmetric (corr[k] == corr[n - k] exactly), which an FFT round trip reproduces
// only to 1 ulp — so the reference's own order inside a twin pair of lags is round-off noise (tools/lag_twin_noise.py). Ties are
// broken towards the smaller lag.
#include "mtl_common.h"

#include <mutex>

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, int op) {   // op 0 sum, 1 min, 2 max; blockDim = 256
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o, 64);
        v = op == 0 ? v + w : (op == 1 ? fminf(v, w) : fmaxf(v, w));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return op == 0 ? (a + b) + (c + d) : (op == 1 ? fminf(fminf(a, b), fminf(c, d)) : fmaxf(fmaxf(a, b), fmaxf(c, d)));
}

// one workgroup per selected series (b, c). dynamic LDS: xs[L] | cos[L] | sin[L] | red[4]
// out_stats [B, Cs, 4] = (min, max, median, trend 0/1); corr_ws [B, Cs, L/2 + 1] = the power spectrum |X[k]|^2
// TABLE = false (windows beyond 13 650 points: the two tables no longer fit the 160 KB LDS next to the series): the twiddles are evaluated per term —
// the same sincospif of the same reduced index, so both variants produce identical bits.
template <bool TABLE>
__global__ __launch_bounds__(256) void series_stats_kernel(const float* __restrict__ x, float* __restrict__ out_stats, float* __restrict__ corr_ws,
                                                           int L, int C, int c0, int Cs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ct = lds + L;
    float* sn = TABLE ? ct + L : ct;
    float* red = TABLE ? sn + L : lds + L;
    const int bc = blockIdx.x, b = bc / Cs, c = c0 + bc % Cs, tid = threadIdx.x;
    if (TABLE)
        for (int m = tid; m < L; m += 256) sincospif(2.0f * (float)m / (float)L, &sn[m], &ct[m]);
    const float* src = x + (int64_t)b * L * C + c;
    float mn = INFINITY, mx = -INFINITY;
    for (int t = tid; t < L; t += 256) {
        const float v = src[(int64_t)t * C];
        xs[t] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = block_reduce(mn, red, 1);
    mx = block_reduce(mx, red, 2);
    float ds = 0.f;
    for (int t = tid; t + 1 < L; t += 256) ds += xs[t + 1] - xs[t];
    ds = block_reduce(ds, red, 0);
    // lower median = the element of rank (L - 1) / 2: x_i with  #{x_j < x_i} <= k < #{x_j <= x_i}
    const int k = (L - 1) / 2;
    float med = -INFINITY;
    for (int i = tid; i < L; i += 256) {
        const float v = xs[i];
        int lt = 0, le = 0;
        for (int j = 0; j < L; ++j) {
            const float w = xs[j];
            lt += w < v;
            le += w <= v;
        }
        if (lt <= k && k < le) med = v;      // every i that qualifies holds the same value
    }
    med = block_reduce(med, red, 2);
    if (tid == 0) {
        float* o = out_stats + (int64_t)bc * 4;
        o[0] = mn; o[1] = mx; o[2] = med; o[3] = ds > 0.f ? 1.f : 0.f;
    }
    const int nh = L / 2 + 1;
    for (int kk = tid; kk < nh; kk += 256) {       // |X[k]|^2, X[k] = sum_t x[t] e^{-2 pi i k t / L}
        float re = 0.f, im = 0.f;
        int m = 0;
        for (int t = 0; t < L; ++t) {
            float cv, sv;
            if (TABLE) { cv = ct[m]; sv = sn[m]; }
            else sincospif(2.0f * (float)m / (float)L, &sv, &cv);
            re += xs[t] * cv;
            im += xs[t] * sv;
            m += kk;
            if (m >= L) m -= L;
        }
        corr_ws[(int64_t)bc * nh + kk] = re * re + im * im;
    }
}

// one workgroup per sample: channel-mean power spectrum -> inverse real transform of irfft's default length n = 2 (nh - 1) for lags
// 0 .. n/2, mirrored -> top-k by (value desc, lag asc). dynamic LDS: pw[nh] | cos[n] | corr[n]; out_lags [B, n_lags] (floats: exact below 2^24)
template <bool TABLE>
__global__ __launch_bounds__(256) void top_lags_kernel(const float* __restrict__ corr_ws, float* __restrict__ out_lags, int L, int Cs, int n_lags) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nh = L / 2 + 1, n = 2 * (nh - 1);
    float* pw = lds;
    float* ct = pw + nh;
    float* corr = TABLE ? ct + n : ct;
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < nh; k += 256) {
        float s = 0.f;
        for (int c = 0; c < Cs; ++c) s += corr_ws[((int64_t)b * Cs + c) * nh + k];
        pw[k] = s / (float)Cs;       // (irfft is linear: the mean over channels commutes with it)
    }
    if (TABLE)
        for (int m = tid; m < n; m += 256) ct[m] = cospif(2.0f * (float)m / (float)n);
    __syncthreads();
    for (int j = tid; j <= n / 2; j += 256) {
        float acc = 0.f;
        int m = j;
        for (int k = 1; k < nh - 1; ++k) {
            acc += pw[k] * (TABLE ? ct[m] : cospif(2.0f * (float)m / (float)n));
            m += j;
            if (m >= n) m -= n;
        }
        const float v = (pw[0] + 2.0f * acc + ((j & 1) ? -pw[nh - 1] : pw[nh - 1])) / (float)n;
        corr[j] = v;
        if (j > 0 && j < n - j) corr[n - j] = v;
    }
    __syncthreads();
    for (int r = 0; r < n_lags; ++r) {
        float best = -INFINITY;
        int arg = 0x7fffffff;
        for (int k = tid; k < n; k += 256) {
            const float v = corr[k];
            if (v > best || (v == best && k < arg)) { best = v; arg = k; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
        }
        if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = arg; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (bv[w] > bv[0] || (bv[w] == bv[0] && bi[w] < bi[0])) { bv[0] = bv[w]; bi[0] = bi[w]; }
            out_lags[(int64_t)b * n_lags + r] = (float)bi[0];
            corr[bi[0]] = -INFINITY;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t mtl_input_stats_workspace_bytes(int64_t B, int64_t L, int64_t n_channels) {
    return (size_t)B * (size_t)n_channels * (size_t)(L / 2 + 1) * sizeof(float);
}

extern "C" int mtl_input_stats(const float* x, float* out_stats, float* out_lags, void* workspace, size_t workspace_bytes, int64_t B, int64_t L,
                               int64_t C, int64_t channel, int64_t n_lags, void* stream) {
    if (!x || !out_stats || !out_lags || !workspace || B <= 0 || L < 2 || C <= 0 || channel >= C || n_lags <= 0 || n_lags > L) return MTL_ERR_ARG;
    const int c0 = channel < 0 ? 0 : (int)channel, Cs = channel < 0 ? (int)C : 1;
    if (workspace_bytes < mtl_input_stats_workspace_bytes(B, L, Cs)) return MTL_ERR_WORKSPACE;
    // LDS: series + cos + sin tables while they fit the 160 KB (L <= 13 300), the series alone beyond that (twiddles per term, L <= 39 900);
    // the lag kernel: spectrum + cos table + correlation (L <= 15 900), or without the table (L <= 26 600)
    constexpr size_t kLds = 156 * 1024;
    if (n_lags > 2 * (L / 2)) return MTL_ERR_UNSUPPORTED;
    const size_t lds_tab = (size_t)(3 * L + 4) * sizeof(float), lds_plain = (size_t)(L + 4) * sizeof(float);
    const size_t lag_tab = (size_t)(L / 2 + 1 + 4 * (L / 2)) * sizeof(float), lag_plain = (size_t)(L / 2 + 1 + 2 * (L / 2)) * sizeof(float);
    if (lds_plain > kLds || lag_plain > kLds) return MTL_ERR_UNSUPPORTED;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)series_stats_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        (void)hipFuncSetAttribute((const void*)series_stats_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        (void)hipFuncSetAttribute((const void*)top_lags_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        (void)hipFuncSetAttribute((const void*)top_lags_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    });
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (lds_tab <= kLds)
        hipLaunchKernelGGL(series_stats_kernel<true>, dim3((unsigned)(B * Cs)), dim3(256), lds_tab, st, x, out_stats, reinterpret_cast<float*>(workspace), (int)L, (int)C, c0, Cs);
    else
        hipLaunchKernelGGL(series_stats_kernel<false>, dim3((unsigned)(B * Cs)), dim3(256), lds_plain, st, x, out_stats, reinterpret_cast<float*>(workspace), (int)L, (int)C, c0, Cs);
    if (lag_tab <= kLds)
        hipLaunchKernelGGL(top_lags_kernel<true>, dim3((unsigned)B), dim3(256), lag_tab, st, reinterpret_cast<const float*>(workspace), out_lags, (int)L, Cs, (int)n_lags);
    else
        hipLaunchKernelGGL(top_lags_kernel<false>, dim3((unsigned)B), dim3(256), lag_plain, st, reinterpret_cast<const float*>(workspace), out_lags, (int)L, Cs, (int)n_lags);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
