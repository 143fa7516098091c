// mtl_tokenizer.hip — fused patch tokeniser: RevIN statistics + normalise + replicate-pad/unfold index map +
// k=3 circular token conv + (concat) relayout, one workgroup per (sample, channel) series.
//
// HBM-bound and tiny (B*L*C*4 B in, B*C*P*d_patch*2 B out); the point of the fusion is ONE launch instead of
// the reference's ~10 ATen kernels, with the normalised series, the conv weight and the patch map all in LDS.
#include "mtl_common.h"

namespace {

// THE patch index map (bit-exact parity target): source time index of element j of patch p.
// ReplicationPad1d((0, stride)) then unfold(size=patch_len, step=stride)  (R:models/layers/embed.py:160-163,190)
__device__ __forceinline__ int patch_src_index(int p, int j, int L, int stride) {
    const int t = p * stride + j;
    return t < L - 1 ? t : L - 1;
}

__global__ void patch_index_map_kernel(int32_t* __restrict__ idx, int P, int L, int patch_len, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P * patch_len) idx[i] = patch_src_index(i / patch_len, i % patch_len, L, stride);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// dynamic LDS: xn[L] | w[d_patch*patch_len*3] | red[4]
__global__ __launch_bounds__(256) void tokenize_fwd_kernel(const float* __restrict__ x, const float* __restrict__ conv_w,
                                                           bf16_t* __restrict__ out, float* __restrict__ mean_out,
                                                           float* __restrict__ stdev_out, int L, int C, int patch_len, int stride,
                                                           int d_patch, int P, int64_t ld_out, int concat, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xn = lds;
    float* w = lds + L;
    float* red = w + d_patch * patch_len * 3;
    const int bc = blockIdx.x, b = bc / C, c = bc % C;
    const int tid = threadIdx.x;
    const float* xs = x + (int64_t)b * L * C + c;

    float s = 0.f;
    for (int t = tid; t < L; t += 256) {
        const float v = xs[(int64_t)t * C];
        xn[t] = v;
        s += v;
    }
    for (int i = tid; i < d_patch * patch_len * 3; i += 256) w[i] = conv_w[i];
    const float mean = block_sum(s, red) / (float)L;
    float q = 0.f;
    for (int t = tid; t < L; t += 256) {
        const float dlt = xn[t] - mean;
        q += dlt * dlt;
    }
    const float stdev = sqrtf(block_sum(q, red) / (float)L + eps);  // sqrt(var_biased + eps)  R:RevIN.py:43
    if (tid == 0) {
        mean_out[bc] = mean;
        stdev_out[bc] = stdev;
    }
    for (int t = tid; t < L; t += 256) xn[t] = (xn[t] - mean) / stdev;  // (x - mean) / stdev  R:RevIN.py:52-53
    __syncthreads();

    // out[p][o] = sum_k sum_j W[o][j][k] * patch[(p + k - 1) mod P][j]
    const int64_t row0 = concat ? (int64_t)b * P : (int64_t)bc * P;
    const int col0 = concat ? c * d_patch : 0;
    const int n_out = P * d_patch;
    for (int e = tid; e < n_out; e += 256) {
        const int p = e / d_patch, o = e % d_patch;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int pp = p + k - 1;
            pp = pp < 0 ? pp + P : (pp >= P ? pp - P : pp);
            const float* wk = w + o * patch_len * 3 + k;
            for (int j = 0; j < patch_len; ++j) acc += wk[j * 3] * xn[patch_src_index(pp, j, L, stride)];
        }
        out[(row0 + p) * ld_out + col0 + o] = f32_to_bf16(acc);
    }
    // zero the K-padding columns of this series' rows (GEMM operands are padded to K % 64 == 0)
    const int used = concat ? C * d_patch : d_patch;
    const int pad = (int)ld_out - used;
    if (pad > 0 && (!concat || c == 0)) {
        for (int e = tid; e < P * pad; e += 256) out[(row0 + e / pad) * ld_out + used + e % pad] = 0;
    }
}

// dW partial for one series: partial[bc][o][j][k] = sum_p dout[p][o] * patch[(p + k - 1) mod P][j]
// dynamic LDS: xn[L] | red[4]
__global__ __launch_bounds__(256) void tokenize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean_in,
                                                           const float* __restrict__ stdev_in, const bf16_t* __restrict__ dout,
                                                           float* __restrict__ partial, int L, int C, int patch_len, int stride,
                                                           int d_patch, int P, int64_t ld_out, int concat) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xn = lds;
    const int bc = blockIdx.x, b = bc / C, c = bc % C;
    const int tid = threadIdx.x;
    const float* xs = x + (int64_t)b * L * C + c;
    const float mean = mean_in[bc], stdev = stdev_in[bc];
    for (int t = tid; t < L; t += 256) xn[t] = (xs[(int64_t)t * C] - mean) / stdev;
    __syncthreads();
    const int64_t row0 = concat ? (int64_t)b * P : (int64_t)bc * P;
    const int col0 = concat ? c * d_patch : 0;
    const int nw = d_patch * patch_len * 3;
    for (int e = tid; e < nw; e += 256) {
        const int o = e / (patch_len * 3), j = (e / 3) % patch_len, k = e % 3;
        float acc = 0.f;
        for (int p = 0; p < P; ++p) {
            int pp = p + k - 1;
            pp = pp < 0 ? pp + P : (pp >= P ? pp - P : pp);
            acc += bf16_to_f32(dout[(row0 + p) * ld_out + col0 + o]) * xn[patch_src_index(pp, j, L, stride)];
        }
        partial[(int64_t)bc * nw + e] = acc;
    }
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n_series, int nw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nw) return;
    float s = 0.f;
    for (int i = 0; i < n_series; ++i) s += partial[(int64_t)i * nw + e];
    dw[e] = s;
}

}  // namespace

extern "C" int mtl_patch_index_map(int32_t* idx, int64_t L, int64_t patch_len, int64_t stride, void* stream) {
    if (!idx || L < patch_len || patch_len <= 0 || stride <= 0) return MTL_ERR_ARG;
    const int P = (int)((L + stride - patch_len) / stride + 1);
    const int n = P * (int)patch_len;
    hipLaunchKernelGGL(patch_index_map_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, P, (int)L, (int)patch_len,
                       (int)stride);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_patch_tokenize_fwd(const float* x, const float* conv_w, void* out, float* mean, float* stdev, int64_t B, int64_t L,
                                      int64_t C, int64_t patch_len, int64_t stride, int64_t d_patch, int64_t ld_out, int concat,
                                      float eps, void* stream) {
    if (!x || !conv_w || !out || !mean || !stdev || B <= 0 || C <= 0 || L < patch_len || patch_len <= 0 || stride <= 0 || d_patch <= 0)
        return MTL_ERR_ARG;
    const int P = (int)((L + stride - patch_len) / stride + 1);
    if (ld_out < (concat ? C * d_patch : d_patch)) return MTL_ERR_ARG;
    const size_t lds_bytes = (size_t)(L + d_patch * patch_len * 3 + 4) * sizeof(float);
    if (lds_bytes > 64 * 1024) return MTL_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tokenize_fwd_kernel, dim3((unsigned)(B * C)), dim3(256), lds_bytes, (hipStream_t)stream, x, conv_w, (bf16_t*)out,
                       mean, stdev, (int)L, (int)C, (int)patch_len, (int)stride, (int)d_patch, P, ld_out, concat, eps);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_patch_tokenize_bwd(const float* x, const float* mean, const float* stdev, const void* dout, float* partial, float* dw,
                                      int64_t B, int64_t L, int64_t C, int64_t patch_len, int64_t stride, int64_t d_patch,
                                      int64_t ld_out, int concat, void* stream) {
    if (!x || !mean || !stdev || !dout || !partial || !dw || B <= 0 || C <= 0 || L < patch_len) return MTL_ERR_ARG;
    const int P = (int)((L + stride - patch_len) / stride + 1);
    const size_t lds_bytes = (size_t)(L + 4) * sizeof(float);
    if (lds_bytes > 64 * 1024) return MTL_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tokenize_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), lds_bytes, st, x, mean, stdev, (const bf16_t*)dout, partial,
                       (int)L, (int)C, (int)patch_len, (int)stride, (int)d_patch, P, ld_out, concat);
    const int nw = (int)(d_patch * patch_len * 3);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((nw + 255) / 256), dim3(256), 0, st, partial, dw, (int)(B * C), nw);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
