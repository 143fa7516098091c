// mtl_tokenizer.hip — fused patch tokeniser: RevIN statistics + normalise + replicate-pad/unfold index map +
// k=3 circular token conv + (concat) relayout, one workgroup per (sample, channel) series.
//
// HBM-bound and tiny (B*L*C*4 B in, B*C*P*d_patch*2 B out); the point of the fusion is ONE launch instead of
// the reference's ~10 ATen kernels, with the normalised series, the conv weight and the patch map all in LDS.
#include "mtl_common.h"

#include <mutex>

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
                                                           int d_patch, int P, int64_t ld_out, int concat, float eps, uint32_t drop_thr, uint32_t drop_seed) {
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
        if (drop_thr) acc = drop_keep(drop_base(drop_seed, 0u), (uint32_t)(row0 + p), (uint32_t)(col0 + o), drop_thr) ? acc * drop_scale_of(drop_thr) : 0.f;
        out[(row0 + p) * ld_out + col0 + o] = f32_to_bf16(acc);
    }
    // zero the K-padding columns of this series' rows (GEMM operands are padded to K % 64 == 0)
    const int used = concat ? C * d_patch : d_patch;
    const int pad = (int)ld_out - used;
    if (pad > 0 && (!concat || c == 0)) {
        for (int e = tid; e < P * pad; e += 256) out[(row0 + e / pad) * ld_out + used + e % pad] = 0;
    }
}

// Fast path (patch_len == 16, stride % 4 == 0, 256 % d_patch == 0): every thread owns ONE output channel with its
// 48 conv weights in registers and walks the patches; the patch windows are contiguous 64-B slices of the replicate-padded
// normalised series in LDS (exactly the reference's pad + unfold), read as 16-B broadcasts.
// dynamic LDS: xpad[L + stride] | red[4]
__global__ __launch_bounds__(256) void tokenize_fwd_fast_kernel(const float* __restrict__ x, const float* __restrict__ conv_w,
                                                                bf16_t* __restrict__ out, float* __restrict__ mean_out,
                                                                float* __restrict__ stdev_out, int L, int C, int stride, int d_patch,
                                                                int P, int64_t ld_out, int concat, float eps, uint32_t drop_thr, uint32_t drop_seed) {
    constexpr int PL = 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xp = lds;
    float* red = lds + ((L + stride + 3) & ~3);
    const int bc = blockIdx.x, b = bc / C, c = bc % C;
    const int tid = threadIdx.x;
    const float* xs = x + (int64_t)b * L * C + c;
    float s = 0.f;
    for (int t = tid; t < L; t += 256) {
        const float v = xs[(int64_t)t * C];
        xp[t] = v;
        s += v;
    }
    const int o = tid % d_patch, pg = tid / d_patch, npg = 256 / d_patch;
    float w[3][PL];
#pragma unroll
    for (int j = 0; j < PL; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k][j] = conv_w[(o * PL + j) * 3 + k];
    const float mean = block_sum(s, red) / (float)L;
    float q = 0.f;
    for (int t = tid; t < L; t += 256) {
        const float dlt = xp[t] - mean;
        q += dlt * dlt;
    }
    const float stdev = sqrtf(block_sum(q, red) / (float)L + eps);
    if (tid == 0) {
        mean_out[bc] = mean;
        stdev_out[bc] = stdev;
    }
    for (int t = tid; t < L; t += 256) xp[t] = (xp[t] - mean) / stdev;
    __syncthreads();
    const float last = xp[L - 1];
    for (int t = L + tid; t < L + stride; t += 256) xp[t] = last;   // ReplicationPad1d((0, stride))
    __syncthreads();
    const int64_t row0 = concat ? (int64_t)b * P : (int64_t)bc * P;
    const int col0 = concat ? c * d_patch : 0;
    for (int p = pg; p < P; p += npg) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int pp = p + k - 1;
            pp = pp < 0 ? pp + P : (pp >= P ? pp - P : pp);
            const float4* win = reinterpret_cast<const float4*>(xp + pp * stride);
#pragma unroll
            for (int j4 = 0; j4 < PL / 4; ++j4) {
                const float4 v = win[j4];
                acc += w[k][j4 * 4] * v.x + w[k][j4 * 4 + 1] * v.y + w[k][j4 * 4 + 2] * v.z + w[k][j4 * 4 + 3] * v.w;
            }
        }
        // PatchEmbedding's dropout (R:models/layers/embed.py:197), mask indexed by the element's position in `out`
        if (drop_thr) acc = drop_keep(drop_base(drop_seed, 0u), (uint32_t)(row0 + p), (uint32_t)(col0 + o), drop_thr) ? acc * drop_scale_of(drop_thr) : 0.f;
        out[(row0 + p) * ld_out + col0 + o] = f32_to_bf16(acc);
    }
    const int used = concat ? C * d_patch : d_patch;
    const int pad = (int)ld_out - used;
    if (pad > 0 && (!concat || c == 0)) {
        for (int e = tid; e < P * pad; e += 256) out[(row0 + e / pad) * ld_out + used + e % pad] = 0;
    }
}

// dW partial for one series: partial[bc][o][j][k] = sum_p dout[p][o] * patch[(p + k - 1) mod P][j]
// dynamic LDS: xn[L] | dout tile [P][d_patch] fp32
__global__ __launch_bounds__(256) void tokenize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean_in,
                                                           const float* __restrict__ stdev_in, const bf16_t* __restrict__ dout,
                                                           float* __restrict__ partial, int L, int C, int patch_len, int stride,
                                                           int d_patch, int P, int64_t ld_out, int concat, uint32_t drop_thr, uint32_t drop_seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xn = lds;
    const int bc = blockIdx.x, b = bc / C, c = bc % C;
    const int tid = threadIdx.x;
    const float* xs = x + (int64_t)b * L * C + c;
    const float mean = mean_in[bc], stdev = stdev_in[bc];
    for (int t = tid; t < L; t += 256) xn[t] = (xs[(int64_t)t * C] - mean) / stdev;
    float* dt = lds + L;
    const int64_t row0 = concat ? (int64_t)b * P : (int64_t)bc * P;
    const int col0 = concat ? c * d_patch : 0;
    for (int e = tid; e < P * d_patch; e += 256) {
        float v = bf16_to_f32(dout[(row0 + e / d_patch) * ld_out + col0 + e % d_patch]);
        if (drop_thr) v = drop_keep(drop_base(drop_seed, 0u), (uint32_t)(row0 + e / d_patch), (uint32_t)(col0 + e % d_patch), drop_thr) ? v * drop_scale_of(drop_thr) : 0.f;
        dt[e] = v;      // gradient through the forward's dropout mask
    }
    __syncthreads();
    const int nw = d_patch * patch_len * 3;
    // consecutive threads -> consecutive output channels o (conflict-free dt reads, broadcast xn reads)
    for (int e = tid; e < nw; e += 256) {
        const int o = e % d_patch, jk = e / d_patch, j = jk / 3, k = jk % 3;
        float acc = 0.f;
        for (int p = 0; p < P; ++p) {
            int pp = p + k - 1;
            pp = pp < 0 ? pp + P : (pp >= P ? pp - P : pp);
            acc += dt[p * d_patch + o] * xn[patch_src_index(pp, j, L, stride)];
        }
        partial[(int64_t)bc * nw + (o * patch_len + j) * 3 + k] = acc;
    }
}

// Fast path of the weight gradient (patch_len == 16, stride % 4 == 0, 256 % d_patch == 0), the mirror of tokenize_fwd_fast_kernel: a
// thread owns ONE output channel and every (256 / d_patch)-th patch, keeps its 48 partial sums in registers and reads the patch
// windows as 16-B broadcasts of the replicate-padded normalised series (the generic kernel reads LDS twice per multiply-add:
// 53 us for 0.15 GFLOP). The patch groups are then summed through LDS in a fixed order.
// dynamic LDS: xpad[L + stride] | dout tile [P][d_patch] fp32 (re-used as [npg][48 d_patch] for the group sums)
__global__ __launch_bounds__(256) void tokenize_bwd_fast_kernel(const float* __restrict__ x, const float* __restrict__ mean_in,
                                                                const float* __restrict__ stdev_in, const bf16_t* __restrict__ dout,
                                                                float* __restrict__ partial, int L, int C, int stride, int d_patch, int P,
                                                                int64_t ld_out, int concat, uint32_t drop_thr, uint32_t drop_seed) {
    constexpr int PL = 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xp = lds;
    float* dt = lds + ((L + stride + 3) & ~3);
    const int bc = blockIdx.x, b = bc / C, c = bc % C;
    const int tid = threadIdx.x;
    const float* xs = x + (int64_t)b * L * C + c;
    const float mean = mean_in[bc], stdev = stdev_in[bc];
    for (int t = tid; t < L + stride; t += 256) xp[t] = (xs[(int64_t)(t < L - 1 ? t : L - 1) * C] - mean) / stdev;   // ReplicationPad1d((0, stride))
    const int64_t row0 = concat ? (int64_t)b * P : (int64_t)bc * P;
    const int col0 = concat ? c * d_patch : 0;
    const uint32_t dbase = drop_base(drop_seed, 0u);
    for (int e = tid; e < P * d_patch; e += 256) {
        float v = bf16_to_f32(dout[(row0 + e / d_patch) * ld_out + col0 + e % d_patch]);
        if (drop_thr) v = drop_keep(dbase, (uint32_t)(row0 + e / d_patch), (uint32_t)(col0 + e % d_patch), drop_thr) ? v * drop_scale_of(drop_thr) : 0.f;
        dt[e] = v;      // gradient through the forward's dropout mask
    }
    __syncthreads();
    const int o = tid % d_patch, pg = tid / d_patch, npg = 256 / d_patch;
    float acc[3][PL];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < PL; ++j) acc[k][j] = 0.f;
    for (int p = pg; p < P; p += npg) {
        const float gv = dt[p * d_patch + o];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int pp = p + k - 1;
            pp = pp < 0 ? pp + P : (pp >= P ? pp - P : pp);
            const float4* win = reinterpret_cast<const float4*>(xp + pp * stride);
#pragma unroll
            for (int j4 = 0; j4 < PL / 4; ++j4) {
                const float4 v = win[j4];
                acc[k][j4 * 4] += gv * v.x; acc[k][j4 * 4 + 1] += gv * v.y; acc[k][j4 * 4 + 2] += gv * v.z; acc[k][j4 * 4 + 3] += gv * v.w;
            }
        }
    }
    __syncthreads();                      // everyone is done with the dout tile: its space takes the group sums
    float* red = dt;                      // [pg][(j * 3 + k)][o]: consecutive threads -> consecutive o (conflict-free)
    const int nw = d_patch * PL * 3;
#pragma unroll
    for (int j = 0; j < PL; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) red[pg * nw + (j * 3 + k) * d_patch + o] = acc[k][j];
    __syncthreads();
    for (int e = tid; e < nw; e += 256) {
        float sum = 0.f;
        for (int g2 = 0; g2 < npg; ++g2) sum += red[g2 * nw + e];
        const int oo = e % d_patch, jk = e / d_patch;
        partial[(int64_t)bc * nw + oo * PL * 3 + jk] = sum;       // dw layout [o][j][k]
    }
}

// deterministic two-stage reduction over the series: 16 outputs per block, 16 series lanes per output
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n_series, int nw) {
    __shared__ float part[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + tx;
    float s = 0.f;
    if (e < nw)
        for (int i = ty; i < n_series; i += 16) s += partial[(int64_t)i * nw + e];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < nw) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][tx];
        dw[e] = t;
    }
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
                                      float eps, float drop_p, uint32_t drop_seed, void* stream) {
    if (!x || !conv_w || !out || !mean || !stdev || B <= 0 || C <= 0 || L < patch_len || patch_len <= 0 || stride <= 0 || d_patch <= 0)
        return MTL_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return MTL_ERR_ARG;
    const uint32_t drop_thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    const int P = (int)((L + stride - patch_len) / stride + 1);
    if (ld_out < (concat ? C * d_patch : d_patch)) return MTL_ERR_ARG;
    if (patch_len == 16 && stride % 4 == 0 && d_patch <= 256 && 256 % d_patch == 0) {
        const size_t fast_bytes = (size_t)(((L + stride + 3) & ~3) + 4) * sizeof(float);
        if (fast_bytes <= 64 * 1024) {
            hipLaunchKernelGGL(tokenize_fwd_fast_kernel, dim3((unsigned)(B * C)), dim3(256), fast_bytes, (hipStream_t)stream, x, conv_w,
                               (bf16_t*)out, mean, stdev, (int)L, (int)C, (int)stride, (int)d_patch, P, ld_out, concat, eps, drop_thr, drop_seed);
            MTL_CHECK_LAUNCH();
            return MTL_OK;
        }
    }
    const size_t lds_bytes = (size_t)(L + d_patch * patch_len * 3 + 4) * sizeof(float);
    if (lds_bytes > 64 * 1024) return MTL_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tokenize_fwd_kernel, dim3((unsigned)(B * C)), dim3(256), lds_bytes, (hipStream_t)stream, x, conv_w, (bf16_t*)out,
                       mean, stdev, (int)L, (int)C, (int)patch_len, (int)stride, (int)d_patch, P, ld_out, concat, eps, drop_thr, drop_seed);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_patch_tokenize_bwd(const float* x, const float* mean, const float* stdev, const void* dout, float* partial, float* dw,
                                      int64_t B, int64_t L, int64_t C, int64_t patch_len, int64_t stride, int64_t d_patch,
                                      int64_t ld_out, int concat, float drop_p, uint32_t drop_seed, void* stream) {
    if (!x || !mean || !stdev || !dout || !partial || !dw || B <= 0 || C <= 0 || L < patch_len) return MTL_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return MTL_ERR_ARG;
    const uint32_t drop_thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    const int P = (int)((L + stride - patch_len) / stride + 1);
    const size_t lds_bytes = (size_t)(L + P * d_patch) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int nw = (int)(d_patch * patch_len * 3);
    const size_t fast_tile = (size_t)P * d_patch > (size_t)nw * (256 / (d_patch > 0 && d_patch <= 256 ? d_patch : 256)) ? (size_t)P * d_patch : (size_t)nw * (256 / (d_patch <= 256 ? d_patch : 256));
    const size_t fast_bytes = ((size_t)((L + stride + 3) & ~3) + fast_tile) * sizeof(float);
    if (patch_len == 16 && stride % 4 == 0 && d_patch <= 256 && 256 % d_patch == 0 && fast_bytes <= 160 * 1024) {
        static std::once_flag once;
        std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)tokenize_bwd_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        hipLaunchKernelGGL(tokenize_bwd_fast_kernel, dim3((unsigned)(B * C)), dim3(256), fast_bytes, st, x, mean, stdev, (const bf16_t*)dout, partial,
                           (int)L, (int)C, (int)stride, (int)d_patch, P, ld_out, concat, drop_thr, drop_seed);
    } else {
        if (lds_bytes > 64 * 1024) return MTL_ERR_UNSUPPORTED;      // (the generic kernel's static budget; long windows take the fast kernel above)
        hipLaunchKernelGGL(tokenize_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), lds_bytes, st, x, mean, stdev, (const bf16_t*)dout, partial,
                           (int)L, (int)C, (int)patch_len, (int)stride, (int)d_patch, P, ld_out, concat, drop_thr, drop_seed);
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((nw + 15) / 16), dim3(256), 0, st, partial, dw, (int)(B * C), nw);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}
