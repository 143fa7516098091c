// mtl_elementwise.hip — HBM-bound layout / cast / elementwise kernels (all 8-16 B per lane, coalesced).
#include "mtl_common.h"

namespace {

// ---------------------------------------------------------------- f32 -> bf16 cast with zero padding (+ optional transpose)
// 64x64 tile per block; element (r, c) = (r < R && c < Cc) ? src[r][c] : 0.
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int64_t ld_src, bf16_t* __restrict__ dst,
                                                       int64_t ld_dst, bf16_t* __restrict__ dst_t, int64_t ld_dst_t, int64_t R,
                                                       int64_t Cc) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty * 16 + i;
        const int64_t r = r0 + rr, c = c0 + tx;
        const float v = (r < R && c < Cc) ? src[r * ld_src + c] : 0.f;
        const bf16_t b = f32_to_bf16(v);
        tile[rr][tx] = b;
        if (dst && r < R && c < ld_dst) dst[r * ld_dst + c] = b;
    }
    if (!dst_t) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty * 16 + i;
        const int64_t c = c0 + cc, r = r0 + tx;
        if (c < Cc && r < ld_dst_t) dst_t[c * ld_dst_t + r] = tile[tx][cc];
    }
}

template <bool COLSUM>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int64_t ld_src, bf16_t* __restrict__ dst,
                                                             int64_t ld_dst, int64_t R, int64_t Cc, float* __restrict__ colsum) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty * 16 + i;
        const int64_t r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < R && c < Cc) ? src[r * ld_src + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty * 16 + i;
        const int64_t c = c0 + cc, r = r0 + tx;
        if (c < Cc && r < ld_dst) dst[c * ld_dst + r] = tile[tx][cc];
    }
    if constexpr (COLSUM) {     // column sums of src from the staged tile (rows >= R are zero): the bias gradient of a Linear's dY
        if (ty == 0 && c0 + tx < Cc) {
            float sacc = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 64; ++rr) sacc += bf16_to_f32(tile[rr][tx]);
            atomicAdd(colsum + c0 + tx, sacc);
        }
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            u32x2 pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
            *reinterpret_cast<u32x2*>(dst + i) = pk;
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
        }
    }
}

__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(src + i);
            *reinterpret_cast<float4*>(dst + i) = make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                                                              __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u));
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = bf16_to_f32(src[j]);
        }
    }
}

// column sums: grid (ceil(Cc/64), row chunks of 256); 256 threads = 64 cols x 4 row lanes; fp32 atomics into dst (pre-zeroed)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t R,
                                                     int64_t Cc) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    const int64_t r0 = (int64_t)blockIdx.y * 256;
    float s = 0.f;
    if (c < Cc) {
        const int64_t rend = (r0 + 256 < R) ? r0 + 256 : R;
        for (int64_t r = r0 + ty; r < rend; r += 4) s += bf16_to_f32(src[r * ld_src + c]);
    }
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < Cc) atomicAdd(dst + c, (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]));
}

// ---------------------------------------------------------------- RoPE (half-split), in place on q|k heads
__device__ __forceinline__ int64_t remap_row(int64_t m, int64_t group_rows, int64_t group_stride, int64_t off) {
    if (group_rows == 0) return m;
    return (m / group_rows) * group_stride + off + (m % group_rows);
}

__global__ void rope_kernel(bf16_t* __restrict__ qkv, int64_t ld, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                            int64_t M, int64_t T, int n_heads, int D, int inverse, int64_t gr, int64_t gs, int64_t ro) {
    const int half = D >> 1;
    const int pairs_per_row = n_heads * (half >> 1);  // two rotation pairs (i, i+1) per thread -> 4-byte accesses
    const int64_t total = M * pairs_per_row;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = remap_row(idx / pairs_per_row, gr, gs, ro);
        const int rem = (int)(idx % pairs_per_row);
        const int h = rem / (half >> 1), i = (rem % (half >> 1)) * 2;
        const int64_t pos = m % T;
        bf16_t* base = qkv + m * ld + (int64_t)h * D;
        const uint32_t lo = *reinterpret_cast<const uint32_t*>(base + i);
        const uint32_t hi = *reinterpret_cast<const uint32_t*>(base + half + i);
        const float x1a = __uint_as_float(lo << 16), x1b = __uint_as_float(lo & 0xffff0000u);
        const float x2a = __uint_as_float(hi << 16), x2b = __uint_as_float(hi & 0xffff0000u);
        const float ca = cos_t[pos * D + i], cb = cos_t[pos * D + i + 1];
        float sa = sin_t[pos * D + i], sb = sin_t[pos * D + i + 1];
        if (inverse) { sa = -sa; sb = -sb; }
        // y1 = x1*c - x2*s ; y2 = x2*c + x1*s   (q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1))
        float y1a, y2a, y1b, y2b;
        rope_pair(x1a, x2a, ca, sa, y1a, y2a);
        rope_pair(x1b, x2b, cb, sb, y1b, y2b);
        *reinterpret_cast<uint32_t*>(base + i) = pack_bf16x2(y1a, y1b);
        *reinterpret_cast<uint32_t*>(base + half + i) = pack_bf16x2(y2a, y2b);
    }
}

// the same rotation with 16-byte accesses: a thread owns 8 consecutive columns i .. i+7 of a head's low half and the matching 8 of its high half
// (two 16-B loads + two 16-B stores of qkv, 2 x 32 B of cos / sin); D % 16 == 0, 16-B aligned rows. Same rope_pair arithmetic: bit-identical.
// (the 4-byte form above moved the Llama-2-7B forward pass' 134 MB per layer at 3.4 TB/s)
__global__ __launch_bounds__(256) void rope_kernel_v8(bf16_t* __restrict__ qkv, int64_t ld, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                      int64_t M, int64_t T, int n_heads, int D, int inverse, int64_t gr, int64_t gs, int64_t ro) {
    const int half = D >> 1, cpr = half >> 3;      // 8-column chunks per head half
    const int per_row = n_heads * cpr;
    const int64_t total = M * per_row;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = remap_row(idx / per_row, gr, gs, ro);
        const int rem = (int)(idx % per_row);
        const int h = rem / cpr, i = (rem % cpr) * 8;
        const int64_t pos = m % T;
        bf16_t* base = qkv + m * ld + (int64_t)h * D;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(base + i);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(base + half + i);
        const float4 c0 = *reinterpret_cast<const float4*>(cos_t + pos * D + i), c1 = *reinterpret_cast<const float4*>(cos_t + pos * D + i + 4);
        float4 s0 = *reinterpret_cast<const float4*>(sin_t + pos * D + i), s1 = *reinterpret_cast<const float4*>(sin_t + pos * D + i + 4);
        const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        u32x4 ylo, yhi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x1a = __uint_as_float(lo[j] << 16), x1b = __uint_as_float(lo[j] & 0xffff0000u);
            const float x2a = __uint_as_float(hi[j] << 16), x2b = __uint_as_float(hi[j] & 0xffff0000u);
            const float sa = inverse ? -sn[2 * j] : sn[2 * j], sb = inverse ? -sn[2 * j + 1] : sn[2 * j + 1];
            float y1a, y2a, y1b, y2b;
            rope_pair(x1a, x2a, c[2 * j], sa, y1a, y2a);
            rope_pair(x1b, x2b, c[2 * j + 1], sb, y1b, y2b);
            ylo[j] = pack_bf16x2(y1a, y1b);
            yhi[j] = pack_bf16x2(y2a, y2b);
        }
        *reinterpret_cast<u32x4*>(base + i) = ylo;
        *reinterpret_cast<u32x4*>(base + half + i) = yhi;
    }
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ h, int64_t M, int64_t F) {
    const int64_t total = M * (F >> 1);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / (F >> 1), f = (idx % (F >> 1)) * 2;
        const uint32_t g = *reinterpret_cast<const uint32_t*>(gu + m * 2 * F + f);
        const uint32_t u = *reinterpret_cast<const uint32_t*>(gu + m * 2 * F + F + f);
        const float g0 = __uint_as_float(g << 16), g1 = __uint_as_float(g & 0xffff0000u);
        const float u0 = __uint_as_float(u << 16), u1 = __uint_as_float(u & 0xffff0000u);
        // silu output is a bf16 tensor in the reference (act_fn on a bf16 Linear output), then multiplied by up
        const float s0 = bf16_to_f32(f32_to_bf16(g0 * sigmoid_f(g0))), s1 = bf16_to_f32(f32_to_bf16(g1 * sigmoid_f(g1)));
        *reinterpret_cast<uint32_t*>(h + m * F + f) = pack_bf16x2(s0 * u0, s1 * u1);
    }
}

// gu rows may be gathered (saved activations keep their physical layout); dh / dgu are compact
template <bool INTERLEAVED>
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dh, bf16_t* __restrict__ dgu, int64_t M,
                                  int64_t F, int64_t gr, int64_t gs, int64_t ro) {
    const int64_t total = M * (F >> 1);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / (F >> 1), f = (idx % (F >> 1)) * 2;
        const int64_t pm = remap_row(m, gr, gs, ro);
        float gv[2], uv[2];
        if (INTERLEAVED) {          // columns (2f, 2f+1, 2f+2, 2f+3) = (g_f, u_f, g_f+1, u_f+1): one 8-byte access
            const u32x2 q = *reinterpret_cast<const u32x2*>(gu + pm * 2 * F + 2 * f);
            gv[0] = __uint_as_float(q[0] << 16); uv[0] = __uint_as_float(q[0] & 0xffff0000u);
            gv[1] = __uint_as_float(q[1] << 16); uv[1] = __uint_as_float(q[1] & 0xffff0000u);
        } else {
            const uint32_t g = *reinterpret_cast<const uint32_t*>(gu + pm * 2 * F + f);
            const uint32_t u = *reinterpret_cast<const uint32_t*>(gu + pm * 2 * F + F + f);
            gv[0] = __uint_as_float(g << 16); gv[1] = __uint_as_float(g & 0xffff0000u);
            uv[0] = __uint_as_float(u << 16); uv[1] = __uint_as_float(u & 0xffff0000u);
        }
        const uint32_t d = *reinterpret_cast<const uint32_t*>(dh + m * F + f);
        float dv[2] = {__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
        float dg[2], du[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float sg = sigmoid_f(gv[e]);
            du[e] = dv[e] * gv[e] * sg;
            dg[e] = dv[e] * uv[e] * sg * (1.0f + gv[e] * (1.0f - sg));
        }
        if (INTERLEAVED) {
            const u32x2 q = {pack_bf16x2(dg[0], du[0]), pack_bf16x2(dg[1], du[1])};
            *reinterpret_cast<u32x2*>(dgu + m * 2 * F + 2 * f) = q;
        } else {
            *reinterpret_cast<uint32_t*>(dgu + m * 2 * F + f) = pack_bf16x2(dg[0], dg[1]);
            *reinterpret_cast<uint32_t*>(dgu + m * 2 * F + F + f) = pack_bf16x2(du[0], du[1]);
        }
    }
}

// ---------------------------------------------------------------- LLM input assembly: one block per (b, t) row
// OUT_BF: the residual stream is bf16 (the reference's dtype "bf16": embedding tables, their sum and the dropped result are bf16 tensors there)
__device__ __forceinline__ float4 rbf4(float4 v) {
    const u32x2 k = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    return make_float4(__uint_as_float(k[0] << 16), __uint_as_float(k[0] & 0xffff0000u), __uint_as_float(k[1] << 16), __uint_as_float(k[1] & 0xffff0000u));
}
template <bool OUT_BF>
__device__ __forceinline__ void st_stream4(void* base, int64_t idx, float4 v) {
    if constexpr (OUT_BF) *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(base) + idx) = (u32x2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
}
template <bool OUT_BF>
__global__ __launch_bounds__(256) void assemble_kernel(const int32_t* __restrict__ ids, int64_t ids_B, const float* __restrict__ embed,
                                                       const bf16_t* __restrict__ x_tok, const float* __restrict__ wpe,
                                                       void* __restrict__ h0, int64_t n_tok, int64_t P, int64_t d, uint32_t drop_thr,
                                                       uint32_t drop_seed) {
    const int64_t T = n_tok + P;
    const int64_t row = blockIdx.x, b = row / T, t = row % T;
    const float* pe = wpe ? wpe + t * d : nullptr;
    // GPT-2 embd_pdrop on inputs_embeds + wpe (HF:models/gpt2/modeling_gpt2.py:579), mask of (seed, flattened row, column) = mtl_dropout_f32's
    const uint32_t dbase = drop_base(drop_seed, 0u);
    const float dscale = drop_scale_of(drop_thr);
    auto drop4 = [&](float4& v, int64_t c) {
        if (drop_thr) {
            const uint2 w = drop_quad(dbase, (uint32_t)row, (uint32_t)c >> 2);
            v.x = (w.x & 0xffffu) >= drop_thr ? v.x * dscale : 0.f; v.y = (w.x >> 16) >= drop_thr ? v.y * dscale : 0.f;
            v.z = (w.y & 0xffffu) >= drop_thr ? v.z * dscale : 0.f; v.w = (w.y >> 16) >= drop_thr ? v.w * dscale : 0.f;
        }
    };
    if (t < n_tok) {
        const int64_t id = ids[(ids_B == 1 ? 0 : b) * n_tok + t];
        const float* e = embed + id * d;
        for (int64_t c = (int64_t)threadIdx.x * 4; c < d; c += 1024) {
            float4 v = *reinterpret_cast<const float4*>(e + c);
            if constexpr (OUT_BF) v = rbf4(v);
            if (pe) {
                float4 p4 = *reinterpret_cast<const float4*>(pe + c);
                if constexpr (OUT_BF) p4 = rbf4(p4);
                v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
                if constexpr (OUT_BF) v = rbf4(v);
            }
            drop4(v, c);
            st_stream4<OUT_BF>(h0, row * d + c, v);
        }
    } else {
        const bf16_t* xr = x_tok + (b * P + (t - n_tok)) * d;
        for (int64_t c = (int64_t)threadIdx.x * 4; c < d; c += 1024) {
            const u32x2 k = *reinterpret_cast<const u32x2*>(xr + c);
            float4 v = make_float4(__uint_as_float(k[0] << 16), __uint_as_float(k[0] & 0xffff0000u), __uint_as_float(k[1] << 16),
                                   __uint_as_float(k[1] & 0xffff0000u));
            if (pe) {
                float4 p4 = *reinterpret_cast<const float4*>(pe + c);
                if constexpr (OUT_BF) p4 = rbf4(p4);
                v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
                if constexpr (OUT_BF) v = rbf4(v);
            }
            drop4(v, c);
            st_stream4<OUT_BF>(h0, row * d + c, v);
        }
    }
}

// backward of the assembly (+ embd dropout): dx_tok[b, p, :] = bf16(mask * dh0[b, n_tok + p, :]) — the token rows only, one pass
template <bool IN_BF>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const void* __restrict__ dh0, bf16_t* __restrict__ dx, int64_t n_tok, int64_t P, int64_t d,
                                                           uint32_t drop_thr, uint32_t drop_seed) {
    const int64_t T = n_tok + P;
    const int64_t r = blockIdx.x, b = r / P, pp = r % P, row = b * T + n_tok + pp;
    const uint32_t dbase = drop_base(drop_seed, 0u);
    const float dscale = drop_scale_of(drop_thr);
    for (int64_t c = (int64_t)threadIdx.x * 4; c < d; c += 1024) {
        float4 v;
        if constexpr (IN_BF) {
            const u32x2 k = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dh0) + row * d + c);
            v = make_float4(__uint_as_float(k[0] << 16), __uint_as_float(k[0] & 0xffff0000u), __uint_as_float(k[1] << 16), __uint_as_float(k[1] & 0xffff0000u));
        } else {
            v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dh0) + row * d + c);
        }
        if (drop_thr) {
            const uint2 w = drop_quad(dbase, (uint32_t)row, (uint32_t)c >> 2);
            v.x = (w.x & 0xffffu) >= drop_thr ? v.x * dscale : 0.f; v.y = (w.x >> 16) >= drop_thr ? v.y * dscale : 0.f;
            v.z = (w.y & 0xffffu) >= drop_thr ? v.z * dscale : 0.f; v.w = (w.y >> 16) >= drop_thr ? v.w * dscale : 0.f;
        }
        *reinterpret_cast<u32x2*>(dx + r * d + c) = (u32x2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    }
}

template <bool IN_BF, bool OUT_BF>
__global__ void revin_denorm_kernel(const void* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ stdev,
                                    void* __restrict__ out, int64_t B, int64_t T, int64_t C) {
    const int64_t total = B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % C, b = i / (T * C);
        const float yv = IN_BF ? bf16_to_f32(reinterpret_cast<const bf16_t*>(y)[i]) : reinterpret_cast<const float*>(y)[i];
        const float v = yv * stdev[b * C + c] + (mean ? mean[b * C + c] : 0.f);
        if (OUT_BF) reinterpret_cast<bf16_t*>(out)[i] = f32_to_bf16(v);
        else reinterpret_cast<float*>(out)[i] = v;
    }
}

inline unsigned grid_for(int64_t items, int block) {
    int64_t g = (items + block - 1) / block;
    if (g > 2048 * 4) g = 2048 * 4;  // grid-stride the rest (GUIDE Guideline 11)
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int mtl_cast_pad_f32_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t,
                                     int64_t R, int64_t Cc, void* stream) {
    if (!src || (!dst && !dst_t) || R <= 0 || Cc <= 0) return MTL_ERR_ARG;
    if ((dst && ld_dst < Cc) || (dst_t && ld_dst_t < R)) return MTL_ERR_ARG;
    const int64_t rmax = (dst_t && ld_dst_t > R) ? ld_dst_t : R;
    const int64_t cmax = (dst && ld_dst > Cc) ? ld_dst : Cc;
    dim3 grid((unsigned)((cmax + 63) / 64), (unsigned)((rmax + 63) / 64));
    hipLaunchKernelGGL(cast_pad_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, (bf16_t*)dst, ld_dst, (bf16_t*)dst_t,
                       ld_dst_t, R, Cc);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t R, int64_t Cc, void* stream) {
    if (!src || !dst || R <= 0 || Cc <= 0 || ld_dst < R) return MTL_ERR_ARG;
    const int64_t rmax = ld_dst > R ? ld_dst : R;
    dim3 grid((unsigned)((Cc + 63) / 64), (unsigned)((rmax + 63) / 64));
    hipLaunchKernelGGL(transpose_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst,
                       R, Cc, (float*)nullptr);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_transpose_colsum_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, float* colsum, int64_t R, int64_t Cc,
                                         void* stream) {
    if (!src || !dst || !colsum || R <= 0 || Cc <= 0 || ld_dst < R) return MTL_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(colsum, 0, (size_t)Cc * sizeof(float), st) != hipSuccess) return MTL_ERR_LAUNCH;
    const int64_t rows = ld_dst > R ? ld_dst : R;
    dim3 grid((unsigned)((Cc + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(transpose_bf16_kernel<true>, grid, dim3(256), 0, st, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, R, Cc, colsum);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return MTL_ERR_ARG;
    if (((uintptr_t)src % 16) || ((uintptr_t)dst % 8)) return MTL_ERR_ALIGN;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return MTL_ERR_ARG;
    if (((uintptr_t)src % 8) || ((uintptr_t)dst % 16)) return MTL_ERR_ALIGN;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

// ---------------------------------------------------------------- channel mixing of the non-concat covariate modes
// y[b, r, o] = bias[o] + sum_{j, c} W[o, j*C + c] * x[b, c, r, j]        x bf16 [B, C, R, J], y bf16 or f32 [B, R, O]
//   "add" (R:models/medtsllm.py:286: mean over the channels)       J = O = 1, W = NULL (every weight 1 / C), no bias
//   "weighted-average" (:288-291, feature_weighting = Linear(C, 1)) J = O = 1
//   "independent" (:371: mean of the per-channel predictions)        J = O = 1, W = NULL, fp32 output
//   "merge-end" (:373-375, Linear(C * n_out, n_out) on the channel-last view)   J = O = n_out
// fp32 arithmetic on the bf16 inputs, one rounding at the output — the same numbers the ATen route (float().mean() / F.linear on the
// permuted fp32 view) produced, without the fp32 copies and permutes. Streaming kernels; the weight gradient is a fixed-order two-stage sum.
template <bool YBF>
__global__ __launch_bounds__(256) void channel_mix_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                              void* __restrict__ y, int64_t B, int C, int64_t R, int J, int O) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (b, r)
    if (idx >= B * R) return;
    const int64_t b = idx / R, r = idx - b * R;
    const float inv_c = 1.0f / (float)C;
    for (int o = 0; o < O; ++o) {
        float acc = 0.f;
        for (int j = 0; j < J; ++j)
            for (int c = 0; c < C; ++c) {
                const float xv = bf16_to_f32(x[((b * C + c) * R + r) * J + j]);
                acc += W ? W[o * J * C + j * C + c] * xv : xv;
            }
        if (!W) acc *= inv_c;
        if (bias) acc += bias[o];
        if (YBF) reinterpret_cast<bf16_t*>(y)[idx * O + o] = f32_to_bf16(acc);
        else reinterpret_cast<float*>(y)[idx * O + o] = acc;
    }
}

// dx[b, c, r, j] = sum_o W[o, j*C + c] * dy[b, r, o]   (bf16, the dtype of x)
template <bool YBF>
__global__ __launch_bounds__(256) void channel_mix_dx_kernel(const void* __restrict__ dy, const float* __restrict__ W, bf16_t* __restrict__ dx, int64_t B,
                                                             int C, int64_t R, int J, int O) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (b, c, r)
    if (idx >= B * C * R) return;
    const int64_t r = idx % R, bc = idx / R, c = bc % C, b = bc / C;
    const float inv_c = 1.0f / (float)C;
    for (int j = 0; j < J; ++j) {
        float acc = 0.f;
        for (int o = 0; o < O; ++o) {
            const float g = YBF ? bf16_to_f32(reinterpret_cast<const bf16_t*>(dy)[(b * R + r) * O + o]) : reinterpret_cast<const float*>(dy)[(b * R + r) * O + o];
            acc += (W ? W[o * J * C + j * C + c] : inv_c) * g;
        }
        dx[idx * J + j] = f32_to_bf16(acc);
    }
}

// partial[s][o*J*C + j*C + c] = sum over the s-th slice of (b, r) of dy[b, r, o] * x[b, c, r, j]; partial[s][O*J*C + o] = the slice's sum of dy[., ., o]
template <bool YBF>
__global__ __launch_bounds__(256) void channel_mix_dw_kernel(const bf16_t* __restrict__ x, const void* __restrict__ dy, float* __restrict__ partial, int64_t B,
                                                             int C, int64_t R, int J, int O, int S) {
    __shared__ float red[4];
    const int K = J * C, nw = O * K + O;
    const int e = blockIdx.x, s = blockIdx.y;                 // e < O*K: a weight; e >= O*K: bias o = e - O*K
    const bool is_b = e >= O * K;
    const int o = is_b ? e - O * K : e / K, k = is_b ? 0 : e % K, j = k / C, c = k % C;
    const int64_t n = B * R, per = (n + S - 1) / S, i0 = (int64_t)s * per, i1 = i0 + per < n ? i0 + per : n;
    float acc = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float g = YBF ? bf16_to_f32(reinterpret_cast<const bf16_t*>(dy)[i * O + o]) : reinterpret_cast<const float*>(dy)[i * O + o];
        if (is_b) acc += g;
        else {
            const int64_t b = i / R, r = i - b * R;
            acc += g * bf16_to_f32(x[((b * C + c) * R + r) * J + j]);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)s * nw + e] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void channel_mix_dw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, float* __restrict__ dbias, int nw_w, int O, int S) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nw_w + O) return;
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += partial[(int64_t)s * (nw_w + O) + e];
    if (e < nw_w) { if (dW) dW[e] = t; }
    else if (dbias) dbias[e - nw_w] = t;
}

extern "C" int mtl_channel_mix_fwd(const void* x, const float* W, const float* bias, void* y, int y_dtype, int64_t B, int64_t C, int64_t R, int64_t J,
                                   int64_t O, void* stream) {
    if (!x || !y || B <= 0 || C <= 0 || R <= 0 || J <= 0 || O <= 0 || (y_dtype != MTL_BF16 && y_dtype != MTL_F32)) return MTL_ERR_ARG;
    if (!W && (J != 1 || O != 1 || bias)) return MTL_ERR_ARG;              // the plain mean has one output and no bias
    const dim3 grid((unsigned)((B * R + 255) / 256));
    if (y_dtype == MTL_BF16) hipLaunchKernelGGL(channel_mix_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, W, bias, y, B, (int)C, R, (int)J, (int)O);
    else hipLaunchKernelGGL(channel_mix_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, W, bias, y, B, (int)C, R, (int)J, (int)O);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" size_t mtl_channel_mix_workspace_bytes(int64_t C, int64_t J, int64_t O) { return (size_t)64 * (size_t)(O * J * C + O) * sizeof(float); }

extern "C" int mtl_channel_mix_bwd(const void* x, const float* W, const void* dy, int dy_dtype, void* dx, float* dW, float* dbias, void* workspace,
                                   int64_t B, int64_t C, int64_t R, int64_t J, int64_t O, void* stream) {
    if (!x || !dy || B <= 0 || C <= 0 || R <= 0 || J <= 0 || O <= 0 || (dy_dtype != MTL_BF16 && dy_dtype != MTL_F32)) return MTL_ERR_ARG;
    if (!W && (J != 1 || O != 1)) return MTL_ERR_ARG;
    if ((dW || dbias) && !workspace) return MTL_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool ybf = dy_dtype == MTL_BF16;
    if (dx) {
        const dim3 grid((unsigned)((B * C * R + 255) / 256));
        if (ybf) hipLaunchKernelGGL(channel_mix_dx_kernel<true>, grid, dim3(256), 0, st, dy, W, (bf16_t*)dx, B, (int)C, R, (int)J, (int)O);
        else hipLaunchKernelGGL(channel_mix_dx_kernel<false>, grid, dim3(256), 0, st, dy, W, (bf16_t*)dx, B, (int)C, R, (int)J, (int)O);
    }
    if (dW || dbias) {
        const int S = 64, nw_w = (int)(O * J * C);
        const dim3 grid((unsigned)(nw_w + O), (unsigned)S);
        if (ybf) hipLaunchKernelGGL(channel_mix_dw_kernel<true>, grid, dim3(256), 0, st, (const bf16_t*)x, dy, (float*)workspace, B, (int)C, R, (int)J, (int)O, S);
        else hipLaunchKernelGGL(channel_mix_dw_kernel<false>, grid, dim3(256), 0, st, (const bf16_t*)x, dy, (float*)workspace, B, (int)C, R, (int)J, (int)O, S);
        hipLaunchKernelGGL(channel_mix_dw_reduce_kernel, dim3((unsigned)((nw_w + O + 255) / 256)), dim3(256), 0, st, (const float*)workspace, dW, dbias, nw_w, (int)O, S);
    }
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

// row sums of a bf16 matrix in fp32: one wave per row (the mapping layer's bias gradient = row sums of d source)
__global__ __launch_bounds__(256) void rowsum_kernel(const bf16_t* __restrict__ src, int64_t ld, float* __restrict__ dst, int64_t R, int64_t Cc) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const bf16_t* r = src + row * ld;
    float s = 0.f;
    for (int64_t c = lane; c < Cc; c += 64) s += bf16_to_f32(r[c]);
    s = wave_sum(s);
    if (lane == 0) dst[row] = s;
}

extern "C" int mtl_rowsum_bf16(const void* src, int64_t ld_src, float* dst, int64_t R, int64_t Cc, void* stream) {
    if (!src || !dst || R <= 0 || Cc <= 0) return MTL_ERR_ARG;
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, dst, R, Cc);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_colsum_bf16(const void* src, int64_t ld_src, float* dst, int64_t R, int64_t Cc, void* stream) {
    if (!src || !dst || R <= 0 || Cc <= 0) return MTL_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dst, 0, (size_t)Cc * sizeof(float), st) != hipSuccess) return MTL_ERR_LAUNCH;
    dim3 grid((unsigned)((Cc + 63) / 64), (unsigned)((R + 255) / 256));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, (const bf16_t*)src, ld_src, dst, R, Cc);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_rope_inplace_rows(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, int64_t M, int64_t T,
                                     int64_t n_rot_heads, int64_t D, int inverse, int64_t group_rows, int64_t group_stride,
                                     int64_t row_offset, void* stream) {
    if (!qkv || !cos_t || !sin_t || M <= 0 || T <= 0 || n_rot_heads <= 0) return MTL_ERR_ARG;
    if (D % 4 != 0 || ld % 2 != 0) return MTL_ERR_ALIGN;
    const int64_t items = M * n_rot_heads * (D / 4);
    if (D % 16 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (reinterpret_cast<uintptr_t>(cos_t) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(sin_t) & 15) == 0)
        hipLaunchKernelGGL(rope_kernel_v8, dim3(grid_for(items / 4, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv, ld, cos_t, sin_t, M, T,
                           (int)n_rot_heads, (int)D, inverse, group_rows, group_stride, row_offset);
    else
    hipLaunchKernelGGL(rope_kernel, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv, ld, cos_t, sin_t, M, T,
                       (int)n_rot_heads, (int)D, inverse, group_rows, group_stride, row_offset);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_rope_inplace(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, int64_t M, int64_t T,
                                int64_t n_rot_heads, int64_t D, int inverse, void* stream) {
    return mtl_rope_inplace_rows(qkv, ld, cos_t, sin_t, M, T, n_rot_heads, D, inverse, 0, 0, 0, stream);
}

extern "C" int mtl_swiglu_fwd(const void* gu, void* h, int64_t M, int64_t F, void* stream) {
    if (!gu || !h || M <= 0 || F <= 0) return MTL_ERR_ARG;
    if (F % 2 != 0) return MTL_ERR_ALIGN;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(M * F / 2, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu, (bf16_t*)h, M, F);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_swiglu_bwd_rows(const void* gu, const void* dh, void* dgu, int64_t M, int64_t F, int64_t group_rows,
                                   int64_t group_stride, int64_t row_offset, int interleaved, void* stream) {
    if (!gu || !dh || !dgu || M <= 0 || F <= 0) return MTL_ERR_ARG;
    if (F % 2 != 0) return MTL_ERR_ALIGN;
    if (interleaved)
        hipLaunchKernelGGL(swiglu_bwd_kernel<true>, dim3(grid_for(M * F / 2, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu,
                           (const bf16_t*)dh, (bf16_t*)dgu, M, F, group_rows, group_stride, row_offset);
    else
    hipLaunchKernelGGL(swiglu_bwd_kernel<false>, dim3(grid_for(M * F / 2, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu,
                       (const bf16_t*)dh, (bf16_t*)dgu, M, F, group_rows, group_stride, row_offset);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_swiglu_bwd(const void* gu, const void* dh, void* dgu, int64_t M, int64_t F, void* stream) {
    return mtl_swiglu_bwd_rows(gu, dh, dgu, M, F, 0, 0, 0, 0, stream);
}

extern "C" int mtl_assemble_bwd_t(const void* dh0, int dh0_dtype, void* dx_tok, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p,
                                  uint32_t drop_seed, void* stream) {
    if (!dh0 || !dx_tok || B <= 0 || P <= 0 || d <= 0 || n_tok < 0 || drop_p < 0.f || drop_p >= 1.f) return MTL_ERR_ARG;
    if (dh0_dtype != MTL_F32 && dh0_dtype != MTL_BF16) return MTL_ERR_ARG;
    if (d % 4 != 0) return MTL_ERR_ALIGN;
    const uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    if (dh0_dtype == MTL_BF16)
        hipLaunchKernelGGL(assemble_bwd_kernel<true>, dim3((unsigned)(B * P)), dim3(256), 0, (hipStream_t)stream, dh0, (bf16_t*)dx_tok, n_tok, P, d, thr, drop_seed);
    else
        hipLaunchKernelGGL(assemble_bwd_kernel<false>, dim3((unsigned)(B * P)), dim3(256), 0, (hipStream_t)stream, dh0, (bf16_t*)dx_tok, n_tok, P, d, thr, drop_seed);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_assemble_bwd(const float* dh0, void* dx_tok, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p,
                                uint32_t drop_seed, void* stream) {
    return mtl_assemble_bwd_t(dh0, MTL_F32, dx_tok, B, n_tok, P, d, drop_p, drop_seed, stream);
}

extern "C" int mtl_assemble_llm_input_t(const int32_t* ids, int64_t ids_B, const float* embed, const void* x_tok, const float* wpe,
                                        void* h0, int h0_dtype, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p, uint32_t drop_seed,
                                        void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f) return MTL_ERR_ARG;
    if (!x_tok || !h0 || B <= 0 || P <= 0 || d <= 0 || n_tok < 0) return MTL_ERR_ARG;
    if (h0_dtype != MTL_F32 && h0_dtype != MTL_BF16) return MTL_ERR_ARG;
    if (n_tok > 0 && (!ids || !embed || (ids_B != 1 && ids_B != B))) return MTL_ERR_ARG;
    if (d % 4 != 0) return MTL_ERR_ALIGN;
    const uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    if (h0_dtype == MTL_BF16)
        hipLaunchKernelGGL(assemble_kernel<true>, dim3((unsigned)(B * (n_tok + P))), dim3(256), 0, (hipStream_t)stream, ids, ids_B, embed,
                           (const bf16_t*)x_tok, wpe, h0, n_tok, P, d, thr, drop_seed);
    else
        hipLaunchKernelGGL(assemble_kernel<false>, dim3((unsigned)(B * (n_tok + P))), dim3(256), 0, (hipStream_t)stream, ids, ids_B, embed,
                           (const bf16_t*)x_tok, wpe, h0, n_tok, P, d, thr, drop_seed);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_assemble_llm_input(const int32_t* ids, int64_t ids_B, const float* embed, const void* x_tok, const float* wpe,
                                      float* h0, int64_t B, int64_t n_tok, int64_t P, int64_t d, float drop_p, uint32_t drop_seed,
                                      void* stream) {
    return mtl_assemble_llm_input_t(ids, ids_B, embed, x_tok, wpe, h0, MTL_F32, B, n_tok, P, d, drop_p, drop_seed, stream);
}

extern "C" int mtl_revin_denorm(const void* y, int y_dtype, const float* mean, const float* stdev, void* out, int out_dtype, int64_t B, int64_t T,
                                int64_t C, void* stream) {
    if (!y || !stdev || !out || B <= 0 || T <= 0 || C <= 0) return MTL_ERR_ARG;
    if ((y_dtype != MTL_F32 && y_dtype != MTL_BF16) || (out_dtype != MTL_F32 && out_dtype != MTL_BF16)) return MTL_ERR_ARG;
    const dim3 grid(grid_for(B * T * C, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (y_dtype == MTL_BF16 && out_dtype == MTL_BF16) hipLaunchKernelGGL((revin_denorm_kernel<true, true>), grid, block, 0, st, y, mean, stdev, out, B, T, C);
    else if (y_dtype == MTL_BF16) hipLaunchKernelGGL((revin_denorm_kernel<true, false>), grid, block, 0, st, y, mean, stdev, out, B, T, C);
    else if (out_dtype == MTL_BF16) hipLaunchKernelGGL((revin_denorm_kernel<false, true>), grid, block, 0, st, y, mean, stdev, out, B, T, C);
    else hipLaunchKernelGGL((revin_denorm_kernel<false, false>), grid, block, 0, st, y, mean, stdev, out, B, T, C);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

namespace {
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* x, float* y, int64_t M, int d, uint32_t thr, float scale, uint32_t seed) {
    const int64_t nq = (int64_t)M * (d / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (d / 4);
        const int c = (int)(i - row * (d / 4)) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + row * d + c);
        const uint32_t dbase = drop_base(seed, 0u);
        const uint2 wq = drop_quad(dbase, (uint32_t)row, (uint32_t)c >> 2);
                        const uint32_t w0 = wq.x, w1 = wq.y;
        v.x = (w0 & 0xffffu) >= thr ? v.x * scale : 0.f; v.y = (w0 >> 16) >= thr ? v.y * scale : 0.f;
        v.z = (w1 & 0xffffu) >= thr ? v.z * scale : 0.f; v.w = (w1 >> 16) >= thr ? v.w * scale : 0.f;
        *reinterpret_cast<float4*>(y + row * d + c) = v;
    }
}
}  // namespace

extern "C" int mtl_dropout_f32(const float* x, float* y, int64_t M, int64_t d, float p, uint32_t seed, void* stream) {
    if (!x || !y || M <= 0 || d <= 0 || p < 0.f || p >= 1.f) return MTL_ERR_ARG;
    if (d % 4 != 0) return MTL_ERR_ALIGN;
    const int64_t nq = M * (d / 4);
    const unsigned blocks = (unsigned)((nq + 255) / 256 < 8192 ? (nq + 255) / 256 : 8192);
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, M, (int)d,
                       drop_threshold(p), drop_scale_of(drop_threshold(p)), seed);
    MTL_CHECK_LAUNCH();
    return MTL_OK;
}

extern "C" int mtl_abi_version(void) { return MTL_ABI_VERSION; }

extern "C" int mtl_build_flags(void) {
    int f = 0;
#ifdef MTL_DIAG
    f |= MTL_BUILD_DIAG_ENV;
#endif
#if defined(MTL_DIAG_NOHASH) || defined(MTL_DIAG_ATTN_NODROP) || defined(MTL_DIAG_W4VAR) || defined(MTL_DIAG_WRONG)
    f |= MTL_BUILD_DIAG_WRONG;
#endif
    return f;
}

extern "C" const char* mtl_strerror(int code) {
    switch (code) {
        case MTL_OK: return "ok";
        case MTL_ERR_ARG: return "invalid argument (null pointer / non-positive size / inconsistent shape)";
        case MTL_ERR_ALIGN: return "alignment requirement violated (K % 64, ld % 8, 16-byte pointers, d % 4)";
        case MTL_ERR_UNSUPPORTED: return "unsupported configuration (head_dim, epilogue or size out of range)";
        case MTL_ERR_LAUNCH: return "HIP kernel launch failed";
        case MTL_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown error";
    }
}
