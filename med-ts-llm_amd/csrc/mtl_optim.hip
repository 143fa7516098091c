// Adam / AdamW step of every trainable tensor in ONE streaming launch (R:tasks/base.py:97,99 builds torch.optim.Adam /
// AdamW; the loop bodies call optimizer.step(), R:tasks/forecasting.py:27). HBM-bound: 28 B/element (read p,g,m,v;
// write p,m,v) + 2 B for the optional bf16 shadow of the updated weight, which replaces the separate
// f32 -> bf16 re-cast (another 4 B read) the next forward would otherwise do.
#include "mtl_common.h"

namespace {

constexpr int MAX_T = MTL_ADAM_MAX_TENSORS;
constexpr int THREADS = 256;
constexpr int VEC_PER_THREAD = 4;                          // float4 per thread
constexpr int64_t CHUNK = (int64_t)THREADS * 4 * VEC_PER_THREAD;   // elements per block

struct AdamTable {
    mtl_adam_tensor t[MAX_T];
    int first_block[MAX_T + 1];
    int count;
};

struct AdamHyper { float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt; int decoupled; };

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamHyper& h) {
    if (h.wd != 0.f) {
        if (h.decoupled) p *= 1.f - h.lr * h.wd;           // AdamW
        else g += h.wd * p;                                // Adam L2
    }
    m += (g - m) * (1.f - h.beta1);                        // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + (1.f - h.beta2) * g * g;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    return p - (h.lr / h.bc1) * (m / denom);
}

__global__ __launch_bounds__(THREADS) void adam_multi_kernel(const AdamTable tab, const AdamHyper h) {
    // block -> tensor: linear search over <= MAX_T prefix entries (wave-uniform)
    int ti = 0;
    const int b = blockIdx.x;
    while (ti + 1 < tab.count && b >= tab.first_block[ti + 1]) ++ti;
    const mtl_adam_tensor t = tab.t[ti];
    const int64_t base = (int64_t)(b - tab.first_block[ti]) * CHUNK;
    const bool p16 = t.param_dtype == MTL_BF16;          // setup.dtype = "bf16": the parameter (and its gradient) ARE bf16; moments stay fp32
    const bf16_t* gp16 = reinterpret_cast<const bf16_t*>(t.g);
    bf16_t* pp16 = reinterpret_cast<bf16_t*>(t.p);
    const bool vec = !p16 && (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0;
    // bf16 parameter + gradient: 8-byte vectors of both, 16-byte vectors of the fp32 moments (element-wise 2-byte accesses ran the [1024, 50257] mapping
    // weight's update at 575 us against 366 us for the fp32 parameter that moves MORE bytes: profiles/r06_bf16_kernel_stats.txt)
    const bool vec16 = p16 && (((uintptr_t)t.p | (uintptr_t)t.g) & 7) == 0 && (((uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0;
    bf16_t* sh = reinterpret_cast<bf16_t*>(t.shadow);
#pragma unroll
    for (int i = 0; i < VEC_PER_THREAD; ++i) {
        const int64_t e0 = base + ((int64_t)i * THREADS + threadIdx.x) * 4;
        if (e0 >= t.n) break;
        float pp[4], gg[4], mm[4], vv[4];
        const int nv = (t.n - e0) >= 4 ? 4 : (int)(t.n - e0);
        if (vec && nv == 4) {
            // streamed once per step: non-temporal so that 2 GB of optimiser state does not evict the model from L2 / MALL
            const f32x4 p4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.p + e0)), g4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.g + e0));
            const f32x4 m4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.m + e0)), v4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.v + e0));
            pp[0] = p4.x; pp[1] = p4.y; pp[2] = p4.z; pp[3] = p4.w;
            gg[0] = g4.x; gg[1] = g4.y; gg[2] = g4.z; gg[3] = g4.w;
            mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
            vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
        } else if (vec16 && nv == 4) {
            const u32x2 pk = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(pp16 + e0)), gk = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(gp16 + e0));
            const f32x4 m4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.m + e0)), v4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.v + e0));
            pp[0] = __uint_as_float(pk[0] << 16); pp[1] = __uint_as_float(pk[0] & 0xffff0000u); pp[2] = __uint_as_float(pk[1] << 16); pp[3] = __uint_as_float(pk[1] & 0xffff0000u);
            gg[0] = __uint_as_float(gk[0] << 16); gg[1] = __uint_as_float(gk[0] & 0xffff0000u); gg[2] = __uint_as_float(gk[1] << 16); gg[3] = __uint_as_float(gk[1] & 0xffff0000u);
            mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
            vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = e < nv;
                if (p16) { pp[e] = ok ? bf16_to_f32(pp16[e0 + e]) : 0.f; gg[e] = ok ? bf16_to_f32(gp16[e0 + e]) : 0.f; }
                else { pp[e] = ok ? t.p[e0 + e] : 0.f; gg[e] = ok ? t.g[e0 + e] : 0.f; }
                mm[e] = ok ? t.m[e0 + e] : 0.f; vv[e] = ok ? t.v[e0 + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[e] = adam_one(pp[e], gg[e], mm[e], vv[e], h);
        if (vec && nv == 4) {
            __builtin_nontemporal_store((f32x4){pp[0], pp[1], pp[2], pp[3]}, reinterpret_cast<f32x4*>(t.p + e0));
            __builtin_nontemporal_store((f32x4){mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<f32x4*>(t.m + e0));
            __builtin_nontemporal_store((f32x4){vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<f32x4*>(t.v + e0));
        } else if (vec16 && nv == 4) {
            // one rounding of the updated parameter per step; the shadow below takes the ROUNDED values, so that it stays bit-identical to the parameter
            const u32x2 pk = {pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3])};
            __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(pp16 + e0));
            __builtin_nontemporal_store((f32x4){mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<f32x4*>(t.m + e0));
            __builtin_nontemporal_store((f32x4){vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<f32x4*>(t.v + e0));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nv) {
                    if (p16) pp16[e0 + e] = f32_to_bf16(pp[e]);      // one rounding per step (torch.optim.Adam on bf16 tensors rounds every intermediate)
                    else t.p[e0 + e] = pp[e];
                    t.m[e0 + e] = mm[e]; t.v[e0 + e] = vv[e];
                }
        }
        if (sh) {   // bf16 shadow [rows, ld_shadow] of the fp32 [rows, cols] weight (n < 2^32 checked on the host)
            const uint32_t cols = (uint32_t)t.cols;
            uint32_t row = (uint32_t)e0 / cols, col = (uint32_t)e0 - row * cols;
            if (nv == 4 && col + 4 <= cols) {
                // the four elements sit in one row: ONE 8-byte store (2-byte aligned when cols is odd, as for the [1024, 50257] mapping
                // weight: legal in the unaligned access mode compute queues run in) instead of four 2-byte ones
                typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(2)));
                *reinterpret_cast<u32x2_u*>(sh + (int64_t)row * t.ld_shadow + col) = (u32x2_u){pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3])};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < nv) sh[(int64_t)row * t.ld_shadow + col] = f32_to_bf16(pp[e]);
                    if (++col == cols) { col = 0; ++row; }
                }
            }
        }
    }
}

}  // namespace

extern "C" int mtl_adam_step(const mtl_adam_tensor* tensors, int count, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int decoupled, int64_t step, void* stream) {
    if (count < 0 || (count > 0 && !tensors) || step < 1) return MTL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AdamHyper h;
    h.lr = lr; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.wd = weight_decay; h.decoupled = decoupled;
    h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    for (int i = 0; i < count;) {       // one launch per table of up to MAX_T non-empty tensors; `i` carries across tables
        AdamTable tab;
        tab.count = 0;
        int blocks = 0;
        double bytes = 0;
        for (; i < count && tab.count < MAX_T; ++i) {
            const mtl_adam_tensor& t = tensors[i];
            if (t.n == 0) continue;
            if (!t.p || !t.g || !t.m || !t.v || t.n < 0) return MTL_ERR_ARG;
            if (t.shadow && (t.cols <= 0 || t.ld_shadow < t.cols || t.n % t.cols != 0 || t.n >= ((int64_t)1 << 32))) return MTL_ERR_ARG;
            tab.t[tab.count] = t;
            tab.first_block[tab.count] = blocks;
            blocks += (int)((t.n + CHUNK - 1) / CHUNK);
            if (t.param_dtype != MTL_F32 && t.param_dtype != MTL_BF16) return MTL_ERR_ARG;
            bytes += (double)t.n * ((t.param_dtype == MTL_BF16 ? 22.0 : 28.0) + (t.shadow ? 2.0 : 0.0));      // p, g, m, v read; p, m, v written (+ bf16 shadow)
            ++tab.count;
        }
        tab.first_block[tab.count] = blocks;
        if (blocks == 0) continue;
        MTL_LAUNCH("adam_multi_kernel", bytes, 1, adam_multi_kernel, dim3(blocks), dim3(THREADS), 0, st, tab, h);
        MTL_CHECK_LAUNCH();
    }
    return MTL_OK;
}
