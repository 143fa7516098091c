// mtl_common.h — shared device helpers for the MedTsLLM gfx950 kernels.
// CDNA4-only code (wave64, MFMA, LDS-DMA); there is deliberately no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/medtsllm_hip.h"

typedef uint16_t bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MTL_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16 (same rounding PyTorch uses; NaN kept quiet)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// gelu_new (HF:activations.py:65-66) and its derivative, fp32
__device__ __forceinline__ float gelu_new_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float t = tanhf(k0 * (x + k1 * x * x * x));
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float dgelu_new_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float x2 = x * x;
    float t = tanhf(k0 * (x + k1 * x * x2));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k0 * (1.0f + 3.0f * k1 * x2);
}

#define MTL_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return MTL_ERR_LAUNCH;        \
    } while (0)
