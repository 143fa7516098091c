// mtl_common.h — shared device helpers for the MedTsLLM gfx950 kernels.
// CDNA4-only code (wave64, MFMA, LDS-DMA); there is deliberately no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/medtsllm_hip.h"

typedef uint16_t bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MTL_WAVE 64

// Environment switches exist in DIAGNOSTIC builds only (-DMTL_DIAG, tools/build_variant.sh): the product library reads no environment variable and
// keeps no mutable global besides the opt-in launch profiler, so nothing in a user's shell can change its dispatch. mtl_build_flags() reports the build.
#ifdef MTL_DIAG
#include <cstdlib>
inline int mtl_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int mtl_env_int(const char*, int dflt) { return dflt; }
#endif

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round-to-nearest-even: written as plain conversions so that hipcc emits the gfx950 hardware packed
// convert (v_cvt_pk_bf16_f32: 1 instruction per PAIR instead of ~6 VALU ops per element for a software RNE).
typedef __attribute__((ext_vector_type(2))) float mtl_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 mtl_bf16x2;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const mtl_f32x2 v = {lo, hi};
    const mtl_bf16x2 b = __builtin_convertvector(v, mtl_bf16x2);
    return __builtin_bit_cast(uint32_t, b);
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

// gelu_new (HF:activations.py:65-66) and its derivative, fp32.
// 0.5*x*(1+tanh(u)) == x*sigmoid(2u), u = k0*(x + k1*x^3): one v_exp_f32 + one v_rcp_f32 instead of tanhf.
__device__ __forceinline__ float gelu_new_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.0f * k0 * (x + k1 * x * x * x);
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
    return x * s;
}
__device__ __forceinline__ float dgelu_new_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float x2 = x * x;
    const float u2 = 2.0f * k0 * (x + k1 * x * x2);
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
    // d/dx [x*s(2u)] = s + x * s*(1-s) * 2u'   with u' = k0*(1 + 3*k1*x^2)
    return s + x * s * (1.0f - s) * (2.0f * k0 * (1.0f + 3.0f * k1 * x2));
}

// dropout: counter-based keep mask, a pure function of (seed, stream, a, b) so that a backward kernel regenerates exactly the
// forward's mask; tests replicate it on the host (tests/helpers.py). Attention: stream = batch*head, (a, b) = (query, key);
// matrices: stream = 0, (a, b) = (row, column).
// Cost matters: the mask is evaluated for every attention probability in four kernels per layer. One hash (drop_quad) yields TWO words =
// the decisions of the FOUR elements (a, 4*quad .. 4*quad + 3) as uniform 16-bit fields, and the mixing uses only full-rate 24-bit
// multiply-adds (v_mad_u32_u24) — 32-bit integer multiplies are quarter rate on CDNA: 12 VALU operations per four elements instead of
// ~30 per element for the murmur-style hash of round 1 (attention forward 23.8 -> 20.5 us per layer).
// A keep threshold is a 16-bit number: the effective rate is thr / 65536 (p = 0.1 -> 0.09999), and the kept values are scaled by
// the EXACT 1 / (1 - thr / 65536), so the mask is unbiased. a, quad < 2^24 (beyond that the pattern repeats).
__host__ __device__ __forceinline__ uint32_t drop_base(uint32_t seed, uint32_t stream) {   // wave-uniform: scalar unit, once per kernel
    uint32_t a = seed ^ (stream * 0x9E3779B1u);
    a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
    return a;
}
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
// mask words of the FOUR elements (a, 4 quad .. 4 quad + 3): .x carries elements 0 / 1 in its low / high 16 bits, .y elements 2 / 3.
// One multiply-xorshift round on the 24-bit multiplier (full rate) gives .x, one more gives .y: 12 VALU operations per four
// elements (the first version hashed every PAIR with 12). Statistics of the fields (keep rate, neighbour / row / stream correlations,
// chi-square of the bytes) were checked on the host replica before adoption (tests/helpers.py::drop_u16, tests/test_host_logic.py).
__device__ __forceinline__ uint2 drop_quad(uint32_t base, uint32_t a, uint32_t quad) {
#ifdef MTL_DIAG_NOHASH      // diagnostic builds only (tools/build_variant.sh): what the hash itself costs a kernel — every element kept, WRONG results
    return make_uint2(0xffffffffu ^ (base & 1u), 0xffffffffu ^ (a & 1u));
#endif
    uint32_t h = mad24(a, 0x9E3779u, base) ^ __umul24(quad, 0x85EBCBu);
    h ^= h >> 15; h = mad24(h, 0xC2B2AFu, h >> 24);
    h ^= h >> 13;
    uint32_t g = mad24(h, 0x27D4EBu, h >> 8);
    g ^= g >> 15;
    return make_uint2(h, g);
}
// the 16-bit field of element b (its quad's words given): keep iff field >= threshold
__device__ __forceinline__ uint32_t drop_field(uint2 words, uint32_t b) {
    const uint32_t w = (b & 2u) ? words.y : words.x;
    return (b & 1u) ? w >> 16 : w & 0xffffu;
}
__device__ __forceinline__ bool drop_keep(uint32_t base, uint32_t a, uint32_t b, uint32_t thr) { return drop_field(drop_quad(base, a, b >> 2), b) >= thr; }
// dK/dV kernels: a lane owns ONE key and meets four consecutive queries (rows a0 .. a0 + 3) per tile, and the four lanes of a DPP
// quad (keys 4j .. 4j + 3, key0 % 4 == 0) meet the SAME queries: lane m hashes row a0 + m once and the quad exchanges the word pairs
// with quad_perm broadcasts (12 + 8 operations per four elements instead of 4 x 12). fields[r] = the lane's 16-bit field of row a0 + r.
template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {      // the value lane R of the caller's DPP quad holds
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xf, 0xf, false);
}
__device__ __forceinline__ void drop_fields_shared(uint32_t base, uint32_t a0, uint32_t key, uint32_t (&fields)[4]) {
    const uint2 w = drop_quad(base, a0 + (key & 3u), key >> 2);
    const uint32_t shift = (key & 1u) * 16u;
    const uint32_t pick_y = 0u - ((key >> 1) & 1u);      // all-ones where the lane's field lives in .y. Bitwise select, NOT `?:`: the compiler
                                                        // turns a lane-dependent choice between two DPP reads into a branch, and a DPP read
                                                        // under a partial EXEC mask returns 0 for the lanes parked in the other arm
                                                        // (tools/probes/dpp_quad.hip)
#define MTL_QUAD_FIELD(R)                                                     \
    do {                                                                      \
        const uint32_t x = quad_bcast<R>(w.x), y = quad_bcast<R>(w.y);        \
        fields[R] = ((x ^ ((x ^ y) & pick_y)) >> shift) & 0xffffu;            \
    } while (0)
    MTL_QUAD_FIELD(0); MTL_QUAD_FIELD(1); MTL_QUAD_FIELD(2); MTL_QUAD_FIELD(3);
#undef MTL_QUAD_FIELD
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    const float t = p * 65536.0f;
    return t >= 65535.0f ? 65535u : (t <= 0.f ? 0u : (uint32_t)t);
}
__host__ __device__ __forceinline__ float drop_scale_of(uint32_t thr) { return 65536.0f / (float)(65536u - thr); }
// seed of layer i's k-th dropout site (k = 0 attention probabilities, 1 attention-branch residual, 2 MLP-branch residual)
__host__ __device__ __forceinline__ uint32_t drop_site_seed(uint32_t seed, int layer, int k) {
    return seed ^ (0x9E3779B9u * (uint32_t)(3 * layer + k + 1));
}

// rotary embedding of one (x1, x2) = (x[j], x[j + D/2]) pair: y1 = x1 c - x2 s, y2 = x2 c + x1 s (HF rotate_half). ONE operation order for the
// stand-alone kernel and the fused epilogues, so that they agree bit for bit.
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& y1, float& y2) {
    y1 = __fmaf_rn(x1, c, -__fmul_rn(x2, s));
    y2 = __fmaf_rn(x2, c, __fmul_rn(x1, s));
}

#define MTL_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return MTL_ERR_LAUNCH;        \
    } while (0)

// ---------------------------------------------------------------- optional launch profiler (bench.py's roofline legs)
// Off by default: one relaxed flag test per launch. While on (mtl_prof_enable), a launch goes through hipExtLaunchKernelGGL with
// its own start / stop event pair: their elapsed time is the kernel's begin -> end on the device — the duration rocprofv3
// --kernel-trace reports for the same dispatch — with no launch gap inside, so nothing has to be calibrated away.
// `name` is the kernel as rocprofv3 prints it (without the anonymous-namespace prefix and the parameter list); `work` is the
// launch's ALGORITHMIC work: FLOPs (kind 0, MFMA family) or HBM bytes (kind 1, HBM family). Defined in mtl_gemm.hip.
namespace mtlprof {
bool enabled();
bool begin(hipEvent_t* e0, hipEvent_t* e1);
void end(hipEvent_t e0, hipEvent_t e1, const char* name, double work, int kind);
}  // namespace mtlprof

#define MTL_LAUNCH(NAME, WORK, KIND, KERNEL, GRID, BLOCK, LDS, ST, ...)                                    \
    do {                                                                                                   \
        hipEvent_t pe0__, pe1__;                                                                           \
        if (mtlprof::enabled() && mtlprof::begin(&pe0__, &pe1__)) {                                        \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, pe0__, pe1__, 0, __VA_ARGS__);             \
            mtlprof::end(pe0__, pe1__, NAME, WORK, KIND);                                                  \
        } else {                                                                                           \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__);                                 \
        }                                                                                                  \
    } while (0)
