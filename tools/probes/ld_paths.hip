// probe: is the ~20-30 B/clk/CU cap of the GEMM operand path a property of the LDS-DMA (global_load_lds) or of the vector-memory path as a
// whole? One 512-thread workgroup per CU streams two GEMM-shaped operand panels (256 rows x 128 B per round each, row stride LD bytes,
// L2-resident per XCD) with a two-round pipeline (counted vmcnt): MODE 0 = both panels through LDS-DMA (the GEMM kernels' way),
// MODE 1 = both through plain global_load_dwordx4 into registers, MODE 2 = panel A into registers + panel B through LDS-DMA.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/ld_paths tools/probes/ld_paths.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, uint32_t* out, int rounds, int ld, int kwrap, size_t xcd_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 stages x 64 KB
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* pa = src + (size_t)(blockIdx.x & 7) * xcd_stride;     // the workgroups of an XCD share one A and one B panel
    const char* pb = pa + (size_t)256 * ld;
    const size_t lane_off = (size_t)(tid >> 3) * ld + (tid & 7) * 16;
    u32x4 ra[2][4], rb[2][4];
    u32x4 sink = {0, 0, 0, 0};
    auto issue = [&](int r, int buf) {
        const size_t ko = (size_t)(r % kwrap) * 128;
        char* st = smem + buf * 65536;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* ga = pa + (size_t)i * 64 * ld + lane_off + ko;
            const char* gb = pb + (size_t)i * 64 * ld + lane_off + ko;
            if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_void_t*)ga, (lds_void_t*)(st + (i * 512 + wave * 64) * 16), 16, 0, 0);
            else ra[buf][i] = *reinterpret_cast<const u32x4*>(ga);
            if (MODE == 1) rb[buf][i] = *reinterpret_cast<const u32x4*>(gb);
            else __builtin_amdgcn_global_load_lds((gbl_void_t*)gb, (lds_void_t*)(st + 32768 + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    issue(0, 0);
#pragma unroll 1
    for (int r = 0; r < rounds; r += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            issue(r + b + 1, b ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (MODE != 0) { _Pragma("unroll") for (int i = 0; i < 4; ++i) sink ^= ra[b][i]; }
            if (MODE == 1) { _Pragma("unroll") for (int i = 0; i < 4; ++i) sink ^= rb[b][i]; }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((sink[0] ^ sink[1] ^ sink[2] ^ sink[3]) == 0x12345678u) out[blockIdx.x] = 1;
}


// ---- second question: the same LDS-DMA pipeline on COLD lines that G workgroups of an XCD request together (a GEMM's operand panels on their first
// touch per XCD), as a function of the bytes in flight per CU: round = NL x 8 KB per workgroup, DEPTH rounds in the ring, DEPTH - 1 in flight.
// TOUCH = D > 0: a ninth wave (its own vmcnt) touches, D rounds ahead, the 1 / share of the stream's lines that falls to this workgroup (one dword per
// 128-B line): every line is then missed ONCE per XCD, by one requester, and the LDS-DMA of all sharers finds it in the L2.
template <int NL, int DEPTH, int TOUCH>
__global__ __launch_bounds__(576) void k2(const char* __restrict__ src, int rounds, int ld, int share, size_t rep_off, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, grp = slot / share;        // `share` workgroups of an XCD walk the same lines
    const int kwrap = ld / 128;
    const size_t panel = (size_t)NL * 64 * ld;                                       // NL * 64 rows
    const size_t npan = (size_t)(rounds + kwrap - 1) / kwrap;
    const char* sbase = src + rep_off + ((size_t)(xcd * (32 / share) + grp) * npan) * panel;
    const char* base = sbase + (size_t)(tid >> 3) * ld + (tid & 7) * 16;
    if (wave == 8) {                      // the toucher
        uint32_t acc = 0;
        const int lane = tid & 63, rank = slot % share;
        for (int r = 0; r < rounds; ++r) {
            if (TOUCH > 0 && r + TOUCH < rounds) {
                const int rt = r + TOUCH;
                for (int row = rank + share * lane; row < NL * 64; row += share * 64) {     // fire and forget: nobody waits for a touch
                    const char* ptr = sbase + (size_t)(rt / kwrap) * panel + (size_t)(rt % kwrap) * 128 + (size_t)row * ld;
                    asm volatile("global_load_dword %0, %1, off" : "+v"(acc) : "v"(ptr) : "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (acc == 0x12345678u) out[blockIdx.x] = acc;
        return;
    }
    auto issue = [&](int r) {
        const char* s0 = base + (size_t)(r / kwrap) * panel + (size_t)(r % kwrap) * 128;
        char* st = smem + (r % DEPTH) * (NL * 8192);
#pragma unroll
        for (int i = 0; i < NL; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(s0 + (size_t)i * 64 * ld), (lds_void_t*)(st + (i * 512 + wave * 64) * 16), 16, 0, 0);
    };
    for (int r = 0; r < DEPTH - 1; ++r) issue(r);
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if (r + DEPTH - 1 < rounds) issue(r + DEPTH - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NL) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NL, int DEPTH, int TOUCH = 0>
void run2(const char* src, int ld, int share, size_t bufsz, uint32_t* out = nullptr) {
    const int rounds = 256 * 8 / NL;                 // 16 MB per workgroup
    const size_t lds = (size_t)DEPTH * NL * 8192;
    if (lds > 160 * 1024) return;
    hipFuncSetAttribute((const void*)k2<NL, DEPTH, TOUCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t per_rep = (size_t)8 * (32 / share) * 17 * 1024 * 1024;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        const size_t off = (per_rep * rep) % (bufsz - per_rep);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k2<NL, DEPTH, TOUCH>), dim3(256), dim3(576), lds, 0, src, rounds, ld, share, off, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)NL * 8192 * rounds;
    printf("cold, %2d WGs/XCD share a stream, touch-ahead %d, round %3d KB, %d in flight = %3d KB: %7.3f us per 64 KB  %6.1f GB/s per CU  (HBM side %5.2f TB/s)\n", share, TOUCH, NL * 8,
           DEPTH - 1, (DEPTH - 1) * NL * 8, best * 1e3 / rounds * 8 / NL, bytes / (best * 1e-3) / 1e9, bytes * 8 * (32 / share) / (best * 1e-3) / 1e12);
}

template <int MODE>
void run(const char* src, uint32_t* out, int blocks, int ld, int kwrap, size_t xcd_stride, const char* what) {
    const int rounds = 800;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 131072, 0, src, out, rounds, ld, kwrap, xcd_stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = 65536.0 * rounds;
    printf("%-40s ld=%5d blocks=%3d: %7.3f us/round  %6.1f GB/s per CU  %6.2f TB/s total  (%5.1f B/clk/CU at 2.1 GHz)\n", what, ld, blocks,
           best * 1e3 / rounds, bytes / (best * 1e-3) / 1e9, bytes * blocks / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 2.1e9);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t total = 64u << 20;
    char* src; uint32_t* out;
    hipMalloc(&src, total + (1u << 20)); hipMemset(src, 1, total + (1u << 20)); hipMalloc(&out, 4096 * 4);
    for (int ld : {2048, 8192}) {
        const int kwrap = ld / 128;
        const size_t xs = (size_t)512 * ld;           // one A + one B panel per XCD: 1 MB (ld = 2048) / 4 MB (ld = 8192: the whole L2)
        for (int blocks : {256, 64}) {
            run<0>(src, out, blocks, ld, kwrap, xs, "both panels via LDS-DMA");
            run<1>(src, out, blocks, ld, kwrap, xs, "both panels via global_load -> VGPR");
            run<2>(src, out, blocks, ld, kwrap, xs, "A -> VGPR, B via LDS-DMA");
        }
    }
    hipFree(src);
    const size_t big = (size_t)6 << 30;
    hipMalloc(&src, big); hipMemset(src, 1, big);
    for (int share : {32, 8, 4}) {
        run2<8, 2>(src, 8192, share, big);      // 64 KB rounds, one in flight (the 2-stage 256 x 256 kernel)
        run2<4, 3>(src, 8192, share, big);      // 32 KB rounds, 64 KB in flight
        run2<4, 4>(src, 8192, share, big);      //               96 KB
        run2<4, 5>(src, 8192, share, big);      //              128 KB (all of the LDS)
        run2<2, 9>(src, 8192, share, big);      // 16 KB rounds, 128 KB
        run2<8, 2, 1>(src, 8192, share, big, out);
        run2<8, 2, 2>(src, 8192, share, big, out);
        run2<8, 2, 3>(src, 8192, share, big, out);
        run2<8, 2, 4>(src, 8192, share, big, out);
    }
    return 0;
}
