// probe: cost of one "issue N LDS-DMA loads -> s_waitcnt vmcnt(0) -> s_barrier" round per workgroup, vs N (loads per wave),
// waves per workgroup and ring depth (loads left in flight across the barrier). Data is L2/MALL resident (16 MB window).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/glds_rate tools/probes/glds_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

template <int NL, int DEPTH>   // NL loads (1 KB each) per wave per round; DEPTH rounds in flight (1 = wait for the round just issued)
__global__ void k(const char* __restrict__ src, long long* out, int rounds, size_t window) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, nw = blockDim.x >> 6;
    const char* base = src + ((size_t)blockIdx.x * 65536) % (window / 2 - (1u << 20)) + tid * 16;
    auto issue = [&](int r) {
        char* dst = smem + (r % DEPTH) * (nw * NL * 1024) + wave * NL * 1024;
        const char* s = base + ((size_t)r * nw * NL * 1024) % (window / 2);
#pragma unroll
        for (int i = 0; i < NL; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (size_t)(wave * NL + i) * 1024), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
    };
    for (int r = 0; r < DEPTH - 1; ++r) issue(r);
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        issue(r + DEPTH - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NL) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int NL, int DEPTH>
void run(const char* src, long long* out, int waves, int blocks, size_t window) {
    const int rounds = 400;
    const size_t lds = (size_t)DEPTH * waves * NL * 1024;
    if (lds > 160 * 1024) return;
    hipFuncSetAttribute((const void*)k<NL, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NL, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, out, rounds, window);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NL, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, out, rounds, window);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double kb = (double)waves * NL;
    printf("NL=%2d depth=%d waves=%2d blocks=%4d : %7.0f clk/round (s_memtime, 100 MHz ticks x?) %7.2f us/round  %5.1f KB/round/WG  -> %6.1f GB/s per WG, %6.2f TB/s total\n",
           NL, DEPTH, waves, blocks, avg / rounds, ms * 1e3 / rounds, kb, kb * 1024 / (ms * 1e-3 / rounds) / 1e9, kb * 1024 * blocks / (ms * 1e-3 / rounds) / 1e12);
}

int main() {
    const size_t window = 16u << 20;
    char* src; long long* out;
    hipMalloc(&src, window + (1u << 20)); hipMemset(src, 1, window + (1u << 20)); hipMalloc(&out, 4096 * 8);
    for (int blocks : {256, 512}) {
        for (int waves : {4, 8, 16}) {
            run<1, 1>(src, out, waves, blocks, window);
            run<2, 1>(src, out, waves, blocks, window);
            run<4, 1>(src, out, waves, blocks, window);
            run<6, 1>(src, out, waves, blocks, window);
            run<8, 1>(src, out, waves, blocks, window);
            run<4, 2>(src, out, waves, blocks, window);
            run<6, 2>(src, out, waves, blocks, window);
            run<4, 3>(src, out, waves, blocks, window);
        }
    }
    return 0;
}
