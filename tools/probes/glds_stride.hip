// probe: does the ROW STRIDE of an LDS-DMA tile load matter? Each wave instruction fetches 8 rows x 128 B (as a GEMM tile
// stage does); rows are `stride` bytes apart. stride = 128 is the contiguous case of glds_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

template <int NL, int DEPTH>
__global__ void k(const char* __restrict__ src, int rounds, size_t stride, size_t rows_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const size_t row0 = ((size_t)(blockIdx.x % 64) * 8 * NL * nw) % (rows_total - 8 * NL * nw - 8);   // 4-8 blocks share a row panel
    auto issue = [&](int r) {
        char* dst = smem + (r % DEPTH) * (nw * NL * 1024) + wave * NL * 1024;
        const size_t koff = ((size_t)r * 128) % (stride >= 256 ? stride - 128 : 1);   // walk along K inside the row
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const size_t row = row0 + (size_t)(wave * NL + i) * 8 + (lane >> 3);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + row * stride + (stride >= 256 ? koff : 0) + (lane & 7) * 16), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
        }
    };
    for (int r = 0; r < DEPTH - 1; ++r) issue(r);
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        issue(r + DEPTH - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NL) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NL, int DEPTH>
void run(const char* src, int waves, int blocks, size_t stride, size_t rows_total) {
    const int rounds = 48;
    const size_t lds = (size_t)DEPTH * waves * NL * 1024;
    hipFuncSetAttribute((const void*)k<NL, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NL, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, rounds, stride, rows_total);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<NL, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, rounds, stride, rows_total);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double kb = (double)waves * NL;
    printf("stride=%6zu NL=%d depth=%d waves=%2d blocks=%4d : %6.2f us/round -> %6.1f GB/s per WG, %6.2f TB/s total\n", stride, NL, DEPTH, waves, blocks,
           ms * 1e3 / rounds, kb * 1024 / (ms * 1e-3 / rounds) / 1e9, kb * 1024 * blocks / (ms * 1e-3 / rounds) / 1e12);
}

int main() {
    const size_t bytes = 256u << 20;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    for (size_t stride : {(size_t)128, (size_t)1536, (size_t)6144, (size_t)8192, (size_t)6144 + 128}) {
        const size_t rows_total = stride == 128 ? 65536 : (12u << 20) / stride;   // ~12 MB footprint: L2/MALL resident, like a re-read GEMM operand
        for (int blocks : {256, 512}) {
            run<6, 1>(src, 4, blocks, stride, rows_total);
            run<6, 2>(src, 4, blocks, stride, rows_total);
            run<4, 1>(src, 8, blocks, stride, rows_total);
        }
    }
    return 0;
}
