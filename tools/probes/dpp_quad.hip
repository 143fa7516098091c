// probe: drop_fields_shared (lane-quad exchange of the dropout mask words with DPP quad_perm broadcasts) == per-element hashing
#include "../../med-ts-llm_amd/csrc/mtl_common.h"
#include <cstdio>
__global__ void probe(uint32_t base, uint32_t* out_a, uint32_t* out_b) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const uint32_t key = 192u + (threadIdx.x >> 6) * 16u + l15, a0 = 32u + g * 4u;
    uint32_t fa[4], fb[4];
    drop_fields_shared(base, a0, key, fa);
    for (int r = 0; r < 4; ++r) fb[r] = drop_field(drop_quad(base, a0 + r, key >> 2), key);
    for (int r = 0; r < 4; ++r) { out_a[threadIdx.x * 4 + r] = fa[r]; out_b[threadIdx.x * 4 + r] = fb[r]; }
}
int main() {
    uint32_t *a, *b, ha[1024], hb[1024];
    hipMalloc(&a, 4096); hipMalloc(&b, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, 0x12345678u, a, b);
    hipMemcpy(ha, a, 4096, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) if (ha[i] != hb[i]) { if (bad < 8) printf("lane %d r %d: shared %04x direct %04x\n", i / 4, i % 4, ha[i], hb[i]); ++bad; }
    printf("mismatches: %d of 1024\n", bad);
    return bad != 0;
}
