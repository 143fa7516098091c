// probe: what the memory system makes of the residual GEMM's epilogue burst on its own (no GEMM in front): out = in + 1 over an [8192 x 768] fp32
// matrix = 25.2 MB read + 25.2 MB written, cold buffers (rotating through 16 pairs = 800 MB > L2 + MALL),
//   stream : 16 B per thread, flat, as many workgroups as the matrix has 1024-float4 pieces (the shape a memcpy-like kernel would use)
//   tile   : 256 workgroups x 512 threads, one 256 x 96 tile each, the wave / lane -> (row, 4 columns) map of the GEMM's register epilogue
//            (wave = 64 rows x 48 columns; a lane owns row l15 of each 16-row block and 4 consecutive columns per 16-column tile)
//   tile8  : the same with the paired-column layout (8 consecutive columns per lane: two float4)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/burst_floor tools/probes/burst_floor.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int M = 8192, N = 768;

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, float4* __restrict__ out, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { float4 v = in[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; out[i] = v; }
}

template <int PAIR>
__global__ __launch_bounds__(512) void k_tile(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const int tm = blockIdx.x / 8, tn = blockIdx.x % 8, wr = wave >> 1, wc = wave & 1;
    const int m0 = tm * 256 + wr * 64 + l15, n0 = tn * 96 + wc * 48;
    float4 v[4][3];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            const int col = PAIR && ni < 2 ? g * 8 + ni * 4 : ni * 16 + g * 4;      // paired: tiles 0 / 1 side by side, the third plain
            v[mi][ni] = *reinterpret_cast<const float4*>(in + (size_t)(m0 + mi * 16) * N + n0 + col);
        }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            const int col = PAIR && ni < 2 ? g * 8 + ni * 4 : ni * 16 + g * 4;
            float4 w = v[mi][ni]; w.x += 1.f; w.y += 1.f; w.z += 1.f; w.w += 1.f;
            *reinterpret_cast<float4*>(out + (size_t)(m0 + mi * 16) * N + n0 + col) = w;
        }
}

int main() {
    constexpr int NB = 16;
    const size_t bytes = (size_t)M * N * 4;
    std::vector<float*> in(NB), out(NB);
    for (int i = 0; i < NB; ++i) { (void)hipMalloc(&in[i], bytes); (void)hipMalloc(&out[i], bytes); (void)hipMemset(in[i], 0, bytes); (void)hipMemset(out[i], 0, bytes); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* what, auto launch) {
        float best = 1e9f, sum = 0.f; int cnt = 0;
        for (int rep = 0; rep < 48; ++rep) {
            const int b = rep % NB;
            (void)hipEventRecord(e0);
            launch(in[b], out[b]);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= NB) { best = ms < best ? ms : best; sum += ms; ++cnt; }
        }
        printf("%-8s: best %6.2f us  mean %6.2f us   %5.2f TB/s (read + write, mean)\n", what, best * 1e3, sum / cnt * 1e3, 2.0 * bytes / (sum / cnt * 1e-3) / 1e12);
    };
    const int n4 = M * N / 4;
    time("stream", [&](float* a, float* b) { hipLaunchKernelGGL(k_stream, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    time("tile", [&](float* a, float* b) { hipLaunchKernelGGL(k_tile<0>, dim3(256), dim3(512), 0, 0, a, b); });
    time("tile8", [&](float* a, float* b) { hipLaunchKernelGGL(k_tile<1>, dim3(256), dim3(512), 0, 0, a, b); });
    time("stream", [&](float* a, float* b) { hipLaunchKernelGGL(k_stream, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    return 0;
}
