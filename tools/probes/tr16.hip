// Probe: semantics of ds_read_b64_tr_b16 (gfx950). Prints, per lane, which LDS element indices it received.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int addr_elems = mode == 0 ? lane * 4 : (lane & 15) * 64 + (lane >> 4) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_elems));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 512);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d (addr = %s)\n", mode, mode == 0 ? "lane*4 elems" : "(lane&15)*64 + (lane>>4)*4 elems");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
