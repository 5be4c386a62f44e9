// Which XCD does a workgroup run on?  s_getreg_b32 HW_REG_XCC_ID (id 20, bits 3:0) per block of a 64-block grid.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/xcc_id tools/probes/xcc_id.hip && /tmp/xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  int h[256];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    printf("launch %d:", rep);
    for (int b = 0; b < 64; ++b) printf(" %d", h[b]);
    printf("\n");
  }
  return 0;
}
