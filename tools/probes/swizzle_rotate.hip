#include <hip/hip_runtime.h>
__global__ void k(int* out) {
  int v = threadIdx.x;
  int a = __builtin_amdgcn_ds_swizzle(v, 0xC000 | (0 << 10) | (1 << 5));
  int b = __builtin_amdgcn_ds_swizzle(v, 0xC000 | (1 << 10) | (1 << 5));
  out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
}
int main() {
  int* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) { for (int i = 0; i < 64; ++i) printf("%d ", h[r * 64 + i]); printf("\n"); }
  return 0;
}
