// Micro-benchmark: cost of random (table-lookup) LDS reads per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 2048
template <int BYTES, int ENTRIES>
__global__ void __launch_bounds__(512) k(double* out, const unsigned* idx, int nidx) {
  __shared__ __attribute__((aligned(16))) char tab[BYTES * ENTRIES];
  for (int i = threadIdx.x; i < BYTES * ENTRIES / 4; i += blockDim.x) ((float*)tab)[i] = i * 1e-6f;
  __syncthreads();
  unsigned k0 = idx[(blockIdx.x * blockDim.x + threadIdx.x) % nidx];
  double acc = 0;
  unsigned kk = k0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      kk = (kk * 1664525u + 1013904223u);
      unsigned e = (kk >> 8) % ENTRIES;
      if (BYTES == 16) { double2 v = *(const double2*)(tab + e * 16); acc += v.x + v.y; }
      if (BYTES == 8) { double v = *(const double*)(tab + e * 8); acc += v; }
      if (BYTES == 4) { float v = *(const float*)(tab + e * 4); acc += v; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int BYTES, int ENTRIES> void run(const char* name, unsigned* didx, int nidx) {
  double* d; (void)hipMalloc(&d, 512 * 512 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<BYTES, ENTRIES>), dim3(512), dim3(512), 0, 0, d, didx, nidx);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<BYTES, ENTRIES>), dim3(512), dim3(512), 0, 0, d, didx, nidx);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double winstr = 512.0 * 512 / 64 * ITERS * 8;          // wave-instructions
  // 512 blocks over 256 CUs: 2 blocks per CU run concurrently (LDS permitting)
  printf("%-34s %8.3f ms  -> %.1f LDS cycles per wave-instr per CU @2.1GHz\n", name, ms, ms * 1e-3 * 2.1e9 / (winstr / 256));
  (void)hipFree(d);
}
int main() {
  const int n = 1 << 16; unsigned* h = (unsigned*)malloc(n * 4); srand(1);
  for (int i = 0; i < n; ++i) h[i] = rand();
  unsigned* didx; (void)hipMalloc(&didx, n * 4); (void)hipMemcpy(didx, h, n * 4, hipMemcpyHostToDevice);
  run<16, 4096>("b128 random, 4096 x 16B (64KB)", didx, n);
  run<16, 1024>("b128 random, 1024 x 16B (16KB)", didx, n);
  run<16, 64>("b128 random, 64 x 16B", didx, n);
  run<16, 16>("b128 random, 16 x 16B", didx, n);
  run<8, 4096>("b64 random, 4096 x 8B", didx, n);
  run<8, 8192>("b64 random, 8192 x 8B (64KB)", didx, n);
  run<4, 8192>("b32 random, 8192 x 4B", didx, n);
  run<16, 1>("b128 broadcast", didx, n);
  return 0;
}
