// Micro-benchmark: per-instruction throughput of the fp64 VALU ops the fill kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int OP>
__global__ void k(double* out, double seed) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  double b = seed * 0.5, c = 1.0000001;
  int acc = 0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fma(a[i], c, b);
      if (OP == 1) a[i] = a[i] + b;
      if (OP == 2) a[i] = __builtin_fmax(a[i], b + i);
      if (OP == 3) { acc += (int)a[i]; a[i] += 1.0; }
      if (OP == 4) a[i] = __builtin_amdgcn_fract(a[i]) + c;
      if (OP == 5) a[i] = a[i] * c;
      if (OP == 6) { float f = (float)a[i]; f = __builtin_fmaf(f, 1.0001f, 0.5f); a[i] = f; }
      if (OP == 7) a[i] = (a[i] < b) ? c : a[i];
    }
  }
  double s = acc;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int per_iter) {
  double* d; hipMalloc(&d, 2048 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 1.5);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 1.5);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = 2048.0 * 256 * ITERS * 8 * per_iter;
  printf("%-22s %8.3f ms  %7.2f T lane-ops/s  => %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, ms, ops / ms / 1e9,
         1024 * 2.4e9 / (ops / 64 / (ms * 1e-3)));
  hipFree(d);
}
int main() {
  run<0>("v_fma_f64", 1); run<1>("v_add_f64", 1); run<2>("v_max_f64(+add)", 2); run<3>("cvt_i32_f64(+2)", 3);
  run<4>("v_fract_f64(+add)", 2); run<5>("v_mul_f64", 1); run<6>("cvt f64<->f32 + fma32", 3); run<7>("cmp+cndmask x2", 3);
  return 0;
}
