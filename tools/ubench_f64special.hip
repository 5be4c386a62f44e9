// Micro-benchmark: throughput of the fp64 "special" VALU ops the scaled-probability kernel uses
// (v_ldexp_f64, v_frexp_mant_f64, v_frexp_exp_i32_f64, v_cvt_f64_i32) next to v_fma_f64.
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_f64special tools/ubench_f64special.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int OP>
__global__ void k(double* out, double seed, int n) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  int acc = n;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fma(a[i], 1.0000001, 0.5);
      if (OP == 1) a[i] = __builtin_ldexp(a[i], n);
      if (OP == 2) a[i] = __builtin_amdgcn_frexp_mant(a[i]);
      if (OP == 3) { acc += __builtin_amdgcn_frexp_exp(a[i]); }
      if (OP == 4) { a[i] = (double)(acc + i); }
      if (OP == 5) { acc = __builtin_amdgcn_update_dpp(acc, acc + i, 0x138, 0xf, 0xf, false); }
      if (OP == 6) { acc = max(acc, i) - n; }
    }
    if (OP == 3) for (int i = 0; i < 8; ++i) a[i] += 1.0;      // (+8 adds per 8 frexp_exp: subtracted below)
  }
  double s = acc;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double per_iter) {
  double* d; hipMalloc(&d, 2048 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 1.5, 1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 1.5, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = 2048.0 * 256 / 64, instr = waves * ITERS * 8;
  printf("%-28s %8.3f ms  => %.2f cycles per loop-body instance per SIMD @2.4GHz (%s)\n", name, ms,
         ms * 1e-3 * 2.4e9 * 1024 / instr, per_iter > 1 ? "body has extra ops, see source" : "one instruction");
  hipFree(d);
}
int main() {
  run<0>("v_fma_f64", 1); run<1>("v_ldexp_f64", 1); run<2>("v_frexp_mant_f64", 1); run<3>("v_frexp_exp_i32_f64 + add_f64+add_i32", 3);
  run<4>("v_cvt_f64_i32 (+add_i32)", 2); run<5>("v_mov_dpp (+add_i32)", 2); run<6>("v_max_i32+sub", 2);
  return 0;
}
