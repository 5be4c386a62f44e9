// Achievable HBM write bandwidth on MI355X for the store patterns the fill kernels use.
//   pattern 0: plain streaming stores, 16 B per lane, consecutive lanes contiguous (1 KiB per wave-instruction)
//   pattern 1: the fills' pattern: every wave writes 1 KiB chunks into P planes, advancing 1 KiB per iteration
//   pattern 2: as 0 with non-temporal stores
// hipcc --offload-arch=gfx950 -O3 -o ubench_store tools/ubench_store.hip && ./ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));

__global__ void k_stream(d2v* __restrict__ p, size_t n, int nt) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const d2v v = {1.5, 2.5};
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    if (nt) __builtin_nontemporal_store(v, p + k);
    else p[k] = v;
  }
}

// W waves per workgroup, each wave owns a contiguous region per plane and writes 1 KiB per iteration and plane
template <int P>
__global__ void k_planes(d2v* __restrict__ p, size_t per_wave /* d2v per wave per plane */, size_t plane /* d2v */) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  d2v* base = p + wave * per_wave + lane;
  const d2v v = {1.5, 2.5};
  for (size_t it = 0; it < per_wave / 64; ++it) {
#pragma unroll
    for (int q = 0; q < P; ++q) base[q * plane + it * 64] = v;
  }
}

int main() {
  const size_t bytes = (size_t)48 << 30;
  d2v* p;
  if (hipMalloc((void**)&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n = bytes / 16;
  for (int pat = 0; pat < 5; ++pat) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (pat == 0 || pat == 2) hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, p, n, pat == 2);
      else {
        const int P = pat == 1 ? 5 : (pat == 3 ? 3 : 10);
        const size_t waves = 8192;                       // 512 workgroups x 16 waves, as the headline bench
        const size_t plane = n / P;
        const size_t per_wave = (plane / waves) & ~(size_t)63;
        if (P == 5) hipLaunchKernelGGL(k_planes<5>, dim3(512), dim3(1024), 0, 0, p, per_wave, plane);
        if (P == 3) hipLaunchKernelGGL(k_planes<3>, dim3(512), dim3(1024), 0, 0, p, per_wave, plane);
        if (P == 10) hipLaunchKernelGGL(k_planes<10>, dim3(512), dim3(1024), 0, 0, p, per_wave, plane);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const char* names[] = {"streaming 16 B/lane", "5 planes, 1 KiB per wave and plane per iteration", "streaming, non-temporal",
                           "3 planes", "10 planes"};
    printf("%-52s %8.3f ms  %7.1f GB/s\n", names[pat], best, bytes / (best * 1e-3) / 1e9);
  }
  return 0;
}
