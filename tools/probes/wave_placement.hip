// Where do the two wavefronts of a 128-thread workgroup land?  Prints, for a grid shaped like the banded sweep's
// (n workgroups x 2 waves, LDS bytes per workgroup as given), how many workgroups have both waves on one SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip && ./wave_placement [workgroups] [lds_bytes] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>

__global__ void k_where(unsigned* out, int spin) {
  extern __shared__ char lds[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the wave resident for a while so that the whole grid is co-resident as in the real kernel
  long long t0 = clock64();
  while (clock64() - t0 < spin) { }
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[2 * w] = hw;
    out[2 * w + 1] = xcc;
  }
  if (spin < 0) lds[threadIdx.x] = 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int ldsb = argc > 2 ? atoi(argv[2]) : 55000;
  const int threads = argc > 3 ? atoi(argv[3]) : 128;
  const int wpw = threads / 64;
  unsigned* d;
  hipMalloc(&d, sizeof(unsigned) * 2 * n * wpw);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_where), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL(k_where, dim3(n), dim3(threads), ldsb, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n * wpw);
  hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
  int same = 0;
  std::map<unsigned, int> per_cu, per_simd;
  for (int b = 0; b < n; ++b) {
    std::map<unsigned, int> simds;
    for (int w = 0; w < wpw; ++w) {
      const unsigned hw = h[2 * (b * wpw + w)], xcc = h[2 * (b * wpw + w) + 1] & 0xF;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      simds[simd]++;
      per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
      per_simd[(xcc << 18) | (se << 10) | (sh << 6) | (cu << 2) | simd]++;
    }
    for (auto& s : simds) if (s.second > 1) { ++same; break; }
    if (b < 6) {
      printf("workgroup %d:", b);
      for (int w = 0; w < wpw; ++w) printf(" [xcc %u hw_id %08x simd %u cu %u]", h[2 * (b * wpw + w) + 1] & 0xF, h[2 * (b * wpw + w)],
                                           (h[2 * (b * wpw + w)] >> 4) & 3, (h[2 * (b * wpw + w)] >> 8) & 0xF);
      printf("\n");
    }
  }
  std::map<int, int> hist, hist_simd;
  for (auto& c : per_cu) hist[c.second]++;
  for (auto& c : per_simd) hist_simd[c.second]++;
  printf("%d workgroups of %d waves, %d bytes of LDS: %d have two waves on one SIMD; distinct CUs %zu\n", n, wpw, ldsb, same, per_cu.size());
  for (auto& x : hist) printf("  CUs holding %d waves: %d\n", x.first, x.second);
  for (auto& x : hist_simd) printf("  SIMDs holding %d waves: %d\n", x.first, x.second);
  return 0;
}
