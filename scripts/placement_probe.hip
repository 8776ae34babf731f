// Developer probe: where do the waves of co-resident 128-thread workgroups land?  (HW_ID: simd, cu, se; XCC_ID)
// hipcc --offload-arch=gfx950 -O2 -o /tmp/probe scripts/placement_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void probe(unsigned *out, int spin) {
  extern __shared__ double lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = hw;
    out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc;
  }
  // stay resident so that the whole grid is placed at once
  double a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0000001 + 0.5;
  if (a == 12345.678) lds[threadIdx.x] = a;
}
int main() {
  const int grid = 1024, T = 128, waves = T / 64;
  unsigned *d;
  hipMalloc(&d, sizeof(unsigned) * grid * waves * 2);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 39 * 1024);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(T), 38 * 1024, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(grid * waves * 2);
  hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
  // key = (xcc, se, cu) -> list of (block, wave, simd)
  std::map<unsigned, std::vector<unsigned>> cus;
  for (int b = 0; b < grid; b++)
    for (int w = 0; w < waves; w++) {
      unsigned hw = h[(b * waves + w) * 2], xcc = h[(b * waves + w) * 2 + 1] & 0xf;
      unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
      cus[key].push_back((b << 8) | (w << 4) | simd);
    }
  printf("distinct CUs: %zu\n", cus.size());
  int shown = 0;
  long hist[4] = {0, 0, 0, 0};
  for (auto &kv : cus) {
    int cnt0[4] = {0, 0, 0, 0};
    for (unsigned e : kv.second)
      if (((e >> 4) & 15) == 0) cnt0[e & 3]++;
    int mx = 0;
    for (int i = 0; i < 4; i++) mx = cnt0[i] > mx ? cnt0[i] : mx;
    hist[mx > 3 ? 3 : mx]++;
    if (shown < 6) {
      printf("xcc %u se %u cu %u:", kv.first >> 16, (kv.first >> 8) & 255, kv.first & 15);
      for (unsigned e : kv.second) printf("  b%u.w%u@simd%u", e >> 8, (e >> 4) & 15, e & 3);
      printf("\n");
      shown++;
    }
  }
  printf("CUs by the largest number of wave-0s sharing one SIMD: 1 -> %ld, 2 -> %ld, 3+ -> %ld (0 -> %ld)\n", hist[1], hist[2], hist[3], hist[0]);
  return 0;
}
