// Developer microbenchmark (gfx950): issue-to-issue latency of DEPENDENT fp64 operations of one wave alone on its SIMD -- what a link of
// the reference order's sequential sums costs.   hipcc --offload-arch=gfx950 -O3 scripts/micro/fp64_latency.hip -o /tmp/fp64_latency && /tmp/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define BC(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define BC16 BC(0) BC(1) BC(2) BC(3) BC(4) BC(5) BC(6) BC(7) BC(8) BC(9) BC(10) BC(11) BC(12) BC(13) BC(14) BC(15)
__global__ void k(double *out, long long *cyc, int reps) {
  double acc = threadIdx.x * 1e-3, v = 1.0 + threadIdx.x * 1e-6, one = 1.0;
  long long t0 = clock64();
  for (int i = 0; i < reps; i++) asm volatile("s_nop 1\n\t" BC16 : "+v"(acc) : "v"(v), "v"(one));
  long long t1 = clock64();
  double a2 = acc;
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(v));
  }
  long long t2 = clock64();
  double a3 = a2;
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a3) : "v"(v));
  }
  long long t3 = clock64();
  double b0 = a3, b1 = a3 + 1, b2 = a3 + 2, b3 = a3 + 3; // four independent chains: the issue rate
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(v));
  }
  long long t4 = clock64();
  out[threadIdx.x] = b0 + b1 + b2 + b3;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
  double *o; long long *c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 32);
  const int reps = 4096;
  k<<<1, 64>>>(o, c, reps); k<<<1, 64>>>(o, c, reps);
  long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
  const double n = 16.0 * reps;
  printf("clock64 ticks per dependent link: v_fmac_f64_dpp row_newbcast %.2f, v_add_f64 %.2f, v_fma_f64 %.2f; per v_add_f64 of four independent chains %.2f\n", h[0] / n, h[1] / n, h[2] / n, h[3] / n);
  return 0;
}
