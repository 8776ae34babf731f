// Developer micro-benchmark (one wave, shader clock): dependent latency of the chain forms the reference order's sequential sums
// could take on gfx950 -- v_fmac_f64_dpp row_newbcast (what solver_ref.hip: seq_sum_dpp uses), v_add_f64 on ready operands,
// v_mov_b64_dpp + v_add_f64.     hipcc --offload-arch=gfx950 -O3 scripts/ubench_chain.hip -o /tmp/ubench_chain && /tmp/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define FM(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define FM16 FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
#define MV(K) "v_mov_b64_dpp %" #K ", %16 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
__global__ void k(double *out, long long *cyc, double seed) {
  double v = seed + threadIdx.x * 1e-3, acc = 0.0, one = 1.0;
  const int ITERS = 2000;
  long long t0 = clock64();
  for (int it = 0; it < ITERS; it++) asm volatile("s_nop 1\n\t" FM16 FM16 : "+v"(acc) : "v"(v), "v"(one));
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = (t1 - t0) / ITERS;
  double t[16];
  for (int i = 0; i < 16; i++) t[i] = v + i;
  double a2 = 0.0;
  t0 = clock64();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int i = 0; i < 16; i++) a2 = a2 + t[i];
    asm volatile("" : "+v"(a2));
  }
  t1 = clock64();
  if (threadIdx.x == 0) cyc[1] = (t1 - t0) / ITERS;
  double a3 = 0.0;
  t0 = clock64();
  for (int it = 0; it < ITERS; it++) {
    double m[16];
    asm volatile("s_nop 1\n\t" MV(0) MV(1) MV(2) MV(3) MV(4) MV(5) MV(6) MV(7) MV(8) MV(9) MV(10) MV(11) MV(12) MV(13) MV(14) MV(15)
                 : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5]), "=&v"(m[6]), "=&v"(m[7]), "=&v"(m[8]), "=&v"(m[9]),
                   "=&v"(m[10]), "=&v"(m[11]), "=&v"(m[12]), "=&v"(m[13]), "=&v"(m[14]), "=&v"(m[15])
                 : "v"(v));
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int i = 0; i < 16; i++) a3 = a3 + m[i];
    asm volatile("" : "+v"(a3));
  }
  t1 = clock64();
  if (threadIdx.x == 0) cyc[2] = (t1 - t0) / ITERS;
  out[threadIdx.x] = acc + a2 + a3;
}
int main() {
  double *o; long long *c;
  hipMalloc(&o, 64 * 8); hipMalloc(&c, 8 * 8);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.25);
  long long h[3]; hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
  printf("cycles per 32-term chain, one wave alone: fmac_dpp %lld | add on ready operands %lld | 16 mov_dpp + 32 adds %lld\n", h[0], h[1], h[2]);
  return 0;
}
