// Developer micro-benchmark (shader clock): does a VALU instruction whose EXEC holds only the first row of 16 lanes take fewer
// cycles on gfx950?  The reference order's sequential sums are chains of v_fmac_f64_dpp of which one lane's result is wanted
// (solver_ref.hip: seq_sum_dpp); at two waves per SIMD the chains of two trajectories share one VALU.  Chains of 32, full EXEC
// against EXEC = 0xffff, one wave per SIMD (256 threads) and two (512).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_exec.hip -o /tmp/ubench_exec && /tmp/ubench_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define FM(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define FM16 FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
__global__ void k(double *out, long long *cyc, double seed) {
  double v = seed + threadIdx.x * 1e-3, acc = 0.0, one = 1.0;
  const int ITERS = 2000;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < ITERS; it++) asm volatile("s_nop 1\n\t" FM16 FM16 : "+v"(acc) : "v"(v), "v"(one));
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[(threadIdx.x >> 6) * 2] = (t1 - t0) / ITERS;
  __syncthreads();
  double acc2 = 0.0;
  t0 = clock64();
  for (int it = 0; it < ITERS; it++)
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffff\n\ts_nop 1\n\t" FM16 FM16 "s_mov_b64 exec, s[20:21]\n\t" : "+v"(acc2) : "v"(v), "v"(one) : "s20", "s21");
  t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[(threadIdx.x >> 6) * 2 + 1] = (t1 - t0) / ITERS;
  out[threadIdx.x] = acc + acc2;
}
int main() {
  double *o; long long *c;
  hipMalloc(&o, 512 * 8); hipMalloc(&c, 16 * 8);
  for (int T : {64, 256, 512}) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(T), 0, 0, o, c, 1.25);
    long long h[16]; hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
    printf("%d threads: cycles per 32-term fmac_dpp chain, per wave: full EXEC", T);
    for (int w = 0; w < T / 64; w++) printf(" %lld", h[2 * w]);
    printf(" | EXEC = 0xffff");
    for (int w = 0; w < T / 64; w++) printf(" %lld", h[2 * w + 1]);
    printf("\n");
  }
  return 0;
}
