// Developer microbenchmark (gfx950): ISSUE rate of independent fp64 instructions of one wave alone on its SIMD (eight independent chains,
// so that no link waits for its predecessor), beside fp32.   hipcc --offload-arch=gfx950 -O3 scripts/micro/fp64_rate.hip -o /tmp/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(OP) OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8\n\t"
#define F8(OP) OP " %0, %0, %8, %8\n\t" OP " %1, %1, %8, %8\n\t" OP " %2, %2, %8, %8\n\t" OP " %3, %3, %8, %8\n\t" OP " %4, %4, %8, %8\n\t" OP " %5, %5, %8, %8\n\t" OP " %6, %6, %8, %8\n\t" OP " %7, %7, %8, %8\n\t"
template <int WHICH> __device__ long long run(double &sink, int reps) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, v = 1.0 + 1e-9 * threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < reps; i++) {
    if (WHICH == 0) asm volatile(R8("v_add_f64") R8("v_add_f64") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v));
    if (WHICH == 1) asm volatile(R8("v_mul_f64") R8("v_mul_f64") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v));
    if (WHICH == 2) asm volatile(F8("v_fma_f64") F8("v_fma_f64") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v));
  }
  long long t1 = clock64();
  sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  return t1 - t0;
}
__device__ long long run32(float &sink, int reps) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, v = 1.0f + 1e-6f * threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < reps; i++) asm volatile(F8("v_fma_f32") F8("v_fma_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v));
  long long t1 = clock64();
  sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  return t1 - t0;
}
__global__ void k(double *out, long long *cyc, int reps) {
  double s; float f;
  long long c0 = run<0>(s, reps); out[threadIdx.x] = s;
  long long c1 = run<1>(s, reps); out[threadIdx.x] += s;
  long long c2 = run<2>(s, reps); out[threadIdx.x] += s;
  long long c3 = run32(f, reps); out[threadIdx.x] += f;
  if (threadIdx.x == 0) { cyc[0] = c0; cyc[1] = c1; cyc[2] = c2; cyc[3] = c3; }
}
int main() {
  double *o; long long *c; (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 32);
  const int reps = 4096;
  k<<<1, 64>>>(o, c, reps); k<<<1, 64>>>(o, c, reps);
  long long h[4]; (void)hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
  const double n = 16.0 * reps;
  printf("clock64 ticks per independent wave64 instruction (one wave on its SIMD): v_add_f64 %.2f, v_mul_f64 %.2f, v_fma_f64 %.2f, v_fma_f32 %.2f\n", h[0] / n, h[1] / n, h[2] / n, h[3] / n);
  return 0;
}
