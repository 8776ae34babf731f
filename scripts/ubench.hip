// Developer micro-benchmark: issue cost and dependent latency of the instructions the solver's
// wave-0 code is made of (one wave, shader clock).   hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ inline double mov_dpp(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double rl(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
#define REP 64
#define ITERS 200
__global__ void k(double *out, long long *cyc, double seed) {
  double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0000001;
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = seed + i + threadIdx.x;
  long long t0, t1;
  int t = 0;
#define BEGIN t0 = clock64();
#define END(name) t1 = clock64(); if (threadIdx.x == 0) cyc[t] = t1 - t0; t++; asm volatile("" : "+v"(a)); for (int i = 0; i < 8; i++) asm volatile("" : "+v"(x[i]));
  // 0: dependent f64 fma chain
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++) a = __builtin_fma(a, c, b);
  } END("fma dep")
  // 1: 8 independent fma chains
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], c, b);
  } END("fma indep")
  // 2: dependent f64 add chain
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++) a = a + b;
  } END("add dep")
  // 3: independent adds
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = x[i] + b;
  } END("add indep")
  // 4: dependent butterfly level (2 dpp + add), counts 3 instr per rep
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++) a = a + mov_dpp<0xB1>(a);
  } END("dpp level dep")
  // 5: 8 independent butterfly levels
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = x[i] + mov_dpp<0xB1>(x[i]);
  } END("dpp level indep")
  // 6: row_mirror level independent
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = x[i] + mov_dpp<0x140>(x[i]);
  } END("dpp row_mirror indep")
  // 7: permlane16 swap level independent
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x[i]), __double2loint(x[i]), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x[i]), __double2hiint(x[i]), false, false);
        x[i] = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
      }
  } END("permlane16 level indep")
  // 8: readlane + fma with scalar operand, dependent (lane-broadcast chain)
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++) a = __builtin_fma(rl(a, r & 31), c, b);
  } END("readlane+fma dep")
  // 9: readlane (independent of chain) feeding independent fmas
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_fma(rl(a, r * 8 + i), c, x[i]);
  } END("readlane indep")
  // 10: dependent mul
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++) a = a * c;
  } END("mul dep")
  // 11: v_cndmask pair + mul (mask at use) independent
  BEGIN for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = (threadIdx.x < 31 ? x[i] : 0.0) * c;
  } END("cndmask+mul indep")
  double s = a;
  for (int i = 0; i < 8; i++) s += x[i];
  out[threadIdx.x] = s;
}
// ---- the block of the solver's two-loop recursion, operands in registers, no memory traffic
__device__ inline void swap16(double v, double &x, double &y) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]);
  y = __hiloint2double(b[1], a[1]);
}
__device__ inline double div_by_rcp(double a, double b, double y) {
  double q0 = a * y;
  double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}
__device__ inline int step_of_lane(int l) {
  const int b0 = l & 1, b1 = (l >> 1) & 1, b2 = (l >> 2) & 1, b3 = (l >> 3) & 1;
  return ((b2 ^ b3) << 2) | ((b1 ^ b2) << 1) | (b0 ^ b2);
}
__host__ __device__ constexpr int lane_of_step(int u) {
  return (((u >> 2) & 1) << 2) | ((((u >> 1) ^ (u >> 2)) & 1) << 1) | ((u ^ (u >> 2)) & 1);
}
__device__ __forceinline__ double sum8t(const double (&v)[8], int lane) {
  asm volatile("" : "+v"(lane));
  const bool k0 = ((lane ^ (lane >> 2)) & 1) != 0;
  const bool k1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
  const bool k2 = (((lane >> 2) ^ (lane >> 3)) & 1) != 0;
  double w[4], z[2];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double keep = k0 ? v[2 * k + 1] : v[2 * k], send = k0 ? v[2 * k] : v[2 * k + 1];
    w[k] = keep + mov_dpp<0xB1>(send);
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const double keep = k1 ? w[2 * k + 1] : w[2 * k], send = k1 ? w[2 * k] : w[2 * k + 1];
    z[k] = keep + mov_dpp<0x4E>(send);
  }
  double r;
  {
    const double keep = k2 ? z[1] : z[0], send = k2 ? z[0] : z[1];
    r = keep + mov_dpp<0x141>(send);
  }
  r += mov_dpp<0x140>(r);
  double x, y;
  swap16(r, x, y);
  return x + y;
}
__global__ void blockbench(double *out, long long *cyc, double seed, int mode) {
  const int lane = threadIdx.x;
  const bool act = lane < 31;
  double s[8], y[8], coef[7], ys = 3.0 + seed, ri = 1.0 / ys, dreg = seed + lane;
  for (int i = 0; i < 8; i++) { s[i] = seed * (i + 1) + lane; y[i] = seed - i * 0.001 * lane; }
  for (int i = 0; i < 7; i++) coef[i] = 1e-3 * (i + 1);
  long long t0 = clock64();
  for (int it = 0; it < 500; it++) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = (act ? s[q] : 0.0) * dreg;
    double acc = sum8t(v, lane);
    if (mode >= 1) {
      int st = step_of_lane(lane);
      asm volatile("" : "+v"(st));
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const double t = div_by_rcp(acc, ys, ri);
        const double au = rl(t, lane_of_step(u));
        if (u < 7) {
          const double nacc = __builtin_fma(-au, coef[u], acc);
          acc = st > u ? nacc : acc;
        }
        if (mode >= 2) dreg = __builtin_fma(-au * 1e-9, act ? y[u] : 0.0, dreg);
      }
    } else {
      dreg += acc * 1e-9;
    }
  }
  long long t1 = clock64();
  if (lane == 0) cyc[mode] = t1 - t0;
  out[lane] = dreg;
}
// ---- issue cost of memory instructions in a burst (all L2 / LDS hits; one wave)
typedef const double __attribute__((address_space(1))) *gptr_t;
__global__ void membench(const double *buf, double *out, long long *cyc, int stride) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = buf[i];
  __syncthreads();
  gptr_t g = (gptr_t)buf;
  double acc = 0.0;
  long long t0, t1;
  // 0: 16 global loads (64-bit VGPR address each), consumed after the burst
  t0 = clock64();
  for (int it = 0; it < 200; it++) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = g[(size_t)(i * stride + (it & 7)) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i];
  }
  t1 = clock64();
  if (lane == 0) cyc[0] = t1 - t0;
  // 1: the same with scalar base + 32-bit lane offset written as inline asm (saddr form)
  t0 = clock64();
  for (int it = 0; it < 200; it++) {
    double v[16];
    const unsigned off = (unsigned)lane * 8u;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const double *row = buf + (size_t)(i * stride + (it & 7)) * 64; // uniform
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v[i]) : "v"(off), "s"(row) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("" : "+v"(v[i]));
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i];
  }
  t1 = clock64();
  if (lane == 0) cyc[1] = t1 - t0;
  // 2: 16 LDS reads (one double per lane), consumed after the burst
  t0 = clock64();
  for (int it = 0; it < 200; it++) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = lds[((i * 5 + (it & 7)) * 64 + lane) & 4095];
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i];
  }
  t1 = clock64();
  if (lane == 0) cyc[2] = t1 - t0;
  // 3: 16 adds only (to subtract)
  t0 = clock64();
  for (int it = 0; it < 200; it++) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = acc * (i + it);
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i];
  }
  t1 = clock64();
  if (lane == 0) cyc[3] = t1 - t0;
  out[lane] = acc;
}
__global__ void chase(const int *next, int start, int steps, long long *cyc, int *sink) {
  int p = start;
  long long t0 = clock64();
  for (int i = 0; i < steps; i++) p = next[p];
  long long t1 = clock64();
  cyc[0] = t1 - t0;
  *sink = p;
}
__global__ void lds_chase(long long *cyc, int *sink) {
  __shared__ int nx[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) nx[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  int p = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < 1000; i++) p = nx[p];
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  sink[threadIdx.x] = p;
}
static void run_chase(size_t bytes, const char *name) {
  size_t n = bytes / 4;
  std::vector<int> h(n);
  // stride of 4 KB + 64 B so every hop is a new line / page-ish; cycle through the whole buffer
  size_t stride = (4096 + 64) / 4;
  for (size_t i = 0; i < n; i++) h[i] = (int)((i + stride) % n);
  int *d, *sink; long long *cyc;
  hipMalloc(&d, bytes); hipMalloc(&sink, 256); hipMalloc(&cyc, 64);
  hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  int steps = 4000;
  for (int rep = 0; rep < 3; rep++) chase<<<1, 1>>>(d, 0, steps, cyc, sink);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-45s %7.1f cycles per dependent load\n", name, (double)c / steps);
  hipFree(d); hipFree(sink); hipFree(cyc);
}
int main() {
  {
    double *out; long long *cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    for (int mode = 0; mode < 3; mode++) { blockbench<<<1, 64>>>(out, cyc, 1.25, mode); blockbench<<<1, 64>>>(out, cyc, 1.25, mode); }
    hipDeviceSynchronize();
    long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    printf("block: products + transposed 8-way sum          %7.1f cycles per block\n", h[0] / 500.0);
    printf("block: + lane-distributed solve                 %7.1f cycles per block\n", h[1] / 500.0);
    printf("block: + direction update                       %7.1f cycles per block\n", h[2] / 500.0);
  }
  {
    double *buf, *out; long long *cyc;
    hipMalloc(&buf, 8u << 20); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    hipMemset(buf, 0, 8u << 20);
    for (int rep = 0; rep < 2; rep++) membench<<<1, 64>>>(buf, out, cyc, 4);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("burst of 16 global loads (VGPR address) + 16 adds    %7.1f cycles per burst\n", h[0] / 200.0);
    printf("burst of 16 global loads (saddr, inline asm) + adds  %7.1f cycles per burst\n", h[1] / 200.0);
    printf("burst of 16 LDS reads + 16 adds                      %7.1f cycles per burst\n", h[2] / 200.0);
    printf("16 muls + 16 adds                                    %7.1f cycles per burst\n", h[3] / 200.0);
  }
  run_chase(256 << 10, "global load chase, 256 KB (L2)");
  run_chase(16 << 20, "global load chase, 16 MB (MALL?)");
  run_chase(1024u << 20, "global load chase, 1 GB (HBM)");
  {
    int *sink; long long *cyc; hipMalloc(&sink, 256); hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 2; rep++) lds_chase<<<1, 64>>>(cyc, sink);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-45s %7.1f cycles per dependent LDS read\n", "LDS chase", (double)c / 1000);
  }
  double *out; long long *cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64 * 8);
  hipMemset(cyc, 0, 64 * 8);
  for (int rep = 0; rep < 2; rep++) k<<<1, 64>>>(out, cyc, 1.25);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[] = {"fma dep", "fma indep x8", "add dep", "add indep x8", "dpp(quad)+add dep [3 instr]", "dpp(quad)+add indep x8 [3 instr]",
                         "dpp(row_mirror)+add indep x8 [3 instr]", "permlane16_swap+add indep x8 [3-5 instr]", "readlane x2 + fma dep [3 instr]",
                         "readlane x2 + fma indep x8 [3 instr]", "mul dep", "cndmask x2 + mul indep x8 [3 instr]"};
  for (int i = 0; i < 12; i++) printf("%-45s %7.2f cycles per rep\n", names[i], (double)h[i] / (REP * ITERS));
  return 0;
}
