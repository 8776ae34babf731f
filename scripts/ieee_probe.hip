// Probe: are fp64 sqrt, division, and mul/add (no contraction) on gfx950 bit-identical to the host's IEEE results?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
__global__ void k(const double* a, const double* b, double* s, double* q, double* r, double* m, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { s[i] = sqrt(a[i]); q[i] = a[i] / b[i]; r[i] = 1.0 / b[i]; m[i] = a[i] * b[i] + a[i]; }
}
int main() {
  const int n = 1 << 22;
  std::mt19937_64 g(7);
  std::vector<double> a(n), b(n);
  for (int i = 0; i < n; i++) {
    double e1 = std::ldexp(1.0, (int)(g() % 80) - 40), e2 = std::ldexp(1.0, (int)(g() % 80) - 40);
    a[i] = (1.0 + (g() >> 11) * 0x1.0p-53) * e1;
    b[i] = (1.0 + (g() >> 11) * 0x1.0p-53) * e2 * ((g() & 1) ? 1 : -1);
  }
  double *da, *db, *ds, *dq, *dr, *dm;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dm, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, ds, dq, dr, dm, n);
  std::vector<double> s(n), q(n), r(n), m(n);
  hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 8, hipMemcpyDeviceToHost);
  long bs = 0, bq = 0, br = 0, bm = 0;
  for (int i = 0; i < n; i++) {
    double hs = std::sqrt(a[i]), hq = a[i] / b[i], hr = 1.0 / b[i];
    volatile double p = a[i] * b[i]; double hm = p + a[i];
    bs += std::memcmp(&hs, &s[i], 8) != 0; bq += std::memcmp(&hq, &q[i], 8) != 0;
    br += std::memcmp(&hr, &r[i], 8) != 0; bm += std::memcmp(&hm, &m[i], 8) != 0;
  }
  printf("n=%d mismatches: sqrt=%ld div=%ld rcp=%ld muladd=%ld\n", n, bs, bq, br, bm);
  return 0;
}
