// Developer check (host, OpenMP): the two-phase exp / log / sincos of dftpav_amd/csrc/cr_trig.h (a quick phase with Ziv's rounding test, the
// double-double series behind it) against the accurate phase alone, on N random arguments per range, and how often the quick phase
// hands over.  A quick phase whose error bound were wrong would show here first: one argument in ~1 000 lies within the test's 2^-64
// of a rounding boundary, i.e. ~N / 1 000 of the arguments probe the bound.
//   g++ -O2 -fopenmp -std=c++17 -ffp-contract=off -I dftpav_amd/csrc scripts/cr_quick_check.cpp -o /tmp/cr_quick_check && /tmp/cr_quick_check 200000000
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <omp.h>
static long long g_fall[3];
#pragma omp threadprivate(g_fall)
#define DFTPAV_CR_FALLBACK(which) (++g_fall[which])
#include "cr_trig.h"
using namespace dftpav::crt;
int main(int argc, char **argv) {
  const long long N = argc > 1 ? atoll(argv[1]) : 20000000;
  struct R { int fn; double lo, hi; bool logspace; const char *name; };
  const R ranges[] = {{0, -745.0, 0.0, false, "exp [-745, 0]"},   {0, -40.0, 0.0, false, "exp [-40, 0] (the soft-max weights)"}, {0, -1e-3, 1e-3, false, "exp near 0"},
                      {0, 0.0, 709.0, false, "exp [0, 709]"},     {1, 1.0, 64.0, false, "log [1, 64] (the sums of weights)"},     {1, -700.0, 700.0, true, "log e^[-700, 700]"},
                      {1, 1.0 - 1e-6, 1.0 + 1e-6, false, "log near 1"},
                      {2, -3.2, 3.2, false, "sin / cos [-3.2, 3.2] (junction angles)"}, {2, -1.0e3, 1.0e3, false, "sin / cos [-1e3, 1e3]"},
                      {2, 1.5707, 1.5709, false, "sin / cos near pi/2"},                 {2, -1.0e-4, 1.0e-4, false, "sin / cos near 0"},
                      {2, 1.0e6, 1.0e12, false, "sin / cos [1e6, 1e12] (far trial points)"}};
  int bad_total = 0;
  for (const R &rg : ranges) {
    long long bad = 0, fall = 0;
#pragma omp parallel reduction(+ : bad, fall)
    {
      std::mt19937_64 gen(12345 + 977 * (long long)(&rg - ranges) + 31 * omp_get_thread_num());
      std::uniform_real_distribution<double> u(rg.lo, rg.hi);
      g_fall[0] = g_fall[1] = g_fall[2] = 0;
#pragma omp for schedule(static)
      for (long long i = 0; i < N; i++) {
        double x = u(gen);
        if (rg.logspace) x = exp_cr_impl<false>(x);
        double a, b, a2 = 0.0, b2 = 0.0;
        if (rg.fn == 2) {
          sincos_impl<true>(x, a, a2);
          sincos_impl<false>(x, b, b2);
        } else {
          a = rg.fn == 0 ? exp_cr_impl<true>(x) : log_cr_impl<true>(x);
          b = rg.fn == 0 ? exp_cr_impl<false>(x) : log_cr_impl<false>(x);
        }
        if (std::memcmp(&a, &b, 8) != 0 || std::memcmp(&a2, &b2, 8) != 0) {
          if (bad < 5) std::printf("  MISMATCH %s x = %a: two-phase %a accurate %a\n", rg.name, x, a, b);
          bad++;
        }
      }
      fall += g_fall[rg.fn];
    }
    std::printf("%-40s %lld arguments, %lld mismatches, quick phase handed over %lld times (1 in %.0f)\n", rg.name, N, bad, fall, fall ? (double)N / fall : 0.0);
    bad_total += bad != 0;
  }
  return bad_total;
}
