// orl_log(): natural logarithm, bit-identical to glibc 2.35's log() as CPython's
// math.log reaches it on x86-64 CPUs with FMA+AVX2 (the `__log_fma` ifunc target).
//
// Why this exists: the reference's only transcendental is the log inside
// random.expovariate (optical_rl_gym/envs/rmsa_env.py:548-553 ->
// /usr/lib/python3.10/random.py expovariate: `-_log(1.0 - self.random())/lambd`).
// glibc's log is faithfully, not correctly, rounded, so the device has to walk the
// same table and the same fused/unfused operation sequence to reproduce arrival and
// holding times to the last bit.  The sequence below is the published algorithm
// (glibc sysdeps/ieee754/dbl-64/e_log.c, Szabolcs Nagy's table-driven log: N=128
// subintervals, order-6 polynomial, order-12 polynomial near 1) with the FMA
// contractions GCC chose for the -mfma -mavx2 build spelled out explicitly as fma()
// calls.  Compile every TU that includes this header with -ffp-contract=off so the
// compiler neither adds nor removes a fusion.
//
// Domain handled: finite normal x > 0 (expovariate only ever passes 1-u in
// [2^-53, 1]).  x == 1 returns +0.0 like glibc.  Zero, negatives, subnormals, inf
// and NaN are outside the hot path and are not restated.
//
// tests/test_log_restatement.py pins this against the host libm on >= 10^7 inputs.
#pragma once
#include <stdint.h>
#include "orl_log_data.h"

#if defined(__HIPCC__)
#define ORL_LOG_FN __device__ __forceinline__
#define ORL_LOG_TABLE_QUAL __device__ const
#else
#define ORL_LOG_FN static inline
#define ORL_LOG_TABLE_QUAL static const
#endif

ORL_LOG_TABLE_QUAL double orl_log_poly[5] = ORL_LOG_POLY_INIT;    // A[0..4]
ORL_LOG_TABLE_QUAL double orl_log_poly1[11] = ORL_LOG_POLY1_INIT; // B[0..10]
ORL_LOG_TABLE_QUAL double orl_log_tab[256] = ORL_LOG_TAB_INIT;    // {invc, logc} x 128

ORL_LOG_FN double orl_log_asdouble(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
ORL_LOG_FN uint64_t orl_log_asuint64(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}

ORL_LOG_FN double orl_log(double x) {
  const uint64_t ix = orl_log_asuint64(x);
  // |x - 1| small: 1 - 2^-4 <= x < 1 + 0x1.09p-4  (unsigned wrap compare, as glibc)
  if (ix - 0x3fee000000000000ull < 0x0003090000000000ull) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double* B = orl_log_poly1;
    const double r = x - 1.0;
    const double r2 = r * r;
    const double r3 = r * r2;
    const double a = __builtin_fma(r2, B[3], __builtin_fma(r, B[2], B[1]));
    const double b = __builtin_fma(r2, B[6], __builtin_fma(r, B[5], B[4]));
    const double c = __builtin_fma(r3, B[10], __builtin_fma(r2, B[9], __builtin_fma(r, B[8], B[7])));
    const double q = __builtin_fma(__builtin_fma(c, r3, b), r3, a);
    // split r = rhi + rlo with rhi holding the top 26 bits
    const double t = __builtin_fma(r, 0x1p27, r);
    const double rhi = __builtin_fma(-0x1p27, r, t);
    const double rlo = r - rhi;
    const double rhi2 = rhi * rhi;
    const double hi = __builtin_fma(rhi2, B[0], r);
    double lo = __builtin_fma(rhi2, B[0], r - hi);
    lo = __builtin_fma(B[0] * rlo, r + rhi, lo);
    return hi + __builtin_fma(q, r3, lo);
  }
  // x = 2^k z, z in [OFF, 2 OFF), OFF = 0x3fe6000000000000; table index = top 7 mantissa bits
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
  const double invc = orl_log_tab[2 * i];
  const double logc = orl_log_tab[2 * i + 1];
  const double z = orl_log_asdouble(iz);
  const double* A = orl_log_poly;
  const double kd = (double)k;
  const double r = __builtin_fma(z, invc, -1.0);
  const double w = __builtin_fma(kd, ORL_LOG_LN2HI, logc);
  const double hi = r + w;
  const double lo = __builtin_fma(kd, ORL_LOG_LN2LO, (w - hi) + r);
  const double r2 = r * r;
  const double r3 = r * r2;
  const double p = __builtin_fma(__builtin_fma(r, A[4], A[3]), r2, __builtin_fma(r, A[2], A[1]));
  return __builtin_fma(r3, p, __builtin_fma(r2, A[0], lo)) + hi;
}
