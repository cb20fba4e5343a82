// fastmath.h - fp64 log and atan2 of the prism-integral kernels (build.hip), table-reduced, division-light.
//
// The prism rows (gravity_field.f90:131-195, magnetic_field.f90:321-457) spend their time in log / atan2 / sqrt of doubles: two
// logs and one atan2 per grid node and observation.  The device libm versions cost ~85 / ~110 VALU instructions each (double-
// double internals, full IEEE division, every special case inline).  These versions take ~20 / ~45:
//
//   log x    x = 2^e m, m in [1, 2); i = top 7 mantissa bits; r = fma(m, c_i, -1) with c_i ~ 1 / (1 + (i + 1/2) / 128), |r| <= 2^-8;
//            log x = e ln2 + (-log c_i) + log1p(r), log1p by a degree-7 Taylor polynomial (truncation < 2^-67).
//   atan2    t = min(|y|, |x|) / max(|y|, |x|) is never formed: with t_i = i / 64 the node below a float estimate of t,
//            u = (mn - t_i mx) / (mx + t_i mn) (one division by Newton iterations on v_rcp_f64), |u| < 2^-6;
//            atan t = atan(t_i) + u - u^3/3 + u^5/5 - u^7/7 + u^9/9; the octant of (y, x) picks one of four tabulated constants
//            (atan(t_i), pi/2 -+ atan(t_i), pi - atan(t_i) as hi + lo) so that no reflection rounds.
//
// Accuracy (tests/test_fastmath.py, host build of this header against the host libm on 1e7 random arguments of the prism
// kernels' ranges): log <= 0.51 ulp of max(|result|, 1 ulp at 1), atan2 <= 1.14 ulp of the result (0.52 ulp of pi/4) - the device libm they replace:
// (<= 1-2 ulp); arguments outside the fast range (zero, denormal, inf, nan, |.| beyond 1e+-30 for atan2) take the libm call.
// tables: math_tables.h (tools/gen_math_tables.py), copied into LDS by the kernels (`FastMathTables`).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#define TFX_HD __host__ __device__ __forceinline__
#define TFX_TABLE_CONST __device__ const
#else
#define TFX_HD inline
#define TFX_TABLE_CONST static const
#endif
#include "math_tables.h"

namespace tfx {

struct FastMathTables {
    const double *logt;    // 3 * TFX_LOG_TAB_N: c_i, -log(c_i) as hi + lo
    const double *atant;   // 8 * TFX_ATAN_TAB_N: the octant-pair constant of node i = i / 64 as hi + lo, four cases (math_tables.h)
};
constexpr int FASTMATH_TABLE_DOUBLES = 3 * TFX_LOG_TAB_N + 8 * TFX_ATAN_TAB_N;

TFX_HD uint64_t fm_bits(double x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return (uint64_t)__double_as_longlong(x);
#else
    uint64_t b;
    std::memcpy(&b, &x, 8);
    return b;
#endif
}
TFX_HD double fm_double(uint64_t b)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __longlong_as_double((long long)b);
#else
    double x;
    std::memcpy(&x, &b, 8);
    return x;
#endif
}

// 1 / d for a normal d well inside the exponent range: hardware estimate + two Newton steps (< 1 ulp)
TFX_HD double fm_rcp(double d)
{
#ifdef __HIP_DEVICE_COMPILE__
    double r = __builtin_amdgcn_rcp(d);
#else
    double r = (double)(1.0f / (float)d);
#endif
    r = std::fma(std::fma(-d, r, 1.0), r, r);
    r = std::fma(std::fma(-d, r, 1.0), r, r);
    return r;
}

// natural logarithm, x > 0
TFX_HD double fast_log(double x, const FastMathTables &tb)
{
    const uint64_t b = fm_bits(x);
    const uint32_t hi = (uint32_t)(b >> 32);
    // zero, negative, denormal, inf, nan: the library call (never on the prism kernels' regular arguments)
    if (hi - 0x00100000u >= 0x7fe00000u) return std::log(x);
    const int e = (int)(hi >> 20) - 1023;
    const int i = (int)((hi >> 13) & 127u);
    const double m = fm_double((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double c = tb.logt[3 * i], Lh = tb.logt[3 * i + 1], Ll = tb.logt[3 * i + 2];
    const double r = std::fma(m, c, -1.0);
    // log1p(r) - r = r^2 (-1/2 + r (1/3 + r (-1/4 + r (1/5 + r (-1/6 + r / 7)))))
    double p = std::fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = std::fma(r, p, 1.0 / 5.0);
    p = std::fma(r, p, -1.0 / 4.0);
    p = std::fma(r, p, 1.0 / 3.0);
    p = std::fma(r, p, -0.5);
    const double r2 = r * r;
    const double ed = (double)e;
    // e ln2_hi + hi(-log c_i) is exact (26-bit constant, hi a multiple of 2^-42): the only rounding that counts is the last add
    const double small = std::fma(r2, p, std::fma(ed, TFX_LN2_LO, Ll));
    return std::fma(ed, TFX_LN2_HI, Lh) + (r + small);
}

// atan2(y, x), IEEE conventions for the regular cases; everything unusual goes to the library
TFX_HD double fast_atan2(double y, double x, const FastMathTables &tb)
{
    const double a = std::fabs(y), bb = std::fabs(x);
    const double mx = a > bb ? a : bb, mn = a > bb ? bb : a;
    if (!(mx > 1e-30 && mx < 1e30)) return std::atan2(y, x);
    // nearest table node to t = mn / mx (a float estimate is enough: the reduction below is exact for ANY node)
#ifdef __HIP_DEVICE_COMPILE__
    const float tq = (float)mn * __builtin_amdgcn_rcpf((float)mx);
#else
    const float tq = (float)mn / (float)mx;
#endif
    // the node BELOW t (not the nearest): q >= 0, so the small results next to a node (t ~ 1/128 against node 1/64) are not the
    // difference of two numbers twice their size - the last addition rounds in ulps of the result (|u| < 2^-6: u^11 / 11 < 2^-69)
    int i = (int)(tq * 64.0f);
    i = i < 0 ? 0 : (i > 64 ? 64 : i);
    const double ti = (double)i * 0.015625;
    const double num = std::fma(-ti, mx, mn), den = std::fma(ti, mn, mx);
    // den + den_lo = mx + ti mn exactly (mx - den is exact: den lies in [mx, 2 mx]); the quotient is corrected against the exact
    // denominator, so u carries the rounding of num (none when mn - ti mx cancels) and of the last fma only
    const double den_lo = std::fma(ti, mn, mx - den);
    const double rd = fm_rcp(den);
    double u = num * rd;
    u = std::fma(std::fma(-den_lo, u, std::fma(-den, u, num)), rd, u);
    const double u2 = u * u;
    double p = std::fma(u2, 1.0 / 9.0, -1.0 / 7.0);
    p = std::fma(u2, p, 1.0 / 5.0);
    p = std::fma(u2, p, -1.0 / 3.0);
    // atan(mn / mx) = atan(t_i) + q, q = u - u^3/3 + ...; the octant of (y, x) selects the constant q is added to or subtracted from
    // (atan(t_i), pi/2 - atan(t_i), pi/2 + atan(t_i), pi - atan(t_i), each as hi + lo): the reflections of rounds 2-4
    // (pi/2 - r, pi - r) cost a rounding each - up to 1.6 ulp - this form has ONE rounding that counts, the last addition
    const bool swap = a > bb, xneg = (fm_bits(x) >> 63) != 0;
    const int k = xneg ? (swap ? 2 : 3) : (swap ? 1 : 0);
    double q = std::fma(u * u2, p, u);
    if (k & 1) q = -q;
    const double *c = tb.atant + 2 * (k * TFX_ATAN_TAB_N + i);
    const double r = c[0] + (q + c[1]);
    return (fm_bits(y) >> 63) ? -r : r;
}

}  // namespace tfx
