// Host build of csrc/fastmath.h against the long double libm: prints the largest error in ulps of the double result
// (tests/test_fastmath.py).  Arguments are drawn like the prism kernels form them: corners (XX, YY, ZZ) of cells up to a
// few 1e5 m from the observation, arg4 = R + XX, arg3 = atan2(XX YY, ZZ R), plus log-uniform magnitudes and exact ties.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "fastmath.h"

static double ulp_of(double v)
{
    v = std::fabs(v);
    if (v < 2.2250738585072014e-308) return 4.9406564584124654e-324;
    int e;
    std::frexp(v, &e);
    return std::ldexp(1.0, e - 53);
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-1.0, 1.0), U01(0.0, 1.0);
    tfx::FastMathTables tb{tfx_log_tab, tfx_atan_tab};
    double worst_log = 0, worst_log_abs = 0, worst_atan = 0, worst_atan_abs = 0;
    double wl_arg = 0, wa_y = 0, wa_x = 0;
    for (long k = 0; k < n; ++k) {
        double XX, YY, ZZ;
        const int mode = (int)(k % 4);
        if (mode == 0) { XX = 3e4 * U(rng); YY = 3e4 * U(rng); ZZ = 1e4 * U01(rng) + 0.1; }
        else if (mode == 1) { XX = std::pow(10.0, 6 * U(rng)) * (U(rng) < 0 ? -1 : 1); YY = std::pow(10.0, 6 * U(rng)) * (U(rng) < 0 ? -1 : 1); ZZ = std::pow(10.0, 6 * U(rng)) * (U(rng) < 0 ? -1 : 1); }
        else if (mode == 2) { XX = 100.0 * std::floor(300 * U(rng)); YY = 100.0 * std::floor(300 * U(rng)); ZZ = 50.0 * std::floor(1 + 200 * U01(rng)); }
        else { XX = 2e5 * U(rng); YY = 10.0 * U(rng); ZZ = 1e3 * U(rng); }
        const double R = std::sqrt(XX * XX + YY * YY + ZZ * ZZ);
        const double args[3] = {R + XX, R + YY, std::pow(2.0, 40 * U(rng))};
        for (double a : args) {
            if (!(a > 0)) continue;
            const double got = tfx::fast_log(a, tb);
            const long double want = logl((long double)a);
            // |log| < 1: absolute error against 1 ulp at 1 (the kernels multiply the log by a coordinate: absolute error counts)
            const double scale = std::fabs((double)want) < 1.0 ? 1.1102230246251565e-16 : ulp_of((double)want);
            const double err = (double)fabsl((long double)got - want) / scale;
            if (err > worst_log) { worst_log = err; wl_arg = a; }
            const double rel = (double)fabsl((long double)got - want) / ulp_of((double)want);
            if (rel > worst_log_abs) worst_log_abs = rel;
        }
        const double ys[2] = {XX * YY, ZZ}, xs[2] = {ZZ * R, XX};
        for (int j = 0; j < 2; ++j) {
            const double got = tfx::fast_atan2(ys[j], xs[j], tb);
            const long double want = atan2l((long double)ys[j], (long double)xs[j]);
            const double err = (double)fabsl((long double)got - want) / ulp_of((double)want);
            if (err > worst_atan) { worst_atan = err; wa_y = ys[j]; wa_x = xs[j]; }
            const double ea = (double)fabsl((long double)got - want) / ulp_of(std::fmax(1.0, std::fabs((double)want)));
            if (ea > worst_atan_abs) worst_atan_abs = ea;
        }
    }
    // exact ties / axes / signed zeros against the double libm
    const double sp[][2] = {{1, 1}, {-1, 1}, {1, -1}, {-1, -1}, {0.0, 1}, {-0.0, 1}, {0.0, -1}, {-0.0, -1}, {1, 0.0}, {1, -0.0}, {-1, 0.0},
                            {0.0, 0.0}, {-0.0, -0.0}, {1e-40, 1}, {1, 1e40}, {3, 4e-31}};
    int bad = 0;
    for (auto &p : sp) {
        const double got = tfx::fast_atan2(p[0], p[1], tb), want = std::atan2(p[0], p[1]);
        if (!(std::fabs(got - want) <= ulp_of(want)) || std::signbit(got) != std::signbit(want)) { ++bad; std::printf("special atan2(%g, %g): %a vs %a\n", p[0], p[1], got, want); }
    }
    const double ls[] = {1.0, 2.0, 0.5, 0.0, -1.0, 4.9406564584124654e-324, 1e-310, 1.7976931348623157e308, 0.99999999999999989, 1.0000000000000002};
    for (double a : ls) {
        const double got = tfx::fast_log(a, tb), want = std::log(a);
        const bool same = (std::isnan(got) && std::isnan(want)) || got == want || std::fabs(got - want) <= 1.2e-16 * std::fmax(1.0, std::fabs(want));
        if (!same) { ++bad; std::printf("special log(%a): %a vs %a\n", a, got, want); }
    }
    std::printf("log_ulp %.3f (arg %a) log_rel_ulp %.3f atan2_ulp %.3f (y %a x %a) atan2_abs_ulp %.3f special_bad %d\n", worst_log, wl_arg, worst_log_abs, worst_atan, wa_y, wa_x, worst_atan_abs, bad);
    return 0;
}
