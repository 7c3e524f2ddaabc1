// exact_log.h — ln(x) of a binary32 x, CORRECTLY ROUNDED to binary32, identical on host and device.
//
// Why it exists: `cutoff = sqrt(max(9 + 2 ln(opacity), 1e-6))` (src/render/gaussian.wgsl:229-235) feeds the 2DGS
// degeneracy DECISIONS (`|d| < 1e-4`, `extent < 1e-4`, src/render/gaussian_2d.wgsl:49-78,104-132), and `extent` is an
// ill-conditioned cancellation: one ulp of ln(opacity) flips it (round 2: seed 321 of the medium sweep, hardware
// v_log_f32 on the device vs libm in the oracle). WGSL leaves the precision of `log` open, so the arithmetic
// contract pins it to the one value every implementation can agree on: the correctly rounded one.
//
// How: only IEEE binary64 +, -, *, /, fma and integer bit operations (each correctly rounded on x86 and on
// gfx950), in a fixed order, so host build and device give the same bits BY CONSTRUCTION:
//   x = 2^e * m, m in [0.71875, 1.4375);  r = m * c_j  with c_j ~ 1/m from a 32-entry table (10-bit c_j: the product
//   is exact), |r - 1| <= 2^-5;  ln x = e ln2 - ln c_j + 2 atanh(s), s = (r-1)/(r+1) as a double-double quotient;
//   the three leading terms are summed exactly (two-sums), the tail (< 2^-13 of the result) in plain doubles:
//   total error ~2^-65 relative. The (head, tail) pair is then rounded to ODD in binary64 and converted once to
//   binary32 (round-to-odd at 53 bits followed by round-to-nearest at 24 bits = correct rounding of the exact sum).
// That the result IS the correctly rounded ln for EVERY positive binary32 input (2 139 095 039 of them: an error of
// 2^-65 could still straddle a rounding boundary) is checked exhaustively against x87 `logl` (2^-63) by
// scripts/exact_log/check_exhaustive.cpp and on the inputs nearest to a rounding boundary against mpmath at 300 bits
// (scripts/exact_log/check_hard_cases.py); tests/test_device_math_host.py samples it on every CPU run and
// tests/test_gpu_parity.py runs the device build over all inputs.
// Constants: scripts/exact_log/make_table.py (mpmath).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define BGS_XL_HD __host__ __device__ __forceinline__
#else
#define BGS_XL_HD static inline
#endif

namespace bgs {

BGS_XL_HD uint64_t xl_d2u(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u; memcpy(&u, &d, 8); return u;
#endif
}
BGS_XL_HD double xl_u2d(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; memcpy(&d, &u, 8); return d;
#endif
}
BGS_XL_HD double xl_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// s + err == a + b exactly (Knuth; no magnitude precondition)
BGS_XL_HD double xl_two_sum(double a, double b, double& err) {
    const double s = a + b;
    const double bb = s - a;
    err = (a - (s - bb)) + (b - bb);
    return s;
}

struct XlRow { double c, nl_hi, nl_lo; };  // c_j, -ln(c_j) as head + tail

// The double-double (head, tail) whose exact sum is ln(x) to ~2^-65; x positive, finite, non-zero.
BGS_XL_HD void ln_f32_dd(float x, double& head, double& tail) {
    static constexpr XlRow tab[32] = {
        {0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0},
        {0x1.e900000000000p-1, 0x1.788595a3577bap-5, 0x1.e5ef898b67923p-59},
        {0x1.db00000000000p-1, 0x1.333d7f8183f4bp-4, 0x1.a92afc8ef70b1p-58},
        {0x1.cd80000000000p-1, 0x1.a956d3ecade63p-4, 0x1.e5300b12bd55ep-58},
        {0x1.c100000000000p-1, 0x1.0ce7ecdccc28dp-3, -0x1.692a0055dc959p-57},
        {0x1.b500000000000p-1, 0x1.4462b9dc9b3dcp-3, -0x1.629c46c186385p-58},
        {0x1.a980000000000p-1, 0x1.7b0091651528cp-3, 0x1.4069f303518c8p-57},
        {0x1.9f00000000000p-1, 0x1.ae2ca6f672bd4p-3, 0x1.ab5ca9eaa088ap-57},
        {0x1.9480000000000p-1, 0x1.e2a877a6b2c12p-3, -0x1.fa21e3df99430p-58},
        {0x1.8b00000000000p-1, 0x1.09aa572e6c6d4p-2, 0x1.43c2e68684d53p-57},
        {0x1.8180000000000p-1, 0x1.22981fbef797bp-2, -0x1.0b04ac06cebe0p-59},
        {0x1.7880000000000p-1, 0x1.3ac8ca38e5c5fp-2, -0x1.f7de015f253eep-56},
        {0x1.7000000000000p-1, 0x1.522ae0738a3d8p-2, -0x1.8f7e9b38a6979p-57},
        {0x1.6800000000000p-1, 0x1.68ac83e9c6a14p-2, 0x1.a64eadd740178p-58},
        {0x1.6080000000000p+0, -0x1.478cd5959b3d9p-2, -0x1.37e191a12fb48p-58},
        {0x1.5900000000000p+0, -0x1.31871c9544185p-2, 0x1.51acc4c09b379p-60},
        {0x1.5200000000000p+0, -0x1.1c898c16999fbp-2, 0x1.0e5c62aff1c44p-60},
        {0x1.4b00000000000p+0, -0x1.071b85fcd590dp-2, -0x1.d1707f97bde80p-58},
        {0x1.4480000000000p+0, -0x1.e598ed5a87e2fp-3, 0x1.a5e78f4c50659p-58},
        {0x1.3e00000000000p+0, -0x1.bc286742d8cd6p-3, -0x1.4fce744870f55p-58},
        {0x1.3800000000000p+0, -0x1.9525a9cf456b4p-3, -0x1.d904c1d4e2e26p-57},
        {0x1.3200000000000p+0, -0x1.6d60fe719d21dp-3, 0x1.caae268ecd179p-57},
        {0x1.2c80000000000p+0, -0x1.483bccce6e3ddp-3, -0x1.29391fb1b4b22p-57},
        {0x1.2700000000000p+0, -0x1.2266f190a5acbp-3, -0x1.f547bf1809e88p-57},
        {0x1.2200000000000p+0, -0x1.fec9131dbeabbp-4, 0x1.5746b9981b36cp-58},
        {0x1.1d00000000000p+0, -0x1.b78c82bb0eda1p-4, -0x1.0878cf0327e21p-61},
        {0x1.1800000000000p+0, -0x1.6f0d28ae56b4cp-4, 0x1.906d99184b992p-58},
        {0x1.1380000000000p+0, -0x1.2cb0283f5de1fp-4, 0x1.d359a8fde8adep-60},
        {0x1.0f00000000000p+0, -0x1.d276b8adb0b52p-5, -0x1.1e3c53257fd47p-61},
        {0x1.0a80000000000p+0, -0x1.494acc34d911cp-5, -0x1.e295bf491ccc5p-59},
        {0x1.0600000000000p+0, -0x1.7b91b07d5b11bp-6, 0x1.5b602ace3a510p-60},
        {0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0},
    };
    const double LN2_HI = 0x1.62e42fefa3a00p-1;    // 44 significant bits: e * LN2_HI is exact
    const double LN2_LO = -0x1.0ca86c3898d00p-49;

    const uint64_t bits = xl_d2u((double)x);       // a binary32 subnormal is a normal binary64
    int e = (int)(bits >> 52) - 1023;
    const uint32_t j = (uint32_t)(bits >> 47) & 31u;
    double m = xl_u2d((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);  // [1, 2)
    if (j >= 14u) { m *= 0.5; e += 1; }            // [0.71875, 1.4375)
    const XlRow row = tab[j];
    const double f = m * row.c - 1.0;              // exact: 24 x 10-bit product, then a difference near 1
    const double d = f + 2.0;                      // exact (36 significant bits at most)
    const double q = f / d;
    const double ql = xl_fma(-q, d, f) / d;        // s = q + ql to ~2^-105
    const double z = q * q;
    // 2 atanh(s) = 2 s + s z (2/3 + 2 z/5 + 2 z^2/7 + ...), z < 2^-12: the next term is below 2^-72 of the head
    double p = 0x1.3b13b13b13b14p-3;
    p = xl_fma(p, z, 0x1.745d1745d1746p-3);
    p = xl_fma(p, z, 0x1.c71c71c71c71cp-3);
    p = xl_fma(p, z, 0x1.2492492492492p-2);
    p = xl_fma(p, z, 0x1.999999999999ap-2);
    p = xl_fma(p, z, 0x1.5555555555555p-1);
    const double t = (q * z) * p;
    const double E = (double)e;
    double e1, e2;
    const double s1 = xl_two_sum(E * LN2_HI, row.nl_hi, e1);
    head = xl_two_sum(s1, 2.0 * q, e2);
    tail = (e1 + e2) + ((xl_fma(E, LN2_LO, row.nl_lo) + 2.0 * ql) + t);
}

// Correctly rounded binary32 ln(x). ln(+0) = ln(-0) = -inf, ln(x < 0) = NaN, ln(inf) = inf, ln(NaN) = NaN.
BGS_XL_HD float ln_f32_cr(float x) {
    if (x != x) return x;
    if (x == 0.0f) return -__builtin_inff();
    if (x < 0.0f) return __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    double head, tail, err;
    ln_f32_dd(x, head, tail);
    const double s = xl_two_sum(head, tail, err);
    uint64_t sb = xl_d2u(s);
    // round to odd: if the sum was inexact and the nearest double is even, take its odd neighbour on the side of
    // the true value; a binary32 conversion of a round-to-odd binary64 is the correct rounding of the exact sum
    if (err != 0.0 && (sb & 1ull) == 0ull) sb += ((err > 0.0) == (s > 0.0)) ? 1ull : ~0ull;
    return (float)xl_u2d(sb);
}

// what bgs_selftest_ln_f32 sums over a range of inputs (order-independent, wrap-around): host and device agree on
// the sum iff (overwhelmingly) they agree on every result
BGS_XL_HD uint64_t ln_selftest_mix(uint32_t in_bits, uint32_t out_bits) {
    uint64_t v = ((uint64_t)in_bits << 32 | out_bits) * 0x9E3779B97F4A7C15ull;
    v ^= v >> 29;
    return v * 0xBF58476D1CE4E5B9ull;
}

}  // namespace bgs
