// exp() for the kernels that evaluate it in a loop (ABCD march, calibration march, Penman-Monteith).
//
// The device library's f64 exp is a 13-fma sequence, but the compiler issues its Horner steps as v_fmac_f64 (dst += a*b
// with the coefficient pre-loaded in dst), which destroys the coefficient register: every call re-materialises its ten
// 64-bit coefficients with 19 v_mov_b32 -- 45 % of the call's instructions, ~14 % of an ABCD month.  xh_exp is the SAME
// sequence (same constants, read off the library's code; same order of operations; same overflow / underflow selects),
// so its results are bit-identical to exp(), but the coefficients are passed in as values the compiler cannot see
// through (xh_exp_consts pins them in vector registers once per thread): the Horner steps become three-operand
// v_fma_f64 and nothing is re-materialised inside the loop.
#pragma once
#include <hip/hip_runtime.h>

struct XhExpConsts {
    double c[10];      // c2 .. c11 of the library's polynomial for exp(r) - 1 - r, |r| <= ln2 / 2
};

__device__ __forceinline__ XhExpConsts xh_exp_consts() {
    const unsigned long long bits[10] = {0x3fe000000000000bull, 0x3fc5555555555511ull, 0x3fa55555555502a1ull,
                                         0x3f81111111122322ull, 0x3f56c16c1852b7b0ull, 0x3f2a01a014761f6eull,
                                         0x3efa01997c89e6b0ull, 0x3ec71dee623fde64ull, 0x3e928af3fca7ab0cull,
                                         0x3e5ade156a5dcb37ull};
    XhExpConsts k;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        k.c[i] = __longlong_as_double((long long)bits[i]);
        asm volatile("" : "+v"(k.c[i]));      // opaque from here on: kept in registers, never rebuilt from literals
    }
    return k;
}

__device__ __forceinline__ double xh_exp(double x, const XhExpConsts &K) {
    const double log2e = __longlong_as_double(0x3ff71547652b82fell);
    const double neg_ln2_hi = __longlong_as_double((long long)0xbfe62e42fefa39efull);
    const double neg_ln2_lo = __longlong_as_double((long long)0xbc7abc9e3b39803full);
    const double k = __builtin_rint(x * log2e);
    double r = __builtin_fma(neg_ln2_hi, k, x);
    r = __builtin_fma(neg_ln2_lo, k, r);
    double p = __builtin_fma(K.c[9], r, K.c[8]);
#pragma unroll
    for (int i = 7; i >= 0; --i) p = __builtin_fma(r, p, K.c[i]);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    double e = __builtin_ldexp(p, (int)k);
    e = (1024.0 < x) ? __builtin_inf() : e;       // as the library: guards the integer conversion, NaN falls through
    e = (-1075.0 > x) ? 0.0 : e;
    return e;
}

// exp(x) for x <= 0 where 1e-11 relative is enough (Penman-Monteith's pow(rh / 100, vpd / beta), :323: its result is one
// of three addends of a class's ET, and PET is held to 1e-9): the same reduction, the polynomial two terms shorter
// (degree 9: the dropped terms are < r^10 / 10! = 7e-12 for |r| <= ln2 / 2), no overflow select.
__device__ __forceinline__ double xh_exp_nonpos(double x, const XhExpConsts &K) {
    const double log2e = __longlong_as_double(0x3ff71547652b82fell);
    const double neg_ln2_hi = __longlong_as_double((long long)0xbfe62e42fefa39efull);
    const double neg_ln2_lo = __longlong_as_double((long long)0xbc7abc9e3b39803full);
    const double k = __builtin_rint(x * log2e);
    double r = __builtin_fma(neg_ln2_hi, k, x);
    r = __builtin_fma(neg_ln2_lo, k, r);
    double p = __builtin_fma(K.c[7], r, K.c[6]);
#pragma unroll
    for (int i = 5; i >= 0; --i) p = __builtin_fma(r, p, K.c[i]);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    const double e = __builtin_ldexp(p, (int)k);
    return (-1075.0 > x) ? 0.0 : e;               // also x = -inf (rh = 0)
}

// sqrt for arguments that are zero or of ordinary magnitude (the ABCD month update's rpt^2 - w b / a; Penman-Monteith's
// temperature ratio).  The library's f64 sqrt is v_rsq_f64 + the Goldschmidt / Newton steps below, wrapped in a
// rescaling for arguments below 2^-767 and a class test (22 instructions); its argument here, rpt^2 - w b / a, is zero
// or of ordinary magnitude, so the same steps run bare (10 instructions + the zero / infinity select): identical
// results, correctly rounded, 9 instructions fewer on the march's dependent chain.  Negative -> NaN like sqrt.
__device__ __forceinline__ double xh_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    const double g3 = __builtin_fma(d1, h1, g2);
    return (x == 0.0 || x == __builtin_inf()) ? x : g3;
}
