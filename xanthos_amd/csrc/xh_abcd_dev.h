// ABCD month update shared by the runoff kernels (xh_abcd.hip) and the calibration objective (xh_calib.hip).
// Arithmetic follows xanthos/runoff/abcd.py:abcd_dist (:171-228) and set_rain_and_snow (:141-169) term by term.
#pragma once
#include <hip/hip_runtime.h>

#include "xh_math.h"

namespace xh_abcd_dev {

constexpr double TRAIN = 2.5;   // abcd.py:99
constexpr double TSNOW = 0.6;   // abcd.py:100
constexpr double TSPAN = TRAIN - TSNOW;        // the divisor of the mixed rain / snow class (:160, :192), as numpy forms it
constexpr double INV_TSPAN = 1.0 / TSPAN;

struct AbcdPar {
    double a2, b, b_over_a, c, d, d1, m;
    double inv_a2, inv_b, inv_d1;      // reciprocals of the three per-cell divisors of the month update (see finish_par)
};

// The month update divides by 2a, b and d + 1 -- constants of the cell (or calibration member) -- and the division by
// 2a sits on the sequential chain of the march (13 dependent instructions as an IEEE quotient).  The reciprocals are
// taken once per march; quot() then forms x / d as x * (1/d) plus one residual correction, three dependent
// instructions, which gives the correctly rounded quotient (the last step of the IEEE sequence, without its scaling
// for extreme exponents).  A bare x * (1/d) is not good enough here: y = rpt - sqrt(rpt^2 - w b / a) cancels, and one
// ulp in rpt showed up as 1e-11 in soil moisture against numpy.
__device__ __forceinline__ void finish_par(AbcdPar &P, double a) {
    P.a2 = a * 2.0;                                                   // :54-56
    P.b_over_a = P.b / a;
    P.d1 = P.d + 1.0;
    P.inv_a2 = 1.0 / P.a2;
    P.inv_b = 1.0 / P.b;
    P.inv_d1 = 1.0 / P.d1;
}

__device__ __forceinline__ double quot(double x, double d, double inv_d) {
    const double q = x * inv_d;
    return __builtin_fma(__builtin_fma(-d, q, x), inv_d, q);
}

struct AbcdState {
    double snowpack, sm, gw;
};

// The month update is split in two so that a thread can evaluate the state-INDEPENDENT part (rain / snow split,
// melt factor, exp(-PET/b)) for a whole tile of months with instruction-level parallelism, and keep only the short
// state recurrence sequential.  ABCD is a dependent fp64 chain per cell with ~1 wave per SIMD on the chip (67,420
// cells = 1,054 waves), i.e. latency-bound: hoisting exp and the divisions of the split out of the chain is worth
// far more than any memory optimisation.  Operations and their order are exactly those of abcd_dist.
struct AbcdPre {
    double rain, snow, frac, decay, pet;   // decay = exp(-PET/b) (:211)
    int kind;                              // 1 = all rain (melt = pack*m), 2 = mixed (melt = pack*m*frac), 0 = no melt
};

// set_rain_and_snow (:141-169) for one (cell, month): NaN tmin matches no class => rain = snow = 0.
// kind: 1 = all rain (melt = pack*m), 2 = mixed (melt = pack*m*frac), 0 = no melt.
__device__ __forceinline__ void abcd_split(bool snow_on, double precip, double tmin, double &rain, double &snow,
                                           double &frac, int &kind) {
    rain = precip;
    snow = 0.0;
    frac = 0.0;
    kind = 0;
    if (snow_on) {
        const bool allrain = tmin > TRAIN;
        const bool mixed = (tmin <= TRAIN) && (tmin >= TSNOW);
        const bool allsnow = tmin < TSNOW;
        // both quotients have the constant divisor 1.9: reciprocal + one residual correction (quot) gives the correctly
        // rounded quotient in 3 instructions instead of the 12 of a general division
        frac = quot(TRAIN - tmin, TSPAN, INV_TSPAN);
        rain = 0.0;
        if (mixed) {
            snow = quot(precip * (TRAIN - tmin), TSPAN, INV_TSPAN);
            rain = precip - snow;
        }
        if (allrain) rain = precip;
        if (allsnow) snow = precip;
        kind = allrain ? 1 : (mixed ? 2 : 0);
    }
}

__device__ __forceinline__ AbcdPre abcd_pre(const AbcdPar &P, const XhExpConsts &K, bool snow_on, double pet, double precip,
                                            double tmin) {
    AbcdPre r;
    r.pet = pet;
    r.decay = xh_exp(quot(-pet, P.b, P.inv_b), K);                    // :211
    abcd_split(snow_on, precip, tmin, r.rain, r.snow, r.frac, r.kind);
    return r;
}

// State recurrence of abcd_dist (:171-228). `first` = month 0 of a march: no snow-melt term in W (:200-201).
// FASTQ (the calibration marches only, whose objective is held to 1e-9 of the oracle's, not to its bits): the groundwater
// quotient as a bare product with 1 / (d + 1) -- one rounding more than the quotient, 2 of the march's 89 VALU instructions per
// member, cell and month.  (The quotient by 2a stays exact in either form: y = rpt - sqrt(...) cancels.)
template <bool FASTQ = false>
__device__ __forceinline__ void abcd_step(const AbcdPar &P, AbcdState &s, bool snow_on, bool first, const AbcdPre &r,
                                          double &aet, double &q) {
    double snm = 0.0;
    if (snow_on) {
        s.snowpack = s.snowpack + r.snow;                             // :180-183
        const double pm = s.snowpack * P.m;
        // :191-194: pack*m (all rain), pack*m*frac (mixed), 0 (all snow / no class); x * 1.0 is x, bit for bit
        const double mf = r.kind == 2 ? r.frac : 1.0;
        const double melt = pm * mf;
        snm = r.kind == 0 ? 0.0 : melt;
        s.snowpack = s.snowpack - snm;                                // :197
    }
    const double w = first ? r.rain + s.sm : r.rain + s.sm + snm;     // :200-203
    const double rpt = quot(w + P.b, P.a2, P.inv_a2);                 // :206-207
    const double y = rpt - xh_sqrt(rpt * rpt - (w * P.b_over_a));     // :208
    const double sm1 = y * r.decay;                                   // :211
    const double awet = w - y;
    const double c_awet = P.c * awet;
    s.gw = FASTQ ? (s.gw + c_awet) * P.inv_d1 : quot(s.gw + c_awet, P.d1, P.inv_d1);      // :219-221
    double e = y - sm1;                                               // :224-226
    e = (0.0 >= e) ? 0.0 : e;                                         // np.maximum(0, e): NaN e stays NaN
    e = (r.pet <= e || r.pet != r.pet) ? r.pet : e;                   // np.minimum(pet, e): NaN propagates
    s.sm = y - e;                                                     // :227
    aet = e;
    q = (awet - c_awet) + P.d * s.gw;                                 // :228
}

__device__ __forceinline__ void abcd_month(const AbcdPar &P, const XhExpConsts &K, AbcdState &s, bool snow_on, bool first,
                                           double pet, double precip, double tmin, double &aet, double &q) {
    const AbcdPre r = abcd_pre(P, K, snow_on, pet, precip, tmin);
    abcd_step(P, s, snow_on, first, r, aet, q);
}

__device__ __forceinline__ AbcdPar load_par(const double *__restrict__ pars, int row, bool snow_on) {
    const double *p = pars + (int64_t)row * 5;
    AbcdPar P;
    const double a = p[0];
    P.b = p[1] * 1000.0;                                              // :48
    P.c = p[2];
    P.d = p[3];
    P.m = snow_on ? p[4] : 0.0;
    finish_par(P, a);
    return P;
}

}  // namespace xh_abcd_dev
