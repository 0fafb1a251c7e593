// ABCD month update shared by the runoff kernels (xh_abcd.hip) and the calibration objective (xh_calib.hip).
// Arithmetic follows xanthos/runoff/abcd.py:abcd_dist (:171-228) and set_rain_and_snow (:141-169) term by term.
#pragma once
#include <hip/hip_runtime.h>

namespace xh_abcd_dev {

constexpr double TRAIN = 2.5;   // abcd.py:99
constexpr double TSNOW = 0.6;   // abcd.py:100

struct AbcdPar {
    double a2, b, b_over_a, c, d, d1, m;
};

struct AbcdState {
    double snowpack, sm, gw;
};

// One month of abcd_dist (:171-228). `first` = month 0 of a march: no snow-melt term in W (:200-201).
__device__ __forceinline__ void abcd_month(const AbcdPar &P, AbcdState &s, bool snow_on, bool first, double pet,
                                           double precip, double tmin, double &aet, double &q) {
    double rain = precip, snm = 0.0;
    if (snow_on) {
        // set_rain_and_snow (:141-169): NaN tmin matches no class => rain = snow = 0
        const bool allrain = tmin > TRAIN;
        const bool mixed = (tmin <= TRAIN) && (tmin >= TSNOW);
        const bool allsnow = tmin < TSNOW;
        const double frac = (TRAIN - tmin) / (TRAIN - TSNOW);
        double snow = 0.0;
        rain = 0.0;
        if (mixed) {
            snow = precip * (TRAIN - tmin) / (TRAIN - TSNOW);
            rain = precip - snow;
        }
        if (allrain) rain = precip;
        if (allsnow) snow = precip;
        s.snowpack = s.snowpack + snow;                               // :180-183
        if (allrain) snm = s.snowpack * P.m;                          // :191
        if (mixed) snm = (s.snowpack * P.m) * frac;                   // :192-193
        s.snowpack = s.snowpack - snm;                                // :197
    }
    const double w = first ? rain + s.sm : rain + s.sm + snm;         // :200-203
    const double rpt = (w + P.b) / P.a2;                              // :206-207
    const double y = rpt - sqrt(rpt * rpt - (w * P.b_over_a));        // :208
    const double sm1 = y * exp(-pet / P.b);                           // :211
    const double awet = w - y;
    const double c_awet = P.c * awet;
    s.gw = (s.gw + c_awet) / P.d1;                                    // :219-221
    double e = y - sm1;                                               // :224-226
    e = (0.0 >= e) ? 0.0 : e;                                         // np.maximum(0, e): NaN e stays NaN
    e = (pet <= e || pet != pet) ? pet : e;                           // np.minimum(pet, e): NaN propagates
    s.sm = y - e;                                                     // :227
    aet = e;
    q = (awet - c_awet) + P.d * s.gw;                                 // :228
}

__device__ __forceinline__ AbcdPar load_par(const double *__restrict__ pars, int row, bool snow_on) {
    const double *p = pars + (int64_t)row * 5;
    AbcdPar P;
    const double a = p[0];
    P.b = p[1] * 1000.0;                                              // :48
    P.c = p[2];
    P.d = p[3];
    P.m = snow_on ? p[4] : 0.0;
    P.a2 = a * 2.0;                                                   // :54-56
    P.b_over_a = P.b / a;
    P.d1 = P.d + 1.0;
    return P;
}

}  // namespace xh_abcd_dev
