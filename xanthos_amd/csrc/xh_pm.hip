// Penman-Monteith monthly PET on gfx950: one fused kernel over (cell, month).
//
// Replaces xanthos/pet/penman_monteith.py: run_pmpet (:394-477), SetData (:17-99), et_veg (:223-334),
// et_water (:337-361), et_snow (:364-377).  The reference evaluates ~250 whole-array numpy expressions on
// [nlcs, ncell, 12] tensors per simulated year; here each thread owns two consecutive months of one cell, reads
// the six forcings as 16-byte coalesced loads (arrays are [ncell, nmonths], month fastest), evaluates every
// land class in registers against per-class tables staged in LDS, and writes PET once.
// Algorithmic HBM traffic: 6 x 8 B forcing + 8 B PET per cell-month (+ land cover, amortised) = 61.3 B.
//
// Arithmetic follows the reference expression by expression (same association order; compiled with
// -ffp-contract=off) except where a quotient is only ever used as a divisor again: the parallel resistances and their
// caps are carried as reciprocals (1/ra = 1/rc + 1/rr, min -> max), per-class constants are multiplied by their
// host-computed reciprocals, 1/cc is formed as gsum / numerator, rc / (lai fwet) as rc (1/lai) (1/fwet) with 1/lai from a
// table and 1/fwet shared by the classes, and the three quotients that make up a class's ET (:306-327) are put over one
// common denominator (one reciprocal per class instead of three).  Differences from numpy therefore come from the
// exp/log/sqrt implementations, from fdiv() below and from those regroupings -- a few ulp each, 5e-13 in PET overall.
#include <algorithm>

#include "xh_common.h"
#include "xh_math.h"
#include "xh_stage.h"

namespace {

constexpr double LAMBDA1 = 2.46e6;   // penman_monteith.py:76
constexpr double CP = 1006.0;        // :77
constexpr double SIGMA = 4.9e-3;     // :78
constexpr double SIGMA2 = 5.67e-8;   // :79
constexpr double GAMMA = 0.67;       // :80

// rows of the per-class parameter block
enum { V_CL = 0, V_BETA, V_RSLIMIT, V_TOPEN, V_TCLOSE, V_TSPAN, V_VCLOSE, V_VOPEN, V_VSPAN, V_RBLMIN, V_RBLMAX,
       V_RBLSPAN, V_RC, V_INVRC, V_EMISS, V_INVTSPAN, V_INVVSPAN, V_INVBETA, V_INVRSLIMIT, PM_NVEC };

#ifndef XH_PM_LCT_AHEAD
#define XH_PM_LCT_AHEAD 1
#endif

struct PmTablesDev {
    int nlcs, n_lc_years, water_idx, snow_idx, start_year, nyears;
    double wind_pow;                     // (2/10)^0.11 (:99)
    double vec[XH_MAX_LCS][PM_NVEC];     // [class][row]: the rows of one class are contiguous, so its ~19 scalar loads merge into a few wide ones
    double one_m_alpha[XH_MAX_LCS][12], lai[XH_MAX_LCS][12], fc[XH_MAX_LCS][12], inv_lai[XH_MAX_LCS][12];
};

// The per-(class, month) tables are indexed by the thread's month and live in LDS; the per-class vectors are indexed by
// the class loop counter alone, i.e. wave-uniform: they are read straight from the (read-only, restrict) table in global
// memory, which the compiler turns into scalar loads -- as LDS reads they were 85 % of the kernel's 46 M LDS
// wave-instructions, each ~16 cycles of a lone wave's issue.
// (Round 6 tried the other way round once more, VERDICT round 5 item 6: the class block staged in LDS and read with broadcast
// reads loses 5 % -- pm_pet 1.72 -> 1.81 ms --, and with class l + 1 read ahead into registers while class l computes 59 %
// -- 2.73 ms: the 40 extra registers cost the third wave per SIMD.  profiles/round6/pm_class_lds_ab.txt.  Not built in.)
struct PmLds {
    double one_m_alpha[XH_MAX_LCS][12], lai[XH_MAX_LCS][12], fc[XH_MAX_LCS][12], inv_lai[XH_MAX_LCS][12];
};

__device__ __forceinline__ int days_in_month(int year, int moy) {
    const int d = (moy == 1) ? 28 : ((moy == 3 || moy == 5 || moy == 8 || moy == 10) ? 30 : 31);
    const bool leap = (year % 4 == 0 && year % 100 != 0) || (year % 400 == 0);   // calendar.isleap (:57)
    return d + ((moy == 1 && leap) ? 1 : 0);
}

// a / b as a * (1 / b), the reciprocal from v_rcp_f64 refined by Newton steps (see frcp): within a few ulp of the IEEE quotient
// for the magnitudes that occur here (no scaling for operands near the exponent limits; b == 0 gives NaN, not an
// infinity -- every denominator below is guarded or strictly positive) in 6 instructions instead of the ~13 of the
// correctly rounded sequence.  The kernel is bound by its divisions (~145 per cell-month as written in the reference,
// ~50 after the regroupings above): 4.9 ms -> 3.7 ms -> 3.1 ms per 67,420 x 600 launch.  PET still agrees with numpy to 5e-13 relative (exp / log dominate; tolerance 1e-6).
// Round 4: ONE Newton step (two until then).  v_rcp_f64 delivers ~26 bits, one step squares the error: ~1e-15 relative per
// quotient, PET 8.6e-12 against numpy on the bench world (7.9e-12 with two steps) and every PM test and fuzz case unchanged
// in outcome, for 3.2 % of the kernel's time (1.95 -> 1.89 ms).
__device__ __forceinline__ double frcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
#ifdef XH_PM_RCP2      // the second step (round 1 - 3)
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
#endif
    return r;
}
__device__ __forceinline__ double fdiv(double a, double b) { return a * frcp(b); }

struct PmCell {          // per-cell quantities shared by the months a thread handles
    double p;            // air pressure (calc_p :185-188)
};

// The month update in two parts, so that a thread can run the class loop for TWO months at once (k_pm_pet2): everything a
// month contributes that does not depend on the land class ...
struct PmMonth {
    double sx, vpd, rcorr, gcu, fwet, one_m_fwet, inv_fwet, g, rl_term, inv_secs, rho_cp, inv_rr, rs_secs, sig_t4_dz, p_cp, vpd_log, secs_lambda, T, TN, W, dz;
    int moy;
};

// Floating-point contraction (round 5): inside pm_prep and pm_class -- and only there -- a * b + c within ONE expression is
// formed as an fma (`#pragma clang fp contract(on)`, decided by the front end per expression), although the library is built
// with -ffp-contract=off.  Both PM kernels inline these two functions, so they keep producing the same bits as each other
// (the fed order runs the paired kernel, the stage-by-stage order the other one; the suites compare them bit for bit); the
// accumulation over the classes outside stays two rounded operations like the reference's `arr *= lct; np.sum`.  Whole-file
// contraction was measured first (profiles/round5/contract_ab.txt: pm_pet 1.761 -> 1.705 ms) and broke exactly that equality.
#ifndef XH_PM_CONTRACT
#define XH_PM_CONTRACT 1
#endif
__device__ __forceinline__ PmMonth pm_prep(const XhExpConsts &K, double p, double T, double TN, double RH, double W, double RS,
                                           double RL, double TP, int moy, double dz) {
#if XH_PM_CONTRACT
#pragma clang fp contract(on)
#endif
    // ---- terms shared by every land class (SetData :83-99, et_veg :226-282)
    const double esx = 6.10588 * xh_exp(fdiv(17.32491 * T, T + 238.102), K);
    const double vap = esx * (RH * 0.01);                             // constant divisors are multiplied by their reciprocal
    const double tk1 = T + 238.1;
    const double sx = fdiv(238.1 * 17.325 * esx, tk1 * tk1);
    const double vpd = esx - vap;
    const double xr = (273.15 + T) * (1.0 / 293.15);
    const double sq = xh_sqrt(xr);                                    // (the library's sqrt without its rescaling for tiny arguments:
    const double rcorr = fdiv(p, 101300.0 * (xr * sq * xh_sqrt(sq)));   //  same bits, 9 instructions fewer each)  pow(x, 1.75) = x * x^(1/2) * x^(1/4)
    const double gcu = 0.00001 * rcorr;
    const double rh = RH > 99.9999 ? 99.9 : RH;                       // calc_rh :205-209
    const double r100 = rh * 0.01;
    const double r2 = r100 * r100, r4 = r2 * r2, r8 = r4 * r4;
    double fwet = rh < 70.0 ? 0.0 : rh;                               // calc_fwet :165-172
    fwet = rh >= 70.0 ? r8 : fwet;
    fwet = rh >= 80.0 ? r8 * r2 : fwet;
    fwet = rh >= 90.0 ? r8 * r4 : fwet;
    fwet = rh >= 95.0 ? r8 * r8 : fwet;
    const double g = (moy == 0) ? 0.0 : 1.6198 * (T - TP);            // calc_g :212-216
    const double t273 = T + 273.0;
    const double t273_2 = t273 * t273;
    const double sig_t4 = SIGMA * (t273_2 * t273_2);
    const double rl_term = RL * 86400.0 * dz;
    const double secs = 86400.0 * dz;
    const double tk = T + 273.15;
    const double rho = fdiv(p, tk * 287.058);
    const double rho_cp = rho * CP;
    const double inv_rr = fdiv(4.0 * SIGMA2 * (tk * tk * tk), rho_cp);    // 1 / rr, rr = rho CP / (4 sigma tk^3) (:268-269): only 1 / rr is used
    const double inv_secs = fdiv(1.0, secs);
    const double log_r100 = log(r100);
    const double one_m_fwet = 1.0 - fwet;
    const double inv_fwet = fwet == 0.0 ? 1.0 : frcp(fwet);          // rc / (lai fwet) below, once for all classes
    // products every class would form again (the association differs from the reference's left-to-right by an ulp)
    const double rs_secs = RS * 86400.0 * dz;                         // oma RS 86400 dz (:150-152)
    const double sig_t4_dz = sig_t4 * dz;                             // sigma T^4 emiss dz
    const double p_cp = p * 0.01 * CP * (1.0 / (LAMBDA1 * 0.622));    // :307
    const double vpd_log = vpd * log_r100;                            // exponent of :323 is vpd_log / beta
    const double secs_lambda = secs * (1.0 / LAMBDA1);                // 86400 dz / lambda1 of :306-327

    PmMonth M;
    M.sx = sx;
    M.vpd = vpd;
    M.rcorr = rcorr;
    M.gcu = gcu;
    M.fwet = fwet;
    M.one_m_fwet = one_m_fwet;
    M.inv_fwet = inv_fwet;
    M.g = g;
    M.rl_term = rl_term;
    M.inv_secs = inv_secs;
    M.rho_cp = rho_cp;
    M.inv_rr = inv_rr;
    M.rs_secs = rs_secs;
    M.sig_t4_dz = sig_t4_dz;
    M.p_cp = p_cp;
    M.vpd_log = vpd_log;
    M.secs_lambda = secs_lambda;
    M.T = T;
    M.TN = TN;
    M.W = W;
    M.dz = dz;
    M.moy = moy;
    return M;
}

// ... and the evapotranspiration of ONE land class in that month (et_veg :223-334, et_water :337-361, et_snow :364-377).
__device__ __forceinline__ double pm_class(const PmLds &L, const PmTablesDev *__restrict__ tab, const XhExpConsts &K, int l,
                                           int water_idx, int snow_idx, double wind_pow, const PmMonth &M) {
#if XH_PM_CONTRACT
#pragma clang fp contract(on)
#endif
    const double sx = M.sx;
    const double vpd = M.vpd;
    const double rcorr = M.rcorr;
    const double gcu = M.gcu;
    const double fwet = M.fwet;
    const double one_m_fwet = M.one_m_fwet;
    const double inv_fwet = M.inv_fwet;
    const double g = M.g;
    const double rl_term = M.rl_term;
    const double inv_secs = M.inv_secs;
    const double rho_cp = M.rho_cp;
    const double inv_rr = M.inv_rr;
    const double rs_secs = M.rs_secs;
    const double sig_t4_dz = M.sig_t4_dz;
    const double p_cp = M.p_cp;
    const double vpd_log = M.vpd_log;
    const double secs_lambda = M.secs_lambda;
    const double T = M.T;
    const double TN = M.TN;
    const double W = M.W;
    const double dz = M.dz;
    const int moy = M.moy;
    double et;
    const double oma = (l == water_idx) ? L.one_m_alpha[0][moy]
                                        : ((l == snow_idx) ? L.one_m_alpha[6][moy] : L.one_m_alpha[l][moy]);
        if (l == snow_idx) {
            // et_snow (:364-377): emissivity 0.85, albedo of land class 6
            const double rnl = sig_t4_dz * 0.85 - rl_term;
            double rn = oma * rs_secs - rnl;
            rn = fmax(rn, 0.0);
            et = (rn * inv_secs) * dz * 0.6 * (1.0 / 2845.0);
            et = fmax(et, 0.0);
        } else if (l == water_idx) {
            // et_water (:337-361): emissivity 0.98, albedo of land class 0
            const double rsn = oma * rs_secs;
            const double rnl = sig_t4_dz * 0.98 - rl_term;
            double rn = rsn - rnl;
            rn = fmax(rn, 0.0);
            const double qt = 0.5 * rsn - (moy <= 5 ? 0.8 : 1.3) * rnl;
            double ax = (rn - qt) * inv_secs;
            ax = fmax(ax, 0.0);
            const double ewetx = (rn * inv_secs) * dz * 0.6 * (1.0 / 2845.0);
            const double wind2 = W * wind_pow;
            const double ewety = fdiv(dz * 86400.0 * (sx * ax + GAMMA * 6.43 * (0.5 + 0.54 * wind2) * vpd),
                                      (sx + GAMMA) * LAMBDA1);
            et = T < -1.0 ? ewetx : ewety;
            et = fmax(et, 0.0);
        } else {
            // et_veg (:223-334)
            const double topen = tab->vec[l][V_TOPEN], tclose = tab->vec[l][V_TCLOSE];
            // The three piecewise-linear factors (calc_mtmin :102-114, calc_vpd :117-129, calc_rtotc :132-145): the linear
            // piece first, then the two plateaus in the reference's order of assignment (no input is NaN here, so one of
            // the three branches always applies and the value the reference starts from is never seen)
            double mtmin = (TN - tclose) * tab->vec[l][V_INVTSPAN];
            mtmin = TN >= topen ? 1.0 : mtmin;
            mtmin = TN <= tclose ? 0.1 : mtmin;
            const double vclose = tab->vec[l][V_VCLOSE], vopen = tab->vec[l][V_VOPEN], inv_vspan = tab->vec[l][V_INVVSPAN];
            const double vlin = (vclose - vpd) * inv_vspan;
            const bool v_lo = vpd <= vopen, v_hi = vpd >= vclose;
            double mvpd = vlin;
            mvpd = v_lo ? 1.0 : mvpd;
            mvpd = v_hi ? 0.1 : mvpd;
            const double gs1 = tab->vec[l][V_CL] * mtmin * mvpd * rcorr;    // :242
            const double rblmin = tab->vec[l][V_RBLMIN], rblmax = tab->vec[l][V_RBLMAX];
            double rtotc = rblmax - tab->vec[l][V_RBLSPAN] * (vclose - vpd) * inv_vspan;      // the reference's association
            rtotc = v_lo ? rblmax : rtotc;
            rtotc = v_hi ? rblmin : rtotc;

            const double rnl = sig_t4_dz * tab->vec[l][V_EMISS] - rl_term;    // calc_a :148-162
            const double rn = oma * rs_secs - rnl;
            const double a = rn * inv_secs;

            const double lai = L.lai[l][moy], fc = L.fc[l][moy];
            const double ac = fc * a;
            const double asoil = (1.0 - fc) * a - g;
            double rtot = rtotc * rcorr;
            rtot = fmin(rtot, 80.0);                                  // one v_min_f64 instead of compare + two selects: no operand
                                                                      // here can be NaN (the loader's nan_to_num, guarded quotients)
            const double rc = tab->vec[l][V_RC], inv_rc = tab->vec[l][V_INVRC], rslimit = tab->vec[l][V_RSLIMIT];
            // Resistances in parallel / capped: only their reciprocals are used below, so they are formed directly:
            // 1 / (x rr / (x + rr)) = 1/x + 1/rr, and min(r, rtot) becomes max(1/r, 1/rtot) (same NaN selection).
            const double inv_rtot = fdiv(1.0, rtot);
            double inv_ra = inv_rc + inv_rr;                          // 1 / (rc rr / (rc + rr))
            inv_ra = fmax(inv_ra, inv_rtot);                          // ra = min(ra, rtot)

            const double gsum = gs1 + inv_rc + gcu;                   // calc_cc :192-197
            // cc (:192-197) and rs = 1 / cc (:285-291) in one step: gsum < 1e-4 -> cc = 1e4; else fwet == 1 or lai < 1e-4 ->
            // cc = 1e-5; else cc = cnum / gsum, and cc == 0 -> rs = 1e5.  fwet == 1 makes cnum zero, so two tests cover the
            // three ways to 1e5.
            const double cnum = inv_rc * (gs1 + gcu) * lai * one_m_fwet;
            double rs = (lai < 0.0001 || cnum == 0.0) ? 100000.0 : fdiv(gsum, cnum);
            rs = gsum < 0.0001 ? 0.0001 : rs;
            rs = fmin(rs, rslimit);

            const double lf = lai * fwet;                             // :296-301
            const double lai_fwet = lf == 0.0 ? 1.0 : lf;
            const double inv_rslimit = tab->vec[l][V_INVRSLIMIT];
            // rc / (lai fwet) as rc (1/lai) (1/fwet): 1/lai from the table, 1/fwet shared by the classes (lai fwet == 0 with
            // lai > 1e-5 means fwet == 0: the reference divides by 1 then)
            const double rhc_raw = rc * (lf == 0.0 ? 1.0 : L.inv_lai[l][moy] * inv_fwet);
            const bool use_rhc = lai > 0.00001 && rhc_raw <= rslimit;   // else the cap (:299-301): rhc = rslimit
            const double rhc = use_rhc ? rhc_raw : rslimit;
            const double inv_rhc = use_rhc ? lai_fwet * inv_rc : inv_rslimit;
            double inv_rhrc = inv_rhc + inv_rr;                       // 1 / (rhc rr / (rhc + rr))
            inv_rhrc = fmax(inv_rhrc, inv_rtot);                      // rhrc = min(rhrc, rtot)

            // the three quotients of :306-327 (canopy evaporation, soil evaporation, transpiration) over one common
            // denominator: one reciprocal instead of three
            // (rh < 70 makes fwet zero and fc == 0 makes ac zero: the reference's two np.where(..., 0, ...) of :309-310 and :328 only
            // ever replace a zero by a zero)
            // Every numerator carries 86400 dz and every denominator lambda1: both leave the class loop as secs_lambda.
            const double rcv = rho_cp * vpd;
            const double m_apres = (sx * ac + rcv * fc * inv_rhrc) * fwet;      // :306-307
            const double x_apres = sx + p_cp * rhc * inv_rhrc;

            const double inv_rasoil = inv_rtot + inv_rr;              // 1 / (rtot rr / (rtot + rr))
            // ewet_soil + esoilpot pow(rh/100, vpd/beta) (:314-323): one numerator, fwet + (1 - fwet) pow(...) of it
            const double m_soil = (sx * asoil + rho_cp * (1.0 - fc) * vpd * inv_rasoil) *
                                  (fwet + one_m_fwet * xh_exp_nonpos(vpd_log * tab->vec[l][V_INVBETA], K));   // vpd >= 0, log(rh / 100) <= 0
            const double x_soil = sx + GAMMA * rtot * inv_rasoil;

            const double m_trans = (sx * ac + rcv * fc * inv_ra) * one_m_fwet;  // :326-327
            const double x_trans = sx + GAMMA * (1.0 + rs * inv_ra);

            const double x_as = x_apres * x_soil;
            const double num = (m_trans * x_as + m_apres * (x_trans * x_soil)) + m_soil * (x_trans * x_apres);
            et = secs_lambda * (num * frcp(x_trans * x_as));
            et = fmax(et, 0.0);
        }
    return et;
}

// One (cell, month): returns PET. lct_cell[l * lct_stride] = land-cover fraction of class l, totpct = their sum (0 -> 0.01).
__device__ __forceinline__ double pm_month(const PmLds &L, const PmTablesDev *__restrict__ tab, const XhExpConsts &K,
                                           int nlcs, int water_idx, int snow_idx, double wind_pow,
                                           double p, double T, double TN, double RH, double W, double RS,
                                           double RL, double TP, int moy, double dz,
                                           const double *__restrict__ lct_cell, int lct_stride, double totpct) {
    const PmMonth M = pm_prep(K, p, T, TN, RH, W, RS, RL, TP, moy, dz);
    double acc = 0.0;
#if XH_PM_LCT_AHEAD
    // the land-cover fraction of class l + 1 is fetched while class l is computed: as written in the reference's order (the
    // fraction read where it is used) the load sat at the end of every iteration with s_waitcnt vmcnt(0) right behind it
    double frac = nlcs > 0 ? lct_cell[0] : 0.0;
    for (int l = 0; l < nlcs; ++l) {
        const double frac_next = lct_cell[(l + 1 < nlcs ? l + 1 : l) * lct_stride];
        const double et = pm_class(L, tab, K, l, water_idx, snow_idx, wind_pow, M);
        const double term = et * frac;                                // arr *= lct (:467)
        acc += term;                                                  // np.sum over classes, in order (:470); 0 + x is x
        frac = frac_next;
    }
#else
    for (int l = 0; l < nlcs; ++l) {
        const double et = pm_class(L, tab, K, l, water_idx, snow_idx, wind_pow, M);
        const double term = et * lct_cell[l * lct_stride];            // arr *= lct (:467)
        acc += term;                                                  // np.sum over classes, in order (:470); 0 + x is x
    }
#endif
    return fdiv(acc, totpct);
}

// Thread <-> (cell, pair of consecutive months). nmonths is a multiple of 12, hence even.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_pm_pet(const PmTablesDev *__restrict__ tab,
                                                const int *__restrict__ lc_of_year, int64_t ncell, int nmonths,
                                                int m_begin, int m_count, const double *__restrict__ tas, const double *__restrict__ tmin,
                                                const double *__restrict__ rhs, const double *__restrict__ wind,
                                                const double *__restrict__ rsds, const double *__restrict__ rlds,
                                                const double *__restrict__ tairprev,
                                                const double *__restrict__ lct, const double *__restrict__ pressure,
                                                double *__restrict__ pet) {
    __shared__ PmLds L;
    const int nlcs = tab->nlcs;
    {
        for (int i = threadIdx.x; i < nlcs * 12; i += blockDim.x) {
            const int l = i / 12, m = i % 12;
            L.one_m_alpha[l][m] = tab->one_m_alpha[l][m];
            L.lai[l][m] = tab->lai[l][m];
            L.fc[l][m] = tab->fc[l][m];
            L.inv_lai[l][m] = tab->inv_lai[l][m];
        }
    }
    __syncthreads();
    const int water_idx = tab->water_idx, snow_idx = tab->snow_idx, start_year = tab->start_year;
    const int n_lc_years = tab->n_lc_years;
    const double wind_pow = tab->wind_pow;
    const XhExpConsts K = xh_exp_consts();
    const int half = m_count >> 1;                  // pairs of months per cell in [m_begin, m_begin + m_count)
    const int64_t total = ncell * (int64_t)half;
    for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = item / half;
        const int m0 = m_begin + (int)(item - c * half) * 2;
        const int64_t off = c * nmonths + m0;
        const double2 T = *reinterpret_cast<const double2 *>(tas + off);
        const double2 TN = *reinterpret_cast<const double2 *>(tmin + off);
        const double2 RH = *reinterpret_cast<const double2 *>(rhs + off);
        const double2 W = *reinterpret_cast<const double2 *>(wind + off);
        const double2 RS = *reinterpret_cast<const double2 *>(rsds + off);
        const double2 RL = *reinterpret_cast<const double2 *>(rlds + off);
        double2 TP;
        if (tairprev) {
            TP = *reinterpret_cast<const double2 *>(tairprev + off);
        } else if (c > 0) {
            TP = *reinterpret_cast<const double2 *>(tas + off - nmonths);   // previous CELL (data_load.py:128-129)
        } else {
            TP = make_double2(0.0, 0.0);
        }
        const int yr_i = m0 / 12;                   // both months are in the same year (m0 even, 12 even)
        const int year = start_year + yr_i;
        const int moy = m0 - yr_i * 12;
        const double *lct_cell = lct + c * (int64_t)nlcs * n_lc_years + lc_of_year[yr_i];
        double totpct = 0.0;
        for (int l = 0; l < nlcs; ++l) {
            const double v = lct_cell[l * n_lc_years];
            totpct = (l == 0) ? v : totpct + v;
        }
        totpct = totpct == 0.0 ? 0.01 : totpct;     // :47
        const double p = pressure[c];               // calc_p (:185-188), once per cell by k_pm_pressure
        double2 out;
        out.x = pm_month(L, tab, K, nlcs, water_idx, snow_idx, wind_pow, p, T.x, TN.x, RH.x, W.x, RS.x, RL.x, TP.x, moy,
                         (double)days_in_month(year, moy), lct_cell, n_lc_years, totpct);
        out.y = pm_month(L, tab, K, nlcs, water_idx, snow_idx, wind_pow, p, T.y, TN.y, RH.y, W.y, RS.y, RL.y, TP.y,
                         moy + 1, (double)days_in_month(year, moy + 1), lct_cell, n_lc_years, totpct);
        *reinterpret_cast<double2 *>(pet + off) = out;
    }
}

// The same thread <-> (cell, two months), but BOTH months go through one class loop: two independent dependency chains per
// wave and the per-class scalars loaded once per pair.  Same functions, same order of operations per month: bit-identical
// to k_pm_pet.  It needs up to 256 registers (two waves per SIMD instead of three), which is what a lone wave wants: beside
// a routing wave of a fed run (xh_fused.hip, mode 1) exactly one PM wave fits per SIMD, and one wave of k_pm_pet there runs
// at ~1/6 of the kernel's own speed for want of independent work (profiles/round4/feed_first_block.txt).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_pm_pet2(const PmTablesDev *__restrict__ tab,
                                                const int *__restrict__ lc_of_year, int64_t ncell, int nmonths,
                                                int m_begin, int m_count, const double *__restrict__ tas, const double *__restrict__ tmin,
                                                const double *__restrict__ rhs, const double *__restrict__ wind,
                                                const double *__restrict__ rsds, const double *__restrict__ rlds,
                                                const double *__restrict__ tairprev,
                                                const double *__restrict__ lct, const double *__restrict__ pressure,
                                                double *__restrict__ pet) {
    __shared__ PmLds L;
    const int nlcs = tab->nlcs;
    for (int i = threadIdx.x; i < nlcs * 12; i += blockDim.x) {
        const int l = i / 12, m = i % 12;
        L.one_m_alpha[l][m] = tab->one_m_alpha[l][m];
        L.lai[l][m] = tab->lai[l][m];
        L.fc[l][m] = tab->fc[l][m];
        L.inv_lai[l][m] = tab->inv_lai[l][m];
    }
    __syncthreads();
    const int water_idx = tab->water_idx, snow_idx = tab->snow_idx, start_year = tab->start_year;
    const int n_lc_years = tab->n_lc_years;
    const double wind_pow = tab->wind_pow;
    const XhExpConsts K = xh_exp_consts();
    const int half = m_count >> 1;
    const int64_t total = ncell * (int64_t)half;
    for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = item / half;
        const int m0 = m_begin + (int)(item - c * half) * 2;
        const int64_t off = c * nmonths + m0;
        const double2 T = *reinterpret_cast<const double2 *>(tas + off);
        const double2 TN = *reinterpret_cast<const double2 *>(tmin + off);
        const double2 RH = *reinterpret_cast<const double2 *>(rhs + off);
        const double2 W = *reinterpret_cast<const double2 *>(wind + off);
        const double2 RS = *reinterpret_cast<const double2 *>(rsds + off);
        const double2 RL = *reinterpret_cast<const double2 *>(rlds + off);
        double2 TP;
        if (tairprev) {
            TP = *reinterpret_cast<const double2 *>(tairprev + off);
        } else if (c > 0) {
            TP = *reinterpret_cast<const double2 *>(tas + off - nmonths);   // previous CELL (data_load.py:128-129)
        } else {
            TP = make_double2(0.0, 0.0);
        }
        const int yr_i = m0 / 12;
        const int year = start_year + yr_i;
        const int moy = m0 - yr_i * 12;
        const double *lct_cell = lct + c * (int64_t)nlcs * n_lc_years + lc_of_year[yr_i];
        double totpct = 0.0;
        for (int l = 0; l < nlcs; ++l) {
            const double v = lct_cell[l * n_lc_years];
            totpct = (l == 0) ? v : totpct + v;
        }
        totpct = totpct == 0.0 ? 0.01 : totpct;     // :47
        const double p = pressure[c];
        const PmMonth A = pm_prep(K, p, T.x, TN.x, RH.x, W.x, RS.x, RL.x, TP.x, moy, (double)days_in_month(year, moy));
        const PmMonth B = pm_prep(K, p, T.y, TN.y, RH.y, W.y, RS.y, RL.y, TP.y, moy + 1, (double)days_in_month(year, moy + 1));
        double acc_a = 0.0, acc_b = 0.0;
        for (int l = 0; l < nlcs; ++l) {
            const double w = lct_cell[l * n_lc_years];
            const double et_a = pm_class(L, tab, K, l, water_idx, snow_idx, wind_pow, A);
            const double et_b = pm_class(L, tab, K, l, water_idx, snow_idx, wind_pow, B);
            acc_a += et_a * w;                      // arr *= lct (:467); np.sum over classes, in order (:470)
            acc_b += et_b * w;
        }
        double2 out;
        out.x = fdiv(acc_a, totpct);
        out.y = fdiv(acc_b, totpct);
        *reinterpret_cast<double2 *>(pet + off) = out;
    }
}

// calc_p (:185-188): air pressure from elevation, a per-cell constant (the pow() cost every (cell, month pair) ~5 % of
// the kernel's instructions when it was evaluated in place)
__global__ void __launch_bounds__(256) k_pm_pressure(int64_t ncell, const double *__restrict__ elev,
                                                     double *__restrict__ pressure) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncell) pressure[c] = 101325.0 * pow(1.0 - 0.0065 * elev[c] / 288.15, 5.2558);
}

int land_cover_index(int year, const std::vector<int> &sorted_years) {   // SetData :33-43
    if (year >= sorted_years.back()) return (int)sorted_years.size() - 1;
    for (size_t i = 0; i < sorted_years.size(); ++i)
        if (sorted_years[i] - year >= -4) return (int)i;
    return (int)sorted_years.size() - 1;
}

}  // namespace

int xh_pm_prepare(xh_ctx *ctx, const xh_pm_tables *t, int64_t ncell, int32_t nmonths, int32_t start_year,
                  int32_t n_lc_years, const int32_t *h_lc_years, int32_t water_idx, int32_t snow_idx, xh_pm_setup *out) {
    XH_REQUIRE(ctx, t && h_lc_years, "xh_pm_pet: NULL argument");
    XH_REQUIRE(ctx, ncell >= 0 && nmonths > 0 && nmonths % 12 == 0, "xh_pm_pet: nmonths must be a positive multiple of 12");
    XH_REQUIRE(ctx, n_lc_years >= 1, "xh_pm_pet: need at least one land-cover year");
    const int nlcs = t->nlcs;
    if (nlcs > XH_MAX_LCS) return xh_fail(ctx, XH_ERR_LIMIT, "xh_pm_pet: nlcs=%d exceeds XH_MAX_LCS=%d", nlcs, XH_MAX_LCS);
    // the reference reads albedo rows 0 and 6 for water and snow regardless of the indices (:361, :377)
    XH_REQUIRE(ctx, nlcs >= 7, "xh_pm_pet: nlcs=%d; the reference indexes land classes 0 and 6, so nlcs >= 7", nlcs);
    XH_REQUIRE(ctx, water_idx >= 0 && water_idx < nlcs && snow_idx >= 0 && snow_idx < nlcs,
               "xh_pm_pet: water_idx/snow_idx out of range");
    out->ncell = ncell;
    out->nmonths = nmonths;
    if (ncell == 0) return XH_OK;
    const int nyears = nmonths / 12;
    PmTablesDev h;
    memset(&h, 0, sizeof(h));
    h.nlcs = nlcs;
    h.n_lc_years = n_lc_years;
    h.water_idx = water_idx;
    h.snow_idx = snow_idx;
    h.start_year = start_year;
    h.nyears = nyears;
    h.wind_pow = pow(2.0 / 10.0, 0.11);
    for (int l = 0; l < nlcs; ++l) {
        h.vec[l][V_CL] = t->cL[l];
        h.vec[l][V_BETA] = t->beta[l];
        h.vec[l][V_RSLIMIT] = t->rslimit[l];
        h.vec[l][V_TOPEN] = t->Tminopen[l];
        h.vec[l][V_TCLOSE] = t->Tminclose[l];
        h.vec[l][V_TSPAN] = t->Tminopen[l] - t->Tminclose[l];
        h.vec[l][V_VCLOSE] = t->VPDclose[l];
        h.vec[l][V_VOPEN] = t->VPDopen[l];
        h.vec[l][V_VSPAN] = t->VPDclose[l] - t->VPDopen[l];
        h.vec[l][V_RBLMIN] = t->RBLmin[l];
        h.vec[l][V_RBLMAX] = t->RBLmax[l];
        h.vec[l][V_RBLSPAN] = t->RBLmax[l] - t->RBLmin[l];
        h.vec[l][V_RC] = t->rc[l];
        h.vec[l][V_INVRC] = 1.0 / t->rc[l];
        h.vec[l][V_EMISS] = t->emiss[l];
        h.vec[l][V_INVTSPAN] = 1.0 / h.vec[l][V_TSPAN];          // reciprocals of per-class constants: the kernel multiplies
        h.vec[l][V_INVVSPAN] = 1.0 / h.vec[l][V_VSPAN];
        h.vec[l][V_INVBETA] = 1.0 / t->beta[l];
        h.vec[l][V_INVRSLIMIT] = 1.0 / t->rslimit[l];
        for (int m = 0; m < 12; ++m) {
            const double lai = t->lai[l * 12 + m], lmin = t->laimin[l * 12 + m], lmax = t->laimax[l * 12 + m];
            double den = exp(-0.5 * lmin) - exp(-0.5 * lmax);           // :257-261
            den = den == 0.0 ? 1.0 : den;
            double fc = (exp(-0.5 * lmin) - exp(-0.5 * lai)) / den;
            fc = fc > 1.0 ? 1.0 : fc;
            h.one_m_alpha[l][m] = 1.0 - t->alpha[l * 12 + m];
            h.lai[l][m] = lai;
            h.fc[l][m] = fc;
            h.inv_lai[l][m] = lai > 0.00001 ? 1.0 / lai : 0.0;      // only read where lai > 1e-5 (:296-301)
        }
    }
    std::vector<int> sorted(h_lc_years, h_lc_years + n_lc_years);
    std::sort(sorted.begin(), sorted.end());
    std::vector<int> lc_of_year(nyears);
    for (int y = 0; y < nyears; ++y) lc_of_year[y] = land_cover_index(start_year + y, sorted);

    void *d_tab = nullptr;
    const size_t tab_bytes = (sizeof(PmTablesDev) + 255) & ~size_t(255);
    const size_t lcy_bytes = (sizeof(int) * nyears + 255) & ~size_t(255);
    // The same tables, years and cell count as the last call, and nobody has used the scratch slot since: what is on the
    // device is what would be uploaded again (a pipeline run step after step; each upload needs a stream synchronisation,
    // i.e. an idle device for a launch latency).  The pressure array is refilled by the first xh_pm_enqueue either way.
    std::vector<char> key(sizeof(h) + sizeof(int) * nyears + sizeof(int64_t));
    memcpy(key.data(), &h, sizeof(h));
    memcpy(key.data() + sizeof(h), lc_of_year.data(), sizeof(int) * nyears);
    memcpy(key.data() + sizeof(h) + sizeof(int) * nyears, &ncell, sizeof(int64_t));
    const bool cached = ctx->scratch[0] && ctx->pm_cache_gen == ctx->scratch_gen[0] && ctx->pm_cache == key &&
                        ctx->scratch_bytes[0] >= tab_bytes + lcy_bytes + sizeof(double) * ncell;
    int rc = XH_OK;
    if (cached) {
        d_tab = ctx->scratch[0];
    } else {
        rc = xh_scratch(ctx, 0, tab_bytes + lcy_bytes + sizeof(double) * ncell, &d_tab);
        if (rc) return rc;
    }
    int *d_lcy = reinterpret_cast<int *>(static_cast<char *>(d_tab) + tab_bytes);
    out->d_pressure = reinterpret_cast<double *>(static_cast<char *>(d_tab) + tab_bytes + lcy_bytes);
    out->pressure_done = false;
    if (!cached) {
        XH_HIP(ctx, hipMemcpyAsync(d_tab, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(ctx, hipMemcpyAsync(d_lcy, lc_of_year.data(), sizeof(int) * nyears, hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // h / lc_of_year are stack/heap locals
        ctx->pm_cache.swap(key);
        ctx->pm_cache_gen = ctx->scratch_gen[0];
    }

    out->d_tab = d_tab;
    out->d_lcy = d_lcy;
    return XH_OK;
}

int xh_pm_enqueue(xh_ctx *ctx, hipStream_t st, xh_pm_setup &s, int m_begin, int m_count, const double *d_tas,
                  const double *d_tmin, const double *d_rhs, const double *d_wind, const double *d_rsds,
                  const double *d_rlds, const double *d_tairprev, const double *d_lct, const double *d_elev,
                  double *d_pet) {
    if (s.ncell == 0 || m_count <= 0) return XH_OK;
    XH_REQUIRE(ctx, m_begin >= 0 && m_begin % 2 == 0 && m_count % 2 == 0 && m_begin + m_count <= s.nmonths,
               "xh_pm_pet: bad month block");
    const int64_t items = s.ncell * (int64_t)(m_count / 2);
    // grid-stride loop with a whole number of passes per thread: a block of months (xh_run_fused) is only ~1.5 passes of
    // the capped grid, and a ragged last pass would leave a quarter of the chip idle for it
    // (threads per workgroup: the caller's choice -- one-wave workgroups fit a SIMD's free registers whatever the other SIMDs
    // of the CU hold: the fillers of a fed run --, default 256)
    const int bsz = (s.block == 64 || s.block == 128) ? s.block : 256;
    const int64_t cap = (int64_t)ctx->prop.multiProcessorCount * 32 * (256 / bsz);
    const int64_t need = (items + bsz - 1) / bsz;
    const int64_t passes = (need + cap - 1) / cap;
    int64_t blocks = (need + passes - 1) / passes;
    xh_span sp = xh_span_begin_on(ctx, "pm_pet", st);
    if (!s.pressure_done) {
        hipLaunchKernelGGL(k_pm_pressure, dim3((unsigned)((s.ncell + 255) / 256)), dim3(256), 0, st, s.ncell, d_elev,
                           s.d_pressure);
        s.pressure_done = true;
    }
    // the caller's choice of variant (xh_pm_setup::paired: the fillers of a fed run), default k_pm_pet
    if (s.paired)
        hipLaunchKernelGGL(k_pm_pet2, dim3((unsigned)blocks), dim3(bsz), 0, st, static_cast<const PmTablesDev *>(s.d_tab),
                           s.d_lcy, s.ncell, s.nmonths, m_begin, m_count, d_tas, d_tmin, d_rhs, d_wind, d_rsds, d_rlds,
                           d_tairprev, d_lct, s.d_pressure, d_pet);
    else
        hipLaunchKernelGGL(k_pm_pet, dim3((unsigned)blocks), dim3(bsz), 0, st, static_cast<const PmTablesDev *>(s.d_tab),
                           s.d_lcy, s.ncell, s.nmonths, m_begin, m_count, d_tas, d_tmin, d_rhs, d_wind, d_rsds, d_rlds,
                           d_tairprev, d_lct, s.d_pressure, d_pet);
    xh_span_end(sp);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" int xh_pm_pet(xh_ctx *ctx, const xh_pm_tables *t, int64_t ncell, int32_t nmonths, int32_t start_year,
                         int32_t n_lc_years, const int32_t *h_lc_years, int32_t water_idx, int32_t snow_idx,
                         const double *d_tas, const double *d_tmin, const double *d_rhs, const double *d_wind,
                         const double *d_rsds, const double *d_rlds, const double *d_tairprev, const double *d_lct,
                         const double *d_elev, double *d_pet) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_tas && d_tmin && d_rhs && d_wind && d_rsds && d_rlds && d_lct && d_elev && d_pet,
               "xh_pm_pet: NULL argument");
    xh_pm_setup s;
    int rc = xh_pm_prepare(ctx, t, ncell, nmonths, start_year, n_lc_years, h_lc_years, water_idx, snow_idx, &s);
    if (rc || ncell == 0) return rc;
    return xh_pm_enqueue(ctx, ctx->stream, s, 0, nmonths, d_tas, d_tmin, d_rhs, d_wind, d_rsds, d_rlds, d_tairprev, d_lct,
                         d_elev, d_pet);
}
