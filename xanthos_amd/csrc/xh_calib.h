// Internal interface between the calibration objective (xh_calib.hip) and the device-side differential evolution
// (xh_calib_de.hip).  Not part of the C-ABI.
#pragma once
#include <vector>

#include "xh_common.h"

struct xh_calib_basin {
    int ncell, chunk0, nchunks, pad;
    const double *pet, *pr, *tn, *area;      // [month, cell] forcing of this basin; area may be NULL (mm_per_mth)
    // member-lane layout only: the member-independent rain / snow split of every (month, cell), made once per problem
    double *rain, *snow, *frac;
    int *kind;
};

// A set of basins laid out for the objective kernels: tables and work arrays in ONE device allocation.
struct xh_calib_problem {
    int nbasins = 0, nmonths = 0, spinup = 0, nmembers = 0, npar = 0;
    size_t nchunks = 0;
    // 0: lanes <-> cells, 4 members per thread (any population).  1: lanes <-> members, 16 cells per wave, forcing as
    // scalar loads, no cross-lane sums (populations that fill waves of 64 members: 100 instead of 145 wave-instructions
    // per member-cell-month).  Chosen by xh_calib_problem_plan; the work arrays below are laid out accordingly.
    int member_lanes = 0;
    mutable bool split_done = false;          // member-lane layout: rain / snow split computed (set by the first enqueue)
    xh_calib_basin *d_basins = nullptr;
    int *d_chunk_basin = nullptr;
    double *d_obs = nullptr;                  // [nbasins, nmonths]
    double *d_sm0 = nullptr, *d_gw0 = nullptr;   // [nbasins, nmembers]
    double *d_dec = nullptr;                  // [nchunks, nmembers, 6]
    int *d_cnt = nullptr;                     // [nchunks, nmembers, 6]
    double *d_part = nullptr;                 // [nchunks, nmembers, nmonths]
    double *d_series = nullptr;               // [nbasins, nmembers, nmonths]
    double *d_series_m = nullptr;             // member-lane layout: [nbasins, nmonths, nmembers]
};

// Validates the arguments, fills the host tables and returns the bytes xh_calib_problem_place needs.
int xh_calib_problem_plan(xh_ctx *ctx, int32_t nbasins, const int64_t *h_ncell, int32_t nmonths, int32_t spinup,
                          int32_t nmembers, int32_t npar, const double *const *h_pet_t,
                          const double *const *h_precip_t, const double *const *h_tmin_t,
                          const double *const *h_area, std::vector<xh_calib_basin> &basins,
                          std::vector<int> &chunk_basin, size_t *bytes, int *member_lanes);
// Carves the problem out of `buf` (device, 256-byte aligned, at least `bytes` long) and uploads tables and obs.
int xh_calib_problem_place(xh_ctx *ctx, xh_calib_problem &P, int32_t nmonths, int32_t spinup, int32_t nmembers,
                           int32_t npar, std::vector<xh_calib_basin> &basins, const std::vector<int> &chunk_basin,
                           const double *h_obs, void *buf, int member_lanes);
// member-lane layout: d_series in the [nbasins, nmembers, nmonths] order of the C-ABI (a transpose of d_series_m)
int xh_calib_series_out(xh_ctx *ctx, const xh_calib_problem &P);
// Enqueues one evaluation of every basin's population on the context's stream: d_pars [nbasins, nmembers, npar] ->
// d_ed [nbasins, nmembers] (ED = 1 - KGE).  d_active [nbasins] (may be NULL): basins with 0 are skipped and their
// d_ed entries left untouched.  No host synchronisation.
int xh_calib_enqueue(xh_ctx *ctx, const xh_calib_problem &P, const double *d_pars, const int *d_active, double *d_ed);
