// Internal stage interface: each stage of the hot path as "prepare once, enqueue on a stream", so that xh_run_fused
// (xh_fused.hip) can pipeline blocks of months over several streams.  The public entry points xh_pm_pet, xh_abcd and
// xh_route_series are the one-stream, whole-series case of the same functions.  Not part of the C-ABI.
#pragma once
#include "xh_common.h"

struct xh_pm_setup {
    const void *d_tab = nullptr;      // PmTablesDev
    const int *d_lcy = nullptr;       // land-cover column of each year
    int64_t ncell = 0;
    int nmonths = 0;
    double *d_pressure = nullptr;     // [ncell] calc_p, filled by the first xh_pm_enqueue
    bool pressure_done = false;
    bool paired = false;              // k_pm_pet2: both months of a thread in one class loop (256 registers, two chains per wave)
    int block = 0;                    // threads per workgroup (0 = 256); 64: one-wave workgroups, which fit the registers a SIMD has
                                      // left beside a routing wave whatever the CU's other SIMDs hold (the fillers of a fed run)
};
// Validates, builds the per-class tables and uploads them (waits for the context's stream once).
int xh_pm_prepare(xh_ctx *ctx, const xh_pm_tables *t, int64_t ncell, int32_t nmonths, int32_t start_year,
                  int32_t n_lc_years, const int32_t *h_lc_years, int32_t water_idx, int32_t snow_idx, xh_pm_setup *out);
// PET of months [m_begin, m_begin + m_count) (both multiples of 2) of every cell, on stream `st`.
int xh_pm_enqueue(xh_ctx *ctx, hipStream_t st, xh_pm_setup &s, int m_begin, int m_count, const double *d_tas,
                  const double *d_tmin, const double *d_rhs, const double *d_wind, const double *d_rsds,
                  const double *d_rlds, const double *d_tairprev, const double *d_lct, const double *d_elev,
                  double *d_pet);

struct xh_abcd_setup {
    int64_t ncell = 0;
    int nmonths = 0, spinup = 0, n_groups = 0;
    int *d_ptr = nullptr, *d_cells = nullptr, *d_bidx = nullptr, *d_pidx = nullptr;
    double *d_dec = nullptr, *d_sm0 = nullptr, *d_gw0 = nullptr, *d_state = nullptr;
};
int xh_abcd_prepare(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t spinup, int32_t n_groups,
                    const int32_t *h_basin_index, const int32_t *h_par_index, int64_t npar_rows, xh_abcd_setup *out);
// spin-up march + per-basin December means (leaves sm0 / gw0 in the setup's scratch)
int xh_abcd_enqueue_spinup(xh_ctx *ctx, hipStream_t st, const xh_abcd_setup &s, const double *d_pars,
                           const double *d_pet, const double *d_precip, const double *d_tmin);
// simulation months [m_begin, m_end) (multiples of 2; m_begin = 0 starts from the basin means, later blocks continue
// from the state the previous block left in the setup's scratch)
int xh_abcd_enqueue_sim(xh_ctx *ctx, hipStream_t st, const xh_abcd_setup &s, int m_begin, int m_end,
                        const double *d_pars, const double *d_pet, const double *d_precip, const double *d_tmin,
                        double *d_aet, double *d_q, double *d_sav, double *d_q_staged);
// (d_q_staged: NULL, or the routing kernel's copy of the runoff, [ceil(nmonths / 16)][ncell][16]; FlowFeed in xh_mrtm_flow.h)

// xh_route_series with the runoff arriving while the routing kernel runs (xh_mrtm.hip; xh_run_fused mode 1).  XH_ERR_LIMIT
// when this plan / schedule cannot be routed that way (nothing has been enqueued then): the caller finishes the runoff
// first and calls xh_route_series.
struct FlowFeed;
// The typed plan's "do the same cells fire?" check enqueued ahead of the call that needs its answer (xh_mrtm.hip); _cancel
// drops an answer that was not used.
int xh_route_series_fed(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months, const int32_t *h_ndays,
                        double dt, const double *d_flow_dist, const double *d_velocity, const double *d_area,
                        const double *d_runoff, const double *d_S0, double *d_chstorage, double *d_avgchflow,
                        int32_t flags, const FlowFeed *feed);
