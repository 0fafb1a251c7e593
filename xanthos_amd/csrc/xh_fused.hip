// xh_run_fused: Penman-Monteith -> ABCD -> MRTM for one set of cells as ONE call (gfx950).
//
// The reference hands whole arrays from stage to stage (components.py:344-370: calculate_pet -> calculate_runoff ->
// calculate_routing), so each stage starts when the previous one has finished.  PM is bound by fp64 issue; the ABCD
// march is bound by its dependent chain and leaves most issue slots empty.  Here PM runs in blocks of `block_months`
// months on the context's stream and the ABCD march follows one block behind on a second stream:
//
//   stream A   PM block 0 | PM block 1 | PM block 2 | ...                      | routing (all months)
//   stream B              | ABCD spin-up, basin means | sim block 0 | sim block 1 | ...
//
// so that a block of PET is consumed while it is still in L2 / Infinity Cache (67,420 x 96 x 8 B = 52 MB) and the march
// of block k runs on the issue slots PM's block k + 1 leaves empty; snowpack / soil moisture / groundwater are carried
// from block to block in a [3, ncell] scratch array.  Routing follows when the last block of runoff exists.
//
// Tried and dropped (round 2, measured on MI355X): starting the time-skewed routing kernel on the FIRST block of runoff
// (the kernel polling a device word for the months behind it).  The poll and the coherent loads cost the routing kernel
// 17 % even when it never waited (32.6 -> 38.2 ms: more live registers through the sub-step loop), and with PM's and
// ABCD's waves taking the register file the routing units were no longer all resident: their bounded waits timed out
// and the call was re-routed.  The dataflow kernel wants the chip to itself.
// Results are identical to the three separate calls: the same kernels, the same order of operations.
#include <algorithm>
#include <string>

#include "xh_stage.h"

namespace {

int fused_resources(xh_ctx *ctx, size_t n_events) {
    if (!ctx->side_stream[0]) XH_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream[0], hipStreamNonBlocking));
    while (ctx->side_events.size() < n_events) {
        hipEvent_t e = nullptr;
        XH_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->side_events.push_back(e);
    }
    return XH_OK;
}

}  // namespace

extern "C" int xh_run_fused(xh_ctx *ctx, const xh_fused_args *a) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, a, "xh_run_fused: NULL argument");
    XH_REQUIRE(ctx, a->d_tas && a->d_tmin && a->d_rhs && a->d_wind && a->d_rsds && a->d_rlds && a->d_lct && a->d_elev &&
                        a->d_pars && a->d_precip && a->d_pet && a->d_q,
               "xh_run_fused: NULL array (PET and runoff are intermediate results and must be given)");
    XH_REQUIRE(ctx, !a->plan || (a->h_ndays && a->d_flow_dist && a->d_velocity && a->d_area),
               "xh_run_fused: routing needs ndays, flow distance, velocity and area");
    const int nmonths = a->nmonths;
    int block = a->block_months > 0 ? a->block_months : 96;
    XH_REQUIRE(ctx, block % 48 == 0, "xh_run_fused: block_months must be a multiple of 48 (whole years and whole lines)");
    XH_REQUIRE(ctx, nmonths > 0 && nmonths % 12 == 0, "xh_run_fused: nmonths must be whole years (PM runs year by year)");
    XH_REQUIRE(ctx, a->abcd_spinup >= 1 && a->abcd_spinup <= nmonths, "xh_run_fused: abcd_spinup out of range");
    xh_pm_setup pm;
    int rc = xh_pm_prepare(ctx, a->pm, a->ncell, nmonths, a->start_year, a->n_lc_years, a->h_lc_years, a->water_idx,
                           a->snow_idx, &pm);
    if (rc) return rc;
    xh_abcd_setup ab;
    rc = xh_abcd_prepare(ctx, a->ncell, nmonths, a->abcd_spinup, a->n_groups, a->h_basin_index, a->h_par_index,
                         a->npar_rows, &ab);
    if (rc || a->ncell == 0) return rc;
    const int nblk = (nmonths + block - 1) / block;
    rc = fused_resources(ctx, (size_t)nblk + 3);
    if (rc) return rc;
    hipStream_t A = ctx->stream, B = ctx->side_stream[0];
    hipEvent_t ev_start = ctx->side_events[0], ev_done = ctx->side_events[1];
    hipEvent_t *ev_pm = &ctx->side_events[3];

    // From here on work is enqueued on two streams: every way out -- errors included -- goes through the join below, so
    // that later calls on the context's stream (and xh_sync) are ordered behind whatever stream B still runs and no
    // ABCD kernel keeps writing d_q / d_aet / d_sav / the state scratch behind the caller's back.
    auto enqueue = [&]() -> int {
        XH_HIP(ctx, hipEventRecord(ev_start, A));                // everything enqueued before this call
        XH_HIP(ctx, hipStreamWaitEvent(B, ev_start, 0));
        for (int k = 0; k < nblk; ++k) {
            const int m0 = k * block, cnt = std::min(block, nmonths - m0);
            int r = xh_pm_enqueue(ctx, A, pm, m0, cnt, a->d_tas, a->d_tmin, a->d_rhs, a->d_wind, a->d_rsds, a->d_rlds,
                                  a->d_tairprev, a->d_lct, a->d_elev, a->d_pet);
            if (r) return r;
            XH_HIP(ctx, hipEventRecord(ev_pm[k], A));
        }
        // spin-up needs PET of months [0, spinup)
        XH_HIP(ctx, hipStreamWaitEvent(B, ev_pm[(a->abcd_spinup - 1) / block], 0));
        int r = xh_abcd_enqueue_spinup(ctx, B, ab, a->d_pars, a->d_pet, a->d_precip, a->d_abcd_tmin);
        if (r) return r;
        for (int k = 0; k < nblk; ++k) {
            const int m0 = k * block, m1 = std::min(nmonths, m0 + block);
            XH_HIP(ctx, hipStreamWaitEvent(B, ev_pm[k], 0));
            r = xh_abcd_enqueue_sim(ctx, B, ab, m0, m1, a->d_pars, a->d_pet, a->d_precip, a->d_abcd_tmin, a->d_aet,
                                    a->d_q, a->d_sav);
            if (r) return r;
        }
        XH_HIP(ctx, hipGetLastError());
        return XH_OK;
    };
    rc = enqueue();
    const std::string first_error = rc ? ctx->err : std::string();
    const hipError_t j1 = hipEventRecord(ev_done, B);
    const hipError_t j2 = j1 == hipSuccess ? hipStreamWaitEvent(A, ev_done, 0) : j1;      // join
    if (j2 != hipSuccess) {                                      // cannot order the streams: wait for B here instead
        (void)hipStreamSynchronize(B);
        if (!rc) rc = xh_fail(ctx, XH_ERR_HIP, "xh_run_fused: joining the side stream failed: %s", hipGetErrorString(j2));
    }
    if (rc) {
        if (!first_error.empty()) ctx->err = first_error;
        return rc;
    }
    if (a->plan)
        return xh_route_series(ctx, a->plan, nmonths, a->routing_spinup, a->h_ndays, a->dt, a->d_flow_dist, a->d_velocity,
                               a->d_area, a->d_q, a->d_S0, a->d_chstorage, a->d_avgchflow, nullptr, nullptr,
                               a->route_flags);
    return XH_OK;
}
