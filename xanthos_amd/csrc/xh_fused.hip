// xh_run_fused: Penman-Monteith -> ABCD -> MRTM for one set of cells as ONE pipelined call (gfx950).
//
// The reference hands whole arrays from stage to stage (components.py:344-370: calculate_pet -> calculate_runoff ->
// calculate_routing), so each stage starts when the previous one has finished.  On the device the three stages stress
// different things -- PM is bound by fp64 issue, the ABCD march by its dependent chain (it leaves most issue slots
// empty), routing by the latency of one wave per unit (it leaves ~3/4 of them empty) -- so they are overlapped:
//
//   stream A   PM block 0 | PM block 1 | PM block 2 | ...                         (blocks of `block_months` months)
//   stream B              | ABCD spin-up, basin means | ABCD sim block 0 | sim block 1 | ...   -> months_ready
//   stream C                                                              | routing, all months ......................
//
// * a block of PET is consumed by the ABCD march while it is still in L2 / Infinity Cache (block = 67,420 x 96 x 8 B =
//   52 MB) and PM's next block runs on the issue slots the march leaves empty; the march carries snowpack / soil
//   moisture / groundwater from block to block in a [3, ncell] scratch array;
// * routing starts as soon as the first block of runoff exists: the time-skewed kernel polls the device word
//   `months_ready` before it loads a month of runoff (agent-coherent loads; the runoff kernel's stores are written back
//   at its end, before the word is advanced on the same stream).  It consumes 120 months in ~5 ms, the upstream stages
//   produce them in ~0.6 ms, so after the first block it never waits;
// * if the routing falls back to a kernel without month flags (general graphs, very short months) it simply waits for
//   the event that marks the last block of runoff.
// Results are identical to the three separate calls: the same kernels, the same order of operations.
#include "xh_stage.h"

namespace {

__global__ void k_set_u32(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int fused_resources(xh_ctx *ctx, size_t n_events) {
    for (int i = 0; i < 2; ++i)
        if (!ctx->side_stream[i]) XH_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream[i], hipStreamNonBlocking));
    while (ctx->side_events.size() < n_events) {
        hipEvent_t e = nullptr;
        XH_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->side_events.push_back(e);
    }
    if (!ctx->d_months_ready) XH_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_months_ready), 64));
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
    xh_pm_setup pm;
    int rc = xh_pm_prepare(ctx, a->pm, a->ncell, nmonths, a->start_year, a->n_lc_years, a->h_lc_years, a->water_idx,
                           a->snow_idx, &pm);
    if (rc) return rc;
    xh_abcd_setup ab;
    rc = xh_abcd_prepare(ctx, a->ncell, nmonths, a->abcd_spinup, a->n_groups, a->h_basin_index, a->h_par_index,
                         a->npar_rows, &ab);
    if (rc || a->ncell == 0) return rc;
    const int nblk = (nmonths + block - 1) / block;
    rc = fused_resources(ctx, (size_t)2 * nblk + 3);
    if (rc) return rc;
    hipStream_t A = ctx->stream, B = ctx->side_stream[0], C = ctx->side_stream[1];
    hipEvent_t ev_start = ctx->side_events[0], ev_route = ctx->side_events[1], ev_spin = ctx->side_events[2];
    hipEvent_t *ev_pm = &ctx->side_events[3], *ev_ab = &ctx->side_events[3 + nblk];

    XH_HIP(ctx, hipEventRecord(ev_start, A));                    // everything enqueued before this call
    XH_HIP(ctx, hipStreamWaitEvent(B, ev_start, 0));
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, B, ctx->d_months_ready, 0u);
    for (int k = 0; k < nblk; ++k) {
        const int m0 = k * block, cnt = std::min(block, nmonths - m0);
        rc = xh_pm_enqueue(ctx, A, pm, m0, cnt, a->d_tas, a->d_tmin, a->d_rhs, a->d_wind, a->d_rsds, a->d_rlds,
                           a->d_tairprev, a->d_lct, a->d_elev, a->d_pet);
        if (rc) return rc;
        XH_HIP(ctx, hipEventRecord(ev_pm[k], A));
    }
    // spin-up needs PET of months [0, spinup)
    XH_HIP(ctx, hipStreamWaitEvent(B, ev_pm[(a->abcd_spinup - 1) / block], 0));
    rc = xh_abcd_enqueue_spinup(ctx, B, ab, a->d_pars, a->d_pet, a->d_precip, a->d_abcd_tmin);
    if (rc) return rc;
    XH_HIP(ctx, hipEventRecord(ev_spin, B));
    for (int k = 0; k < nblk; ++k) {
        const int m0 = k * block, m1 = std::min(nmonths, m0 + block);
        XH_HIP(ctx, hipStreamWaitEvent(B, ev_pm[k], 0));
        rc = xh_abcd_enqueue_sim(ctx, B, ab, m0, m1, a->d_pars, a->d_pet, a->d_precip, a->d_abcd_tmin, a->d_aet, a->d_q,
                                 a->d_sav);
        if (rc) return rc;
        hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, B, ctx->d_months_ready, (unsigned)m1);
        XH_HIP(ctx, hipEventRecord(ev_ab[k], B));
    }
    XH_HIP(ctx, hipGetLastError());
    bool routed = false;
    if (a->plan) {
        xh_route_overlap ov;
        ov.stream = C;
        ov.start = ev_ab[0];
        ov.all_ready = ev_ab[nblk - 1];
        ov.d_months_ready = ctx->d_months_ready;
        const size_t before = ctx->pending_routes.size();
        rc = xh_route_enqueue(ctx, a->plan, nmonths, a->routing_spinup, a->h_ndays, a->dt, a->d_flow_dist, a->d_velocity,
                              a->d_area, a->d_q, a->d_S0, a->d_chstorage, a->d_avgchflow, nullptr, nullptr,
                              a->route_flags, &ov);
        if (rc) return rc;
        routed = ctx->pending_routes.size() > before;
        XH_HIP(ctx, hipEventRecord(ev_route, C));
        XH_HIP(ctx, hipStreamWaitEvent(A, ev_route, 0));
    }
    XH_HIP(ctx, hipStreamWaitEvent(A, ev_ab[nblk - 1], 0));      // join: later calls on the context see every output
    if (routed) return xh_fault_collect(ctx);
    return XH_OK;
}
