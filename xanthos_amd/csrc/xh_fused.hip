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
//
// Round 4, mode 1 (XH_FUSED_FEED): the other way round -- the routing kernel gets the chip first and the rest of PM and
// ABCD runs beside it.  Routing's spin-up pass and the first months of its simulation pass only need the first B0 =
// max(spin-ups) months of runoff, rounded up to 16; that is 7.9 ms of its 23 at the full grid, and PM + ABCD of the other
// months is 2.2 ms of work:
//
//   stream A   PM [0,B0) | ABCD spin-up, means, sim [0,B0) | routing kernel (all months) ................... | join
//   stream B                                                  gate | PM [B0,n) | ABCD sim [B0,n) | ready = n
//
// What makes this work where round 2's attempt did not: (1) the routing kernel is launched onto an IDLE chip and the side
// stream's first kernel (k_gate) holds everything else back until the routing kernel reports that each of its workgroups is
// resident and has claimed its SIMD (place_epoch) -- its units are then never displaced, the others take the wave slots
// that are left (one PM or ABCD wave fits beside a routing wave's 256 registers); (2) the routing waves raise their issue
// priority (s_setprio) above the fillers'; (3) the only thing the sub-step loop pays is one scalar comparison per MONTH
// (month index against the months known to be final), the months-ready word is polled only when that fails; (4) no
// coherent loads, no invalidates: ABCD writes the routing kernel's runoff a second time in a staged layout
// [16-month line][cell] whose 128-byte lines are each written whole by one launch and never read before the flag says so
// (FlowFeed, xh_mrtm_flow.h), so no cache can hold a stale copy.  Same kernels, same arithmetic: bit-identical outputs.
#include <algorithm>
#include <string>

#include "xh_mrtm_flow.h"
#include "xh_stage.h"

namespace {

int fused_resources(xh_ctx *ctx, size_t n_events) {
    if (!ctx->side_stream[0]) {
        // LOWEST priority, and not only because its kernels are the fillers of mode 1: the runtime multiplexes streams onto
        // a few hardware queues per priority level, and two streams that share a queue run their kernels in submission
        // order -- the side stream's kernels would sit behind the routing kernel that is waiting for their months (a
        // bounded wait, a fault, a re-route: observed).  Streams of different priority never share a queue.
        int least = 0, greatest = 0;
        XH_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        // (the context's stream has the default priority 0; a device whose range has nothing below it gets the other end)
        const int prio = least != 0 ? least : greatest;
        // one priority level only: the side stream would share the context stream's hardware queues and its kernels could sit
        // behind the routing kernel that waits for them -- no fed order on such a device (run_fed turns the call down)
        ctx->feed_queue_ok = least != greatest;
        XH_HIP(ctx, hipStreamCreateWithPriority(&ctx->side_stream[0], hipStreamNonBlocking, prio));
        if (xh_flow_debug()) fprintf(stderr, "[libxanthos_hip] side stream priority %d (range %d .. %d)\n", prio, least, greatest);
    }
    while (ctx->side_events.size() < n_events) {
        hipEvent_t e = nullptr;
        XH_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->side_events.push_back(e);
    }
    return XH_OK;
}

__global__ void k_set_word(unsigned *w, unsigned v) {
    if (threadIdx.x == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Holds the side stream back until the routing kernel of this call has placed its workgroups (it writes `epoch`), at most
// `limit_ticks` of the 100 MHz counter: correctness never depends on it (the stream also waits for an event), only who
// gets the wave slots first.
__global__ void k_gate(unsigned *w, unsigned epoch, unsigned long long limit_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // (polled with an atomic add of 0: never answered from a cached copy of the line)
    while ((int)(__hip_atomic_fetch_add(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// mode 1 of xh_run_fused (see the top of this file).  XH_ERR_LIMIT when the routing cannot be fed: too few months (nothing
// has been enqueued, *runoff_done false) or a plan / flags the time-skewed dataflow kernel does not take (PM and ABCD have
// then been enqueued for the whole series on the context's stream, *runoff_done true: the caller routes the ordinary way).
int run_fed(xh_ctx *ctx, const xh_fused_args *a, xh_pm_setup &pm, xh_abcd_setup &ab, bool *runoff_done) {
    *runoff_done = false;
    const int nmonths = a->nmonths;
    // First block: at least the spin-ups (ABCD's needs PET of those months, the routing's pass over them comes first), and
    // enough months to keep the routing kernel busy until the side stream has delivered the rest.  Beside the routing waves
    // PM + ABCD take ~12 us per month of the full grid (one PM wave per SIMD, below the routing waves' priority, in one-wave
    // workgroups: four-wave ones need a free slot on all four SIMDs of a CU at once and ran at 1/6 of the stand-alone rate)
    // and hand everything over at the end; the routing advances one month in ~33 us and asks for month B0 after its spin-up
    // pass and B0 more months: (spin-up + B0) x 33 >= (months - B0) x 12, plus a quarter as margin.  600 months with 120 months
    // of spin-up: 90 -> the spin-ups' 128 decide (measured: 24.60 ms per step against 24.70 with 160 months in front, same
    // box, three alternating runs each; profiles/round4/feed_first_block.txt has the sweep with the slower side kernels of
    // the round's start, when a quarter of the series was needed).
    int b0 = (std::max(a->abcd_spinup, a->routing_spinup) + 15) & ~15;
    {
        const long long need = (12ll * nmonths - 33ll * a->routing_spinup) * 5 / (45 * 4);
        b0 = std::max(b0, (int)((std::max(need, 0ll) + 15) & ~15ll));
    }
    b0 = std::max(b0, 32);
    // (more months in front do not pay: 128 / 160 / 192 / 224 / 256 at the full grid give 13.9 / 14.0 / 14.1 / 14.1 / 14.3 ms per
    // step, round 6: profiles/round6/fed_penalty.txt)
    hipStream_t A = ctx->stream, B = ctx->side_stream[0];
    auto pm_block = [&](hipStream_t st, int m0, int m1) {
        return xh_pm_enqueue(ctx, st, pm, m0, m1 - m0, a->d_tas, a->d_tmin, a->d_rhs, a->d_wind, a->d_rsds, a->d_rlds, a->d_tairprev,
                             a->d_lct, a->d_elev, a->d_pet);
    };
    auto sim_block = [&](hipStream_t st, int m0, int m1, double *staged) {
        return xh_abcd_enqueue_sim(ctx, st, ab, m0, m1, a->d_pars, a->d_pet, a->d_precip, a->d_abcd_tmin, a->d_aet, a->d_q,
                                   a->d_sav, staged);
    };
    if (nmonths - b0 < 64) return XH_ERR_LIMIT;
    // nothing has been enqueued yet: a device without a second priority level, or a context on which a fed call has already
    // waited in vain for its side stream (kernels serialised by a profiler's counter collection, AMD_SERIALIZE_KERNEL), runs
    // stage by stage
    if (!ctx->feed_queue_ok || ctx->feed_disabled) return XH_ERR_LIMIT;
    // the routing kernel's copy of the runoff and the two words of the hand-over (each on a line of its own)
    const size_t lines = (size_t)(nmonths + 15) / 16;
    const size_t need = lines * (size_t)a->ncell * 128 + 256;
    if (need > ctx->feed_bytes) {
        int rc = xh_settle(ctx);
        if (rc) return rc;
        if (ctx->d_feed) XH_HIP(ctx, hipFree(ctx->d_feed));
        ctx->d_feed = nullptr;
        ctx->feed_bytes = 0;
        XH_HIP(ctx, hipMalloc(&ctx->d_feed, need));
        ctx->feed_bytes = need;
        XH_HIP(ctx, hipMemsetAsync(ctx->d_feed, 0, 256, A));
    }
    unsigned *w_ready = static_cast<unsigned *>(ctx->d_feed), *w_epoch = w_ready + 32;
    double *staged = reinterpret_cast<double *>(static_cast<char *>(ctx->d_feed) + 256);
    hipEvent_t ev_blk0 = ctx->side_events[0], ev_done = ctx->side_events[1];

    int rc;
    rc = pm_block(A, 0, b0);
    if (rc) return rc;
    rc = xh_abcd_enqueue_spinup(ctx, A, ab, a->d_pars, a->d_pet, a->d_precip, a->d_abcd_tmin);
    if (rc) return rc;
    rc = sim_block(A, 0, b0, staged);
    if (rc) return rc;
    // (the months-ready word is set to b0 by the routing launch's own argument kernel: k_mrtm_wave_args)
    XH_HIP(ctx, hipEventRecord(ev_blk0, A));
    FlowFeed feed;
    feed.q_staged = staged;
    feed.ncell = a->ncell;
    feed.months_ready = w_ready;
    feed.ready_at_launch = (unsigned)b0;
    feed.place_epoch = w_epoch;
    feed.epoch = ++ctx->feed_epoch;
    rc = xh_route_series_fed(ctx, a->plan, nmonths, a->routing_spinup, a->h_ndays, a->dt, a->d_flow_dist, a->d_velocity,
                             a->d_area, a->d_q, a->d_S0, a->d_chstorage, a->d_avgchflow, a->route_flags, &feed);
    if (rc == XH_ERR_LIMIT) {      // not routed: the rest of the runoff on this stream, then the ordinary call (caller)
        rc = pm_block(A, b0, nmonths);
        if (rc) return rc;
        rc = sim_block(A, b0, nmonths, nullptr);
        *runoff_done = rc == XH_OK;
        return rc ? rc : XH_ERR_LIMIT;
    }
    if (rc) return rc;
    // The routing kernel is in stream A's queue.  Everything below goes to stream B: ordered behind the first block by
    // the event (ABCD's state scratch, the spin-up means), held back by the gate until the routing units are in place.
    std::string first_error;
    auto rest = [&]() -> int {
        XH_HIP(ctx, hipStreamWaitEvent(B, ev_blk0, 0));
        constexpr unsigned long long gate_ticks = 20000ull * 100ull;      // 20 ms of the 100 MHz counter
        {   // (its timer also tells callers how many calls were routed this way: xh_timing_get "feed_gate")
            xh_span sp = xh_span_begin_on(ctx, "feed_gate", B);
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, B, w_epoch, feed.epoch, gate_ticks);
            xh_span_end(sp);
        }
        // (the paired PM kernel: one wave per SIMD is all that fits beside a routing wave, and it is the variant whose lone
        // wave has two independent chains to issue from)
        pm.paired = true;
        pm.block = 64;
        int r = pm_block(B, b0, nmonths);
        pm.paired = false;
        pm.block = 0;
        if (r) return r;
        r = sim_block(B, b0, nmonths, staged);
        if (r) return r;
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(64), 0, B, w_ready, (unsigned)nmonths);
        XH_HIP(ctx, hipGetLastError());
        return XH_OK;
    };
    rc = rest();
    if (rc) first_error = ctx->err;
    // What went to stream B reads none of the routing's outputs: it does not count as "work enqueued behind the routing call"
    // when a fault of this call is settled (xh_fault_check: the call is then routed again from the complete runoff array and
    // the caller hears nothing, as after a stage-by-stage call).
    if (!ctx->pending_routes.empty() && ctx->pending_routes.back().plan == a->plan && ctx->pending_routes.back().chs == a->d_chstorage)
        ctx->pending_routes.back().seq_after = ctx->work_seq;
    // join, whatever happened: later calls on the context's stream (and xh_sync) are ordered behind stream B
    const hipError_t j1 = hipEventRecord(ev_done, B);
    const hipError_t j2 = j1 == hipSuccess ? hipStreamWaitEvent(A, ev_done, 0) : j1;
    if (j2 != hipSuccess) {
        (void)hipStreamSynchronize(B);
        if (!rc) rc = xh_fail(ctx, XH_ERR_HIP, "xh_run_fused: joining the side stream failed: %s", hipGetErrorString(j2));
    }
    if (!rc) {      // PET / AET / Q / Sav are final once ev_done has fired: a side gather of them need not wait for the routing
        ctx->runoff_event = ev_done;
        ctx->runoff_event_fresh = true;
        ctx->runoff_seq = ctx->work_seq;      // (any later entry point that enqueues work on the context makes the event stale)
    }
    if (rc) {
        // the routing kernel may be waiting for months that will never come: raise the fault word so that it gives up
        unsigned *fault = nullptr;
        if (xh_fault_word(ctx, &fault) == XH_OK) hipLaunchKernelGGL(k_set_word, dim3(1), dim3(64), 0, B, fault, 1u);
        if (!first_error.empty()) ctx->err = first_error;
    }
    return rc;
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
    int nblk = (nmonths + block - 1) / block;
    rc = fused_resources(ctx, (size_t)nblk + 3);
    if (rc) return rc;
    if (a->mode == 1 && a->plan) {
        bool runoff_done = false;
        rc = run_fed(ctx, a, pm, ab, &runoff_done);
        if (rc != XH_ERR_LIMIT) return rc;
        if (runoff_done)      // PM and ABCD are complete on the context's stream: the ordinary routing call
            return xh_route_series(ctx, a->plan, nmonths, a->routing_spinup, a->h_ndays, a->dt, a->d_flow_dist, a->d_velocity,
                                   a->d_area, a->d_q, a->d_S0, a->d_chstorage, a->d_avgchflow, nullptr, nullptr,
                                   a->route_flags);
        // (too few months, or no fed order on this context: nothing has been enqueued; the pipeline below does the whole series,
        // in ONE block when the fed order was refused for the device's sake -- PM, then ABCD, then the routing)
        if (!ctx->feed_queue_ok || ctx->feed_disabled) {
            block = (nmonths + 47) / 48 * 48;
            nblk = 1;
        }
    }
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
                                    a->d_q, a->d_sav, nullptr);
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
