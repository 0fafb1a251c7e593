// Host side of the two time-skewed dataflow routing kernels (k_mrtm_wave: xh_mrtm_wave.hip, bit-exact, one specialisation per
// row shape, minutes to compile; k_mrtm_rsum: xh_mrtm_rsum.hip, reassociated form): ring / record / placement set-up and the
// launch.  A translation unit of its own so that neither kernel is recompiled for a change here.
#include <algorithm>
#include "xh_mrtm_wave_unit.h"
#undef A

namespace {

// The argument block of the next launch, and its counters (stream `ready` / `done` words, placement) back to zero: one
// small kernel in front of every routing launch (a memset of its own cost the stream another ~5 us of turn-around).
__global__ void k_mrtm_wave_args(WaveArgs a, WaveArgs *dst, uint4 *cnt, unsigned cnt_vec) {
    if (threadIdx.x == 0) *dst = a;
    // fed run: the months-ready word still holds the previous call's "all months"; back to what exists at this launch
    if (threadIdx.x == 0 && a.months_ready)
        __hip_atomic_store(const_cast<unsigned *>(a.months_ready), a.ready_at_launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned i = threadIdx.x; i < cnt_vec; i += blockDim.x) cnt[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace

// placement words of the process's last dataflow launch: printed by xh_fault_check when a launch faults (diagnosis only -- the
// copy fails harmlessly if the plan has been destroyed since; with several contexts it may be another context's launch)
static unsigned *g_last_place = nullptr;
unsigned *xh_wave_last_place() { return g_last_place; }

int wave_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st) {
    if (!fp || fp->n_units == 0) return XH_OK;
    // a lane must have left month it - 1 before the unit's clock reaches month it + 1 (one pending snapshot per lane)
    if (!fp->skew_ok || fp->max_imports > 8 * SK_R || fp->max_exports > 8 * SK_R || s.ntmin < fp->skew_lmax + 2 * GROUP) {
        if (xh_flow_debug())
            fprintf(stderr, "time-skewed kernel not used: skew_ok %d, imports %d, outlets %d, shortest month %d sub-steps, largest lag %d\n",
                    (int)fp->skew_ok, fp->max_imports, fp->max_exports, s.ntmin, fp->skew_lmax);
        return XH_ERR_LIMIT;
    }
    // The kernel addresses a cell's row of runoff as (uniform base + 32-bit byte offset) (row_off in wave_unit): a grid whose
    // last routed row starts at or beyond 4 GiB is left to the round-2 kernels, which use 64-bit row offsets
    // (XH_WAVE_ROW_LIMIT: the limit in bytes, for the test that exercises this on a small grid).
    {
        uint64_t limit = (uint64_t)1 << 32;
        if (const char *env = getenv("XH_WAVE_ROW_LIMIT")) limit = std::min<uint64_t>(limit, strtoull(env, nullptr, 10));
        const uint64_t stride = io.feed ? 128u : (uint64_t)s.nmonths * 8u;
        if ((uint64_t)(fp->max_cell + 1) * stride > limit || (io.feed && (int64_t)fp->max_cell >= io.feed->ncell)) {
            if (xh_flow_debug())
                fprintf(stderr, "round-3 time-skewed kernel not used: %d rows x %d months x 8 B exceed its 32-bit row offsets\n",
                        fp->max_cell + 1, s.nmonths);
            return XH_ERR_LIMIT;
        }
    }
    // Ring size.  A consumer asks for ~2 CH + lag sub-steps ahead of its clock, a producer may run RS - CH - lag ahead.  A
    // stream that jumps over k pipeline levels (a tributary that joins the main stem far downstream: its consumer also
    // waits for units k levels below the producer) needs the lead of all of them in its ring: every level trails the one
    // above by RING + lag + CH + GROUP + up to CH of check granularity ~ 400-450 sub-steps.  With a shorter ring
    // nobody deadlocks, but the producer is held at the ring limit, its other consumers starve, and every linked unit ends
    // up waiting a quarter of the time (round 2: 32.7 instead of 26.7 ms with 2,048 sub-steps and a 6-level jump).
    int rs = 2048;
    while (rs < 8 * CH + 4 * fp->skew_lmax || rs < 1024 + (2 * CH + 256) * fp->skew_span) rs *= 2;
    if (const char *env = getenv("XH_FLOW_RS")) {      // experiments: a power of two
        const int v = atoi(env);
        if (v >= 2048 && (v & (v - 1)) == 0) rs = v;
    }
    const size_t x_streams = (size_t)std::max(fp->n_edges, 1) * (size_t)rs * sizeof(v2d);
    if (x_streams + (size_t)rs * 16 > ((size_t)1 << 32)) return XH_ERR_LIMIT;      // 32-bit ring offsets; idle block lanes point at ~(RS * 16 - 1), beyond them
    const size_t x_cnt = ((size_t)(fp->n_edges + fp->n_units + PLACE_WORDS) * sizeof(unsigned) + 255) & ~size_t(255);
    const size_t x_rec0 = ((size_t)(s.nit + 3) * sizeof(MonthRec) + 255) & ~size_t(255);
    const size_t x_rec = x_rec0 + (((size_t)(s.nit + 2) * sizeof(FinRec) + 255) & ~size_t(255));
    if (x_streams + x_cnt + x_rec > fp->x_bytes) {
        if (fp->d_x) {
            XH_HIP(ctx, hipStreamSynchronize(st));
            XH_HIP(ctx, hipFree(fp->d_x));
            fp->d_x = nullptr;
        }
        XH_HIP(ctx, hipMalloc(&fp->d_x, x_streams + x_cnt + x_rec));
        fp->x_bytes = x_streams + x_cnt + x_rec;
        fp->rec_key = 0;
    }
    unsigned *cnt = reinterpret_cast<unsigned *>(static_cast<char *>(fp->d_x) + x_streams);
    MonthRec *d_rec = reinterpret_cast<MonthRec *>(static_cast<char *>(fp->d_x) + x_streams + x_cnt);
    FinRec *d_fin = reinterpret_cast<FinRec *>(static_cast<char *>(fp->d_x) + x_streams + x_cnt + x_rec0);
    // (cnt is zeroed by k_mrtm_wave_args, below)
    // where month m of a cell's row lies in the runoff source (WaveArgs::q_row_stride)
    const FlowFeed *feed = io.feed;
    auto q_off = [&](int m) -> long long {
        if (!feed) return (long long)m * 8;
        return (long long)(m >> 4) * feed->ncell * 128 + (long long)(m & 15) * 8;
    };
    // The records only change with the schedule or the layout of the runoff source: a caller that routes the same months
    // again (a scenario sweep, the bench loop) finds them on the device already.
    uint64_t rec_key = 1469598103934665603ull;
    {
        auto mix = [&](const void *p, size_t nb) {
            const unsigned char *b = static_cast<const unsigned char *>(p);
            for (size_t i = 0; i < nb; ++i) rec_key = (rec_key ^ b[i]) * 1099511628211ull;
        };
        const long long lay[4] = {s.nit, s.total, feed ? (long long)feed->ncell : -1ll, (long long)(size_t)fp->d_x};
        mix(lay, sizeof(lay));
        mix(s.h_m, sizeof(int) * (size_t)s.nit);
        mix(s.h_nt, sizeof(int) * (size_t)s.nit);
        mix(s.h_g, sizeof(int) * (size_t)(s.nit + 1));
        mix(s.h_wr, (size_t)s.nit);
        mix(s.h_secs, sizeof(double) * (size_t)s.nit);
        if (rec_key == 0) rec_key = 1;
    }
    if (rec_key != fp->rec_key) {   // the schedule as one record per iteration (+ three zero records: the month bookkeeping looks two ahead)
        fp->rec_key = 0;
        fp->h_rec.assign((size_t)(s.nit + 3) * sizeof(MonthRec), 0);
        MonthRec *h = reinterpret_cast<MonthRec *>(fp->h_rec.data());
        for (int it = 0; it < s.nit; ++it) {
            h[it].m = s.h_m[it];
            h[it].nt = s.h_nt[it];
            h[it].g = s.h_g[it];
            h[it].write = s.h_wr[it];
            h[it].secs = s.h_secs[it];
            h[it].q_off = q_off(s.h_m[it]);
        }
        h[s.nit].g = s.total;
        fp->h_fin.assign((size_t)(s.nit + 2) * sizeof(FinRec), 0);
        FinRec *hf = reinterpret_cast<FinRec *>(fp->h_fin.data());
        for (int it = 0; it <= s.nit; ++it) {
            if (it >= 1) {
                hf[it].m_prev_w = h[it - 1].m | (h[it - 1].write ? FIN_WRITE : 0);
                hf[it].nt_prev = h[it - 1].nt;
            }
            hf[it].nt_prev = std::max(hf[it].nt_prev, 1);
            hf[it].secs_next1 = it + 1 < s.nit ? h[it + 1].secs : 1.0;
            hf[it].m_next2 = it + 2 < s.nit ? h[it + 2].m : 0;
            hf[it].q_off_next2 = q_off(hf[it].m_next2);
            hf[it].g_next1 = it + 1 <= s.nit ? h[it + 1].g : s.total;
        }
        hf[s.nit + 1].nt_prev = 1;
        hf[s.nit + 1].secs_next1 = 1.0;
        // pageable source: the copy is staged before the call returns, so h_rec may be rewritten by the next launch
        XH_HIP(ctx, hipMemcpyAsync(d_rec, h, fp->h_rec.size(), hipMemcpyHostToDevice, st));
        XH_HIP(ctx, hipMemcpyAsync(d_fin, hf, fp->h_fin.size(), hipMemcpyHostToDevice, st));
        fp->rec_key = rec_key;
    }

    // every unit resident at once (see flow_launch for the LDS-share sizing)
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // the reassociated plan (xh_flow_rsum.cpp) has a kernel of its own: same argument block, same protocol (xh_mrtm_rsum.hip)
    const void *kernel = fp->rsum ? wave_rsum_kernel() : wave_exact_kernel();
    const size_t lds_static = (size_t)RING * NSLOT * sizeof(v2d) + LANES * sizeof(uint2) + 2 * LANES * sizeof(unsigned) +
                              LANES * sizeof(double) + 64 +     // + fend_sh, unit_sh / prio_sh, padded
                              (fp->rsum ? 2 * LANES * sizeof(unsigned) + LANES * sizeof(double) : 0);      // (k_mrtm_rsum: the folded leaves' halves)
    // single-sum plans: their pair units (the tail of the claim list) get a CU to themselves (wave_claim); XH_RSUM_EXCL=0: A/B
    int n_excl = 0;
    if (fp->rsum && fp->n_special >= 0 && fp->n_pair_units > 0 && xh_env_on("XH_RSUM_EXCL", true))
        n_excl = std::min(fp->n_pair_units, 0xffff);
    int n_wg = 0, resident = 0;
    size_t lds = 0;
    {
        // spare workgroups (k_mrtm_wave, "which unit this workgroup runs"): one per CU, as long as two waves per SIMD hold all
        n_wg = fp->n_units + cus;
        if (n_wg > 8 * cus) n_wg = std::max(fp->n_units, 8 * cus);
        {
            const char *env = getenv("XH_FLOW_SPARE");            // experiments only
            if (env) n_wg = fp->n_units + std::max(atoi(env), 0);
        }
        // LDS share per workgroup: what spreads the launch over the CUs.  Three more per CU than the even split (one more until
        // round 6): the workgroups of an XCD must ALL find a CU of that XCD with a contiguous share free before any of them
        // moves on, and beside another context's kernels that hold LDS of their own the last ones did not -- 999 units + 256
        // spares = 157 workgroups per XCD against 32 CUs x 6 shares, and the launch sat out its 5 s placement bound (then
        // routed by the workgroup-per-network kernel: tests/test_gpu_fullsize.py::test_config3_twenty_repetitions_and_
        // background_load failed in 4 of 5 runs; tools/bg_load_probe.py).  With three, two waves per SIMD are the limit again;
        // the kernel is no slower on an idle device (12.3-12.5 ms, every unit alone on its SIMD).
        const int per_cu = (n_wg + cus - 1) / cus + 3;
        const size_t share = ((size_t)(160 * 1024) / (size_t)per_cu) & ~size_t(1023);
        lds = share > lds_static + 1024 ? share - lds_static : 0;
        XH_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        XH_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, LANES, lds));
        if ((int64_t)(resident - 1) * cus < n_wg || n_wg > 8 * cus) return XH_ERR_LIMIT;
    }

    WaveArgs a;
    a.cell_of_slot = static_cast<const int *>(fp->d_cell_of_slot.p);
    a.lag = static_cast<const int *>(fp->d_lag.p);
    a.ghost_lag = static_cast<const int *>(fp->d_ghost_lag.p);
    a.export_edge = static_cast<const int *>(fp->d_export_edge.p);
    a.ghost_edge = static_cast<const int *>(fp->d_ghost_edge.p);
    a.edge_cons_unit = static_cast<const int *>(fp->d_edge_cons_unit.p);
    a.ent2 = static_cast<const unsigned *>(fp->d_ent2.p);
    a.eprev = static_cast<const unsigned *>(fp->d_eprev.p);
    a.unit_p = static_cast<const int *>(fp->d_unit_p.p);
    a.unit_order = static_cast<const int *>(fp->d_unit_order.p);
    a.n_units = fp->n_units;
    a.unit_lmax = static_cast<const int *>(fp->d_unit_lmax.p);
    a.unit_glmax = static_cast<const int *>(fp->d_unit_glmax.p);
    a.total_slots = (int64_t)fp->n_units * LANES;
    a.nmonths = s.nmonths;
    a.nit = s.nit;
    a.total = s.total;
    a.odd_ok = s.nt_even ? 0 : 1;
    a.rec = d_rec;
    a.fin = d_fin;
    a.lane_flags = static_cast<const unsigned char *>(fp->d_lane_flags.p);
    a.dt = s.dt;
    a.dtinv = 1.0 / s.dt;
    a.flow_dist = io.flow_dist;
    a.velocity = io.velocity;
    a.area = io.area;
    a.runoff = feed ? feed->q_staged : io.runoff;
    a.q_row_stride = feed ? 128u : (unsigned)s.nmonths * 8u;
    a.ready_at_launch = feed ? feed->ready_at_launch : UINT_MAX;
    a.months_ready = feed ? feed->months_ready : nullptr;
    a.place_epoch = feed ? feed->place_epoch : nullptr;
    a.epoch = feed ? feed->epoch : 0u;
    a.n_excl = n_excl;
    // XH_ROUTE_FENCED=1: agent-scope release / acquire fences around the stream counters -- the publication the HIP memory model
    // asks for (+76 % on this kernel: profiles/round5/fence_mid_ab.txt); the default is the measured `sc1` hand-off (check())
    a.fenced = xh_env_on("XH_ROUTE_FENCED", false) ? 1 : 0;
    a.S0 = io.S0;
    a.chs = io.chs;
    a.avg = io.avg;
    a.S_end = io.S_end;
    a.F_end = io.F_end;
    a.xbuf = static_cast<char *>(fp->d_x);
    a.xbytes = (unsigned)x_streams;
    a.ring_mask_b = (unsigned)rs * 16u - 1u;
    a.rs = rs;
    a.ready = cnt;
    a.done = cnt + fp->n_edges;
    a.place = cnt + fp->n_edges + fp->n_units;
    g_last_place = a.place;
    unsigned *fault = nullptr;
    int rc = xh_fault_word(ctx, &fault);
    if (rc) return rc;
    a.fault = fault;
    // XH_ROUTE_TEST_FAULT raises the fault word before the launch, as a timed-out wait of another unit would: every
    // unit that has to wait gives up and the call is re-routed.
    if (s.test_fault) XH_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(fault), (int)FAULT_TEST, 1, st));
    a.stats = nullptr;
    {
        const char *env = getenv("XH_FLOW_STATS");
        if (env && env[0] == '1') {
            if (!fp->d_stats) XH_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&fp->d_stats), (size_t)fp->n_units * 48));
            a.stats = fp->d_stats;
        }
    }
    a.fold_cell = static_cast<const int *>(fp->d_fold_cell.p);
    if (!fp->d_skew_args) XH_HIP(ctx, hipMalloc(&fp->d_skew_args, sizeof(WaveArgs) + 256));
    // stream-ordered: the previous launch has finished reading the block before this one rewrites it
    hipLaunchKernelGGL(k_mrtm_wave_args, dim3(1), dim3(256), 0, st, a, static_cast<WaveArgs *>(fp->d_skew_args),
                       reinterpret_cast<uint4 *>(cnt), (unsigned)(x_cnt / sizeof(uint4)));
    {
        const WaveArgs *d_args = static_cast<const WaveArgs *>(fp->d_skew_args);
        void *kargs[] = {&d_args};
        XH_HIP(ctx, hipLaunchKernel(kernel, dim3((unsigned)n_wg), dim3(LANES), kargs, lds, st));
    }
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}
