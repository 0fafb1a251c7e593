// MRTM routing, the call layer (gfx950 host code + two comparison kernels): what happens AROUND a routing launch.
//
//   * xh_route_series / xh_route_series_fed: the C-ABI entries.  A call that ran on a dataflow kernel is remembered until a
//     synchronisation has confirmed that no bounded wait timed out (xh_fault_collect, xh_ctx.hip); xh_route_rerun routes it
//     again -- on the plan of pairs after a guard trip of a prepared plan, with one workgroup per network after a timeout.
//   * the cross-checks: XH_ROUTE_VALIDATE (flag or environment), the FIRST dataflow call of a plan on a box / build / runtime
//     without a pass on record, and every XH_ROUTE_VALIDATE_EVERY-th call of a long-lived plan are routed again by the
//     barrier-only workgroup-per-network kernel and compared on the device -- bit for bit for the bit-exact kernels, within
//     1e-9 for the reassociated form (the streams between dataflow units rest on an ordering assumption outside the HIP
//     memory model: xh_mrtm_wave_unit.h, check()).
// Reference: the month loops of xanthos/components.py:273-294 are what one call replaces; none of this exists there.
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>

#include "xh_mrtm_plan.h"

// Comparison of two arrays the way the tests compare with the oracle (numpy.array_equal(..., equal_nan=True)): equal values,
// or NaN in both -- the payload and sign of a NaN depend on the order in which a kernel's instructions met it, and the two
// kernels differ there (first seen when the first call of every plan became a checked call: 3 values of a fuzz case with NaN
// runoff): XH_ROUTE_VALIDATE
__global__ void __launch_bounds__(256) k_count_diff(const unsigned long long *a, const unsigned long long *b, int64_t n,
                                                    unsigned long long *count) {
    unsigned long long local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = __longlong_as_double((long long)a[i]), y = __longlong_as_double((long long)b[i]);
        local += (x == y || (x != x && y != y)) ? 0ull : 1ull;
    }
    if (local) atomicAdd(count, local);
}

// The same for a call routed by the reassociated form (XH_ROUTE_REASSOC): equal to rounding, not bit for bit.  A value counts
// when it is farther from the checker's than 1e-9 of it (+ 1e-9 of the array's scale `tiny`, for storages that the excess-flow
// rule has just emptied), or NaN on one side only.
__global__ void __launch_bounds__(256) k_count_far(const double *a, const double *b, int64_t n, double rel, double tiny,
                                                   unsigned long long *count) {
    unsigned long long local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = a[i], y = b[i];
        const bool xn = x != x, yn = y != y;
        local += (xn || yn) ? (xn != yn ? 1ull : 0ull) : ((fabs(x - y) <= rel * fabs(y) + tiny) ? 0ull : 1ull);
    }
    if (local) atomicAdd(count, local);
}

// XH_ROUTE_VALIDATE: the call has just been routed by a dataflow kernel into the caller's arrays; route it again with one
// workgroup per network (barriers, no streams, no reliance on the ordering of write-through stores) into scratch arrays
// and compare every output bit.  Synchronous; a debugging / CI mode.
static int route_validate(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months, const int32_t *h_ndays,
                          double dt, const double *d_flow_dist, const double *d_velocity, const double *d_area,
                          const double *d_runoff, const double *d_S0, const double *d_chs, const double *d_avg,
                          const double *d_S_end, const double *d_F_end, int32_t flags) {
    int rc = xh_settle(ctx);          // a fault of the dataflow run is settled (re-routed) first: then there is nothing to validate
    if (rc) return rc;
    const size_t nc = (size_t)plan->ncell, big = nc * (size_t)nmonths * sizeof(double);
    double *t_chs = nullptr, *t_avg = nullptr, *t_S = nullptr, *t_F = nullptr;
    unsigned long long *d_cnt = nullptr, h_cnt = 0;
    auto release = [&]() {
        for (void *p : {(void *)t_chs, (void *)t_avg, (void *)t_S, (void *)t_F, (void *)d_cnt})
            if (p) (void)hipFree(p);
    };
    hipError_t e = hipSuccess;
    if (d_chs) e = hipMalloc(reinterpret_cast<void **>(&t_chs), big);
    if (e == hipSuccess && d_avg) e = hipMalloc(reinterpret_cast<void **>(&t_avg), big);
    if (e == hipSuccess && d_S_end) e = hipMalloc(reinterpret_cast<void **>(&t_S), nc * sizeof(double));
    if (e == hipSuccess && d_F_end) e = hipMalloc(reinterpret_cast<void **>(&t_F), nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long));
    if (e != hipSuccess) {
        release();
        return xh_fail(ctx, XH_ERR_HIP, "XH_ROUTE_VALIDATE: no memory for the second set of outputs");
    }
    bool used = false;
    const int routed_by = plan->last_tree_kernel;
    const bool by_rsum = plan->last_rsum;       // routed by the reassociated form: equal to rounding, compared within 1e-9
    rc = route_series_impl(ctx, plan, nmonths, spinup_months, h_ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff, d_S0,
                           t_chs, t_avg, t_S, t_F, (flags | XH_ROUTE_NO_DATAFLOW) & ~XH_ROUTE_TEST_FAULT, &used);
    plan->last_tree_kernel = routed_by;
    plan->last_rsum = by_rsum;
    if (!rc) {
        (void)hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), ctx->stream);
        // (storages in m3, flows in m3/s: the absolute terms are far below anything a grid cell holds or passes)
        auto cmp = [&](const double *a, const double *b, size_t n, double tiny) {
            if (!(a && b && n)) return;
            if (by_rsum)
                hipLaunchKernelGGL(k_count_far, dim3(1024), dim3(256), 0, ctx->stream, a, b, (int64_t)n, 1e-9, tiny, d_cnt);
            else
                hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, ctx->stream,
                                   reinterpret_cast<const unsigned long long *>(a),
                                   reinterpret_cast<const unsigned long long *>(b), (int64_t)n, d_cnt);
        };
        cmp(d_chs, t_chs, nc * (size_t)nmonths, 1e-3);
        cmp(d_avg, t_avg, nc * (size_t)nmonths, 1e-9);
        cmp(d_S_end, t_S, nc, 1e-3);
        cmp(d_F_end, t_F, nc, 1e-9);
        if (hipMemcpyAsync(&h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = xh_fail(ctx, XH_ERR_HIP, "XH_ROUTE_VALIDATE: comparison failed to run");
    }
    release();
    if (rc) return rc;
    plan->validated += 1;
    if (h_cnt)
        return xh_fail(ctx, XH_ERR_DEVICE, "XH_ROUTE_VALIDATE: %llu output values of the dataflow routing kernel differ from "
                       "the workgroup-per-network kernel%s", h_cnt, by_rsum ? " by more than 1e-9 (reassociated form)" : "");
    return XH_OK;
}

static int route_series_call(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months,
                             const int32_t *h_ndays, double dt, const double *d_flow_dist,
                             const double *d_velocity, const double *d_area, const double *d_runoff,
                             const double *d_S0, double *d_chstorage, double *d_avgchflow, double *d_S_end,
                             double *d_F_end, int32_t flags, const FlowFeed *feed);

extern "C" int xh_route_series(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months,
                               const int32_t *h_ndays, double dt, const double *d_flow_dist,
                               const double *d_velocity, const double *d_area, const double *d_runoff,
                               const double *d_S0, double *d_chstorage, double *d_avgchflow, double *d_S_end,
                               double *d_F_end, int32_t flags) {
    return route_series_call(ctx, plan, nmonths, spinup_months, h_ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff, d_S0,
                             d_chstorage, d_avgchflow, d_S_end, d_F_end, flags, nullptr);
}

// The runoff source of the routing kernel is the staged copy named by `feed`, filled while the kernel runs; d_runoff is the
// [ncell, nmonths] array the same months end up in, and what a re-run after a fault reads (complete by then).
int xh_route_series_fed(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months, const int32_t *h_ndays,
                        double dt, const double *d_flow_dist, const double *d_velocity, const double *d_area,
                        const double *d_runoff, const double *d_S0, double *d_chstorage, double *d_avgchflow,
                        int32_t flags, const FlowFeed *feed) {
    return route_series_call(ctx, plan, nmonths, spinup_months, h_ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff, d_S0,
                             d_chstorage, d_avgchflow, nullptr, nullptr, flags, feed);
}

// Marker of a passed first-call check: <dir>/route_ok_<device>_<build>_<topology>; dir = $XH_CACHE_DIR or
// $HOME/.cache/xanthos_amd.  Failing to read or write it only means the check runs again.
// (form: 0 the bit-exact kernels, 1 the reassociated form, 2 the prepared reassociated plan: folded leaves "_rf", single sums "_rs")
static std::string first_check_path(const xh_ctx *ctx, const xh_route_plan *plan, int form) {
    const std::string dir = xh_cache_dir();
    if (dir.empty()) return std::string();
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const char *t) {
        for (; *t; ++t) h = (h ^ (unsigned char)*t) * 1099511628211ull;
    };
    mix(ctx->prop.name);
    mix(ctx->prop.gcnArchName);
    mix(__DATE__ " " __TIME__);      // this translation unit's build: a new library build checks again
    {   // the HIP runtime and the driver the pass was recorded under: the ordering the streams rely on is theirs as much as
        // the silicon's (XH_TEST_RUNTIME_TAG: appended, so that a test can stand in for "another runtime")
        int rt = 0, drv = 0;
        (void)hipRuntimeGetVersion(&rt);
        (void)hipDriverGetVersion(&drv);
        char ver[96];
        const char *tag = getenv("XH_TEST_RUNTIME_TAG");
        snprintf(ver, sizeof(ver), "rt%d drv%d %s", rt, drv, tag ? tag : "");
        mix(ver);
    }
    char name[160];
    snprintf(name, sizeof(name), "/route_ok_%016llx_%016llx_%lld_%lld%s", (unsigned long long)h, (unsigned long long)plan->topo_hash,
             (long long)plan->ncell, (long long)(plan->flow ? plan->flow->n_units : 0),
             form == 2 ? ((plan->flow_rsum_fold && plan->flow_rsum_fold->n_special >= 0) ? "_rs" : "_rf") : form == 1 ? "_r" : "");
    return dir + name;
}

static bool &first_checked_of(xh_route_plan *plan, int form) {
    return form == 2 ? plan->first_checked_fold : form == 1 ? plan->first_checked_rsum : plan->first_checked;
}

static int last_form(const xh_route_plan *plan) {
    return !plan->last_rsum ? 0 : (plan->flow_rsum_fold && plan->last_rsum_plan == plan->flow_rsum_fold) ? 2 : 1;
}

static bool first_check_needed(xh_ctx *ctx, xh_route_plan *plan, int form) {
    static const bool enabled = xh_env_on("XH_ROUTE_VALIDATE_FIRST", true);
    bool &checked = first_checked_of(plan, form);
    if (!enabled || checked || !plan->flow) return false;
    const std::string path = first_check_path(ctx, plan, form);
    if (!path.empty()) {
        if (FILE *f = fopen(path.c_str(), "r")) {
            fclose(f);
            checked = true;
            return false;
        }
    }
    return true;
}

static void first_check_passed(xh_ctx *ctx, xh_route_plan *plan, int form) {
    first_checked_of(plan, form) = true;
    const std::string path = first_check_path(ctx, plan, form);
    if (path.empty()) return;
    const std::string dir = path.substr(0, path.rfind('/'));
    for (size_t i = 1; i <= dir.size(); ++i)      // mkdir -p
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);
    if (FILE *f = fopen(path.c_str(), "w")) {
        fprintf(f, "dataflow routing equal to the workgroup-per-network kernel, %s, on %s\n", form ? "within 1e-9" : "bit for bit", ctx->prop.name);
        fclose(f);
    }
}

static int route_series_call(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months,
                             const int32_t *h_ndays, double dt, const double *d_flow_dist,
                             const double *d_velocity, const double *d_area, const double *d_runoff,
                             const double *d_S0, double *d_chstorage, double *d_avgchflow, double *d_S_end,
                             double *d_F_end, int32_t flags, const FlowFeed *feed) {
    bool used_flow = false;
    if (plan && plan->skip_calls > 0 && (flags & XH_ROUTE_TEST_FAULT) == 0) {      // recently faulted: see xh_route_plan
        if (!feed) plan->skip_calls -= 1;      // (a fed call is turned down below and comes back as an ordinary one: counted there)
        flags |= XH_ROUTE_NO_DATAFLOW;
    }
    static const bool validate_env = xh_env_on("XH_ROUTE_VALIDATE", false);
    bool validate = validate_env || (flags & XH_ROUTE_VALIDATE) != 0;
    flags &= ~XH_ROUTE_VALIDATE;
    // first dataflow call of this plan on a box / build that has not passed the cross-check yet: checked like XH_ROUTE_VALIDATE
    const bool plain_call = plan && (flags & (XH_ROUTE_NO_DATAFLOW | XH_ROUTE_FORCE_FALLBACK | XH_ROUTE_ATOMIC | XH_ROUTE_TEST_FAULT)) == 0;
    const bool want_rsum = plan && reassoc_wanted(flags) && (flags & XH_ROUTE_NO_SKEW) == 0;
    const int want_form = !want_rsum ? 0 : (plan->flow_rsum_fold && !plan->fold_disabled && dt == plan->fold_dt) ? 2 : 1;
    bool first_check = !validate && plain_call && first_check_needed(ctx, plan, want_form);
    if (plain_call && plan->flow) {
        const char *ev = getenv("XH_ROUTE_VALIDATE_EVERY");      // (read per call: a long-lived caller may change its mind)
        const int64_t every = ev ? (int64_t)atoll(ev) : (int64_t)1000;
        if (plan->validate_due && !feed) {      // the fed call that was due came back as an ordinary one: checked now
            first_check = first_check || !validate;
            plan->validate_due = false;
        } else {
            plan->dataflow_calls += 1;
            if (!validate && !first_check && every > 0 && plan->dataflow_calls % every == 0) {      // handled like the first one
                first_check = true;
                plan->validate_due = feed != nullptr;      // (a fed call cannot be checked at once: turned down below)
            }
        }
    }
    validate = validate || first_check;
    // a fed call cannot be cross-checked at once (the second routing would read runoff that does not exist yet), nor
    // routed by anything but the dataflow kernel that knows how to wait for it
    // (XH_ROUTE_TEST_FAULT is taken: the fault word is raised in front of the launch, the units that have to wait give up, and
    // the call is settled like any faulted one -- routed again from the runoff array, complete by then)
    if (feed && (validate || (flags & (XH_ROUTE_NO_DATAFLOW | XH_ROUTE_NO_SKEW | XH_ROUTE_FORCE_FALLBACK | XH_ROUTE_ATOMIC)) != 0))
        return XH_ERR_LIMIT;
    int rc = route_series_impl(ctx, plan, nmonths, spinup_months, h_ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff,
                               d_S0, d_chstorage, d_avgchflow, d_S_end, d_F_end, flags, &used_flow, feed);
    if (rc || !used_flow) return rc;
    // remember the call until a synchronisation has confirmed that no bounded wait timed out (xh_fault_check)
    xh_route_record r;
    r.plan = plan;
    r.nmonths = nmonths;
    r.spinup_months = spinup_months;
    r.flags = flags;
    r.ndays.assign(h_ndays, h_ndays + nmonths);
    r.dt = dt;
    r.flow_dist = d_flow_dist;
    r.velocity = d_velocity;
    r.area = d_area;
    r.runoff = d_runoff;
    r.S0 = d_S0;
    r.chs = d_chstorage;
    r.avg = d_avgchflow;
    r.S_end = d_S_end;
    r.F_end = d_F_end;
    r.seq_after = ctx->work_seq;
    r.fed = feed != nullptr;
    ctx->pending_routes.push_back(std::move(r));
    rc = xh_fault_collect(ctx);
    if (rc || !validate) return rc;
    const int64_t reroutes_before = ctx->reroutes;
    rc = route_validate(ctx, plan, nmonths, spinup_months, h_ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff, d_S0,
                        d_chstorage, d_avgchflow, d_S_end, d_F_end, flags);
    // (a call that had to be re-routed was not routed by the dataflow kernel in the end: nothing was checked)
    if (rc == XH_OK && ctx->reroutes == reroutes_before && (first_check || !first_checked_of(plan, last_form(plan))))
        first_check_passed(ctx, plan, last_form(plan));
    return rc;
}

int xh_route_rerun(xh_ctx *ctx, const xh_route_record &r, bool dataflow_pairs) {
    bool used_flow = false;
    if (!dataflow_pairs) r.plan->reroutes += 1;      // (guard re-runs are counted apart: xh_route_plan_rsum_info[6])
    int flags = r.flags & ~XH_ROUTE_TEST_FAULT;
    if (dataflow_pairs) {
        // A guard of the PREPARED reassociated plan tripped (a folded leaf that can fire with this call's data after all,
        // negative runoff or initial storage, a negative outflow leaving a halo): the prepared plan is given up until the
        // plan is prepared for other data, XH_ROUTE_NO_PLAIN below routes on the plan of pairs, which assumes nothing
        r.plan->guard_trips += 1;
        if (r.plan->last_rsum && r.plan->last_rsum_plan == r.plan->flow_rsum_fold) r.plan->fold_disabled = true;
        flags |= XH_ROUTE_NO_PLAIN;
    } else {
        flags |= XH_ROUTE_NO_DATAFLOW;
    }
    return route_series_impl(ctx, r.plan, r.nmonths, r.spinup_months, r.ndays.data(), r.dt, r.flow_dist, r.velocity,
                             r.area, r.runoff, r.S0, r.chs, r.avg, r.S_end, r.F_end, flags, &used_flow);
}

