// MRTM routing as a dataflow of single-wave units (gfx950): the partition of the tree networks into units
// (flow_plan_build, also used by the time-skewed kernel of xh_mrtm_skew.hip, which is the default) and the lock-step
// kernel k_mrtm_flow, kept for schedules whose months are shorter than the skewed kernel's lane lags and as an
// independently written second implementation that the tests hold to the same bits.
//
// One workgroup per river network (xh_mrtm.hip) is bounded by the largest network: all of its cells share one CU,
// which then issues ~70 instructions x (cells / 64) per sub-step while 200 other CUs idle.  But the dependency in
//     dS_i/dt = sum_{j upstream of i} F_j - F_i + lateral_i          (mrtm.py:50-51)
// runs one way only: a cell needs the flows of the cells UPSTREAM of it, never those downstream.  A tributary can
// therefore be integrated months ahead of the river it joins.  This file cuts every tree-shaped network into
// connected pieces of at most 64 cells, packs pieces of equal pipeline depth into UNITS of 64 lanes (one wave, one
// cell per lane), and links the units by one-way streams in HBM:
//
//   - inside a unit the sub-step loop is what xh_mrtm.hip does, minus the workgroup barriers: the two flow exchanges
//     per sub-step go through LDS, whose operations complete in issue order for a single wave;
//   - a piece's outlet lane appends {trial flow F, adjusted flow F2} of every sub-step to its stream (one 16-byte
//     cached store per sub-step, so the L2 merges them into whole lines); at the end of a month it publishes the
//     month: s_waitcnt vmcnt(0) -> agent-scope release -> s_waitcnt vmcnt(0) -> relaxed agent-scope store of the
//     month counter.  (Write-through sc1 stores without the release were measured at the same speed but 9.8 GB
//     instead of 3.6 GB of HBM writes per 720-month run, because every 8-byte store goes to the fabric.)
//   - the consuming unit waits for that counter at the start of the same month (relaxed agent-scope polls with
//     s_sleep, one agent-scope acquire), then its "ghost" lanes read the stream eight sub-steps ahead into
//     registers and drop each pair into ghost slots of the LDS flow buffers, where the consuming cell's gather
//     finds them like any other neighbour.  The values and the order of every sum are unchanged, so results stay
//     bit-identical to numpy/scipy;
//   - streams are rings of RING months; a producer that would lap its consumer waits on the consumer's progress
//     counter.  Units only ever wait on units strictly upstream (data) or downstream (ring space) of themselves,
//     pieces of one unit have the same depth, and the launch keeps every unit resident, so the waits cannot cycle.
//     Every spin is bounded by the 100 MHz real-time counter and raises the context's fault word instead of hanging.
//
// Throughput is then set by the sub-step latency of ONE wave (a few hundred cycles) instead of the instruction
// issue of the largest network, and the whole chip is busy: 1,121 units over 256 CUs for the 67,420-cell grid.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

#include "xh_flow_plan.h"
#include "xh_mrtm_flow.h"

namespace {

constexpr int W_MAX = 9;          // terms per row: 8 D8 neighbours + the diagonal
constexpr int LANES = 64;         // cells per unit (one per lane)
constexpr int NPAIR = 2 * LANES + 1;       // LDS pairs per flow buffer: cells, ghost slots (one per lane), constant zero
constexpr int RING = 4;           // months of stream kept in HBM per edge
constexpr int PF = 8;             // sub-steps of ghost prefetch held in registers
constexpr unsigned FAULT_DATA_WAIT = 1, FAULT_RING_WAIT = 2;
constexpr unsigned FAULT_TEST = 99;
constexpr unsigned long long SPIN_LIMIT_TICKS = 500000000ull;   // 5 s of the 100 MHz real-time counter (see xh_mrtm_skew.hip)

struct FlowArgs {
    const int *cell_of_slot;        // [units*64] global cell id or -1
    const unsigned *ent;            // [W_MAX][units*64] byte offsets into a flow buffer
    const int *export_edge;         // [units*64] stream this lane's cell feeds, or -1
    const int *ghost_edge;          // [units*64] stream that ghost slot `lane` of the unit imports, or -1
    const int *edge_cons_unit;      // [edges]
    const int *unit_terms;          // [units] longest row (terms) of each unit
    int64_t total_slots;
    int nmonths, nit, ntmax;
    const int *sched_m, *sched_nt;
    const double *sched_secs;
    const unsigned char *sched_write;
    double dt, dtinv;
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *chs, *avg, *S_end, *F_end;
    double2 *xbuf;                  // [edges][RING][ntmax] {F, F2}
    unsigned *ready;                // [edges] months published
    unsigned *done;                 // [units] months consumed
    unsigned *fault;
    unsigned long long *stats;      // [units][6] optional cycle accounting (XH_FLOW_STATS=1)
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All lanes with `need` wait until *p >= target.  Returns false (and raises the fault word) on timeout / fault.
__device__ __forceinline__ bool wave_wait_ge(bool need, const unsigned *p, unsigned target, unsigned *fault,
                                             unsigned code) {
    bool ok = !need;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (!ok) ok = ld_relaxed(p) >= target;
        if (__all(ok)) return true;
        if (ld_relaxed(fault) != 0) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) {
            __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// Whole series for one unit; WU = terms gathered per row (the unit's longest row, rounded up to 3, 5, 7 or 9): LDS
// instructions from a lone wave are slow, so a unit without big confluences should not issue nine reads per gather.
template <int WU>
__device__ __forceinline__ void flow_unit(const FlowArgs &a, double2 *lds) {
    double2 *bufA = lds;               // trial flows {F, -F}: cells 0..63, ghosts 64..127, zero 128
    double2 *bufB = lds + NPAIR;       // adjusted flows {F2, -F2}
    // Gather addresses as absolute 32-bit LDS addresses, formed once: the dynamic-LDS base is a link-time constant the
    // compiler cannot fold, so "base + offset" inside the loop costs one extra VALU instruction per gathered term.
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    typedef __attribute__((address_space(3))) const char lds_cchar;
    lds_cchar *ldsA = (lds_cchar *)bufA;
    constexpr unsigned B_OFF = NPAIR * sizeof(double2);      // bufB = bufA + B_OFF: an immediate offset in ds_read
    const int lane = threadIdx.x, unit = blockIdx.x;
    const int64_t slot = (int64_t)unit * LANES + lane;

    const int gc = a.cell_of_slot[slot];
    const bool valid = gc >= 0;
    const double tauinv = valid ? a.velocity[gc] / a.flow_dist[gc] : 0.0;      // mrtm.py:40
    const double area = valid ? a.area[gc] : 0.0;
    double S = (valid && a.S0) ? a.S0[gc] : 0.0;
    double F = 0.0;
    lds_cchar *e[WU];
#pragma unroll
    for (int w = 0; w < WU; ++w) e[w] = ldsA + a.ent[(int64_t)w * a.total_slots + slot];
    const int xedge = a.export_edge[slot];
    const int gedge = a.ghost_edge[slot];
    const bool has_x = xedge >= 0, has_g = gedge >= 0;
    const bool any_x = __any(has_x), any_g = __any(has_g);
    const unsigned *ready_p = a.ready + (has_g ? gedge : 0);
    const unsigned *done_p = a.done + (has_x ? a.edge_cons_unit[xedge] : 0);

    bufA[LANES + lane] = make_double2(0.0, 0.0);
    bufB[LANES + lane] = make_double2(0.0, 0.0);
    if (lane == 0) {
        bufA[NPAIR - 1] = make_double2(0.0, 0.0);
        bufB[NPAIR - 1] = make_double2(0.0, 0.0);
    }
    const double dt = a.dt, dtinv = a.dtinv;
    double qn = valid ? a.runoff[(int64_t)gc * a.nmonths + a.sched_m[0]] : 0.0;
    bool alive = true;
    unsigned long long cyc_loop = 0, cyc_wait_data = 0, cyc_wait_ring = 0;
    double ob_s[8], ob_a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ob_s[j] = ob_a[j] = 0.0;
    const unsigned long long cyc_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();

    for (int it = 0; it < a.nit && alive; ++it) {
        const int m = a.sched_m[it], nt = a.sched_nt[it];
        const double secs = a.sched_secs[it];
        const double erl = (qn * area) * 1000.0 / secs;                        // mrtm.py:45
        double favg = 0.0;
        if (it + 1 < a.nit) qn = valid ? a.runoff[(int64_t)gc * a.nmonths + a.sched_m[it + 1]] : 0.0;

        const int rs = it % RING;
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        if (any_g) {      // the streams this unit imports must hold month `it`
            alive = wave_wait_ge(has_g, ready_p, (unsigned)it + 1u, a.fault, FAULT_DATA_WAIT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const unsigned long long w1 = __builtin_amdgcn_s_memtime();
        if (any_x && it >= RING && alive)   // ring slot `rs` must have been consumed (month it - RING)
            alive = wave_wait_ge(has_x, done_p, (unsigned)(it - RING) + 1u, a.fault, FAULT_RING_WAIT);
        cyc_wait_data += w1 - w0;
        cyc_wait_ring += __builtin_amdgcn_s_memtime() - w1;
        if (!alive) break;
        const double2 *iptr = a.xbuf + ((int64_t)(has_g ? gedge : 0) * RING + rs) * a.ntmax;
        double2 *optr = a.xbuf + ((int64_t)(has_x ? xedge : 0) * RING + rs) * a.ntmax;

        double2 q[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) q[j] = (has_g && j < nt) ? iptr[j] : make_double2(0.0, 0.0);

        auto substep = [&](int t, double2 gv) {
            F = S * tauinv;                                                    // mrtm.py:50
            bufA[lane] = make_double2(F, -F);
            if (has_g) reinterpret_cast<double *>(bufA + LANES + lane)[0] = gv.x;   // ghosts are only ever added (+1 terms)
            __builtin_amdgcn_wave_barrier();
            double v[WU];
#pragma unroll
            for (int w = 0; w < WU; ++w) v[w] = *(lds_cdouble *)e[w];
            double acc = 0.0;                                                  // UM.dot(F), row order (mrtm.py:51)
#pragma unroll
            for (int w = 0; w < WU; ++w) acc += v[w];
            const double dsdt = acc + erl;
            const bool sx = (dsdt * dt) < (-S);                                // mrtm.py:54
            const double f2 = sx ? (dsdt + F) + S * dtinv : F;                 // mrtm.py:60
            if (has_x) optr[t] = make_double2(F, f2);
            // mrtm.py:56-76: the flows are gathered a second time only "if Sx.any()".  Here "any" is decided per unit:
            // if no cell of this unit fired and no imported flow was adjusted upstream (F2 == F), every F2 this unit
            // gathers equals the F it already gathered, so the second sum is bit-identical to the first.
            if (__any(sx || (has_g && gv.x != gv.y))) {
                S = sx ? 0.0 : S;                                              // mrtm.py:63
                bufB[lane] = make_double2(f2, -f2);
                if (has_g) reinterpret_cast<double *>(bufB + LANES + lane)[0] = gv.y;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int w = 0; w < WU; ++w) v[w] = *(lds_cdouble *)(e[w] + B_OFF);
                double acc2 = 0.0;                                             // UM.dot(F) with the adjusted flows
#pragma unroll
                for (int w = 0; w < WU; ++w) acc2 += v[w];
                const double dsdt2 = acc2 + erl;                               // mrtm.py:68
                S = sx ? S : S + dsdt2 * dt;                                   // mrtm.py:69
            } else {
                S = S + dsdt * dt;                                             // mrtm.py:76
            }
            F = f2;
            favg += f2;                                                        // mrtm.py:78
            __builtin_amdgcn_wave_barrier();
        };

        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        int t0 = 0;
        for (; t0 + PF <= nt; t0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const double2 gv = q[j];
                if (has_g && t0 + j + PF < nt) q[j] = iptr[t0 + j + PF];
                substep(t0 + j, gv);
            }
        }
        for (int t = t0; t < nt; ++t) substep(t, (has_g) ? iptr[t] : make_double2(0.0, 0.0));
        cyc_loop += __builtin_amdgcn_s_memtime() - c0;

        // Outputs leave as whole 64-byte groups of 8 months per cell (a lone 8-byte store per month made the L2 write
        // the same line back many times: 13 GB of HBM writes for 0.65 GB of output).
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            ob_s[j] = ob_s[j + 1];
            ob_a[j] = ob_a[j + 1];
        }
        ob_s[7] = S;
        ob_a[7] = favg / (double)nt;                                           // mrtm.py:80
        if (a.sched_write[it] && valid) {
            if ((m & 7) == 7) {
                const int64_t o = (int64_t)gc * a.nmonths + (m - 7);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if (a.chs) *reinterpret_cast<double2 *>(a.chs + o + j) = make_double2(ob_s[j], ob_s[j + 1]);
                    if (a.avg) *reinterpret_cast<double2 *>(a.avg + o + j) = make_double2(ob_a[j], ob_a[j + 1]);
                }
            } else if (m == a.nmonths - 1) {                                    // last, partial group
                const int r = (m & 7) + 1;
                const int64_t o = (int64_t)gc * a.nmonths + (m + 1 - r);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j >= 8 - r) {
                        if (a.chs) a.chs[o + j - (8 - r)] = ob_s[j];
                        if (a.avg) a.avg[o + j - (8 - r)] = ob_a[j];
                    }
            }
        }
        if (any_x) {      // publish month `it`
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // cached stream stores: one L2 write-back per month
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (has_x) __hip_atomic_store(a.ready + xedge, (unsigned)it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (any_g && lane == 0)   // this month's imports are consumed: their ring slots may be reused
            __hip_atomic_store(a.done + unit, (unsigned)it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (valid) {
        if (a.S_end) a.S_end[gc] = S;
        if (a.F_end) a.F_end[gc] = F;
    }
    if (a.stats && lane == 0) {      // shader cycles inside the sub-step loops / whole unit, real-time ticks, shape
        unsigned long long *st = a.stats + (int64_t)unit * 6;
        st[4] = cyc_wait_data;
        st[5] = cyc_wait_ring;
        st[0] = cyc_loop;
        st[1] = __builtin_amdgcn_s_memtime() - cyc_begin;
        st[2] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        // HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20): where this wave ran (placement diagnostics)
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        st[3] = (unsigned long long)WU | (any_g ? 16u : 0u) | (any_x ? 32u : 0u) | ((unsigned long long)hw << 8) |
                ((unsigned long long)(xcc & 15u) << 40);
    }
}

__global__ void __launch_bounds__(LANES) k_mrtm_flow(FlowArgs a) {
    // Static LDS: its base is a compile-time constant that folds into the ds_read/ds_write offset field.  (The base of
    // `extern __shared__` memory is resolved after instruction selection and cost one v_add per gathered term.)  The
    // launch still requests dynamic LDS, unused, purely to bound the workgroups per CU (see flow_launch).
    __shared__ __attribute__((aligned(16))) double2 lds[2 * NPAIR];
    const int wu = a.unit_terms[blockIdx.x];        // uniform per workgroup
    if (wu <= 3) flow_unit<3>(a, lds);
    else if (wu <= 5) flow_unit<5>(a, lds);
    else if (wu <= 7) flow_unit<7>(a, lds);
    else flow_unit<W_MAX>(a, lds);
}

template <typename T>
int put(xh_ctx *ctx, FlowBuf &b, const std::vector<T> &v) {
    XH_HIP(ctx, hipMalloc(&b.p, v.empty() ? 16 : v.size() * sizeof(T)));
    if (!v.empty()) XH_HIP(ctx, hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return XH_OK;
}

}  // namespace

void flow_plan_destroy(FlowPlan *fp) {
    if (!fp) return;
    FlowBuf *bufs[] = {&fp->d_cell_of_slot, &fp->d_ent,      &fp->d_export_edge, &fp->d_ghost_edge,
                       &fp->d_edge_cons_unit, &fp->d_unit_terms, &fp->d_lag,        &fp->d_ghost_lag,
                       &fp->d_ent2,         &fp->d_eprev,       &fp->d_unit_order,  &fp->d_unit_p,     &fp->d_unit_lmax,   &fp->d_unit_glmax,
                       &fp->d_lane_flags,   &fp->d_fold_cell};
    for (FlowBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (fp->d_x) (void)hipFree(fp->d_x);
    if (fp->d_skew_args) (void)hipFree(fp->d_skew_args);
    if (fp->d_stats) (void)hipFree(fp->d_stats);
    delete fp;
}

void flow_plan_info(const FlowPlan *fp, int64_t info[5]) {
    info[0] = fp ? fp->n_units : 0;
    info[1] = fp ? fp->n_edges : 0;
    info[2] = fp ? fp->depth : 0;
    info[3] = fp ? fp->n_cells : 0;
    info[4] = fp ? fp->max_imports : 0;
}

// The partition itself is host code without any HIP in it (xh_flow_plan.cpp: also built with sanitizers for the host
// fuzzer, tests/plan_fuzz); here its tables go to the device.  The experiment switches of the planner are read once.
FlowPlanOptions flow_plan_options(const xh_ctx *ctx) {
    FlowPlanOptions o;
    o.simds = 4 * (ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 0);
    if (const char *e = getenv("XH_FLOW_PIECE_CAP")) o.piece_cap = std::min(std::max(atoi(e), 1), LANES);
    o.debug = xh_flow_debug();
    return o;
}

int flow_tables_host(FlowPlanOptions opt, int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                     const std::vector<int> &comp, int ncomp, std::vector<char> &handled, FlowTables &t, std::string &err) {
    handled.assign(n, 0);
    t = FlowTables();
    if (n == 0) return 0;
    opt.capable = nullptr;      // every unit in pair form
    if (flow_tables_build(n, indptr, indices, sign, comp.data(), ncomp, opt, handled, t, err) != 0) return -1;
    if (t.n_units == 0) return 0;
    if (xh_flow_check()) {      // the invariants the host fuzzer holds the planner to, on this very plan
        const std::string bad = flow_tables_check(n, indptr, indices, sign, handled, t);
        if (!bad.empty()) {
            err = "flow plan check: " + bad;
            return -1;
        }
    }
    return 0;
}

int flow_plan_build(xh_ctx *ctx, int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                    const std::vector<int> &comp, int ncomp, std::vector<char> &handled, FlowPlan **out) {
    *out = nullptr;
    FlowTables t;
    std::string err;
    const FlowPlanOptions opt = flow_plan_options(ctx);
    // The all-pairs partition of a grid is the same every time (topology, planner options, library build): 50-70 ms of host
    // time at the full grid, a few to read back.  Kept in the per-box cache (XH_CACHE_DIR; XH_ROUTE_LEARN_CACHE=0 switches the
    // caches of partitions off) and held to the planner's own invariant checker before it is used.
    std::string cache;
    if (xh_plan_cache_on() && n > 0 && !opt.debug) {
        const std::string dir = xh_cache_dir();
        if (!dir.empty()) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](const void *p, size_t nbytes) {
                const unsigned char *b = static_cast<const unsigned char *>(p);
                for (size_t i = 0; i < nbytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
            };
            const int64_t nnz = indptr[n];
            mix(indptr, sizeof(int64_t) * (size_t)(n + 1));
            if (nnz) mix(indices, sizeof(int32_t) * (size_t)nnz);
            if (nnz) mix(sign, (size_t)nnz);
            const int knobs[5] = {opt.simds, opt.piece_cap, opt.chain, opt.cut_rule, opt.tlimit};
            mix(knobs, sizeof(knobs));
            const char *stamp = __DATE__ " " __TIME__;
            mix(stamp, strlen(stamp));
            char name[96];
            snprintf(name, sizeof(name), "/pairs_%016llx_%d.tables", (unsigned long long)h, n);
            cache = dir + name;
            if (flow_tables_load(cache.c_str(), t) && t.n_units > 0 && (int64_t)t.cell_of_slot.size() == (int64_t)t.n_units * LANES) {
                handled.assign(n, 0);
                bool ok = true;
                for (int c : t.cell_of_slot) {
                    if (c >= n) ok = false;
                    else if (c >= 0) handled[c] = 1;
                }
                if (ok && flow_tables_check(n, indptr, indices, sign, handled, t).empty()) return flow_plan_upload(ctx, t, out);
            }
        }
    }
    if (flow_tables_host(opt, n, indptr, indices, sign, comp, ncomp, handled, t, err) != 0)
        return xh_fail(ctx, XH_ERR_ARG, "%s", err.c_str());
    if (!cache.empty() && t.n_units > 0) {
        const std::string dir = cache.substr(0, cache.rfind('/'));
        for (size_t i = 1; i <= dir.size(); ++i)      // mkdir -p
            if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);
        (void)flow_tables_save(t, cache.c_str());
    }
    return flow_plan_upload(ctx, t, out);
}

int flow_plan_upload(xh_ctx *ctx, const FlowTables &t, FlowPlan **out) {
    *out = nullptr;
    if (t.n_units == 0) return XH_OK;
    FlowPlan *fp = new FlowPlan();
    fp->skew_ok = t.skew_ok;
    fp->skew_lmax = t.skew_lmax;
    fp->skew_span = t.skew_span;
    fp->n_units = t.n_units;
    fp->n_edges = t.n_edges;
    fp->depth = t.depth;
    fp->n_cells = t.n_cells;
    fp->max_imports = t.max_imports;
    fp->max_exports = t.max_exports;
    fp->rsum = t.rsum;
    for (int c : t.cell_of_slot) fp->max_cell = std::max(fp->max_cell, c);
    for (int c : t.fold_of_slot) fp->max_cell = std::max(fp->max_cell, c);
    int rc = put(ctx, fp->d_cell_of_slot, t.cell_of_slot);
    rc |= put(ctx, fp->d_ent, t.ent);
    rc |= put(ctx, fp->d_export_edge, t.export_edge);
    rc |= put(ctx, fp->d_ghost_edge, t.ghost_edge);
    rc |= put(ctx, fp->d_edge_cons_unit, t.edge_cons_unit);
    rc |= put(ctx, fp->d_unit_terms, t.unit_terms);
    rc |= put(ctx, fp->d_lag, t.lag);
    rc |= put(ctx, fp->d_ghost_lag, t.ghost_lag);
    rc |= put(ctx, fp->d_ent2, t.ent2);
    rc |= put(ctx, fp->d_eprev, t.eprev);
    rc |= put(ctx, fp->d_unit_p, t.unit_p);
    rc |= put(ctx, fp->d_unit_order, t.unit_order);
    rc |= put(ctx, fp->d_unit_lmax, t.unit_lmax);
    rc |= put(ctx, fp->d_unit_glmax, t.unit_glmax);
    rc |= put(ctx, fp->d_lane_flags, t.lane_flags);
    if (!t.fold_of_slot.empty()) rc |= put(ctx, fp->d_fold_cell, t.fold_of_slot);
    fp->n_folded = t.n_folded;
    fp->n_special = t.n_special;
    fp->n_pair_units = t.n_pair_units;
    if (rc) {
        flow_plan_destroy(fp);
        return XH_ERR_HIP;
    }
    *out = fp;
    return XH_OK;
}

int flow_stats_fetch(xh_ctx *ctx, FlowPlan *fp, std::vector<unsigned long long> &out) {
    out.clear();
    if (!fp || !fp->d_stats) return XH_OK;
    out.resize((size_t)fp->n_units * 6);
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    XH_HIP(ctx, hipMemcpy(out.data(), fp->d_stats, out.size() * 8, hipMemcpyDeviceToHost));
    return XH_OK;
}

int flow_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st) {
    if (!fp || fp->n_units == 0) return XH_OK;
    // exchange block: streams, then the counters (zeroed every call)
    const size_t x_streams = (size_t)std::max(fp->n_edges, 1) * RING * (size_t)s.ntmax * sizeof(double2);
    const size_t x_cnt = ((size_t)(fp->n_edges + fp->n_units) * sizeof(unsigned) + 255) & ~size_t(255);
    fp->rec_key = 0;      // (k_mrtm_wave keeps its month records in this buffer too: this launch lays it out its own way)
    if (x_streams + x_cnt > fp->x_bytes) {
        if (fp->d_x) {
            XH_HIP(ctx, hipStreamSynchronize(st));
            XH_HIP(ctx, hipFree(fp->d_x));
            fp->d_x = nullptr;
        }
        XH_HIP(ctx, hipMalloc(&fp->d_x, x_streams + x_cnt));
        fp->x_bytes = x_streams + x_cnt;
    }
    unsigned *cnt = reinterpret_cast<unsigned *>(static_cast<char *>(fp->d_x) + x_streams);
    XH_HIP(ctx, hipMemsetAsync(cnt, 0, x_cnt, st));

    // Every unit should be resident at once (a unit that is not resident stalls the units feeding it through the
    // ring back-pressure until it is dispatched).  Asking for a share of the CU's LDS bounds the workgroups per CU,
    // which also spreads the units over the chip; the share leaves room for ONE MORE workgroup per CU than the even
    // split needs, because the even split itself (5 x 32 KiB on a 160 KiB CU) was measured to admit one fewer than
    // hipOccupancyMaxActiveBlocksPerMultiprocessor reports (routing 18.0 ms -> 10.7 ms for 120 months).
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int per_cu = (fp->n_units + cus - 1) / cus + 1;
    const size_t lds_static = 2 * (size_t)NPAIR * sizeof(double2);
    const size_t share = ((size_t)(160 * 1024) / (size_t)per_cu) & ~size_t(1023);
    size_t lds = share > lds_static + 1024 ? share - lds_static : 0;      // dynamic part on top of the static buffers
    XH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mrtm_flow), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    int resident = 0;
    XH_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_mrtm_flow, LANES, lds));
    if ((int64_t)(resident - 1) * cus < fp->n_units)
        return xh_fail(ctx, XH_ERR_LIMIT, "dataflow routing: %d units but only (%d - 1) x %d can be resident", fp->n_units,
                       resident, cus);

    FlowArgs a;
    a.cell_of_slot = static_cast<const int *>(fp->d_cell_of_slot.p);
    a.ent = static_cast<const unsigned *>(fp->d_ent.p);
    a.export_edge = static_cast<const int *>(fp->d_export_edge.p);
    a.ghost_edge = static_cast<const int *>(fp->d_ghost_edge.p);
    a.edge_cons_unit = static_cast<const int *>(fp->d_edge_cons_unit.p);
    a.unit_terms = static_cast<const int *>(fp->d_unit_terms.p);
    a.total_slots = (int64_t)fp->n_units * LANES;
    a.nmonths = s.nmonths;
    a.nit = s.nit;
    a.ntmax = s.ntmax;
    a.sched_m = s.d_m;
    a.sched_nt = s.d_nt;
    a.sched_secs = s.d_secs;
    a.sched_write = s.d_wr;
    a.dt = s.dt;
    a.dtinv = 1.0 / s.dt;
    a.flow_dist = io.flow_dist;
    a.velocity = io.velocity;
    a.area = io.area;
    a.runoff = io.runoff;
    a.S0 = io.S0;
    a.chs = io.chs;
    a.avg = io.avg;
    a.S_end = io.S_end;
    a.F_end = io.F_end;
    a.xbuf = static_cast<double2 *>(fp->d_x);
    a.ready = cnt;
    a.done = cnt + fp->n_edges;
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
    // the fault word is zeroed on the context stream; `st` may be a different stream ordered after it by the caller
    hipLaunchKernelGGL(k_mrtm_flow, dim3((unsigned)fp->n_units), dim3(LANES), lds, st, a);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}
