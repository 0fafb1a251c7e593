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
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

#include "xh_mrtm_flow.h"

namespace {

constexpr int W_MAX = 9;          // terms per row: 8 D8 neighbours + the diagonal
constexpr int LANES = 64;         // cells per unit (one per lane)
constexpr int G_MAX = 16;         // imported streams, and outlets, per unit (16 = two block-transfer rounds of the skewed kernel)
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
                       &fp->d_ent2,         &fp->d_eprev,       &fp->d_unit_order,  &fp->d_unit_p,     &fp->d_unit_lmax,   &fp->d_unit_glmax};
    for (FlowBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (fp->d_x) (void)hipFree(fp->d_x);
    if (fp->d_skew_args) (void)hipFree(fp->d_skew_args);
    if (fp->d_stats) (void)hipFree(fp->d_stats);
    if (fp->d_trace) (void)hipFree(fp->d_trace);
    delete fp;
}

void flow_plan_info(const FlowPlan *fp, int64_t info[5]) {
    info[0] = fp ? fp->n_units : 0;
    info[1] = fp ? fp->n_edges : 0;
    info[2] = fp ? fp->depth : 0;
    info[3] = fp ? fp->n_cells : 0;
    info[4] = fp ? fp->max_imports : 0;
}

int flow_plan_build(xh_ctx *ctx, int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                    const std::vector<int> &comp, int ncomp, std::vector<char> &handled, FlowPlan **out) {
    *out = nullptr;
    handled.assign(n, 0);
    if (n == 0) return XH_OK;

    // ---- which networks are plain trees: rows are {-1 on the diagonal, +1 elsewhere}, every cell feeds <= 1 row,
    //      at most W_MAX terms per row, no cycle
    std::vector<int> ds(n, -1);
    std::vector<char> comp_ok(ncomp, 1);
    for (int r = 0; r < n; ++r) {
        int ndiag = 0;
        if (indptr[r + 1] - indptr[r] > W_MAX) comp_ok[comp[r]] = 0;
        for (int64_t j = indptr[r]; j < indptr[r + 1]; ++j) {
            const int c = indices[j];
            if (sign[j] < 0) {
                if (c == r) ++ndiag;
                else comp_ok[comp[r]] = 0;
            } else {
                if (c == r || ds[c] >= 0) comp_ok[comp[r]] = 0;
                ds[c] = r;
            }
        }
        if (ndiag != 1) comp_ok[comp[r]] = 0;
    }
    {   // cycles: follow the downstream pointers with three colours
        std::vector<char> colour(n, 0);
        std::vector<int> path;
        for (int s = 0; s < n; ++s) {
            if (colour[s]) continue;
            path.clear();
            int v = s;
            while (v >= 0 && colour[v] == 0) {
                colour[v] = 1;
                path.push_back(v);
                v = ds[v];
            }
            if (v >= 0 && colour[v] == 1) comp_ok[comp[v]] = 0;       // ran into the current path: a cycle
            for (int p : path) colour[p] = 2;
        }
    }

    // ---- bottom-up cut into connected pieces of <= LANES cells with <= G_MAX imported streams
    std::vector<int> nchild(n, 0);
    for (int c = 0; c < n; ++c)
        if (ds[c] >= 0) nchild[ds[c]]++;
    std::vector<int> child_ptr(n + 1, 0);
    for (int c = 0; c < n; ++c) child_ptr[c + 1] = child_ptr[c] + nchild[c];
    std::vector<int> child(child_ptr[n]);
    {
        std::vector<int> fill(child_ptr.begin(), child_ptr.end() - 1);
        for (int c = 0; c < n; ++c)
            if (ds[c] >= 0) child[fill[ds[c]]++] = c;
    }
    // longest side of every row either side of its diagonal (the time-skewed kernel reads that many pairs per sub-step)
    std::vector<int> cell_pre(n, 0), cell_post(n, 0);
    for (int c = 0; c < n; ++c) {
        bool past = false;
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
            if (indices[j] == c) past = true;
            else ++(past ? cell_post[c] : cell_pre[c]);
        }
    }
    const int simds = 4 * (ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 0);

    // ---- partition for one piece capacity `cap`: bottom-up cut into connected pieces of <= cap cells with <= G_MAX
    //      imported streams, then pieces packed into units of LANES cells.
    //      A smaller capacity than LANES costs streams (every cut is one) and buys units: pieces of 33..64 cells cannot
    //      share a unit, so the largest capacity leaves every other unit ~10 lanes short of full (67,420 cells: 1,121
    //      units at 64; 1,059 at 36 -- 1,054 would be every lane used -- with 1,699 streams instead of 861).  Units beyond the SIMD count
    //      share a SIMD with another unit and the slowest unit paces the run, so units are what counts.
    struct Partition {
        std::vector<int> queue, piece, closed_roots, piece_of_root, piece_size, piece_imp, piece_depth;
        std::vector<int> edge_prod_cell, edge_cons_cell, edge_of_prod;
        std::vector<int> unit_of_piece, unit_cells_n, unit_imp_n, unit_depth;
        std::vector<char> reached;
        int nunit = 0, nedge = 0, maxdepth = 0;
    };
    auto make_partition = [&](int cap, Partition &P) {
        P = Partition();
        std::vector<int> &queue = P.queue;
        queue.reserve(n);
        std::vector<int> left(nchild);
        for (int c = 0; c < n; ++c)
            if (comp_ok[comp[c]] && nchild[c] == 0) queue.push_back(c);
        std::vector<int> dsu(n);
        std::iota(dsu.begin(), dsu.end(), 0);
        auto find = [&](int x) {
            while (dsu[x] != x) {
                dsu[x] = dsu[dsu[x]];
                x = dsu[x];
            }
            return x;
        };
        std::vector<int> open_cnt(n, 0), open_imp(n, 0);
        std::vector<int> &closed_roots = P.closed_roots;            // piece roots in closing order (upstream pieces first)
        std::vector<int> kids;
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const int v = queue[qi];
            kids.assign(child.begin() + child_ptr[v], child.begin() + child_ptr[v + 1]);
            std::sort(kids.begin(), kids.end(), [&](int x, int y) {
                return open_cnt[x] != open_cnt[y] ? open_cnt[x] < open_cnt[y] : x < y;
            });
            int total = 1, imp = (int)kids.size();
            unsigned keep = 0;                          // bit i: kids[i]'s open piece joins v's
            for (size_t i = 0; i < kids.size(); ++i) {
                const int c = kids[i];
                if (total + open_cnt[c] <= cap && imp - 1 + open_imp[c] <= G_MAX) {
                    keep |= 1u << i;
                    total += open_cnt[c];
                    imp += open_imp[c] - 1;
                }
            }
            static const bool cut_rule = !(getenv("XH_FLOW_CUTRULE") && getenv("XH_FLOW_CUTRULE")[0] == '0');
            if (cut_rule && keep + 1 != (1u << kids.size()) && cell_pre[v] >= 3 && kids.size() <= 8) {
                // Not every child fits, and v's row has a long front side: which children become streams decides how many
                // pairs v reads per sub-step (a stream may only open the chain that sums the front side on the way, see
                // the time-skewed layout below).  Among the choices that fit: fewest reads for v, then most cells kept.
                auto v_reads = [&](unsigned kp) {
                    int j = 0, k = 0;
                    bool in_prefix = true;
                    for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
                        const int src = indices[e];
                        if (src == v) break;
                        bool kept = false;
                        for (size_t i = 0; i < kids.size(); ++i)
                            if (kids[i] == src) kept = (kp >> i) & 1u;
                        if (in_prefix && (kept || k == 0)) ++j;
                        else in_prefix = false;
                        ++k;
                    }
                    const int dir = j >= 2 ? 1 + k - j : k;
                    return (k - dir >= 2 ? dir + 1 : k) + cell_post[v];
                };
                // (the capacity is there for the packing; a piece around such a cell may grow up to a whole unit if that is
                // what it takes to keep its chain)
                int best_reads = v_reads(keep), best_total = total;
                for (unsigned kp = 0; kp < (1u << kids.size()); ++kp) {
                    int t = 1, im = (int)kids.size();
                    for (size_t i = 0; i < kids.size(); ++i)
                        if ((kp >> i) & 1u) {
                            t += open_cnt[kids[i]];
                            im += open_imp[kids[i]] - 1;
                        }
                    if (t > LANES || im > G_MAX) continue;
                    const int r = v_reads(kp);
                    const bool over = t > cap, best_over = best_total > cap;
                    if (r < best_reads || (r == best_reads && (over != best_over ? !over : t > best_total))) {
                        best_reads = r;
                        best_total = t;
                        keep = kp;
                    }
                }
                total = 1;
                imp = (int)kids.size();
                for (size_t i = 0; i < kids.size(); ++i)
                    if ((keep >> i) & 1u) {
                        total += open_cnt[kids[i]];
                        imp += open_imp[kids[i]] - 1;
                    }
            }
            for (size_t i = 0; i < kids.size(); ++i) {
                const int c = kids[i];
                if ((keep >> i) & 1u) dsu[find(c)] = v;      // c's open piece joins v's
                else closed_roots.push_back(c);             // c's piece is final; its outlet streams into v
            }
            open_cnt[v] = total;
            open_imp[v] = imp;
            if (ds[v] < 0) {
                closed_roots.push_back(v);
            } else if (--left[ds[v]] == 0) {
                queue.push_back(ds[v]);
            }
        }
        // cells of ok networks that were never reached (cannot happen for trees) stay unhandled
        P.reached.assign(n, 0);
        for (int v : queue) P.reached[v] = 1;

        // ---- pieces, their stream edges and pipeline depth
        const int npiece = (int)closed_roots.size();
        P.piece_of_root.assign(n, -1);
        for (int p = 0; p < npiece; ++p) P.piece_of_root[closed_roots[p]] = p;
        P.piece.assign(n, -1);
        P.piece_size.assign(npiece, 0);
        P.piece_imp.assign(npiece, 0);
        P.piece_depth.assign(npiece, 0);
        std::vector<int> ppre(npiece, 0), ppost(npiece, 0), pdir(npiece, 0);
        for (int c = 0; c < n; ++c)
            if (P.reached[c]) {
                const int q = P.piece_of_root[find(c)];
                P.piece[c] = q;
                P.piece_size[q]++;
                ppre[q] = std::max(ppre[q], cell_pre[c]);
                ppost[q] = std::max(ppost[q], cell_post[c]);
            }
        // front-side terms a cell still reads one by one when its unit is chained (see the time-skewed layout below): the
        // prefix of cells of its own piece (the first may be an imported stream) counts as one
        for (int c = 0; c < n; ++c)
            if (P.reached[c]) {
                const int q = P.piece[c];
                int j = 0, k = 0;
                bool in_prefix = true;
                for (int64_t e = indptr[c]; e < indptr[c + 1]; ++e) {
                    const int src = indices[e];
                    if (src == c) break;
                    if (in_prefix && (P.piece[src] == q || k == 0)) ++j;
                    else in_prefix = false;
                    ++k;
                }
                pdir[q] = std::max(pdir[q], j >= 2 ? 1 + k - j : k);
            }
        P.edge_of_prod.assign(n, -1);                        // one stream per closed piece that has a downstream cell
        for (int p = 0; p < npiece; ++p) {                   // closing order: upstream pieces come first
            const int r = closed_roots[p];
            if (ds[r] >= 0) {
                const int cp = P.piece[ds[r]];
                P.edge_of_prod[r] = (int)P.edge_prod_cell.size();
                P.edge_prod_cell.push_back(r);
                P.edge_cons_cell.push_back(ds[r]);
                P.piece_imp[cp]++;
                P.piece_depth[cp] = std::max(P.piece_depth[cp], P.piece_depth[p] + 1);
            }
        }
        P.nedge = (int)P.edge_prod_cell.size();
        P.maxdepth = npiece ? *std::max_element(P.piece_depth.begin(), P.piece_depth.end()) : 0;

        // ---- packing.  Pieces with a stream in or out: equal depth per unit (a unit then only ever waits for units
        //      strictly upstream or downstream of it), first-fit decreasing.  Pieces without streams -- whole small
        //      networks -- wait for nobody and go wherever lanes are free; the `cheap_units` cheapest of them (fewest row
        //      terms) are kept together instead: units for the SIMDs that must hold two waves (see the numbering below).
        auto has_out = [&](int p) { return ds[closed_roots[p]] >= 0; };
        // pairs a unit reads per sub-step: chained when that saves a read (front side - one-by-one terms >= 2)
        auto reads = [](int pre, int dir, int post) {
            return std::max(pre - dir >= 2 ? dir + 1 : pre, 1) + std::max(post, 1);
        };
        auto terms_of = [&](int p) { return reads(ppre[p], pdir[p], ppost[p]); };
        static const int tlimit = getenv("XH_FLOW_TLIMIT") ? atoi(getenv("XH_FLOW_TLIMIT")) : 5;
        std::vector<int> dep, fre;
        for (int p = 0; p < npiece; ++p) (P.piece_imp[p] > 0 || has_out(p) ? dep : fre).push_back(p);
        std::stable_sort(dep.begin(), dep.end(), [&](int x, int y) {
            return P.piece_depth[x] != P.piece_depth[y] ? P.piece_depth[x] < P.piece_depth[y]
                                                        : P.piece_size[x] > P.piece_size[y];
        });
        int cheap_units = 0;
        for (int round = 0; round < 4; ++round) {
            P.unit_of_piece.assign(npiece, -1);
            P.unit_cells_n.clear();
            P.unit_imp_n.clear();
            P.unit_depth.clear();
            std::vector<int> unit_out_n;                                   // outlets: <= G_MAX too
            std::vector<int> upre, udir, upost;                            // longest sides of the unit's rows
            auto new_unit = [&](int depth) {
                P.unit_cells_n.push_back(0);
                P.unit_imp_n.push_back(0);
                unit_out_n.push_back(0);
                upre.push_back(0);
                udir.push_back(0);
                upost.push_back(0);
                P.unit_depth.push_back(depth);
                return (int)P.unit_cells_n.size() - 1;
            };
            auto put_piece = [&](int p, int u) {
                P.unit_of_piece[p] = u;
                P.unit_cells_n[u] += P.piece_size[p];
                P.unit_imp_n[u] += P.piece_imp[p];
                unit_out_n[u] += has_out(p) ? 1 : 0;
                upre[u] = std::max(upre[u], ppre[p]);
                udir[u] = std::max(udir[u], pdir[p]);
                upost[u] = std::max(upost[u], ppost[p]);
            };
            // a piece joins a unit only if the unit then reads no more pairs per sub-step than `tlimit`, or than the piece
            // or the unit need on their own: the slowest unit paces the run, and it is the one with the longest rows
            auto class_ok = [&](int p, int u) {
                const int t = reads(std::max(upre[u], ppre[p]), std::max(udir[u], pdir[p]), std::max(upost[u], ppost[p]));
                return P.unit_cells_n[u] == 0 ||
                       t <= std::max(tlimit, std::max(terms_of(p), reads(upre[u], udir[u], upost[u])));
            };
            {
                size_t first_open = 0;
                int cur_depth = -1;
                for (int p : dep) {
                    if (P.piece_depth[p] != cur_depth) {
                        cur_depth = P.piece_depth[p];
                        first_open = P.unit_cells_n.size();
                    }
                    int u = -1;
                    for (size_t b = first_open; b < P.unit_cells_n.size(); ++b)
                        if (P.unit_cells_n[b] + P.piece_size[p] <= LANES && P.unit_imp_n[b] + P.piece_imp[p] <= G_MAX &&
                            unit_out_n[b] + (has_out(p) ? 1 : 0) <= G_MAX && class_ok(p, (int)b)) {
                            u = (int)b;
                            break;
                        }
                    if (u < 0) u = new_unit(cur_depth);
                    put_piece(p, u);
                    while (first_open < P.unit_cells_n.size() && P.unit_cells_n[first_open] >= LANES) ++first_open;
                }
            }
            // the cheap units: free pieces by (row terms, size), filled one unit after the other
            std::vector<int> by_terms(fre);
            std::stable_sort(by_terms.begin(), by_terms.end(), [&](int x, int y) {
                return terms_of(x) != terms_of(y) ? terms_of(x) < terms_of(y) : P.piece_size[x] < P.piece_size[y];
            });
            std::vector<char> taken(npiece, 0);
            {
                int made = 0, u = -1;
                for (int p : by_terms) {
                    if (terms_of(p) > 3) break;
                    if (u < 0 || P.unit_cells_n[u] + P.piece_size[p] > LANES) {
                        if (made == cheap_units) break;
                        u = new_unit(0);
                        ++made;
                    }
                    put_piece(p, u);
                    taken[p] = 1;
                }
            }
            // the other free pieces: largest first, each into the fullest unit that still takes it
            std::vector<int> by_size;
            for (int p : fre)
                if (!taken[p]) by_size.push_back(p);
            std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) { return P.piece_size[x] > P.piece_size[y]; });
            {
                // units by free lanes: bucket[f] = units with f free lanes
                std::vector<std::vector<int>> bucket(LANES + 1);
                for (int u = 0; u < (int)P.unit_cells_n.size(); ++u) bucket[LANES - P.unit_cells_n[u]].push_back(u);
                for (int p : by_size) {
                    const int sz = P.piece_size[p];
                    int u = -1;
                    for (int pass = 0; pass < 2 && u < 0; ++pass)          // second pass: any unit with room
                        for (int f = sz; f <= LANES && u < 0; ++f)
                            for (size_t i = bucket[f].size(); i-- > 0;)
                                if (pass == 1 || class_ok(p, bucket[f][i])) {
                                    u = bucket[f][i];
                                    bucket[f].erase(bucket[f].begin() + (long)i);
                                    break;
                                }
                    if (u < 0) u = new_unit(0);
                    put_piece(p, u);
                    bucket[LANES - P.unit_cells_n[u]].push_back(u);
                }
            }
            P.nunit = (int)P.unit_cells_n.size();
            const int need = simds > 0 ? std::max(P.nunit - simds, 0) : 0;
            if (need <= cheap_units) break;
            cheap_units = need + (round > 0 ? 2 : 0);      // the cheap units themselves may add a unit or two
        }
    };
    Partition P;
    {
        // the capacity with the fewest units wins (ties: the larger capacity = fewer streams)
        static const int caps[] = {LANES, 56, 48, 44, 40, 36, 32};
        int forced = 0;
        if (const char *env = getenv("XH_FLOW_PIECE_CAP")) forced = std::min(std::max(atoi(env), 1), LANES);      // experiments
        Partition Q;
        bool have = false;
        long best_score = 0;
        for (int cap : caps) {
            if (forced) cap = forced;
            make_partition(cap, Q);
            // units beyond the SIMD count share a SIMD; only units without streams may (see the numbering below): a unit
            // with streams that has to share one slows every unit it is linked to, which costs far more than a few units
            int indep = 0;
            {
                std::vector<char> coupled(Q.nunit, 0);
                for (size_t p = 0; p < Q.closed_roots.size(); ++p)
                    if (Q.piece_imp[p] > 0 || ds[Q.closed_roots[p]] >= 0) coupled[Q.unit_of_piece[p]] = 1;
                for (int u = 0; u < Q.nunit; ++u) indep += coupled[u] ? 0 : 1;
            }
            const int extra = simds > 0 ? std::max(Q.nunit - simds, 0) : 0;
            const long score = 1000000L * std::max(2 * extra - indep, 0) + 1000L * Q.nunit + Q.nedge / 8;
            if (getenv("XH_FLOW_DEBUG"))
                fprintf(stderr, "flow plan: piece capacity %d -> %d units, %d streams, %d units without streams\n", cap,
                        Q.nunit, Q.nedge, indep);
            if (!have || score < best_score) {
                std::swap(P, Q);
                best_score = score;
                have = true;
            }
            if (forced || (simds > 0 && P.nunit <= simds)) break;       // every unit has a SIMD of its own: good enough
        }
    }
    std::vector<int> &queue = P.queue, &piece = P.piece, &closed_roots = P.closed_roots, &piece_of_root = P.piece_of_root;
    std::vector<int> &piece_size = P.piece_size, &piece_imp = P.piece_imp, &piece_depth = P.piece_depth;
    std::vector<int> &edge_prod_cell = P.edge_prod_cell, &edge_cons_cell = P.edge_cons_cell, &edge_of_prod = P.edge_of_prod;
    std::vector<int> &unit_of_piece = P.unit_of_piece, &unit_cells_n = P.unit_cells_n, &unit_imp_n = P.unit_imp_n;
    std::vector<int> &unit_depth = P.unit_depth;
    std::vector<char> &reached = P.reached;
    (void)reached;
    (void)piece_size;
    (void)piece_imp;
    const int npiece = (int)closed_roots.size();
    const int nedge = P.nedge, maxdepth = P.maxdepth;
    const int nunit = P.nunit;
    if (nunit == 0) return XH_OK;

    // ---- which unit runs where.  With more units than SIMDs some SIMDs hold two waves; the slowest unit paces the run,
    //      and two waves on a SIMD take about as long as their instruction streams put together (+50 % for a 5-term unit
    //      next to a 2-term one).  A unit with streams passes its delay on to every unit downstream and, through the ring
    //      limits, upstream of it; a unit without streams only delays itself.  So the SIMDs with two waves should hold
    //      units without streams, a cheap one next to a dearer one that gets issue priority.  Which workgroup lands on
    //      which SIMD cannot be planned: it follows the workgroup id only on an idle device (measured: with the ABCD
    //      kernel's last waves still draining, 38 SIMDs instead of 34 got two workgroups, ids unrelated, and the call
    //      took 31 ms instead of 25.6).  The kernel therefore lets every workgroup find out where it runs and claim its
    //      unit from this list (xh_mrtm_skew.hip, top of k_mrtm_skew): units without streams by rising cost, then the others.
    std::vector<int> unit_order(nunit);      // filled below, once the row shapes of the units are known

    // ---- slots, ghosts, gather offsets
    const int64_t ts = (int64_t)nunit * LANES;
    std::vector<int> cell_of_slot(ts, -1), export_edge(ts, -1), ghost_edge(ts, -1), slot_of_cell(n, -1);
    std::vector<int> fill(nunit, 0), gfill(nunit, 0), edge_cons_unit(nedge), edge_ghost(nedge);
    for (int c = 0; c < n; ++c)
        if (piece[c] >= 0) {
            const int u = unit_of_piece[piece[c]];
            const int s = fill[u]++;
            cell_of_slot[(int64_t)u * LANES + s] = c;
            slot_of_cell[c] = s;
            handled[c] = 1;
        }
    for (int ed = 0; ed < nedge; ++ed) {
        const int u = unit_of_piece[piece[edge_cons_cell[ed]]];
        const int g = gfill[u]++;
        edge_cons_unit[ed] = u;
        edge_ghost[ed] = g;
        ghost_edge[(int64_t)u * LANES + g] = ed;
        const int pc = edge_prod_cell[ed];
        export_edge[(int64_t)unit_of_piece[piece[pc]] * LANES + slot_of_cell[pc]] = ed;
    }
    std::vector<unsigned> ent((size_t)W_MAX * ts, (unsigned)(NPAIR - 1) * 16u);
    std::vector<int> unit_terms(nunit, 1);
    for (int c = 0; c < n; ++c) {
        if (piece[c] < 0) continue;
        const int u = unit_of_piece[piece[c]];
        unit_terms[u] = std::max(unit_terms[u], (int)(indptr[c + 1] - indptr[c]));
        const int64_t slot = (int64_t)u * LANES + slot_of_cell[c];
        int w = 0;
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j, ++w) {
            const int src = indices[j];
            unsigned off;
            if (piece[src] >= 0 && unit_of_piece[piece[src]] == u) {
                off = (unsigned)slot_of_cell[src] * 16u + (sign[j] < 0 ? 8u : 0u);
            } else {                                    // the outlet of an upstream piece in another unit
                const int ed = edge_of_prod[src];
                if (ed < 0 || edge_cons_unit[ed] != u)
                    return xh_fail(ctx, XH_ERR_ARG, "flow plan: inconsistent stream edge at cell %d", c);
                off = (unsigned)(LANES + edge_ghost[ed]) * 16u;
            }
            ent[(size_t)w * ts + slot] = off;
        }
    }

    // ---- time-skewed layout (xh_mrtm_skew.hip).  Lane lags: a cell `h` edges above its piece's outlet runs
    //      2 * (H - h) sub-steps behind the unit's clock (H = tallest piece of the unit, an imported stream counting as
    //      one more level), so that every flow a cell gathers was produced exactly two iterations earlier.  The lags
    //      of a unit are shifted so that its outlets' lag is a multiple of 16 (stream stores of 16 sub-steps never
    //      wrap inside a group).  Row terms are split at the diagonal: SK_P before, SK_P after.
    constexpr int SK_P = 4;
    constexpr unsigned SK_ZERO = 2u * LANES * 16u;
    bool skew_ok = true;
    // Chained units (xh_mrtm_skew.hip, CHAIN): the cells that feed a cell from in front of its diagonal, as far as they
    // are lanes of the unit from the first one on (an imported stream ends the chain: its pair is dropped into LDS by the
    // block transfers, not computed by a lane), pass a running sum along their stored order; the fed cell reads the last
    // one's pair as one term and the rest of its front side term by term.  A unit is chained when that saves at least
    // one read per sub-step: longest front side (reads saved + 1 for the running pair) at least 2 shorter.
    std::vector<char> unit_chain(nunit, 0);
    std::vector<int> chain_extra(n, 0), chain_prev(n, -1), chain_len(n, 0);   // levels below the last of the chain; cell before
    std::vector<int> edge_reader(edge_cons_cell);      // the cell whose lane reads an imported pair out of LDS
    {
        static const bool chain_env = !(getenv("XH_FLOW_CHAIN") && getenv("XH_FLOW_CHAIN")[0] == '0');
        std::vector<int> upre(nunit, 0), udir(nunit, 0);
        auto front = [&](int c, int u, int &k) {       // k = terms in front of the diagonal, returns the chainable prefix:
            int j = 0;                                 // lanes of the unit, the first one possibly an imported stream
            bool in_prefix = true;
            k = 0;
            for (int64_t e = indptr[c]; e < indptr[c + 1]; ++e) {
                const int src = indices[e];
                if (src == c) break;
                const bool inu = piece[src] >= 0 && unit_of_piece[piece[src]] == u;
                if (in_prefix && (inu || k == 0)) ++j;
                else in_prefix = false;
                ++k;
            }
            return j;
        };
        for (int c = 0; c < n; ++c) {
            if (piece[c] < 0) continue;
            const int u = unit_of_piece[piece[c]];
            int k;
            const int j = front(c, u, k);
            upre[u] = std::max(upre[u], k);
            udir[u] = std::max(udir[u], j >= 2 ? 1 + k - j : k);
        }
        for (int u = 0; u < nunit; ++u) unit_chain[u] = chain_env && upre[u] - udir[u] >= 2 && udir[u] <= 2;
        for (int c = 0; c < n; ++c) {
            if (piece[c] < 0) continue;
            const int u = unit_of_piece[piece[c]];
            if (!unit_chain[u]) continue;
            int k;
            const int j = front(c, u, k);
            if (j < 2) continue;
            chain_len[c] = j;
            int prev = -1;
            for (int i = 0; i < j; ++i) {
                const int t = indices[indptr[c] + i];
                const bool inu = piece[t] >= 0 && unit_of_piece[piece[t]] == u;
                if (!inu) {             // an imported stream opens the chain: the next cell adds its flows to the ghost pair
                    edge_reader[edge_of_prod[t]] = indices[indptr[c] + 1];
                    prev = -2 - edge_of_prod[t];
                    continue;
                }
                chain_extra[t] = j - 1 - i;
                chain_prev[t] = prev;
                prev = t;
            }
        }
    }
    std::vector<int> hgt(n, 0), unit_h(nunit, 0);
    for (size_t qi = queue.size(); qi-- > 0;) {          // reverse bottom-up order: downstream cells first
        const int c = queue[qi];
        if (piece[c] < 0) continue;
        hgt[c] = (piece_of_root[c] == piece[c]) ? 0 : hgt[ds[c]] + 1 + chain_extra[c];
        int &uh = unit_h[unit_of_piece[piece[c]]];
        uh = std::max(uh, hgt[c]);
    }
    for (int ed = 0; ed < nedge; ++ed) {
        int &uh = unit_h[edge_cons_unit[ed]];
        uh = std::max(uh, hgt[edge_reader[ed]] + 1);
    }
    std::vector<int> lag(ts, 0), ghost_lag(ts, 0), unit_p(nunit, 0x11), unit_lmax(nunit, 0), unit_glmax(nunit, 0);
    std::vector<unsigned> ent2((size_t)2 * SK_P * ts, SK_ZERO), eprev(ts, SK_ZERO);
    for (int u = 0; u < nunit; ++u) unit_lmax[u] = (2 * unit_h[u] + 15) & ~15;
    for (int c = 0; c < n; ++c) {
        if (piece[c] < 0) continue;
        const int u = unit_of_piece[piece[c]];
        const int64_t slot = (int64_t)u * LANES + slot_of_cell[c];
        lag[slot] = unit_lmax[u] - 2 * hgt[c];
        if (chain_prev[c] >= 0) eprev[slot] = (unsigned)slot_of_cell[chain_prev[c]] * 16u;
        else if (chain_prev[c] <= -2) eprev[slot] = (unsigned)(LANES + edge_ghost[-2 - chain_prev[c]]) * 16u;
        int npre = 0, npost = 0, seen = 0;
        bool past = false;
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
            const int src = indices[j];
            if (src == c) {
                past = true;
                continue;
            }
            if (!past && ++seen < chain_len[c]) continue;      // summed on the way: only the last of the chain is read
            unsigned off;
            if (piece[src] >= 0 && unit_of_piece[piece[src]] == u) off = (unsigned)slot_of_cell[src] * 16u;
            else off = (unsigned)(LANES + edge_ghost[edge_of_prod[src]]) * 16u;
            int &k = past ? npost : npre;
            if (k >= SK_P) {
                skew_ok = false;
                continue;
            }
            ent2[(size_t)((past ? SK_P : 0) + k) * ts + slot] = off;
            ++k;
        }
        unit_p[u] = std::max(unit_p[u] & 15, npre) | (std::max((unit_p[u] >> 4) & 15, npost) << 4) | (unit_chain[u] ? 0x100 : 0);
    }
    for (int ed = 0; ed < nedge; ++ed) {
        const int u = edge_cons_unit[ed];
        const int gl = unit_lmax[u] - 2 * (hgt[edge_reader[ed]] + 1);
        ghost_lag[(int64_t)u * LANES + edge_ghost[ed]] = gl;
        unit_glmax[u] = std::max(unit_glmax[u], gl);
    }

    std::vector<int> unit_exp(nunit, 0);
    for (int ed = 0; ed < nedge; ++ed) unit_exp[unit_of_piece[piece[edge_prod_cell[ed]]]]++;
    {   // the claim list (see "which unit runs where"): units without streams by rising cost, then the others.  Cost as
        // measured (tools/flow_stats.py): ~25 cycles per pair read per sub-step, ~15 for imports, ~15 for outlets.
        std::vector<int> cost(nunit);
        for (int u = 0; u < nunit; ++u)
            cost[u] = 25 * ((unit_p[u] & 15) + ((unit_p[u] >> 4) & 15) + ((unit_p[u] & 0x100) ? 1 : 0)) +
                      (unit_imp_n[u] > 0 ? 15 : 0) + (unit_exp[u] > 0 ? 15 : 0);
        auto coupled = [&](int u) { return unit_imp_n[u] > 0 || unit_exp[u] > 0; };
        std::iota(unit_order.begin(), unit_order.end(), 0);
        std::stable_sort(unit_order.begin(), unit_order.end(), [&](int x, int y) {
            return coupled(x) != coupled(y) ? !coupled(x) : cost[x] < cost[y];
        });
    }
    if (getenv("XH_FLOW_DEBUG")) {      // partition statistics on stderr
        std::vector<int> hp(8, 0), hi(9, 0), hx(9, 0), hl(10, 0), hpp(25, 0);
        int n_chain = 0;
        auto bucket = [](int v) { return v == 0 ? 0 : v <= 1 ? 1 : v <= 2 ? 2 : v <= 4 ? 3 : v <= 8 ? 4 : v <= 16 ? 5 : v <= 32 ? 6 : 7; };
        for (int u = 0; u < nunit; ++u) {
            hp[std::max(unit_p[u] & 15, (unit_p[u] >> 4) & 15)]++;
            hpp[(unit_p[u] & 15) * 5 + ((unit_p[u] >> 4) & 15)]++;
            if (unit_p[u] & 0x100) ++n_chain;
            hi[bucket(unit_imp_n[u])]++;
            hx[bucket(unit_exp[u])]++;
            hl[std::min(unit_lmax[u] / 16, 9)]++;
        }
        fprintf(stderr, "flow plan: %d units, %d pieces, %d edges, depth %d, skew_ok %d\n", nunit, npiece, nedge,
                maxdepth + 1, (int)skew_ok);
        {   // pieces by the longest (pre, post) side of their rows, and the cells per class
            std::vector<int> ppre(npiece, 0), ppost(npiece, 0), hpc(25, 0), hcells(25, 0);
            for (int c = 0; c < n; ++c) {
                if (piece[c] < 0) continue;
                int a = 0, b = 0;
                bool past = false;
                for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
                    if (indices[j] == c) past = true;
                    else ++(past ? b : a);
                }
                ppre[piece[c]] = std::max(ppre[piece[c]], a);
                ppost[piece[c]] = std::max(ppost[piece[c]], b);
            }
            for (int q = 0; q < npiece; ++q) {
                hpc[ppre[q] * 5 + ppost[q]]++;
                hcells[ppre[q] * 5 + ppost[q]] += piece_size[q];
            }
            fprintf(stderr, "  pieces (cells) by (pre, post):");
            for (int a = 0; a <= 4; ++a)
                for (int b = 0; b <= 4; ++b)
                    if (hpc[a * 5 + b]) fprintf(stderr, " (%d,%d) %d (%d)", a, b, hpc[a * 5 + b], hcells[a * 5 + b]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "  chained units: %d\n", n_chain);
        fprintf(stderr, "  units by P (1..4):");
        for (int k = 1; k <= 4; ++k) fprintf(stderr, " %d", hp[k]);
        fprintf(stderr, "\n  units by (pre, post) terms:");
        for (int a = 1; a <= 4; ++a)
            for (int b = 1; b <= 4; ++b) fprintf(stderr, " (%d,%d) %d", a, b, hpp[a * 5 + b]);
        fprintf(stderr, "\n  units by imports (0,1,2,<=4,<=8,<=16,<=32,more):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hi[k]);
        fprintf(stderr, "\n  units by exports (0,1,2,<=4,<=8,<=16,<=32,more):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hx[k]);
        fprintf(stderr, "\n  units by lmax/16 (0..9+):");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", hl[k]);
        fprintf(stderr, "\n");
    }

    if (const char *dump = getenv("XH_FLOW_DUMP")) {      // partition as int32 rows [n]: downstream cell, piece, unit, height
        if (FILE *f = fopen(dump, "wb")) {
            std::vector<int> row(n);
            fwrite(&n, sizeof(int), 1, f);
            fwrite(ds.data(), sizeof(int), n, f);
            fwrite(piece.data(), sizeof(int), n, f);
            for (int c = 0; c < n; ++c) row[c] = piece[c] >= 0 ? unit_of_piece[piece[c]] : -1;
            fwrite(row.data(), sizeof(int), n, f);
            fwrite(hgt.data(), sizeof(int), n, f);
            fclose(f);
        }
    }

    FlowPlan *fp = new FlowPlan();
    fp->skew_ok = skew_ok;
    fp->skew_lmax = *std::max_element(unit_lmax.begin(), unit_lmax.end());
    {   // longest jump of a stream over pipeline levels: the ring of such a stream has to hold what the levels in between
        // need as lead (xh_mrtm_skew.hip, ring size)
        int span = 1;
        for (int ed = 0; ed < nedge; ++ed)
            span = std::max(span, piece_depth[piece[edge_cons_cell[ed]]] - piece_depth[piece[edge_prod_cell[ed]]]);
        fp->skew_span = span;
        if (getenv("XH_FLOW_DEBUG")) fprintf(stderr, "  longest stream jump: %d levels\n", span);
    }
    fp->n_units = nunit;
    fp->n_edges = nedge;
    fp->depth = maxdepth + 1;
    fp->n_cells = (int)std::count(handled.begin(), handled.end(), (char)1);
    fp->max_imports = *std::max_element(unit_imp_n.begin(), unit_imp_n.end());
    fp->max_exports = *std::max_element(unit_exp.begin(), unit_exp.end());
    int rc = put(ctx, fp->d_cell_of_slot, cell_of_slot);
    rc |= put(ctx, fp->d_ent, ent);
    rc |= put(ctx, fp->d_export_edge, export_edge);
    rc |= put(ctx, fp->d_ghost_edge, ghost_edge);
    rc |= put(ctx, fp->d_edge_cons_unit, edge_cons_unit);
    rc |= put(ctx, fp->d_unit_terms, unit_terms);
    rc |= put(ctx, fp->d_lag, lag);
    rc |= put(ctx, fp->d_ghost_lag, ghost_lag);
    rc |= put(ctx, fp->d_ent2, ent2);
    rc |= put(ctx, fp->d_eprev, eprev);
    rc |= put(ctx, fp->d_unit_p, unit_p);
    rc |= put(ctx, fp->d_unit_order, unit_order);
    rc |= put(ctx, fp->d_unit_lmax, unit_lmax);
    rc |= put(ctx, fp->d_unit_glmax, unit_glmax);
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
    if (const char *path = getenv("XH_FLOW_TRACE")) {
        if (fp->d_trace && fp->trace_words) {
            std::vector<unsigned> tr(fp->trace_words);
            XH_HIP(ctx, hipMemcpy(tr.data(), fp->d_trace, tr.size() * 4, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "wb")) {
                const int nu = fp->n_units;
                fwrite(&nu, 4, 1, f);
                fwrite(tr.data(), 4, tr.size(), f);
                fclose(f);
            }
        }
    }
    return XH_OK;
}

int flow_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st) {
    if (!fp || fp->n_units == 0) return XH_OK;
    // exchange block: streams, then the counters (zeroed every call)
    const size_t x_streams = (size_t)std::max(fp->n_edges, 1) * RING * (size_t)s.ntmax * sizeof(double2);
    const size_t x_cnt = ((size_t)(fp->n_edges + fp->n_units) * sizeof(unsigned) + 255) & ~size_t(255);
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
    int per_cu = (fp->n_units + cus - 1) / cus + 1;
    {
        const char *env = getenv("XH_FLOW_PER_CU_EXTRA");     // experiments only
        if (env) per_cu += atoi(env);
    }
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
