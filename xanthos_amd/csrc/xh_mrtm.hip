// Modified River Transport Model (MRTM) channel routing on gfx950.
//
// Replaces xanthos/routing/mrtm.py:streamrouting (:16-82) and the spin-up / simulation month loops of
// xanthos/components.py:calculate_routing (:273-294).  The reference does, per 3-hour sub-step, one or two
// scipy CSR mat-vecs with UM = UP - I plus ~8 numpy vector ops, 8*nday sub-steps per month, one Python call per
// month.  600 months are ~146,000 strictly sequential sub-steps, so launch-per-sub-step is hopeless.
//
// Design: the flow graph splits into independent river networks (weakly-connected components of UM).  A
// network (or a bin of small networks) is a "unit" that one workgroup owns for the WHOLE series: per-cell state
// (storage S, running mean flow, 1/tau, lateral inflow) lives in registers, the flows F exchanged between
// neighbours live in LDS, and the month and sub-step loops run inside one persistent kernel with two workgroup
// barriers per sub-step.  HBM is touched once per cell-month (read runoff, write ChStorage / Avg_ChFlow).
//
// Exactness: row i of UM.dot(F) is accumulated as 0 + sum_j sign_j * F[col_j] in stored (ascending column)
// order, exactly like scipy's csr_matvec, and the reference's "any excess flow" branch is evaluated in the
// equivalent two-phase per-cell form (identical when no cell fires), with -ffp-contract=off.
// Results are bit-identical to the numpy/scipy path for identical runoff.
//
// Networks that do not fit a workgroup (more than 3072 cells, or a row with more than 9 entries) are routed by
// the global-memory kernels at the bottom (two launches per sub-step; optional fp64 atomic scatter-add variant).
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <atomic>
#include <numeric>
#include <thread>

#include "xh_common.h"
#include <sys/stat.h>
#include <unistd.h>

#include "xh_mrtm_flow.h"
#include "xh_mrtm_plan.h"

namespace {

constexpr int W_MAX = 9;             // 8 D8 neighbours + the diagonal
constexpr int W_BASE = 3;            // terms per row gathered branch-free (diagonal + two tributaries)
constexpr int UNIT_MAX_CELLS = 3072;  // 44 B of LDS per cell: pair buffers + tail-term table
constexpr int BIN_CELLS = 256;       // small networks share a single-wave workgroup of up to this many cells
constexpr int N_CLASS = XH_ROUTE_N_CLASS;

struct UnitClass {
    int nt, kc;
};
// shapes tried in order: first one with nt*kc >= cells
constexpr UnitClass CLASSES[N_CLASS] = {{64, 1}, {64, 2}, {64, 4}, {256, 2}, {256, 4}, {1024, 2}, {768, 4}, {768, 4}};

struct RouteArgs {
    const int *unit_slot0;           // [units of this launch] first slot of each unit
    const int *cell_of_slot;         // [total_slots] global cell id, -1 = padding
    const unsigned *ent;             // [W_MAX][total_slots] byte offset of each gathered term in the pair buffer
    const unsigned char *cnt;        // [total_slots] number of terms
    int64_t total_slots;
    int nmonths, nit;
    const int *sched_m;              // [nit] month index routed at iteration it (spin-up months first)
    const int *sched_nt;             // [nit] sub-steps
    const double *sched_secs;        // [nit] nday*24*3600
    const unsigned char *sched_write;  // [nit] 1 = store outputs (simulation pass)
    double dt, dtinv;
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *chs, *avg, *S_end, *F_end;
};

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// acc[k] = 0 + sum of the row's terms in stored order, read from the pair buffer at `base`.
// Stripe 0 holds the unit's highest-degree cells (cells are sorted by number of terms), so it carries all W_MAX term
// offsets in registers and gathers them branch-free; the other stripes carry W_BASE and fall back to the LDS table
// xe [W_MAX - W_BASE][NP] (offsets in 8-byte units) only in units with more confluence cells than stripe 0 holds.
template <int KC>
__device__ __forceinline__ void gather_terms(const char *base, const unsigned (&e0)[W_MAX], const unsigned (&e)[KC][W_BASE],
                                             double (&acc)[KC], const int (&W)[KC], const unsigned short *xe, int NP,
                                             int NT, int tid) {
    double v0[W_MAX], v[KC][W_BASE];
#pragma unroll
    for (int w = 0; w < W_MAX; ++w) v0[w] = *reinterpret_cast<const double *>(base + e0[w]);
#pragma unroll
    for (int k = 1; k < KC; ++k)
#pragma unroll
        for (int w = 0; w < W_BASE; ++w) v[k][w] = *reinterpret_cast<const double *>(base + e[k][w]);
    {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < W_MAX; ++w) s += v0[w];
        acc[0] = s;
    }
#pragma unroll
    for (int k = 1; k < KC; ++k) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < W_BASE; ++w) s += v[k][w];
        acc[k] = s;
    }
#pragma unroll
    for (int k = 1; k < KC; ++k) {
        if (W[k] > W_BASE) {                                 // wave-uniform and rare
            for (int w = W_BASE; w < W[k]; ++w)
                acc[k] += *reinterpret_cast<const double *>(base + 8u * xe[(w - W_BASE) * NP + k * NT + tid]);
        }
    }
}

// One workgroup = one routing unit for the whole series.  blockDim.x = NT, each thread owns KC cells
// (slot l = k*NT + tid, so consecutive lanes touch consecutive LDS pairs).
template <int KC, int NTMAX>
__global__ void __launch_bounds__(NTMAX) k_mrtm_units(RouteArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 lds[];
    const int NT = blockDim.x, tid = threadIdx.x;
    const int NP = NT * KC;
    double2 *bufA = lds;                // trial flows  {F, -F}
    double2 *bufB = lds + (NP + 1);     // final flows  {F2, -F2}; slot NP of each buffer is the constant {0,0}
    unsigned short *xe = reinterpret_cast<unsigned short *>(lds + 2 * (NP + 1));   // [W_MAX - W_BASE][NP]
    const char *baseA = reinterpret_cast<const char *>(bufA);
    const char *baseB = reinterpret_cast<const char *>(bufB);
    const int slot0 = a.unit_slot0[blockIdx.x];

    // Gather terms: the first W_BASE terms of every row are read unconditionally (rows with fewer terms point at
    // the constant-zero slot, adding +0.0 is exact), so the loads of all stripes issue back to back with no branch
    // in between; rows with more terms (confluences of 3+ tributaries, a few per cent of the cells, sorted to the
    // front of the unit) finish in a wave-uniform tail loop.
    int gc[KC];
    double S[KC], tauinv[KC], area[KC], erl[KC], favg[KC], F[KC], qn[KC];
    unsigned e0[W_MAX], e[KC][W_BASE];
    int W[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int64_t slot = (int64_t)slot0 + k * NT + tid;
        gc[k] = a.cell_of_slot[slot];
        const bool valid = gc[k] >= 0;
        tauinv[k] = valid ? a.velocity[gc[k]] / a.flow_dist[gc[k]] : 0.0;     // mrtm.py:40
        area[k] = valid ? a.area[gc[k]] : 0.0;
        S[k] = (valid && a.S0) ? a.S0[gc[k]] : 0.0;
        F[k] = 0.0;
        W[k] = __builtin_amdgcn_readfirstlane(wave_max_i32((int)a.cnt[slot]));   // most terms of any row in this stripe
#pragma unroll
        for (int w = 0; w < W_BASE; ++w) e[k][w] = a.ent[(int64_t)w * a.total_slots + slot];
        if (k == 0) {
#pragma unroll
            for (int w = 0; w < W_MAX; ++w) e0[w] = a.ent[(int64_t)w * a.total_slots + slot];
        }
        for (int w = W_BASE; w < W_MAX; ++w)
            xe[(w - W_BASE) * NP + k * NT + tid] = (unsigned short)(a.ent[(int64_t)w * a.total_slots + slot] >> 3);
        qn[k] = valid ? a.runoff[(int64_t)gc[k] * a.nmonths + a.sched_m[0]] : 0.0;
    }
    if (tid == 0) {
        bufA[NP] = make_double2(0.0, 0.0);
        bufB[NP] = make_double2(0.0, 0.0);
    }
    const double dt = a.dt, dtinv = a.dtinv;

    for (int it = 0; it < a.nit; ++it) {
        const int m = a.sched_m[it], nt = a.sched_nt[it];
        const double secs = a.sched_secs[it];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            erl[k] = (qn[k] * area[k]) * 1000.0 / secs;                       // mrtm.py:45
            favg[k] = 0.0;
        }
        if (it + 1 < a.nit) {                                                  // prefetch next month's runoff
            const int m2 = a.sched_m[it + 1];
#pragma unroll
            for (int k = 0; k < KC; ++k)
                qn[k] = gc[k] >= 0 ? a.runoff[(int64_t)gc[k] * a.nmonths + m2] : 0.0;
        }
        for (int t = 0; t < nt; ++t) {
            bool sx[KC];
            double acc[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                F[k] = S[k] * tauinv[k];                                       // mrtm.py:50
                bufA[k * NT + tid] = make_double2(F[k], -F[k]);
            }
            __syncthreads();
            gather_terms<KC>(baseA, e0, e, acc, W, xe, NP, NT, tid);        // UM.dot(F), row order (mrtm.py:51)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const double dsdt = acc[k] + erl[k];
                sx[k] = (dsdt * dt) < (-S[k]);                                 // mrtm.py:54
                const double f2 = sx[k] ? (dsdt + F[k]) + S[k] * dtinv : F[k]; // mrtm.py:60
                S[k] = sx[k] ? 0.0 : S[k];                                     // mrtm.py:63
                F[k] = f2;
                bufB[k * NT + tid] = make_double2(f2, -f2);
            }
            __syncthreads();
            gather_terms<KC>(baseB, e0, e, acc, W, xe, NP, NT, tid);        // UM.dot(F) again, adjusted flows
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const double dsdt = acc[k] + erl[k];                           // mrtm.py:68
                S[k] = sx[k] ? S[k] : S[k] + dsdt * dt;                        // mrtm.py:69 / :76
                favg[k] += F[k];                                               // mrtm.py:78
            }
        }
        if (a.sched_write[it]) {
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                if (gc[k] >= 0) {
                    const int64_t o = (int64_t)gc[k] * a.nmonths + m;
                    if (a.chs) a.chs[o] = S[k];
                    if (a.avg) a.avg[o] = favg[k] / (double)nt;                // mrtm.py:80
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        if (gc[k] >= 0) {
            if (a.S_end) a.S_end[gc[k]] = S[k];
            if (a.F_end) a.F_end[gc[k]] = F[k];
        }
    }
}

// ------------------------------------------------------------------------------------------- global fallback
struct FbArgs {
    int n;                           // cells routed by the fallback
    const int *cells;                // [n] global cell ids
    const int *ptr;                  // [n+1] CSR over `cells` order
    const int *col;                  // global cell id of each term
    const signed char *sgn;
    const int *ds;                   // [n] single downstream cell (atomic variant), -1 = none
    int nmonths;
    double dt, dtinv;
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *S, *F, *F2, *favg, *erl, *inflow;   // [ncell] work arrays in cell order
    unsigned char *sx;
    double *chs, *avg, *S_end, *F_end;
};

__global__ void __launch_bounds__(256) k_fb_init(FbArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    a.S[c] = a.S0 ? a.S0[c] : 0.0;
    a.F2[c] = 0.0;
}

__global__ void __launch_bounds__(256) k_fb_month_begin(FbArgs a, int m, double secs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    a.erl[c] = (a.runoff[(int64_t)c * a.nmonths + m] * a.area[c]) * 1000.0 / secs;
    a.favg[c] = 0.0;
    a.F[c] = a.S[c] * (a.velocity[c] / a.flow_dist[c]);
    if (a.inflow) a.inflow[c] = 0.0;
}

// phase A: trial step, excess-flow test, adjusted flow F2
template <bool ATOMIC>
__global__ void __launch_bounds__(256) k_fb_phase_a(FbArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    double acc = 0.0;
    if (ATOMIC) {
        acc = a.inflow[c] - a.F[c];
        a.inflow[c] = 0.0;
    } else {
        for (int j = a.ptr[i]; j < a.ptr[i + 1]; ++j) acc += a.sgn[j] > 0 ? a.F[a.col[j]] : -a.F[a.col[j]];
    }
    const double dsdt = acc + a.erl[c];
    const double S = a.S[c], F = a.F[c];
    const bool sx = (dsdt * a.dt) < (-S);
    a.F2[c] = sx ? (dsdt + F) + S * a.dtinv : F;
    if (sx) a.S[c] = 0.0;
    a.sx[c] = sx ? 1 : 0;
}

// phase B: final storage update, running mean, next trial flow
template <bool ATOMIC>
__global__ void __launch_bounds__(256) k_fb_phase_b(FbArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    double acc = 0.0;
    if (ATOMIC) {
        acc = a.inflow[c] - a.F2[c];
        a.inflow[c] = 0.0;
    } else {
        for (int j = a.ptr[i]; j < a.ptr[i + 1]; ++j) acc += a.sgn[j] > 0 ? a.F2[a.col[j]] : -a.F2[a.col[j]];
    }
    const double dsdt = acc + a.erl[c];
    double S = a.S[c];
    if (!a.sx[c]) S = S + dsdt * a.dt;
    a.S[c] = S;
    a.favg[c] += a.F2[c];
    a.F[c] = S * (a.velocity[c] / a.flow_dist[c]);
}

// scatter-add of each cell's outflow to its downstream cell: global_atomic_add_f64 (-munsafe-fp-atomics)
__global__ void __launch_bounds__(256) k_fb_scatter(FbArgs a, int which) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    const int d = a.ds[i];
    if (d >= 0) unsafeAtomicAdd(&a.inflow[d], which ? a.F2[c] : a.F[c]);
}

__global__ void __launch_bounds__(256) k_fb_month_end(FbArgs a, int m, int nt, int write) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    if (write) {
        const int64_t o = (int64_t)c * a.nmonths + m;
        if (a.chs) a.chs[o] = a.S[c];
        if (a.avg) a.avg[o] = a.favg[c] / (double)nt;
    }
}

__global__ void __launch_bounds__(256) k_fb_finish(FbArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int c = a.cells[i];
    if (a.S_end) a.S_end[c] = a.S[c];
    if (a.F_end) a.F_end[c] = a.F2[c];
}

template <typename T>
int upload(xh_ctx *ctx, DevBuf &b, const std::vector<T> &v) {
    b.bytes = v.size() * sizeof(T);
    XH_HIP(ctx, hipMalloc(&b.p, b.bytes ? b.bytes : 16));
    if (b.bytes) XH_HIP(ctx, hipMemcpy(b.p, v.data(), b.bytes, hipMemcpyHostToDevice));
    return XH_OK;
}

}  // namespace

namespace {

int find_root(std::vector<int> &parent, int x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}

void free_buf(DevBuf &b) {
    if (b.p && !b.pooled) (void)hipFree(b.p);
    b.p = nullptr;
    b.pooled = false;
}

// The tables of a plan in ONE device allocation and ONE copy: xh_route_plan_create used to make some thirty hipMalloc +
// hipMemcpy pairs (tens of milliseconds beside run_model()'s forcing upload, which holds the same runtime locks).
struct UploadPool {
    struct Item { DevBuf *buf; size_t off, bytes; };
    std::vector<char> host;
    std::vector<Item> items;
    template <typename T>
    void add(DevBuf &b, const std::vector<T> &v) {
        const size_t bytes = v.size() * sizeof(T), off = host.size();
        host.resize(off + ((bytes ? bytes : 16) + 255) / 256 * 256);      // empty tables still get an address of their own
        if (bytes) memcpy(host.data() + off, v.data(), bytes);
        items.push_back({&b, off, bytes});
    }
    int commit(xh_ctx *ctx, void **base) {
        XH_HIP(ctx, hipMalloc(base, host.size() ? host.size() : 256));
        if (!host.empty()) XH_HIP(ctx, hipMemcpy(*base, host.data(), host.size(), hipMemcpyHostToDevice));
        for (const Item &it : items) {
            it.buf->p = (char *)*base + it.off;
            it.buf->bytes = it.bytes;
            it.buf->pooled = true;
        }
        return XH_OK;
    }
};

template <int KC, int NTMAX>
void launch_units(const xh_route_plan *plan, int cls, RouteArgs args, hipStream_t st, bool rest_only) {
    const int nt = CLASSES[cls].nt;
    size_t lds = 2 * (size_t)(nt * KC + 1) * sizeof(double2) + (size_t)(W_MAX - W_BASE) * nt * KC * sizeof(unsigned short);
    // Placement control: the dispatcher stacks small workgroups on one CU until a resource runs out, and a stack of
    // routing waves saturates that CU's LDS pipe while other CUs idle.  Asking for an even share of the 160 KiB
    // makes at most ceil(units / CUs) workgroups fit per CU, so the units spread over the whole chip.
    {
        const int cus = plan->ctx->prop.multiProcessorCount > 0 ? plan->ctx->prop.multiProcessorCount : 256;
        const int64_t per_cu = ((rest_only ? plan->n_rest_units : plan->n_units) + cus - 1) / cus;
        const size_t share = ((size_t)(160 * 1024) / (size_t)(per_cu > 0 ? per_cu : 1)) & ~size_t(1023);
        if (share > lds) lds = share;
    }
    const std::vector<int> &list = rest_only ? plan->rest_units[cls] : plan->class_units[cls];
    args.unit_slot0 = static_cast<const int *>((rest_only ? plan->d_rest_units[cls] : plan->d_class_units[cls]).p);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mrtm_units<KC, NTMAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_mrtm_units<KC, NTMAX>), dim3((unsigned)list.size()), dim3(nt), lds, st, args);
}

}  // namespace

static void route_plan_free(xh_route_plan *plan, bool settle);

// Which form routes the tree networks of a call.  The call's flag decides (XH_ROUTE_EXACT / XH_ROUTE_REASSOC), else
// XH_ROUTE_REASSOC=1 / 0 in the environment, else the library default: the reassociated form (round 5: <= 3e-12 of the
// bit-exact kernel on every routed value of the full grid against a gate of 1e-6, 15.3 against 22.6 ms).
#ifndef XH_REASSOC_DEFAULT
#define XH_REASSOC_DEFAULT 1
#endif
static int reassoc_env() {
    static const int v = [] {
        const char *e = getenv("XH_ROUTE_REASSOC");
        return !e ? -1 : (e[0] == '0' ? 0 : 1);
    }();
    return v;
}
bool reassoc_wanted(int flags) {
    if (flags & XH_ROUTE_EXACT) return false;
    if (flags & XH_ROUTE_REASSOC) return true;
    return reassoc_env() >= 0 ? reassoc_env() == 1 : XH_REASSOC_DEFAULT != 0;
}

// The reassociated partition of the plan's tree networks, from the per-box cache or from the host planner
// (xh_flow_rsum.cpp).  `foldable` (or nullptr): the leaves the parents' lanes may carry (FlowPlanOptions::foldable).  The
// partition of a grid is the same every time (topology, planner options, foldable set, library build): it is kept in the
// cache beside the bit-exact one (flow_plan_build) and held to the planner's own invariant checker before it is used.
// False: the planner has nothing for this grid.
static bool rsum_tables_get(xh_ctx *ctx, xh_route_plan *plan, const unsigned char *foldable, const unsigned char *capable,
                            FlowTables &t, std::vector<char> &handled) {
    std::string err;
    FlowPlanOptions opt = flow_plan_options(ctx);
    opt.foldable = foldable;
    opt.capable = capable;
    if (const char *e = getenv("XH_RSUM_HALO")) opt.halo = std::max(atoi(e), 0);      // cells in pair form below a cell that may leave negative storage
    std::string cache;
    if (xh_plan_cache_on() && !opt.debug) {
        const std::string dir = xh_cache_dir();
        if (!dir.empty()) {
            uint64_t h = plan->topo_hash;
            // (the planner's own version: a planner-only rebuild must not find the partitions of the one before it)
            const int knobs[5] = {opt.simds, opt.piece_cap, opt.halo, opt.pair_imports, flow_rsum_planner_version()};
            for (size_t i = 0; i < sizeof(knobs); ++i) h = (h ^ reinterpret_cast<const unsigned char *>(knobs)[i]) * 1099511628211ull;
            for (const char *b = __DATE__ " " __TIME__; *b; ++b) h = (h ^ (unsigned char)*b) * 1099511628211ull;
            for (const unsigned char *set : {foldable, capable}) {
                h = (h ^ (set ? 1u : 0u)) * 1099511628211ull;
                if (set)
                    for (int64_t c = 0; c < plan->ncell; ++c) h = (h ^ set[c]) * 1099511628211ull;
            }
            char name[96];
            snprintf(name, sizeof(name), "/rsum%s%s_%016llx_%lld.tables", foldable ? "f" : "", capable ? "s" : "", (unsigned long long)h,
                     (long long)plan->ncell);
            cache = dir + name;
        }
    }
    if (!cache.empty() && flow_tables_load(cache.c_str(), t) && t.rsum && t.n_units > 0 &&
        (int64_t)t.cell_of_slot.size() == (int64_t)t.n_units * 64 && (foldable || t.n_folded == 0) && (capable || t.n_special < 0)) {
        handled.assign((size_t)plan->ncell, 0);
        bool ok = true;
        auto take = [&](int c) {
            if (c >= plan->ncell) ok = false;
            else if (c >= 0) handled[c] = 1;
        };
        for (int c : t.cell_of_slot) take(c);
        for (int c : t.fold_of_slot) take(c);
        if (ok && flow_tables_check_rsum((int)plan->ncell, plan->h_indptr.data(), plan->h_indices.data(), plan->h_sign.data(),
                                         handled, t).empty())
            return true;
        t = FlowTables();
    }
    if (flow_tables_build_rsum((int)plan->ncell, plan->h_indptr.data(), plan->h_indices.data(), plan->h_sign.data(),
                               plan->h_comp.data(), plan->h_ncomp, opt, handled, t, err) != 0 || t.n_units == 0)
        return false;
    if (!cache.empty()) {
        const std::string dir = cache.substr(0, cache.rfind('/'));
        for (size_t i = 1; i <= dir.size(); ++i)      // mkdir -p
            if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);
        (void)flow_tables_save(t, cache.c_str());
    }
    return true;
}

// The reassociated plan (host planner + upload).  XH_OK with plan->flow_rsum == nullptr and rsum_failed set when the planner
// has nothing for this grid; the call then takes the bit-exact path.
static int rsum_plan_build(xh_ctx *ctx, xh_route_plan *plan) {
    if (plan->flow_rsum || plan->rsum_failed) return XH_OK;
    plan->rsum_failed = true;
    if (!plan->flow || plan->h_indptr.empty()) return XH_OK;
    std::vector<char> handled;
    FlowTables t;
    if (!rsum_tables_get(ctx, plan, nullptr, nullptr, t, handled)) return XH_OK;
    if (t.n_cells != plan->flow->n_cells) return XH_OK;      // must route exactly the cells the bit-exact plan routes
    if (xh_flow_check()) {
        const std::string bad = flow_tables_check_rsum((int)plan->ncell, plan->h_indptr.data(), plan->h_indices.data(),
                                                       plan->h_sign.data(), handled, t);
        if (!bad.empty()) return xh_fail(ctx, XH_ERR_ARG, "reassociated flow plan check: %s", bad.c_str());
    }
    const int rc = flow_plan_upload(ctx, t, &plan->flow_rsum);
    if (rc) return rc;
    plan->rsum_failed = plan->flow_rsum == nullptr;
    return XH_OK;
}

extern "C" int xh_route_plan_create(xh_ctx *ctx, int64_t ncell, const int64_t *h_indptr, const int32_t *h_indices,
                                    const int8_t *h_sign, xh_route_plan **out) {
    if (!ctx || !out) return XH_ERR_ARG;
    *out = nullptr;
    XH_REQUIRE(ctx, h_indptr && ncell >= 0 && ncell < ((int64_t)1 << 31), "xh_route_plan_create: bad argument");
    const int64_t nnz = h_indptr[ncell];
    XH_REQUIRE(ctx, nnz == 0 || (h_indices && h_sign), "xh_route_plan_create: NULL indices");
    XH_REQUIRE(ctx, nnz < ((int64_t)1 << 31), "xh_route_plan_create: too many entries");
    const int n = (int)ncell;

    XH_HIP(ctx, hipSetDevice(ctx->device));                      // may be called from a host thread of its own (run_model())
    const bool timing = getenv("XH_PLAN_TIMING") != nullptr;      // stderr: where the host time of a plan goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t = now();
        fprintf(stderr, "[libxanthos_hip] plan: %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // ---- networks = weakly connected components of the entries
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    std::vector<int> deg(n, 0);
    for (int i = 0; i < n; ++i) {
        XH_REQUIRE(ctx, h_indptr[i + 1] >= h_indptr[i], "xh_route_plan_create: indptr not monotone");
        deg[i] = (int)(h_indptr[i + 1] - h_indptr[i]);
        for (int64_t j = h_indptr[i]; j < h_indptr[i + 1]; ++j) {
            const int c = h_indices[j];
            XH_REQUIRE(ctx, c >= 0 && c < n, "xh_route_plan_create: column %d out of range", c);
            XH_REQUIRE(ctx, h_sign[j] == 1 || h_sign[j] == -1, "xh_route_plan_create: UM entries must be +1 / -1");
            const int ra = find_root(parent, i), rb = find_root(parent, c);
            if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
        }
    }
    std::vector<int> comp(n), comp_size;
    {
        std::vector<int> id(n, -1);
        for (int i = 0; i < n; ++i) {
            const int r = find_root(parent, i);
            if (id[r] < 0) {
                id[r] = (int)comp_size.size();
                comp_size.push_back(0);
            }
            comp[i] = id[r];
            comp_size[comp[i]]++;
        }
    }
    const int ncomp = (int)comp_size.size();
    std::vector<int> comp_maxdeg(ncomp, 0);
    for (int i = 0; i < n; ++i) comp_maxdeg[comp[i]] = std::max(comp_maxdeg[comp[i]], deg[i]);
    std::vector<std::vector<int>> comp_cells(ncomp);
    for (int c = 0; c < ncomp; ++c) comp_cells[c].reserve(comp_size[c]);
    for (int i = 0; i < n; ++i) comp_cells[comp[i]].push_back(i);

    xh_route_plan *plan = new xh_route_plan();
    plan->ctx = ctx;
    plan->ncell = ncell;
    plan->n_networks = ncomp;
    plan->largest_network = ncomp ? *std::max_element(comp_size.begin(), comp_size.end()) : 0;

    lap("validation + components");
    // ---- tree-shaped networks also get a dataflow layout (xh_mrtm_flow.hip)
    std::vector<char> flow_cell;
    {
        const int frc = flow_plan_build(ctx, n, h_indptr, h_indices, h_sign, comp, ncomp, flow_cell, &plan->flow);
        if (frc) {
            route_plan_free(plan, false);
            return frc;
        }
        if (flow_cell.empty()) flow_cell.assign(n, 0);
        if (plan->flow) {
            {
                uint64_t h = 1469598103934665603ull;      // FNV-1a over the CSR structure
                auto mix = [&](const void *p, size_t nbytes) {
                    const unsigned char *b = static_cast<const unsigned char *>(p);
                    for (size_t i = 0; i < nbytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
                };
                mix(h_indptr, sizeof(int64_t) * (size_t)(n + 1));
                if (nnz) mix(h_indices, sizeof(int32_t) * (size_t)nnz);
                if (nnz) mix(h_sign, (size_t)nnz);
                plan->topo_hash = h;
            }
            plan->h_indptr.assign(h_indptr, h_indptr + n + 1);
            plan->h_indices.assign(h_indices, h_indices + nnz);
            plan->h_sign.assign(h_sign, h_sign + nnz);
            plan->h_comp = comp;
            plan->h_ncomp = ncomp;
            if (reassoc_wanted(0)) {       // the form calls without a flag will ask for: its partition is made with the plan
                const int rrc = rsum_plan_build(ctx, plan);
                if (rrc) {
                    route_plan_free(plan, false);
                    return rrc;
                }
            }
        }
    }
    std::vector<char> comp_flow(ncomp, 0);
    for (int i = 0; i < n; ++i)
        if (flow_cell[i]) comp_flow[comp[i]] = 1;

    lap("dataflow plan");
    // ---- units: big networks alone, small ones first-fit-decreasing into bins of BIN_CELLS.  Two passes keep
    //      every unit either wholly routed by the dataflow kernel or wholly not.
    std::vector<int> order(ncomp);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return comp_size[x] > comp_size[y]; });
    std::vector<std::vector<int>> units;          // cells of each unit
    std::vector<char> unit_is_flow;
    std::vector<int> fb_cells, fb_rest_cells;
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int> bin_free;                // remaining capacity of open bins
        std::vector<int> bin_unit;
        size_t first_open = 0;
        for (int ci : order) {
            if ((comp_flow[ci] != 0) != (pass == 0)) continue;
            const int sz = comp_size[ci];
            if (sz > UNIT_MAX_CELLS || comp_maxdeg[ci] > W_MAX) {
                fb_cells.insert(fb_cells.end(), comp_cells[ci].begin(), comp_cells[ci].end());
                if (pass == 1) fb_rest_cells.insert(fb_rest_cells.end(), comp_cells[ci].begin(), comp_cells[ci].end());
                continue;
            }
            if (sz > BIN_CELLS / 2) {
                units.push_back(comp_cells[ci]);
                unit_is_flow.push_back(pass == 0);
                continue;
            }
            bool placed = false;
            for (size_t b = first_open; b < bin_free.size(); ++b) {
                if (bin_free[b] >= sz) {
                    std::vector<int> &u = units[bin_unit[b]];
                    u.insert(u.end(), comp_cells[ci].begin(), comp_cells[ci].end());
                    bin_free[b] -= sz;
                    placed = true;
                    break;
                }
            }
            if (!placed) {
                bin_unit.push_back((int)units.size());
                bin_free.push_back(BIN_CELLS - sz);
                units.push_back(comp_cells[ci]);
                unit_is_flow.push_back(pass == 0);
            }
            while (first_open < bin_free.size() && bin_free[first_open] == 0) ++first_open;
        }
    }
    std::sort(fb_cells.begin(), fb_cells.end());
    std::sort(fb_rest_cells.begin(), fb_rest_cells.end());

    // ---- slot layout
    std::vector<int> cell_of_slot;
    std::vector<int> slot_of_cell(n, -1), unit_slot0, unit_np;
    for (size_t ui = 0; ui < units.size(); ++ui) {
        auto &u = units[ui];
        // cells with many terms first, so that only the first wave stripes run long gather loops
        std::stable_sort(u.begin(), u.end(), [&](int x, int y) { return deg[x] > deg[y]; });
        int cls = -1;
        for (int k = 0; k < N_CLASS; ++k)
            if (CLASSES[k].nt * CLASSES[k].kc >= (int)u.size()) {
                cls = k;
                break;
            }
        const int np = CLASSES[cls].nt * CLASSES[cls].kc;
        const int s0 = (int)cell_of_slot.size();
        plan->class_units[cls].push_back(s0);
        if (!unit_is_flow[ui]) {
            plan->rest_units[cls].push_back(s0);
            plan->n_rest_units++;
        }
        unit_slot0.push_back(s0);
        unit_np.push_back(np);
        for (int l = 0; l < np; ++l) {
            const int c = l < (int)u.size() ? u[l] : -1;
            cell_of_slot.push_back(c);
            if (c >= 0) slot_of_cell[c] = s0 + l;
        }
        plan->largest_unit = std::max<int64_t>(plan->largest_unit, (int64_t)u.size());
    }
    plan->n_units = (int64_t)units.size();
    plan->total_slots = (int64_t)cell_of_slot.size();
    const int64_t ts = plan->total_slots;
    std::vector<unsigned> ent((size_t)W_MAX * ts);
    std::vector<unsigned char> cnt(ts, 0);
    for (size_t ui = 0; ui < units.size(); ++ui) {
        const int s0 = unit_slot0[ui], np = unit_np[ui];
        const unsigned zero_off = (unsigned)np * 16u;
        for (int l = 0; l < np; ++l) {
            const int slot = s0 + l, c = cell_of_slot[slot];
            for (int w = 0; w < W_MAX; ++w) ent[(size_t)w * ts + slot] = zero_off;
            if (c < 0) continue;
            int w = 0;
            for (int64_t j = h_indptr[c]; j < h_indptr[c + 1]; ++j, ++w) {
                const int src = slot_of_cell[h_indices[j]] - s0;      // same unit by construction
                ent[(size_t)w * ts + slot] = (unsigned)src * 16u + (h_sign[j] < 0 ? 8u : 0u);
            }
            cnt[slot] = (unsigned char)w;
        }
    }

    lap("workgroup units + slots");
    // ---- fallback CSR (subset) and whole-graph CSR
    auto build_csr = [&](const std::vector<int> &cells, std::vector<int> &ptr, std::vector<int> &col,
                         std::vector<signed char> &sgn, std::vector<int> &ds, bool &single) {
        ptr.assign(cells.size() + 1, 0);
        col.clear();
        sgn.clear();
        for (size_t i = 0; i < cells.size(); ++i) {
            const int c = cells[i];
            for (int64_t j = h_indptr[c]; j < h_indptr[c + 1]; ++j) {
                col.push_back(h_indices[j]);
                sgn.push_back((signed char)h_sign[j]);
            }
            ptr[i + 1] = (int)col.size();
        }
        // downstream cell of each cell = the row in which it appears with +1 (atomic variant needs exactly <= 1,
        // and every row's only negative term on its diagonal)
        std::vector<int> ds_cell(n, -1);
        single = true;
        for (int r = 0; r < n; ++r)
            for (int64_t j = h_indptr[r]; j < h_indptr[r + 1]; ++j) {
                const int c = h_indices[j];
                if (h_sign[j] > 0) {
                    if (ds_cell[c] >= 0) single = false;
                    ds_cell[c] = r;
                } else if (c != r) {
                    single = false;
                }
            }
        for (int r = 0; r < n; ++r) {
            int ndiag = 0;
            for (int64_t j = h_indptr[r]; j < h_indptr[r + 1]; ++j)
                if (h_indices[j] == r && h_sign[j] < 0) ++ndiag;
            if (ndiag != 1) single = false;
        }
        ds.resize(cells.size());
        for (size_t i = 0; i < cells.size(); ++i) ds[i] = ds_cell[cells[i]];
    };
    int rc = XH_OK;
    UploadPool pool;
    {
        std::vector<int> ptr, col, ds;
        std::vector<signed char> sgn;
        build_csr(fb_cells, ptr, col, sgn, ds, plan->fb_single_ds);
        plan->n_fb = (int64_t)fb_cells.size();
        pool.add(plan->d_fb_cells, fb_cells);
        pool.add(plan->d_fb_ptr, ptr);
        pool.add(plan->d_fb_col, col);
        pool.add(plan->d_fb_sgn, sgn);
        pool.add(plan->d_fb_ds, ds);
        build_csr(fb_rest_cells, ptr, col, sgn, ds, plan->fb_rest_single_ds);
        plan->n_fb_rest = (int64_t)fb_rest_cells.size();
        pool.add(plan->d_fbr_cells, fb_rest_cells);
        pool.add(plan->d_fbr_ptr, ptr);
        pool.add(plan->d_fbr_col, col);
        pool.add(plan->d_fbr_sgn, sgn);
        pool.add(plan->d_fbr_ds, ds);
        std::vector<int> all(n);
        std::iota(all.begin(), all.end(), 0);
        build_csr(all, ptr, col, sgn, ds, plan->all_single_ds);
        plan->all_nnz = (int64_t)col.size();
        pool.add(plan->d_all_cells, all);
        pool.add(plan->d_all_ptr, ptr);
        pool.add(plan->d_all_col, col);
        pool.add(plan->d_all_sgn, sgn);
        pool.add(plan->d_all_ds, ds);
    }
    pool.add(plan->d_cell_of_slot, cell_of_slot);
    pool.add(plan->d_ent, ent);
    pool.add(plan->d_cnt, cnt);
    for (int k = 0; k < N_CLASS; ++k) {
        pool.add(plan->d_class_units[k], plan->class_units[k]);
        pool.add(plan->d_rest_units[k], plan->rest_units[k]);
    }
    rc |= pool.commit(ctx, &plan->d_pool);
    // (the streams of the per-class kernels and of the global fallback are made on first use: the dataflow
    // kernels, which route every tree-shaped grid, never need them)
    if (hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess) rc |= XH_ERR_HIP;
    for (int k = 0; k <= N_CLASS; ++k)
        if (hipEventCreateWithFlags(&plan->ev_join[k], hipEventDisableTiming) != hipSuccess) rc |= XH_ERR_HIP;
    if (rc) {
        route_plan_free(plan, false);
        return xh_fail(ctx, XH_ERR_HIP, "xh_route_plan_create: device allocation failed");
    }
    lap("tables + uploads");
    *out = plan;
    return XH_OK;
}

// settle = false: teardown of a plan that xh_route_plan_create could not finish.  Nothing was ever routed on it, so no
// call in flight refers to it, and the context's stream and fault bookkeeping are left alone: run_model() makes the plan
// on a host thread while the main thread uploads forcing on the same context (pipeline.py, plan_async).
static void route_plan_free(xh_route_plan *plan, bool settle) {
    if (!plan) return;
    if (settle) (void)xh_sync(plan->ctx);        // settles (and if needed re-runs) routing calls still in flight
    for (int k = 0; k < N_CLASS; ++k) {
        free_buf(plan->d_class_units[k]);
        free_buf(plan->d_rest_units[k]);
        if (plan->streams[k]) (void)hipStreamDestroy(plan->streams[k]);
    }
    if (plan->fb_stream) (void)hipStreamDestroy(plan->fb_stream);
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    for (int k = 0; k <= N_CLASS; ++k)
        if (plan->ev_join[k]) (void)hipEventDestroy(plan->ev_join[k]);
    DevBuf *bufs[] = {&plan->d_cell_of_slot, &plan->d_ent, &plan->d_cnt, &plan->d_fb_cells, &plan->d_fb_ptr,
                      &plan->d_fb_col, &plan->d_fb_sgn, &plan->d_fb_ds, &plan->d_all_cells, &plan->d_all_ptr,
                      &plan->d_all_col, &plan->d_all_sgn, &plan->d_all_ds, &plan->d_fbr_cells, &plan->d_fbr_ptr,
                      &plan->d_fbr_col, &plan->d_fbr_sgn, &plan->d_fbr_ds};
    for (DevBuf *b : bufs) free_buf(*b);
    if (plan->d_pool) (void)hipFree(plan->d_pool);
    flow_plan_destroy(plan->flow);
    flow_plan_destroy(plan->flow_rsum);
    flow_plan_destroy(plan->flow_rsum_fold);
    delete plan;
}

extern "C" void xh_route_plan_destroy(xh_route_plan *plan) { route_plan_free(plan, true); }

extern "C" int xh_route_plan_info(const xh_route_plan *plan, int64_t info[16]) {
    if (!plan || !info) return XH_ERR_ARG;
    info[0] = plan->n_networks;
    info[1] = plan->largest_network;
    info[2] = plan->n_units;
    info[3] = plan->n_fb;
    info[4] = plan->largest_unit;
    info[5] = plan->total_slots;
    info[6] = plan->all_single_ds ? 1 : 0;
    int64_t fi[5];
    flow_plan_info(plan->last_rsum ? plan->last_rsum_plan : plan->flow, fi);
    info[7] = fi[0];                  // dataflow units
    info[8] = fi[1];                  // stream edges
    info[9] = fi[2];                  // pipeline depth
    info[10] = fi[3];                 // cells routed by the dataflow kernel
    info[11] = fi[4];                 // most imported streams of a unit
    info[12] = (plan->last_rsum && plan->last_rsum_plan) ? plan->last_rsum_plan->skew_lmax
                                                    : ((plan->flow && plan->flow->skew_ok) ? plan->flow->skew_lmax : -1);     // deepest lane lag (sub-steps)
    info[13] = plan->last_tree_kernel;
    info[14] = plan->reroutes;
    info[15] = plan->validated;
    return XH_OK;
}

extern "C" int xh_route_plan_stats(xh_route_plan *plan, int64_t max_words, uint64_t *h_words, int64_t *n_words) {
    if (!plan || !n_words) return XH_ERR_ARG;
    std::vector<unsigned long long> st;
    int rc = flow_stats_fetch(plan->ctx, plan->last_rsum ? plan->last_rsum_plan : plan->flow, st);
    if (rc) return rc;
    *n_words = (int64_t)st.size();
    if (h_words)
        for (int64_t i = 0; i < (int64_t)st.size() && i < max_words; ++i) h_words[i] = st[i];
    return XH_OK;
}

// Which cells can fire (mrtm.py:54: dSdt * dt < -S).  With non-negative inflows dSdt >= -F = -S * tauinv, so a cell whose
// tauinv * dt stays below 1 cannot (rounding: three operations of relative error 2^-53 each against a margin of 2^-20;
// negative inputs are outside the argument and are what the guards of the prepared plans are for).  A NaN ratio counts as "can".
constexpr double CAPABLE_THRESHOLD = 1.0 - 1.0 / 1048576.0;

void xh_route_confirm(const xh_route_record &r) { r.plan->fault_streak = 0; }

extern "C" int xh_route_plan_prepare(xh_ctx *ctx, xh_route_plan *plan, const double *h_flow_dist, const double *h_velocity,
                                     double dt) {
    if (!ctx || !plan) return XH_ERR_ARG;
    XH_REQUIRE(ctx, plan->ctx == ctx, "xh_route_plan_prepare: plan belongs to another context");
    XH_REQUIRE(ctx, h_flow_dist && h_velocity && dt > 0.0, "xh_route_plan_prepare: bad argument");
    XH_HIP(ctx, hipSetDevice(ctx->device));                      // may be called from a host thread of its own (run_model())
    // (nothing to prepare for a process whose calls route in the bit-exact form by default: XH_ROUTE_REASSOC=0)
    if (reassoc_wanted(0)) {
        // the PREPARED reassociated plan (xanthos_hip.h, xh_route_plan_rsum_info: folded leaves, single sums) needs
        // exactly what this call brings: which cells can fire.  XH_FLOW_FOLD=0 / XH_RSUM_SINGLE=0: without the one / the other.
        // A plan may be prepared again: other velocities, lengths or dt that change WHICH cells can fire replace the prepared
        // plan (and lift a guard trip's ban, which was about the old one); the same sets are a cheap no-op.
        static const bool fold_on = xh_env_on("XH_FLOW_FOLD", true), single_on = xh_env_on("XH_RSUM_SINGLE", true);
        if ((!fold_on && !single_on) || !plan->flow || plan->h_indptr.empty()) return XH_OK;
        const size_t n = (size_t)plan->ncell;
        std::vector<unsigned char> foldable(n, 0), capable(n, 0);
        size_t nfold = 0;
        uint64_t key = 1469598103934665603ull;
        for (size_t c = 0; c < n; ++c) {
            const double tauinv = h_velocity[c] / h_flow_dist[c];
            const bool leaf = plan->h_indptr[c + 1] - plan->h_indptr[c] == 1;
            capable[c] = (tauinv * dt <= CAPABLE_THRESHOLD) ? 0 : 1;                                    // (NaN: can fire)
            foldable[c] = (leaf && tauinv >= 0.0 && tauinv * dt <= CAPABLE_THRESHOLD) ? 1 : 0;          // (NaN: not foldable)
            nfold += foldable[c];
            key = (key ^ (unsigned)(capable[c] | (foldable[c] << 1))) * 1099511628211ull;
        }
        {
            unsigned char b[sizeof(double)];
            memcpy(b, &dt, sizeof(dt));
            for (unsigned char x : b) key = (key ^ x) * 1099511628211ull;
            if (key == 0) key = 1;
        }
        if (plan->fold_tried && key == plan->prep_key) return XH_OK;
        if (plan->flow_rsum_fold) {      // prepared for other data: the old plan may still be routing
            XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (plan->last_rsum_plan == plan->flow_rsum_fold) {
                plan->last_rsum_plan = nullptr;
                plan->last_rsum = false;
            }
            flow_plan_destroy(plan->flow_rsum_fold);
            plan->flow_rsum_fold = nullptr;
            plan->first_checked_fold = false;
        }
        plan->fold_tried = true;
        plan->prep_key = key;
        plan->fold_disabled = false;
        const bool fold = fold_on && nfold > 0;
        if (!fold && !single_on) return XH_OK;
        std::vector<char> handled;
        FlowTables t;
        if (!rsum_tables_get(ctx, plan, fold ? foldable.data() : nullptr, single_on ? capable.data() : nullptr, t, handled) ||
            (t.n_folded == 0 && t.n_special < 0))
            return XH_OK;
        if (t.n_cells != plan->flow->n_cells) return XH_OK;
        if (xh_flow_check()) {
            const std::string bad = flow_tables_check_rsum((int)plan->ncell, plan->h_indptr.data(), plan->h_indices.data(),
                                                           plan->h_sign.data(), handled, t);
            if (!bad.empty()) return xh_fail(ctx, XH_ERR_ARG, "reassociated flow plan check (prepared plan): %s", bad.c_str());
        }
        const int rcu = flow_plan_upload(ctx, t, &plan->flow_rsum_fold);
        if (rcu) return rcu;
        plan->fold_dt = dt;
        return XH_OK;
    }
    return XH_OK;
}

void xh_route_backoff(xh_route_plan *plan) {      // once per fault event and plan
    plan->fault_streak = std::min(plan->fault_streak + 1, 6);
    plan->skip_calls = 4 << plan->fault_streak;      // 8, 16, ... 256 calls without the dataflow kernels
}

extern "C" int xh_route_plan_rsum_info(const xh_route_plan *plan, int64_t info[8]) {
    if (!plan || !info) return XH_ERR_ARG;
    const FlowPlan *fp = plan->last_rsum ? plan->last_rsum_plan : nullptr;
    info[0] = fp ? fp->n_units : 0;
    info[1] = fp ? fp->n_folded : 0;
    info[2] = plan->fold_disabled ? 1 : 0;
    info[3] = plan->flow_rsum_fold ? plan->flow_rsum_fold->n_folded : 0;
    info[4] = fp ? fp->n_special : -1;
    info[5] = plan->flow_rsum_fold ? plan->flow_rsum_fold->n_special : -1;
    info[6] = plan->guard_trips;
    info[7] = fp ? fp->n_pair_units : 0;
    return XH_OK;
}

int route_series_impl(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months,
                             const int32_t *h_ndays, double dt, const double *d_flow_dist, const double *d_velocity,
                             const double *d_area, const double *d_runoff, const double *d_S0, double *d_chstorage,
                             double *d_avgchflow, double *d_S_end, double *d_F_end, int32_t flags, bool *used_flow,
                             const FlowFeed *feed) {
    if (!ctx || !plan) return XH_ERR_ARG;
    XH_REQUIRE(ctx, plan->ctx == ctx, "xh_route_series: plan belongs to another context");
    XH_REQUIRE(ctx, h_ndays && d_flow_dist && d_velocity && d_area && d_runoff, "xh_route_series: NULL argument");
    XH_REQUIRE(ctx, nmonths > 0 && spinup_months >= 0 && spinup_months <= nmonths,
               "xh_route_series: need 0 <= spinup_months (%d) <= nmonths (%d)", spinup_months, nmonths);
    XH_REQUIRE(ctx, dt > 0.0, "xh_route_series: dt must be positive");
    if (plan->ncell == 0) return XH_OK;

    // month schedule: spin-up months 0..spinup-1 (outputs discarded: the simulation pass overwrites them,
    // components.py:273-294), then every month
    const int nit = spinup_months + nmonths;
    std::vector<int> sm(nit), snt(nit);
    std::vector<double> ssecs(nit);
    std::vector<unsigned char> swr(nit);
    for (int it = 0; it < nit; ++it) {
        const int m = it < spinup_months ? it : it - spinup_months;
        XH_REQUIRE(ctx, h_ndays[m] > 0, "xh_route_series: ndays[%d] = %d", m, h_ndays[m]);
        sm[it] = m;
        ssecs[it] = (double)((int64_t)h_ndays[m] * 24 * 3600);
        snt[it] = (int)(ssecs[it] / dt);                                        // int(nday*24*3600/dt), mrtm.py:35
        XH_REQUIRE(ctx, snt[it] >= 1, "xh_route_series: dt longer than a month");
        swr[it] = it >= spinup_months ? 1 : 0;
    }
    std::vector<int> sg(nit + 1, 0);                     // first global sub-step of each iteration
    for (int it = 0; it < nit; ++it) {
        XH_REQUIRE(ctx, (int64_t)sg[it] + snt[it] < (int64_t)1 << 30, "xh_route_series: more than 2^30 sub-steps");
        sg[it + 1] = sg[it] + snt[it];
    }
    const size_t sched_bytes = (size_t)nit * (3 * sizeof(int) + sizeof(double) + 1) + 64;
    void *sbuf = nullptr;
    int rc = xh_scratch(ctx, 2, sched_bytes, &sbuf);
    if (rc) return rc;
    double *d_secs = static_cast<double *>(sbuf);
    int *d_m = reinterpret_cast<int *>(d_secs + nit);
    int *d_nt = d_m + nit;
    int *d_g = d_nt + nit;
    unsigned char *d_wr = reinterpret_cast<unsigned char *>(d_g + nit + 1);
    // The schedule on the device, for the kernels that read it there -- every one but k_mrtm_wave, whose launch turns the host
    // arrays into month records: uploaded in front of the first launch that needs it (five small copies that a call routed
    // by k_mrtm_wave alone, i.e. every tree-shaped grid, no longer pays for).
    bool sched_on_device = false;
    auto sched_upload = [&]() -> int {
        if (sched_on_device) return XH_OK;
        sched_on_device = true;
        XH_HIP(ctx, hipMemcpyAsync(d_secs, ssecs.data(), sizeof(double) * nit, hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(ctx, hipMemcpyAsync(d_m, sm.data(), sizeof(int) * nit, hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(ctx, hipMemcpyAsync(d_nt, snt.data(), sizeof(int) * nit, hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(ctx, hipMemcpyAsync(d_g, sg.data(), sizeof(int) * (nit + 1), hipMemcpyHostToDevice, ctx->stream));
        XH_HIP(ctx, hipMemcpyAsync(d_wr, swr.data(), (size_t)nit, hipMemcpyHostToDevice, ctx->stream));
        return XH_OK;
    };
    // Reassociated form (XH_ROUTE_REASSOC, flag or environment): a partition of its own, nothing to learn -- the typed / adaptive
    // machinery below is for the bit-exact kernel only.
    bool use_rsum = reassoc_wanted(flags) && plan->flow &&
                    (flags & (XH_ROUTE_FORCE_FALLBACK | XH_ROUTE_NO_DATAFLOW | XH_ROUTE_NO_SKEW | XH_ROUTE_ATOMIC)) == 0;
    if (use_rsum) {
        if ((rc = rsum_plan_build(ctx, plan)) != XH_OK) return rc;
        use_rsum = plan->flow_rsum != nullptr;
    }
    flags &= ~(XH_ROUTE_REASSOC | XH_ROUTE_EXACT);
    FlowPlan *tree_plan = plan->flow;
    FlowPlan *rsum_plan = plan->flow_rsum;
    if (use_rsum && plan->flow_rsum_fold && !plan->fold_disabled && dt == plan->fold_dt && (flags & XH_ROUTE_NO_PLAIN) == 0)
        rsum_plan = plan->flow_rsum_fold;      // leaves that cannot fire carried by their parents' lanes (guarded in the kernel)
    if (use_rsum) tree_plan = rsum_plan;

    const bool force_fb = (flags & XH_ROUTE_FORCE_FALLBACK) != 0;
    const bool atomic = (flags & XH_ROUTE_ATOMIC) != 0;
    bool use_flow = !force_fb && plan->flow != nullptr && (flags & XH_ROUTE_NO_DATAFLOW) == 0;
    // fed call: every cell must be routed by k_mrtm_wave (the other kernels read the runoff array itself, at once)
    if (feed && (!use_flow || plan->n_rest_units > 0 || plan->n_fb_rest > 0)) return XH_ERR_LIMIT;

    xh_span sp = xh_span_begin(ctx, "mrtm_route");
    // (kernels on the class streams wait for ev_fork: whatever they read must be in the stream before it is recorded)
    if (!(use_flow && plan->n_rest_units == 0 && plan->n_fb_rest == 0) && (rc = sched_upload()) != XH_OK) return rc;
    XH_HIP(ctx, hipEventRecord(plan->ev_fork, ctx->stream));
    int njoin = 0;
    plan->last_tree_kernel = 0;
    plan->last_rsum = false;
    if (use_flow) {     // tree-shaped networks: single-wave dataflow units on the context's own stream
        int ntmax = 0, ntmin = INT_MAX;
        bool nt_even = true;
        for (int v : snt) {
            ntmax = std::max(ntmax, v);
            ntmin = std::min(ntmin, v);
            nt_even = nt_even && (v & 1) == 0;
        }
        const FlowSched fs{nmonths, nit, ntmax, ntmin, sg[nit], d_m, d_nt, d_g, d_secs, d_wr, dt,
                           (flags & XH_ROUTE_TEST_FAULT) != 0, nt_even, sm.data(), snt.data(), sg.data(), ssecs.data(),
                           swr.data()};
        const FlowIO fio{d_flow_dist, d_velocity, d_area, d_runoff, d_S0, d_chstorage, d_avgchflow, d_S_end, d_F_end, feed};
        // time-skewed units first; months shorter than the deepest lane lag (long dt) and grids beyond the kernel's 32-bit row
        // offsets use the lock-step kernel (k_mrtm_flow)
        rc = XH_ERR_LIMIT;
        plan->last_tree_kernel = 2;
        if ((flags & XH_ROUTE_NO_SKEW) == 0) rc = wave_launch(ctx, tree_plan, fs, fio, ctx->stream);
        // (which plan actually ran: a month shorter than the plan's lane lags, the 32-bit row limit or residency send the call
        // from the reassociated plan back to the bit-exact one -- everything below speaks of THAT plan then)
        FlowPlan *ran = tree_plan;
        if (rc == XH_ERR_LIMIT && tree_plan != plan->flow && (flags & XH_ROUTE_NO_SKEW) == 0) {
            ran = plan->flow;
            rc = wave_launch(ctx, plan->flow, fs, fio, ctx->stream);
        }
        plan->last_rsum = rc == XH_OK && use_rsum && ran == rsum_plan && ran->rsum;
        if (plan->last_rsum) plan->last_rsum_plan = rsum_plan;
        if (plan->last_rsum) plan->last_tree_kernel = 4;
        if (feed && rc == XH_ERR_LIMIT) {      // nothing was launched: the caller completes the runoff and calls again
            xh_span_cancel(sp);
            return XH_ERR_LIMIT;
        }
        if (rc == XH_ERR_LIMIT) {
            plan->last_tree_kernel = 1;
            if ((rc = sched_upload()) != XH_OK) return rc;
            rc = flow_launch(ctx, plan->flow, fs, fio, ctx->stream);
        }
        if (rc == XH_ERR_LIMIT) {
            use_flow = false;   // units cannot all be resident on this device: one workgroup per network instead
            plan->last_tree_kernel = 0;
        } else if (rc) {
            return rc;
        }
    }
    const int64_t n_fb = force_fb ? plan->ncell : (use_flow ? plan->n_fb_rest : plan->n_fb);
    const int64_t n_lds_units = use_flow ? plan->n_rest_units : plan->n_units;
    if (((!force_fb && n_lds_units > 0) || n_fb > 0) && !sched_on_device) {
        // only after the dataflow kernels turned a tree-only grid down: nothing of this call is running yet
        rc = sched_upload();
        if (rc) return rc;
        XH_HIP(ctx, hipEventRecord(plan->ev_fork, ctx->stream));
    }
    if (!force_fb && n_lds_units > 0) {
        RouteArgs a;
        a.unit_slot0 = nullptr;
        a.cell_of_slot = static_cast<const int *>(plan->d_cell_of_slot.p);
        a.ent = static_cast<const unsigned *>(plan->d_ent.p);
        a.cnt = static_cast<const unsigned char *>(plan->d_cnt.p);
        a.total_slots = plan->total_slots;
        a.nmonths = nmonths;
        a.nit = nit;
        a.sched_m = d_m;
        a.sched_nt = d_nt;
        a.sched_secs = d_secs;
        a.sched_write = d_wr;
        a.dt = dt;
        a.dtinv = 1.0 / dt;
        a.flow_dist = d_flow_dist;
        a.velocity = d_velocity;
        a.area = d_area;
        a.runoff = d_runoff;
        a.S0 = d_S0;
        a.chs = d_chstorage;
        a.avg = d_avgchflow;
        a.S_end = d_S_end;
        a.F_end = d_F_end;
        // largest classes first so the long-running workgroups start first
        for (int cls = N_CLASS - 1; cls >= 0; --cls) {
            if ((use_flow ? plan->rest_units[cls] : plan->class_units[cls]).empty()) continue;
            hipStream_t st = nullptr;
            if (!plan->streams[cls]) XH_HIP(ctx, hipStreamCreateWithFlags(&plan->streams[cls], hipStreamNonBlocking));
            st = plan->streams[cls];
            XH_HIP(ctx, hipStreamWaitEvent(st, plan->ev_fork, 0));
            switch (cls) {
                case 0: launch_units<1, 64>(plan, cls, a, st, use_flow); break;
                case 1: launch_units<2, 64>(plan, cls, a, st, use_flow); break;
                case 2: launch_units<4, 64>(plan, cls, a, st, use_flow); break;
                case 3: launch_units<2, 256>(plan, cls, a, st, use_flow); break;
                case 4: launch_units<4, 256>(plan, cls, a, st, use_flow); break;
                case 5: launch_units<2, 1024>(plan, cls, a, st, use_flow); break;
                case 6: launch_units<4, 768>(plan, cls, a, st, use_flow); break;
                case 7: break;
            }
            XH_HIP(ctx, hipGetLastError());
            XH_HIP(ctx, hipEventRecord(plan->ev_join[njoin], st));
            XH_HIP(ctx, hipStreamWaitEvent(ctx->stream, plan->ev_join[njoin], 0));
            ++njoin;
        }
    }
    if (n_fb > 0) {
        const bool single = force_fb ? plan->all_single_ds : (use_flow ? plan->fb_rest_single_ds : plan->fb_single_ds);
        const DevBuf &b_cells = force_fb ? plan->d_all_cells : (use_flow ? plan->d_fbr_cells : plan->d_fb_cells);
        const DevBuf &b_ptr = force_fb ? plan->d_all_ptr : (use_flow ? plan->d_fbr_ptr : plan->d_fb_ptr);
        const DevBuf &b_col = force_fb ? plan->d_all_col : (use_flow ? plan->d_fbr_col : plan->d_fb_col);
        const DevBuf &b_sgn = force_fb ? plan->d_all_sgn : (use_flow ? plan->d_fbr_sgn : plan->d_fb_sgn);
        const DevBuf &b_ds = force_fb ? plan->d_all_ds : (use_flow ? plan->d_fbr_ds : plan->d_fb_ds);
        if (atomic && !single) {
            xh_span_end(sp);
            return xh_fail(ctx, XH_ERR_ARG, "xh_route_series: XH_ROUTE_ATOMIC needs one downstream cell per cell");
        }
        // work arrays in cell order: S, F, F2, favg, erl, inflow (6 doubles) + sx flag
        void *wbuf = nullptr;
        const size_t nc = (size_t)plan->ncell;
        rc = xh_scratch(ctx, 3, nc * (6 * sizeof(double) + 1) + 64, &wbuf);
        if (rc) return rc;
        FbArgs f;
        f.n = (int)n_fb;
        f.cells = static_cast<const int *>(b_cells.p);
        f.ptr = static_cast<const int *>(b_ptr.p);
        f.col = static_cast<const int *>(b_col.p);
        f.sgn = static_cast<const signed char *>(b_sgn.p);
        f.ds = static_cast<const int *>(b_ds.p);
        f.nmonths = nmonths;
        f.dt = dt;
        f.dtinv = 1.0 / dt;
        f.flow_dist = d_flow_dist;
        f.velocity = d_velocity;
        f.area = d_area;
        f.runoff = d_runoff;
        f.S0 = d_S0;
        f.S = static_cast<double *>(wbuf);
        f.F = f.S + nc;
        f.F2 = f.F + nc;
        f.favg = f.F2 + nc;
        f.erl = f.favg + nc;
        f.inflow = atomic ? f.erl + nc : nullptr;
        f.sx = reinterpret_cast<unsigned char *>(f.erl + 2 * nc);
        f.chs = d_chstorage;
        f.avg = d_avgchflow;
        f.S_end = d_S_end;
        f.F_end = d_F_end;
        hipStream_t st = nullptr;
        if (!plan->fb_stream) XH_HIP(ctx, hipStreamCreateWithFlags(&plan->fb_stream, hipStreamNonBlocking));
        st = plan->fb_stream;
        XH_HIP(ctx, hipStreamWaitEvent(st, plan->ev_fork, 0));
        const dim3 grid((unsigned)((n_fb + 255) / 256)), block(256);
        hipLaunchKernelGGL(k_fb_init, grid, block, 0, st, f);
        for (int it = 0; it < nit; ++it) {
            hipLaunchKernelGGL(k_fb_month_begin, grid, block, 0, st, f, sm[it], ssecs[it]);
            for (int t = 0; t < snt[it]; ++t) {
                if (atomic) {
                    hipLaunchKernelGGL(k_fb_scatter, grid, block, 0, st, f, 0);
                    hipLaunchKernelGGL(k_fb_phase_a<true>, grid, block, 0, st, f);
                    hipLaunchKernelGGL(k_fb_scatter, grid, block, 0, st, f, 1);
                    hipLaunchKernelGGL(k_fb_phase_b<true>, grid, block, 0, st, f);
                } else {
                    hipLaunchKernelGGL(k_fb_phase_a<false>, grid, block, 0, st, f);
                    hipLaunchKernelGGL(k_fb_phase_b<false>, grid, block, 0, st, f);
                }
            }
            hipLaunchKernelGGL(k_fb_month_end, grid, block, 0, st, f, sm[it], snt[it], (int)swr[it]);
        }
        hipLaunchKernelGGL(k_fb_finish, grid, block, 0, st, f);
        XH_HIP(ctx, hipGetLastError());
        XH_HIP(ctx, hipEventRecord(plan->ev_join[N_CLASS], st));
        XH_HIP(ctx, hipStreamWaitEvent(ctx->stream, plan->ev_join[N_CLASS], 0));
    }
    xh_span_end(sp);
    *used_flow = use_flow;
    return XH_OK;
}
