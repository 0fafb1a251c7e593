// Internal interface between xh_mrtm.hip (plan + API), xh_mrtm_flow.hip (tree partition + lock-step dataflow kernel) and the
// time-skewed dataflow kernels on the same partition (xh_mrtm_wave.hip, xh_mrtm_rsum.hip; launch: xh_mrtm_wave_launch.hip).
#pragma once
#include <string>
#include <vector>

#include "xh_common.h"
#include "xh_flow_plan.h"

struct FlowBuf {
    void *p = nullptr;
};

// Partition of the tree-shaped river networks into single-wave units (64 lanes = 64 cells) linked by one-way streams.
struct FlowPlan {
    int n_units = 0, n_edges = 0, depth = 0, n_cells = 0, max_imports = 0, max_exports = 0;
    bool rsum = false;                   // reassociated form (xh_flow_rsum.cpp tables, k_mrtm_rsum): wave_launch only
    int max_cell = -1;                   // largest grid index of a routed cell (the time-skewed kernel's 32-bit row offsets)
    FlowBuf d_cell_of_slot, d_ent, d_export_edge, d_ghost_edge, d_edge_cons_unit, d_unit_terms;
    // time-skewed layout; skew_ok = every row has <= 4 terms either side of its diagonal
    bool skew_ok = false;
    int skew_lmax = 0;                   // largest lane lag of any unit (sub-steps)
    int skew_span = 1;                   // most pipeline levels a stream jumps over (consumer depth - producer depth)
    FlowBuf d_lag, d_ghost_lag, d_ent2, d_eprev, d_unit_p, d_unit_lmax, d_unit_glmax, d_unit_order;
    // per-call exchange buffers (grow-only)
    void *d_x = nullptr;
    size_t x_bytes = 0;
    unsigned long long *d_stats = nullptr;
    void *d_skew_args = nullptr;         // argument block of the time-skewed kernel (rewritten, stream-ordered, by every launch)
    std::vector<char> h_rec, h_fin;      // host copies of the month records of the last launch (sources of asynchronous copies)
    uint64_t rec_key = 0;                // hash of the schedule / runoff layout the records on the device were made for (0: none)
    FlowBuf d_fold_cell;                 // reassociated form with folded leaves: [units*64] the leaf a lane carries, or -1 (NULL: none)
    int n_folded = 0;
    int n_special = -1;      // reassociated form: -1 = pairs of sums, >= 0 = single-sum plan with that many cells in pair units
    int n_pair_units = 0;    // single-sum plan: its pair units (the tail of the claim list; the kernel gives them CUs of their own)
    FlowBuf d_lane_flags;    // single-sum plans: [units*64] bit 0 the cell may fire, bit 1 an exit lane (xh_flow_rsum.cpp)
};

struct FlowSched {
    int nmonths, nit, ntmax, ntmin, total;
    const int *d_m, *d_nt;            // [nit] month index / sub-steps of each iteration
    const int *d_g;                   // [nit + 1] first global sub-step of each iteration; d_g[nit] = total
    const double *d_secs;             // [nit]
    const unsigned char *d_wr;        // [nit] 1 = simulation pass (store outputs)
    double dt;
    bool test_fault;                  // XH_ROUTE_TEST_FAULT: unit 0 raises the fault word and stops (tests of the re-route)
    bool nt_even;                     // every month has an even number of sub-steps (dt = 3 h: 8 per day)
    const int *h_m, *h_nt, *h_g;      // the same schedule on the host ([nit], h_g [nit + 1])
    const double *h_secs;
    const unsigned char *h_wr;
};

// Routing that starts before the whole runoff series exists (xh_run_fused, mode 1; k_mrtm_wave only).  The runoff is read
// from a STAGED copy the ABCD kernels write beside the [ncell, nmonths] array: [line v][cell][16 months], one 128-byte line
// per (16 months, cell), every line written whole by ONE kernel and read only after `months_ready` says so -- a line is
// never in anybody's cache before it is final, so the hand-over needs no invalidate and no coherent loads, only the flag.
struct FlowFeed {
    const double *q_staged = nullptr;     // [ceil(nmonths / 16)][ncell][16]
    int64_t ncell = 0;                    // cells of the staged array (its line stride is ncell * 128 bytes)
    const unsigned *months_ready = nullptr;      // device word: months [0, *months_ready) of every cell are final
    unsigned ready_at_launch = 0;         // its value when the kernel is launched
    unsigned *place_epoch = nullptr;      // device word the kernel sets to `epoch` once every workgroup is resident and placed
    unsigned epoch = 0;
};

struct FlowIO {
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *chs, *avg, *S_end, *F_end;
    const FlowFeed *feed = nullptr;   // wave_launch only: runoff arrives while the kernel runs (see FlowFeed)
};

// Partition every tree-shaped river network (each cell drains to at most one cell, no cycle, standard UP - I rows)
// into single-wave units linked by one-way streams (every unit in pair form).  handled[c] = 1 for the cells these units route.
int flow_plan_build(xh_ctx *ctx, int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                    const std::vector<int> &comp, int ncomp, std::vector<char> &handled, FlowPlan **out);
// The two halves of flow_plan_build.  flow_tables_host is plain host work (no HIP call, no context).  flow_plan_upload
// allocates and fills the device tables.
FlowPlanOptions flow_plan_options(const xh_ctx *ctx);      // device size + the XH_FLOW_* switches of the environment
int flow_tables_host(FlowPlanOptions opt, int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                     const std::vector<int> &comp, int ncomp, std::vector<char> &handled, FlowTables &t, std::string &err);
int flow_plan_upload(xh_ctx *ctx, const FlowTables &t, FlowPlan **out);
void flow_plan_destroy(FlowPlan *fp);
// info: [0] units, [1] stream edges, [2] pipeline depth (levels), [3] cells, [4] max imports of a unit
void flow_plan_info(const FlowPlan *fp, int64_t info[5]);
// Enqueue the persistent dataflow kernel on `st`. Returns XH_ERR_LIMIT if the units cannot all be resident.
int flow_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st);
// Same contract for the time-skewed kernels (k_mrtm_wave / k_mrtm_rsum, by the plan's kind); XH_ERR_LIMIT also when the
// schedule does not suit them (months shorter than the deepest lane lag, rows wider than 4 + 1 + 4, rows beyond 32-bit
// offsets) -- the caller then uses flow_launch.
int wave_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st);
const void *wave_rsum_kernel();      // k_mrtm_rsum (xh_mrtm_rsum.hip): the kernel wave_launch starts for a reassociated plan
const void *wave_exact_kernel();     // k_mrtm_wave (xh_mrtm_wave.hip)
// Per-unit cycle accounting of the last launch (only when XH_FLOW_STATS=1): 6 words per unit
// {shader cycles in sub-step loops, shader cycles total, 100 MHz ticks total, shape bits, data-wait, ring-wait cycles}.
int flow_stats_fetch(xh_ctx *ctx, FlowPlan *fp, std::vector<unsigned long long> &out);
