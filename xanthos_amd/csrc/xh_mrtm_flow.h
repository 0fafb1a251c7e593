// Internal interface between xh_mrtm.hip (plan + API) and xh_mrtm_flow.hip (tree-partitioned dataflow routing).
#pragma once
#include <vector>

#include "xh_common.h"

struct FlowPlan;   // opaque: defined in xh_mrtm_flow.hip

struct FlowSched {
    int nmonths, nit, ntmax;
    const int *d_m, *d_nt;            // [nit] month index / sub-steps of each iteration
    const double *d_secs;             // [nit]
    const unsigned char *d_wr;        // [nit] 1 = simulation pass (store outputs)
    double dt;
};

struct FlowIO {
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *chs, *avg, *S_end, *F_end;
};

// Partition every tree-shaped river network (each cell drains to at most one cell, no cycle, standard UP - I rows)
// into single-wave units linked by one-way monthly streams.  handled[c] = 1 for the cells these units route.
int flow_plan_build(xh_ctx *ctx, int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                    const std::vector<int> &comp, int ncomp, std::vector<char> &handled, FlowPlan **out);
void flow_plan_destroy(FlowPlan *fp);
// info: [0] units, [1] stream edges, [2] pipeline depth (levels), [3] cells, [4] max imports of a unit
void flow_plan_info(const FlowPlan *fp, int64_t info[5]);
// Enqueue the persistent dataflow kernel on `st`. Returns XH_ERR_LIMIT if the units cannot all be resident.
int flow_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st);
// Per-unit cycle accounting of the last launch (only when XH_FLOW_STATS=1): 4 words per unit
// {shader cycles in sub-step loops, shader cycles total, 100 MHz ticks total, shape bits}.
int flow_stats_fetch(xh_ctx *ctx, FlowPlan *fp, std::vector<unsigned long long> &out);
