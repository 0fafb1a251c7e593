// The routing plan (xh_route_plan) as the two halves of the routing code see it: xh_mrtm.hip (kernels for general graphs, plan
// create / prepare / info, route_series_impl: which kernel routes a call) and xh_mrtm_check.hip (the call layer: first-call
// and periodic cross-checks, XH_ROUTE_VALIDATE, the record of unconfirmed calls and their re-runs after a device fault).
#pragma once
#include <string>
#include <vector>

#include "xh_common.h"
#include "xh_mrtm_flow.h"

constexpr int XH_ROUTE_N_CLASS = 8;      // workgroup shapes of the workgroup-per-network kernel (xh_mrtm.hip: CLASSES)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool pooled = false;        // p points into an UploadPool's allocation (freed with the pool, not on its own)
};

struct xh_route_plan {
    xh_ctx *ctx = nullptr;
    int64_t ncell = 0, n_networks = 0, largest_network = 0, n_units = 0, largest_unit = 0, total_slots = 0;
    // LDS units, grouped by class
    std::vector<int> class_units[XH_ROUTE_N_CLASS];       // slot0 of the units of each class (every network)
    DevBuf d_class_units[XH_ROUTE_N_CLASS];
    std::vector<int> rest_units[XH_ROUTE_N_CLASS];        // only the units of networks the dataflow kernel does not route
    DevBuf d_rest_units[XH_ROUTE_N_CLASS];
    FlowPlan *flow = nullptr;                    // tree-shaped networks as single-wave dataflow units of the BIT-EXACT kernels
    // Reassociated form (XH_ROUTE_REASSOC; xh_flow_rsum.cpp, k_mrtm_rsum): a partition of its own over the same cells, made
    // with the plan when the environment asks for the form, else on the first call that does.  Needs nothing of a call's data.
    FlowPlan *flow_rsum = nullptr;
    // ... and the same with FOLDED LEAVES (xh_flow_rsum.cpp, FlowPlanOptions::foldable): which leaves cannot fire depends on
    // velocity, flow distance and dt, so this one is made by xh_route_plan_prepare from the host copies (XH_FLOW_FOLD=1); the
    // kernel guards the assumption and a trip routes the call again on flow_rsum and switches the folded plan off
    FlowPlan *flow_rsum_fold = nullptr;
    double fold_dt = 0.0;
    bool fold_disabled = false;
    bool fold_tried = false;      // prepare() has asked the planner (it may have had nothing to fold)
    uint64_t prep_key = 0;        // ... for these sets of cells that can fire / leaves that cannot, and this dt
    bool first_checked_fold = false;
    FlowPlan *last_rsum_plan = nullptr;          // the plan the last reassociated call ran on
    bool rsum_failed = false;                    // the planner turned the grid down once: not tried again
    bool last_rsum = false;                      // the last call was routed by k_mrtm_rsum
    bool first_checked_rsum = false;
    std::vector<int64_t> h_indptr;
    std::vector<int32_t> h_indices;
    std::vector<int8_t> h_sign;
    std::vector<int> h_comp;
    int h_ncomp = 0;
    int guard_trips = 0;                         // calls routed again on the plan of pairs after a guard of the prepared plan tripped
    int last_tree_kernel = 0;                    // last xh_route_series: 0 none, 1 monthly streams, 2 time-skewed
    int64_t reroutes = 0;                        // calls re-run with one workgroup per network after a device fault
    // After a fault the dataflow kernels are skipped for the next `skip_calls` calls of this plan (the device is shared:
    // every further attempt would first sit out a bounded wait), doubling with every fault in a row up to 256 calls; a
    // dataflow call confirmed fault-free resets the streak (xh_route_confirm).
    int fault_streak = 0, skip_calls = 0;
    int64_t validated = 0;                       // calls cross-checked against the workgroup-per-network kernel (XH_ROUTE_VALIDATE)
    // The dataflow kernels' streams rest on an ordering assumption outside the HIP memory model (xh_mrtm_wave.hip, check()).
    // So that no product run is unverified on a new box, the FIRST dataflow call of a plan is cross-checked like
    // XH_ROUTE_VALIDATE unless a marker file says this library build already passed on this device with this topology
    // (route_first_check_*; XH_ROUTE_VALIDATE_FIRST=0 switches it off).
    bool first_checked = false;
    // ... and every XH_ROUTE_VALIDATE_EVERY-th dataflow call of a long-lived plan is cross-checked again (default 1,000; 0 =
    // never): one clean pass says little about call 10^4 of a server that routes scenarios all day (~0.25 s each time)
    int64_t dataflow_calls = 0;
    bool validate_due = false;
    uint64_t topo_hash = 0;
    int64_t n_rest_units = 0, n_fb_rest = 0;
    bool fb_rest_single_ds = true;
    DevBuf d_fbr_cells, d_fbr_ptr, d_fbr_col, d_fbr_sgn, d_fbr_ds;
    DevBuf d_cell_of_slot, d_ent, d_cnt;
    // fallback
    int64_t n_fb = 0;
    bool fb_single_ds = true;
    DevBuf d_fb_cells, d_fb_ptr, d_fb_col, d_fb_sgn, d_fb_ds;
    // whole-graph copies so XH_ROUTE_FORCE_FALLBACK can route everything
    DevBuf d_all_cells, d_all_ptr, d_all_col, d_all_sgn, d_all_ds;
    bool all_single_ds = true;
    int64_t all_nnz = 0;
    hipStream_t streams[XH_ROUTE_N_CLASS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[XH_ROUTE_N_CLASS + 1] = {};
    hipStream_t fb_stream = nullptr;
    void *d_pool = nullptr;     // the allocation behind the tables uploaded at create (UploadPool)
};

// xh_mrtm.hip
int route_series_impl(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months, const int32_t *h_ndays, double dt,
                      const double *d_flow_dist, const double *d_velocity, const double *d_area, const double *d_runoff,
                      const double *d_S0, double *d_chstorage, double *d_avgchflow, double *d_S_end, double *d_F_end,
                      int32_t flags, bool *used_flow, const FlowFeed *feed = nullptr);
bool reassoc_wanted(int flags);          // which form routes the tree networks of a call with these flags
