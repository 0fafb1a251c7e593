// Host-side planner of the dataflow routing kernels: cuts the tree-shaped river networks into connected pieces, packs
// them into single-wave units and emits the tables the kernels read.  Plain C++ on host vectors -- no HIP, no device
// memory -- so that the same translation unit builds into libxanthos_hip.so and, with -fsanitize=address,undefined,
// into the host-only fuzzer of tests/plan_fuzz (round 2's 680-line flow_plan_build only ever ran behind GPU tests).
//
// Reference semantics the tables must preserve: mrtm.py:50-51 -- row i of UM.dot(F) is accumulated in stored (ascending
// column) order, which flow_tables_check re-derives from the tables term by term.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct FlowPlanOptions {
    int simds = 0;               // SIMDs of the device (units beyond them share a SIMD); 0 = unknown
    int piece_cap = 0;           // cells per piece; 0 = try the built-in capacities and keep the best partition
    bool chain = true;           // sum long front sides on the way along chains of lanes
    bool cut_rule = true;        // chain-aware choice of the children that become streams
    int tlimit = 5;              // pair reads per sub-step a unit may reach by taking in another piece
    bool balance_lds = false;    // order the claim list's tail by LDS load (XH_WAVE_BALANCE=1: every CU one unit of each quarter)
    int lane_trials = 0;         // > 0: move the cells of a unit to lanes on which its gather meets fewer LDS bank conflicts (swaps tried per unit)
    bool debug = false;          // partition statistics on stderr
    // Reassociated planner only (xh_flow_rsum.cpp): foldable[c] != 0 for LEAF cells (no upstream neighbour) that cannot fire
    // (velocity * dt / length below 1) -- such a cell's outflow is a one-fma recurrence that the lane of its downstream cell can
    // carry in registers, so it needs no lane, no LDS slot and no level of lag of its own.  nullptr: nothing is folded.
    const unsigned char *foldable = nullptr;
    // Reassociated planner: capable[c] != 0 for the cells that can fire (velocity * dt / length close to or above 1).  Given,
    // it makes the SINGLE-SUM partition (xh_flow_rsum.cpp): lanes pass one running sum instead of a pair; the cells that may
    // fire AND have an upstream neighbour that may -- capable cells and `halo` cells downstream of every capable cell with a
    // capable neighbour -- sit in pair units of their own.  nullptr: the plan of pairs.  (The bit-exact planner ignores it:
    // all of its units gather pairs.  Its "typed" partitions -- 8-byte units for the cells without such a neighbour, rounds 3
    // to 5 -- went when the single-sum form replaced them: docs/HISTORY.md.)
    const unsigned char *capable = nullptr;
    int halo = 8;
    int pair_imports = 8;        // ... and take on at most this many imported streams per unit (one import round of the kernel: 12.4-12.6
                                 // against 12.7 ms with 16 -- round 6, same box)
};

struct FlowTables {
    int n_units = 0, n_edges = 0, depth = 0, n_cells = 0, max_imports = 0, max_exports = 0;
    bool skew_ok = false;        // every row has <= 4 terms either side of its diagonal
    int skew_lmax = 0;           // largest lane lag of any unit (sub-steps)
    int skew_span = 1;           // most pipeline levels a stream jumps over
    bool rsum = false;           // reassociated form (xh_flow_rsum.cpp): ent2[0] = the cell's inflow entry, eprev = its chain entry, unit_p = 0x400 | reads
    std::vector<int> cell_of_slot, export_edge, ghost_edge, edge_cons_unit, unit_terms;
    std::vector<unsigned> ent;                                   // lock-step kernel: [9][units*64]
    std::vector<int> lag, ghost_lag, unit_p, unit_lmax, unit_glmax, unit_order;
    std::vector<unsigned> ent2, eprev;                           // time-skewed kernels: [8][units*64], [units*64]
    // kept for checks and diagnostics
    std::vector<int> edge_prod_cell, edge_cons_cell, unit_depth, piece_of_cell, unit_of_cell, height_of_cell, ds;
    std::vector<unsigned char> lane_flags;                       // single-sum plans: [units*64] bit 0 the cell may fire, bit 1 an exit lane (all 0 otherwise)
    std::vector<int> ghost_prod;                                 // [units*64] producer cell of imported stream k of the unit
    std::vector<int> fold_of_slot;                               // reassociated form: [units*64] the leaf cell folded into the lane's cell, or -1 (empty: none)
    int n_folded = 0;
    int n_special = -1;                                          // reassociated form: -1 = plan of pairs; >= 0 = single-sum plan with that many cells in pair units
    int n_pair_units = 0;                                        // single-sum plan: its pair units (the last ones of unit_order)
};

// ---- shared by the two planners (xh_flow_plan.cpp: sums in stored order; xh_flow_rsum.cpp: reassociated sums)
namespace xh_flow {
struct Tree {
    int n = 0;
    const int64_t *indptr = nullptr;
    const int32_t *indices = nullptr;
    const int8_t *sign = nullptr;
    std::vector<int> ds, nchild, child_ptr, child, cell_pre, cell_post;
    std::vector<char> ok;                 // per cell: its network is a plain tree
};
// which networks are plain trees (rows {-1 on the diagonal, +1 elsewhere}, every cell feeds <= 1 row, <= 9 terms per row,
// no cycle); downstream pointers, children lists, row shapes
void tree_analyse(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign, const int *comp, int ncomp,
                  Tree &t);
}  // namespace xh_flow

// Partition every tree-shaped network (each cell drains to at most one cell, no cycle, rows = {-1 on the diagonal, +1
// elsewhere}, at most 9 terms).  handled[c] = 1 for the cells the tables route.  Returns 0, or -1 with `err` set.
int flow_tables_build(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign, const int *comp,
                      int ncomp, const FlowPlanOptions &opt, std::vector<char> &handled, FlowTables &out,
                      std::string &err);

// Invariants of a built plan (every cell in exactly one slot; streams run strictly down the pipeline; <= 16 imports
// and outlets per unit; lane lags even and consistent with the "two iterations earlier" rule; every row, expanded
// through its chains, equals the CSR row in stored order).  Empty = fine.
std::string flow_tables_check(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                              const std::vector<char> &handled, const FlowTables &t);

// The reassociated ("tolerance") form: every upstream neighbour of a cell passes a running sum along a chain, a cell reads
// the chain's total and its own predecessor -- two values per sub-step for every unit, sums equal to the CSR row's to
// rounding, not bit for bit (xh_flow_rsum.cpp).  Same contract as flow_tables_build / flow_tables_check; routes the same cells.
int flow_tables_build_rsum(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign, const int *comp,
                           int ncomp, const FlowPlanOptions &opt, std::vector<char> &handled, FlowTables &out,
                           std::string &err);
std::string flow_tables_check_rsum(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                                   const std::vector<char> &handled, const FlowTables &t);
// bumped with every change of the reassociated planner that alters its partitions (part of the key of the per-box cache)
int flow_rsum_planner_version();

// Tables to / from a file (the per-box cache of xh_route_plan_prepare: a partition costs tens of milliseconds, reading it
// back a few).  flow_tables_load returns false on any mismatch of format or size; callers hold what they read to
// flow_tables_check before using it.
bool flow_tables_save(const FlowTables &t, const char *path);
bool flow_tables_load(const char *path, FlowTables &t);
