/*
 * xanthos_hip.h -- C-ABI of libxanthos_hip.so: the MI355X (gfx950) implementation of the Xanthos monthly
 * PET -> runoff -> routing hot path.
 *
 * The reference (JGCRI/xanthos v2.4.1) is pure Python/numpy and has no FFI of its own; its "plugin API" for this
 * path is a set of module-level Python functions that xanthos/components.py calls.  Each entry point below states
 * the reference function (file:line under /root/reference) whose work it replaces.  The Python stubs that bind
 * these symbols with ctypes are in xanthos_amd/_hip.py; INTEGRATION.md shows the lines a Xanthos maintainer would
 * add to pet/penman_monteith.py, runoff/abcd.py and routing/mrtm.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, fp64 only.  All 2-D arrays are [ncell, nmonths], C order (month
 *     fastest) -- exactly the reference's in-memory layout, so uploads/downloads are plain copies.
 *   - Pointers named d_* are DEVICE pointers (from xh_malloc, or any HIP allocation on the context's device);
 *     pointers named h_* are HOST pointers (small index/table arrays read during the call).
 *   - Every function returns XH_OK (0) or an error code; xh_last_error(ctx) gives the message.
 *   - Kernels are enqueued on the context's own HIP stream and the call returns without waiting; xh_sync,
 *     xh_memcpy_d2h and xh_timing_get wait for the stream.  Calls on one context must not be concurrent
 *     (one context per host thread); different contexts are independent.
 *   - A call never falls back to the CPU: if the device or the code object is unusable it fails.
 */
#ifndef XANTHOS_HIP_H
#define XANTHOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XH_OK 0
#define XH_ERR_ARG 1     /* bad argument (NULL, size, unsupported value)          */
#define XH_ERR_HIP 2     /* a HIP runtime call or kernel failed                   */
#define XH_ERR_LIMIT 3   /* problem exceeds a compiled-in limit (e.g. land classes) */
#define XH_ERR_DEVICE 4  /* routing kernel reported a device-side fault (spin timeout) */

#define XH_MAX_LCS 32    /* max land-cover classes in xh_pm_pet */

typedef struct xh_ctx xh_ctx;
typedef struct xh_route_plan xh_route_plan;

/* ------------------------------------------------------------------ context, memory, timing */
int xh_abi_version(void);                                  /* bumps when a signature changes */
int xh_device_count(int *n);
int xh_ctx_create(int device, xh_ctx **out);
void xh_ctx_destroy(xh_ctx *ctx);
const char *xh_last_error(const xh_ctx *ctx);              /* ctx may be NULL: last error of a failed xh_ctx_create */
int xh_device_name(xh_ctx *ctx, char *buf, size_t len);

int xh_malloc(xh_ctx *ctx, size_t bytes, void **d_ptr);
int xh_free(xh_ctx *ctx, void *d_ptr);
int xh_memcpy_h2d(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int xh_memcpy_d2h(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* waits for the stream */
int xh_memcpy_d2d(xh_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* Page-locked host memory and transfers that do not wait: a loader / writer that keeps its arrays in xh_host_alloc
 * memory moves them at PCIe rate and overlaps them with kernels; the buffers must stay untouched until xh_sync. */
int xh_host_alloc(xh_ctx *ctx, size_t bytes, void **h_ptr);
int xh_host_free(xh_ctx *ctx, void *h_ptr);
int xh_memcpy_h2d_async(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int xh_memcpy_d2h_async(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* File <-> HBM without a host copy of the data: `bytes` of `path` from byte `offset` (the body of a .npy: the loader of
 * data_load.py:186-195, :342-350 keeps np.load's memory map instead of a 324 MB host array) are mapped read-only and
 * copied out of the mapping (34 GB/s from the page cache on the MI355X box against 8 GB/s for read() + copy; `threads`
 * is ignored).  xh_download_file is the writer's side (data_writer/out_writer.py np.save: the caller writes the header,
 * this call the body; the file is created when missing, never truncated): device -> page-locked slots -> write(), the
 * copy of one chunk under the write of the one before (15 GB/s against 6 GB/s for a download + np.save).  Both wait for
 * earlier work of the context and return when the transfer is complete.  XH_ERR_ARG: file missing / too short / write
 * error (xh_last_error has errno's text). */
int xh_upload_file(xh_ctx *ctx, void *d_dst, const char *path, uint64_t offset, size_t bytes, int threads);
int xh_download_file(xh_ctx *ctx, const void *d_src, const char *path, uint64_t offset, size_t bytes, int threads);
/* Several device arrays to several files at once (n <= 16), one writer thread per file: the output variables of one run. */
int xh_download_files(xh_ctx *ctx, int n, const void *const *d_srcs, const char *const *paths, const uint64_t *offsets,
                      const size_t *bytes);
int xh_memset(xh_ctx *ctx, void *d_ptr, int value, size_t bytes);
int xh_sync(xh_ctx *ctx);
/* Gather / scatter whole rows of a [nrows_total, ncols] device array by row index (shard packing, samples). */
int xh_gather_rows(xh_ctx *ctx, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                   double *d_dst);
int xh_scatter_rows(xh_ctx *ctx, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                    double *d_dst);
/* [rows, cols] -> [cols, rows] */
int xh_transpose(xh_ctx *ctx, const double *d_src, int64_t rows, int64_t cols, double *d_dst);

/* HIP-event timing of the kernels each entry point launches, accumulated per kernel name on the context's stream.
 * Names: "pm_pet", "abcd_spinup", "abcd_basin_mean", "abcd_sim", "mrtm_route", "calib_abcd", "calib_kge", "calib_de",
 * "agg_time", "agg_spatial", "drought_thresh", "drought_stats".  xh_timing_get waits for the stream, then returns total milliseconds and launch count. */
int xh_timing_reset(xh_ctx *ctx);
/* a caller-named span on the context's stream, read back with xh_timing_get like the library's own timers (one open
 * at a time): e.g. what a step still spends in the write-out gather after the routing kernel has ended                */
int xh_mark_begin(xh_ctx *ctx, const char *name);
int xh_mark_end(xh_ctx *ctx);
int xh_timing_enable(xh_ctx *ctx, int on);
int xh_timing_get(xh_ctx *ctx, const char *name, double *total_ms, int64_t *launches);

/* ------------------------------------------------------------------ Penman-Monteith PET
 * Replaces pet/penman_monteith.py:run_pmpet (:394-477) with SetData (:17-99), et_veg (:223-334),
 * et_water (:337-361) and et_snow (:364-377) fused into one kernel over (cell, month).
 * Host tables are the DataLoader fields of data_load.py:94-117.                                         */
typedef struct xh_pm_tables {
    int32_t nlcs;                       /* number of land-cover classes (>= 7: rows 0 and 6 are hard-wired, :361,:377) */
    const double *cL, *beta, *rslimit, *Tminopen, *Tminclose, *VPDclose, *VPDopen, *RBLmin, *RBLmax, *rc,
        *emiss;                         /* host, [nlcs] each (gcam_ET_para.csv columns 0-2, 5-12)      */
    const double *alpha, *lai, *laimin, *laimax; /* host, [nlcs*12] each, row = class, col = month of year */
} xh_pm_tables;

int xh_pm_pet(xh_ctx *ctx, const xh_pm_tables *h_tab,
              int64_t ncell, int32_t nmonths,       /* nmonths = 12 * number of years                    */
              int32_t start_year,
              int32_t n_lc_years, const int32_t *h_lc_years, /* years of the land-cover slices (pm_lc_years) */
              int32_t water_idx, int32_t snow_idx,
              const double *d_tas, const double *d_tmin, const double *d_rhs, const double *d_wind,
              const double *d_rsds, const double *d_rlds,
              const double *d_tairprev,             /* [ncell,nmonths] or NULL: NULL = tas of the previous CELL,
                                                       zeros for cell 0 (data_load.py:128-129)            */
              const double *d_lct,                  /* [ncell, nlcs, n_lc_years]                          */
              const double *d_elev,                 /* [ncell]                                            */
              double *d_pet);                       /* out [ncell, nmonths]                               */

/* ------------------------------------------------------------------ ABCD runoff
 * Replaces runoff/abcd.py:abcd_execute / abcd_parallel / _run_basins (:314-422) and the ABCD class
 * (:41-311): spin-up, per-basin December means (set_vals :246-282), simulation.
 *   h_basin_index[ncell] : dense 0-based group id used for the spin-up means (cells of one basin share it)
 *   h_par_index[ncell]   : row of d_pars used by each cell (basin row for abcd_execute; cell row for ABCD(pars))
 *   d_pars[npar_rows,5]  : a, b (x1000 applied inside, abcd.py:48), c, d, m
 *   d_tmin may be NULL   : no snow component (abcd.py:44,51,144-146)
 * Outputs may individually be NULL.  spinup < 25 is rejected like the reference's IndexError (:258-266).  */
int xh_abcd(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t spinup, int32_t n_groups,
            const int32_t *h_basin_index, const int32_t *h_par_index, int64_t npar_rows, const double *d_pars,
            const double *d_pet, const double *d_precip, const double *d_tmin,
            double *d_aet, double *d_q, double *d_sav,
            double *d_sm0, double *d_gw0);          /* optional out [n_groups]: post-spin-up initial state */

/* ------------------------------------------------------------------ MRTM routing
 * xh_route_plan_create replaces the per-call topology work of routing/mrtm.py:upstream_genmatrix (:194-230):
 * it takes UM = UP - I as CSR (row i = sorted columns j with +1 for "j flows into i" and -1 on the diagonal),
 * finds the independent river networks and lays them out for the LDS-resident kernel.  Row order of the
 * entries is preserved so sums are accumulated exactly as scipy's csr mat-vec does.                       */
int xh_route_plan_create(xh_ctx *ctx, int64_t ncell, const int64_t *h_indptr, const int32_t *h_indices,
                         const int8_t *h_sign, xh_route_plan **out);
void xh_route_plan_destroy(xh_route_plan *plan);
/* info[0]=networks, [1]=largest network (cells), [2]=workgroup-per-network units, [3]=cells routed by the global
 * fallback, [4]=largest such unit (cells), [5]=padded slots, [6]=1 if every cell has one downstream cell,
 * [7]=dataflow units, [8]=stream edges between them, [9]=pipeline depth, [10]=cells routed by the dataflow kernel,
 * [11]=most imported streams of one unit, [12]=deepest lane lag of the time-skewed layout in sub-steps (-1: layout
 * not available), [13]=kernel that routed the tree networks in the last xh_route_series call on this plan (0 none,
 * 1 lock-step units with monthly streams -- months shorter than the lane lags, or rows beyond the time-skewed kernels'
 * 32-bit offsets --, 2 time-skewed units of the bit-exact kernel k_mrtm_wave, 4 the reassociated kernel k_mrtm_rsum: the
 * default), [14]=calls of this plan re-run with one workgroup per
 * network after a device fault, [15]=calls cross-checked by XH_ROUTE_VALIDATE */
int xh_route_plan_info(const xh_route_plan *plan, int64_t info[16]);

/* (Rounds 3-5 had a second, PLAIN form of the bit-exact kernel's units here -- selected by XH_ROUTE_TYPED or learnt call by
 * call, with xh_route_plan_typed_info.  Retired in round 6: the bit-exact kernels are the checker and the XH_ROUTE_EXACT
 * option now, every unit of theirs in pair form; what knows which cells can fire is the PREPARED plan of the default form,
 * below.) */
/* Optional, with the HOST copies of the arrays the calls will pass on the device: makes the PREPARED plan of the default
 * (reassociated) routing form -- xh_route_plan_rsum_info below says what is in it -- from WHICH cells can fire at these
 * velocities, lengths and dt.  The package's own callers do it for the user (the pipeline on its planning thread beside the
 * forcing upload; routing.mrtm.route_series / streamrouting from the L, ChV and dt they hold).  May be called again: the same
 * data is a no-op, other data replaces the prepared plan.  Results never depend on it (the kernel guards what the plan
 * assumes; a trip routes the call again on the plan of pairs).  Partitions are kept per box under $XH_CACHE_DIR or
 * ~/.cache/xanthos_amd (XH_ROUTE_LEARN_CACHE=0: not).  A no-op in a process whose default form is the bit-exact one. */
int xh_route_plan_prepare(xh_ctx *ctx, xh_route_plan *plan, const double *h_flow_dist, const double *h_velocity, double dt);

/* Diagnostics: with XH_FLOW_STATS=1 in the environment the dataflow kernel records, per unit, {shader cycles inside the
 * sub-step loops, shader cycles total, 100 MHz ticks total, shape bits + placement, cycles waiting for data, cycles waiting for ring space}; this call waits for the
 * stream and copies up to max_words 64-bit words (6 per unit) of the last xh_route_series launch. */
int xh_route_plan_stats(xh_route_plan *plan, int64_t max_words, uint64_t *h_words, int64_t *n_words);

/* Host-side topology (integer work, no device needed).
 * xh_mrtm_downstream replaces routing/mrtm.py:downstream + make_flowdirgrid (:85-120, :233-258): D8 decode, the
 * longitude wrap quirk ((col+1) mod ncol for off-grid columns), off-grid rows -> self, ocean/self targets -> -1.
 *   h_ilon / h_ilat [ncell]: 1-based column / row of each cell (coords[:,3], coords[:,4]); h_id [ncell]: 1-based
 *   cell ids (coords[:,0]); h_flowdir [ncell]: D8 codes (-9999 = none).  h_dsid out [ncell], 1-based or -1.
 * xh_mrtm_upstream replaces routing/mrtm.py:upstream (:123-191): for each cell its 8 in-grid neighbours (no
 * date-line wrap), those draining into it first, then the count.  h_upid out [ncell, 9].
 * xh_mrtm_um_csr replaces upstream_genmatrix (:194-230): UM = UP - I as CSR with ascending columns;
 *   h_indptr [ncell+1], h_indices / h_sign sized ncell + sum(upid[:,8]).                                      */
int xh_mrtm_downstream(int64_t ncell, int32_t nrow, int32_t ncol, const int64_t *h_id, const int32_t *h_ilon,
                       const int32_t *h_ilat, const double *h_flowdir, int64_t *h_dsid);
int xh_mrtm_upstream(int64_t ncell, int32_t nrow, int32_t ncol, const int64_t *h_id, const int32_t *h_ilon,
                     const int32_t *h_ilat, const int64_t *h_dsid, int64_t *h_upid);
int xh_mrtm_um_csr(int64_t ncell, const int64_t *h_upid, int64_t *h_indptr, int32_t *h_indices, int8_t *h_sign);

/* xh_route_series replaces routing/mrtm.py:streamrouting (:16-82) and the month loops of
 * components.py:calculate_routing (:273-294): spin-up over months [0, spinup_months) then all nmonths,
 * nt = int(nday*86400/dt) explicit-Euler sub-steps per month, all inside one persistent kernel.
 *   h_ndays[nmonths]; d_runoff [ncell, nmonths] (mm/month); d_flow_dist, d_velocity, d_area [ncell]
 *   d_S0 [ncell] or NULL (zeros); outputs d_chstorage / d_avgchflow [ncell, nmonths] (either may be NULL),
 *   d_S_end / d_F_end [ncell] optional.  streamrouting() itself is the nmonths = 1, spinup = 0 case.
 * flags: XH_ROUTE_ATOMIC routes with global fp64 atomic scatter-adds (non bit-reproducible variant).
 * The dataflow kernels (tree networks) keep every unit resident and let units wait for each other, each wait bounded
 * by a 5 s timeout (two orders of magnitude above the kernel's run time at the full grid).  On a device shared with
 * another routing call the units may not all fit: the timeout then raises a sticky device fault, and the next
 * synchronising call (xh_sync, xh_memcpy_d2h) re-routes every call enqueued since the last synchronisation with one
 * workgroup per network (no waits between workgroups) before it returns -- XH_OK if nothing else was enqueued behind
 * the routing, XH_ERR_DEVICE (routing outputs valid, later results not) otherwise.  xh_route_plan_info[14] counts such
 * re-runs.  After such a fault the plan's next 8 calls (16, 32 ... 256 when faults repeat) skip the dataflow kernels, so
 * a device that stays shared does not cost a timeout per call; a fault-free dataflow call resets the back-off.
 * Bit-exactness of the dataflow kernels rests on a hardware assumption stated at xh_mrtm_wave.hip ("MEMORY-ORDERING
 * ASSUMPTION": write-through stream stores retire in order under s_waitcnt vmcnt); XH_ROUTE_VALIDATE checks a call
 * against the kernel that does not need it.                                                                        */
#define XH_ROUTE_DEFAULT 0
#define XH_ROUTE_FORCE_FALLBACK 1   /* route every network with the global-memory kernels (testing)        */
#define XH_ROUTE_ATOMIC 2           /* with the fallback: scatter-add outflow with global_atomic_add_f64   */
#define XH_ROUTE_NO_DATAFLOW 4      /* one workgroup per network even for tree-shaped networks (testing)   */
#define XH_ROUTE_NO_SKEW 8          /* dataflow units in lock-step with monthly streams, not time-skewed   */
#define XH_ROUTE_TEST_FAULT 16      /* testing: the dataflow kernel raises its fault word as a timed-out wait would */
#define XH_ROUTE_VALIDATE 32        /* cross-check: after a dataflow kernel routed the call, route it again with one
                                       workgroup per network (barriers only, no streams between units) and compare every
                                       output bit; synchronous; XH_ERR_DEVICE on a difference.  Also switched on for every
                                       call by XH_ROUTE_VALIDATE=1 in the environment.  xh_route_plan_info[15] counts the
                                       validated calls.                                                            */
/* (64 was XH_ROUTE_TYPED until round 5: ignored now) */
#define XH_ROUTE_REASSOC 128        /* tree networks by the REASSOCIATED form of the time-skewed kernel (k_mrtm_rsum): the row sum
                                       of mrtm.py:50-51 travels as running sums along chains of lanes (two LDS reads per
                                       sub-step for every unit instead of up to six) and the update of mrtm.py:54-69 is fused.
                                       Equal to the reference to rounding -- <= 1e-9 relative on every routed value at the
                                       full grid, identical NaN masks; the gate is 1e-6 -- NOT bit for bit.  THE DEFAULT since
                                       round 5 for calls that carry neither this flag nor XH_ROUTE_EXACT; XH_ROUTE_REASSOC=0 /
                                       1 in the environment moves that default.  xh_route_plan_info[13] = 4 when it routed
                                       the call (months shorter than its lane lags, networks that are not trees: the
                                       bit-exact kernels).  XH_ROUTE_VALIDATE then compares within 1e-9.               */
#define XH_ROUTE_EXACT 256          /* the bit-exact kernels for this call (every row sum in scipy's stored order) whatever
                                       the default says                                                                  */
/* The reassociated plan the last call ran on: info[8] = {its units; the leaves it folded into their downstream cells' lanes;
 * 1 if a guard trip has switched the prepared plan off; folded leaves of the prepared plan; cells in pair form of the plan
 * the last call ran on, -1 if ALL its lanes pass pairs of sums; the same for the prepared plan (-1: pairs, or none prepared);
 * calls routed again on the plan of pairs after a guard trip, so far; pair units of the plan the last call ran on}.
 * The PREPARED plan is the one xh_route_plan_prepare makes from the call's velocities, lengths and dt (reassociated form
 * only; XH_FLOW_FOLD=0 / XH_RSUM_SINGLE=0 switch its two parts off):
 *  - leaves that cannot fire (velocity * dt / length < 1) of river networks small enough to have no streams are carried by
 *    their parents' lanes, which frees enough lanes for every unit to have a SIMD of its own;
 *  - the lanes pass ONE running sum (of the adjusted flows) instead of the pair {sum F, sum F2}: only a cell that may fire
 *    AND has an upstream neighbour that may needs both -- the cells that can fire by construction with such a neighbour and
 *    a halo of XH_RSUM_HALO (8) cells downstream of them, where the reference's corner S1 >= 0 > S2 (mrtm.py:54, :66-69)
 *    sends negative flows -- and those few sit in pair units of their own.
 * Both rest on which cells can fire; the kernel guards the assumptions (folded leaves' storage, lateral inflow and initial
 * storage >= 0, the outflow of the halos' exit cells >= 0) and a trip routes the call again on the plan of pairs.          */
int xh_route_plan_rsum_info(const xh_route_plan *plan, int64_t info[8]);
int xh_route_series(xh_ctx *ctx, xh_route_plan *plan, int32_t nmonths, int32_t spinup_months,
                    const int32_t *h_ndays, double dt,
                    const double *d_flow_dist, const double *d_velocity, const double *d_area,
                    const double *d_runoff, const double *d_S0,
                    double *d_chstorage, double *d_avgchflow, double *d_S_end, double *d_F_end, int32_t flags);

/* ------------------------------------------------------------------ the whole path as one pipelined call
 * xh_run_fused replaces the stage-after-stage hand-over of components.py:simulation (:344-370: calculate_pet ->
 * calculate_runoff -> calculate_routing): Penman-Monteith, ABCD (spin-up, basin means, simulation) and MRTM are enqueued
 * together -- PM runs in blocks of `block_months` months and the ABCD march follows one block behind on a second stream
 * (PET still in cache, the march on the issue slots PM leaves empty, its state carried from block to block); routing
 * follows when the last block of runoff exists.  Results are bit-identical to xh_pm_pet + xh_abcd +
 * xh_route_series with the same arguments (same kernels, same arithmetic).  Arguments mean what they mean there.
 * d_pet and d_q must be given (they are the stages' hand-over and outputs of the model); d_aet, d_sav, d_chstorage,
 * d_avgchflow may be NULL; plan = NULL stops after the runoff.  block_months = 0 picks the default (96); it must be a
 * multiple of 48.  Asynchronous; on return the context's stream is ordered after every output.                    */
typedef struct xh_fused_args {
    int64_t ncell;
    int32_t nmonths, start_year;
    const xh_pm_tables *pm;
    int32_t n_lc_years;
    const int32_t *h_lc_years;
    int32_t water_idx, snow_idx;
    const double *d_tas, *d_tmin, *d_rhs, *d_wind, *d_rsds, *d_rlds, *d_tairprev, *d_lct, *d_elev;
    int32_t abcd_spinup, n_groups;
    const int32_t *h_basin_index, *h_par_index;
    int64_t npar_rows;
    const double *d_pars, *d_precip, *d_abcd_tmin;
    xh_route_plan *plan;
    int32_t routing_spinup;
    const int32_t *h_ndays;
    double dt;
    const double *d_flow_dist, *d_velocity, *d_area, *d_S0;
    int32_t route_flags;
    double *d_pet, *d_aet, *d_q, *d_sav, *d_chstorage, *d_avgchflow;
    int32_t block_months;
    int32_t mode;       /* 0: PM blocks with the ABCD march one block behind, then routing (above).  1 (needs plan): the first
                         * max(spin-ups) months of PM and ABCD, then the ROUTING KERNEL, and the remaining months of PM and ABCD
                         * beside it on a second stream, fed to the running kernel through a months-ready word; falls back to
                         * the stage-by-stage order when the plan is not routed by the time-skewed dataflow kernel. Same results. */
} xh_fused_args;
int xh_run_fused(xh_ctx *ctx, const xh_fused_args *args);

/* ------------------------------------------------------------------ calibration objective
 * Replaces calibrate/calibrate_abcd.py:basin_runoff + objective_kge (:134-213) for set_calibrate = 0, batched
 * over a population: every member runs ABCD on the basin's cells, the simulated runoff is summed over cells per
 * month (nansum; x area x 1e-6 if d_area != NULL) and scored against h_obs with ED = 1 - KGE.
 *   d_pet_t / d_precip_t / d_tmin_t : [nmonths, ncell_b] (cell fastest; use xh_transpose), d_tmin_t may be NULL
 *   h_pars [nmembers, npar] with npar = 5 (a,b,c,d,m) or 4 (no snow)
 *   h_ed [nmembers] out; h_series [nmembers, nmonths] optional out                                          */
int xh_calib_objective(xh_ctx *ctx, int64_t ncell_b, int32_t nmonths, int32_t spinup, int32_t nmembers,
                       int32_t npar, const double *h_pars, const double *d_pet_t, const double *d_precip_t,
                       const double *d_tmin_t, const double *d_area, const double *h_obs, double *h_ed,
                       double *h_series);

/* ------------------------------------------------------------------ output aggregation
 * xh_agg_time replaces data_writer/out_writer.py:agg_to_year (:237-248) and the mm -> km3 scaling of write()
 * (:111-112): out[c, g] = f(in[c, g*group .. (g+1)*group)) x (d_scale ? d_scale[c] : 1), with f = NaN-skipping sum
 * (mode 0; an all-NaN block gives 0, as pandas does) or NaN-skipping mean (mode 1; all-NaN gives NaN).
 * group = 12 aggregates months to years; group = 1, mode 0 is a plain per-cell scaling (NaN kept).  Mode 2 is
 * np.sum over the block in numpy's own order (eight accumulators, NaN propagates): the yearly totals of
 * accessible/accessible.py:41-42, bit for bit.
 * xh_agg_spatial replaces out_writer.py:agg_spatial (:250-265): out[k, t] = NaN-skipping sum over the cells with
 * h_group[c] == k (h_group is 0-based, -1 = cell not aggregated); groups without cells give NaN.            */
int xh_agg_time(xh_ctx *ctx, int64_t ncell, int32_t ncols, int32_t group, int32_t mode, const double *d_scale,
                const double *d_in, double *d_out);
int xh_agg_spatial(xh_ctx *ctx, int64_t ncell, int32_t ncols, int32_t n_groups, const int32_t *h_group,
                   const double *d_in, double *d_out);
/* Loader transform on the device: np.nan_to_num in place, as data_load.py applies to the PM forcings
 * and the ABCD tmin (:120-125, :194-195): NaN -> 0, +inf / -inf -> +/- largest finite double.                */
int xh_nan_to_num(xh_ctx *ctx, double *d_arr, int64_t n);

/* ------------------------------------------------------------------ drought statistics
 * xh_drought_thresholds replaces drought/drought_stats.py:getthresh (:150-171) as called by calculate_thresholds
 * (:69-83): for every cell and each of the nper periods of the year, the quantile over the nyear samples
 * d_hydro[c, month0 + y*nper + p] (y < nyear), by numpy's "linear" method.  The caller passes what depends only on
 * (nyear, q): the ranks k_prev / k_next of the two order statistics and the weight gamma = (nyear-1) q - k_prev
 * (xanthos_amd/drought/drought_stats.py computes them the way numpy does).  A NaN sample gives a NaN threshold.
 *   d_hydro [ncell, nmonths]; d_thresh out [nper, ncell] (the layout of the reference's thresholds file).
 * xh_drought_stats replaces droughtstats (:85-148): severity / intensity / duration [ncell, nmonths] (any may be
 * NULL) from d_hydro [ncell, nmonths] and d_thresh [nthresh, ncell]; month t uses threshold row t mod nthresh.   */
int xh_drought_thresholds(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t month0, int32_t nyear, int32_t nper,
                          int32_t k_prev, int32_t k_next, double gamma, const double *d_hydro, double *d_thresh);
int xh_drought_stats(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t nthresh, const double *d_hydro,
                     const double *d_thresh, double *d_severity, double *d_intensity, double *d_duration);

/* The same objective for SEVERAL basins in one launch, each basin with its own population: one basin alone is only
 * months x ~1.7 us of dependent chain, far too little to fill the chip.  h_ncell [nbasins]; h_pars [nbasins, nmembers,
 * npar]; h_pet_t / h_precip_t / h_tmin_t / h_area: host arrays of nbasins DEVICE pointers ([nmonths, ncell_b] each;
 * h_tmin_t NULL for npar = 4, h_area NULL for mm_per_mth); h_obs [nbasins, nmonths]; h_ed [nbasins, nmembers] out;
 * h_series [nbasins, nmembers, nmonths] optional out.                                                        */
int xh_calib_objective_multi(xh_ctx *ctx, int32_t nbasins, const int64_t *h_ncell, int32_t nmonths, int32_t spinup,
                             int32_t nmembers, int32_t npar, const double *h_pars, const double *const *h_pet_t,
                             const double *const *h_precip_t, const double *const *h_tmin_t,
                             const double *const *h_area, const double *h_obs, double *h_ed, double *h_series);

/* ------------------------------------------------------------------ differential evolution on the device
 * Replaces the scipy.optimize.differential_evolution call of calibrate/calibrate_abcd.py:calibrate_basin (:103-112,
 * SciPy defaults: best1bin, Latin-hypercube start, dither (0.5, 1), recombination 0.7, tol 0.01, polish off) AND the
 * serial loop over basins of calibrate_all (:256-262): all basins of the session search at once in lock-step
 * generations; populations, trial vectors, energies and convergence flags stay in HBM and a generation is five
 * kernel launches with no host work.  Selection is generation-synchronous (SciPy's updating='deferred').
 * Random numbers are counter-based on (seed, h_basin_key[b], generation, member): a basin's search does not depend
 * on the other basins of the session or on the device, so basin-sharded multi-GPU runs reproduce one-GPU runs.
 *   xh_calib_de_create : basins as for xh_calib_objective_multi (device forcing [nmonths, ncell_b] per basin, which
 *                        must outlive the session); h_basin_key [nbasins] or NULL (= 0..nbasins-1); h_lo / h_hi
 *                        [npar] parameter bounds (:62-64); nmembers = popsize x npar in SciPy's terms
 *   xh_calib_de_init   : Latin-hypercube population and its energies
 *   xh_calib_de_step   : ngen generations; a basin stops when std(E) <= atol + tol |mean(E)| (SciPy's test);
 *                        *h_n_active = basins still searching after the last generation
 *   xh_calib_de_result : h_x [nbasins, npar] best parameters, h_fun [nbasins] = ED = 1 - KGE, h_nfev, h_nit,
 *                        h_active (each may be NULL)
 *   xh_calib_de_state  : which = 0 population + energies, 1 last trial vectors + their energies (unit cube),
 *                        2 last trial vectors scaled to parameter space; [nbasins, nmembers, npar] / [nbasins, nmembers]
 *   xh_calib_de_set_state : overwrite population and energies and re-activate every basin (tests, restarts)   */
typedef struct xh_calib_de xh_calib_de;
int xh_calib_de_create(xh_ctx *ctx, int32_t nbasins, const int64_t *h_ncell, const uint64_t *h_basin_key,
                       int32_t nmonths, int32_t spinup, int32_t nmembers, int32_t npar,
                       const double *const *h_pet_t, const double *const *h_precip_t, const double *const *h_tmin_t,
                       const double *const *h_area, const double *h_obs, const double *h_lo, const double *h_hi,
                       uint64_t seed, xh_calib_de **out);
void xh_calib_de_destroy(xh_calib_de *de);
int xh_calib_de_init(xh_calib_de *de);
int xh_calib_de_step(xh_calib_de *de, int32_t ngen, double tol, double atol, double mut_lo, double mut_hi,
                     double recombination, int32_t *h_n_active);
int xh_calib_de_result(xh_calib_de *de, double *h_x, double *h_fun, int64_t *h_nfev, int32_t *h_nit,
                       int32_t *h_active);
int xh_calib_de_state(xh_calib_de *de, int32_t which, double *h_vectors, double *h_energy);
int xh_calib_de_set_state(xh_calib_de *de, const double *h_pop, const double *h_energy, int32_t generation);

/* ------------------------------------------------------------------ multi-GPU write-out (RCCL over xGMI)
 * The reference has no distributed path; BASELINE's north star shards the 235 basins over the GPUs of a node with a
 * single RCCL gather at write-out.  One process per GPU; the launcher (torchrun, mpirun, anything) starts the ranks
 * and carries the 128-byte id from rank 0 to the others.  RCCL is bound at run time (dlopen of librccl.so.1).
 *   xh_comm_unique_id   : ncclGetUniqueId (rank 0); len >= 128
 *   xh_comm_create      : ncclCommInitRank on the context's device (collective over all ranks)
 *   xh_comm_gather_rows : every rank contributes nvar device arrays [h_counts[rank], ncols] (h_d_local: host array of
 *                         nvar device pointers -- the pipeline's own output buffers, nothing is staged or padded on the
 *                         senders); grouped ncclSend / ncclRecv of the exact sizes on the context's stream.  On the root
 *                         d_perm [sum h_counts] (device) holds, rank-major, the destination row of every gathered row and
 *                         h_d_out the nvar device arrays [total rows, ncols] that receive them in grid order.
 *                         Asynchronous like every call: xh_sync to wait.                                       */
typedef struct xh_comm xh_comm;
int xh_comm_unique_id(char *id, size_t len);
int xh_comm_create(xh_ctx *ctx, int32_t nranks, int32_t rank, const char *id, size_t len, xh_comm **out);
void xh_comm_destroy(xh_comm *comm);
/*   xh_comm_info        : info[4] = {ranks, this rank, ncclSend calls, ncclRecv calls issued through this communicator so far}.
 *                         XH_COMM_SELF_LOOP=1 in the environment when a communicator is created (testing on a one-GPU box, with
 *                         the REAL librccl): the root also sends its OWN rows to itself -- ncclSend and ncclRecv to the own
 *                         rank inside the group, which RCCL allows -- so they take the whole data path of a remote rank's rows
 *                         (send kernel, receive into the staging area, row scatter to grid order) instead of the local short cut. */
int xh_comm_info(const xh_comm *comm, int64_t info[4]);
int xh_comm_gather_rows(xh_ctx *ctx, xh_comm *comm, int32_t root, int32_t nvar, const double *const *h_d_local,
                        int64_t ncols, const int64_t *h_counts, const int64_t *d_perm, double *const *h_d_out);
/*   xh_comm_gather_rows_side : the same gather on the context's gather stream (a hardware queue of its own), ordered behind
 *                         the kernels that produced the arrays -- everything enqueued on the context so far, or, right after
 *                         an xh_run_fused in mode 1, the side stream that completed PET / AET / Q / Sav while the routing
 *                         kernel already runs -- and beside whatever the context does next: the write-out of the four
 *                         arrays the routing does not touch travels over xGMI while the routing kernel runs.  Give this
 *                         stream a communicator of its own.
 *   xh_comm_join        : orders the context's stream behind the side gather (xh_sync and every other synchronising
 *                         call do so too).                                                                            */
int xh_comm_gather_rows_side(xh_ctx *ctx, xh_comm *comm, int32_t root, int32_t nvar, const double *const *h_d_local,
                             int64_t ncols, const int64_t *h_counts, const int64_t *d_perm, double *const *h_d_out);
int xh_comm_join(xh_ctx *ctx);

/* ------------------------------------------------------------------ bench support (not on the hot path)
 * Fills the eight forcing arrays of the synthetic benchmark world on the device (same distributions as
 * xanthos_amd/synth.py:make_forcing); d_lat [ncell] degrees; nan_frac = share of cells whose precipitation is NaN.
 * d_cell_ids [ncell] or NULL: GLOBAL cell index of each row (the random streams are keyed on it, so a rank can generate
 * just its shard of a world; -1 = no cell: a row of zeros).  With only d_tas non-NULL, only temperature is generated
 * (the tairprev rows of a shard: the previous GLOBAL cell's temperature, data_load.py:128-129).             */
int xh_synth_forcing(xh_ctx *ctx, uint64_t seed, double nan_frac, int64_t ncell, int32_t nmonths, const double *d_lat,
                     const int64_t *d_cell_ids, double *d_tas, double *d_tmin, double *d_rhs, double *d_wind, double *d_rsds,
                     double *d_rlds, double *d_precip, double *d_abcd_tmin);

#ifdef __cplusplus
}
#endif
#endif /* XANTHOS_HIP_H */
