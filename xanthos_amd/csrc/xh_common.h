// Internal declarations shared by the libxanthos_hip.so translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "xanthos_hip.h"

// The library's switches (README lists them).  One place reads each of them.
static inline bool xh_env_on(const char *name, bool dflt) {      // "0" = off, anything else = on, unset = dflt
    const char *e = getenv(name);
    return e ? e[0] != '0' : dflt;
}
static inline std::string xh_cache_dir() {      // per-box cache of partitions and first-check markers; "" = none
    if (const char *d = getenv("XH_CACHE_DIR")) return d;
    if (const char *h = getenv("HOME")) return std::string(h) + "/.cache/xanthos_amd";
    return std::string();
}
static inline bool xh_plan_cache_on() { return xh_env_on("XH_ROUTE_LEARN_CACHE", true); }      // partitions kept per box
static inline bool xh_flow_debug() { return getenv("XH_FLOW_DEBUG") != nullptr; }               // partition statistics on stderr
static inline bool xh_flow_check() { return getenv("XH_FLOW_CHECK") != nullptr; }               // planner invariants on every plan

struct xh_timer_slot {
    double done_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// A routing call that ran on the dataflow kernels and has not been confirmed fault-free by a synchronisation yet.
// If the device fault word turns out to be set (units of two dataflow kernels could not all be resident and a bounded
// wait timed out), the synchronising call re-runs these calls with one workgroup per network (XH_ROUTE_NO_DATAFLOW),
// which has no dependencies between workgroups.
struct xh_route_record {
    xh_route_plan *plan = nullptr;
    int32_t nmonths = 0, spinup_months = 0, flags = 0;
    std::vector<int32_t> ndays;
    double dt = 0.0;
    const double *flow_dist = nullptr, *velocity = nullptr, *area = nullptr, *runoff = nullptr, *S0 = nullptr;
    double *chs = nullptr, *avg = nullptr, *S_end = nullptr, *F_end = nullptr;
    uint64_t seq_after = 0;        // ctx->work_seq right after the call was enqueued
    bool fed = false;              // routed while the side stream still produced its runoff (xh_run_fused mode 1)
};
int xh_route_rerun(xh_ctx *ctx, const xh_route_record &r, bool dataflow_pairs);      // xh_mrtm.hip: workgroup per network, or the dataflow kernel with every unit in pair form
void xh_route_confirm(const xh_route_record &r);                // xh_mrtm.hip: the call's dataflow kernel ran fault-free
void xh_route_backoff(xh_route_plan *plan);                     // xh_mrtm.hip: a fault was seen on this plan: skip the dataflow kernels for a while

struct xh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool timing = true;
    std::map<std::string, xh_timer_slot> timers;
    std::vector<hipEvent_t> event_pool;
    // grow-only scratch buffers (device)
    void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[4] = {0, 0, 0, 0};
    uint64_t scratch_gen[4] = {0, 0, 0, 0};      // bumped by every xh_scratch call on the slot: "did anybody else write there since?"
    // what xh_pm_prepare / xh_abcd_prepare uploaded last (host copies) and the slot generation right after: a pipeline that
    // is run again and again with the same tables and cell lists skips the uploads and the stream synchronisation they need
    std::vector<char> pm_cache, abcd_cache;
    uint64_t pm_cache_gen = 0, abcd_cache_gen = 0;
    hipDeviceProp_t prop;
    // device-side fault word (bounded spins of the dataflow routing kernel) + pinned host mirror
    unsigned *d_fault = nullptr;
    unsigned *h_fault = nullptr;
    bool fault_pending = false;
    std::vector<xh_route_record> pending_routes;
    uint64_t work_seq = 0;         // bumped by every entry point that enqueues work on the stream (kernels, copies, row movers)
    int64_t reroutes = 0;          // routing calls re-run after a device fault
    // xh_run_fused: side stream [0] (lowest priority: the fillers beside the routing kernel) and events of the pipelines;
    // side stream [1] (highest priority: a queue of its own): the write-out gather of PET / AET / Q / Sav beside the routing
    hipStream_t side_stream[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> side_events;
    hipEvent_t runoff_event = nullptr;     // xh_run_fused mode 1: recorded on side stream [0] when PET / AET / Q / Sav are final
    bool runoff_event_fresh = false;       // ... by the last call, and not consumed by a side gather yet
    hipEvent_t gather_event = nullptr;     // last work of side stream [1]
    bool gather_pending = false;           // side stream [1] holds work the context's stream has not been joined with
    // xh_mark_begin / xh_mark_end: one caller-named span on the context's stream
    hipEvent_t mark_a = nullptr, mark_b = nullptr;
    bool mark_open = false;
    std::string mark_name;
    // xh_run_fused mode 1: hand-over words (months ready, placement epoch) + the routing kernel's staged runoff (grow-only)
    void *d_feed = nullptr;
    size_t feed_bytes = 0;
    unsigned feed_epoch = 0;
    // xh_upload_file / xh_download_file: page-locked chunk ring of the file movers (grow-only)
    void *io_ring = nullptr;
    size_t io_ring_bytes = 0;
    // fed order (xh_run_fused mode 1): work_seq at the end of the call that left `runoff_event` (a side gather may use the event
    // only while nothing else has been enqueued on the context since); whether the side stream has a hardware queue of its own
    // (a priority level apart from the context's stream -- without one the fed order cannot run); switched off after a fed
    // call's routing kernel waited in vain for its months (serialised kernels, e.g. under a counter-collecting profiler)
    uint64_t runoff_seq = 0;
    bool feed_queue_ok = true, feed_disabled = false;
};

// Fault code of the routing kernel's plain units: an input outside the argument that lets them gather one value per term
// (a cell that cannot fire did).  Nothing timed out; the call is routed again in pair form (xh_fault_check).
#define XH_FAULT_GUARD 4u
#define XH_ROUTE_NO_PLAIN 0x4000      /* internal flag of xh_route_series: every dataflow unit in pair form */

// Device fault word: kernels set it non-zero instead of hanging; the next synchronising call reports XH_ERR_DEVICE.
int xh_fault_word(xh_ctx *ctx, unsigned **d_word);   // lazily allocated and zeroed ONCE: the word is sticky until a check clears it
int xh_fault_collect(xh_ctx *ctx);                   // enqueue device -> pinned host copy after the kernel
int xh_fault_check(xh_ctx *ctx);                     // after a stream sync: XH_ERR_DEVICE if the word was set
// Stream synchronisation + xh_fault_check.  EVERY entry point that synchronises the context stream, releases or overwrites
// device memory goes through this, so a routing call that has to be re-run is re-run before its inputs / outputs can be
// freed or replaced, and the caller hears about it (XH_ERR_DEVICE when work enqueued behind the routing read invalid data).
int xh_settle(xh_ctx *ctx);

int xh_fail(xh_ctx *ctx, int code, const char *fmt, ...);
extern std::string g_xh_create_error;

#define XH_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return xh_fail((ctx), XH_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                    \
    } while (0)

#define XH_REQUIRE(ctx, cond, ...)                                   \
    do {                                                             \
        if (!(cond)) return xh_fail((ctx), XH_ERR_ARG, __VA_ARGS__); \
    } while (0)

// side stream [1], created on first use (highest priority: never shares a hardware queue with the context's stream)
int xh_gather_stream(xh_ctx *ctx, hipStream_t *out);
// row movers on a given stream (xh_gather_rows / xh_scatter_rows are these on the context's stream)
int xh_move_rows_on(xh_ctx *ctx, hipStream_t st, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                    double *d_dst, int scatter);

// scratch slot `which` with at least `bytes` bytes (device memory owned by the context)
int xh_scratch(xh_ctx *ctx, int which, size_t bytes, void **out);

// RAII-free timing helpers: record a start/stop event pair around kernels of one entry point
struct xh_span {
    xh_ctx *ctx;
    const char *name;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t stream = nullptr;
};
xh_span xh_span_begin(xh_ctx *ctx, const char *name);
xh_span xh_span_begin_on(xh_ctx *ctx, const char *name, hipStream_t stream);   // kernels of a pipelined call on a side stream
void xh_span_end(xh_span &s);
void xh_span_cancel(xh_span &s);

static inline int xh_is_leap_gregorian(int y) { return (y % 4 == 0 && y % 100 != 0) || (y % 400 == 0); }
