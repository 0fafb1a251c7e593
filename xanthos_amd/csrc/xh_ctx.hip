// Context, device memory, row gather/scatter, transpose and HIP-event timing for libxanthos_hip.so.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "xh_common.h"

std::string g_xh_create_error;

int xh_fail(xh_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    // (a routing plan may be made on a host thread of its own while the main thread works on the same context:
    // the error text is the one field both may write)
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (ctx)
        ctx->err = buf;
    else
        g_xh_create_error = buf;
    return code;
}

int xh_scratch(xh_ctx *ctx, int which, size_t bytes, void **out) {
    int rc_device = XH_OK;      // XH_ERR_DEVICE of the settle below: reported after the buffer has been replaced
    ctx->scratch_gen[which] += 1;
    if (bytes > ctx->scratch_bytes[which]) {
        if (ctx->scratch[which]) {
            // the buffer may be read by a routing call that still has to be confirmed (or re-run): settle before freeing
            const int rc = xh_settle(ctx);
            if (rc && rc != XH_ERR_DEVICE) return rc;
            rc_device = rc;
            XH_HIP(ctx, hipFree(ctx->scratch[which]));
            ctx->scratch[which] = nullptr;
            ctx->scratch_bytes[which] = 0;
        }
        size_t want = bytes + (bytes >> 2) + 4096;
        XH_HIP(ctx, hipMalloc(&ctx->scratch[which], want));
        ctx->scratch_bytes[which] = want;
    }
    *out = ctx->scratch[which];
    // A re-routed call whose invalid outputs later work had already consumed: the settle has cleared the bookkeeping, so
    // this is the one place the caller can still hear about it (xh_common.h, contract of xh_settle).
    return rc_device;
}

int xh_fault_word(xh_ctx *ctx, unsigned **d_word) {
    if (!ctx->d_fault) {
        XH_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_fault), 64));
        XH_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_fault), 64, hipHostMallocDefault));
        *ctx->h_fault = 0;
        // zeroed once: a fault stays set (and makes later dataflow launches give up at once) until xh_fault_check
        // has seen it, so a second call enqueued behind a faulting one cannot wipe the evidence
        XH_HIP(ctx, hipMemsetAsync(ctx->d_fault, 0, 64, ctx->stream));
    }
    *d_word = ctx->d_fault;
    return XH_OK;
}

int xh_fault_collect(xh_ctx *ctx) {
    if (!ctx->d_fault) return XH_OK;
    XH_HIP(ctx, hipMemcpyAsync(ctx->h_fault, ctx->d_fault, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    ctx->fault_pending = true;
    return XH_OK;
}

int xh_fault_check(xh_ctx *ctx) {
    if (!ctx->fault_pending) return XH_OK;
    ctx->fault_pending = false;
    const unsigned code = *ctx->h_fault;
    *ctx->h_fault = 0;
    std::vector<xh_route_record> pending;
    pending.swap(ctx->pending_routes);
    if (!code) {
        for (const xh_route_record &r : pending) xh_route_confirm(r);
        return XH_OK;
    }
    // Either a bounded wait between routing units timed out (the device is shared and the units were not all resident)
    // or a guard of a PREPARED routing plan tripped (XH_FAULT_GUARD: a folded leaf that can fire after all, negative runoff or
    // initial storage, a negative outflow leaving a halo of pair units): the dataflow kernels of these calls left invalid
    // outputs.  Clear the word and route them again -- after a guard fault on the plan of pairs (still the dataflow kernel;
    // the prepared plan is given up), otherwise, or if that faults too, with one workgroup per network (no waits between
    // workgroups).
    XH_HIP(ctx, hipMemsetAsync(ctx->d_fault, 0, 64, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t seq_now = ctx->work_seq;
    const bool later_work = !pending.empty() && seq_now != pending.back().seq_after;
    bool pairs_first = code == XH_FAULT_GUARD;
    if (pairs_first) {
        fprintf(stderr, "[libxanthos_hip] a guard of the prepared routing plan tripped (a cell that cannot overdraw its channel by "
                "velocity * dt / length did: negative runoff or initial storage, other velocities than the plan was prepared "
                "for, a negative outflow beyond a halo): routing %zu call(s) again on the plan of pairs\n",
                pending.size());
        for (const xh_route_record &r : pending) {
            int rc = xh_route_rerun(ctx, r, true);
            if (rc) return rc;
        }
        XH_HIP(ctx, hipMemcpyAsync(ctx->h_fault, ctx->d_fault, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const unsigned again = *ctx->h_fault;
        *ctx->h_fault = 0;
        if (again) {
            XH_HIP(ctx, hipMemsetAsync(ctx->d_fault, 0, 64, ctx->stream));
            XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            pairs_first = false;
        }
    }
    if (!pairs_first) {
        fprintf(stderr, "[libxanthos_hip] routing fault %u (a bounded wait between routing units timed out: the device is "
                "shared and the units were not all resident); re-routing %zu call(s) with one workgroup per network\n",
                code, pending.size());
        {
            extern unsigned *xh_wave_last_place();
            unsigned w[16] = {0};
            if (xh_wave_last_place() && hipMemcpy(w, xh_wave_last_place(), sizeof(w), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[libxanthos_hip]   placement words: registered %u seconds %u firsts %u t3 %u decided4 %u c5 %u c6 %u t7 %u | leaders %u t10 %u displaced %u decided12 %u claimed13 %u\n",
                        w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[9], w[10], w[11], w[12], w[13]);
        }
        if (ctx->d_feed) {      // a fed call (xh_run_fused mode 1): how far the side stream had come
            unsigned w[48] = {0};
            if (hipMemcpy(w, ctx->d_feed, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[libxanthos_hip]   fed routing: months ready %u, placement epoch %u of %u\n", w[0], w[32],
                        ctx->feed_epoch);
                // a fed call whose routing kernel gave up waiting for data: most likely the side stream's kernels do not run
                // beside it here (kernels serialised by a profiler's counter passes or AMD_SERIALIZE_KERNEL) -- stage by
                // stage from now on instead of one 5 s wait per call after every back-off
                bool any_fed = false;
                for (const xh_route_record &r : pending) any_fed = any_fed || r.fed;
                if (code == 1 && any_fed) {
                    ctx->feed_disabled = true;
                    fprintf(stderr, "[libxanthos_hip]   the fed stage order is switched off for this context\n");
                }
            }
        }
        std::vector<xh_route_plan *> plans;
        for (const xh_route_record &r : pending) {
            int rc = xh_route_rerun(ctx, r, false);
            if (rc) return rc;
            ctx->reroutes += 1;
            if (std::find(plans.begin(), plans.end(), r.plan) == plans.end()) plans.push_back(r.plan);
        }
        for (xh_route_plan *p : plans) xh_route_backoff(p);
    }
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (pending.empty())
        return xh_fail(ctx, XH_ERR_DEVICE, "routing kernel fault %u and no record of the call: outputs are invalid", code);
    if (later_work)
        return xh_fail(ctx, XH_ERR_DEVICE, "routing kernel fault %u: the routing outputs were recomputed %s and are valid "
                       "now, but results of calls enqueued after xh_route_series read the invalid ones and must be recomputed",
                       code, pairs_first ? "by the dataflow kernel on the plan of pairs (a guard of the prepared plan had tripped)"
                                         : "with the workgroup-per-network kernel");
    return XH_OK;
}

int xh_settle(xh_ctx *ctx) {
    if (ctx->gather_pending) {      // a side gather nobody joined: it reads the pipeline's outputs, so it is part of "everything"
        XH_HIP(ctx, hipStreamSynchronize(ctx->side_stream[1]));
        ctx->gather_pending = false;
    }
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return xh_fault_check(ctx);
}

int xh_gather_stream(xh_ctx *ctx, hipStream_t *out) {
    if (!ctx->side_stream[1]) {
        int least = 0, greatest = 0;
        XH_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        XH_HIP(ctx, hipStreamCreateWithPriority(&ctx->side_stream[1], hipStreamNonBlocking, greatest != 0 ? greatest : least));
        XH_HIP(ctx, hipEventCreateWithFlags(&ctx->gather_event, hipEventDisableTiming));
    }
    *out = ctx->side_stream[1];
    return XH_OK;
}

// XH_ROCTX=1: every span also opens / closes a roctx range (bound at run time: libroctx64 of the ROCm on the box), so that a
// rocprofv3 --marker-trace timeline shows the stages by name around their kernels (tracing).
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
};
Roctx &roctx() {
    static Roctx r = [] {
        Roctx x;
        const char *env = getenv("XH_ROCTX");
        if (!env || env[0] != '1') return x;
        for (const char *n : {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                x.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                break;
            }
        }
        x.on = x.push && x.pop;
        return x;
    }();
    return r;
}
}  // namespace

xh_span xh_span_begin(xh_ctx *ctx, const char *name) { return xh_span_begin_on(ctx, name, ctx->stream); }

xh_span xh_span_begin_on(xh_ctx *ctx, const char *name, hipStream_t stream) {
    xh_span s{ctx, name};
    s.stream = stream;
    if (roctx().on) (void)roctx().push(name);
    ctx->work_seq += 1;
    // new work on the context's stream: the "runoff is final" event a fed call left for a side gather is no longer the
    // thing to wait for (xh_comm_gather_rows_side then orders itself behind the context's stream)
    if (stream == ctx->stream) ctx->runoff_event_fresh = false;
    if (!ctx->timing) return s;
    auto take = [&](hipEvent_t &e) {
        if (!ctx->event_pool.empty()) {
            e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
    };
    take(s.a);
    take(s.b);
    if (s.a && s.b) (void)hipEventRecord(s.a, stream);
    return s;
}

void xh_span_end(xh_span &s) {
    if (roctx().on) (void)roctx().pop();
    if (!s.ctx->timing || !s.a || !s.b) return;
    (void)hipEventRecord(s.b, s.stream);
    s.ctx->timers[s.name].pending.emplace_back(s.a, s.b);
}

void xh_span_cancel(xh_span &s) {      // nothing was launched after all: the span leaves no timing record
    if (roctx().on) (void)roctx().pop();
    if (s.a) s.ctx->event_pool.push_back(s.a);
    if (s.b) s.ctx->event_pool.push_back(s.b);
    s.a = s.b = nullptr;
}

extern "C" {

int xh_abi_version(void) { return 5; }

int xh_device_count(int *n) {
    if (!n) return XH_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return xh_fail(nullptr, XH_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return XH_OK;
}

int xh_ctx_create(int device, xh_ctx **out) {
    if (!out) return xh_fail(nullptr, XH_ERR_ARG, "xh_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return xh_fail(nullptr, XH_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return xh_fail(nullptr, XH_ERR_ARG, "device %d out of range [0,%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return xh_fail(nullptr, XH_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    xh_ctx *ctx = new xh_ctx();
    ctx->device = device;
    e = hipGetDeviceProperties(&ctx->prop, device);
    if (e != hipSuccess) {
        delete ctx;
        return xh_fail(nullptr, XH_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return xh_fail(nullptr, XH_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return XH_OK;
}

void xh_ctx_destroy(xh_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->timers)
        for (auto &p : kv.second.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 4; ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    for (int i = 0; i < 2; ++i)
        if (ctx->side_stream[i]) {
            (void)hipStreamSynchronize(ctx->side_stream[i]);
            (void)hipStreamDestroy(ctx->side_stream[i]);
        }
    for (auto e : ctx->side_events) (void)hipEventDestroy(e);
    if (ctx->gather_event) (void)hipEventDestroy(ctx->gather_event);
    if (ctx->d_feed) (void)hipFree(ctx->d_feed);
    if (ctx->d_fault) (void)hipFree(ctx->d_fault);
    if (ctx->h_fault) (void)hipHostFree(ctx->h_fault);
    if (ctx->io_ring) (void)hipHostFree(ctx->io_ring);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *xh_last_error(const xh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_xh_create_error.c_str(); }

int xh_device_name(xh_ctx *ctx, char *buf, size_t len) {
    if (!ctx || !buf || !len) return XH_ERR_ARG;
    snprintf(buf, len, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount);
    return XH_OK;
}

int xh_malloc(xh_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return XH_ERR_ARG;
    XH_HIP(ctx, hipSetDevice(ctx->device));
    XH_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 16));
    return XH_OK;
}

int xh_free(xh_ctx *ctx, void *d_ptr) {
    if (!ctx) return XH_ERR_ARG;
    if (!d_ptr) return XH_OK;
    // routing calls not yet confirmed fault-free keep raw pointers to their inputs and outputs (pending_routes): a
    // re-route must happen BEFORE any of that memory is released
    const int rc = xh_settle(ctx);
    XH_HIP(ctx, hipFree(d_ptr));
    return rc;
}

int xh_memcpy_h2d(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return XH_ERR_ARG;
    int rc = XH_OK;
    if (ctx->fault_pending) rc = xh_settle(ctx);      // the copy may overwrite the inputs of a call that must be re-routed
    ctx->work_seq += 1;
    XH_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host buffer may be reused on return
    return rc;
}

int xh_memcpy_d2h(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx || (bytes && (!h_dst || !d_src))) return XH_ERR_ARG;
    int rc = XH_OK;
    if (ctx->fault_pending) rc = xh_settle(ctx);      // settle routing calls in flight first: a re-route must precede the copy
    XH_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

int xh_host_alloc(xh_ctx *ctx, size_t bytes, void **h_ptr) {
    if (!ctx || !h_ptr) return XH_ERR_ARG;
    XH_HIP(ctx, hipSetDevice(ctx->device));
    XH_HIP(ctx, hipHostMalloc(h_ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return XH_OK;
}

int xh_host_free(xh_ctx *ctx, void *h_ptr) {
    if (!ctx) return XH_ERR_ARG;
    if (!h_ptr) return XH_OK;
    const int rc = xh_settle(ctx);
    XH_HIP(ctx, hipHostFree(h_ptr));
    return rc;
}

int xh_memcpy_h2d_async(xh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return XH_ERR_ARG;
    ctx->work_seq += 1;      // work that may read the outputs of a routing call still to be confirmed (xh_fault_check)
    XH_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return XH_OK;
}

int xh_memcpy_d2h_async(xh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx || (bytes && (!h_dst || !d_src))) return XH_ERR_ARG;
    ctx->work_seq += 1;      // work that may read the outputs of a routing call still to be confirmed (xh_fault_check)
    XH_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return XH_OK;
}

int xh_memcpy_d2d(xh_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return XH_ERR_ARG;
    ctx->work_seq += 1;      // work that may read the outputs of a routing call still to be confirmed (xh_fault_check)
    XH_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return XH_OK;
}

int xh_memset(xh_ctx *ctx, void *d_ptr, int value, size_t bytes) {
    if (!ctx || (bytes && !d_ptr)) return XH_ERR_ARG;
    ctx->work_seq += 1;
    XH_HIP(ctx, hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return XH_OK;
}

int xh_sync(xh_ctx *ctx) {
    if (!ctx) return XH_ERR_ARG;
    return xh_settle(ctx);
}

int xh_timing_reset(xh_ctx *ctx) {
    if (!ctx) return XH_ERR_ARG;
    const int rc_settle = xh_settle(ctx);
    if (rc_settle) return rc_settle;
    for (auto &kv : ctx->timers) {
        for (auto &p : kv.second.pending) {
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
        kv.second.done_ms = 0.0;
        kv.second.launches = 0;
    }
    return XH_OK;
}

int xh_timing_enable(xh_ctx *ctx, int on) {
    if (!ctx) return XH_ERR_ARG;
    ctx->timing = on != 0;
    return XH_OK;
}

int xh_timing_get(xh_ctx *ctx, const char *name, double *total_ms, int64_t *launches) {
    if (!ctx || !name) return XH_ERR_ARG;
    const int rc_settle = xh_settle(ctx);
    if (rc_settle) return rc_settle;
    auto it = ctx->timers.find(name);
    double ms = 0.0;
    int64_t n = 0;
    if (it != ctx->timers.end()) {
        xh_timer_slot &s = it->second;
        for (auto &p : s.pending) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) {
                s.done_ms += t;
                s.launches += 1;
            }
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
        s.pending.clear();
        ms = s.done_ms;
        n = s.launches;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return XH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ row movers
// One workgroup per row chunk; lanes run along the row so both sides are coalesced 16-B accesses.
__global__ void __launch_bounds__(256) k_gather_rows(const double *__restrict__ src, const int64_t *__restrict__ rows,
                                                     int64_t nrows, int64_t ncols, double *__restrict__ dst,
                                                     int scatter) {
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int64_t g = rows[r];
        const double *s = scatter ? src + r * ncols : src + g * ncols;
        double *d = scatter ? dst + g * ncols : dst + r * ncols;
        for (int64_t c = threadIdx.x; c < ncols; c += blockDim.x) d[c] = s[c];
    }
}

// 32x32 LDS tile transpose (+1 padding column: conflict-free column reads of 8-byte elements)
__global__ void __launch_bounds__(256) k_transpose(const double *__restrict__ src, int64_t rows, int64_t cols,
                                                   double *__restrict__ dst) {
    __shared__ double tile[32][33];
    const int64_t tr = (int64_t)blockIdx.y * 32, tc = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = tr + j, c = tc + tx;
        if (r < rows && c < cols) tile[j][tx] = src[r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = tc + j, r = tr + tx;
        if (r < rows && c < cols) dst[c * rows + r] = tile[tx][j];
    }
}

extern "C" {

}  // extern "C"

int xh_move_rows_on(xh_ctx *ctx, hipStream_t st, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                    double *d_dst, int scatter) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_src && d_rows && d_dst && nrows >= 0 && ncols >= 0, "xh_gather/scatter_rows: bad argument");
    if (nrows == 0 || ncols == 0) return XH_OK;
    ctx->work_seq += 1;
    int grid = (int)(nrows < 65536 ? nrows : 65536);
    hipLaunchKernelGGL(k_gather_rows, dim3(grid), dim3(256), 0, st, d_src, d_rows, nrows, ncols, d_dst, scatter);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" {

static int move_rows(xh_ctx *ctx, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                     double *d_dst, int scatter) {
    if (!ctx) return XH_ERR_ARG;
    return xh_move_rows_on(ctx, ctx->stream, d_src, d_rows, nrows, ncols, d_dst, scatter);
}

// A named HIP-event span on the context's stream, for callers that want a device time of their own next to the library's
// kernel timers (xh_timing_get): e.g. what a step spends in the write-out gather after the routing has ended.
int xh_mark_begin(xh_ctx *ctx, const char *name) {
    if (!ctx || !name) return XH_ERR_ARG;
    XH_REQUIRE(ctx, !ctx->mark_open, "xh_mark_begin: a mark is already open");
    ctx->mark_name = name;
    xh_span s = xh_span_begin(ctx, ctx->mark_name.c_str());
    ctx->mark_a = s.a;
    ctx->mark_b = s.b;
    ctx->mark_open = true;
    return XH_OK;
}

int xh_mark_end(xh_ctx *ctx) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, ctx->mark_open, "xh_mark_end: no mark is open");
    xh_span s{ctx, ctx->mark_name.c_str()};
    s.a = ctx->mark_a;
    s.b = ctx->mark_b;
    s.stream = ctx->stream;
    xh_span_end(s);
    ctx->mark_open = false;
    return XH_OK;
}

int xh_gather_rows(xh_ctx *ctx, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                   double *d_dst) {
    return move_rows(ctx, d_src, d_rows, nrows, ncols, d_dst, 0);
}

int xh_scatter_rows(xh_ctx *ctx, const double *d_src, const int64_t *d_rows, int64_t nrows, int64_t ncols,
                    double *d_dst) {
    return move_rows(ctx, d_src, d_rows, nrows, ncols, d_dst, 1);
}

int xh_transpose(xh_ctx *ctx, const double *d_src, int64_t rows, int64_t cols, double *d_dst) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_src && d_dst && rows >= 0 && cols >= 0, "xh_transpose: bad argument");
    if (rows == 0 || cols == 0) return XH_OK;
    ctx->work_seq += 1;
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    XH_REQUIRE(ctx, grid.y <= 65535u, "xh_transpose: too many rows");
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, ctx->stream, d_src, rows, cols, d_dst);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

}  // extern "C"
