// Multi-GPU write-out: ONE gather of every rank's output rows to the root over RCCL (xGMI inside a node).
//
// The reference has no distributed path (its only parallelism is a joblib thread pool over basin chunks,
// abcd.py:357-391).  BASELINE's north star partitions the 235 basins over the GPUs of a node "with a single RCCL
// gather over xGMI at write-out": each rank keeps its shard's [n_local, nmonths] outputs in HBM and, once per run,
// sends them to the root, which drops every row at its place in grid order.
//
// The gather is a grouped ncclSend / ncclRecv of the EXACT shard sizes (RCCL has no gather-v; no padding to the
// largest shard, no staging copy on the senders: the send buffers are the pipeline's own output arrays), enqueued on
// the context's stream like every kernel of the library.  xGMI is point to point: the root receives from its N - 1
// peers over N - 1 different links at once, so the gather is bound by the root's HBM write rate, not by one link.
// On the root the received blocks land rank-major in a staging area and one row-scatter kernel per variable moves
// them (and the root's own rows, straight from its output arrays) to grid order.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the soname PyTorch-ROCm also loads, so a process that already
// runs torch.distributed shares one RCCL): single-GPU users of libxanthos_hip.so do not need it at all.
#include <dlfcn.h>

#include <cstdlib>

#include <mutex>
#include <rccl/rccl.h>

#include "xh_common.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

void rccl_load(RcclApi &api);

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(api); });
    return api;
}

void rccl_load(RcclApi &api) {
    // XH_RCCL_LIBRARY: full path of the RCCL build to bind (a site's own build; the test-only stand-in of tests/fake_rccl in a
    // process that has PyTorch's bundled copy loaded under the same soname, where the name alone would find that one)
    const char *names[] = {getenv("XH_RCCL_LIBRARY"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        if (!n || !n[0]) continue;
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) {
        api.error = std::string("cannot load librccl.so.1: ") + dlerror();
        return;
    }
    auto sym = [&](const char *name) {
        void *p = dlsym(api.handle, name);
        if (!p && api.error.empty()) api.error = std::string("librccl has no symbol ") + name;
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
}

}  // namespace

struct xh_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    void *d_stage = nullptr;      // root: received blocks, rank-major
    size_t stage_bytes = 0;
    bool self_loop = false;       // XH_COMM_SELF_LOOP=1 at creation: the root's own rows travel through ncclSend / ncclRecv too
    int64_t n_send = 0, n_recv = 0;
};

#define XH_NCCL(ctx, call)                                                                                     \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return xh_fail((ctx), XH_ERR_HIP, "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r_), __FILE__, \
                           __LINE__);                                                                          \
    } while (0)

extern "C" {

int xh_comm_unique_id(char *id, size_t len) {
    if (!id || len < sizeof(ncclUniqueId)) return xh_fail(nullptr, XH_ERR_ARG, "xh_comm_unique_id: need %zu bytes",
                                                          sizeof(ncclUniqueId));
    RcclApi &api = rccl();
    if (!api.error.empty()) return xh_fail(nullptr, XH_ERR_HIP, "%s", api.error.c_str());
    ncclUniqueId u;
    ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) return xh_fail(nullptr, XH_ERR_HIP, "ncclGetUniqueId: %s", api.GetErrorString(r));
    memcpy(id, &u, sizeof(u));
    return XH_OK;
}

int xh_comm_create(xh_ctx *ctx, int32_t nranks, int32_t rank, const char *id, size_t len, xh_comm **out) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, out && id && len >= sizeof(ncclUniqueId) && nranks >= 1 && rank >= 0 && rank < nranks,
               "xh_comm_create: bad argument");
    *out = nullptr;
    RcclApi &api = rccl();
    if (!api.error.empty()) return xh_fail(ctx, XH_ERR_HIP, "%s", api.error.c_str());
    XH_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    xh_comm *c = new xh_comm();
    c->nranks = nranks;
    c->rank = rank;
    ncclResult_t r = api.CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return xh_fail(ctx, XH_ERR_HIP, "ncclCommInitRank(%d of %d) failed: %s", rank, nranks, api.GetErrorString(r));
    }
    c->self_loop = getenv("XH_COMM_SELF_LOOP") && getenv("XH_COMM_SELF_LOOP")[0] == '1';
    *out = c;
    return XH_OK;
}

int xh_comm_info(const xh_comm *c, int64_t info[4]) {
    if (!c || !info) return XH_ERR_ARG;
    info[0] = c->nranks;
    info[1] = c->rank;
    info[2] = c->n_send;
    info[3] = c->n_recv;
    return XH_OK;
}

void xh_comm_destroy(xh_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->d_stage) (void)hipFree(c->d_stage);
    delete c;
}

static int gather_rows_on(xh_ctx *ctx, hipStream_t st, xh_comm *c, int32_t root, int32_t nvar, const double *const *h_d_local,
                          int64_t ncols, const int64_t *h_counts, const int64_t *d_perm, double *const *h_d_out);

int xh_comm_gather_rows(xh_ctx *ctx, xh_comm *c, int32_t root, int32_t nvar, const double *const *h_d_local,
                        int64_t ncols, const int64_t *h_counts, const int64_t *d_perm, double *const *h_d_out) {
    if (!ctx || !c) return XH_ERR_ARG;
    return gather_rows_on(ctx, ctx->stream, c, root, nvar, h_d_local, ncols, h_counts, d_perm, h_d_out);
}

// The same gather on the context's GATHER stream (a queue of its own), so that it runs beside whatever the context's stream
// does next -- the routing kernel, which needs none of PET / AET / Q / Sav's bytes to move.  Ordered behind the kernels that
// produced the arrays: behind everything enqueued on the context's stream so far, or -- after an xh_run_fused in mode 1,
// whose runoff is completed by a side stream while the routing kernel already sits in the context's stream -- behind that
// side stream's last kernel.  xh_comm_join (or any synchronising call) orders the context's stream behind the gather.
// Use a communicator of its own for this stream (RCCL serialises the operations of one communicator).
int xh_comm_gather_rows_side(xh_ctx *ctx, xh_comm *c, int32_t root, int32_t nvar, const double *const *h_d_local,
                             int64_t ncols, const int64_t *h_counts, const int64_t *d_perm, double *const *h_d_out) {
    if (!ctx || !c) return XH_ERR_ARG;
    hipStream_t g = nullptr;
    int rc = xh_gather_stream(ctx, &g);
    if (rc) return rc;
    // (the event of a fed call stands for "PET / AET / Q / Sav are final" only while nothing else has been enqueued on the
    // context since -- kernels, copies, row movers all bump work_seq; otherwise order behind the context's stream)
    if (ctx->runoff_event_fresh && ctx->runoff_event && ctx->work_seq == ctx->runoff_seq) {
        XH_HIP(ctx, hipStreamWaitEvent(g, ctx->runoff_event, 0));
        ctx->runoff_event_fresh = false;
    } else {
        XH_HIP(ctx, hipEventRecord(ctx->gather_event, ctx->stream));
        XH_HIP(ctx, hipStreamWaitEvent(g, ctx->gather_event, 0));
    }
    ctx->gather_pending = true;
    // A side gather that moves none of the routing's outputs is not "work behind the routing call" for the settling of a
    // routing fault (xh_fault_check): the re-routed call leaves what it gathered untouched.
    const uint64_t seq_before = ctx->work_seq;
    bool routed_arrays = false;
    if (!ctx->pending_routes.empty() && h_d_local) {
        const xh_route_record &pr = ctx->pending_routes.back();
        for (int v = 0; v < nvar; ++v)
            routed_arrays = routed_arrays || (h_d_local[v] && (h_d_local[v] == pr.chs || h_d_local[v] == pr.avg ||
                                                                h_d_local[v] == pr.S_end || h_d_local[v] == pr.F_end));
    }
    rc = gather_rows_on(ctx, g, c, root, nvar, h_d_local, ncols, h_counts, d_perm, h_d_out);
    if (!routed_arrays && !ctx->pending_routes.empty() && ctx->pending_routes.back().seq_after == seq_before)
        ctx->pending_routes.back().seq_after = ctx->work_seq;
    XH_HIP(ctx, hipEventRecord(ctx->gather_event, g));
    return rc;
}

int xh_comm_join(xh_ctx *ctx) {
    if (!ctx) return XH_ERR_ARG;
    if (!ctx->gather_pending) return XH_OK;
    XH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gather_event, 0));
    ctx->gather_pending = false;
    return XH_OK;
}

static int gather_rows_on(xh_ctx *ctx, hipStream_t st, xh_comm *c, int32_t root, int32_t nvar, const double *const *h_d_local,
                          int64_t ncols, const int64_t *h_counts, const int64_t *d_perm, double *const *h_d_out) {
    XH_REQUIRE(ctx, root >= 0 && root < c->nranks && nvar > 0 && h_d_local && h_counts && ncols > 0,
               "xh_comm_gather_rows: bad argument");
    RcclApi &api = rccl();
    const int64_t n_local = h_counts[c->rank];
    XH_REQUIRE(ctx, n_local >= 0, "xh_comm_gather_rows: negative count");
    for (int v = 0; v < nvar; ++v) XH_REQUIRE(ctx, h_d_local[v] || n_local == 0, "xh_comm_gather_rows: NULL local array");
    ctx->work_seq += 1;      // reads outputs a routing call may still have to recompute (xh_fault_check)
    // Inside a group the first failure is remembered and the group is still closed: returning between GroupStart and
    // GroupEnd would leave the group open on this thread and break every later RCCL call.
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    auto note = [&](ncclResult_t r, const char *name) {
        if (r != ncclSuccess && first == ncclSuccess) {
            first = r;
            what = name;
        }
    };
    if (c->rank != root) {
        if (n_local == 0) return XH_OK;
        XH_NCCL(ctx, api.GroupStart());
        for (int v = 0; v < nvar && first == ncclSuccess; ++v) {
            note(api.Send(h_d_local[v], (size_t)(n_local * ncols), ncclDouble, root, c->comm, st), "ncclSend");
            c->n_send += 1;
        }
        note(api.GroupEnd(), "ncclGroupEnd");
        if (first != ncclSuccess)
            return xh_fail(ctx, XH_ERR_HIP, "xh_comm_gather_rows: %s failed: %s", what, api.GetErrorString(first));
        return XH_OK;
    }
    XH_REQUIRE(ctx, d_perm && h_d_out, "xh_comm_gather_rows: the root needs d_perm and the output arrays");
    // self-loop (testing, XH_COMM_SELF_LOOP=1): the root's own rows are sent to and received from its own rank inside the group,
    // i.e. they count as remote rows and land in the staging area like everybody else's
    const bool loop = c->self_loop && n_local > 0;
    int64_t remote = 0, before_me = 0;
    for (int r = 0; r < c->nranks; ++r) {
        XH_REQUIRE(ctx, h_counts[r] >= 0, "xh_comm_gather_rows: negative count");
        if (r != root || loop) remote += h_counts[r];
        if (r < root) before_me += h_counts[r];
    }
    const size_t need = (size_t)remote * ncols * nvar * sizeof(double);
    int rc_device = XH_OK;      // XH_ERR_DEVICE of the settle below, returned once the gather has been enqueued
    if (need > c->stage_bytes) {
        const int rcs = xh_settle(ctx);
        if (rcs && rcs != XH_ERR_DEVICE) return rcs;
        rc_device = rcs;
        if (c->d_stage) XH_HIP(ctx, hipFree(c->d_stage));
        c->d_stage = nullptr;
        c->stage_bytes = 0;
        XH_HIP(ctx, hipMalloc(&c->d_stage, need ? need : 16));
        c->stage_bytes = need;
    }
    double *stage = static_cast<double *>(c->d_stage);
    // staging layout: [variable][remote ranks in rank order][rows][ncols]
    if (remote > 0) {
        XH_NCCL(ctx, api.GroupStart());
        for (int v = 0; v < nvar && first == ncclSuccess; ++v) {
            int64_t off = 0;
            for (int r = 0; r < c->nranks && first == ncclSuccess; ++r) {
                if ((r == root && !loop) || h_counts[r] == 0) continue;
                note(api.Recv(stage + ((int64_t)v * remote + off) * ncols, (size_t)(h_counts[r] * ncols), ncclDouble, r,
                              c->comm, st), "ncclRecv");
                c->n_recv += 1;
                off += h_counts[r];
            }
            if (loop && first == ncclSuccess) {
                note(api.Send(h_d_local[v], (size_t)(n_local * ncols), ncclDouble, root, c->comm, st), "ncclSend");
                c->n_send += 1;
            }
        }
        note(api.GroupEnd(), "ncclGroupEnd");
        if (first != ncclSuccess)
            return xh_fail(ctx, XH_ERR_HIP, "xh_comm_gather_rows: %s failed: %s", what, api.GetErrorString(first));
    }
    // d_perm is rank-major over ALL ranks: rows [0, before_me) and [before_me + n_local, total) are remote
    for (int v = 0; v < nvar; ++v) {
        XH_REQUIRE(ctx, h_d_out[v], "xh_comm_gather_rows: NULL output array");
        int rc;
        if (loop) {      // every row, the root's own included, sits in the staging area in rank order = the order of d_perm
            rc = xh_move_rows_on(ctx, st, stage + (int64_t)v * remote * ncols, d_perm, remote, ncols, h_d_out[v], 1);
            if (rc) return rc;
            continue;
        }
        if (before_me > 0) {
            rc = xh_move_rows_on(ctx, st, stage + (int64_t)v * remote * ncols, d_perm, before_me, ncols, h_d_out[v], 1);
            if (rc) return rc;
        }
        if (n_local > 0) {
            rc = xh_move_rows_on(ctx, st, h_d_local[v], d_perm + before_me, n_local, ncols, h_d_out[v], 1);
            if (rc) return rc;
        }
        if (remote - before_me > 0) {
            rc = xh_move_rows_on(ctx, st, stage + ((int64_t)v * remote + before_me) * ncols, d_perm + before_me + n_local,
                                 remote - before_me, ncols, h_d_out[v], 1);
            if (rc) return rc;
        }
    }
    return rc_device;
}

}  // extern "C"
