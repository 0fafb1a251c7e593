// TEST-ONLY stand-in for librccl.so.1 (never shipped, never linked into libxanthos_hip.so).
//
// RCCL refuses two ranks on one device, and the boxes this repository is tested on have ONE MI355X.  So that
// xh_comm_gather_rows (csrc/xh_comm.hip) -- its root-side staging offsets, the three scatter ranges around the root's
// own rows, root != 0, empty shards -- can still run with nranks > 1, the GPU tests start their rank processes with
// this directory first on LD_LIBRARY_PATH: libxanthos_hip.so binds RCCL at run time with dlopen("librccl.so.1") and
// then finds this library.  It implements exactly the eight entry points xh_comm.hip uses, with the semantics the
// library relies on: point-to-point sends and receives between ranks of one communicator, matched in issue order per
// (source, destination) pair, grouped between ncclGroupStart / ncclGroupEnd, ordered after the work already on the
// stream.  Transport: a host bounce through files in /dev/shm (written under a temporary name and renamed, so a
// receiver never sees a partial message); everything happens synchronously inside ncclGroupEnd.  Slow and simple on
// purpose: the thing under test is xh_comm.hip's bookkeeping, not a transport.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
    std::string id;
    int nranks = 0, rank = 0;
    std::map<int, unsigned> sent, received;      // messages so far per peer
};

namespace {

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    ncclComm *comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::string msg_path(const ncclComm *c, int src, int dst, unsigned seq) {
    // XH_FAKE_RCCL_DIR: where the messages travel (default /dev/shm; a test that moves gigabytes points it at a disk)
    const char *dir = getenv("XH_FAKE_RCCL_DIR");
    char b[512];
    snprintf(b, sizeof(b), "%s/xh_fake_rccl_%s_%d_%d_%u", (dir && dir[0]) ? dir : "/dev/shm", c->id.c_str(), src, dst, seq);
    return b;
}

ncclResult_t run(const Op &op) {
    ncclComm *c = op.comm;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> host(op.bytes);
    if (op.send) {
        if (hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        const std::string path = msg_path(c, c->rank, op.peer, c->sent[op.peer]++), tmp = path + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return ncclSystemError;
        const size_t w = fwrite(host.data(), 1, op.bytes, f);
        fclose(f);
        if (w != op.bytes || rename(tmp.c_str(), path.c_str()) != 0) return ncclSystemError;
        return ncclSuccess;
    }
    const std::string path = msg_path(c, op.peer, c->rank, c->received[op.peer]++);
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ncclSystemError;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    if ((size_t)st.st_size != op.bytes) {      // the sizes of a send and its receive must agree, as in RCCL
        unlink(path.c_str());
        return ncclInvalidArgument;
    }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return ncclSystemError;
    const size_t r = fread(host.data(), 1, op.bytes, f);
    fclose(f);
    unlink(path.c_str());
    if (r != op.bytes) return ncclSystemError;
    if (hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t flush() {
    ncclResult_t first = ncclSuccess;
    // sends first: every rank's sends complete without waiting for anybody, so receives cannot deadlock
    for (int pass = 0; pass < 2; ++pass)
        for (const Op &op : g_ops)
            if (op.send == (pass == 0)) {
                const ncclResult_t r = run(op);
                if (r != ncclSuccess && first == ncclSuccess) first = r;
            }
    g_ops.clear();
    return first;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{send, buf, count * type_bytes(t), peer, comm, st});
    return g_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof(id->internal), "FAKE%lx_%lx", (long)getpid(), (long)now);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || strncmp(id.internal, "FAKE", 4) != 0) return ncclInvalidArgument;
    ncclComm *c = new ncclComm();
    id.internal[sizeof(id.internal) - 1] = 0;
    c->id = id.internal;
    c->nranks = nranks;
    c->rank = rank;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    return post(true, const_cast<void *>(buf), count, t, peer, comm, st);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    return post(false, buf, count, t, peer, comm, st);
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake rccl)";
        case ncclUnhandledCudaError: return "HIP error (fake rccl)";
        case ncclSystemError: return "system error or 120 s without the matching send (fake rccl)";
        case ncclInvalidArgument: return "invalid argument or send / receive size mismatch (fake rccl)";
        case ncclInvalidUsage: return "invalid usage (fake rccl)";
        default: return "error (fake rccl)";
    }
}

}  // extern "C"
