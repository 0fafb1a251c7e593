// File <-> HBM movers of the loader / writer either side of the hot path.
//
// The reference loads every forcing array with np.load and keeps it on the host (data_load.py:186-195, :342-350); at the
// full grid that is 8 x 324 MB which then has to cross PCIe.  xh_upload_file reads a byte range of a file (the body of a
// .npy) with a few host threads, each pread()-ing 8 MiB chunks into its own page-locked slots and sending them on with
// asynchronous copies on its own stream, so the page-cache reads and the PCIe transfers overlap and no pageable 324 MB
// intermediate exists.  xh_download_file(s) is the mirror for the writer (data_writer/out_writer.py: np.save of the
// outputs): one thread per file, copy of chunk i + 1 under the write of chunk i.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <thread>

#include "xh_common.h"

namespace {

constexpr size_t IO_CHUNK = 8u << 20;
constexpr int IO_SLOTS = 2;            // per thread
constexpr int IO_MAX_THREADS = 16;

int io_ring(xh_ctx *ctx, int threads) {
    const size_t need = (size_t)threads * IO_SLOTS * IO_CHUNK;
    if (ctx->io_ring_bytes >= need) return XH_OK;
    if (ctx->io_ring) XH_HIP(ctx, hipHostFree(ctx->io_ring));
    ctx->io_ring = nullptr;
    ctx->io_ring_bytes = 0;
    XH_HIP(ctx, hipHostMalloc(&ctx->io_ring, need, hipHostMallocDefault));
    ctx->io_ring_bytes = need;
    return XH_OK;
}

int io_threads(int threads) {
    if (threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        threads = hw ? (int)hw : 4;
        if (threads > 8) threads = 8;
    }
    return threads > IO_MAX_THREADS ? IO_MAX_THREADS : threads;
}

// One worker: chunks i = next++ of the range; `up` = file -> device, else device -> file.
void io_worker(int device, int fd, uint64_t offset, char *dev, size_t bytes, char *slots, std::atomic<size_t> *next,
               std::atomic<int> *err, bool up) {
    hipStream_t st = nullptr;
    hipEvent_t ev[IO_SLOTS] = {};
    size_t pend_off[IO_SLOTS] = {}, pend_len[IO_SLOTS] = {};
    bool used[IO_SLOTS] = {};
    auto fail = [&](int code) {
        int z = 0;
        err->compare_exchange_strong(z, code);
    };
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        fail(-1);
        return;
    }
    for (int s = 0; s < IO_SLOTS; ++s)
        if (hipEventCreateWithFlags(&ev[s], hipEventDisableTiming) != hipSuccess) fail(-1);
    const size_t nchunks = (bytes + IO_CHUNK - 1) / IO_CHUNK;
    // device -> file: the slot's chunk is written out once its copy has landed
    auto drain = [&](int s) {
        if (!used[s]) return;
        if (hipEventSynchronize(ev[s]) != hipSuccess) fail(-1);
        if (!up) {
            size_t done = 0;
            while (done < pend_len[s] && err->load() == 0) {
                const ssize_t w = pwrite(fd, slots + (size_t)s * IO_CHUNK + done, pend_len[s] - done,
                                         (off_t)(offset + pend_off[s] + done));
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) {
                    fail(errno ? errno : EIO);
                    break;
                }
                done += (size_t)w;
            }
        }
        used[s] = false;
    };
    int s = 0;
    while (err->load() == 0) {
        const size_t i = next->fetch_add(1);
        if (i >= nchunks) break;
        const size_t off = i * IO_CHUNK, len = bytes - off < IO_CHUNK ? bytes - off : IO_CHUNK;
        char *slot = slots + (size_t)s * IO_CHUNK;
        drain(s);
        if (up) {
            size_t done = 0;
            while (done < len) {
                const ssize_t r = pread(fd, slot + done, len - done, (off_t)(offset + off + done));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) {
                    fail(r == 0 ? ENODATA : errno);
                    break;
                }
                done += (size_t)r;
            }
            if (done < len) break;
            if (hipMemcpyAsync(dev + off, slot, len, hipMemcpyHostToDevice, st) != hipSuccess) fail(-1);
        } else {
            if (hipMemcpyAsync(slot, dev + off, len, hipMemcpyDeviceToHost, st) != hipSuccess) fail(-1);
        }
        if (hipEventRecord(ev[s], st) != hipSuccess) fail(-1);
        used[s] = true;
        pend_off[s] = off;
        pend_len[s] = len;
        s = (s + 1) % IO_SLOTS;
    }
    for (int k = 0; k < IO_SLOTS; ++k) drain((s + k) % IO_SLOTS);
    if (hipStreamSynchronize(st) != hipSuccess) fail(-1);
    for (int k = 0; k < IO_SLOTS; ++k)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    (void)hipStreamDestroy(st);
}

struct IoJob {
    void *dev;
    const char *path;
    uint64_t offset;
    size_t bytes;
};

// Upload: ONE job, its chunks dealt to `threads` workers.  Download: one worker per job (file) -- buffered writes to one
// file are serialised by the file system (one inode lock), so a second writer on the same file only adds contention,
// but different files do run side by side; within a file the copy of chunk i + 1 overlaps the write of chunk i.
int io_run(xh_ctx *ctx, const IoJob *jobs, int njobs, int threads, bool up) {
    if (!ctx || njobs < 0 || (njobs && !jobs)) return XH_ERR_ARG;
    for (int j = 0; j < njobs; ++j)
        if (!jobs[j].path || (jobs[j].bytes && !jobs[j].dev)) return XH_ERR_ARG;
    if (njobs > IO_MAX_THREADS) return xh_fail(ctx, XH_ERR_LIMIT, "at most %d files per call", IO_MAX_THREADS);
    // earlier work on the context's stream may still read (upload) or write (download) the device ranges; a routing call
    // that has to be re-run is re-run here
    const int rc = xh_settle(ctx);
    if (rc != XH_OK && rc != XH_ERR_DEVICE) return rc;
    ctx->work_seq += 1;
    if (njobs == 0 || (up && jobs[0].bytes == 0)) return rc;
    int fds[IO_MAX_THREADS];
    auto close_all = [&](int upto) {
        int bad = 0;
        for (int j = 0; j < upto; ++j)
            if (close(fds[j]) != 0 && !up) bad = errno ? errno : EIO;
        return bad;
    };
    for (int j = 0; j < njobs; ++j) {
        fds[j] = up ? open(jobs[j].path, O_RDONLY) : open(jobs[j].path, O_WRONLY | O_CREAT, 0644);
        if (fds[j] < 0) {
            const int e = errno;
            close_all(j);
            return xh_fail(ctx, XH_ERR_ARG, "%s: %s", jobs[j].path, strerror(e));
        }
        struct stat sb;
        if (up && (fstat(fds[j], &sb) != 0 || (uint64_t)sb.st_size < jobs[j].offset + jobs[j].bytes)) {
            close_all(j + 1);
            return xh_fail(ctx, XH_ERR_ARG, "%s: shorter than offset %llu + %zu bytes", jobs[j].path,
                           (unsigned long long)jobs[j].offset, jobs[j].bytes);
        }
    }
    if (up) {
        threads = io_threads(threads);
        const size_t nchunks = (jobs[0].bytes + IO_CHUNK - 1) / IO_CHUNK;
        if ((size_t)threads > nchunks) threads = nchunks ? (int)nchunks : 1;
    } else {
        threads = njobs;
    }
    const int rr = io_ring(ctx, threads);
    if (rr != XH_OK) {
        close_all(njobs);
        return rr;
    }
    std::atomic<size_t> next[IO_MAX_THREADS];
    for (auto &x : next) x.store(0);
    std::atomic<int> err{0};
    std::thread pool[IO_MAX_THREADS];
    for (int t = 0; t < threads; ++t) {
        const IoJob &job = jobs[up ? 0 : t];
        pool[t] = std::thread(io_worker, ctx->device, fds[up ? 0 : t], job.offset, (char *)job.dev, job.bytes,
                              (char *)ctx->io_ring + (size_t)t * IO_SLOTS * IO_CHUNK, &next[up ? 0 : t], &err, up);
    }
    for (int t = 0; t < threads; ++t) pool[t].join();
    const int cerr = close_all(njobs);
    if (cerr && err.load() == 0) err.store(cerr);
    if (err.load() > 0) return xh_fail(ctx, XH_ERR_ARG, "%s: %s", jobs[0].path, strerror(err.load()));
    if (err.load() < 0) return xh_fail(ctx, XH_ERR_HIP, "%s: a copy of the file transfer failed", jobs[0].path);
    return rc;
}

}  // namespace

extern "C" {

int xh_upload_file(xh_ctx *ctx, void *d_dst, const char *path, uint64_t offset, size_t bytes, int threads) {
    const IoJob job{d_dst, path, offset, bytes};
    return io_run(ctx, &job, 1, threads, true);
}

int xh_download_file(xh_ctx *ctx, const void *d_src, const char *path, uint64_t offset, size_t bytes, int threads) {
    const IoJob job{const_cast<void *>(d_src), path, offset, bytes};
    return io_run(ctx, &job, 1, threads, false);
}

int xh_download_files(xh_ctx *ctx, int n, const void *const *d_srcs, const char *const *paths, const uint64_t *offsets,
                      const size_t *bytes) {
    if (n < 0 || n > IO_MAX_THREADS || (n && (!d_srcs || !paths || !offsets || !bytes))) return XH_ERR_ARG;
    IoJob jobs[IO_MAX_THREADS];
    for (int j = 0; j < n; ++j) jobs[j] = IoJob{const_cast<void *>(d_srcs[j]), paths[j], offsets[j], bytes[j]};
    return io_run(ctx, jobs, n, 0, false);
}

}  // extern "C"
