// File <-> HBM movers of the loader / writer either side of the hot path.
//
// The reference loads every forcing array with np.load and keeps it on the host (data_load.py:186-195, :342-350); at the
// full grid that is 8 x 324 MB which then has to cross PCIe.  Measured on the MI355X box (tools/io_experiment.py, 4 x 324 MB
// from the page cache): read() into a host array + copy 8 GB/s; threads pread()-ing into page-locked slots + asynchronous
// copies 5-8 GB/s (the page-cache copy is the bottleneck, whatever the number of threads); a copy straight out of a
// read-only MAPPING of the file 34 GB/s (the runtime pins the page-cache pages and the DMA engines read them in place).
// So xh_upload_file maps the byte range and hands the mapping to hipMemcpy -- no host copy of the data at all.
// The other direction is the opposite: a copy into a shared mapping of a fresh file runs at 2.8 GB/s (a page fault per
// 4 KB), device -> pageable array -> write() at 6 GB/s, and xh_download_file(s) -- one thread per file, device -> page-locked
// slot i + 1 under the write() of slot i -- at 15 GB/s.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <thread>

#include "xh_common.h"

namespace {

constexpr size_t IO_CHUNK = 8u << 20;
constexpr int IO_SLOTS = 2;            // per writer thread
constexpr int IO_MAX_FILES = 16;

int io_ring(xh_ctx *ctx, int writers) {
    const size_t need = (size_t)writers * IO_SLOTS * IO_CHUNK;
    if (ctx->io_ring_bytes >= need) return XH_OK;
    if (ctx->io_ring) XH_HIP(ctx, hipHostFree(ctx->io_ring));
    ctx->io_ring = nullptr;
    ctx->io_ring_bytes = 0;
    XH_HIP(ctx, hipHostMalloc(&ctx->io_ring, need, hipHostMallocDefault));
    ctx->io_ring_bytes = need;
    return XH_OK;
}

struct IoJob {
    const char *dev;
    const char *path;
    uint64_t offset;
    size_t bytes;
};

// One writer: the chunks of one device array, in order, into one file.  The copy of chunk i + 1 into the other slot is in
// flight while chunk i is written.
void io_writer(int device, int fd, IoJob job, char *slots, std::atomic<int> *err) {
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
    auto drain = [&](int s) {      // the slot's chunk is written out once its copy has landed
        if (!used[s]) return;
        if (hipEventSynchronize(ev[s]) != hipSuccess) fail(-1);
        size_t done = 0;
        while (done < pend_len[s] && err->load() == 0) {
            const ssize_t w = pwrite(fd, slots + (size_t)s * IO_CHUNK + done, pend_len[s] - done,
                                     (off_t)(job.offset + pend_off[s] + done));
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) {
                fail(errno ? errno : EIO);
                break;
            }
            done += (size_t)w;
        }
        used[s] = false;
    };
    int s = 0;
    for (size_t off = 0; off < job.bytes && err->load() == 0; off += IO_CHUNK) {
        const size_t len = job.bytes - off < IO_CHUNK ? job.bytes - off : IO_CHUNK;
        drain(s);
        if (hipMemcpyAsync(slots + (size_t)s * IO_CHUNK, job.dev + off, len, hipMemcpyDeviceToHost, st) != hipSuccess) fail(-1);
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

// One writer thread per job (file): buffered writes to one file are serialised by the file system (one inode lock), so
// a second writer on the same file only adds contention; different files do run side by side.
int io_download(xh_ctx *ctx, const IoJob *jobs, int njobs) {
    if (!ctx || njobs < 0 || (njobs && !jobs)) return XH_ERR_ARG;
    for (int j = 0; j < njobs; ++j)
        if (!jobs[j].path || (jobs[j].bytes && !jobs[j].dev)) return XH_ERR_ARG;
    if (njobs > IO_MAX_FILES) return xh_fail(ctx, XH_ERR_LIMIT, "at most %d files per call", IO_MAX_FILES);
    const int rc = xh_settle(ctx);      // earlier work may still write the sources; a routing call that must be re-run is re-run here
    if (rc != XH_OK && rc != XH_ERR_DEVICE) return rc;
    ctx->work_seq += 1;
    if (njobs == 0) return rc;
    int fds[IO_MAX_FILES];
    auto close_all = [&](int upto) {
        int bad = 0;
        for (int j = 0; j < upto; ++j)
            if (close(fds[j]) != 0) bad = errno ? errno : EIO;
        return bad;
    };
    for (int j = 0; j < njobs; ++j) {
        fds[j] = open(jobs[j].path, O_WRONLY | O_CREAT, 0644);
        if (fds[j] < 0) {
            const int e = errno;
            close_all(j);
            return xh_fail(ctx, XH_ERR_ARG, "%s: %s", jobs[j].path, strerror(e));
        }
    }
    const int rr = io_ring(ctx, njobs);
    if (rr != XH_OK) {
        close_all(njobs);
        return rr;
    }
    std::atomic<int> err{0};
    std::thread pool[IO_MAX_FILES];
    for (int t = 0; t < njobs; ++t)
        pool[t] = std::thread(io_writer, ctx->device, fds[t], jobs[t], (char *)ctx->io_ring + (size_t)t * IO_SLOTS * IO_CHUNK, &err);
    for (int t = 0; t < njobs; ++t) pool[t].join();
    const int cerr = close_all(njobs);
    if (cerr && err.load() == 0) err.store(cerr);
    if (err.load() > 0) return xh_fail(ctx, XH_ERR_ARG, "%s: %s", jobs[0].path, strerror(err.load()));
    if (err.load() < 0) return xh_fail(ctx, XH_ERR_HIP, "%s: a copy of the file transfer failed", jobs[0].path);
    return rc;
}

}  // namespace

extern "C" {

int xh_upload_file(xh_ctx *ctx, void *d_dst, const char *path, uint64_t offset, size_t bytes, int threads) {
    (void)threads;                                   // kept in the signature: the mapped copy needs no host threads
    if (!ctx || !path || (bytes && !d_dst)) return XH_ERR_ARG;
    const int rc = xh_settle(ctx);                   // earlier work may still read the destination; re-routes happen here
    if (rc != XH_OK && rc != XH_ERR_DEVICE) return rc;
    ctx->work_seq += 1;
    if (bytes == 0) return rc;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return xh_fail(ctx, XH_ERR_ARG, "%s: %s", path, strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size < offset + bytes) {
        close(fd);
        return xh_fail(ctx, XH_ERR_ARG, "%s: shorter than offset %llu + %zu bytes", path, (unsigned long long)offset, bytes);
    }
    const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE);
    const uint64_t start = offset - offset % page;
    const size_t span = (size_t)(offset - start) + bytes;
    void *map = mmap(nullptr, span, PROT_READ, MAP_SHARED, fd, (off_t)start);   // shared, like numpy.memmap(mode="r")
    close(fd);                                       // the mapping keeps the file
    if (map == MAP_FAILED) return xh_fail(ctx, XH_ERR_ARG, "%s: mmap: %s", path, strerror(errno));
    (void)madvise(map, span, MADV_SEQUENTIAL);
    hipError_t e = hipMemcpyAsync(d_dst, (const char *)map + (offset - start), bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the mapping goes away below
    munmap(map, span);
    if (e != hipSuccess) return xh_fail(ctx, XH_ERR_HIP, "%s: copy from the mapped file: %s", path, hipGetErrorString(e));
    return rc;
}

int xh_download_file(xh_ctx *ctx, const void *d_src, const char *path, uint64_t offset, size_t bytes, int threads) {
    (void)threads;                                   // one writer per file
    const IoJob job{(const char *)d_src, path, offset, bytes};
    return io_download(ctx, &job, 1);
}

int xh_download_files(xh_ctx *ctx, int n, const void *const *d_srcs, const char *const *paths, const uint64_t *offsets,
                      const size_t *bytes) {
    if (n < 0 || n > IO_MAX_FILES || (n && (!d_srcs || !paths || !offsets || !bytes))) return XH_ERR_ARG;
    IoJob jobs[IO_MAX_FILES];
    for (int j = 0; j < n; ++j) jobs[j] = IoJob{(const char *)d_srcs[j], paths[j], offsets[j], bytes[j]};
    return io_download(ctx, jobs, n);
}

}  // extern "C"
