// Output aggregation on the device (first "next" row after the hot path).
//
// Replaces the array math of xanthos/data_writer/out_writer.py: agg_to_year (:237-248, pandas groupby over blocks of
// 12 month columns: NaN-skipping sum, or mean for channel flow), the mm -> km3 conversion of write() (:111-112,
// rows x area / 1e6) and agg_spatial (:250-265, NaN-skipping sum of the cells of each basin / country / region;
// ids without cells give NaN rows).  The six outputs are already in HBM after the pipeline, yearly aggregation
// shrinks what crosses PCIe (or the multi-GPU gather) 12x.
#include <algorithm>
#include <cmath>
#include <vector>

#include "xh_common.h"

namespace {

// np.sum of n <= 128 contiguous doubles, in numpy's order (pairwise_sum of numpy/_core/src/umath/loops_utils.h.src,
// numpy >= 1.9): fewer than 8 values are added left to right; otherwise eight accumulators take the values 8 at a
// time, are combined as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), and the remaining n % 8 values are added one by one.
// NaN propagates (accessible.py:42 relies on that to drop whole cell-years).
__device__ __forceinline__ double np_sum(const double *p, int n) {
    if (n < 8) {
        double r = 0.0;                       // numpy starts from the identity written to the output (-0.0 aside)
        for (int i = 0; i < n; ++i) r += p[i];
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = p[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += p[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += p[i];
    return 0.0 + res;                         // the reduction starts from the identity: -0.0 totals come out +0.0
}

// thread <-> (cell, group of `group` consecutive columns)
__global__ void __launch_bounds__(256) k_agg_time(int64_t ncell, int ncols, int group, int mode,
                                                  const double *__restrict__ scale, const double *__restrict__ in,
                                                  double *__restrict__ out) {
    const int ng = ncols / group;
    const int64_t total = ncell * (int64_t)ng;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / ng;
        const int g = (int)(i - c * ng);
        const double *p = in + c * (int64_t)ncols + (int64_t)g * group;
        double sum = 0.0;
        int cnt = 0;
        for (int j = 0; j < group; ++j) {
            const double v = p[j];
            if (v == v) {
                sum += v;
                ++cnt;
            }
        }
        double r = mode == 0 ? sum : (cnt ? sum / (double)cnt : NAN);     // pandas: sum skips NaN (all-NaN -> 0), mean -> NaN
        if (group == 1 && mode == 0) r = p[0];                             // plain conversion keeps NaN
        if (mode == 2) r = np_sum(p, group);                               // np.sum along a contiguous axis
        if (scale) r = r * scale[c];
        out[i] = r;
    }
}

// block <-> (group id, tile of 256 columns); cells of the group are summed in index order
__global__ void __launch_bounds__(256) k_agg_spatial(int ncols, const int *__restrict__ ptr, const int *__restrict__ cells,
                                                     const double *__restrict__ in, double *__restrict__ out) {
    const int k = blockIdx.x;
    const int t = blockIdx.y * blockDim.x + threadIdx.x;
    if (t >= ncols) return;
    const int lo = ptr[k], hi = ptr[k + 1];
    double sum = 0.0;
    for (int i = lo; i < hi; ++i) {
        const double v = in[(int64_t)cells[i] * ncols + t];
        if (v == v) sum += v;
    }
    out[(int64_t)k * ncols + t] = (hi > lo) ? sum : NAN;
}

// np.nan_to_num in place (data_load.py:120-125,:194-195: NaN -> 0, +/-inf -> +/-DBL_MAX), loader transform N3
__global__ void __launch_bounds__(256) k_nan_to_num(double *__restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = a[i];
        if (v != v) a[i] = 0.0;
        else if (v == INFINITY) a[i] = 1.7976931348623157e308;
        else if (v == -INFINITY) a[i] = -1.7976931348623157e308;
    }
}

}  // namespace

extern "C" int xh_nan_to_num(xh_ctx *ctx, double *d_arr, int64_t n) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, n >= 0 && (d_arr || n == 0), "xh_nan_to_num: bad argument");
    if (n == 0) return XH_OK;
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_nan_to_num, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_arr, n);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" int xh_agg_time(xh_ctx *ctx, int64_t ncell, int32_t ncols, int32_t group, int32_t mode, const double *d_scale,
                           const double *d_in, double *d_out) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_in && d_out && ncell >= 0 && ncols > 0, "xh_agg_time: bad argument");
    XH_REQUIRE(ctx, group >= 1 && ncols % group == 0, "xh_agg_time: ncols (%d) is not a multiple of group (%d)", ncols, group);
    XH_REQUIRE(ctx, mode >= 0 && mode <= 2, "xh_agg_time: mode must be 0 (sum), 1 (mean) or 2 (numpy sum)");
    XH_REQUIRE(ctx, mode != 2 || group <= 128, "xh_agg_time: numpy-order sums are implemented for blocks of <= 128 values");
    if (ncell == 0) return XH_OK;
    const int64_t total = ncell * (int64_t)(ncols / group);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    xh_span sp = xh_span_begin(ctx, "agg_time");
    hipLaunchKernelGGL(k_agg_time, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ncell, (int)ncols, (int)group, (int)mode,
                       d_scale, d_in, d_out);
    xh_span_end(sp);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" int xh_agg_spatial(xh_ctx *ctx, int64_t ncell, int32_t ncols, int32_t n_groups, const int32_t *h_group,
                              const double *d_in, double *d_out) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, h_group && d_in && d_out && ncell >= 0 && ncols > 0 && n_groups > 0, "xh_agg_spatial: bad argument");
    XH_REQUIRE(ctx, ncell < ((int64_t)1 << 31), "xh_agg_spatial: too many cells");
    std::vector<int> ptr(n_groups + 1, 0), cells;
    for (int64_t c = 0; c < ncell; ++c) {
        const int k = h_group[c];
        XH_REQUIRE(ctx, k >= -1 && k < n_groups, "xh_agg_spatial: group %d of cell %lld out of range", k, (long long)c);
        if (k >= 0) ptr[k + 1]++;
    }
    for (int k = 0; k < n_groups; ++k) ptr[k + 1] += ptr[k];
    cells.resize(ptr[n_groups]);
    {
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (int64_t c = 0; c < ncell; ++c)
            if (h_group[c] >= 0) cells[fill[h_group[c]]++] = (int)c;
    }
    void *buf = nullptr;
    const size_t bytes = (ptr.size() + cells.size()) * sizeof(int) + 64;
    int rc = xh_scratch(ctx, 2, bytes, &buf);
    if (rc) return rc;
    int *d_ptr = static_cast<int *>(buf), *d_cells = d_ptr + ptr.size();
    XH_HIP(ctx, hipMemcpyAsync(d_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (!cells.empty())
        XH_HIP(ctx, hipMemcpyAsync(d_cells, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    xh_span sp = xh_span_begin(ctx, "agg_spatial");
    hipLaunchKernelGGL(k_agg_spatial, dim3((unsigned)n_groups, (unsigned)((ncols + 255) / 256)), dim3(256), 0, ctx->stream,
                       (int)ncols, d_ptr, d_cells, d_in, d_out);
    xh_span_end(sp);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}
