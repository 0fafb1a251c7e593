// Drought statistics on the device: the two array computations of
// xanthos/drought/drought_stats.py.
//
//   k_drought_thresh  getthresh (:150-171): the q-quantile over the years of a reference period, per cell and per
//                     period of the year -- np.percentile(..., axis=0) with numpy's default "linear" method.  The
//                     host passes the two order statistics that bracket the virtual index (n - 1) q and the
//                     interpolation weight, computed exactly as numpy does (they depend only on n and q); the kernel
//                     selects the two order statistics of each (cell, period) by rank counting and interpolates with
//                     numpy's _lerp: a + (b - a) g, or b - (b - a)(1 - g) when g >= 0.5.  A NaN anywhere in the
//                     sample gives NaN, as np.percentile does.
//   k_drought_stats   droughtstats (:85-148): per cell, a march over the months: duration D (months under the
//                     threshold so far), severity S (accumulated relative shortfall) and intensity I = S / D, all
//                     zero outside a drought.  The threshold of month t is row t mod K of the table (:133).
//
// Arrays keep the package's [ncell, nmonths] layout (the reference transposes to [ntime, ngrid] internally, :39-41);
// thresholds are [K, ncell] as the reference stores them, so a wave reads them coalesced.  Bound: HBM (8 B in, 24 B
// out per cell-month) with the same row-walking access pattern as the ABCD kernel.
#include <cmath>

#include "xh_common.h"

namespace {

// thread <-> (cell, period)
__global__ void __launch_bounds__(256) k_drought_thresh(int64_t ncell, int nmonths, int month0, int nyear, int nper,
                                                        int k_prev, int k_next, double gamma,
                                                        const double *__restrict__ hydro, double *__restrict__ thresh) {
    const int64_t total = ncell * (int64_t)nper;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / ncell);                 // consecutive threads = consecutive cells of one period
        const int64_t c = i - (int64_t)p * ncell;
        const double *v = hydro + c * (int64_t)nmonths + month0 + p;       // sample y is v[y * nper]
        double a = 0.0, b = 0.0;
        bool has_nan = false;
        for (int y = 0; y < nyear; ++y) {
            const double vy = v[(int64_t)y * nper];
            has_nan |= (vy != vy);
            int rank = 0;                               // position of sample y in the sorted sample (ties by index)
            for (int z = 0; z < nyear; ++z) {
                const double vz = v[(int64_t)z * nper];
                rank += (vz < vy || (vz == vy && z < y)) ? 1 : 0;
            }
            a = rank == k_prev ? vy : a;
            b = rank == k_next ? vy : b;
        }
        const double d = b - a;
        double r = a + d * gamma;                                           // numpy _lerp
        r = gamma >= 0.5 ? b - d * (1.0 - gamma) : r;
        thresh[(int64_t)p * ncell + c] = has_nan ? NAN : r;
    }
}

// thread <-> cell
__global__ void __launch_bounds__(256) k_drought_stats(int64_t ncell, int nmonths, int nthresh,
                                                       const double *__restrict__ hydro, const double *__restrict__ thresh,
                                                       double *__restrict__ S_out, double *__restrict__ I_out,
                                                       double *__restrict__ D_out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const double *h = hydro + c * (int64_t)nmonths;
    double S = 0.0, D = 0.0;
    int m = 0;
    for (int t0 = 0; t0 < nmonths; t0 += 2) {           // nmonths even: 16-byte loads and stores
        const double2 hv = *reinterpret_cast<const double2 *>(h + t0);
        double s2[2], i2[2], d2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double hy = k == 0 ? hv.x : hv.y;
            const double th = thresh[(int64_t)m * ncell + c];
            m = (m + 1 == nthresh) ? 0 : m + 1;
            const bool dry = hy < th;                                       // :137 (NaN compares false -> no drought)
            D = dry ? D + 1.0 : 0.0;                                        // :138
            S = dry ? S + (th - hy) / th : 0.0;                             // :139
            s2[k] = S;
            d2[k] = D;
            i2[k] = dry ? S / D : 0.0;                                      // :140
        }
        const int64_t o = c * (int64_t)nmonths + t0;
        if (S_out) *reinterpret_cast<double2 *>(S_out + o) = make_double2(s2[0], s2[1]);
        if (I_out) *reinterpret_cast<double2 *>(I_out + o) = make_double2(i2[0], i2[1]);
        if (D_out) *reinterpret_cast<double2 *>(D_out + o) = make_double2(d2[0], d2[1]);
    }
}

}  // namespace

extern "C" int xh_drought_thresholds(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t month0, int32_t nyear,
                                     int32_t nper, int32_t k_prev, int32_t k_next, double gamma, const double *d_hydro,
                                     double *d_thresh) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_hydro && d_thresh && ncell >= 0 && nmonths > 0, "xh_drought_thresholds: bad argument");
    XH_REQUIRE(ctx, nper >= 1 && nyear >= 1 && month0 >= 0 && (int64_t)month0 + (int64_t)nyear * nper <= nmonths,
               "xh_drought_thresholds: reference period (month %d, %d x %d) outside the %d months", month0, nyear, nper,
               nmonths);
    XH_REQUIRE(ctx, k_prev >= 0 && k_prev < nyear && k_next >= 0 && k_next < nyear && gamma >= 0.0 && gamma <= 1.0,
               "xh_drought_thresholds: order statistics %d, %d / weight %g invalid for %d samples", k_prev, k_next, gamma,
               nyear);
    if (ncell == 0) return XH_OK;
    const int64_t total = ncell * (int64_t)nper;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->prop.multiProcessorCount * 32;
    if (blocks > cap) blocks = cap;
    xh_span sp = xh_span_begin(ctx, "drought_thresh");
    hipLaunchKernelGGL(k_drought_thresh, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ncell, (int)nmonths, (int)month0,
                       (int)nyear, (int)nper, (int)k_prev, (int)k_next, gamma, d_hydro, d_thresh);
    xh_span_end(sp);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" int xh_drought_stats(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t nthresh, const double *d_hydro,
                                const double *d_thresh, double *d_severity, double *d_intensity, double *d_duration) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_hydro && d_thresh && ncell >= 0 && nmonths > 0 && nthresh >= 1, "xh_drought_stats: bad argument");
    XH_REQUIRE(ctx, nmonths % 2 == 0, "xh_drought_stats: nmonths (%d) must be even (whole years)", nmonths);
    if (ncell == 0) return XH_OK;
    xh_span sp = xh_span_begin(ctx, "drought_stats");
    hipLaunchKernelGGL(k_drought_stats, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, ctx->stream, ncell, (int)nmonths,
                       (int)nthresh, d_hydro, d_thresh, d_severity, d_intensity, d_duration);
    xh_span_end(sp);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}
