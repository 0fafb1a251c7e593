// Bench support (NOT on the hot path): fill the synthetic benchmark forcing on the device.
//
// Same counter-based SplitMix64 streams and distributions as xanthos_amd/synth.py:make_forcing,
// so a 67,420 x 600 world (2.6 GB over eight arrays) is generated in milliseconds instead of minutes of numpy.
// Values agree with the numpy generator up to the last bits of log/cos/sqrt; parity tests always feed the SAME
// arrays to the HIP path and to the oracle, so the two generators never need to agree exactly.
#include <cmath>

#include "xh_common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double uni(uint64_t seed, uint64_t stream, uint64_t idx) {
    const uint64_t key = splitmix64(seed * 0xD1342543DE82EF95ull + stream * 0xA0761D6478BD642Full);
    return (double)(splitmix64(idx ^ key) >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double nrm(uint64_t seed, uint64_t stream, uint64_t idx) {
    const double u1 = uni(seed, 2 * stream, idx), u2 = uni(seed, 2 * stream + 1, idx);
    return sqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * M_PI * u2);
}

__global__ void __launch_bounds__(256) k_synth(uint64_t seed, double nan_frac, int64_t ncell, int nmonths, const double *__restrict__ lat,
                                               const int64_t *__restrict__ cell_ids, double *tas, double *tmin, double *rhs, double *wind, double *rsds,
                                               double *rlds, double *precip, double *abcd_tmin) {
    const int64_t total = ncell * (int64_t)nmonths;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / nmonths;
        const int m = (int)(i - c * nmonths);
        const int64_t gcid = cell_ids ? cell_ids[c] : c;            // global cell: the random streams follow it
        if (gcid < 0) {                                             // "no such cell" (tairprev row of cell 0): zeros
            if (tas) tas[i] = 0.0;
            continue;
        }
        const uint64_t idx = (uint64_t)gcid * 4096ull + (uint64_t)m;
        const double la = lat[c];
        const double coslat = cos(la * (M_PI / 180.0));
        const double hemi = la > 0.0 ? 1.0 : (la < 0.0 ? -1.0 : 0.0);
        const double season = sin(2.0 * M_PI * ((double)(m % 12) - 3.5) / 12.0);
        const double t = -10.0 + 35.0 * coslat * coslat + 10.0 * hemi * season + 2.0 * nrm(seed, 1, idx);
        const double tn = t - (2.0 + 8.0 * uni(seed, 4, idx));
        double rh = 65.0 + 20.0 * nrm(seed, 3, idx);
        rh = rh < 5.0 ? 5.0 : (rh > 100.0 ? 100.0 : rh);
        if (tas) tas[i] = t;
        if (!tmin) continue;                                        // tas only (tairprev rows of a shard)
        tmin[i] = tn;
        rhs[i] = rh;
        wind[i] = 0.5 + 7.5 * uni(seed, 10, idx);
        rsds[i] = 30.0 + 300.0 * uni(seed, 11, idx);
        rlds[i] = 150.0 + 280.0 * uni(seed, 12, idx);
        double pr = -40.0 * (log(1.0 - uni(seed, 13, idx)) + log(1.0 - uni(seed, 14, idx)));
        if (uni(seed, 15, (uint64_t)gcid) < nan_frac) pr = NAN;     // missing-data cells (precip keeps NaN, data_load.py:186)
        precip[i] = pr;
        abcd_tmin[i] = tn;
    }
}

}  // namespace

extern "C" int xh_synth_forcing(xh_ctx *ctx, uint64_t seed, double nan_frac, int64_t ncell, int32_t nmonths, const double *d_lat,
                                const int64_t *d_cell_ids, double *d_tas, double *d_tmin, double *d_rhs, double *d_wind, double *d_rsds,
                                double *d_rlds, double *d_precip, double *d_abcd_tmin) {
    if (!ctx) return XH_ERR_ARG;
    const bool all = d_tmin && d_rhs && d_wind && d_rsds && d_rlds && d_precip && d_abcd_tmin;
    const bool none = !d_tmin && !d_rhs && !d_wind && !d_rsds && !d_rlds && !d_precip && !d_abcd_tmin;
    XH_REQUIRE(ctx, d_lat && d_tas && (all || none), "xh_synth_forcing: NULL argument (all eight arrays, or d_tas only)");
    XH_REQUIRE(ctx, ncell >= 0 && nmonths > 0 && nmonths < 4096, "xh_synth_forcing: bad size");
    if (ncell == 0) return XH_OK;
    const int64_t total = ncell * (int64_t)nmonths;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, seed, nan_frac, ncell, (int)nmonths, d_lat,
                       d_cell_ids, d_tas, d_tmin, d_rhs, d_wind, d_rsds, d_rlds, d_precip, d_abcd_tmin);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}
