// ABCD(+snow) monthly water balance on gfx950.
//
// Replaces xanthos/runoff/abcd.py: ABCD.spinup / set_vals / simulate (:246-311), abcd_dist (:171-228),
// set_rain_and_snow (:141-169) and the basin-chunk driver _run_basins / abcd_parallel / abcd_execute (:314-422).
//
// Three launches:
//   k_abcd<SPINUP>      one thread per cell marches the first `spinup` months from (SM, GW) = (100, 500) and keeps
//                       soil moisture / groundwater of the last three Decembers (rows -1, -13, -25; :255-259)
//   k_abcd_basin_mean   one workgroup per basin: nan-mean over the basin's cells per December, mean of the three
//                       (fixed-order tree reduction in LDS => bitwise reproducible)
//   k_abcd<SIM>         one thread per cell marches all months from the basin means and writes AET, Q, Sav
//
// Memory: arrays stay in the reference layout [ncell, nmonths] (month fastest).  A thread walks its own row in
// tiles of TM months: 16-byte loads of the next tile are issued before the current tile is computed (software
// prefetch in registers), results leave as 16-byte stores.  State (snowpack, soil moisture, groundwater) lives in
// registers for the whole march.  Algorithmic HBM bytes per simulated cell-month: 24 read + 24 written
// (+ 24 x spinup/nmonths for the spin-up pass) = 52.8 B at spinup 120 / 600 months.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "xh_abcd_dev.h"
#include "xh_common.h"
#include "xh_stage.h"

namespace {

constexpr int TM = 8;           // months per register tile = one 64-byte sector per row and array (rows are 16-B aligned: nmonths is even)

using namespace xh_abcd_dev;

struct Tile {
    double pet[TM], pr[TM], tn[TM];
};

__device__ __forceinline__ void load_tile(Tile &t, const double *__restrict__ pet, const double *__restrict__ pr,
                                          const double *__restrict__ tn, int64_t off) {
#pragma unroll
    for (int j = 0; j < TM; j += 2) {
        const double2 a = *reinterpret_cast<const double2 *>(pet + off + j);
        const double2 b = *reinterpret_cast<const double2 *>(pr + off + j);
        t.pet[j] = a.x, t.pet[j + 1] = a.y;
        t.pr[j] = b.x, t.pr[j + 1] = b.y;
        const double2 c = *reinterpret_cast<const double2 *>(tn + off + j);   // tn is never NULL here (see caller):
        t.tn[j] = c.x, t.tn[j + 1] = c.y;                                      // branch-free loads keep vmcnt waits exact
    }
}

// SPINUP = true : march months [0, nsteps) without output, record the Decembers in dec[6][ncell]
// SPINUP = false: march months [0, nsteps) from sm0/gw0 of the cell's basin, write aet / q / sav
template <bool SPINUP>
__global__ void __launch_bounds__(64) k_abcd(int64_t ncell, int nmonths, int nsteps, const int *__restrict__ par_index,
                                             const int *__restrict__ basin_index, const double *__restrict__ pars,
                                             const double *__restrict__ pet, const double *__restrict__ precip,
                                             const double *__restrict__ tmin, const double *__restrict__ sm0,
                                             const double *__restrict__ gw0, double *__restrict__ dec,
                                             double *__restrict__ aet, double *__restrict__ q,
                                             double *__restrict__ sav) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const bool snow_on = tmin != nullptr;
    const AbcdPar P = load_par(pars, par_index[c], snow_on);
    const XhExpConsts K = xh_exp_consts();
    AbcdState s;
    s.snowpack = 0.0;                                                 // SN0 (:98); also reset before simulate()
    if (SPINUP) {
        s.sm = 100.0;                                                 // inv[1] (:82-83)
        s.gw = 500.0;                                                 // inv[2] (:84)
    } else {
        const int b = basin_index[c];
        s.sm = sm0[b];
        s.gw = gw0[b];
    }
    const int64_t row = c * (int64_t)nmonths;
    const int ntiles = nsteps / TM;
    // Ping-pong register tiles A / B, loop unrolled by two: the loads of the tile after next are in flight while a tile
    // is computed, and no tile is ever copied (a `cur = nxt` copy made the compiler wait for the loads it had just
    // issued: s_waitcnt vmcnt(0) at the bottom of every iteration).
    auto compute_tile = [&](const Tile &tl, int m0) {
        double oa[TM], oq[TM], os[TM];
        AbcdPre pre[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) pre[j] = abcd_pre(P, K, snow_on, tl.pet[j], tl.pr[j], tl.tn[j]);   // independent: ILP
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            abcd_step(P, s, snow_on, (m0 + j) == 0, pre[j], oa[j], oq[j]);
            os[j] = s.sm;
            if (SPINUP) {
                const int m = m0 + j;
                const int k = (m == nsteps - 1) ? 0 : ((m == nsteps - 13) ? 1 : ((m == nsteps - 25) ? 2 : -1));
                if (k >= 0) {
                    dec[(int64_t)k * ncell + c] = s.sm;
                    dec[(int64_t)(3 + k) * ncell + c] = s.gw;
                }
            }
        }
        if (!SPINUP) {
#pragma unroll
            for (int j = 0; j < TM; j += 2) {
                if (aet) *reinterpret_cast<double2 *>(aet + row + m0 + j) = make_double2(oa[j], oa[j + 1]);
                if (q) *reinterpret_cast<double2 *>(q + row + m0 + j) = make_double2(oq[j], oq[j + 1]);
                if (sav) *reinterpret_cast<double2 *>(sav + row + m0 + j) = make_double2(os[j], os[j + 1]);
            }
        }
    };
    const double *tn_rows = tmin ? tmin : precip;   // without snow the values are loaded but never used (snow_on false)
    Tile A, B;
    if (ntiles > 0) load_tile(A, pet, precip, tn_rows, row);
    for (int t = 0; t < ntiles; t += 2) {
        if (t + 1 < ntiles) load_tile(B, pet, precip, tn_rows, row + (int64_t)(t + 1) * TM);
        compute_tile(A, t * TM);
        if (t + 2 < ntiles) load_tile(A, pet, precip, tn_rows, row + (int64_t)(t + 2) * TM);
        if (t + 1 < ntiles) compute_tile(B, (t + 1) * TM);
    }
    for (int m = ntiles * TM; m < nsteps; ++m) {                      // tail months (spin-up length is arbitrary)
        double oa, oq;
        abcd_month(P, K, s, snow_on, m == 0, pet[row + m], precip[row + m], tmin ? tmin[row + m] : 0.0, oa, oq);
        if (SPINUP) {
            const int k = (m == nsteps - 1) ? 0 : ((m == nsteps - 13) ? 1 : ((m == nsteps - 25) ? 2 : -1));
            if (k >= 0) {
                dec[(int64_t)k * ncell + c] = s.sm;
                dec[(int64_t)(3 + k) * ncell + c] = s.gw;
            }
        } else {
            if (aet) aet[row + m] = oa;
            if (q) q[row + m] = oq;
            if (sav) sav[row + m] = s.sm;
        }
    }
}

// ---------------------------------------------------------------------------------------------- tiled kernel
// Same march, different data movement.  In k_abcd a wave's 16-byte load touches 64 different cache lines (one per
// cell row) and its stores are quarter-line fragments; the texture-address path then serialises ~64 line requests per
// instruction and, with one wave per SIMD, nothing hides it.  Here ONE WAVE owns CPW cells and moves whole 128-byte
// lines: lane (r, k) = (lane / 8, lane % 8) copies the 16-byte chunk k of row r (+ 8 rows per pass), so an
// instruction covers 8 complete lines; tiles of 16 months are staged in LDS ([cell][17] doubles: conflict-free both
// for the copy lanes and for the compute lanes), the lanes < CPW march their cell through the tile reading month j at
// [cell][j], results overwrite the inputs in place (PET -> AET, precipitation -> Q, tmin -> soil moisture) and leave
// as whole lines.  Rows start at byte c x nmonths x 8, which is a multiple of 128 only for every other cell at 600
// months, so each row's tiles are shifted by s(c) = (c x nmonths) mod 16 months: tile t of cell c holds months
// [16 t - s, 16 t - s + 16) and EVERY global access is a full aligned line.  The next tile's lines are loaded into
// registers while the current one is computed.  CPW = 32 gives twice the waves (two per SIMD), whose dependent chains
// overlap.
constexpr int TMS = 16;            // months per LDS tile = one 128-byte line per row and array
constexpr int TLD = TMS + 1;       // padded row length (doubles)

// (two waves per SIMD: at most 256 registers, AGPRs included -- CPW = 32 relies on its two waves per SIMD, and in a fed run
// (xh_fused.hip, mode 1) one of these waves has to fit beside a routing wave's 256)
template <bool SPINUP, int CPW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CPW < 64 ? 2 : 1, CPW < 64 ? 2 : 1))) k_abcd_tile(int64_t ncell, int nmonths, int nsteps, int m_begin, int m_end,
                                                  double *__restrict__ state,      // [3][ncell] carried between month blocks
                                                  const int *__restrict__ par_index, const int *__restrict__ basin_index,
                                                  const double *__restrict__ pars, const double *__restrict__ pet,
                                                  const double *__restrict__ precip, const double *__restrict__ tmin,
                                                  const double *__restrict__ sm0, const double *__restrict__ gw0,
                                                  double *__restrict__ dec, double *__restrict__ aet,
                                                  double *__restrict__ q, double *__restrict__ sav,
                                                  double *__restrict__ q_staged) {      // [ceil(nmonths / 16)][ncell][16] or NULL
    constexpr int NP = CPW / 8;                          // copy passes per tile (8 rows each)
    __shared__ double T[3][CPW * TLD];
    const int lane = threadIdx.x;
    const int64_t cell0 = (int64_t)blockIdx.x * CPW;
    const bool snow_on = tmin != nullptr;
    const double *__restrict__ src[3] = {pet, precip, tmin ? tmin : precip};
    double *__restrict__ dst[3] = {aet, q, sav};

    // ---- copy role
    const int ck = lane & 7, cr = lane >> 3;
    int64_t crow[NP];                                    // element offset of the row, or -1
    int cshift[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int64_t c = cell0 + p * 8 + cr;
        crow[p] = c < ncell ? c * (int64_t)nmonths : -1;
        cshift[p] = c < ncell ? (int)((c * (int64_t)nmonths) & (TMS - 1)) : 0;
    }
    // months [m_begin, m_end) of the march (both even): tiles that can hold one of them for some shift (0 .. 14)
    const int tile0 = m_begin / TMS, ntiles = (m_end + (TMS - 2) + TMS - 1) / TMS;
    double2 R[3][NP];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int m = t * TMS - cshift[p] + 2 * ck;           // first of the chunk's two months
            const bool ok = crow[p] >= 0 && m >= m_begin && m < m_end;
#pragma unroll
            for (int a = 0; a < 3; ++a)
                R[a][p] = ok ? *reinterpret_cast<const double2 *>(src[a] + crow[p] + m) : make_double2(0.0, 0.0);
        }
    };

    // ---- compute role
    const int64_t c = cell0 + lane;
    const bool mine = lane < CPW && c < ncell;
    const int64_t cc = mine ? c : (ncell - 1);
    const AbcdPar P = load_par(pars, par_index[cc], snow_on);
    const XhExpConsts K = xh_exp_consts();
    const int shift = (int)((cc * (int64_t)nmonths) & (TMS - 1));
    AbcdState s;
    s.snowpack = 0.0;
    if (m_begin > 0) {                                   // a later block of months: the state the previous block left
        s.snowpack = state[cc];
        s.sm = state[ncell + cc];
        s.gw = state[2 * ncell + cc];
    } else if (SPINUP) {
        s.sm = 100.0;
        s.gw = 500.0;
    } else {
        const int b = basin_index[cc];
        s.sm = sm0[b];
        s.gw = gw0[b];
    }
    double *mypet = &T[0][(lane < CPW ? lane : 0) * TLD], *mypr = &T[1][(lane < CPW ? lane : 0) * TLD],
           *mytn = &T[2][(lane < CPW ? lane : 0) * TLD];

    load_tile(tile0);
    for (int t = tile0; t < ntiles; ++t) {
        // registers -> LDS (the previous tile's write-out has finished: barrier at the bottom)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int o = (p * 8 + cr) * TLD + 2 * ck;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                T[a][o] = R[a][p].x;
                T[a][o + 1] = R[a][p].y;
            }
        }
        if (t + 1 < ntiles) load_tile(t + 1);            // in flight while this tile is computed
        __syncthreads();
        if (mine) {
            const int mbase = t * TMS - shift;
            // CB months at a time: their state-independent parts are evaluated together (instruction-level parallelism), then
            // the recurrence runs through them.  8 for the 32-cell wave; the wider waves hold more prefetch registers and
            // batch 4 (at 8 they spill 33 - 59 registers per lane, at 4 the 40-cell wave spills none)
            constexpr int CB = CPW == 32 ? 8 : 4;
#pragma unroll
            for (int h = 0; h < TMS; h += CB) {
                double ipet[CB], ipr[CB], itn[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    ipet[j] = mypet[h + j];
                    ipr[j] = mypr[h + j];
                    itn[j] = mytn[h + j];
                }
                if (mbase + h >= m_begin && mbase + h + CB <= m_end) {            // all CB months belong to this block
                    AbcdPre pre[CB];
#pragma unroll
                    for (int j = 0; j < CB; ++j) pre[j] = abcd_pre(P, K, snow_on, ipet[j], ipr[j], itn[j]);
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const int m = mbase + h + j;
                        double oa, oq;
                        abcd_step(P, s, snow_on, m == 0, pre[j], oa, oq);
                        if (SPINUP) {
                            const int k = (m == nsteps - 1) ? 0 : ((m == nsteps - 13) ? 1 : ((m == nsteps - 25) ? 2 : -1));
                            if (k >= 0) {
                                dec[(int64_t)k * ncell + c] = s.sm;
                                dec[(int64_t)(3 + k) * ncell + c] = s.gw;
                            }
                        } else {
                            mypet[h + j] = oa;
                            mypr[h + j] = oq;
                            mytn[h + j] = s.sm;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const int m = mbase + h + j;
                        if (m >= m_begin && m < m_end) {
                            double oa, oq;
                            abcd_month(P, K, s, snow_on, m == 0, ipet[j], ipr[j], itn[j], oa, oq);
                            if (SPINUP) {
                                const int k = (m == nsteps - 1) ? 0 : ((m == nsteps - 13) ? 1 : ((m == nsteps - 25) ? 2 : -1));
                                if (k >= 0) {
                                    dec[(int64_t)k * ncell + c] = s.sm;
                                    dec[(int64_t)(3 + k) * ncell + c] = s.gw;
                                }
                            } else {
                                mypet[h + j] = oa;
                                mypr[h + j] = oq;
                                mytn[h + j] = s.sm;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (!SPINUP) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int m = t * TMS - cshift[p] + 2 * ck;
                if (crow[p] >= 0 && m >= m_begin && m < m_end) {
                    const int o = (p * 8 + cr) * TLD + 2 * ck;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (dst[a]) *reinterpret_cast<double2 *>(dst[a] + crow[p] + m) = make_double2(T[a][o], T[a][o + 1]);
                    // the routing kernel's copy of the runoff when it runs beside this march (xh_fused.hip, mode 1): month m
                    // of cell c at [m / 16][c][m % 16] -- blocks of months that are multiples of 16 leave every 128-byte
                    // line of it written whole by one launch
                    if (q_staged)
                        *reinterpret_cast<double2 *>(q_staged + ((int64_t)(m >> 4) * ncell + (cell0 + p * 8 + cr)) * 16 + (m & 15)) =
                            make_double2(T[1][o], T[1][o + 1]);
                }
            }
            __syncthreads();
        }
    }
    if (state && mine) {
        state[c] = s.snowpack;
        state[ncell + c] = s.sm;
        state[2 * ncell + c] = s.gw;
    }
}

// set_vals (:246-282): per basin, mean over the Decembers {-1,-13,-25} of nanmean over the basin's cells.
// One workgroup per basin; cells of basin b are cells[ptr[b] .. ptr[b+1]).
__global__ void __launch_bounds__(256) k_abcd_basin_mean(const int *__restrict__ ptr, const int *__restrict__ cells,
                                                         int64_t ncell, const double *__restrict__ dec,
                                                         double *__restrict__ sm0, double *__restrict__ gw0) {
    __shared__ double ssum[6][256];
    __shared__ int scnt[6][256];
    const int b = blockIdx.x;
    const int lo = ptr[b], hi = ptr[b + 1];
    double sum[6] = {0, 0, 0, 0, 0, 0};
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int c = cells[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double v = dec[(int64_t)k * ncell + c];
            if (v == v) {
                sum[k] += v;
                cnt[k] += 1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        ssum[k][threadIdx.x] = sum[k];
        scnt[k][threadIdx.x] = cnt[k];
    }
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                ssum[k][threadIdx.x] += ssum[k][threadIdx.x + stride];
                scnt[k][threadIdx.x] += scnt[k][threadIdx.x + stride];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean[6];
        for (int k = 0; k < 6; ++k) mean[k] = ssum[k][0] / (double)scnt[k][0];   // 0/0 = NaN like nanmean of all-NaN
        sm0[b] = ((mean[0] + mean[1]) + mean[2]) / 3.0;
        gw0[b] = ((mean[3] + mean[4]) + mean[5]) / 3.0;
    }
}

}  // namespace

int xh_abcd_prepare(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t spinup, int32_t n_groups,
                    const int32_t *h_basin_index, const int32_t *h_par_index, int64_t npar_rows, xh_abcd_setup *out) {
    XH_REQUIRE(ctx, h_basin_index && h_par_index, "xh_abcd: NULL argument");
    XH_REQUIRE(ctx, ncell >= 0 && nmonths > 0 && nmonths % 2 == 0, "xh_abcd: nmonths must be positive and even");
    XH_REQUIRE(ctx, ncell < (int64_t)1 << 31, "xh_abcd: too many cells");
    // the reference indexes rows -1, -13, -25 of the spin-up series and raises IndexError below 25 (:258-266)
    XH_REQUIRE(ctx, spinup >= 25, "xh_abcd: spin-up of %d months is too short (needs >= 25; abcd.py:258-266)", spinup);
    XH_REQUIRE(ctx, spinup <= nmonths, "xh_abcd: spin-up (%d) exceeds the number of months (%d)", spinup, nmonths);
    XH_REQUIRE(ctx, n_groups >= 1, "xh_abcd: n_groups must be >= 1");
    out->ncell = ncell;
    out->nmonths = nmonths;
    out->spinup = spinup;
    out->n_groups = n_groups;
    if (ncell == 0) return XH_OK;

    // CSR of cells per basin (host, tiny) + index arrays -> scratch
    std::vector<int> ptr(n_groups + 1, 0), cells(ncell), bidx(ncell), pidx(ncell);
    for (int64_t c = 0; c < ncell; ++c) {
        const int b = h_basin_index[c], p = h_par_index[c];
        XH_REQUIRE(ctx, b >= 0 && b < n_groups, "xh_abcd: basin index %d of cell %lld out of range", b, (long long)c);
        XH_REQUIRE(ctx, p >= 0 && p < npar_rows, "xh_abcd: parameter row %d of cell %lld out of range", p, (long long)c);
        ptr[b + 1]++;
        bidx[c] = b;
        pidx[c] = p;
    }
    for (int b = 0; b < n_groups; ++b) ptr[b + 1] += ptr[b];
    {
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (int64_t c = 0; c < ncell; ++c) cells[fill[bidx[c]]++] = (int)c;
    }
    const size_t n_int = (size_t)(n_groups + 1) + 3 * (size_t)ncell;
    const size_t int_bytes = (n_int * sizeof(int) + 255) & ~size_t(255);
    const size_t dbl = (9 * (size_t)ncell + 2 * (size_t)n_groups) * sizeof(double);
    // (as in xh_pm_prepare: the same cell lists as the last call and an untouched scratch slot -> nothing to upload, no
    //  synchronisation; the double arrays behind the lists are work space every launch rewrites before reading)
    std::vector<char> key(sizeof(int) * n_int + 4 * sizeof(int64_t));
    {
        char *k = key.data();
        const int64_t meta[4] = {ncell, nmonths, spinup, n_groups};
        memcpy(k, meta, sizeof(meta));
        k += sizeof(meta);
        memcpy(k, ptr.data(), sizeof(int) * (n_groups + 1));
        k += sizeof(int) * (n_groups + 1);
        memcpy(k, cells.data(), sizeof(int) * ncell);
        k += sizeof(int) * ncell;
        memcpy(k, bidx.data(), sizeof(int) * ncell);
        k += sizeof(int) * ncell;
        memcpy(k, pidx.data(), sizeof(int) * ncell);
    }
    const bool cached = ctx->scratch[1] && ctx->abcd_cache_gen == ctx->scratch_gen[1] && ctx->abcd_cache == key &&
                        ctx->scratch_bytes[1] >= int_bytes + dbl;
    void *buf = nullptr;
    int rc = XH_OK;
    if (cached) {
        buf = ctx->scratch[1];
    } else {
        rc = xh_scratch(ctx, 1, int_bytes + dbl, &buf);
        if (rc) return rc;
    }
    out->d_ptr = static_cast<int *>(buf);
    out->d_cells = out->d_ptr + (n_groups + 1);
    out->d_bidx = out->d_cells + ncell;
    out->d_pidx = out->d_bidx + ncell;
    out->d_dec = reinterpret_cast<double *>(static_cast<char *>(buf) + int_bytes);
    out->d_state = out->d_dec + 6 * ncell;
    out->d_sm0 = out->d_state + 3 * ncell;
    out->d_gw0 = out->d_sm0 + n_groups;
    if (cached) return XH_OK;
    XH_HIP(ctx, hipMemcpyAsync(out->d_ptr, ptr.data(), sizeof(int) * (n_groups + 1), hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipMemcpyAsync(out->d_cells, cells.data(), sizeof(int) * ncell, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipMemcpyAsync(out->d_bidx, bidx.data(), sizeof(int) * ncell, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipMemcpyAsync(out->d_pidx, pidx.data(), sizeof(int) * ncell, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host vectors die at return
    ctx->abcd_cache.swap(key);
    ctx->abcd_cache_gen = ctx->scratch_gen[1];
    return XH_OK;
}

// Which kernel marches: the spin-up on the thread-per-cell kernel -- it writes nothing, so staging its reads through LDS only
// adds work (0.09 ms against 0.18 ms for 120 months) -- and the simulation on the tiled kernel (whole-line traffic: 0.61 ms
// against 0.81 ms for 67,420 x 600).  (XH_ABCD_KERNEL / XH_ABCD_CPW forced other choices until round 5: measured, recorded
// in profiles/round2-3, removed.)
static int abcd_env() { return -1; }

int xh_abcd_enqueue_spinup(xh_ctx *ctx, hipStream_t st, const xh_abcd_setup &s, const double *d_pars,
                           const double *d_pet, const double *d_precip, const double *d_tmin) {
    if (s.ncell == 0) return XH_OK;
    const int64_t ncell = s.ncell;
    const unsigned blocks = (unsigned)((ncell + 63) / 64), blocks32 = (unsigned)((ncell + 31) / 32);
    const int mode = abcd_env() < 0 ? 0 : abcd_env();
    {
        xh_span sp = xh_span_begin_on(ctx, "abcd_spinup", st);
        if (mode == 32)
            hipLaunchKernelGGL((k_abcd_tile<true, 32>), dim3(blocks32), dim3(64), 0, st, ncell, s.nmonths, s.spinup, 0,
                               s.spinup, (double *)nullptr, s.d_pidx, s.d_bidx, d_pars, d_pet, d_precip, d_tmin,
                               (const double *)nullptr, (const double *)nullptr, s.d_dec, (double *)nullptr,
                               (double *)nullptr, (double *)nullptr, (double *)nullptr);
        else if (mode == 64)
            hipLaunchKernelGGL((k_abcd_tile<true, 64>), dim3(blocks), dim3(64), 0, st, ncell, s.nmonths, s.spinup, 0,
                               s.spinup, (double *)nullptr, s.d_pidx, s.d_bidx, d_pars, d_pet, d_precip, d_tmin,
                               (const double *)nullptr, (const double *)nullptr, s.d_dec, (double *)nullptr,
                               (double *)nullptr, (double *)nullptr, (double *)nullptr);
        else
            hipLaunchKernelGGL(k_abcd<true>, dim3(blocks), dim3(64), 0, st, ncell, s.nmonths, s.spinup, s.d_pidx,
                               s.d_bidx, d_pars, d_pet, d_precip, d_tmin, (const double *)nullptr,
                               (const double *)nullptr, s.d_dec, (double *)nullptr, (double *)nullptr,
                               (double *)nullptr);
        xh_span_end(sp);
    }
    {
        xh_span sp = xh_span_begin_on(ctx, "abcd_basin_mean", st);
        hipLaunchKernelGGL(k_abcd_basin_mean, dim3((unsigned)s.n_groups), dim3(256), 0, st, s.d_ptr, s.d_cells, ncell,
                           s.d_dec, s.d_sm0, s.d_gw0);
        xh_span_end(sp);
    }
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

int xh_abcd_enqueue_sim(xh_ctx *ctx, hipStream_t st, const xh_abcd_setup &s, int m_begin, int m_end,
                        const double *d_pars, const double *d_pet, const double *d_precip, const double *d_tmin,
                        double *d_aet, double *d_q, double *d_sav, double *d_q_staged) {
    if (s.ncell == 0 || m_end <= m_begin) return XH_OK;
    XH_REQUIRE(ctx, m_begin >= 0 && m_end <= s.nmonths && m_begin % 2 == 0 && m_end % 2 == 0, "xh_abcd: bad month block");
    const int64_t ncell = s.ncell;
    const unsigned blocks = (unsigned)((ncell + 63) / 64);
    const bool whole = m_begin == 0 && m_end == s.nmonths;
    int mode = abcd_env() < 0 ? 32 : abcd_env();
    if (mode == 0 && (!whole || d_q_staged)) mode = 32;          // only the tiled kernel marches blocks of months / stages the runoff
    // Cells per wave.  Every workgroup marches its cells through the whole block of months, so a grid that does not fit the
    // chip's wave slots at once (two 256-register waves per SIMD) pays for a whole second round: 2,048 workgroups 0.376 ms,
    // 2,049 workgroups 0.537 ms at 600 months (tools/abcd_tail_probe.py, profiles/round4/abcd_tail.txt) -- and 67,420 cells
    // in waves of 32 are 2,107.  The smallest of 32 / 40 / 48 cells per wave that fits in one round is used (40 for the
    // 0.5-degree grid: 1,686 workgroups); beyond that, rounds cannot be avoided and 32 is the cheapest wave.
    int cpw = 32;
    if (mode == 32 && abcd_env() < 0) {
        const int64_t slots = 2 * 4 * (int64_t)(ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256);
        for (int c : {32, 40, 48})
            if ((ncell + c - 1) / c <= slots) {
                cpw = c;
                break;
            }
    }
    const unsigned blocks_cpw = (unsigned)((ncell + cpw - 1) / cpw);
    double *state = whole ? nullptr : s.d_state;
    xh_span sp = xh_span_begin_on(ctx, "abcd_sim", st);
#define XH_ABCD_TILE_SIM(CPWV)                                                                                                  \
    hipLaunchKernelGGL((k_abcd_tile<false, CPWV>), dim3(blocks_cpw), dim3(64), 0, st, ncell, s.nmonths, s.nmonths, m_begin, m_end, \
                       state, s.d_pidx, s.d_bidx, d_pars, d_pet, d_precip, d_tmin, s.d_sm0, s.d_gw0, (double *)nullptr, d_aet,  \
                       d_q, d_sav, d_q_staged)
    if (mode == 32 && cpw == 40)
        XH_ABCD_TILE_SIM(40);
    else if (mode == 32 && cpw == 48)
        XH_ABCD_TILE_SIM(48);
    else if (mode == 32)
        XH_ABCD_TILE_SIM(32);
#undef XH_ABCD_TILE_SIM
    else if (mode == 64)
        hipLaunchKernelGGL((k_abcd_tile<false, 64>), dim3(blocks), dim3(64), 0, st, ncell, s.nmonths, s.nmonths, m_begin,
                           m_end, state, s.d_pidx, s.d_bidx, d_pars, d_pet, d_precip, d_tmin, s.d_sm0, s.d_gw0,
                           (double *)nullptr, d_aet, d_q, d_sav, d_q_staged);
    else
        hipLaunchKernelGGL(k_abcd<false>, dim3(blocks), dim3(64), 0, st, ncell, s.nmonths, s.nmonths, s.d_pidx, s.d_bidx,
                           d_pars, d_pet, d_precip, d_tmin, s.d_sm0, s.d_gw0, (double *)nullptr, d_aet, d_q, d_sav);
    xh_span_end(sp);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" int xh_abcd(xh_ctx *ctx, int64_t ncell, int32_t nmonths, int32_t spinup, int32_t n_groups,
                       const int32_t *h_basin_index, const int32_t *h_par_index, int64_t npar_rows,
                       const double *d_pars, const double *d_pet, const double *d_precip, const double *d_tmin,
                       double *d_aet, double *d_q, double *d_sav, double *d_sm0_out, double *d_gw0_out) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, d_pars && d_pet && d_precip, "xh_abcd: NULL argument");
    xh_abcd_setup s;
    int rc = xh_abcd_prepare(ctx, ncell, nmonths, spinup, n_groups, h_basin_index, h_par_index, npar_rows, &s);
    if (rc || ncell == 0) return rc;
    rc = xh_abcd_enqueue_spinup(ctx, ctx->stream, s, d_pars, d_pet, d_precip, d_tmin);
    if (rc) return rc;
    rc = xh_abcd_enqueue_sim(ctx, ctx->stream, s, 0, nmonths, d_pars, d_pet, d_precip, d_tmin, d_aet, d_q, d_sav, nullptr);
    if (rc) return rc;
    if (d_sm0_out) XH_HIP(ctx, hipMemcpyAsync(d_sm0_out, s.d_sm0, sizeof(double) * n_groups, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_gw0_out) XH_HIP(ctx, hipMemcpyAsync(d_gw0_out, s.d_gw0, sizeof(double) * n_groups, hipMemcpyDeviceToDevice, ctx->stream));
    return XH_OK;
}
