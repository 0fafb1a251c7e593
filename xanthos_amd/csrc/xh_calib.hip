// Batched ABCD calibration objective on gfx950.
//
// Replaces xanthos/calibrate/calibrate_abcd.py:basin_runoff + objective_kge (:134-213) for set_calibrate = 0.
// The reference evaluates ONE parameter vector per call (python + numpy over the basin's cells) and scipy's
// differential evolution calls it 10^3..10^5 times per basin.  Here a whole population is evaluated at once:
//
//   k_calib_march<SPINUP>  lanes <-> cells of the basin (forcing is stored [month, cell], so a wave reads 512
//                          contiguous bytes per month), every thread carries MB members' ABCD state in registers
//                          (forcing is loaded once per MB members).  Spin-up pass: per-wave sums/counts of the
//                          three December soil-moisture / groundwater rows.  Simulation pass: per-wave monthly
//                          sums of runoff (x area x 1e-6 for km3), nansum semantics.
//   k_calib_init           per member: basin mean of the Decembers (one basin id for all cells, :143) in fixed
//                          chunk order
//   k_calib_series / kge   per (member, month): sum the per-wave partials in fixed order; per member: KGE distance
//
// All cross-lane / cross-wave sums run in a fixed order, so results are reproducible run to run.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "xh_abcd_dev.h"
#include "xh_calib.h"
#include "xh_common.h"

namespace {

using namespace xh_abcd_dev;

constexpr int MB = 4;   // members per thread

// Sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS traffic, fixed order => reproducible); every
// lane gets the total.  A ds_bpermute butterfly (__shfl_xor) costs 12 LDS-pipe round trips per fp64 sum and was the
// critical path of the simulation pass, which needs one sum per member and month.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_move(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_sum(double v) {
    double x = v + dpp_move<0x111, 0xf, 0xf>(v);        // row_shr:1
    x += dpp_move<0x112, 0xf, 0xf>(v);                   // row_shr:2
    x += dpp_move<0x113, 0xf, 0xf>(v);                   // row_shr:3   -> sums of 4 consecutive lanes
    x += dpp_move<0x114, 0xf, 0xe>(x);                   // row_shr:4, banks 1-3
    x += dpp_move<0x118, 0xf, 0xc>(x);                   // row_shr:8, banks 2-3 -> lane 15 of each row = row total
    x += dpp_move<0x142, 0xa, 0xf>(x);                   // row_bcast:15 into rows 1 and 3
    x += dpp_move<0x143, 0xc, 0xf>(x);                   // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ AbcdPar member_par(const double *__restrict__ pars, int npar, int member) {
    const double *p = pars + (int64_t)member * npar;
    AbcdPar P;
    const double a = p[0];
    P.b = p[1] * 1000.0;
    P.c = p[2];
    P.d = p[3];
    P.m = npar > 4 ? p[4] : 0.0;
    xh_abcd_dev::finish_par(P, a);
    return P;
}

using CalibBasin = xh_calib_basin;

// grid.x = 64-cell chunks of ALL basins of the call, grid.y = member blocks of MB; block = 64 threads (one wave).
// Every basin carries its own population (pars[basin][member][npar]), so one launch evaluates a whole generation of
// every basin: a single basin is only months x 1.7 us of dependent chain, far too little to fill the chip.
template <bool SPINUP>
__global__ void __launch_bounds__(64) k_calib_march(const CalibBasin *__restrict__ basins,
                                                    const int *__restrict__ chunk_basin,
                                                    const int *__restrict__ active, int nsteps, int nmembers,
                                                    int npar, const double *__restrict__ pars,
                                                    const double *__restrict__ sm0, const double *__restrict__ gw0,
                                                    double *__restrict__ dec_sum,    // [chunk][member][6]
                                                    int *__restrict__ dec_cnt,       // [chunk][member][6]
                                                    double *__restrict__ part) {     // [chunk][member][nsteps]
    const int chunk = blockIdx.x, lane = threadIdx.x;
    const int b = chunk_basin[chunk];
    if (active && !active[b]) return;                                // basin already converged (device-side DE)
    const CalibBasin B = basins[b];
    const int ncell = B.ncell;
    const int c = (chunk - B.chunk0) * 64 + lane;
    const bool valid = c < ncell;
    const int cc = valid ? c : ncell - 1;
    const int mb0 = blockIdx.y * MB;
    const double *__restrict__ pet_t = B.pet, *__restrict__ pr_t = B.pr, *__restrict__ tn_t = B.tn;
    const bool snow_on = tn_t != nullptr;
    const double *area = B.area;
    const double scale = (valid && area) ? area[cc] : 1.0;

    AbcdPar P[MB];
    AbcdState s[MB];
    const XhExpConsts K = xh_exp_consts();
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int mem = min(mb0 + j, nmembers - 1);
        P[j] = member_par(pars, npar, b * nmembers + mem);
        s[j].snowpack = 0.0;
        s[j].sm = SPINUP ? 100.0 : sm0[b * nmembers + mem];
        s[j].gw = SPINUP ? 500.0 : gw0[b * nmembers + mem];
    }
    double pet = pet_t[cc], pr = pr_t[cc], tn = snow_on ? tn_t[cc] : 0.0;
    for (int m = 0; m < nsteps; ++m) {
        const double pet_c = pet, pr_c = pr, tn_c = tn;
        if (m + 1 < nsteps) {                                    // prefetch next month (coalesced across lanes)
            const int64_t o = (int64_t)(m + 1) * ncell + cc;
            pet = pet_t[o];
            pr = pr_t[o];
            tn = snow_on ? tn_t[o] : 0.0;
        }
        const int k = SPINUP ? ((m == nsteps - 1) ? 0 : ((m == nsteps - 13) ? 1 : ((m == nsteps - 25) ? 2 : -1))) : -1;
        // the rain / snow split does not depend on the member: evaluate it once, then only exp(-PET/b) per member
        AbcdPre pre = abcd_pre(P[0], K, snow_on, pet_c, pr_c, tn_c);
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            double aet, q;
            if (j > 0) pre.decay = xh_exp(quot(-pet_c, P[j].b, P[j].inv_b), K);
            abcd_step(P[j], s[j], snow_on, m == 0, pre, aet, q);
            if (SPINUP) {
                if (k >= 0) {                                    // wave-uniform
                    const bool sm_ok = valid && (s[j].sm == s[j].sm), gw_ok = valid && (s[j].gw == s[j].gw);
                    const double ssm = wave_sum(sm_ok ? s[j].sm : 0.0), sgw = wave_sum(gw_ok ? s[j].gw : 0.0);
                    const int nsm = __popcll(__ballot(sm_ok)), ngw = __popcll(__ballot(gw_ok));
                    if (lane == 0 && mb0 + j < nmembers) {
                        const int64_t o = ((int64_t)chunk * nmembers + (mb0 + j)) * 6;
                        dec_sum[o + k] = ssm;
                        dec_sum[o + 3 + k] = sgw;
                        dec_cnt[o + k] = nsm;
                        dec_cnt[o + 3 + k] = ngw;
                    }
                }
            } else {
                double v = area ? q * scale * 1e-6 : q;          // rsim * bsn_areas * 1e-6 (:159) or rsim (:162)
                v = (valid && v == v) ? v : 0.0;                 // nansum
                const double tot = wave_sum(v);
                if (lane == 0 && mb0 + j < nmembers)
                    part[((int64_t)chunk * nmembers + (mb0 + j)) * nsteps + m] = tot;
            }
        }
    }
}

// one thread per (basin, member): basin mean of the three Decembers over the basin's chunks, in chunk order
__global__ void __launch_bounds__(64) k_calib_init(const CalibBasin *__restrict__ basins,
                                                   const int *__restrict__ active, int nbasins, int nmembers,
                                                   const double *__restrict__ dec_sum, const int *__restrict__ dec_cnt,
                                                   double *__restrict__ sm0, double *__restrict__ gw0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbasins * nmembers) return;
    const int b = i / nmembers, mem = i - b * nmembers;
    if (active && !active[b]) return;
    const CalibBasin B = basins[b];
    double sum[6] = {0, 0, 0, 0, 0, 0};
    long long cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int ch = B.chunk0; ch < B.chunk0 + B.nchunks; ++ch) {
        const int64_t o = ((int64_t)ch * nmembers + mem) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            sum[k] += dec_sum[o + k];
            cnt[k] += dec_cnt[o + k];
        }
    }
    double mean[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mean[k] = sum[k] / (double)cnt[k];
    sm0[i] = ((mean[0] + mean[1]) + mean[2]) / 3.0;              // abcd.py:274-278
    gw0[i] = ((mean[3] + mean[4]) + mean[5]) / 3.0;
}

// series[basin][member][month] = sum over the basin's chunks of part[chunk][member][month]
__global__ void __launch_bounds__(256) k_calib_series(const CalibBasin *__restrict__ basins,
                                                      const int *__restrict__ active, int nbasins, int nmembers,
                                                      int nmonths, const double *__restrict__ part,
                                                      double *__restrict__ series) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_basin = (int64_t)nmembers * nmonths;
    if (i >= per_basin * nbasins) return;
    const int b = (int)(i / per_basin);
    if (active && !active[b]) return;
    const int64_t r = i - (int64_t)b * per_basin;                // member * nmonths + month
    const CalibBasin B = basins[b];
    double acc = 0.0;
    for (int ch = B.chunk0; ch < B.chunk0 + B.nchunks; ++ch) acc += part[(int64_t)ch * per_basin + r];
    series[i] = acc;
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) sh[threadIdx.x] += sh[threadIdx.x + stride];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// one workgroup per member: ED = sqrt((r-1)^2 + (sd_m/sd_o - 1)^2 + (mean_m/mean_o - 1)^2) (:196-213)
__global__ void __launch_bounds__(256) k_calib_kge(const int *__restrict__ active, int nmonths, int nmembers,
                                                   const double *__restrict__ series,
                                                   const double *__restrict__ obs_all, double *__restrict__ ed) {
    __shared__ double sh[256];
    const int mem = blockIdx.x;                          // basin * nmembers + member
    if (active && !active[mem / nmembers]) return;       // block-uniform
    const double *x = series + (int64_t)mem * nmonths;
    const double *obs = obs_all + (int64_t)(mem / nmembers) * nmonths;
    double sx = 0.0, so = 0.0;
    for (int m = threadIdx.x; m < nmonths; m += blockDim.x) {
        sx += x[m];
        so += obs[m];
    }
    const double n = (double)nmonths;
    const double mx = block_sum(sx, sh) / n, mo = block_sum(so, sh) / n;
    double vxx = 0.0, voo = 0.0, vxo = 0.0;
    for (int m = threadIdx.x; m < nmonths; m += blockDim.x) {
        const double dx = x[m] - mx, d_o = obs[m] - mo;
        vxx += dx * dx;
        voo += d_o * d_o;
        vxo += dx * d_o;
    }
    vxx = block_sum(vxx, sh);
    voo = block_sum(voo, sh);
    vxo = block_sum(vxo, sh);
    if (threadIdx.x == 0) {
        const double relvar = sqrt(vxx / n) / sqrt(voo / n);     // np.std, population
        const double bias = mx / mo;
        const double c00 = voo / (n - 1.0), c11 = vxx / (n - 1.0), c01 = vxo / (n - 1.0);   // np.corrcoef via np.cov
        double r = c01 / sqrt(c11) / sqrt(c00);
        r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);               // corrcoef clips to [-1, 1]
        ed[mem] = sqrt((r - 1.0) * (r - 1.0) + (relvar - 1.0) * (relvar - 1.0) + (bias - 1.0) * (bias - 1.0));
    }
}


// =============================================================================== member-lane layout
// lanes <-> members, CM cells per wave.  Everything that does not depend on the member -- PET, the rain / snow split,
// the melt class, the cell area -- is the same for all 64 lanes: it arrives through the scalar unit (s_load) from the
// split arrays k_calib_split filled once per problem, and the VALU only sees exp(-PET / b), the month update and three
// instructions of accumulation: the sum over the wave's cells is a plain per-lane sum, no cross-lane reduction.
// ~100 wave-instructions per member-cell-month against ~145 in the cell-lane kernel (whose DPP wave sum alone costs
// 25 per member and month).  The price is a population that fills whole waves of 64 members.
constexpr int CM = 8;    // cells per wave: 24 doubles of state per lane

__global__ void __launch_bounds__(256) k_calib_split(const xh_calib_basin *__restrict__ basins, int nmonths) {
    const xh_calib_basin B = basins[blockIdx.y];
    const bool snow_on = B.tn != nullptr;
    const int64_t n = (int64_t)B.ncell * nmonths;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double rain, snow, frac;
        int kind;
        abcd_split(snow_on, B.pr[i], snow_on ? B.tn[i] : 0.0, rain, snow, frac, kind);
        B.rain[i] = rain;
        B.snow[i] = snow;
        B.frac[i] = frac;
        B.kind[i] = kind;
    }
}

// grid.x = CM-cell chunks of all basins, grid.y = blocks of 64 members; one wave per block
template <bool SPINUP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_calib_march_m(const xh_calib_basin *__restrict__ basins,
                                                      const int *__restrict__ chunk_basin,
                                                      const int *__restrict__ active, int nsteps, int nmembers, int npar,
                                                      const double *__restrict__ pars, const double *__restrict__ sm0,
                                                      const double *__restrict__ gw0,
                                                      double *__restrict__ dec_sum,    // [chunk][6][member]
                                                      int *__restrict__ dec_cnt,       // [chunk][6][member]
                                                      double *__restrict__ part) {     // [chunk][nsteps][member]
    const int chunk = blockIdx.x;
    const int b = chunk_basin[chunk];
    if (active && !active[b]) return;
    const xh_calib_basin B = basins[b];
    const int ncell = B.ncell;
    const int base = (chunk - B.chunk0) * CM;
    const int cnt = min(CM, ncell - base);
    const int mem_raw = blockIdx.y * 64 + threadIdx.x;
    const bool ok = mem_raw < nmembers;
    const int mem = ok ? mem_raw : nmembers - 1;
    const bool snow_on = B.tn != nullptr;
    const AbcdPar P = member_par(pars, npar, b * nmembers + mem);
    const XhExpConsts K = xh_exp_consts();
    AbcdState s[CM];
#pragma unroll
    for (int j = 0; j < CM; ++j) {
        s[j].snowpack = 0.0;
        s[j].sm = SPINUP ? 100.0 : sm0[b * nmembers + mem];
        s[j].gw = SPINUP ? 500.0 : gw0[b * nmembers + mem];
    }
    const double *__restrict__ pet_t = B.pet, *__restrict__ rain_t = B.rain, *__restrict__ snow_t = B.snow,
                 *__restrict__ frac_t = B.frac, *__restrict__ area = B.area;
    const int *__restrict__ kind_t = B.kind;
    // Lane j (< CM) fetches cell j's five values of a month with ordinary coalesced loads, two months ahead, and drops
    // them into a small LDS slot one month ahead; the march reads each cell's values from ONE LDS address (a broadcast
    // read, conflict-free), i.e. with wave-uniform values in vector registers and no memory access on its path.
    // (Left to itself the compiler addresses the uniform values through global loads and waits for every one: the split
    // arrays are written by another kernel of this translation unit, so it will not use the scalar cache.  Broadcasting
    // through v_readlane instead costs 11 readlanes + ~13 register moves per cell and month: 57.2 ms -> see DESIGN.)
    struct Slot {
        double pet, rain, snow, frac;
        int kind, pad;
    };
    __shared__ Slot slot[2][CM];
    const int lane = threadIdx.x;
    const int lj = min(lane & (CM - 1), cnt - 1);
    const double area_l = area ? area[base + lj] : 1.0;
    __shared__ double area_sh[CM];
    if (lane < CM) area_sh[lane] = area_l * 1e-6;
    auto fetch = [&](int m, Slot &v) {
        const int64_t o = (int64_t)m * ncell + base + lj;
        v.pet = pet_t[o];
        v.rain = rain_t[o];
        v.snow = snow_t[o];
        v.frac = frac_t[o];
        v.kind = kind_t[o];
        v.pad = 0;
    };
    Slot nxt;
    fetch(0, nxt);
    if (lane < CM) slot[0][lane] = nxt;
    if (nsteps > 1) fetch(1, nxt);
    __syncthreads();
    for (int m = 0; m < nsteps; ++m) {
        const int k = SPINUP ? ((m == nsteps - 1) ? 0 : ((m == nsteps - 13) ? 1 : ((m == nsteps - 25) ? 2 : -1))) : -1;
        // month m + 1 into the other slot (its loads were issued a month ago), month m + 2 into flight
        if (m + 1 < nsteps && lane < CM) slot[(m + 1) & 1][lane] = nxt;
        if (m + 2 < nsteps) fetch(m + 2, nxt);
        const Slot *cur = slot[m & 1];
        double tot = 0.0, ssm = 0.0, sgw = 0.0;
        int nsm = 0, ngw = 0;
#pragma unroll
        for (int j = 0; j < CM; ++j) {
            if (j < cnt) {                                           // wave-uniform
                AbcdPre pre;
                pre.pet = cur[j].pet;
                pre.rain = cur[j].rain;
                pre.snow = cur[j].snow;
                pre.frac = cur[j].frac;
                pre.kind = cur[j].kind;
                // (the argument of exp() and the groundwater quotient as bare products with the reciprocals: the objective is held
                // to 1e-9 of the oracle's, and these two cost 4 of the march's 89 VALU instructions per member, cell and month)
                pre.decay = xh_exp(-pre.pet * P.inv_b, K);
                double aet, q;
                abcd_step<true>(P, s[j], snow_on, m == 0, pre, aet, q);
                if (SPINUP) {
                    if (k >= 0) {
                        const bool sm_ok = s[j].sm == s[j].sm, gw_ok = s[j].gw == s[j].gw;
                        ssm += sm_ok ? s[j].sm : 0.0;
                        sgw += gw_ok ? s[j].gw : 0.0;
                        nsm += sm_ok ? 1 : 0;
                        ngw += gw_ok ? 1 : 0;
                    }
                } else {
                    const double v = area ? q * area_sh[j] : q;                // rsim * bsn_areas * 1e-6 (:159; area_sh holds area * 1e-6) or rsim (:162)
                    tot += (v == v) ? v : 0.0;                                // nansum
                }
            }
        }
        if (SPINUP) {
            if (k >= 0 && ok) {
                const int64_t o = (int64_t)chunk * 6 * nmembers + mem;
                dec_sum[o + (int64_t)k * nmembers] = ssm;
                dec_sum[o + (int64_t)(3 + k) * nmembers] = sgw;
                dec_cnt[o + (int64_t)k * nmembers] = nsm;
                dec_cnt[o + (int64_t)(3 + k) * nmembers] = ngw;
            }
        } else if (ok) {
            part[((int64_t)chunk * nsteps + m) * nmembers + mem] = tot;
        }
        __syncthreads();      // one wave: orders this month's slot reads before the next month's slot writes
    }
}

__global__ void __launch_bounds__(64) k_calib_init_m(const xh_calib_basin *__restrict__ basins,
                                                     const int *__restrict__ active, int nbasins, int nmembers,
                                                     const double *__restrict__ dec_sum, const int *__restrict__ dec_cnt,
                                                     double *__restrict__ sm0, double *__restrict__ gw0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbasins * nmembers) return;
    const int b = i / nmembers, mem = i - b * nmembers;
    if (active && !active[b]) return;
    const xh_calib_basin B = basins[b];
    double sum[6] = {0, 0, 0, 0, 0, 0};
    long long cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int ch = B.chunk0; ch < B.chunk0 + B.nchunks; ++ch) {
        const int64_t o = (int64_t)ch * 6 * nmembers + mem;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            sum[k] += dec_sum[o + (int64_t)k * nmembers];
            cnt[k] += dec_cnt[o + (int64_t)k * nmembers];
        }
    }
    double mean[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mean[k] = sum[k] / (double)cnt[k];
    sm0[i] = ((mean[0] + mean[1]) + mean[2]) / 3.0;              // abcd.py:274-278
    gw0[i] = ((mean[3] + mean[4]) + mean[5]) / 3.0;
}

// series_m[basin][month][member] = sum over the basin's chunks, in chunk order
__global__ void __launch_bounds__(256) k_calib_series_m(const xh_calib_basin *__restrict__ basins,
                                                        const int *__restrict__ active, int nbasins, int nmembers,
                                                        int nmonths, const double *__restrict__ part,
                                                        double *__restrict__ series_m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_basin = (int64_t)nmembers * nmonths;
    if (i >= per_basin * nbasins) return;
    const int b = (int)(i / per_basin);
    if (active && !active[b]) return;
    const int64_t r = i - (int64_t)b * per_basin;                // month * nmembers + member
    const xh_calib_basin B = basins[b];
    double acc = 0.0;
    for (int ch = B.chunk0; ch < B.chunk0 + B.nchunks; ++ch) acc += part[(int64_t)ch * per_basin + r];
    series_m[i] = acc;
}

// one thread per (basin, member), lanes along the members: ED as in k_calib_kge, sums over the months in order
__global__ void __launch_bounds__(64) k_calib_kge_m(const int *__restrict__ active, int nbasins, int nmonths, int nmembers,
                                                    const double *__restrict__ series_m,
                                                    const double *__restrict__ obs_all, double *__restrict__ ed) {
    const int b = blockIdx.y;
    const int mem = blockIdx.x * blockDim.x + threadIdx.x;
    if (mem >= nmembers || (active && !active[b])) return;
    const double *x = series_m + (int64_t)b * nmonths * nmembers + mem;
    const double *obs = obs_all + (int64_t)b * nmonths;
    double sx = 0.0, so = 0.0;
    for (int m = 0; m < nmonths; ++m) {
        sx += x[(int64_t)m * nmembers];
        so += obs[m];
    }
    const double n = (double)nmonths;
    const double mx = sx / n, mo = so / n;
    double vxx = 0.0, voo = 0.0, vxo = 0.0;
    for (int m = 0; m < nmonths; ++m) {
        const double dx = x[(int64_t)m * nmembers] - mx, d_o = obs[m] - mo;
        vxx += dx * dx;
        voo += d_o * d_o;
        vxo += dx * d_o;
    }
    const double relvar = sqrt(vxx / n) / sqrt(voo / n);         // np.std, population
    const double bias = mx / mo;
    const double c00 = voo / (n - 1.0), c11 = vxx / (n - 1.0), c01 = vxo / (n - 1.0);   // np.corrcoef via np.cov
    double r = c01 / sqrt(c11) / sqrt(c00);
    r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
    ed[b * nmembers + mem] = sqrt((r - 1.0) * (r - 1.0) + (relvar - 1.0) * (relvar - 1.0) + (bias - 1.0) * (bias - 1.0));
}

// d_series [basin][member][month] from d_series_m [basin][month][member] (only when the caller asks for the series)
__global__ void __launch_bounds__(256) k_calib_series_out(int nbasins, int nmembers, int nmonths,
                                                          const double *__restrict__ series_m,
                                                          double *__restrict__ series) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_basin = (int64_t)nmembers * nmonths;
    if (i >= per_basin * nbasins) return;
    const int b = (int)(i / per_basin);
    const int64_t r = i - (int64_t)b * per_basin;
    const int mem = (int)(r / nmonths), m = (int)(r - (int64_t)mem * nmonths);
    series[i] = series_m[(int64_t)b * per_basin + (int64_t)m * nmembers + mem];
}

}  // namespace

int xh_calib_problem_plan(xh_ctx *ctx, int32_t nbasins, const int64_t *h_ncell, int32_t nmonths, int32_t spinup,
                          int32_t nmembers, int32_t npar, const double *const *h_pet_t,
                          const double *const *h_precip_t, const double *const *h_tmin_t,
                          const double *const *h_area, std::vector<xh_calib_basin> &basins,
                          std::vector<int> &chunk_basin, size_t *bytes, int *member_lanes) {
    XH_REQUIRE(ctx, nbasins > 0 && h_ncell && h_pet_t && h_precip_t, "xh_calib_objective: NULL argument");
    XH_REQUIRE(ctx, nmonths > 1 && nmembers > 0, "xh_calib_objective: bad size");
    XH_REQUIRE(ctx, npar == 4 || npar == 5, "xh_calib_objective: npar must be 4 (no snow) or 5");
    XH_REQUIRE(ctx, (npar == 5) == (h_tmin_t != nullptr), "xh_calib_objective: npar = 5 requires tmin and vice versa");
    XH_REQUIRE(ctx, spinup >= 25 && spinup <= nmonths, "xh_calib_objective: spin-up must be in [25, nmonths]");
    // member-lane kernel when the population fills its waves of 64 members to at least 3/4
    const int mblocks = (nmembers + 63) / 64;
    const int ml = (4 * nmembers >= 3 * 64 * mblocks) ? 1 : 0;
    *member_lanes = ml;
    const int cs = ml ? CM : 64;                                 // cells per chunk
    basins.assign(nbasins, xh_calib_basin());
    chunk_basin.clear();
    for (int b = 0; b < nbasins; ++b) {
        XH_REQUIRE(ctx, h_ncell[b] > 0 && h_ncell[b] < ((int64_t)1 << 24), "xh_calib_objective: basin %d has %lld cells",
                   b, (long long)h_ncell[b]);
        XH_REQUIRE(ctx, h_pet_t[b] && h_precip_t[b] && (npar == 4 || h_tmin_t[b]), "xh_calib_objective: NULL forcing");
        xh_calib_basin &B = basins[b];
        B.ncell = (int)h_ncell[b];
        B.chunk0 = (int)chunk_basin.size();
        B.nchunks = (B.ncell + cs - 1) / cs;
        B.pad = 0;
        B.rain = B.snow = B.frac = nullptr;
        B.kind = nullptr;
        B.pet = h_pet_t[b];
        B.pr = h_precip_t[b];
        B.tn = npar == 5 ? h_tmin_t[b] : nullptr;
        B.area = h_area ? h_area[b] : nullptr;
        chunk_basin.insert(chunk_basin.end(), B.nchunks, b);
    }
    XH_REQUIRE(ctx, (nmembers + MB - 1) / MB <= 65535, "xh_calib_objective: too many members");
    const size_t nchunks = chunk_basin.size(), nbm = (size_t)nbasins * nmembers;
    size_t cells = 0;
    for (int b = 0; b < nbasins; ++b) cells += (size_t)h_ncell[b];
    const size_t dbl = (size_t)nbasins * nmonths + 2 * nbm + nchunks * nmembers * 6 + nchunks * nmembers * (size_t)nmonths +
                       nbm * nmonths * (ml ? 2 : 1) + (ml ? 3 * cells * (size_t)nmonths : 0);
    const size_t tab_bytes = ((sizeof(xh_calib_basin) * nbasins + sizeof(int) * nchunks) + 255) & ~size_t(255);
    *bytes = dbl * sizeof(double) + nchunks * nmembers * 6 * sizeof(int) + (ml ? cells * (size_t)nmonths * sizeof(int) : 0) +
             tab_bytes + 512;
    return XH_OK;
}

int xh_calib_problem_place(xh_ctx *ctx, xh_calib_problem &P, int32_t nmonths, int32_t spinup, int32_t nmembers,
                           int32_t npar, std::vector<xh_calib_basin> &basins, const std::vector<int> &chunk_basin,
                           const double *h_obs, void *buf, int member_lanes) {
    XH_REQUIRE(ctx, h_obs != nullptr, "xh_calib_objective: obs is NULL");
    const int nbasins = (int)basins.size();
    const size_t nchunks = chunk_basin.size(), nbm = (size_t)nbasins * nmembers;
    const size_t n_dec = nchunks * nmembers * 6, n_part = nchunks * nmembers * (size_t)nmonths;
    const size_t tab_bytes = ((sizeof(xh_calib_basin) * nbasins + sizeof(int) * nchunks) + 255) & ~size_t(255);
    P.nbasins = nbasins;
    P.nmonths = nmonths;
    P.spinup = spinup;
    P.nmembers = nmembers;
    P.npar = npar;
    P.nchunks = nchunks;
    P.member_lanes = member_lanes;
    P.split_done = false;
    P.d_obs = static_cast<double *>(buf);
    P.d_sm0 = P.d_obs + (size_t)nbasins * nmonths;
    P.d_gw0 = P.d_sm0 + nbm;
    P.d_dec = P.d_gw0 + nbm;
    P.d_part = P.d_dec + n_dec;
    P.d_series = P.d_part + n_part;
    double *next = P.d_series + nbm * nmonths;
    P.d_series_m = nullptr;
    if (member_lanes) {
        P.d_series_m = next;
        next += nbm * nmonths;
        for (xh_calib_basin &B : basins) {                       // rain / snow / frac of every (month, cell) of the basin
            const size_t n = (size_t)B.ncell * nmonths;
            B.rain = next;
            B.snow = next + n;
            B.frac = next + 2 * n;
            next += 3 * n;
        }
    }
    P.d_basins = reinterpret_cast<xh_calib_basin *>(next);
    P.d_chunk_basin = reinterpret_cast<int *>(P.d_basins + nbasins);
    P.d_cnt = reinterpret_cast<int *>(reinterpret_cast<char *>(P.d_basins) + tab_bytes);
    if (member_lanes) {
        int *kind = P.d_cnt + n_dec;
        for (xh_calib_basin &B : basins) {
            B.kind = kind;
            kind += (size_t)B.ncell * nmonths;
        }
    }
    XH_HIP(ctx, hipMemcpyAsync(P.d_obs, h_obs, sizeof(double) * nbasins * nmonths, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipMemcpyAsync(P.d_basins, basins.data(), sizeof(xh_calib_basin) * nbasins, hipMemcpyHostToDevice,
                               ctx->stream));
    XH_HIP(ctx, hipMemcpyAsync(P.d_chunk_basin, chunk_basin.data(), sizeof(int) * nchunks, hipMemcpyHostToDevice,
                               ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // the host tables are the caller's locals
    return XH_OK;
}

static int calib_enqueue_m(xh_ctx *ctx, const xh_calib_problem &P, const double *d_pars, const int *d_active,
                           double *d_ed) {
    const size_t nbm = (size_t)P.nbasins * P.nmembers;
    const dim3 grid((unsigned)P.nchunks, (unsigned)((P.nmembers + 63) / 64)), block(64);
    {
        xh_span sp = xh_span_begin(ctx, "calib_abcd");
        if (!P.split_done) {
            hipLaunchKernelGGL(k_calib_split, dim3(256, (unsigned)P.nbasins), dim3(256), 0, ctx->stream, P.d_basins, P.nmonths);
            P.split_done = true;
        }
        hipLaunchKernelGGL(k_calib_march_m<true>, grid, block, 0, ctx->stream, P.d_basins, P.d_chunk_basin, d_active,
                           P.spinup, P.nmembers, P.npar, d_pars, (const double *)nullptr, (const double *)nullptr,
                           P.d_dec, P.d_cnt, (double *)nullptr);
        hipLaunchKernelGGL(k_calib_init_m, dim3((unsigned)((nbm + 63) / 64)), dim3(64), 0, ctx->stream, P.d_basins,
                           d_active, P.nbasins, P.nmembers, P.d_dec, P.d_cnt, P.d_sm0, P.d_gw0);
        hipLaunchKernelGGL(k_calib_march_m<false>, grid, block, 0, ctx->stream, P.d_basins, P.d_chunk_basin, d_active,
                           P.nmonths, P.nmembers, P.npar, d_pars, P.d_sm0, P.d_gw0, (double *)nullptr, (int *)nullptr,
                           P.d_part);
        xh_span_end(sp);
    }
    {
        xh_span sp = xh_span_begin(ctx, "calib_kge");
        const int64_t n = (int64_t)nbm * P.nmonths;
        hipLaunchKernelGGL(k_calib_series_m, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P.d_basins,
                           d_active, P.nbasins, P.nmembers, P.nmonths, P.d_part, P.d_series_m);
        hipLaunchKernelGGL(k_calib_kge_m, dim3((unsigned)((P.nmembers + 63) / 64), (unsigned)P.nbasins), dim3(64), 0,
                           ctx->stream, d_active, P.nbasins, P.nmonths, P.nmembers, P.d_series_m, P.d_obs, d_ed);
        xh_span_end(sp);
    }
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

int xh_calib_series_out(xh_ctx *ctx, const xh_calib_problem &P) {
    if (!P.member_lanes) return XH_OK;
    const int64_t n = (int64_t)P.nbasins * P.nmembers * P.nmonths;
    hipLaunchKernelGGL(k_calib_series_out, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P.nbasins,
                       P.nmembers, P.nmonths, P.d_series_m, P.d_series);
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

int xh_calib_enqueue(xh_ctx *ctx, const xh_calib_problem &P, const double *d_pars, const int *d_active, double *d_ed) {
    if (P.member_lanes) return calib_enqueue_m(ctx, P, d_pars, d_active, d_ed);
    const int nmblocks = (P.nmembers + MB - 1) / MB;
    const size_t nbm = (size_t)P.nbasins * P.nmembers;
    const dim3 grid((unsigned)P.nchunks, (unsigned)nmblocks), block(64);
    {
        xh_span sp = xh_span_begin(ctx, "calib_abcd");
        hipLaunchKernelGGL(k_calib_march<true>, grid, block, 0, ctx->stream, P.d_basins, P.d_chunk_basin, d_active,
                           P.spinup, P.nmembers, P.npar, d_pars, (const double *)nullptr, (const double *)nullptr,
                           P.d_dec, P.d_cnt, (double *)nullptr);
        hipLaunchKernelGGL(k_calib_init, dim3((unsigned)((nbm + 63) / 64)), dim3(64), 0, ctx->stream, P.d_basins,
                           d_active, P.nbasins, P.nmembers, P.d_dec, P.d_cnt, P.d_sm0, P.d_gw0);
        hipLaunchKernelGGL(k_calib_march<false>, grid, block, 0, ctx->stream, P.d_basins, P.d_chunk_basin, d_active,
                           P.nmonths, P.nmembers, P.npar, d_pars, P.d_sm0, P.d_gw0, (double *)nullptr, (int *)nullptr,
                           P.d_part);
        xh_span_end(sp);
    }
    {
        xh_span sp = xh_span_begin(ctx, "calib_kge");
        const int64_t n = (int64_t)nbm * P.nmonths;
        hipLaunchKernelGGL(k_calib_series, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P.d_basins,
                           d_active, P.nbasins, P.nmembers, P.nmonths, P.d_part, P.d_series);
        hipLaunchKernelGGL(k_calib_kge, dim3((unsigned)nbm), dim3(256), 0, ctx->stream, d_active, P.nmonths,
                           P.nmembers, P.d_series, P.d_obs, d_ed);
        xh_span_end(sp);
    }
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}

extern "C" int xh_calib_objective_multi(xh_ctx *ctx, int32_t nbasins, const int64_t *h_ncell, int32_t nmonths,
                                        int32_t spinup, int32_t nmembers, int32_t npar, const double *h_pars,
                                        const double *const *h_pet_t, const double *const *h_precip_t,
                                        const double *const *h_tmin_t, const double *const *h_area,
                                        const double *h_obs, double *h_ed, double *h_series) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, h_pars && h_obs && h_ed, "xh_calib_objective: NULL argument");
    std::vector<xh_calib_basin> basins;
    std::vector<int> chunk_basin;
    size_t bytes = 0;
    int ml = 0;
    int rc = xh_calib_problem_plan(ctx, nbasins, h_ncell, nmonths, spinup, nmembers, npar, h_pet_t, h_precip_t, h_tmin_t,
                                   h_area, basins, chunk_basin, &bytes, &ml);
    if (rc) return rc;
    const size_t nbm = (size_t)nbasins * nmembers;
    const size_t io_bytes = ((nbm * npar + nbm) * sizeof(double) + 255) & ~size_t(255);
    void *buf = nullptr;
    rc = xh_scratch(ctx, 1, io_bytes + bytes, &buf);
    if (rc) return rc;
    double *d_pars = static_cast<double *>(buf);
    double *d_ed = d_pars + nbm * npar;
    xh_calib_problem P;
    XH_HIP(ctx, hipMemcpyAsync(d_pars, h_pars, sizeof(double) * nbm * npar, hipMemcpyHostToDevice, ctx->stream));
    rc = xh_calib_problem_place(ctx, P, nmonths, spinup, nmembers, npar, basins, chunk_basin, h_obs,
                                static_cast<char *>(buf) + io_bytes, ml);
    if (rc) return rc;
    rc = xh_calib_enqueue(ctx, P, d_pars, nullptr, d_ed);
    if (rc) return rc;
    if (h_series) {
        rc = xh_calib_series_out(ctx, P);
        if (rc) return rc;
    }
    XH_HIP(ctx, hipMemcpyAsync(h_ed, d_ed, sizeof(double) * nbm, hipMemcpyDeviceToHost, ctx->stream));
    if (h_series)
        XH_HIP(ctx, hipMemcpyAsync(h_series, P.d_series, sizeof(double) * nbm * nmonths, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // result buffers are the caller's
    return XH_OK;
}

extern "C" int xh_calib_objective(xh_ctx *ctx, int64_t ncell_b, int32_t nmonths, int32_t spinup, int32_t nmembers,
                                  int32_t npar, const double *h_pars, const double *d_pet_t, const double *d_precip_t,
                                  const double *d_tmin_t, const double *d_area, const double *h_obs, double *h_ed,
                                  double *h_series) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, (npar == 5) == (d_tmin_t != nullptr), "xh_calib_objective: npar = 5 requires tmin and vice versa");
    const double *pet[1] = {d_pet_t}, *pr[1] = {d_precip_t}, *tn[1] = {d_tmin_t}, *ar[1] = {d_area};
    return xh_calib_objective_multi(ctx, 1, &ncell_b, nmonths, spinup, nmembers, npar, h_pars, pet, pr,
                                    d_tmin_t ? tn : nullptr, d_area ? ar : nullptr, h_obs, h_ed, h_series);
}
