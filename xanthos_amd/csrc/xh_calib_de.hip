// Device-side differential evolution for the ABCD calibration (gfx950).
//
// Replaces the scipy.optimize.differential_evolution call of xanthos/calibrate/calibrate_abcd.py:103-112 (SciPy
// defaults: strategy best1bin, popsize x n_parameters members, Latin-hypercube start, dithered mutation in
// (0.5, 1), recombination 0.7, tol 0.01) and the serial basin loop of calibrate_all (:256-262): EVERY basin searches
// at once, in lock-step generations, with populations, trial vectors, energies and convergence flags resident in HBM.
// One generation is
//
//   k_de_trial    one workgroup per basin: arg-min of the energies (first minimum, like np.argmin), the generation's
//                 dither, then per member the best1 mutant (two distinct members other than the candidate), binomial
//                 crossover with one forced gene, re-draw of out-of-bounds genes, scaling to parameter space
//   objective     xh_calib_enqueue (xh_calib.hip) on the trial parameters, skipping converged basins
//   k_de_select   per basin: keep the trial where its energy is <= the member's (SciPy's `updating='deferred'`
//                 semantics, the mode SciPy itself uses for vectorised / parallel objectives), then SciPy's
//                 convergence test std(E) <= atol + tol |mean(E)| (never with an infinite energy in the population)
//
// so a generation costs the host five kernel launches and nothing else: the numpy driver of round 1 spent 1.27 s per
// 512-member generation of 235 basins on the host against 0.078 s of kernels.
//
// Random numbers are counter-based (SplitMix64 of (seed, basin key, generation, member, slot)): a basin's search
// depends only on the seed and its key, not on which other basins share the launch or on which GPU it runs, so a
// basin-sharded multi-GPU calibration returns exactly what one GPU returns.  oracle/de.py restates the same
// generation step in numpy; the tests compare trial vectors bit for bit.
#include <cmath>

#include "xh_calib.h"

struct xh_calib_de {
    xh_ctx *ctx = nullptr;
    xh_calib_problem P;
    void *d_problem = nullptr;       // backing store of P
    int n = 0, d = 0, nb = 0;
    uint64_t seed = 0;
    int32_t generation = 0;          // generations enqueued so far
    bool initialised = false;
    uint64_t *d_key = nullptr;       // [nb] RNG stream of each basin
    double *d_lo = nullptr, *d_hi = nullptr;             // [d]
    double *d_pop = nullptr, *d_trial = nullptr;         // [nb, n, d] unit cube
    double *d_x = nullptr;                               // [nb, n, d] scaled parameters of the vectors being evaluated
    double *d_energy = nullptr, *d_e_trial = nullptr;    // [nb, n]
    int *d_active = nullptr;         // [nb] 1 while the basin is still searching
    int *d_nit = nullptr;            // [nb]
    long long *d_nfev = nullptr;     // [nb]
    int *d_n_active = nullptr;       // [1]
    int *h_n_active = nullptr;       // pinned
};

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// uniform in [0, 1) for (basin key, generation, member, slot); generation -1 is the initial population
__device__ __forceinline__ double de_uniform(uint64_t seed, uint64_t key, int gen, int member, int slot) {
    uint64_t h = splitmix64(seed ^ (key * 0xD1342543DE82EF95ull));
    h = splitmix64(h ^ (uint64_t)(uint32_t)(gen + 1));
    h = splitmix64(h ^ (((uint64_t)(uint32_t)member << 32) | (uint32_t)slot));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// slots of a member's stream
constexpr int SLOT_R0 = 0, SLOT_R1 = 1, SLOT_FILL = 2, SLOT_CROSS = 8, SLOT_REDRAW = 40;   // + gene index
constexpr int SLOT_SCALE = 3;                       // member 0x7fffffff: per-generation dither
constexpr int MEMBER_GEN = 0x7fffffff;

// SciPy's _scale_parameters: 0.5 (lo + hi) + (t - 0.5) |lo - hi|
__device__ __forceinline__ double de_scale(double t, double lo, double hi) {
    return 0.5 * (lo + hi) + (t - 0.5) * fabs(lo - hi);
}

// Latin hypercube start (SciPy init='latinhypercube'): gene j of the members is a random permutation of the n strata,
// jittered inside the stratum.  The permutation is the rank of a random key among the basin's keys.
__global__ void __launch_bounds__(256) k_de_init(int n, int d, uint64_t seed, const uint64_t *__restrict__ keys,
                                                 const double *__restrict__ lo, const double *__restrict__ hi,
                                                 double *__restrict__ pop, double *__restrict__ x) {
    extern __shared__ double sh_key[];               // [n]
    const int b = blockIdx.x;
    const uint64_t key = keys[b];
    for (int j = 0; j < d; ++j) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) sh_key[i] = de_uniform(seed, key, -1, i, SLOT_CROSS + j);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double ki = sh_key[i];
            int rank = 0;
            for (int k = 0; k < n; ++k) {
                const double kk = sh_key[k];
                rank += (kk < ki || (kk == ki && k < i)) ? 1 : 0;
            }
            const double t = ((double)rank + de_uniform(seed, key, -1, i, SLOT_REDRAW + j)) / (double)n;
            const int64_t o = ((int64_t)b * n + i) * d + j;
            pop[o] = t;
            x[o] = de_scale(t, lo[j], hi[j]);
        }
        __syncthreads();
    }
}

// first index of the minimum of e[0..n) (np.argmin; NaN never present: energies are cleaned to +inf)
__device__ __forceinline__ int block_argmin(const double *__restrict__ e, int n, double *sh_v, int *sh_i) {
    double v = INFINITY;
    int at = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = e[i];
        if (x < v || (x == v && i < at)) {
            v = x;
            at = i;
        }
    }
    sh_v[threadIdx.x] = v;
    sh_i[threadIdx.x] = at;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double v2 = sh_v[threadIdx.x + s];
            const int i2 = sh_i[threadIdx.x + s];
            if (v2 < sh_v[threadIdx.x] || (v2 == sh_v[threadIdx.x] && i2 < sh_i[threadIdx.x])) {
                sh_v[threadIdx.x] = v2;
                sh_i[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    const int r = sh_i[0] == 0x7fffffff ? 0 : sh_i[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) k_de_trial(int n, int d, uint64_t seed, int gen, double mut_lo, double mut_hi,
                                                  double recomb, const uint64_t *__restrict__ keys,
                                                  const int *__restrict__ active, const double *__restrict__ lo,
                                                  const double *__restrict__ hi, const double *__restrict__ pop,
                                                  const double *__restrict__ energy, double *__restrict__ trial,
                                                  double *__restrict__ x) {
    __shared__ double sh_v[256];
    __shared__ int sh_i[256];
    const int b = blockIdx.x;
    if (!active[b]) return;
    const uint64_t key = keys[b];
    const double *P = pop + (int64_t)b * n * d;
    const int best = block_argmin(energy + (int64_t)b * n, n, sh_v, sh_i);
    const double scale = mut_lo + de_uniform(seed, key, gen, MEMBER_GEN, SLOT_SCALE) * (mut_hi - mut_lo);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        // two distinct members, both different from the candidate, uniformly (SciPy _select_samples)
        int r0 = (int)(de_uniform(seed, key, gen, i, SLOT_R0) * (double)(n - 1));
        r0 = min(r0, n - 2);
        if (r0 >= i) ++r0;
        int r1 = (int)(de_uniform(seed, key, gen, i, SLOT_R1) * (double)(n - 2));
        r1 = min(r1, n - 3);
        const int s0 = min(i, r0), s1 = max(i, r0);
        if (r1 >= s0) ++r1;
        if (r1 >= s1) ++r1;
        int fill = (int)(de_uniform(seed, key, gen, i, SLOT_FILL) * (double)d);
        fill = min(fill, d - 1);
        for (int j = 0; j < d; ++j) {
            const double mutant = P[(int64_t)best * d + j] + scale * (P[(int64_t)r0 * d + j] - P[(int64_t)r1 * d + j]);
            const bool cross = (de_uniform(seed, key, gen, i, SLOT_CROSS + j) < recomb) || j == fill;
            double t = cross ? mutant : P[(int64_t)i * d + j];
            if (t < 0.0 || t > 1.0 || t != t) t = de_uniform(seed, key, gen, i, SLOT_REDRAW + j);   // _ensure_constraint
            const int64_t o = ((int64_t)b * n + i) * d + j;
            trial[o] = t;
            x[o] = de_scale(t, lo[j], hi[j]);
        }
    }
}

__device__ __forceinline__ double block_sum256(double v, double *sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// first = 1: the energies of the initial population arrive in e_trial (no selection, no convergence test: SciPy
// evolves one generation before the first test)
__global__ void __launch_bounds__(256) k_de_select(int n, int d, int first, double tol, double atol,
                                                   int *__restrict__ active, double *__restrict__ pop,
                                                   const double *__restrict__ trial, double *__restrict__ energy,
                                                   const double *__restrict__ e_trial, int *__restrict__ nit,
                                                   long long *__restrict__ nfev, int *__restrict__ n_active) {
    __shared__ double sh[256];
    const int b = blockIdx.x;
    if (!active[b]) return;
    double *E = energy + (int64_t)b * n;
    const double *ET = e_trial + (int64_t)b * n;
    int n_inf = 0;
    double s1 = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double et = ET[i];
        et = (et == et && fabs(et) != INFINITY) ? et : INFINITY;    // a failed evaluation never wins
        double e = E[i];
        if (first || et <= e) {
            e = et;
            E[i] = et;
            if (!first)
                for (int j = 0; j < d; ++j) {
                    const int64_t o = ((int64_t)b * n + i) * d + j;
                    pop[o] = trial[o];
                }
        }
        n_inf += (e == INFINITY) ? 1 : 0;
        s1 += (e == INFINITY) ? 0.0 : e;
    }
    const double inf_total = block_sum256((double)n_inf, sh);
    const double mean = block_sum256(s1, sh) / (double)n;
    double s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double e = E[i];
        const double dv = (e == INFINITY) ? 0.0 : e - mean;
        s2 += dv * dv;
    }
    const double sd = sqrt(block_sum256(s2, sh) / (double)n);
    if (threadIdx.x == 0) {
        nfev[b] += n;
        if (!first) {
            nit[b] += 1;
            if (inf_total == 0.0 && sd <= atol + tol * fabs(mean)) active[b] = 0;
        }
        if (active[b]) atomicAdd(n_active, 1);
    }
}

// x [nb, d] = scaled best member, fun [nb] = its energy
__global__ void __launch_bounds__(256) k_de_best(int n, int d, const double *__restrict__ lo,
                                                 const double *__restrict__ hi, const double *__restrict__ pop,
                                                 const double *__restrict__ energy, double *__restrict__ x,
                                                 double *__restrict__ fun) {
    __shared__ double sh_v[256];
    __shared__ int sh_i[256];
    const int b = blockIdx.x;
    const int best = block_argmin(energy + (int64_t)b * n, n, sh_v, sh_i);
    if ((int)threadIdx.x < d)
        x[(int64_t)b * d + threadIdx.x] = de_scale(pop[((int64_t)b * n + best) * d + threadIdx.x], lo[threadIdx.x],
                                                   hi[threadIdx.x]);
    if (threadIdx.x == 0) fun[b] = energy[(int64_t)b * n + best];
}

}  // namespace

extern "C" {

void xh_calib_de_destroy(xh_calib_de *de) {
    if (!de) return;
    if (de->ctx) {
        (void)hipSetDevice(de->ctx->device);
        (void)hipStreamSynchronize(de->ctx->stream);
    }
    if (de->d_problem) (void)hipFree(de->d_problem);
    if (de->d_key) (void)hipFree(de->d_key);          // one allocation holds every DE array
    if (de->h_n_active) (void)hipHostFree(de->h_n_active);
    delete de;
}

int xh_calib_de_create(xh_ctx *ctx, int32_t nbasins, const int64_t *h_ncell, const uint64_t *h_basin_key,
                       int32_t nmonths, int32_t spinup, int32_t nmembers, int32_t npar,
                       const double *const *h_pet_t, const double *const *h_precip_t, const double *const *h_tmin_t,
                       const double *const *h_area, const double *h_obs, const double *h_lo, const double *h_hi,
                       uint64_t seed, xh_calib_de **out) {
    if (!ctx) return XH_ERR_ARG;
    XH_REQUIRE(ctx, out && h_obs && h_lo && h_hi, "xh_calib_de_create: NULL argument");
    *out = nullptr;
    XH_REQUIRE(ctx, nmembers >= 4, "xh_calib_de_create: best1bin needs at least 4 members");
    XH_REQUIRE(ctx, nmembers <= 8192, "xh_calib_de_create: at most 8192 members per basin");
    std::vector<xh_calib_basin> basins;
    std::vector<int> chunk_basin;
    size_t bytes = 0;
    int ml = 0;
    int rc = xh_calib_problem_plan(ctx, nbasins, h_ncell, nmonths, spinup, nmembers, npar, h_pet_t, h_precip_t, h_tmin_t,
                                   h_area, basins, chunk_basin, &bytes, &ml);
    if (rc) return rc;
    XH_HIP(ctx, hipSetDevice(ctx->device));
    xh_calib_de *de = new xh_calib_de();
    de->ctx = ctx;
    de->n = nmembers;
    de->d = npar;
    de->nb = nbasins;
    de->seed = seed;
#define DE_TRY(call)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            xh_calib_de_destroy(de);                                                                    \
            return xh_fail(ctx, XH_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                               \
    } while (0)
    DE_TRY(hipMalloc(&de->d_problem, bytes));
    rc = xh_calib_problem_place(ctx, de->P, nmonths, spinup, nmembers, npar, basins, chunk_basin, h_obs, de->d_problem, ml);
    if (rc) {
        xh_calib_de_destroy(de);
        return rc;
    }
    const size_t nb = nbasins, nbn = nb * nmembers, nbnd = nbn * npar;
    // [keys nb u64][nfev nb i64][lo d][hi d][pop][trial][x][energy][e_trial][active nb][nit nb][n_active 1]
    const size_t total = 8 * (2 * nb + 2 * (size_t)npar + 3 * nbnd + 2 * nbn) + 4 * (2 * nb + 16);
    void *buf = nullptr;
    DE_TRY(hipMalloc(&buf, total));
    de->d_key = static_cast<uint64_t *>(buf);
    de->d_nfev = reinterpret_cast<long long *>(de->d_key + nb);
    de->d_lo = reinterpret_cast<double *>(de->d_nfev + nb);
    de->d_hi = de->d_lo + npar;
    de->d_pop = de->d_hi + npar;
    de->d_trial = de->d_pop + nbnd;
    de->d_x = de->d_trial + nbnd;
    de->d_energy = de->d_x + nbnd;
    de->d_e_trial = de->d_energy + nbn;
    de->d_active = reinterpret_cast<int *>(de->d_e_trial + nbn);
    de->d_nit = de->d_active + nb;
    de->d_n_active = de->d_nit + nb;
    DE_TRY(hipHostMalloc(reinterpret_cast<void **>(&de->h_n_active), 64, hipHostMallocDefault));
    *de->h_n_active = nbasins;
    std::vector<uint64_t> keys(nb);
    for (size_t b = 0; b < nb; ++b) keys[b] = h_basin_key ? h_basin_key[b] : (uint64_t)b;
    DE_TRY(hipMemcpyAsync(de->d_key, keys.data(), 8 * nb, hipMemcpyHostToDevice, ctx->stream));
    DE_TRY(hipMemcpyAsync(de->d_lo, h_lo, 8 * (size_t)npar, hipMemcpyHostToDevice, ctx->stream));
    DE_TRY(hipMemcpyAsync(de->d_hi, h_hi, 8 * (size_t)npar, hipMemcpyHostToDevice, ctx->stream));
    DE_TRY(hipMemsetAsync(de->d_nfev, 0, 8 * nb, ctx->stream));
    DE_TRY(hipMemsetAsync(de->d_active, 0, 4 * (2 * nb + 16), ctx->stream));
    DE_TRY(hipStreamSynchronize(ctx->stream));
#undef DE_TRY
    *out = de;
    return XH_OK;
}

static int de_fill_active(xh_calib_de *de, int value) {
    std::vector<int> ones(de->nb, value);
    XH_HIP(de->ctx, hipMemcpyAsync(de->d_active, ones.data(), 4 * (size_t)de->nb, hipMemcpyHostToDevice, de->ctx->stream));
    XH_HIP(de->ctx, hipStreamSynchronize(de->ctx->stream));
    return XH_OK;
}

int xh_calib_de_init(xh_calib_de *de) {
    if (!de) return XH_ERR_ARG;
    xh_ctx *ctx = de->ctx;
    int rc = de_fill_active(de, 1);
    if (rc) return rc;
    XH_HIP(ctx, hipMemsetAsync(de->d_nfev, 0, 8 * (size_t)de->nb, ctx->stream));
    XH_HIP(ctx, hipMemsetAsync(de->d_nit, 0, 4 * (size_t)de->nb, ctx->stream));
    XH_HIP(ctx, hipMemsetAsync(de->d_n_active, 0, 4, ctx->stream));
    de->generation = 0;
    {
        xh_span sp = xh_span_begin(ctx, "calib_de");
        hipLaunchKernelGGL(k_de_init, dim3(de->nb), dim3(256), sizeof(double) * de->n, ctx->stream, de->n, de->d,
                           de->seed, de->d_key, de->d_lo, de->d_hi, de->d_pop, de->d_x);
        xh_span_end(sp);
    }
    rc = xh_calib_enqueue(ctx, de->P, de->d_x, de->d_active, de->d_e_trial);
    if (rc) return rc;
    {
        xh_span sp = xh_span_begin(ctx, "calib_de");
        hipLaunchKernelGGL(k_de_select, dim3(de->nb), dim3(256), 0, ctx->stream, de->n, de->d, 1, 0.0, 0.0,
                           de->d_active, de->d_pop, de->d_trial, de->d_energy, de->d_e_trial, de->d_nit, de->d_nfev,
                           de->d_n_active);
        xh_span_end(sp);
    }
    XH_HIP(ctx, hipGetLastError());
    XH_HIP(ctx, hipMemcpyAsync(de->h_n_active, de->d_n_active, 4, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    de->initialised = true;
    return XH_OK;
}

int xh_calib_de_set_state(xh_calib_de *de, const double *h_pop, const double *h_energy, int32_t generation) {
    if (!de) return XH_ERR_ARG;
    xh_ctx *ctx = de->ctx;
    XH_REQUIRE(ctx, h_pop && h_energy && generation >= 0, "xh_calib_de_set_state: bad argument");
    const size_t nbn = (size_t)de->nb * de->n;
    XH_HIP(ctx, hipMemcpyAsync(de->d_pop, h_pop, 8 * nbn * de->d, hipMemcpyHostToDevice, ctx->stream));
    XH_HIP(ctx, hipMemcpyAsync(de->d_energy, h_energy, 8 * nbn, hipMemcpyHostToDevice, ctx->stream));
    int rc = de_fill_active(de, 1);
    if (rc) return rc;
    de->generation = generation;
    *de->h_n_active = de->nb;
    de->initialised = true;
    return XH_OK;
}

int xh_calib_de_step(xh_calib_de *de, int32_t ngen, double tol, double atol, double mut_lo, double mut_hi,
                     double recombination, int32_t *h_n_active) {
    if (!de) return XH_ERR_ARG;
    xh_ctx *ctx = de->ctx;
    XH_REQUIRE(ctx, de->initialised, "xh_calib_de_step: call xh_calib_de_init first");
    XH_REQUIRE(ctx, ngen >= 0 && mut_lo >= 0.0 && mut_hi >= mut_lo && mut_hi <= 2.0 && recombination >= 0.0 &&
                        recombination <= 1.0 && tol >= 0.0 && atol >= 0.0,
               "xh_calib_de_step: bad argument");
    for (int g = 0; g < ngen; ++g) {
        const int gen = de->generation++;
        xh_span sp = xh_span_begin(ctx, "calib_de");
        hipLaunchKernelGGL(k_de_trial, dim3(de->nb), dim3(256), 0, ctx->stream, de->n, de->d, de->seed, gen, mut_lo,
                           mut_hi, recombination, de->d_key, de->d_active, de->d_lo, de->d_hi, de->d_pop, de->d_energy,
                           de->d_trial, de->d_x);
        xh_span_end(sp);
        int rc = xh_calib_enqueue(ctx, de->P, de->d_x, de->d_active, de->d_e_trial);
        if (rc) return rc;
        XH_HIP(ctx, hipMemsetAsync(de->d_n_active, 0, 4, ctx->stream));
        xh_span sp2 = xh_span_begin(ctx, "calib_de");
        hipLaunchKernelGGL(k_de_select, dim3(de->nb), dim3(256), 0, ctx->stream, de->n, de->d, 0, tol, atol,
                           de->d_active, de->d_pop, de->d_trial, de->d_energy, de->d_e_trial, de->d_nit, de->d_nfev,
                           de->d_n_active);
        xh_span_end(sp2);
    }
    XH_HIP(ctx, hipGetLastError());
    if (ngen > 0) XH_HIP(ctx, hipMemcpyAsync(de->h_n_active, de->d_n_active, 4, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_n_active) *h_n_active = *de->h_n_active;
    return XH_OK;
}

int xh_calib_de_result(xh_calib_de *de, double *h_x, double *h_fun, int64_t *h_nfev, int32_t *h_nit,
                       int32_t *h_active) {
    if (!de) return XH_ERR_ARG;
    xh_ctx *ctx = de->ctx;
    XH_REQUIRE(ctx, de->initialised, "xh_calib_de_result: call xh_calib_de_init first");
    const size_t nb = de->nb;
    void *buf = nullptr;
    int rc = xh_scratch(ctx, 2, 8 * nb * (de->d + 1), &buf);
    if (rc) return rc;
    double *d_x = static_cast<double *>(buf), *d_fun = d_x + nb * de->d;
    hipLaunchKernelGGL(k_de_best, dim3(de->nb), dim3(256), 0, ctx->stream, de->n, de->d, de->d_lo, de->d_hi, de->d_pop,
                       de->d_energy, d_x, d_fun);
    XH_HIP(ctx, hipGetLastError());
    if (h_x) XH_HIP(ctx, hipMemcpyAsync(h_x, d_x, 8 * nb * de->d, hipMemcpyDeviceToHost, ctx->stream));
    if (h_fun) XH_HIP(ctx, hipMemcpyAsync(h_fun, d_fun, 8 * nb, hipMemcpyDeviceToHost, ctx->stream));
    if (h_nfev) XH_HIP(ctx, hipMemcpyAsync(h_nfev, de->d_nfev, 8 * nb, hipMemcpyDeviceToHost, ctx->stream));
    if (h_nit) XH_HIP(ctx, hipMemcpyAsync(h_nit, de->d_nit, 4 * nb, hipMemcpyDeviceToHost, ctx->stream));
    if (h_active) XH_HIP(ctx, hipMemcpyAsync(h_active, de->d_active, 4 * nb, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return XH_OK;
}

int xh_calib_de_state(xh_calib_de *de, int32_t which, double *h_vectors, double *h_energy) {
    if (!de) return XH_ERR_ARG;
    xh_ctx *ctx = de->ctx;
    XH_REQUIRE(ctx, which >= 0 && which <= 2, "xh_calib_de_state: which must be 0 (population), 1 (trial) or 2 (scaled)");
    const size_t nbn = (size_t)de->nb * de->n;
    const double *v = which == 0 ? de->d_pop : (which == 1 ? de->d_trial : de->d_x);
    const double *e = which == 0 ? de->d_energy : de->d_e_trial;
    if (h_vectors) XH_HIP(ctx, hipMemcpyAsync(h_vectors, v, 8 * nbn * de->d, hipMemcpyDeviceToHost, ctx->stream));
    if (h_energy) XH_HIP(ctx, hipMemcpyAsync(h_energy, e, 8 * nbn, hipMemcpyDeviceToHost, ctx->stream));
    XH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return XH_OK;
}

}  // extern "C"
