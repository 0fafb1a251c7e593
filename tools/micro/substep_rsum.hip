// Micro-benchmark (round 5): cost of one routing sub-step in the REASSOCIATED form of k_mrtm_rsum (xh_mrtm_wave_unit.h, RSUM):
// two ds_read_b128 (the cell's inflow {sum F, sum F2}, its chain predecessor's running sum), the fused update of mrtm.py:50-69
// (8 fp64 operations, one compare, two selects), one ds_write_b128 -- alone on the device and with one wave on every SIMD of
// every CU (the LDS shared by four waves), next to the bit-exact pair form of a (2,3) row for comparison.  The figures are the
// `floor_cycles` of bench.py's roofline.critical_path.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o substep_rsum.bin substep_rsum.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_d2;
typedef __attribute__((address_space(3))) double lds_d;
typedef __attribute__((address_space(3))) char lds_c;

constexpr int SLOT_PAIRS = 129;        // 64 cells + 64 imported entries + zero
constexpr int RING = 8;

// MODE 0: reassociated form, reads A and R.  MODE 1: reads A only (no chain read).  MODE 2: no reads at all (single-cell networks).
// MODE 3: MODE 0 without the LDS (arithmetic only: what the VALU work alone costs a lone wave).
// MODE 4: the bit-exact pair form of a (2,3) row (5 ds_read_b128, two sums in stored order) -- round 3's reference point.
// MODE 5: MODE 0 with the reads issued TWO sub-steps ahead of their use (a lane lag of three iterations per level instead of two).
// MODE 6 (round 6): the SINGLE-SUM form of a prepared plan (wave_unit<..., SGL = 1>): two ds_read_b64 (the sum of the adjusted
//         flows of the cell's upstream neighbours, its chain predecessor's), S1 = base + A dt, m = min(S1, 0), F2 = F + m / dt,
//         S = S1 - m, one ds_write_b64 -- 8 fp64 operations.  MODE 7: the same arithmetic alone (no LDS).
// MODE 8: a unit none of whose cells may fire (not built: what dropping the clamp would buy): S = S1, F2 = F -- 5 operations.
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, unsigned long long *cyc, double tauinv, double dt, double dtinv, double erl,
                                         const int *perm, int iters) {
    __shared__ __attribute__((aligned(16))) char lds_all[4][RING * SLOT_PAIRS * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_c *lds = (lds_c *)lds_all[wave];
    for (int i = lane; i < RING * SLOT_PAIRS * 2; i += 64) ((lds_d *)lds)[i] = 1e-3 * (i & 127);
    double S = 1.0 + lane, favg = 0.0;
    const double acoef = 1.0 - tauinv * dt, erldt = erl * dt;
    lds_d2 *own = (lds_d2 *)lds + lane;
    lds_d2 *e[5];
    for (int w = 0; w < 5; ++w) e[w] = (lds_d2 *)lds + perm[w * 64 + lane];
    v2d v[5], vn[5], vnn[2];
    for (int w = 0; w < 5; ++w) v[w] = vn[w] = v2d{0.001, 0.001};
    vnn[0] = vnn[1] = v2d{0.001, 0.001};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            constexpr int NR = (MODE == 0 || MODE == 5 || MODE == 6 || MODE == 8) ? 2 : MODE == 1 ? 1 : MODE == 4 ? 5 : 0;
            if (MODE >= 6) {      // 8-byte entries
                lds_d *own1 = (lds_d *)lds + lane;
                __builtin_amdgcn_sched_barrier(0);
                if (MODE != 7) {
                    vn[0].x = ((lds_d *)e[0])[((j + 7) & 7) * SLOT_PAIRS];
                    vn[1].x = ((lds_d *)e[1])[((j + 7) & 7) * SLOT_PAIRS];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (MODE != 7) __builtin_amdgcn_s_waitcnt(0xC07F | (3 << 8));
                __builtin_amdgcn_sched_barrier(0);
                const double F0 = S * tauinv;
                const double base = __builtin_fma(S, acoef, erldt);
                const double S1 = __builtin_fma(v[0].x, dt, base);
                double f2, o;
                if (MODE == 8) {
                    f2 = F0;
                    S = S1;
                } else {
                    const double m = __builtin_fmin(S1, 0.0);
                    f2 = __builtin_fma(m, dtinv, F0);
                    S = S1 - m;
                }
                o = v[1].x + f2;
                if (MODE != 7) own1[(j & 7) * SLOT_PAIRS] = o;
                else asm volatile("" ::"v"(o));
                favg += f2;
                v[0].x = vn[0].x;
                v[1].x = vn[1].x;
                continue;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NR; ++w) (MODE == 5 ? vnn[w & 1] : vn[w]) = e[w][((j + 7) & 7) * SLOT_PAIRS];
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 5) __builtin_amdgcn_s_waitcnt(0xC07F | ((2 * NR + 2) << 8));
            else if (MODE != 3) __builtin_amdgcn_s_waitcnt(0xC07F | ((NR + 1) << 8));
            __builtin_amdgcn_sched_barrier(0);
            const double F0 = S * tauinv;
            if (MODE == 4) {
                double s1 = 0.0, s2 = 0.0;
                s1 += v[0].x; s2 += v[0].y; s1 += v[1].x; s2 += v[1].y;
                s1 -= F0;
                s1 += v[2].x; s1 += v[3].x; s1 += v[4].x;
                const double dsdt = s1 + erl;
                const bool sx = (dsdt * dt) < (-S);
                const double f2 = sx ? (dsdt + F0) + S * dtinv : F0;
                own[(j & 7) * SLOT_PAIRS] = v2d{F0, f2};
                s2 -= f2;
                s2 += v[2].y; s2 += v[3].y; s2 += v[4].y;
                double Sn = S + (s2 + erl) * dt;
                asm volatile("" : "+v"(Sn));
                S = sx ? 0.0 : Sn;
                favg += f2;
            } else {
                const double base = __builtin_fma(S, acoef, erldt);
                const double S1 = MODE == 2 ? base : __builtin_fma(v[0].x, dt, base);
                const double S2 = MODE == 2 ? base : __builtin_fma(v[0].y, dt, base);
                const bool sx = S1 < 0.0;
                const double f2 = __builtin_fma(__builtin_fmin(S1, 0.0), dtinv, F0);
                const v2d o = (MODE == 0 || MODE == 3 || MODE == 5) ? v2d{v[1].x + F0, v[1].y + f2} : v2d{F0, f2};
                if (MODE != 3) own[(j & 7) * SLOT_PAIRS] = o;
                else asm volatile("" ::"v"(o));
                double Sn = S2;
                asm volatile("" : "+v"(Sn));
                S = sx ? 0.0 : Sn;
                favg += f2;
            }
#pragma unroll
            for (int w = 0; w < NR; ++w) v[w] = vn[w];
            if (MODE == 5) {
                vn[0] = vnn[0];
                vn[1] = vnn[1];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = S + favg;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out;
    unsigned long long *cyc, h;
    int *perm, hp[8 * 64];
    for (int i = 0; i < 8 * 64; ++i) hp[i] = (i * 37 + 11) % 128;
    (void)hipMalloc(&out, 256 * 256 * 8);
    (void)hipMalloc(&cyc, 8);
    (void)hipMalloc(&perm, sizeof(hp));
    (void)hipMemcpy(perm, hp, sizeof(hp), hipMemcpyHostToDevice);
    const int iters = 4096;
#define RUN(M, name)                                                                                                     \
    for (int full = 0; full < 2; ++full) {                                                                               \
        hipLaunchKernelGGL((k<M>), dim3(full ? 256 : 1), dim3(full ? 256 : 64), 0, 0, out, cyc, 1e-4, 10800.0, 1.0 / 10800.0, \
                           1e-3, perm, iters);                                                                           \
        (void)hipDeviceSynchronize();                                                                                    \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                              \
        printf("%-66s %s %.1f cycles per sub-step\n", name, full ? "4 waves on each of 256 CUs" : "one wave alone           ",  \
               (double)h / (iters * 8.0));                                                                               \
    }
    RUN(0, "reassociated: 2 ds_read_b128 (inflow, chain), fused update")
    RUN(1, "reassociated: 1 ds_read_b128 (inflow only)")
    RUN(2, "reassociated: no read (single-cell networks)")
    RUN(3, "reassociated: the arithmetic alone (no LDS)")
    RUN(4, "bit-exact pairs (2,3): 5 ds_read_b128, two sums in stored order")
    RUN(5, "reassociated, reads two sub-steps ahead of their use")
    RUN(6, "single sums: 2 ds_read_b64, 8 fp64 operations, ds_write_b64")
    RUN(7, "single sums: the arithmetic alone (no LDS)")
    RUN(8, "single sums without the clamp (a unit that cannot fire; not built)")
    return 0;
}
