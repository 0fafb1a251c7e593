// Micro-benchmark: where the cycles of one routing sub-step go for ONE wave alone on a SIMD (gfx950).
// Variants add the pieces of xh_mrtm_skew.hip's sub-step (P = 3) one at a time.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o substep_cost.bin substep_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_d2;

template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, double tauinv, double dt, double dtinv, double erl, const int *perm, int nrare) {
    __shared__ __attribute__((aligned(16))) v2d lds[8 * 129];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8 * 129; i += 64) lds[i] = v2d{1e-3 * i, 1e-3 * i};
    lds_d2 *own = (lds_d2 *)lds + lane;
    lds_d2 *e[6];
    for (int w = 0; w < 6; ++w) e[w] = (lds_d2 *)lds + perm[w * 64 + lane];
    double S = 1.0 + lane, favg = 0.0;
    v2d a[3], b[3];
    for (int w = 0; w < 3; ++w) a[w] = b[w] = v2d{0.001, 0.001};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 2048; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v2d an[3], bn[3];
            if (MODE >= 3) {
                if (MODE != 6) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    if (MODE == 3 || MODE == 6 || w == 0) {
                        an[w] = e[w][((j + 7) & 7) * 129]; bn[w] = e[3 + w][((j + 7) & 7) * 129];
                    } else if (MODE == 4) {          // rare terms: only a few lanes read them
                        an[w] = bn[w] = v2d{0.0, 0.0};
                        if (lane < nrare) { an[w] = e[w][((j + 7) & 7) * 129]; bn[w] = e[3 + w][((j + 7) & 7) * 129]; }
                    } else if (MODE == 5) {          // 8-byte reads instead of 16
                        an[w].x = an[w].y = ((__attribute__((address_space(3))) double *)e[w])[((j + 7) & 7) * 258];
                        bn[w].x = bn[w].y = ((__attribute__((address_space(3))) double *)e[3 + w])[((j + 7) & 7) * 258];
                    }
                }
                if (MODE != 6) __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 6) {
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU
                }
            }
            const double F0 = S * tauinv;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int w = 0; w < 3; ++w) { s1 += a[w].x; s2 += a[w].y; }
            s1 -= F0;
#pragma unroll
            for (int w = 0; w < 3; ++w) s1 += b[w].x;
            const double dsdt = s1 + erl;
            double f2, Sn;
            if (MODE >= 1) {
                const bool sx = (dsdt * dt) < (-S);
                f2 = sx ? (dsdt + F0) + S * dtinv : F0;
                if (MODE >= 2) own[(j & 7) * 129] = v2d{F0, f2};
                s2 -= f2;
#pragma unroll
                for (int w = 0; w < 3; ++w) s2 += b[w].y;
                const double dsdt2 = s2 + erl;
                Sn = sx ? 0.0 : S + dsdt2 * dt;
            } else {
                f2 = (dsdt + F0) + S * dtinv;
                s2 -= f2;
#pragma unroll
                for (int w = 0; w < 3; ++w) s2 += b[w].y;
                const double dsdt2 = s2 + erl;
                Sn = S + dsdt2 * dt;
            }
            S = Sn;
            favg += f2;
            if (MODE >= 3) {
#pragma unroll
                for (int w = 0; w < 3; ++w) { a[w] = an[w]; b[w] = bn[w]; }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = S + favg;
    if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; unsigned long long *cyc, h; int *perm, hp[6 * 64];
    for (int i = 0; i < 6 * 64; ++i) hp[i] = (i * 37 + 11) % 128;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&perm, sizeof(hp));
    (void)hipMemcpy(perm, hp, sizeof(hp), hipMemcpyHostToDevice);
    const char *names[] = {"arithmetic only (20 fp64 ops)", "+ compare and 4 v_cndmask", "+ ds_write_b128", "+ 6 ds_read_b128 a sub-step ahead", "2 full ds_read_b128 + 4 on `nrare` lanes", "2 ds_read_b128 + 4 ds_read_b64", "6 ds_read_b128 interleaved with the VALU work"};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1e-4, 10800.0, 1.0 / 10800.0, 1e-3, perm, nrare); (void)hipDeviceSynchronize(); \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.1f cycles per sub-step\n", names[M], (double)h / (2048.0 * 8));
    int nrare = 0;
    RUN(0) RUN(1) RUN(2) RUN(3)
    for (nrare = 1; nrare <= 64; nrare *= 4) { printf("nrare=%d: ", nrare); RUN(4) }
    RUN(5)
    RUN(6)
    return 0;
}
