// Micro-benchmark (round 3): cost of a routing sub-step for a "plain" unit -- no cell whose flow can be adjusted feeds
// any cell of the unit, so one value per gathered term (8 bytes) and ONE row sum are enough -- against the pair form
// of xh_mrtm_skew.hip, alone on the device and with one wave on every SIMD of every CU (LDS shared by four waves).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o substep_plain.bin substep_plain.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_d2;
typedef __attribute__((address_space(3))) double lds_d;
typedef __attribute__((address_space(3))) char lds_c;

constexpr int SLOT_PAIRS = 129;        // pair form: 64 cells + 64 ghosts + zero
constexpr int RING = 8;

// MODE 0: pair form, PRE pre + POST post pair reads (ds_read_b128), two sums, ds_write_b128 of the own pair
// MODE 1: plain form, inbox of NS doubles per lane read as ceil(NS / 2) ds_read_b128, one sum, ds_write_b64 into the
//         consumer's inbox
// MODE 2: plain form with NS separate ds_read_b64
// MODE 3: MODE 1 + scalar guard (s_or of the fired mask)
template <int MODE, int PRE, int POST>
__global__ void __launch_bounds__(256) k(double *out, unsigned long long *cyc, double tauinv, double dt, double dtinv,
                                         double erl, const int *perm, int iters) {
    constexpr int NS = PRE + POST;
    constexpr int NQ = (NS + 1) / 2;
    constexpr int INBOX = (NQ | 1) * 16;                     // bytes per lane, an odd number of 16-byte chunks
    constexpr int SLOTB = MODE == 0 ? SLOT_PAIRS * 16 : 64 * INBOX;
    __shared__ __attribute__((aligned(16))) char lds_all[4][RING * SLOTB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_c *lds = (lds_c *)lds_all[wave];
    for (int i = lane; i < RING * SLOTB / 8; i += 64) ((lds_d *)lds)[i] = 1e-3 * (i & 127);
    double S = 1.0 + lane, favg = 0.0;
    unsigned long long fired = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        lds_d2 *own = (lds_d2 *)lds + lane;
        lds_d2 *ea[PRE], *eb[POST];
        for (int w = 0; w < PRE; ++w) ea[w] = (lds_d2 *)lds + perm[w * 64 + lane];
        for (int w = 0; w < POST; ++w) eb[w] = (lds_d2 *)lds + perm[(4 + w) * 64 + lane];
        v2d a[PRE], b[POST];
        for (int w = 0; w < PRE; ++w) a[w] = v2d{0.001, 0.001};
        for (int w = 0; w < POST; ++w) b[w] = v2d{0.001, 0.001};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v2d an[PRE], bn[POST];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < PRE; ++w) an[w] = ea[w][((j + 7) & 7) * SLOT_PAIRS];
#pragma unroll
                for (int w = 0; w < POST; ++w) bn[w] = eb[w][((j + 7) & 7) * SLOT_PAIRS];
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F | ((PRE + POST + 1) << 8));
                __builtin_amdgcn_sched_barrier(0);
                const double F0 = S * tauinv;
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int w = 0; w < PRE; ++w) { s1 += a[w].x; s2 += a[w].y; }
                s1 -= F0;
#pragma unroll
                for (int w = 0; w < POST; ++w) s1 += b[w].x;
                const double dsdt = s1 + erl;
                const bool sx = (dsdt * dt) < (-S);
                const double f2 = sx ? (dsdt + F0) + S * dtinv : F0;
                own[(j & 7) * SLOT_PAIRS] = v2d{F0, f2};
                s2 -= f2;
#pragma unroll
                for (int w = 0; w < POST; ++w) s2 += b[w].y;
                const double dsdt2 = s2 + erl;
                double Sn = S + dsdt2 * dt;
                asm volatile("" : "+v"(Sn));
                S = sx ? 0.0 : Sn;
                favg += f2;
#pragma unroll
                for (int w = 0; w < PRE; ++w) a[w] = an[w];
#pragma unroll
                for (int w = 0; w < POST; ++w) b[w] = bn[w];
            }
        }
    } else {
        // inbox of this lane: [pre terms, padded to the front][post terms, padded to the back]; the lane's own flow goes
        // into the inbox of its consumer (some other lane, position perm % NS)
        lds_c *inbox = lds + lane * INBOX;
        const int cons = perm[lane] & 63, pos = perm[64 + lane] % NS;
        lds_d *dst = (lds_d *)(lds + cons * INBOX + pos * 8);
        double v[2 * NQ], vn[2 * NQ];
        for (int w = 0; w < 2 * NQ; ++w) v[w] = 0.001;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 2) {
#pragma unroll
                    for (int w = 0; w < NS; ++w) vn[w] = *(lds_d *)(inbox + ((j + 7) & 7) * SLOTB + w * 8);
                } else {
#pragma unroll
                    for (int w = 0; w < NQ; ++w) {
                        const v2d t = *(lds_d2 *)(inbox + ((j + 7) & 7) * SLOTB + w * 16);
                        vn[2 * w] = t.x;
                        vn[2 * w + 1] = t.y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F | (((MODE == 2 ? NS : NQ) + 1) << 8));
                __builtin_amdgcn_sched_barrier(0);
                const double F0 = S * tauinv;
                double s1 = 0.0;
#pragma unroll
                for (int w = 0; w < PRE; ++w) s1 += v[w];
                s1 -= F0;
#pragma unroll
                for (int w = 0; w < POST; ++w) s1 += v[PRE + w];
                const double dsdt = s1 + erl;
                const double d = dsdt * dt;
                const bool sx = d < (-S);
                const double f2 = sx ? (dsdt + F0) + S * dtinv : F0;
                if (MODE == 3) fired |= __builtin_amdgcn_ballot_w64(sx);
                dst[(j & 7) * (SLOTB / 8)] = f2;
                double Sn = S + d;
                asm volatile("" : "+v"(Sn));
                S = sx ? 0.0 : Sn;
                favg += f2;
#pragma unroll
                for (int w = 0; w < 2 * NQ; ++w) v[w] = vn[w];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = S + favg + (double)(fired & 1);
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
#define RUN(M, P, Q, name)                                                                                            \
    for (int full = 0; full < 2; ++full) {                                                                            \
        hipLaunchKernelGGL((k<M, P, Q>), dim3(full ? 256 : 1), dim3(full ? 256 : 64), 0, 0, out, cyc, 1e-4, 10800.0,  \
                           1.0 / 10800.0, 1e-3, perm, iters);                                                         \
        (void)hipDeviceSynchronize();                                                                                 \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                           \
        printf("%-58s %s %.1f cycles per sub-step\n", name, full ? "4 waves on each of 256 CUs" : "one wave alone           ", \
               (double)h / (iters * 8.0));                                                                            \
    }
    RUN(0, 2, 4, "pairs (2,4): 6 ds_read_b128, two sums")
    RUN(0, 2, 3, "pairs (2,3): 5 ds_read_b128, two sums")
    RUN(0, 2, 2, "pairs (2,2): 4 ds_read_b128, two sums")
    RUN(0, 1, 2, "pairs (1,2): 3 ds_read_b128, two sums")
    RUN(0, 1, 1, "pairs (1,1): 2 ds_read_b128, two sums")
    RUN(1, 4, 4, "plain (4,4): 4 ds_read_b128 of 8 values, one sum")
    RUN(1, 2, 4, "plain (2,4): 3 ds_read_b128 of 6 values, one sum")
    RUN(1, 2, 2, "plain (2,2): 2 ds_read_b128 of 4 values, one sum")
    RUN(1, 1, 1, "plain (1,1): 1 ds_read_b128 of 2 values, one sum")
    RUN(2, 2, 4, "plain (2,4): 6 ds_read_b64, one sum")
    RUN(2, 2, 2, "plain (2,2): 4 ds_read_b64, one sum")
    RUN(3, 2, 4, "plain (2,4) + scalar guard on the fired mask")
    return 0;
}
