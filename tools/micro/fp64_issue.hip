// Micro-benchmark: issue cost of fp64 VALU for ONE wave on a SIMD (gfx950): dependent chain vs independent streams.
// hipcc --offload-arch=gfx950 -O3 -o fp64_issue fp64_issue.hip && ./fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64
template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a - threadIdx.x, x3 = b * 2 + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) {            // 4 dependent adds
                x0 += b; x0 += a; x0 += b; x0 += a;
            } else if (MODE == 1) {     // 4 adds, 2 independent chains
                x0 += b; x1 += a; x0 += a; x1 += b;
            } else if (MODE == 2) {     // 4 adds, 4 independent chains
                x0 += b; x1 += a; x2 += b; x3 += a;
            } else if (MODE == 3) {     // 4 dependent muls
                x0 *= b; x0 *= a; x0 *= b; x0 *= a;
            } else if (MODE == 4) {     // dependent add alternating with 32-bit cndmask-like int op on another reg
                x0 += b; asm volatile("v_add_u32 %0, %0, 1" : "+v"(((int *)&x1)[0])); x0 += a; asm volatile("v_add_u32 %0, %0, 1" : "+v"(((int *)&x1)[0]));
            } else if (MODE == 5) {     // 4 dependent fp32 adds (reference)
                float f = (float)x0; f += (float)b; f += (float)a; f += (float)b; f += (float)a; x0 = f;
            }
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const char *names[] = {"4 dependent v_add_f64", "2 chains x 2 v_add_f64", "4 independent v_add_f64", "4 dependent v_mul_f64",
                           "2 dependent v_add_f64 + 2 v_add_u32", "cvt + 4 dependent v_add_f32 + cvt"};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.000001, 0.999999); hipDeviceSynchronize(); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.2f cycles per group of 4\n", names[M], (double)h / (256.0 * REP));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
