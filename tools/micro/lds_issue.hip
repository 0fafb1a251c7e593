// Micro-benchmark: does the issue cost of ds_read_b128 (one wave alone) overlap with VALU work placed between reads?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_issue.bin lds_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

#define READ(r) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr))
#define ADDS(k) { _Pragma("unroll") for (int q = 0; q < k; ++q) { x += b; asm volatile("" : "+v"(x)); } }

template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, double b) {
    __shared__ __attribute__((aligned(16))) v4f lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = v4f{1, 2, 3, 4};
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) v4f *)lds + ((lane * 37 + 11) % 128) * 16;
    double x = lane;
    v4f r0, r1, r2, r3, r4, r5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 4096; ++it) {
        if (MODE == 0) { ADDS(24) }
        if (MODE == 1) { READ(r0); READ(r1); READ(r2); READ(r3); READ(r4); READ(r5); ADDS(24) }
        if (MODE == 2) { READ(r0); ADDS(4) READ(r1); ADDS(4) READ(r2); ADDS(4) READ(r3); ADDS(4) READ(r4); ADDS(4) READ(r5); ADDS(4) }
        if (MODE == 3) { READ(r0); READ(r1); READ(r2); ADDS(24) }
        if (MODE == 4) { READ(r0); ADDS(8) READ(r1); ADDS(8) READ(r2); ADDS(8) }
        if (MODE == 5) { READ(r0); READ(r1); READ(r2); READ(r3); READ(r4); READ(r5); asm volatile("s_waitcnt lgkmcnt(0)"); ADDS(24) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = x + r0.x + r1.x + r2.x + r3.x + r4.x + r5.x;
    if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    const char *names[] = {"24 dependent v_add_f64", "6 ds_read_b128, then 24 adds, wait at the end", "6 x (ds_read_b128 + 4 adds), wait at the end",
                           "3 ds_read_b128, then 24 adds", "3 x (ds_read_b128 + 8 adds)", "6 ds_read_b128, wait, 24 adds"};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001); (void)hipDeviceSynchronize(); \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-50s %.1f cycles per iteration\n", names[M], (double)h / 4096.0);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
