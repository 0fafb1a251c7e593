// Micro-test: global_load_lds_dword (gfx950) -- does lane L's dword land at LDS address M0 + inst_offset + 4 L, and is it
// there after s_waitcnt vmcnt(0)?  (Measured: LDS address = M0 + inst_offset + 4 L -- the instruction offset counts on the LDS side too.)  The routing kernel uses it to fetch next month's runoff without a register in flight.
// hipcc --offload-arch=gfx950 -O3 -o lds_dma.bin lds_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *p, double *o, int stride, unsigned *dbg) {
    __shared__ unsigned stage[256];
    for (int i = threadIdx.x; i < 256; i += 64) stage[i] = 0xdeadbeefu;
    __syncthreads();
    const double *a = p + (size_t)threadIdx.x * stride + 3;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)stage + 64u;      // not 0: tests the base
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_add_u32 m0, m0, 252\n\t"
                 "global_load_lds_dword %1, off offset:4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(a), "s"(base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned lo = stage[16 + threadIdx.x], hi = stage[16 + 64 + threadIdx.x];
    o[threadIdx.x] = __hiloint2double((int)hi, (int)lo);
    for (int i = threadIdx.x; i < 256; i += 64) dbg[i] = stage[i];
}
int main() {
    const int stride = 600;
    double *h = new double[64 * stride], *d, *o, r[64];
    for (int i = 0; i < 64 * stride; ++i) h[i] = 1000.0 * (i / stride) + (i % stride) + 0.25;
    (void)hipMalloc(&d, sizeof(double) * 64 * stride);
    (void)hipMalloc(&o, sizeof(double) * 64);
    (void)hipMemcpy(d, h, sizeof(double) * 64 * stride, hipMemcpyHostToDevice);
    unsigned *dbg, hd[256];
    (void)hipMalloc(&dbg, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, stride, dbg);
    (void)hipMemcpy(hd, dbg, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 160; ++i) printf("%08x%c", hd[i], i % 8 == 7 ? '\n' : ' ');
    (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += r[l] != 1000.0 * l + 3.25;
    printf("global_load_lds_dword: %d of 64 lanes wrong (lane 5: %.2f, expected %.2f)\n", bad, r[5], 5003.25);
    return bad != 0;
}
