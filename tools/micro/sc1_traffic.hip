// Counter calibration for the routing streams: move a KNOWN number of bytes with exactly the
// access shapes of xh_mrtm_wave.hip's streams -- raw-buffer 16-byte stores / loads with the agent-coherent policy (sc1),
// 8 lanes x 16 B = one 128-byte line per outlet and block, one wave per "unit" -- and let rocprofv3 count:
//   rocprofv3 --pmc WRITE_SIZE -d out_w -- ./sc1_traffic.bin <MiB>      rocprofv3 --pmc FETCH_SIZE -d out_r -- ./sc1_traffic.bin <MiB>
// (separate passes).  tools/pmc_to_json.py divides the counters of k_sc1_store / k_sc1_load by the byte count printed
// here to get the factor it then applies to the routing kernel's own stream traffic.
// hipcc --offload-arch=gfx950 -O3 -o sc1_traffic.bin sc1_traffic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;

// every wave owns `lines_per_wave` consecutive groups of 8 lines; lane (k = lane / 8, i = lane % 8) touches 16-byte
// chunk i of line k of the group: eight whole 128-byte lines per instruction, as the block transfers do
__global__ void __launch_bounds__(64) k_sc1_store(char *buf, unsigned bytes, int groups_per_wave) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    const unsigned lane = threadIdx.x, base = blockIdx.x * (unsigned)groups_per_wave * 1024u + lane * 16u;
    for (int g = 0; g < groups_per_wave; ++g)
        __builtin_amdgcn_raw_buffer_store_b128(v4u{lane, (unsigned)g, blockIdx.x, 7u}, r, base + (unsigned)g * 1024u, 0, AUX_SC1);
}
__global__ void __launch_bounds__(64) k_sc1_load(char *buf, unsigned bytes, int groups_per_wave, unsigned *sink) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    const unsigned lane = threadIdx.x, base = blockIdx.x * (unsigned)groups_per_wave * 1024u + lane * 16u;
    unsigned acc = 0;
    for (int g = 0; g < groups_per_wave; ++g) {
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, base + (unsigned)g * 1024u, 0, AUX_SC1);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// the same bytes with ordinary cached accesses, for comparison
__global__ void __launch_bounds__(64) k_plain_store(v4u *buf, int groups_per_wave) {
    const size_t base = (size_t)blockIdx.x * groups_per_wave * 64 + threadIdx.x;
    for (int g = 0; g < groups_per_wave; ++g) buf[base + (size_t)g * 64] = v4u{threadIdx.x, (unsigned)g, blockIdx.x, 7u};
}
__global__ void __launch_bounds__(64) k_plain_load(const v4u *buf, int groups_per_wave, unsigned *sink) {
    const size_t base = (size_t)blockIdx.x * groups_per_wave * 64 + threadIdx.x;
    unsigned acc = 0;
    for (int g = 0; g < groups_per_wave; ++g) {
        const v4u v = buf[base + (size_t)g * 64];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 1024;          // working set; > 256 MiB = past the Infinity Cache
    const int waves = 2048;
    const int groups = (int)(mib * 1024 * 1024 / 1024 / waves);
    const size_t bytes = (size_t)waves * groups * 1024;
    char *buf;
    unsigned *sink;
    (void)hipMalloc(&buf, bytes);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_sc1_store, dim3(waves), dim3(64), 0, 0, buf, (unsigned)bytes, groups);
        hipLaunchKernelGGL(k_sc1_load, dim3(waves), dim3(64), 0, 0, buf, (unsigned)bytes, groups, sink);
        hipLaunchKernelGGL(k_plain_store, dim3(waves), dim3(64), 0, 0, reinterpret_cast<v4u *>(buf), groups);
        hipLaunchKernelGGL(k_plain_load, dim3(waves), dim3(64), 0, 0, reinterpret_cast<const v4u *>(buf), groups, sink);
    }
    (void)hipDeviceSynchronize();
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
