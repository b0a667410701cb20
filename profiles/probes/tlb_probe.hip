// Random 64-byte-line reads over buffers of growing size: does the rate (throughput) or the time per dependent
// load (latency) fall off once the footprint exceeds the TLB reach?   hipcc -O3 --offload-arch=gfx950 tlb_probe.hip -o tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void __launch_bounds__(64)
k_probe(const uint4 *__restrict__ buf, uint64_t n_lines, int iters, int dependent, uint32_t *sink)
{
    uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
        const uint4 a = buf[(x % n_lines) << 2];
        const uint32_t v = a.x ^ a.y ^ a.z ^ a.w;
        acc += v;
        if (dependent) x += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char **argv)
{
    int n_cu = 256;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); n_cu = pr.multiProcessorCount;
    uint32_t *sink; hipMalloc(&sink, 4);
    // default: the round-1 sweep; with arguments: the footprints in GB given on the command line (e.g. 69 119: the K=16 prefix table, table + full SA)
    double sizes_gb[16] = {0.0625, 0.25, 1, 2, 4, 8, 16, 32};
    int n_sizes = 8;
    if (argc > 1) { n_sizes = 0; for (int i = 1; i < argc && n_sizes < 16; i++) sizes_gb[n_sizes++] = atof(argv[i]); }
    for (int si = 0; si < n_sizes; si++) {
        const double gb = sizes_gb[si];
        const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
        void *buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("%.2f GB: alloc failed\n", gb); continue; }
        hipMemset(buf, 1, bytes);
        for (int dep = 0; dep < 2; dep++) for (int wpc : {1, 2, 4, 8, 16}) {
            const int blocks = n_cu * wpc, iters = 400;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k_probe<<<blocks, 64>>>((const uint4 *)buf, bytes / 64, 8, dep, sink);
            hipEventRecord(e0);
            k_probe<<<blocks, 64>>>((const uint4 *)buf, bytes / 64, iters, dep, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double loads = (double)blocks * 64 * iters;
            printf("%6.2f GB  %s  %2d waves/CU: %6.1f G lines/s   %7.1f ns per dependent step\n", gb, dep ? "dependent  " : "independent", wpc, loads / ms / 1e6, ms * 1e6 / iters);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        hipFree(buf);
    }
    return 0;
}
