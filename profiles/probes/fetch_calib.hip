// Calibration of rocprofv3's FETCH_SIZE for RANDOM accesses of a known size (the guide says the counter tallies a 128-B request
// of a wide coalesced stream at 64 B; the seeding kernels read random 64-byte Occ blocks -- what does it report for those, and does a random 128-B
// access cost more than a random 64-B one?).  One configuration per invocation, so that under `rocprofv3 --pmc FETCH_SIZE` the second dispatch
// (the first warms up) is the measured one:
//   ./fetch_calib <footprint GB> <bytes per access: 16 | 64 | 128 | 256> <waves per CU> <iterations> <dependent 0|1>
// Every lane reads `bytes` contiguous bytes at a random `bytes`-aligned place of the buffer per iteration and consumes all of them.
// Prints the bytes requested by the timed dispatch, its duration, accesses/s and GB/s.
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/fetch_calib.hip -o profiles/probes/fetch_calib_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int Q>      // Q = uint4s per access
__global__ void __launch_bounds__(64)
k_calib(const uint4 *__restrict__ buf, uint64_t n_units, int iters, int dependent, uint32_t *sink)
{
    uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
        const uint4 *p = buf + (x % n_units) * Q;
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < Q; q++) { const uint4 a = p[q]; v ^= a.x ^ a.y ^ a.z ^ a.w; }
        acc += v;
        if (dependent) x += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char **argv)
{
    if (argc < 6) { printf("usage: fetch_calib <GB> <bytes per access> <waves per CU> <iters> <dependent>\n"); return 2; }
    const double gb = atof(argv[1]); const int bytes_per = atoi(argv[2]), wpc = atoi(argv[3]), iters = atoi(argv[4]), dep = atoi(argv[5]);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount, blocks = n_cu * wpc;
    const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
    void *buf; uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const uint64_t n_units = bytes / (uint64_t)bytes_per;
    auto launch = [&](int n) {
        if (bytes_per == 16) k_calib<1><<<blocks, 64>>>((const uint4 *)buf, n_units, n, dep, sink);
        else if (bytes_per == 64) k_calib<4><<<blocks, 64>>>((const uint4 *)buf, n_units, n, dep, sink);
        else if (bytes_per == 128) k_calib<8><<<blocks, 64>>>((const uint4 *)buf, n_units, n, dep, sink);
        else k_calib<16><<<blocks, 64>>>((const uint4 *)buf, n_units, n, dep, sink);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(8);
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double acc = (double)blocks * 64 * iters;
    printf("footprint %.2f GB, %d B per access, %d waves/CU, %s: requested %.0f bytes in %.3f ms = %.1f G accesses/s = %.1f GB/s\n", gb, bytes_per, wpc,
           dep ? "dependent" : "independent", acc * bytes_per, ms, acc / ms / 1e6, acc * bytes_per / ms / 1e6);
    return 0;
}
